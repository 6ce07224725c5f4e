import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_amd import native
if sys.argv[1] != "base":
    native.LIB_PATH = os.path.abspath(sys.argv[1])
import pipeline_bench
r = pipeline_bench.main(sys.argv[2:])
print("ABL", sys.argv[1], r["corridor_ms"])
