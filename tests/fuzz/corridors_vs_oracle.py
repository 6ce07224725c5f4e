"""The device corridor stage against the oracle's on random knot-level inputs, at scale: the body of
tests/test_gpu_corridor_pipeline.py::test_device_corridors_equal_oracle_on_random_inputs (6 400 candidates per seed:
horizons of 3-512 knots, 1-64 obstacles, nan / inf / collapsed bounds, both variants; every field of the batch record)
over a range of seeds.

    python tests/fuzz/corridors_vs_oracle.py SEED0 SEEDS       # needs a GPU; ~15 s per seed

Round 3 (first-pass kernels without the serial statement, smaller first-pass lists): 60 seeds = 384 000 candidates,
0 differences.
"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_corridor_pipeline as T
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bad = 0; t0 = time.time()
for s in range(seed0, seed0 + count):
    try:
        T.test_device_corridors_equal_oracle_on_random_inputs.__wrapped__(s) if hasattr(T.test_device_corridors_equal_oracle_on_random_inputs, "__wrapped__") else T.test_device_corridors_equal_oracle_on_random_inputs(s)
    except AssertionError as e:
        bad += 1; print("DIFFERENCE seed", s, str(e)[:300], flush=True)
    except Exception:
        bad += 1; print("ERROR seed", s); traceback.print_exc()
print("seeds", count, "from", seed0, "candidates", count * 6400, "differences", bad, "seconds %.1f" % (time.time() - t0))
