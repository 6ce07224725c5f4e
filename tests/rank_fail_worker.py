"""Worker of tests/test_multi_rank.py::test_a_failing_rank_ends_the_launch: started by torch.distributed.run; the rank
named by FAIL_RANK raises before the first collective, the others enter it."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_amd import dist as D  # noqa: E402

rank = int(os.environ["RANK"])
D.init_process_group("gloo", timeout_s=int(os.environ.get("COLLECTIVE_TIMEOUT_S", "20")))
if rank == int(os.environ.get("FAIL_RANK", "-1")):
    raise RuntimeError("rank %d fails before its first collective" % rank)
c, i = D.global_argmin(torch.tensor([float(rank)], dtype=torch.float64), torch.tensor([rank], dtype=torch.int64))
print("rank", rank, "winner", int(i[0]), flush=True)
torch.distributed.destroy_process_group()
