"""Worker of tests/test_multi_rank.py::test_a_failing_rank_ends_the_launch: started by torch.distributed.run; the rank
named by FAIL_RANK raises before the first collective, the others enter it."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_amd import dist as D  # noqa: E402

rank = int(os.environ["RANK"])
out_dir = os.environ["RESULT_DIR"]


def report(text):
    """One file per rank, written in one piece: the ranks share the launcher's stdout / stderr pipes, where the pieces
    of a multi-argument print() of two ranks interleave."""
    with open(os.path.join(out_dir, "rank%d.txt" % rank), "w") as f:
        f.write(text)


D.init_process_group("gloo", timeout_s=int(os.environ.get("COLLECTIVE_TIMEOUT_S", "20")))
if rank == int(os.environ.get("FAIL_RANK", "-1")):
    report("rank %d fails before its first collective\n" % rank)
    raise RuntimeError("rank %d fails before its first collective" % rank)
c, i = D.global_argmin(torch.tensor([float(rank)], dtype=torch.float64), torch.tensor([rank], dtype=torch.int64))
report("rank %d winner %d\n" % (rank, int(i[0])))
torch.distributed.destroy_process_group()
