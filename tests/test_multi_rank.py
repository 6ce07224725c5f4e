"""N > 1 path on CPU: world_size-2 gloo.  Shards are independent; the only exchange is the
(cost, index) all_gather + local min of spectral_amd/dist.py (bench.py uses the same code)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from spectral_amd.dist import global_argmin, shard_bounds


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, costs, groups, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B = costs.shape[0]
    lo, hi = shard_bounds(B, world, rank)
    local = torch.from_numpy(costs[lo:hi])
    n = hi - lo
    per = n // groups
    bc = torch.empty(groups, dtype=torch.float64); bi = torch.empty(groups, dtype=torch.int64)
    for g in range(groups):  # local arg-min per group, ties -> lowest index, none solved -> -1
        seg = local[g * per:(g + 1) * per]
        j = int(torch.argmin(seg))
        bc[g] = seg[j]; bi[g] = lo + g * per + j if torch.isfinite(seg[j]) else -1
    c, i = global_argmin(bc, bi)
    q.put((rank, c.numpy().copy(), i.numpy().copy()))
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_global_argmin_equals_single_process(world):
    rng = np.random.default_rng(5)
    B, groups = 4096, 4
    costs = rng.normal(size=B) * 1e4
    costs[rng.integers(0, B, 300)] = np.inf                 # failed candidates carry +inf
    costs[100] = costs[3000] = costs.min() - 1.0             # a tie across ranks -> lowest index wins
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, costs, groups, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    per = B // world // groups
    for rank, c, i in res:
        for g in range(groups):
            # group g of the global view = union over ranks of their g-th local group
            cand = np.concatenate([np.arange(r * (B // world) + g * per, r * (B // world) + (g + 1) * per) for r in range(world)])
            j = cand[np.argmin(costs[cand])]
            assert i[g] == j and c[g] == costs[j]
    assert all((res[0][2] == r[2]).all() for r in res)       # every rank agrees
    assert 100 in res[0][2]


def test_shard_bounds_cover_the_batch():
    for B, W in [(65536, 8), (4096, 3), (5, 8), (1, 1)]:
        cover = []
        for r in range(W):
            lo, hi = shard_bounds(B, W, r)
            cover += list(range(lo, hi))
        assert cover == list(range(B))


def test_all_failed_group_reports_minus_one():
    c, i = global_argmin(torch.tensor([float("inf")], dtype=torch.float64), torch.tensor([-1]))
    assert int(i[0]) == -1
