"""The batch bench.py TIMES, under the oracle: `bench.make_workload("scenario1", 65536, 20, v, 0)` -- BASELINE
config 3 (v = 0) and config 4 (v = 1) exactly as rank 0 of the driver's run builds them -- solved by the call bench.py makes
(the library's own choice of form: the lean kernels in two launches, btrapz_last_solve_form() == 11), then

  * on EVERY accepted candidate: the properties the domain offers at any size -- every constraint row of the reference
    (solve_3d.cc:823-888, cuboid_3d.cc:677-689, 826-827) met, the initial state and C2 continuity at every joint
    (solve_3d.cc:896-949), finite control points and cost;
  * on a strided sample of 2 048 candidates spread over the whole batch (every 32nd, not one contiguous slice): the accept
    set and the control points of the oracle's exact solve of the reference's general (P, q, A, l, u) -- acceptance
    solve_3d.cc:1251-1277 -- |ctrl - x*| <= 1e-5 |x*| (north_star: 1e-4), and the objective value.
"""
import os

import numpy as np
import pytest

from helpers import O
from spectral_amd import layout as L
from test_gpu_properties import rows

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("variant", [0, 1], ids=["config 3 (trapezoid)", "config 4 (cuboid)"])
def test_the_batch_bench_times_against_the_oracle(variant):
    import torch
    import bench
    from spectral_amd.solver import BatchSolver
    B, S = 65536, 20
    batch, sh = bench.make_workload("scenario1", B, S, variant, 0)
    solver = BatchSolver(0)
    db = solver.upload(batch)
    o = solver.solve(db, sh, lean=0)            # bench.py's step: solver.solve(db, shared, lean=a.lean), a.lean = 0
    torch.cuda.synchronize()
    assert solver.ctx.last_solve_form() == 11   # lean form, capped launch + resume launch: the kernels of the driver's line
    ctrl, cost = o["ctrl"].cpu().numpy(), o["cost"].cpu().numpy()
    status = o["status"].cpu().numpy()
    ok = status > 0
    assert ok.mean() > (0.98 if variant == 0 else 0.7) and (~ok).any()          # (both kinds are in the batch)
    assert np.isfinite(ctrl[ok]).all() and np.isfinite(cost[ok]).all()
    # ---- every accepted candidate: rows, initial state, continuity ----
    acc = L.Batch(B=int(ok.sum()), S=S, seg=np.ascontiguousarray(batch.seg[:, ok]), init=batch.init[ok], ref_end=batch.ref_end[ok],
                  dl_bounds=batch.dl_bounds[ok])
    ca = ctrl[ok]
    for axis in (0, 1):
        c, val, lo, up = rows(acc, sh, ca, axis)
        scale = 1 + np.maximum(np.abs(lo), np.abs(up))
        assert ((lo - val) / scale).max() <= 1e-7 and ((val - up) / scale).max() <= 1e-7
        t = acc.seg[L.F_T]
        init = acc.init[:, 3 * axis:3 * axis + 3]
        assert np.abs(t[:, 0] * c[:, 0, 0] - init[:, 0]).max() <= 1e-9 * (1 + np.abs(init[:, 0]).max())
        assert np.abs(5 * (c[:, 0, 1] - c[:, 0, 0]) - init[:, 1]).max() <= 1e-8
        assert np.abs(20 * (c[:, 0, 0] - 2 * c[:, 0, 1] + c[:, 0, 2]) - init[:, 2] * t[:, 0]).max() <= 1e-8
        pe = t[:, :-1] * c[:, :-1, 5]; pb = t[:, 1:] * c[:, 1:, 0]
        assert np.abs(pe - pb).max() <= 1e-9 * (1 + np.abs(pe).max())
        assert np.abs((c[:, :-1, 5] - c[:, :-1, 4]) - (c[:, 1:, 1] - c[:, 1:, 0])).max() <= 1e-8
        ae = (c[:, :-1, 3] - 2 * c[:, :-1, 4] + c[:, :-1, 5]) / t[:, :-1]; ab = (c[:, 1:, 0] - 2 * c[:, 1:, 1] + c[:, 1:, 2]) / t[:, 1:]
        assert np.abs(ae - ab).max() <= 1e-8
    # ---- a strided sample over the whole batch against the oracle's exact solve ----
    idx = np.arange(13, B, 32)
    assert len(idx) == 2048
    sub = L.Batch(B=len(idx), S=S, seg=np.ascontiguousarray(batch.seg[:, idx]), init=np.ascontiguousarray(batch.init[idx]),
                  ref_end=np.ascontiguousarray(batch.ref_end[idx]), dl_bounds=np.ascontiguousarray(batch.dl_bounds[idx]))
    x, obj, ost, _ = O.batch_solve(sub, sh, 0, len(idx), exact=True, threads=len(os.sched_getaffinity(0)))
    oa, ka = ost > 0, ok[idx]
    assert np.array_equal(oa, ka), (np.nonzero(oa != ka)[0][:8], ost[oa != ka][:8], status[idx][oa != ka][:8])   # the oracle's accept set
    both = oa & ka
    assert both.sum() > 1400
    rel = np.abs(ctrl[idx][both] - x[both]).max(axis=1) / np.abs(x[both]).max(axis=1)
    assert rel.max() <= 1e-5, rel.max()
    assert (np.abs(cost[idx][both] - obj[both]) <= 1e-6 * (1 + np.abs(obj[both]))).all()
    # the winner bench.py reports is the arg-min of what was accepted
    bi, bc = solver.argmin(o["cost"])
    torch.cuda.synchronize()
    assert int(bi[0]) == int(np.argmin(np.where(ok, cost, np.inf)))
