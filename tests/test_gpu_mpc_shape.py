"""BASELINE.json config 5 as a parity case: 128 ego agents x 512 candidate corridors, 20 segments,
re-solved over a few receding-horizon steps with a per-agent arg-min (groups of 512, GPU-local)."""
import numpy as np
import pytest

from helpers import O
from spectral_amd import layout as L
from spectral_amd import synth

pytestmark = pytest.mark.gpu
AGENTS, CAND, S = 128, 512, 20


def bezier_state(c, t, tau):
    """(p, v, a) of the reference's time-scaled quintic at tau in [0,1] (solve_3d.cc:1366-1388)."""
    from scipy.special import comb
    B = lambda n, i: comb(n, i) * tau ** i * (1 - tau) ** (n - i)
    p = t * sum(c[i] * B(5, i) for i in range(6))
    v = sum(5 * (c[i + 1] - c[i]) * B(4, i) for i in range(5))
    a = sum(20 * (c[i + 2] - 2 * c[i + 1] + c[i]) * B(3, i) for i in range(4)) / t
    return p, v, a


def test_receding_horizon_per_agent_argmin():
    import torch
    from spectral_amd.solver import BatchSolver
    solver = BatchSolver(0)
    batch, sh = synth.make_batch(AGENTS * CAND, S, config=5, agents=AGENTS)
    dt = 0.02                                                     # 50 Hz
    for step in range(3):
        db = solver.upload(batch)
        o = solver.solve(db, sh)
        bi, bc = solver.argmin(o["cost"], group=CAND)             # one winner per agent, no collective
        torch.cuda.synchronize()
        cost = o["cost"].cpu().numpy(); status = o["status"].cpu().numpy(); ctrl = o["ctrl"].cpu().numpy()
        bi = bi.cpu().numpy(); bc = bc.cpu().numpy()
        assert (status > 0).mean() > 0.98, (step, (status > 0).mean())
        assert (batch.init.reshape(AGENTS, CAND, 6) == batch.init.reshape(AGENTS, CAND, 6)[:, :1]).all()
        for a in range(AGENTS):
            seg = cost[a * CAND:(a + 1) * CAND]
            assert bi[a] == a * CAND + int(np.argmin(seg)) and bc[a] == seg.min()
        # the winners of a few agents against the oracle's optimum of the same (updated) problem
        for a in (0, 57, 127):
            w = int(bi[a])
            xs, obj, st, _ = O.batch_solve(batch, sh, w, w + 1, exact=True)
            assert st[0] == 1
            assert np.abs(ctrl[w] - xs[0]).max() <= 1e-5 * np.abs(xs[0]).max()
        # advance every agent along its winner by dt: new initial state for all of its candidates
        for a in range(AGENTS):
            w = int(bi[a])
            t0 = batch.seg[L.F_T, w, 0]
            ps, vs, as_ = bezier_state(ctrl[w, 0:6], t0, dt / t0)
            pl, vl, al = bezier_state(ctrl[w, 6 * S:6 * S + 6], t0, dt / t0)
            batch.init[a * CAND:(a + 1) * CAND] = [ps, vs, as_, pl, vl, al]
