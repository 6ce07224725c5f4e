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


def test_mpc_tool_warm_start_beats_cold_and_matches_oracle(tmp_path):
    """tools/mpc_bench.py (config 5 loop with window roll, eval_states, warm start) on a small fleet: winners agree
    with the oracle's x* at the checked steps, everything stays solved, warm start needs fewer iterations."""
    import json
    import os
    import subprocess
    import sys
    from spectral_amd.layout import Batch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(tag, *extra):
        dump = str(tmp_path / (tag + ".npz"))
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "mpc_bench.py"), "--agents", "8", "--cand", "64",
                              "--steps", "56", "--check", "3", "--dump", dump, *extra], check=True, capture_output=True,
                             text=True, timeout=600).stdout.strip().splitlines()[-1]
        return json.loads(out), np.load(dump)

    _, sh = synth.make_batch(8, S, config=5)
    (warm, wd), (cold, cd) = run("warm"), run("cold", "--cold")
    for r, d in ((warm, wd), (cold, cd)):
        assert r["solved_fraction_min"] >= 0.98, r
        n = int(d["n"])
        assert n == r["dumped_winners"] and n >= 6
        for i in range(n):                              # every dumped winner against the oracle's optimum of ITS problem
            b = Batch(B=1, S=S, seg=np.ascontiguousarray(d["seg_%d" % i][:, None, :]), init=d["init_%d" % i][None],
                      ref_end=d["ref_end_%d" % i][None], dl_bounds=d["dl_bounds_%d" % i][None])
            xs, obj, st, _ = O.batch_solve(b, sh, 0, 1, exact=True)
            assert st[0] == 1
            assert np.abs(d["ctrl_%d" % i] - xs[0]).max() <= 1e-5 * np.abs(xs[0]).max(), (i, int(d["step_%d" % i]))
    assert warm["mean_ipm_iterations"] <= cold["mean_ipm_iterations"] - 2.0, (warm, cold)
