"""HIP path vs the CPU oracle, through the C-ABI (needs an MI355X: pytest -m gpu).

Bar (north_star): control points within 1e-4 relative of the reference solve.  The reference's
own OSQP answer scatters 1e-5..1e-2 around the optimum x* (test_oracle_solvers.py, and it is
0.34 m off on c4 where OSQP ran out of iterations), so the HIP path is held to the optimum
itself: |ctrl - x*|_inf <= 1e-5 |x*|_inf -- ten times tighter than the bar; typical agreement
is 1e-10, the worst cases (tiny or badly conditioned problems, where both solvers sit at their
round-off floor) reach ~2e-6 -- with x* from the oracle's dense interior point (KKT-certified,
cross-checked with HiGHS and tight ADMM)."""
import os

import numpy as np
import pytest

from helpers import O, oracle_qp_from_batch
from spectral_amd import layout as L
from spectral_amd import native, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
RTOL = 1e-5


@pytest.fixture(scope="module")
def ctx():
    return native.Context(0)


def rel(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


@pytest.mark.parametrize("cfg,S,variant", [(2, 10, 0), (3, 20, 0), (4, 20, 1), (9, 7, 0), (8, 13, 1)])
def test_against_committed_xstar(ctx, cfg, S, variant):
    g = np.load(os.path.join(GOLD, "synthetic_xstar.npz"))
    B, S_, v_, nb = g["cfg%d/meta" % cfg]
    assert (S_, v_) == (S, variant)
    batch, sh = synth.make_batch(int(B), S, config=cfg, variant=variant)
    ctrl, cost, status, iters = ctx.solve_host(batch, sh)
    assert (status == 1).all()
    xs, obj = g["cfg%d/xstar" % cfg], g["cfg%d/obj" % cfg]
    for b in range(int(nb)):
        assert rel(ctrl[b], xs[b]) <= RTOL, (b, rel(ctrl[b], xs[b]))
        assert abs(cost[b] - obj[b]) <= 1e-8 * abs(obj[b])
    assert iters.max() < 30


@pytest.mark.parametrize("S,variant", [(20, 0), (20, 1), (10, 0)])
def test_scenario1_batches_against_committed_xstar(ctx, S, variant):
    """BASELINE configs 3 / 4 on the workload bench.py times: scenario_1-shaped corridors (synth.make_scenario1_batch).
    Solvable candidates agree with the oracle's x*; the others (a late slow obstacle in front of a fast ego; empty
    inscribed intervals of the cuboid variant) are flagged by both."""
    g = np.load(os.path.join(GOLD, "scenario1_xstar.npz"))
    key = "S%d_v%d" % (S, variant)
    B, S_, v_, nb = g[key + "/meta"]
    batch, sh = synth.make_scenario1_batch(int(B), S, variant)
    ctrl, cost, status, iters = ctx.solve_host(batch, sh)
    xs, obj, st = g[key + "/xstar"], g[key + "/obj"], g[key + "/status"]
    n = 0
    for b in range(int(nb)):
        if st[b] == 1:
            n += 1
            assert status[b] in (1, 2), (b, status[b])
            assert rel(ctrl[b], xs[b]) <= RTOL, (b, rel(ctrl[b], xs[b]))
            assert abs(cost[b] - obj[b]) <= 1e-7 * abs(obj[b])
        else:
            assert status[b] < 0 and np.isinf(cost[b]), (b, status[b], st[b])
    assert n >= nb // 2


@pytest.mark.parametrize("S,variant,B", [(1, 0, 5), (2, 0, 33), (3, 1, 64), (5, 0, 17), (16, 0, 9), (21, 1, 7), (32, 0, 5),
                                         (33, 0, 3), (64, 0, 2)])
def test_ragged_shapes_against_live_oracle(ctx, S, variant, B):
    """Every packing of segments into a wavefront: 64/S groups, idle tail lanes, partial last wave."""
    batch, sh = synth.make_batch(B, S, config=40 + S, variant=variant)
    ctrl, cost, status, iters = ctx.solve_host(batch, sh)
    xs, obj, st, _ = O.batch_solve(batch, sh, 0, min(B, 6), exact=True, threads=4)
    for b in range(min(B, 6)):
        if st[b] == 1:
            assert status[b] == 1
            assert rel(ctrl[b], xs[b]) <= RTOL, (S, b, rel(ctrl[b], xs[b]))
        else:
            assert status[b] <= 0


def test_single_candidate(ctx):
    batch, sh = synth.make_batch(1, 20, config=3)
    ctrl, cost, status, iters = ctx.solve_host(batch, sh)
    xs, obj, st, _ = O.batch_solve(batch, sh, 0, 1, exact=True)
    assert status[0] == 1 and rel(ctrl[0], xs[0]) <= RTOL


def test_nonuniform_segment_durations(ctx):
    """Bundled-scenario-like t pattern (1, 1, 0.5, 0.5, 0.1, 0.9 ...): P and the joint maps depend on t."""
    batch, sh = synth.make_batch(48, 8, config=77)
    rng = np.random.default_rng(3)
    batch.seg[L.F_T] = rng.choice([0.1, 0.3, 0.5, 0.7, 0.9, 1.0], size=(48, 8))
    batch.seg[L.F_DOWN_BIAS] -= 30.0; batch.seg[L.F_UPP_BIAS] += 30.0   # keep it feasible
    ctrl, cost, status, iters = ctx.solve_host(batch, sh)
    xs, obj, st, _ = O.batch_solve(batch, sh, 0, 12, exact=True, threads=4)
    n = 0
    for b in range(12):
        if st[b] == 1:
            n += 1
            assert status[b] in (1, 2)
            assert rel(ctrl[b], xs[b]) <= 1e-5, (b, rel(ctrl[b], xs[b]))
    assert n >= 8


def test_infeasible_candidates_are_flagged_and_lose_the_argmin(ctx):
    import torch
    from spectral_amd.solver import BatchSolver
    batch, sh = synth.make_batch(64, 10, config=2)
    bad = [3, 17, 40]
    batch.seg[L.F_UPP_BIAS, bad[0], 4] = batch.seg[L.F_DOWN_BIAS, bad[0], 4] - 5.0   # l > u: inconsistent bounds
    batch.seg[L.F_L_UPP_BIAS, bad[1], :] = batch.seg[L.F_L_DOWN_BIAS, bad[1], :] + 1e-3
    batch.seg[L.F_L_UPP_SKEW, bad[1], :] = batch.seg[L.F_L_DOWN_SKEW, bad[1], :]
    batch.init[bad[1], 3] += 2.0                                                   # starts outside a 1 mm corridor
    batch.seg[L.F_DS_HI, bad[2], :] = 0.5                                           # cannot keep up: v <= 0.5 but s must advance
    ctrl, cost, status, iters = ctx.solve_host(batch, sh)
    xs, obj, st, _ = O.batch_solve(batch, sh, 0, 64, exact=True, threads=8)
    for b in bad:
        assert st[b] != 1 and status[b] < 0 and np.isinf(cost[b])
    ok = [b for b in range(64) if b not in bad]
    assert (status[ok] == 1).all()
    solver = BatchSolver(0)
    o = solver.solve(solver.upload(batch), sh)
    bi, bc = solver.argmin(o["cost"])
    torch.cuda.synchronize()
    assert int(bi[0]) == int(np.argmin(cost)) and int(bi[0]) not in bad
    # per-agent arg-min (config 5: groups of candidates), ties/inf handled per group
    bi4, bc4 = solver.argmin(o["cost"], group=16, index_base=1000)
    torch.cuda.synchronize()
    for g in range(4):
        assert int(bi4[g]) == 1000 + 16 * g + int(np.argmin(cost[16 * g:16 * g + 16]))


def test_all_failed_group_returns_minus_one(ctx):
    import torch
    from spectral_amd.solver import BatchSolver
    batch, sh = synth.make_batch(8, 10, config=2)
    batch.seg[L.F_UPP_BIAS] = batch.seg[L.F_DOWN_BIAS] - 1.0
    solver = BatchSolver(0)
    o = solver.solve(solver.upload(batch), sh)
    bi, bc = solver.argmin(o["cost"])
    torch.cuda.synchronize()
    assert int(bi[0]) == -1 and (o["status"].cpu().numpy() == -3).all()


def test_device_path_equals_host_path_and_is_deterministic(ctx):
    import torch
    from spectral_amd.solver import BatchSolver
    batch, sh = synth.make_batch(300, 20, config=3)
    c1, cost1, s1, i1 = ctx.solve_host(batch, sh)
    solver = BatchSolver(0)
    db = solver.upload(batch)
    o = solver.solve(db, sh); torch.cuda.synchronize()
    c2 = o["ctrl"].cpu().numpy().copy()
    o = solver.solve(db, sh); torch.cuda.synchronize()
    c3 = o["ctrl"].cpu().numpy()
    assert (c1 == c2).all() and (c2 == c3).all()        # bit-identical: no atomics, fixed reduction order


def test_sampling_kernel_matches_oracle_sampling(ctx):
    import torch
    from spectral_amd.solver import BatchSolver
    batch, sh = synth.make_batch(16, 10, config=2)
    batch.seg[L.F_T, :, 3] = 5 * 0.1; batch.seg[L.F_T, :, 7] = 3 * 0.1   # (end_t-beg_t)*delta as the reference forms t
    solver = BatchSolver(0)
    db = solver.upload(batch)
    o = solver.solve(db, sh)
    out, npts = solver.sample(db, o["ctrl"], torch.tensor([0, 5, 15]), sh.delta)
    torch.cuda.synchronize()
    ctrl = o["ctrl"].cpu().numpy(); out = out.cpu().numpy(); npts = npts.cpu().numpy()
    for j, b in enumerate([0, 5, 15]):
        cubes = []
        for k in range(10):
            c = O.Cube(); c.t = float(batch.seg[L.F_T, b, k]); cubes.append(c)
        rc, ref = O.sample(cubes, sh.delta, ctrl[b], batch.init[b, :3], batch.init[b, 3:])
        assert rc == 0 and npts[j] == len(ref[0]) == 1 + 8 * 10 + 5 + 3
        for a in range(6):
            assert np.abs(out[j, a, :npts[j]] - ref[a]).max() <= 1e-9 * (1 + np.abs(ref[a]).max())


def test_tolerance_option(ctx):
    """A looser KKT target stops earlier and still meets the 1e-4 bar."""
    batch, sh = synth.make_batch(64, 20, config=3)
    xs, obj, st, _ = O.batch_solve(batch, sh, 0, 8, exact=True, threads=4)
    c_tight, _, s_t, it_t = ctx.solve_host(batch, sh)
    c_loose, _, s_l, it_l = ctx.solve_host(batch, sh, eps=1e-6)
    assert it_l.mean() < it_t.mean()
    for b in range(8):
        assert rel(c_loose[b], xs[b]) <= 1e-4


@pytest.mark.parametrize("variant,b", [(0, 2070), (1, 52944)])
def test_candidates_whose_complementarity_cycles(ctx, variant, b):
    """The two candidates of the bench batches (of 4 x 65 536, all compared with the oracle's x*) on which Mehrotra's
    corrector settles into a two-cycle: residuals at 1e-11, mu going 5e-4 <-> 1.5e-3.  Reported as unsolved until
    round 2; the second chance without the second-order term (btrapz_kernels.hip) converges."""
    from oracle import oracle as O
    from spectral_amd import synth
    from spectral_amd.solver import BatchSolver
    batch, sh = synth.make_scenario1_batch(65536, 20, variant)
    x, obj, st, it = O.batch_solve(batch, sh, b, b + 1, exact=True)
    solver = BatchSolver(0)
    o = solver.solve(solver.upload(batch.slice(b, b + 1)), sh)
    c = o["ctrl"][0].cpu().numpy()
    assert st[0] == 1 and int(o["status"][0]) in (1, 2)
    assert np.abs(c - x[0]).max() <= 1e-5 * np.abs(x[0]).max()


@pytest.mark.parametrize("make", ["scenario1_20_trapezoid", "scenario1_20_cuboid", "generic_20", "scenario1_10"])
def test_acceptance_and_optimum_against_the_live_oracle_on_bench_batches(ctx, make):
    """A 3 072-candidate slice of each bench batch: the kernel accepts exactly the candidates for which the oracle's
    exact solve finds x*, and returns that x*.  (The same comparison over all 4 x 65 536 candidates -- scratch run,
    DESIGN.md section 5 -- ends at 0 / 0 disagreements, worst relative deviation 1.5e-6.)"""
    from oracle import oracle as O
    from spectral_amd.solver import BatchSolver
    batch, sh = {"scenario1_20_trapezoid": lambda: synth.make_scenario1_batch(65536, 20, 0),
                 "scenario1_20_cuboid": lambda: synth.make_scenario1_batch(65536, 20, 1),
                 "generic_20": lambda: synth.make_batch(65536, 20, config=3),
                 "scenario1_10": lambda: synth.make_scenario1_batch(65536, 10, 0)}[make]()
    off, n = 50000, 3072
    part = batch.slice(off, off + n)
    solver = BatchSolver(0)
    o = solver.solve(solver.upload(part), sh)
    st = o["status"].cpu().numpy(); ctrl = o["ctrl"].cpu().numpy()
    x, obj, ost, oit = O.batch_solve(part, sh, 0, n, exact=True, threads=min(8, len(os.sched_getaffinity(0))))
    ka, oa = st > 0, ost > 0
    assert np.array_equal(ka, oa), (np.nonzero(ka != oa)[0][:10] + off, st[ka != oa][:10], ost[ka != oa][:10])
    err = np.abs(ctrl[ka] - x[ka]).max(axis=1) / np.abs(x[ka]).max(axis=1)
    assert ka.sum() >= n // 2 and err.max() <= 1e-5
