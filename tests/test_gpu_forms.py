"""One table, every product form of the solve kernel, the SAME candidates: a change of the method that
lands in one body of the kernel and not in the other (btrapz_kernels.hip / btrapz_lean_body.h), or in one instantiation
and not its sibling, fails here instead of waiting for a fuzz campaign.  Each row: accept set == the oracle's exact solve,
control points within 1e-5 of x* (1e-4 where the form reports "solved inaccurate"), cost == the oracle's objective.
Also: the property "an accepted result is finite and inside its rows" over garbage inputs for the BATCHED entry points
(the round-4 NaN-cost find was behind find_traj only).  Replaces osqp_setup + osqp_solve + the acceptance test of
src/solve_3d.cc:1246-1277 for candidate sets."""
import numpy as np
import pytest

from helpers import O
from spectral_amd import layout as L
from spectral_amd import synth

pytestmark = pytest.mark.gpu
N = 256


@pytest.fixture(scope="module")
def solver():
    from spectral_amd.solver import BatchSolver
    return BatchSolver(0)


@pytest.fixture(scope="module")
def table():
    """256 scenario_1 candidates of 20 segments (a few without a solution), 256 cuboid ones (a quarter without), 256
    generic ones of 10 segments, under weights.txt and under one row from the far end of the reference's weight space."""
    out = {}
    far = [40.86, 0.01, 38.14, 31.77, 41.19, 1.69, 11.55, 43.99, 27.12, 1.26]       # all_weights.txt, smallest s-jerk weight
    for key, (batch, sh) in {"s1": synth.make_scenario1_batch(N, 20, 0), "cub": synth.make_scenario1_batch(N, 20, 1),
                             "g10": synth.make_batch(N, 10, config=2)}.items():
        for wname, w in (("ref", None), ("far", far)):
            import copy
            shw = copy.copy(sh)
            if w is not None:
                shw.w_s = (w[4], w[5], w[0], w[1]); shw.w_l = (w[6], w[7], w[2], w[3]); shw.weight_end_s, shw.weight_end_l = w[8], w[9]
            xs, obj, st, it = O.batch_solve(batch, shw, 0, N, exact=True, threads=8)
            out[(key, wname)] = (batch, shw, xs, obj, st)
    return out


FORMS = {
    "lean": dict(lean=1, cap_iter=-1, split=-1),
    "lean two launches": dict(lean=1, cap_iter=5, split=-1),
    "packed": dict(lean=-1, cap_iter=-1, split=-1),
    "packed two launches": dict(lean=-1, cap_iter=5, split=-1),
    "split": dict(split=1),
    "elastic rows on every candidate": dict(elastic=2),
}


def check(label, r, xs, obj, st, n=N, rescue=False):
    ok_o = st[:n] > 0
    ok_h = r["status"][:n] > 0
    if rescue:      # elastic = 2 also returns least-violation answers for candidates without a solution: the solvable ones must all be there
        assert (ok_h | ~ok_o).all(), (label, np.nonzero(~ok_h & ok_o)[0][:8])
    else:
        assert np.array_equal(ok_h, ok_o), (label, np.nonzero(ok_h != ok_o)[0][:8], r["status"][:n][ok_h != ok_o][:8], st[:n][ok_h != ok_o][:8])
    idx = np.nonzero(ok_o)[0]
    assert idx.size > n // 2
    P = xs.shape[1]
    err = np.abs(r["ctrl"][idx, :P] - xs[idx]).max(axis=1) / np.abs(xs[idx]).max(axis=1)
    # (elastic rows: the relaxed optimum is x* up to delta x the multipliers of the tight rows -- 1.4e-4 on a scenario_1
    #  candidate pressed against an obstacle ramp, 1.4e-3 on a cuboid one under the far weight row)
    tol = np.where(r["status"][idx] == 2, 1e-4, 1e-5) if not rescue else np.full(idx.size, 5e-3)
    assert (err <= tol).all(), (label, idx[err > tol][:8], err.max())
    if not rescue:
        assert np.isfinite(r["cost"][idx]).all() and np.isinf(r["cost"][:n][~ok_o]).all(), label
        assert (np.abs(r["cost"][idx] - obj[idx]) <= 1e-7 * (1.0 + np.abs(obj[idx]))).all(), label


@pytest.mark.parametrize("key", ["s1", "cub", "g10"])
@pytest.mark.parametrize("wname", ["ref", "far"])
def test_every_form_on_the_same_candidates(solver, table, key, wname):
    import torch
    batch, sh, xs, obj, st = table[(key, wname)]
    db = solver.upload(batch)

    def grab(o):
        torch.cuda.synchronize()
        return {k: v.cpu().numpy().copy() for k, v in o.items() if k in ("ctrl", "cost", "status", "iters")}
    seen = {}
    for form, kw in FORMS.items():
        r = grab(solver.solve(db, sh, **kw))
        seen[form] = solver.ctx.last_solve_form()
        check("%s/%s %s" % (key, wname, form), r, xs, obj, st, rescue="elastic" in form)
    assert seen["lean"] == 8 and seen["lean two launches"] == 11 and seen["packed"] == 0 and seen["packed two launches"] == 3 and seen["split"] == 1
    # the ragged entry point on the same record (lean and packed instantiations that carry the end-lane fix-up)
    rec = dict(B=batch.B, seg_stride=batch.S, seg=db.seg, seg_count=torch.full((batch.B,), batch.S, dtype=torch.int32, device=solver.device),
               init=db.init, ref_end=db.ref_end, dl_bounds=db.dl_bounds)
    for form, kw in (("ragged lean", dict(lean=1, cap_iter=-1)), ("ragged packed", dict(lean=-1, cap_iter=-1)), ("ragged lean two launches", dict(lean=1, cap_iter=5))):
        check("%s/%s %s" % (key, wname, form), grab(solver.solve_ragged(rec, sh, **kw)), xs, obj, st)
    # the warm entry point, cold (no guess: its kernels' own cold start) and from the solve under the other weight row
    for form, kw in (("warm entry, lean", dict(lean=1)), ("warm entry, packed", dict(lean=-1))):
        check("%s/%s %s cold" % (key, wname, form), grab(solver.solve(db, sh, keep_multipliers=True, **kw)), xs, obj, st)
        other = table[(key, "far" if wname == "ref" else "ref")][1]
        cold = solver.solve(db, other, keep_multipliers=True, **kw)
        x0 = solver.eval_states(db, cold["ctrl"].clone(), torch.from_numpy(np.cumsum(batch.seg[L.F_T], axis=1)))
        check("%s/%s %s warm" % (key, wname, form), grab(solver.solve(db, sh, warm=dict(x0=x0, lam=cold["lam"].clone()), **kw)), xs, obj, st)


def test_the_long_form_on_the_same_kind_of_candidates(solver):
    """65-256 segments: one axis problem per workgroup (the only form that serves them)."""
    import torch
    batch, sh = synth.make_scenario1_batch(6, 70, 0)
    o = solver.solve(solver.upload(batch), sh)
    torch.cuda.synchronize()
    assert solver.ctx.last_solve_form() == 2
    r = {k: v.cpu().numpy().copy() for k, v in o.items()}
    from helpers import oracle_qp_from_batch
    n = 0
    for b in range(3):        # (the oracle's batch entry point stops at 64 segments: candidate by candidate)
        x, _, info = oracle_qp_from_batch(batch, sh, b).solve_exact(max_iter=120)
        assert (info.status in (1, 2)) == (r["status"][b] > 0)
        if r["status"][b] > 0:
            assert np.abs(r["ctrl"][b] - np.asarray(x)).max() <= 1e-5 * np.abs(x).max()
            n += 1
    assert n >= 1


def garbage(batch, rng, n):
    """Candidates damaged the way the round-4 campaign's finds were: inf / nan / +-1e10 among the bounds, t <= 0, l > u,
    a non-finite initial state or reference."""
    import copy
    b = copy.copy(batch)
    b.seg = batch.seg.copy(); b.init = batch.init.copy(); b.ref_end = batch.ref_end.copy(); b.dl_bounds = batch.dl_bounds.copy()
    bad = rng.choice(batch.B, n, replace=False)
    for i, c in enumerate(bad):
        k = int(rng.integers(batch.S))
        kind = i % 10
        val = [np.inf, -np.inf, np.nan, 1e10, -1e10][i % 5]
        if kind == 0: b.seg[L.F_UPP_BIAS, c, k] = val
        elif kind == 1: b.seg[L.F_DOWN_BIAS, c, k] = val
        elif kind == 2: b.seg[L.F_L_UPP_BIAS, c, k] = val
        elif kind == 3: b.seg[L.F_T, c, k] = [0.0, -1.0, np.nan, np.inf, 1e-300][i % 5]
        elif kind == 4: b.seg[L.F_DS_HI, c, k] = val
        elif kind == 5: b.dl_bounds[c, int(rng.integers(10))] = val
        elif kind == 6: b.init[c, int(rng.integers(6))] = val
        elif kind == 7: b.seg[L.F_UPP_BIAS, c, k] = b.seg[L.F_DOWN_BIAS, c, k] - 1.0          # l > u
        elif kind == 8: b.ref_end[c, int(rng.integers(2))] = val
        else: b.seg[L.F_X_BIAS, c, k] = val
    return b, bad


@pytest.mark.parametrize("make", [lambda: synth.make_scenario1_batch(768, 20, 0), lambda: synth.make_batch(768, 10, config=2), lambda: synth.make_scenario1_batch(768, 20, 1)])
def test_accepted_results_of_the_batched_entry_points_are_finite_and_inside_their_rows(solver, make):
    """Whatever the input, a candidate that comes back with status 1 or 2 has finite control points and a finite cost, and
    -- unless it was rescued (elastic) -- its control points meet the reference's rows (assembled by the oracle) to 1e-6 of
    the bounds' scale; the undamaged candidates keep their results bit for bit (garbage stays inside its own group)."""
    import torch
    from helpers import oracle_qp_from_batch
    batch, sh = make()
    rng = np.random.default_rng(5)
    dirty, bad = garbage(batch, rng, 120)
    db, dbd = solver.upload(batch), solver.upload(dirty)
    clean_mask = np.ones(batch.B, dtype=bool); clean_mask[bad] = False
    rec = dict(B=batch.B, seg_stride=batch.S, seg=dbd.seg, seg_count=torch.full((batch.B,), batch.S, dtype=torch.int32, device=solver.device),
               init=dbd.init, ref_end=dbd.ref_end, dl_bounds=dbd.dl_bounds)
    runs = [("lean", lambda d: solver.solve(d, sh, lean=1, cap_iter=-1, split=-1)), ("lean two launches", lambda d: solver.solve(d, sh, lean=1, cap_iter=4, split=-1)),
            ("packed", lambda d: solver.solve(d, sh, lean=-1, cap_iter=-1, split=-1)), ("rescue pass", lambda d: solver.solve(d, sh, lean=-1, elastic=1, split=-1)),
            ("warm entry", lambda d: solver.solve(d, sh, keep_multipliers=True, lean=1))]
    for label, run in runs:
        ref = {k: v.cpu().numpy().copy() for k, v in run(db).items() if k in ("ctrl", "cost", "status")}
        got = {k: v.cpu().numpy().copy() for k, v in run(dbd).items() if k in ("ctrl", "cost", "status")}
        torch.cuda.synchronize()
        acc = got["status"] > 0
        assert np.isfinite(got["ctrl"][acc]).all() and np.isfinite(got["cost"][acc]).all(), label
        assert np.array_equal(got["status"][clean_mask], ref["status"][clean_mask]), label
        okc = clean_mask & (ref["status"] > 0)
        assert np.array_equal(got["ctrl"][okc], ref["ctrl"][okc]) and np.array_equal(got["cost"][okc], ref["cost"][okc]), label
        if label == "rescue pass":
            continue
        for c in bad[acc[bad]]:                      # an accepted damaged candidate: inside the rows the oracle assembles
            qp = oracle_qp_from_batch(dirty, sh, int(c))
            A = qp.dense()[1]
            Ax = A @ got["ctrl"][c]
            lo, up = np.array(qp.l), np.array(qp.u)
            scale = 1.0 + np.minimum(np.maximum(np.abs(lo), np.abs(up)), 1e9)
            viol = np.maximum(np.maximum(lo - Ax, Ax - up), 0.0) / scale
            assert np.nanmax(viol) <= 1e-6, (label, int(c), float(np.nanmax(viol)))
    o = solver.solve_ragged(rec, sh, lean=1)
    torch.cuda.synchronize()
    acc = o["status"].cpu().numpy() > 0
    assert np.isfinite(o["ctrl"].cpu().numpy()[acc]).all() and np.isfinite(o["cost"].cpu().numpy()[acc]).all()


@pytest.mark.parametrize("family", ["generic20", "scenario1", "generic10 cuboid"])
def test_bounds_left_at_the_references_default_are_no_bounds(solver, family):
    """Rows left at +-1e10 (src/piecewise_jerk_problem.cc:9,25-35) in any field a caller can leave there -- corridor lines,
    velocity intervals, the five dl pairs, the header's acceleration and jerk limits: the solve treats them as absent
    (btrapz_ipm.h, "bounds that are no bounds") and must land on the optimum the oracle finds with the literal 1e10."""
    import copy
    import torch
    rng = np.random.default_rng(17)
    batch, sh = {"generic20": lambda: synth.make_batch(384, 20, config=3), "scenario1": lambda: synth.make_scenario1_batch(384, 20, 0),
                 "generic10 cuboid": lambda: synth.make_batch(384, 10, config=2, variant=1)}[family]()
    b = copy.copy(batch)
    b.seg = batch.seg.copy(); b.dl_bounds = batch.dl_bounds.copy()
    pick = lambda p: rng.random((batch.B, batch.S)) < p
    m = pick(0.15); b.seg[L.F_UPP_BIAS][m] = 1e10; b.seg[L.F_UPP_SKEW][m] = 0.0
    m = pick(0.15); b.seg[L.F_DOWN_BIAS][m] = -1e10; b.seg[L.F_DOWN_SKEW][m] = 0.0
    m = pick(0.10); b.seg[L.F_L_UPP_BIAS][m] = 1e10; b.seg[L.F_L_UPP_SKEW][m] = 0.0; b.seg[L.F_END_L][m] = 1e10
    m = pick(0.10); b.seg[L.F_L_DOWN_BIAS][m] = -1e10; b.seg[L.F_L_DOWN_SKEW][m] = 0.0; b.seg[L.F_BEG_L][m] = -1e10
    m = pick(0.20); b.seg[L.F_DS_HI][m] = 1e10
    m = pick(0.20); b.seg[L.F_DS_LO][m] = -1e10
    m = rng.random(b.dl_bounds.shape) < 0.3
    b.dl_bounds[m] = np.where(np.arange(10) % 2 == 0, -1e10, 1e10)[np.nonzero(m)[1]]
    shf = copy.copy(sh)
    shf.ddl = (-1e10, 1e10); shf.dddl = (sh.dddl[0], 1e10); shf.ddds = (-1e10, sh.ddds[1])
    n = 96
    xs, obj, st, it = O.batch_solve(b, shf, 0, n, exact=True, threads=8)
    assert (st > 0).sum() > n // 2
    db = solver.upload(b)
    for label, kw in (("lean", dict(lean=1, cap_iter=-1, split=-1)), ("packed", dict(lean=-1, cap_iter=-1, split=-1)), ("lean two launches", dict(lean=1, cap_iter=5, split=-1)),
                      ("warm entry", dict(lean=1, keep_multipliers=True))):
        o = solver.solve(db, shf, **kw)
        torch.cuda.synchronize()
        r = {k: v.cpu().numpy().copy() for k, v in o.items() if k in ("ctrl", "cost", "status", "iters")}
        check("%s far bounds %s" % (family, label), r, xs, obj, st, n=n)
