"""btrapz_options.compact: the pre-pass that keeps candidates which cannot start out of the solve launch (an empty
inscribed interval of the cuboid variant, src/cuboid_3d.cc:677-689; an initial state outside segment 0's rows; a joint
whose two sides share no value -- the set-up checks of the solve kernels, applied once, in front).  Scheduling only: every
candidate keeps its status and cost, every solved one its control points and iteration count, bit for bit, in every form
of the launch; what changes is that the launch holds live groups only (seen here through the per-axis records: the other
axis of a dropped candidate is no longer solved)."""
import numpy as np
import pytest

from spectral_amd import layout as L
from spectral_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def solver():
    from spectral_amd.solver import BatchSolver
    return BatchSolver(0)


def damaged(batch, seed=3):
    """A third of the candidates made unstartable in the three ways the set-up knows, plus some garbage."""
    import copy
    rng = np.random.default_rng(seed)
    b = copy.copy(batch)
    b.seg = batch.seg.copy(); b.init = batch.init.copy(); b.dl_bounds = batch.dl_bounds.copy()
    pick = rng.choice(batch.B, batch.B // 3, replace=False)
    for i, c in enumerate(pick):
        k = int(rng.integers(batch.S))
        kind = i % 6
        if kind == 0: b.init[c, 0] += 500.0                                           # initial position outside segment 0
        elif kind == 1: b.init[c, 4] = 50.0                                           # initial lateral velocity outside its row
        elif kind == 2: b.seg[L.F_UPP_BIAS, c, k] = b.seg[L.F_DOWN_BIAS, c, k] - 2.0  # l > u
        elif kind == 3 and k + 1 < batch.S:                                           # a joint with no common value
            b.seg[L.F_L_DOWN_BIAS, c, k + 1] = b.seg[L.F_L_UPP_BIAS, c, k] + b.seg[L.F_L_UPP_SKEW, c, k] + 3.0
            b.seg[L.F_L_UPP_BIAS, c, k + 1] = b.seg[L.F_L_DOWN_BIAS, c, k + 1] + 1.0
            b.seg[L.F_BEG_L, c, k + 1] = b.seg[L.F_L_DOWN_BIAS, c, k + 1]; b.seg[L.F_END_L, c, k + 1] = b.seg[L.F_L_UPP_BIAS, c, k + 1]
        elif kind == 4: b.seg[L.F_T, c, k] = [0.0, -1.0, np.nan][i % 3]
        else: b.seg[L.F_DS_HI, c, k] = [np.nan, np.inf, -5.0][i % 3]
    return b


FORMS = [("lean", dict(lean=1, cap_iter=-1, split=-1)), ("lean two launches", dict(lean=1, cap_iter=5, split=-1)),
         ("packed", dict(lean=-1, cap_iter=-1, split=-1)), ("packed two launches", dict(lean=-1, cap_iter=5, split=-1))]


def _dmg(made, seed):
    return damaged(made[0], seed), made[1]


@pytest.mark.parametrize("make", [lambda: synth.make_scenario1_batch(6144, 20, 1), lambda: _dmg(synth.make_scenario1_batch(6144, 20, 0), 3),
                                  lambda: _dmg(synth.make_batch(4096, 10, config=2), 4), lambda: _dmg(synth.make_batch(900, 33, config=2, variant=1), 5)],
                         ids=["cuboid bench batch", "scenario_1 damaged", "10 segments damaged", "33 segments cuboid damaged"])
def test_compaction_changes_no_result(solver, make):
    import torch
    batch, sh = make()
    db = solver.upload(batch)
    for label, kw in FORMS:
        res = {}
        for compact in (-1, 1):
            o = solver.solve(db, sh, compact=compact, **kw)
            torch.cuda.synchronize()
            res[compact] = ({k: v.cpu().numpy().copy() for k, v in o.items()}, solver.ctx.last_solve_form(), solver.ctx.debug_axis_records(batch.B))
        (a, fa, (ita, sta)), (c, fc, (itc, stc)) = res[-1], res[1]
        assert fa == fc, label                                                    # the same kernels either way
        diff = a["status"] != c["status"]
        assert not diff.any(), (label, np.nonzero(diff)[0][:8], a["status"][diff][:8], c["status"][diff][:8])
        assert np.array_equal(a["cost"], c["cost"]), label
        ok = a["status"] > 0
        assert ok.sum() > batch.B // 3 and (~ok).sum() > batch.B // 8, label       # both kinds are there
        assert np.array_equal(a["ctrl"][ok], c["ctrl"][ok]) and np.array_equal(a["iters"][ok], c["iters"][ok]), label
        # the pre-pass ran: axis problems of dropped candidates -- solved by the plain launch -- were not looked at (status 0)
        skipped = stc == 0
        assert skipped.sum() > batch.B // 16 and (sta[skipped] != 0).all(), label
        assert (c["status"][skipped.any(axis=1)] < 0).all(), label
        # ... and their control points are NaN, written by the pre-pass (no solve kernel ever sees them: ADVICE r5) -- a
        # warm start refuses such an x0 instead of starting from whatever the buffer held
        assert np.isnan(c["ctrl"][skipped.any(axis=1)]).all(), label


def test_compaction_on_ragged_batches(solver):
    """knots -> corridors -> ragged record (jittered c_road_s1_3.txt: a quarter of the candidates have no solution, most
    of them visibly so before the first iteration): the ragged launch with and without the pre-pass."""
    import os
    import torch
    from spectral_amd import knots
    gold = os.path.join(os.path.dirname(__file__), "golden", "inputs")
    W = np.loadtxt(os.path.join(gold, "weights.txt"))
    for name, variant in (("c_road_s1_3", 0), ("c2", 1), ("c1", 1)):    # (c_road_s1_3 has no cuboid corridor at all)
        kb = knots.jittered(knots.parse_corridor_file(os.path.join(gold, name + ".txt")), 4096, seed=21, s_shift=1.0)
        kb.init[::7, 3] += 40.0                                   # every seventh ego starts outside its (lateral) corridor
        sh = synth.shared_params(variant, weights=W)
        sh.ds_ref, sh.dl_ref = kb.header["ds_ref"], kb.header["dl_ref"]
        sh.dds, sh.ddds, sh.ddl, sh.dddl = kb.header["dds"], kb.header["ddds"], kb.header["ddl"], kb.header["dddl"]
        rec = solver.corridor_batch(kb, variant, seg_stride=32)
        for kw in (dict(lean=1, cap_iter=-1), dict(lean=-1, cap_iter=-1), dict(lean=1, cap_iter=6)):
            r = {}
            for compact in (-1, 1):
                o = solver.solve_ragged(rec, sh, compact=compact, **kw)
                torch.cuda.synchronize()
                r[compact] = {k: v.cpu().numpy().copy() for k, v in o.items()}
            a, c = r[-1], r[1]
            assert np.array_equal(a["status"], c["status"]) and np.array_equal(a["cost"], c["cost"]), (name, variant, kw)
            ok = a["status"] > 0
            assert ok.any() and (~ok).sum() > 400
            assert np.array_equal(a["ctrl"][ok], c["ctrl"][ok]) and np.array_equal(a["iters"][ok], c["iters"][ok])


def test_automatic_choice(solver):
    """compact = 0: on for large batches of the cuboid variant and large ragged batches, off elsewhere -- and never with
    a rescue pass or a warm start (their kernels need every axis's record)."""
    import torch
    batch, sh = synth.make_scenario1_batch(24576, 20, 1)
    db = solver.upload(batch)
    solver.solve(db, sh)
    torch.cuda.synchronize()
    it, st = solver.ctx.debug_axis_records(batch.B)
    assert (st == 0).any()                              # cuboid, large: the pre-pass ran
    solver.solve(db, sh, elastic=1)
    torch.cuda.synchronize()
    it, st = solver.ctx.debug_axis_records(batch.B)
    assert not (st == 0).any()                          # rescue pass: every axis problem has a record
    b2, sh2 = synth.make_scenario1_batch(24576, 20, 0)
    solver.solve(solver.upload(b2), sh2)
    torch.cuda.synchronize()
    it, st = solver.ctx.debug_axis_records(b2.B)
    assert not (st == 0).any()                          # trapezoid: off by default
