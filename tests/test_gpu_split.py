"""The split form of the solve kernel (btrapz_options.split; SPLIT in spectral_amd/csrc/btrapz_kernels.hip): one candidate
per wavefront, every segment's fifteen rows spread over three lanes.  Same QP, same method, same decisions as the
three-candidates-per-wavefront form -- held to it and to the oracle's x* (solve_3d.cc:1246-1249 is what both replace)."""
import os

import numpy as np
import pytest

from helpers import O, oracle_qp_from_batch
from spectral_amd import knots, native, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
W = np.loadtxt(os.path.join(GOLD, "inputs", "weights.txt"))


def _solve(solver, db, sh, split):
    import torch
    o = solver.solve(db, sh, split=split)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy().copy() for k, v in o.items()}


@pytest.mark.parametrize("gen,S,variant,B", [("scenario1", 20, 0, 768), ("scenario1", 20, 1, 384), ("generic", 10, 0, 512),
                                             ("scenario1", 7, 0, 300), ("generic", 21, 0, 96), ("generic", 1, 0, 40),
                                             ("generic", 2, 0, 40), ("generic", 3, 1, 40)])
def test_split_form_decides_and_solves_as_the_packed_form(gen, S, variant, B):
    from spectral_amd.solver import BatchSolver
    solver = BatchSolver(0)
    batch, sh = (synth.make_scenario1_batch(B, S, variant) if gen == "scenario1" else synth.make_batch(B, S, config=2, variant=variant))
    db = solver.upload(batch)
    a = _solve(solver, db, sh, -1)
    b = _solve(solver, db, sh, 1)
    assert np.array_equal(a["status"] > 0, b["status"] > 0)                 # the same candidates accepted
    assert np.array_equal(a["status"], b["status"])
    ok = a["status"] > 0
    assert ok.any()
    scale = np.abs(a["ctrl"][ok]).max(axis=1, keepdims=True)
    assert (np.abs(a["ctrl"][ok] - b["ctrl"][ok]) / scale).max() <= 2e-6  # both within 1e-6 of x*, see below
    assert np.abs(a["cost"][ok] - b["cost"][ok]).max() <= 1e-7 * np.abs(a["cost"][ok]).max()
    # same method, same termination tests: iteration counts differ only where rounding moves a score across a threshold
    assert (np.abs(a["iters"] - b["iters"]) <= 1).mean() >= 0.99
    # and against the oracle's exact solve
    xs, obj, st, _ = O.batch_solve(batch, sh, 0, 6, exact=True)
    for i in range(6):
        assert (st[i] in (1, 2)) == bool(ok[i])
        if ok[i]:
            assert np.abs(b["ctrl"][i] - xs[i]).max() <= 1e-5 * np.abs(xs[i]).max()


def test_automatic_choice_takes_the_split_form_for_few_candidates_only():
    """split = 0: a batch that leaves SIMDs idle (2 B wavefronts fit the device) runs the split form, a large one the
    packed form; more than 21 segments always the packed form.  Seen through the results: bit-equal to the forced form."""
    from spectral_amd.solver import BatchSolver
    solver = BatchSolver(0)
    batch, sh = synth.make_scenario1_batch(64, 20, 0)
    db = solver.upload(batch)
    auto, split, packed = _solve(solver, db, sh, 0), _solve(solver, db, sh, 1), _solve(solver, db, sh, -1)
    assert np.array_equal(auto["ctrl"], split["ctrl"]) and not np.array_equal(auto["ctrl"], packed["ctrl"])
    big, sh2 = synth.make_batch(4096, 10, config=2)
    db2 = solver.upload(big)
    auto, packed = _solve(solver, db2, sh2, 0), _solve(solver, db2, sh2, -1)
    assert np.array_equal(auto["ctrl"], packed["ctrl"])
    wide, sh3 = synth.make_batch(30, 22, config=2)
    db3 = solver.upload(wide)
    forced, packed = _solve(solver, db3, sh3, 1), _solve(solver, db3, sh3, -1)
    assert np.array_equal(forced["ctrl"], packed["ctrl"])                 # 22 segments: three lanes per segment do not fit


@pytest.mark.parametrize("name", ["c1", "c2", "c6", "c7", "c_road_s1", "c_road_s1_3"])
def test_find_traj_through_the_split_form(name, monkeypatch):
    """find_traj's single launch uses the split form up to 21 segments (BTRAPZ_SPLIT=0: the packed one): same decision,
    same trajectory, and the oracle's x*."""
    params = native.CParams(*[float(v) for v in W], 3)
    kb = knots.parse_corridor_file(os.path.join(GOLD, "inputs", name + ".txt"))
    for variant in (0, 1):
        monkeypatch.setenv("BTRAPZ_SPLIT", "0")
        c0, t0, x0 = native.find_traj_mem(variant, params, kb)
        it0 = native.lib().btrapz_find_traj_last_iterations()
        monkeypatch.delenv("BTRAPZ_SPLIT")
        c1, t1, x1 = native.find_traj_mem(variant, params, kb)
        it1 = native.lib().btrapz_find_traj_last_iterations()
        assert (c0 == 1e11) == (c1 == 1e11)
        if c0 == 1e11:
            continue
        # (two forms of one method: the same optimum to the solve's tolerance; the costs of c_road_s1 differ by 1.2e-7)
        assert abs(c0 - c1) <= 5e-7 * abs(c0) and np.abs(x0 - x1).max() <= 2e-6 * np.abs(x0).max()
        assert np.abs(t0 - t1).max() <= 2e-6 * max(1.0, np.abs(t0).max())
        assert it1 <= it0 + 3                      # (c_road_s1 sits at the round-off floor: 18 packed / 13 split in round 3, 13 / 15 since round 4's start and corrector)


@pytest.mark.parametrize("gen,S,variant,split", [("scenario1", 20, 0, -1), ("generic", 10, 0, -1), ("scenario1", 20, 1, 1), ("generic", 7, 0, 1)])
def test_unconstrained_start_reaches_the_same_optimum(gen, S, variant, split):
    """btrapz_options.start = 1 (one Newton step of the problem without its inequality rows before the first iteration):
    another starting point of the same strictly convex QP -- same candidates accepted, same optimum."""
    import torch
    from spectral_amd import native
    from spectral_amd.solver import BatchSolver
    if not native.lib().btrapz_build_has_experiments():
        pytest.skip("btrapz_options.start is honoured by -DBTRAPZ_EXPERIMENTS builds only (a measured loss: DESIGN 3.2)")
    solver = BatchSolver(0)
    B = 600
    batch, sh = (synth.make_scenario1_batch(B, S, variant) if gen == "scenario1" else synth.make_batch(B, S, config=2, variant=variant))
    db = solver.upload(batch)
    a = {k: v.cpu().numpy().copy() for k, v in solver.solve(db, sh, split=split).items()}
    b = {k: v.cpu().numpy().copy() for k, v in solver.solve(db, sh, split=split, start=1).items()}
    torch.cuda.synchronize()
    assert np.array_equal(a["status"] > 0, b["status"] > 0)
    ok = a["status"] > 0
    assert (np.abs(a["ctrl"][ok] - b["ctrl"][ok]).max(axis=1) / np.abs(a["ctrl"][ok]).max(axis=1)).max() <= 5e-6
    assert not np.array_equal(a["iters"], b["iters"])          # (it IS another path)
    xs, obj, st, _ = O.batch_solve(batch, sh, 0, 4, exact=True)
    for i in range(4):
        if ok[i]:
            assert np.abs(b["ctrl"][i] - xs[i]).max() <= 1e-5 * np.abs(xs[i]).max()


@pytest.mark.parametrize("gen,S,variant,cap", [("scenario1", 20, 0, 6), ("scenario1", 20, 1, 5), ("generic", 20, 0, 4), ("scenario1", 10, 0, 7),
                                               ("generic", 33, 0, 3), ("scenario1", 20, 0, 1)])
def test_capped_first_launch_and_resume_launch_give_the_one_launch_results_bit_for_bit(gen, S, variant, cap):
    """btrapz_options.cap_iter: a first launch in which a candidate left alone in its wavefront after cap iterations (or
    still iterating four iterations later) hands its iterate over, and a second launch that carries those candidates
    on from exactly that iterate.  Scheduling only: control points, costs, statuses and iteration counts are the
    one-launch solve's, bit for bit."""
    import torch
    from spectral_amd.solver import BatchSolver
    solver = BatchSolver(0)
    B = 3000
    batch, sh = (synth.make_scenario1_batch(B, S, variant) if gen == "scenario1" else synth.make_batch(B, S, config=3, variant=variant))
    db = solver.upload(batch)
    a = {k: v.cpu().numpy().copy() for k, v in solver.solve(db, sh, split=-1, cap_iter=-1).items()}
    b = {k: v.cpu().numpy().copy() for k, v in solver.solve(db, sh, split=-1, cap_iter=cap).items()}
    torch.cuda.synchronize()
    assert np.array_equal(a["status"], b["status"]) and np.array_equal(a["iters"], b["iters"]) and np.array_equal(a["cost"], b["cost"])
    ok = a["status"] > 0
    assert ok.any() and np.array_equal(a["ctrl"][ok], b["ctrl"][ok])
    assert (a["iters"] + 1 > cap + 4).any()          # (some candidates did go through the second launch)
    # the per-axis records behind the candidates' (debug hooks): a candidate's count is its slower axis's, and only
    # axis problems that were still iterating at the hand-over point carry a key of the resume lists
    it_ax, st_ax = solver.ctx.debug_axis_records(B)
    keys = solver.ctx.debug_resume_keys(B)
    assert np.array_equal(it_ax.max(axis=1)[ok], b["iters"][ok]) and ((st_ax > 0).all(axis=1) == ok).all()
    assert (keys >= 0).all() and (keys <= 64).all() and (keys > 0).sum() > 0
    assert (it_ax.T[keys > 0] + 1 > cap).all()      # (not every long runner has one: the hand-over slots are bounded)
    # with the rescue pass behind it, too
    c = {k: v.cpu().numpy().copy() for k, v in solver.solve(db, sh, split=-1, cap_iter=cap, elastic=1).items()}
    d = {k: v.cpu().numpy().copy() for k, v in solver.solve(db, sh, split=-1, cap_iter=-1, elastic=1).items()}
    torch.cuda.synchronize()
    assert np.array_equal(c["status"], d["status"]) and np.array_equal(c["cost"], d["cost"])


@pytest.mark.parametrize("cap", [4, 8])
def test_ragged_batches_in_two_launches_give_the_one_launch_results_bit_for_bit(cap):
    """btrapz_options.cap_iter on a ragged batch (knots -> corridors -> QP): the first launch hands over, the resume
    lists are bucketed by segment count; control points, costs, statuses, iteration counts are the one-launch solve's."""
    import os
    import torch
    from spectral_amd import knots
    from spectral_amd.solver import BatchSolver
    solver = BatchSolver(0)
    gold = os.path.join(os.path.dirname(__file__), "golden", "inputs")
    W = np.loadtxt(os.path.join(gold, "weights.txt"))
    parts = [knots.jittered(knots.parse_corridor_file(os.path.join(gold, n + ".txt")), 1500, seed=3 + i) for i, n in enumerate(("c_road_s1_3", "c1", "c2"))]
    listed = 0
    for kb in parts + [synth.scenario1_knots(1200, 20)]:
        sh = synth.shared_params(0, weights=W)
        sh.ds_ref, sh.dl_ref = kb.header["ds_ref"], kb.header["dl_ref"]
        sh.dds, sh.ddds, sh.ddl, sh.dddl = kb.header["dds"], kb.header["ddds"], kb.header["ddl"], kb.header["dddl"]
        rec = solver.corridor_batch(kb, 0, seg_stride=32)
        a = {k: v.cpu().numpy().copy() for k, v in solver.solve_ragged(rec, sh, cap_iter=-1, lean=-1).items()}
        assert solver.ctx.last_solve_form() == 0
        b = {k: v.cpu().numpy().copy() for k, v in solver.solve_ragged(rec, sh, cap_iter=cap, lean=-1).items()}
        assert solver.ctx.last_solve_form() == 3
        torch.cuda.synchronize()
        assert np.array_equal(a["status"], b["status"]) and np.array_equal(a["iters"], b["iters"]) and np.array_equal(a["cost"], b["cost"])
        ok = a["status"] > 0
        assert ok.any() and np.array_equal(a["ctrl"][ok], b["ctrl"][ok])
        keys = solver.ctx.debug_resume_keys(kb.B)
        cnt = rec["seg_count"].cpu().numpy()
        assert (keys[keys > 0] == np.stack([cnt, cnt])[keys > 0]).all()   # listed by segment count
        listed += int((keys > 0).sum())
    assert listed > 0                                                     # (some problems did go through the second launch)
