"""Horizons beyond the wave-wide kernels' shapes, from knot level to control points.  The reference has no
limit on knots or segments (std::vector throughout: CorridorGeneration src/solve_3d.cc:323-486, CorridorSplit :729-772,
CollisionCheck :488-714); until round 6 the device corridor stage stopped at 512 knots / 64 selected segments and ragged batches
at 64 segments.  Now: more than 512 knots -> corridor_serial_kernel (one lane per candidate, the serial statements of
corridor_core.h on lists in a workspace); candidates of 65..256 segments in a ragged batch -> the long form, one launch per
count.  Held to the oracle: the corridor record bit for bit, the control points to 1e-5 of x*."""
import os

import numpy as np
import pytest

from helpers import O
from spectral_amd import knots, layout as L, synth

pytestmark = pytest.mark.gpu

FIELDS = [(L.F_T, "t"), (L.F_DOWN_BIAS, "down_bias"), (L.F_DOWN_SKEW, "down_skew"), (L.F_UPP_BIAS, "upp_bias"), (L.F_UPP_SKEW, "upp_skew"),
          (L.F_L_DOWN_BIAS, "l_down_bias"), (L.F_L_DOWN_SKEW, "l_down_skew"), (L.F_L_UPP_BIAS, "l_upp_bias"), (L.F_L_UPP_SKEW, "l_upp_skew"),
          (L.F_BEG_L, "beg_l"), (L.F_END_L, "end_l")]


def shared_for(kb, variant):
    sh = synth.shared_params(variant, weights=synth.REFERENCE_WEIGHTS)
    h = kb.header
    sh.ds_ref, sh.dl_ref, sh.dds, sh.ddds, sh.ddl, sh.dddl = h["ds_ref"], h["dl_ref"], h["dds"], h["ddds"], h["ddl"], h["dddl"]
    return sh


@pytest.mark.parametrize("variant", [0, 1])
def test_a_100_segment_jittered_batch_from_knots_to_control_points(variant, tmp_path):
    import torch
    from spectral_amd.solver import BatchSolver
    solver = BatchSolver(0)
    B = 24
    kb = synth.scenario1_knots(B, 100)                       # N = 1 001 knots, two lane corridors, per-candidate events: ragged
    assert kb.N == 1001
    rec = solver.corridor_batch(kb, variant, seg_stride=160)
    sh = shared_for(kb, variant)
    o = solver.solve_ragged(rec, sh)
    torch.cuda.synchronize()
    counts = rec["seg_count"].cpu().numpy()
    assert counts.min() > 64 and counts.max() <= 160 and len(set(counts.tolist())) > 1, counts      # long, and ragged
    assert solver.ctx.last_solve_form() & 16                                                        # the long form served them
    seg = rec["seg"].cpu().numpy(); ctrl = o["ctrl"].cpu().numpy(); status = o["status"].cpu().numpy(); cost = o["cost"].cpu().numpy()
    checked = 0
    for b in (0, 11, 23):
        path = str(tmp_path / ("c%d.txt" % b))
        knots.write_corridor_file(path, kb, b)
        inp = O.ParsedInput(path)
        n, cubes = O.pipeline(variant, inp)
        assert n == counts[b]
        for f, name in FIELDS:                                # the record of the device stage: the oracle's cubes, bit for bit
            if variant == 1 and name.startswith("l_"):
                continue                                      # (the cuboid variant leaves the l lines at their defaults)
            assert np.array_equal(seg[f, b, :n], np.array([getattr(c, name) for c in cubes])), (b, name)
        qp = O.AssembledQp(variant, cubes, O.params_from_weights(synth.REFERENCE_WEIGHTS), inp)
        x, _, info = qp.solve_exact(max_iter=120)
        assert (info.status in (1, 2)) == (status[b] > 0), (b, info.status, status[b])
        if status[b] > 0:
            got = ctrl[b, :12 * n]
            assert np.abs(got - x).max() <= 1e-5 * np.abs(x).max(), (b, np.abs(got - x).max() / np.abs(x).max())
            P, _ = qp.dense()
            assert abs(cost[b] - (0.5 * x @ P @ x + qp.q @ x)) <= 1e-6 * abs(cost[b])
            checked += 1
    assert checked >= 2 or variant == 1
    assert (status > 0).sum() >= (B // 2 if variant == 0 else 0)


def test_long_and_short_candidates_share_a_ragged_batch():
    """One ragged batch, slots for 160 segments: candidates of 70..120 segments (long form, one launch per count) beside
    candidates of 30 and 64 (the bucketed kernel).  Every candidate's result is what it is in a batch of its own kind."""
    import torch
    from spectral_amd.solver import BatchSolver
    solver = BatchSolver(0)
    B = 16
    kb = synth.scenario1_knots(B, 100)
    rec = solver.corridor_batch(kb, 0, seg_stride=160)
    sh = shared_for(kb, 0)
    torch.cuda.synchronize()
    counts = rec["seg_count"].cpu().numpy().copy()
    cut = counts.copy()
    cut[0::4] = 30; cut[1::4] = 64; cut[2::4] = np.minimum(counts[2::4], 70)          # a corridor cut short is a corridor
    rec2 = dict(rec); rec2["seg_count"] = torch.from_numpy(cut.astype(np.int32)).to(rec["seg"].device)
    o = solver.solve_ragged(rec2, sh)
    torch.cuda.synchronize()
    assert solver.ctx.last_solve_form() & 16
    st = o["status"].cpu().numpy(); ctrl = o["ctrl"].cpu().numpy()
    assert (st != -4).all() and (st > 0).sum() >= B // 2                              # nobody is "no usable corridor"
    # the short ones alone, in slots of 64: the same kernel family -> the same bits
    short = np.nonzero(cut <= 64)[0]
    rec3 = dict(B=len(short), seg_stride=64, seg=rec["seg"][:, short, :64].contiguous(), seg_count=rec2["seg_count"][short].contiguous(),
                init=rec["init"][short].contiguous(), ref_end=rec["ref_end"][short].contiguous(), dl_bounds=rec["dl_bounds"][short].contiguous())
    o3 = solver.solve_ragged(rec3, sh, lean=-1)
    torch.cuda.synchronize()
    st3, ctrl3 = o3["status"].cpu().numpy(), o3["ctrl"].cpu().numpy()
    assert np.array_equal(st[short], st3)
    for j, b in enumerate(short):
        if st3[j] > 0:
            n = int(cut[b])
            assert np.abs(ctrl[b, :12 * n] - ctrl3[j, :12 * n]).max() <= 1e-9 * np.abs(ctrl3[j, :12 * n]).max()
    # the long ones alone, as uniform batches of their count (the long form's own entry): bit for bit
    from spectral_amd.layout import Batch
    segs = rec["seg"].cpu().numpy(); init = rec["init"].cpu().numpy(); ref_end = rec["ref_end"].cpu().numpy(); dlb = rec["dl_bounds"].cpu().numpy()
    for b in np.nonzero(cut > 64)[0][:4]:
        n = int(cut[b])
        one = Batch(B=1, S=n, seg=np.ascontiguousarray(segs[:, b:b + 1, :n]), init=init[b:b + 1].copy(), ref_end=ref_end[b:b + 1].copy(), dl_bounds=dlb[b:b + 1].copy())
        o1 = solver.solve(solver.upload(one), sh)
        torch.cuda.synchronize()
        assert int(o1["status"][0].item()) == st[b]
        if st[b] > 0:
            assert np.array_equal(o1["ctrl"][0].cpu().numpy(), ctrl[b, :12 * n])


def test_more_obstacles_than_lanes():
    """65+ obstacle corridors (the wave-wide kernels give one lane to each): the serial kernel, against the wave-wide result of
    the same scene with its obstacles repeated -- a repeated corridor's segments are exact duplicates, which CollisionCheck's
    de-dup removes (solve_3d.cc:617-637), but they DO advance its running hit counter: so the comparison is with the oracle."""
    import tempfile
    import torch
    from spectral_amd.solver import BatchSolver
    solver = BatchSolver(0)
    gold = os.path.join(os.path.dirname(__file__), "golden", "inputs")
    kb0 = knots.jittered(knots.parse_corridor_file(os.path.join(gold, "c_road_s1_3.txt")), 6, seed=2)
    reps = 22                                                 # 3 obstacles x 22 = 66 > 64
    kb = knots.KnotBatch(kb0.B, kb0.N, kb0.num_obs * reps, kb0.delta, np.tile(kb0.s_bounds, (1, reps, 1, 1)), np.tile(kb0.l_bounds, (1, reps, 1, 1)),
                         kb0.ds_bounds, kb0.dl_bounds, kb0.s_ref, kb0.l_ref, kb0.init, dict(kb0.header))
    rec = solver.corridor_batch(kb, 0, seg_stride=64)
    torch.cuda.synchronize()
    counts = rec["seg_count"].cpu().numpy(); seg = rec["seg"].cpu().numpy()
    d = tempfile.mkdtemp()
    for b in range(kb.B):
        path = os.path.join(d, "m%d.txt" % b)
        knots.write_corridor_file(path, kb, b)
        inp = O.ParsedInput(path)
        n, cubes = O.pipeline(0, inp)
        assert n == counts[b], (b, n, counts[b])
        for f, name in FIELDS:
            assert np.array_equal(seg[f, b, :max(n, 0)], np.array([getattr(c, name) for c in cubes[:max(n, 0)]])), (b, name)


@pytest.mark.parametrize("N,num_obs", [(513, 3), (700, 2), (1001, 1), (1500, 2), (71, 65), (130, 90)])
def test_serial_corridor_kernel_equals_the_oracle_on_random_inputs(N, num_obs):
    """corridor_serial_kernel on the fuzz family of the wave-wide kernels' test (slope changes in runs, breaks above and below
    the 0.2 threshold, moving l bounds, collapsed bounds, nan / inf entries, a nan reference knot), at shapes only it serves:
    count and every field of the record as the oracle's corridor stage computes them, both variants."""
    import torch
    from helpers import fuzz_knot_batch, oracle_corridor
    from spectral_amd.solver import BatchSolver
    solver = BatchSolver(0)
    stride = 256
    checked = usable = 0
    for rd in range(3):
        kb = fuzz_knot_batch(9000 + 10 * N + rd, B=6, N=N, num_obs=num_obs)
        for variant in (0, 1):
            rec = solver.corridor_batch(kb, variant, seg_stride=stride)
            torch.cuda.synchronize()
            seg = rec["seg"].cpu().numpy(); cnt = rec["seg_count"].cpu().numpy()
            for b in range(kb.B):
                n, cubes = oracle_corridor(kb, b, variant)
                checked += 1
                want = n if n > 0 else 0
                if n > stride or any(not (c.t > 0) for c in cubes):
                    want = -1
                assert cnt[b] == want, (N, num_obs, rd, variant, b, cnt[b], want)
                if want <= 0:
                    continue
                usable += 1
                for k, c in enumerate(cubes):
                    for f, attr in FIELDS:
                        got, exp = seg[f, b, k], getattr(c, attr)
                        assert got == exp or (np.isnan(got) and np.isnan(exp)), (N, num_obs, rd, variant, b, k, attr, got, exp)
                    lo, hi = 0.0, 1000.0                        # solve_3d.cc:835-841
                    for i in range(c.beg_t, c.end_t + 1):
                        ii = min(max(i, 0), N - 1)
                        lo = np.fmax(kb.ds_bounds[b, ii, 0], lo); hi = np.fmin(kb.ds_bounds[b, ii, 1], hi)
                    i0, i1 = min(10 * k, N - 1), min(10 * k + 1, N - 1)
                    derived = {L.F_DS_LO: lo, L.F_DS_HI: hi, L.F_X_SKEW: (kb.s_ref[b, i1] - kb.s_ref[b, i0]) / kb.delta,
                               L.F_X_BIAS: kb.s_ref[b, i0], L.F_Y_SKEW: (kb.l_ref[b, i1] - kb.l_ref[b, i0]) / kb.delta, L.F_Y_BIAS: kb.l_ref[b, i0]}
                    for f, exp in derived.items():
                        got = seg[f, b, k]
                        assert got == exp or (np.isnan(got) and np.isnan(exp)), (N, num_obs, rd, variant, b, k, f, got, exp)
    assert checked == 36 and usable >= 6
