"""SURVEY 8(f) rank 4 on the device: obstacle prisms -> per-knot bounds (btrapz_prism_bounds_device) against the CPU
restatement (oracle/prism_oracle.py, itself held to the reference's own `Car` / `get_bounds` output by
tests/test_prism_bounds.py), and the whole chain prisms -> bounds -> corridors -> QP -> arg-min on the GPU."""
import json
import os

import numpy as np
import pytest

from helpers import O
from oracle import prism_oracle as P

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
G = json.load(open(os.path.join(GOLD, "prism_goldens.json")))
W = np.loadtxt(os.path.join(GOLD, "inputs", "weights.txt"))


def pack(scenes, P_max):
    arr = np.zeros((len(scenes), P_max, 8))
    for b, cars in enumerate(scenes):
        for p, c in enumerate(cars):
            arr[b, p, :7] = [c["centre"][0], c["centre"][1], c["centre"][2], c.get("vel_s", 0.0), c.get("vel_l", 0.0), c.get("time", 3.0), 1.0]
    return arr


def random_scenes(n, seed, max_cars=4, nice=False):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        cars = []
        for r in range(int(rng.integers(1, max_cars + 1))):
            ahead = rng.uniform() < 0.5
            q = (lambda v, k: round(float(v), k)) if nice else (lambda v, k: float(v))   # "nice" decimals provoke rounding ties
            cars.append(dict(centre=(q(rng.uniform(5, 40), 1), q(rng.uniform(-3.0, 9.0), 2), 0 if ahead else q(rng.uniform(0.1, 3.0), 1)),
                             vel_s=q(rng.uniform(0, 8), 1) + (0.005 if nice else 0.0), vel_l=float(rng.choice([0.0, 0.25, -0.25])),
                             time=float(rng.choice([3.0, 4.0]))))
        out.append(cars)
    return out


def test_device_bounds_equal_the_restatement_bit_for_bit():
    import torch
    from spectral_amd.solver import BatchSolver
    solver = BatchSolver(0)
    scenes = [sc["cars"] for sc in G["scenes"]] + random_scenes(150, 1) + random_scenes(150, 2, nice=True)
    Pm, N, Omax = 4, 71, 9
    sb, lb, n = solver.prism_bounds(torch.from_numpy(pack(scenes, Pm)), N, Omax)
    torch.cuda.synchronize()
    sb, lb, n = sb.cpu().numpy(), lb.cpu().numpy(), n.cpu().numpy()
    for b, cars in enumerate(scenes):
        want = P.prism_bounds(cars, N)
        assert n[b] == len(want), (b, n[b], len(want))
        for j, (ws, wl) in enumerate(want):
            assert np.array_equal(lb[b, j], np.array(wl)), (b, j)
            assert np.array_equal(sb[b, j], np.array(ws)), (b, j, np.abs(sb[b, j] - np.array(ws)).max())
        assert (lb[b, len(want):] == 1e9).all()                       # padding: no reference can enter
    # more strips than the output holds: flagged, not truncated silently
    sb2, lb2, n2 = solver.prism_bounds(torch.from_numpy(pack(scenes[:8], Pm)), N, 2)
    torch.cuda.synchronize()
    assert (n2.cpu().numpy() == -1).sum() >= 6


def test_reference_goldens_through_the_device():
    """The reference's own output (tests/golden/prism_goldens.json), where it is well formed, equals the device's."""
    import torch
    from spectral_amd.solver import BatchSolver
    solver = BatchSolver(0)
    gl = G["globals"]; N = gl["num_of_knots"]
    scenes = [sc for sc in G["scenes"] if not (any(r["l"][0] >= r["l"][1] for r in sc["strips"]) or
                                                any(sc["strips"][i + 1]["l"][0] < sc["strips"][i]["l"][0] for i in range(len(sc["strips"]) - 1)))]
    assert len(scenes) >= 50
    sb, lb, n = solver.prism_bounds(torch.from_numpy(pack([sc["cars"] for sc in scenes], 3)), N, 7)
    torch.cuda.synchronize()
    sb, lb, n = sb.cpu().numpy(), lb.cpu().numpy(), n.cpu().numpy()
    for b, sc in enumerate(scenes):
        assert n[b] == len(sc["strips"])
        for j, r in enumerate(sc["strips"]):
            want = np.tile(np.array([gl["s_l_l"], gl["s_u_l"]]), (N, 1))
            for i, lo, hi in r["s"]:
                want[i] = (lo, hi)
            assert np.array_equal(sb[b, j], want) and (lb[b, j] == np.array(r["l"])).all()


def test_prisms_to_arg_min_end_to_end():
    """Scene -> strips -> corridors -> QP -> winner, all on the device: every solved candidate's corridor segments
    equal the oracle pipeline's on the generated bounds, its control points the oracle's x*, and the trajectory keeps
    clear of the prisms it was planned around."""
    import torch
    from spectral_amd import synth
    from spectral_amd.solver import BatchSolver
    solver = BatchSolver(0)
    B, N, Omax = 64, 71, 5
    rng = np.random.default_rng(7)
    scenes = []
    for b in range(B):          # the harness's constellation (cart_frenet.py:1536-1546): a slower car ahead in the ego's lane, one beside
        scenes.append([dict(centre=(float(rng.uniform(18, 30)), 1.2, 0), vel_s=float(rng.uniform(3, 5)), vel_l=0.0, time=4.0),
                       dict(centre=(float(rng.uniform(5, 15)), 4.2, 0), vel_s=float(rng.uniform(5, 7)), vel_l=0.0, time=4.0)])
    sb, lb, n = solver.prism_bounds(torch.from_numpy(pack(scenes, 2)), N, Omax)
    tt = np.arange(N) * 0.1
    s_ref = np.tile(40.0 / 7.0 * tt, (B, 1)); l_ref = np.tile(np.clip(1.2 + 0.0825 * (np.arange(N) - 15), 1.2, 4.5), (B, 1))
    init = np.zeros((B, 6)); init[:, 1] = 6.0; init[:, 3] = 1.2
    dsb = np.tile(np.array([0.0, 20.0]), (B, N, 1)); dlb = np.tile(np.array([-3.0, 3.0]), (B, N, 1))
    t = torch.from_numpy
    rec = solver.corridor_batch_tensors(0, N, 0.1, sb, lb, t(dsb), t(dlb), t(s_ref), t(l_ref), t(init), seg_stride=24)
    sh = synth.make_scenario1_batch(1, 7, 0)[1]
    out = solver.solve_ragged(rec, sh)
    bi, bc = solver.argmin(out["cost"])
    torch.cuda.synchronize()
    st = out["status"].cpu().numpy(); cnt = rec["seg_count"].cpu().numpy(); ctrl = out["ctrl"].cpu().numpy()
    assert ((st == 1) | (st == 2)).mean() >= 0.5 and int(bi[0]) >= 0
    sbh, lbh = sb.cpu().numpy(), lb.cpu().numpy()
    p = O.params_from_weights(W)
    checked = 0
    for b in range(0, B, 5):
        lists = [O.corridor_generation(0, N, 0.1, sbh[b, o], lbh[b, o]) for o in range(Omax)]
        nseg, cubes = O.collision_check(0, N, 0.1, lists, s_ref[b], l_ref[b])
        assert cnt[b] == max(nseg, 0)
        if nseg < 1 or st[b] < 1:
            continue
        src = type("S", (), {})()
        src.N, src.delta = N, 0.1
        src.dx_bounds, src.dy_bounds, src.x_ref, src.y_ref = dsb[b], dlb[b], s_ref[b], l_ref[b]
        src.init_s, src.init_l = init[b, :3], init[b, 3:]
        src.ds_ref, src.dl_ref, src.dds, src.ddds, src.ddl, src.dddl = sh.ds_ref, sh.dl_ref, sh.dds, sh.ddds, sh.ddl, sh.dddl
        xs, _, info = O.AssembledQp(0, cubes, p, src).solve_exact()
        if info.status == 1:
            assert np.abs(ctrl[b, :12 * nseg] - xs).max() <= 1e-5 * np.abs(xs).max()
            checked += 1
    assert checked >= 3


def _knot_inputs(B, N, seed):
    rng = np.random.default_rng(seed)
    tt = np.arange(N) * 0.1
    s_ref = (rng.uniform(4.0, 7.0, (B, 1)) * tt[None, :]) + rng.uniform(0, 2, (B, 1))
    l_ref = np.clip(1.2 + rng.uniform(0.02, 0.12, (B, 1)) * (np.arange(N)[None, :] - rng.integers(5, 30, (B, 1))), 1.2, 4.5)
    init = np.zeros((B, 6)); init[:, 1] = 6.0; init[:, 3] = 1.2
    dsb = np.tile(np.array([0.0, 20.0]), (B, N, 1)); dlb = np.tile(np.array([-3.0, 3.0]), (B, N, 1))
    return s_ref, l_ref, init, dsb, dlb


@pytest.mark.parametrize("variant,Pm,N,Omax,seg_stride", [(0, 4, 71, 9, 24), (1, 4, 71, 5, 24), (0, 2, 121, 5, 32), (0, 16, 71, 33, 24),
                                                           (0, 3, 201, 7, 40), (0, 6, 31, 13, 16)])
def test_fused_prism_corridor_stage_equals_the_two_launches_bit_for_bit(variant, Pm, N, Omax, seg_stride):
    """btrapz_prism_corridor_batch_device (strips evaluated inside the corridor kernel) against
    btrapz_prism_bounds_device + btrapz_corridor_batch_device: every field of the batch record, the segment counts, the
    strip counts -- including scenes with more strips than O holds, scenes whose lists overflow the first pass, and
    (O * N > 1536) the shapes the fused entry point routes through its own two launches."""
    import torch
    from spectral_amd.solver import BatchSolver
    solver = BatchSolver(0)
    scenes = random_scenes(300, 11 + N + Pm, max_cars=Pm) + random_scenes(300, 12 + N, max_cars=min(Pm, 4), nice=True)
    if Pm <= 4 and N == 71:
        scenes += [sc["cars"][:Pm] for sc in G["scenes"]]
    B = len(scenes)
    pr = torch.from_numpy(pack(scenes, Pm))
    s_ref, l_ref, init, dsb, dlb = _knot_inputs(B, N, 3)
    t = torch.from_numpy
    sb, lb, n = solver.prism_bounds(pr, N, Omax)
    two = solver.corridor_batch_tensors(variant, N, 0.1, sb, lb, t(dsb), t(dlb), t(s_ref), t(l_ref), t(init), seg_stride=seg_stride)
    one = solver.prism_corridor_batch(variant, pr, N, Omax, 0.1, t(dsb), t(dlb), t(s_ref), t(l_ref), t(init), seg_stride=seg_stride)
    torch.cuda.synchronize()
    assert torch.equal(one["n_strips"], n)
    cnt = two["seg_count"].cpu().numpy()
    assert torch.equal(one["seg_count"], two["seg_count"]), (cnt != one["seg_count"].cpu().numpy()).sum()
    assert Omax > 13 or (cnt > 0).mean() > 0.3      # (33 strips leave 4 list slots per strip: every scene overflows, in both)
    a, b = one["seg"].cpu().numpy(), two["seg"].cpu().numpy()
    assert a.tobytes() == b.tobytes(), np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))))[:5]
    assert torch.equal(one["ref_end"], two["ref_end"]) and torch.equal(one["dl_bounds"], two["dl_bounds"])
