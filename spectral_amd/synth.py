"""Synthetic candidate-corridor batches (the workloads of BASELINE.json configs 2-5).

The reference ships no generator (its corridors come from the CommonRoad harness,
cart_frenet.py:833-1030, 384-453).  Two generators, both at the level of the hot path's
input record -- one `Cube` (include/btrapz/cube_type.h:2-24) per segment:

  make_scenario1_batch  BASELINE config 3 / 4: the corridor of the bundled scenario_1 input
                        (src/c1.txt) tiled to S one-second segments, with per-candidate jitter
                        (SURVEY 8d).  scenario1_knots() is the same scene at knot level -- the
                        content of a corridor text file -- for the device corridor stage.
  make_batch            a generic family (smooth random speed profile, random margins and
                        ramps, one lane change), feasible by construction: config 2, config 5
                        and a second figure beside config 3.
"""
import numpy as np

from .layout import (F_T, F_DOWN_BIAS, F_DOWN_SKEW, F_UPP_BIAS, F_UPP_SKEW, F_L_DOWN_BIAS,
                     F_L_DOWN_SKEW, F_L_UPP_BIAS, F_L_UPP_SKEW, F_BEG_L, F_END_L, F_DS_LO, F_DS_HI,
                     F_X_SKEW, F_X_BIAS, F_Y_SKEW, F_Y_BIAS, NUM_SEG_FIELDS, Batch, Shared)

# src/weights.txt (the one Optuna-tuned row the harness loads, trp_wrapper.py:100-115)
REFERENCE_WEIGHTS = (35.73, 41.61, 25.57, 41.59, 0.12, 10.04, 0.71, 14.3, 7.27, 32.13)
SEED_BASE = 0x5EC7A100


def shared_params(variant=0, weights=REFERENCE_WEIGHTS, delta=0.1):
    """Limits from src/c_road_s1_2.txt; weights.txt order == Params order."""
    w = [float(v) for v in weights]
    return Shared(w_s=(w[4], w[5], w[0], w[1]), w_l=(w[6], w[7], w[2], w[3]),
                  weight_end_s=w[8], weight_end_l=w[9], ds_ref=7.0, dl_ref=0.0,
                  dds=(-2.0, 2.0), ddds=(-30.0, 30.0), ddl=(-0.7, 0.7), dddl=(-10.0, 10.0),
                  delta=delta, variant=variant)


def _smoothstep5(x):
    x = np.clip(x, 0.0, 1.0)
    return x * x * x * (10.0 + x * (-15.0 + 6.0 * x))


# ---- scenario_1 (src/c1.txt), tiled ----------------------------------------------------------------------------------
# c1.txt: N = 71 knots of 0.1 s; two lane corridors, l in (1, 3) and (3, 4.5); in BOTH lanes the upper s bound drops
# from "free" (50) to a ramp 19 -> 25 m (3 m/s) over knots 20..40; in the second lane the lower s bound is a ramp
# 10 -> 20 m (5 m/s) over knots 30..50; the reference trajectory runs at 40 m / 7 s and changes lane (l_ref 1.2 -> 4.5
# at 0.0825 per knot, crossing the lane line near knot 37), so the selected corridor (CollisionCheck) is lane 1, then
# lane 2.  Header: ds_ref 10, dl_ref 1, dds (-3, 2), ddds +-30, ddl +-2, dddl +-20, ds in (0, 20), dl in (0, 3).
# Tiled to a horizon of S seconds: the events repeat every 10 s, 57.1 m further on, the lane change alternates
# direction (hence dl in (-3, 3)); per candidate the onset of every obstacle event moves by -1 / 0 / +1 s, its speed by
# +-1 m/s, and v0 ~ U(5, 9).
C1_ROUTE_SPEED = 40.0 / 7.0
C1_HEADER = dict(ds_ref=10.0, dl_ref=1.0, dds=(-3.0, 2.0), ddds=(-30.0, 30.0), ddl=(-2.0, 2.0), dddl=(-20.0, 20.0))


def _scenario1_events(rng, B, S, scale):
    """Per-candidate obstacle events of every 10-s pattern: (onset second, speed) of the upper ramp (both lanes) and
    of the lower ramp (second lane), and the initial speed."""
    npat = (S + 9) // 10
    ev = dict(on_hi=2 + rng.integers(-1, 2, (B, npat)), sp_hi=scale * (3.0 + rng.uniform(-1, 1, (B, npat))),
              on_lo=3 + rng.integers(-1, 2, (B, npat)), sp_lo=scale * (5.0 + rng.uniform(-1, 1, (B, npat))),
              v0=scale * rng.uniform(5.0, 9.0, B))
    return npat, ev


def _scenario1_lref(N):
    """c1's lateral reference, placed so that it crosses the lane line between knots 40 and 41 (and back between 140
    and 141, ...): the lane switch then falls on a segment boundary and every segment lasts one second."""
    kk = np.arange(N, dtype=np.float64)
    l = np.full(N, 1.2)
    for p in range((N + 99) // 100):
        seg = np.clip(1.2 + 0.0825 * (kk - (100 * p + 18.5)), 1.2, 4.5) if p % 2 == 0 else \
            np.clip(4.5 - 0.0825 * (kk - (100 * p + 22.0)), 1.2, 4.5)
        m = kk >= 100 * p
        l[m] = seg[m]
    return l


def make_scenario1_batch(B, S=20, variant=0, seed=None):
    """BASELINE config 3 (variant 0) / config 4 (variant 1): B candidates x S one-second segments, scenario_1-shaped.
    Not feasible by construction: a late, slow obstacle in front of a fast ego leaves no corridor, as on the road."""
    rng = np.random.default_rng(SEED_BASE + 3 + variant if seed is None else seed)
    cub = variant == 1
    # cuboid position rows are clamped to [0, 100] m (cuboid_3d.cc:677-689): the scene is scaled to stay inside
    scale = (95.0 / (C1_ROUTE_SPEED * S)) if (cub and C1_ROUTE_SPEED * S > 95.0) else 1.0
    v_route = C1_ROUTE_SPEED * scale
    sh = shared_params(variant)
    sh.ds_ref, sh.dl_ref = C1_HEADER["ds_ref"], C1_HEADER["dl_ref"]
    sh.dds, sh.ddds, sh.ddl, sh.dddl = C1_HEADER["dds"], C1_HEADER["ddds"], C1_HEADER["ddl"], C1_HEADER["dddl"]
    N = 10 * S + 1
    free_hi = (50.0 + C1_ROUTE_SPEED * max(S - 7, 0)) * scale
    npat, ev = _scenario1_events(rng, B, S, scale)
    kq = np.arange(S)[None, :]                                   # segment = second
    pat = kq // 10
    lane_b = ((kq % 20) >= 4) & ((kq % 20) < 14)                 # second lane from t = 4 s to t = 14 s, and so on
    take = lambda a: np.take_along_axis(a, np.broadcast_to(pat, (B, S)), axis=1)
    ds0 = v_route * 10.0 * pat
    on_hi, sp_hi, on_lo, sp_lo = take(ev["on_hi"]) + 10 * pat, take(ev["sp_hi"]), take(ev["on_lo"]) + 10 * pat, take(ev["sp_lo"])
    in_hi = (kq >= on_hi) & (kq < on_hi + 2)
    in_lo = (kq >= on_lo) & (kq < on_lo + 2) & lane_b
    seg = np.zeros((NUM_SEG_FIELDS, B, S))
    seg[F_T] = 1.0
    seg[F_UPP_BIAS] = np.where(in_hi, ds0 + 19.0 * scale + sp_hi * (kq - on_hi), free_hi)
    seg[F_UPP_SKEW] = np.where(in_hi, sp_hi, 0.0)
    seg[F_DOWN_BIAS] = np.where(in_lo, ds0 + 10.0 * scale + sp_lo * (kq - on_lo), 0.0)
    seg[F_DOWN_SKEW] = np.where(in_lo, sp_lo, 0.0)
    lo_l, hi_l = np.where(lane_b, 3.0, 1.0), np.where(lane_b, 4.5, 3.0)
    seg[F_L_DOWN_BIAS] = lo_l; seg[F_L_UPP_BIAS] = hi_l
    seg[F_BEG_L] = lo_l; seg[F_END_L] = hi_l
    if cub:   # CorridorGeneration of the cuboid variant keeps no l lines (cuboid_3d.cc:301-407): bias 0 / 1000
        seg[F_L_DOWN_BIAS] = 0.0; seg[F_L_UPP_BIAS] = 1000.0
    seg[F_DS_LO] = 0.0; seg[F_DS_HI] = 20.0
    s_ref = v_route * 0.1 * np.arange(N)
    l_ref = _scenario1_lref(N)
    k0 = 10 * np.arange(S)
    seg[F_X_SKEW] = (s_ref[k0 + 1] - s_ref[k0]) / sh.delta; seg[F_X_BIAS] = s_ref[k0]
    seg[F_Y_SKEW] = (l_ref[k0 + 1] - l_ref[k0]) / sh.delta; seg[F_Y_BIAS] = l_ref[k0]
    init = np.zeros((B, 6)); init[:, 1] = ev["v0"]; init[:, 3] = 1.2
    ref_end = np.tile(np.array([s_ref[-1], l_ref[-1]]), (B, 1))
    dlb = np.tile(np.array([-3.0, 3.0] * 5), (B, 1))
    return Batch(B=B, S=S, seg=seg, init=init, ref_end=ref_end, dl_bounds=dlb), sh


def scenario1_knots(B, S=20, seed=None):
    """The same scene at knot level (the content of a corridor text file, trp_wrapper.cpp:39-144) with the SAME random
    draws as make_scenario1_batch: what the device corridor stage (btrapz_corridor_batch_device) takes.  As in c1.txt
    an obstacle ramp covers its last knot too, so the pipeline cuts 0.1-s slivers and the result is ragged (18-24
    segments at S = 20) -- the batch above is the same corridor with every event ending on a segment boundary."""
    from .knots import KnotBatch
    rng = np.random.default_rng(SEED_BASE + 3 if seed is None else seed)
    N = 10 * S + 1
    npat, ev = _scenario1_events(rng, B, S, 1.0)
    k = np.arange(N)[None, :]
    free_hi = 50.0 + C1_ROUTE_SPEED * max(S - 7, 0)
    s_lo = np.zeros((B, 2, N)); s_hi = np.full((B, 2, N), free_hi)
    for p in range(npat):
        ds0 = C1_ROUTE_SPEED * 10.0 * p
        on = 10 * (ev["on_hi"][:, p:p + 1] + 10 * p)
        w = (k >= on) & (k <= on + 20)
        ramp = ds0 + 19.0 + ev["sp_hi"][:, p:p + 1] * 0.1 * (k - on)
        for o in (0, 1):
            s_hi[:, o] = np.where(w, ramp, s_hi[:, o])
        on = 10 * (ev["on_lo"][:, p:p + 1] + 10 * p)
        w = (k >= on) & (k <= on + 20)
        s_lo[:, 1] = np.where(w, ds0 + 10.0 + ev["sp_lo"][:, p:p + 1] * 0.1 * (k - on), s_lo[:, 1])
    l_lo = np.zeros((B, 2, N)); l_hi = np.zeros((B, 2, N))
    l_lo[:, 0] = 1.0; l_hi[:, 0] = 3.0; l_lo[:, 1] = 3.0; l_hi[:, 1] = 4.5
    init = np.zeros((B, 6)); init[:, 1] = ev["v0"]; init[:, 3] = 1.2
    s_ref = np.broadcast_to(C1_ROUTE_SPEED * 0.1 * np.arange(N), (B, N)).copy()
    l_ref = np.broadcast_to(_scenario1_lref(N), (B, N)).copy()
    return KnotBatch(B, N, 2, 0.1, np.stack([s_lo, s_hi], -1), np.stack([l_lo, l_hi], -1),
                     np.broadcast_to(np.array([0.0, 20.0]), (B, N, 2)).copy(),
                     np.broadcast_to(np.array([-3.0, 3.0]), (B, N, 2)).copy(), s_ref, l_ref, init, dict(C1_HEADER))


def make_batch(B, S, config=2, variant=0, seed=None, dtype=np.float64, agents=None, lateral_per_agent=False):
    """Returns (Batch, Shared).  All t_k = 1.0 s (10 knots of 0.1 s), N = 10*S+1.

    agents=G (config 5): the batch is G ego agents x B/G candidate corridors each.  Candidates of one agent
    are alternatives for the SAME ego: they share the initial state and the longitudinal reference and differ
    in corridor margins, obstacle ramps and lane-change target/timing (lateral_per_agent=True: the lane-change
    plan is the agent's too, so that an ego that follows one candidate stays inside the others' corridors --
    the receding-horizon tool tools/mpc_bench.py needs that)."""
    rng = np.random.default_rng(SEED_BASE + config if seed is None else seed)
    if agents:
        assert B % agents == 0
        per = B // agents
        share = lambda draw: np.repeat(draw(agents), per, axis=0)   # one draw per agent, repeated for its candidates
    else:
        share = lambda draw: draw(B)
    sh = shared_params(variant)
    T = float(S)  # horizon in seconds
    N = 10 * S + 1
    tt = np.arange(N) * sh.delta  # knots
    cub = variant == 1
    # longitudinal reference s*(t): v0 + smooth zero-mean acceleration, |a| <= ~1.2
    v0 = share(lambda n: rng.uniform(2.0, 4.5, n) if cub else rng.uniform(4.0, 10.0, n))
    amp = share(lambda n: rng.uniform(0.0, 0.6, (n, 2))) * (0.4 if cub else 1.0)
    ph = share(lambda n: rng.uniform(0, 2 * np.pi, (n, 2)))
    om = 2 * np.pi * np.array([1.0, 2.0]) / max(T, 10.0)
    # v(t) = v0 + sum amp/om * (cos(ph) - cos(om t + ph))
    v = v0[:, None] + ((amp / om)[:, :, None] * (np.cos(ph)[:, :, None] - np.cos(om[None, :, None] * tt[None, None, :] + ph[:, :, None]))).sum(1)
    v = np.maximum(v, 1.0)
    s_star = np.concatenate([np.zeros((B, 1)), np.cumsum(0.5 * (v[:, 1:] + v[:, :-1]) * sh.delta, axis=1)], axis=1)
    # lateral reference: smoothstep between two lanes
    lanes = np.array([1.2, 2.0, 3.5])
    la = share(lambda n: rng.integers(0, 3, n))
    lb = share(lambda n: rng.integers(0, 3, n)) if (agents and lateral_per_agent) else rng.integers(0, 3, B)
    l0, l1 = lanes[la], lanes[lb]
    dl = np.abs(l1 - l0)
    dur = np.maximum(3.0, np.sqrt(5.8 * dl / 0.45))
    t_hi = max(2, int(T - dur.max()) - 1)
    t_start = (share(lambda n: rng.integers(1, t_hi, n)) if (agents and lateral_per_agent) else rng.integers(1, t_hi, B)).astype(float)
    l_star = l0[:, None] + (l1 - l0)[:, None] * _smoothstep5((tt[None, :] - t_start[:, None]) / dur[:, None])

    seg = np.zeros((NUM_SEG_FIELDS, B, S), dtype=dtype)
    k0 = 10 * np.arange(S); k1 = k0 + 10
    s_beg, s_end = s_star[:, k0], s_star[:, k1]
    chord = (s_end - s_beg)  # per 1.0 s
    m_lo = rng.uniform(6.0, 12.0, (B, S)) if cub else rng.uniform(3.0, 10.0, (B, S))
    m_up = rng.uniform(6.0, 12.0, (B, S)) if cub else rng.uniform(3.0, 10.0, (B, S))
    seg[F_T] = 1.0
    seg[F_DOWN_BIAS] = s_beg - m_lo
    seg[F_DOWN_SKEW] = chord
    seg[F_UPP_BIAS] = s_beg + m_up
    seg[F_UPP_SKEW] = chord
    if not cub:
        ramp = rng.uniform(0, 1, (B, S)) < 0.3  # obstacle-style upper ramp (c1.txt: 19 -> 22 -> 25)
        seg[F_UPP_SKEW] = np.where(ramp, 3.0, seg[F_UPP_SKEW])
        seg[F_UPP_BIAS] = np.where(ramp, np.maximum(s_beg, s_end - 3.0) + m_up, seg[F_UPP_BIAS])
    else:
        # axis-aligned box must have a non-empty inscribed interval: lower(end) < upper(beg)
        seg[F_DOWN_BIAS] = np.maximum(s_beg - m_lo, 0.0)
        seg[F_DOWN_SKEW] = 0.0
        seg[F_UPP_BIAS] = s_end + m_up
        seg[F_UPP_SKEW] = 0.0
    l_beg, l_end = l_star[:, k0], l_star[:, k1]
    hw = rng.uniform(0.8, 1.5, (B, S))
    lo_beg, lo_end = l_beg - hw, l_end - hw
    hi_beg, hi_end = l_beg + hw, l_end + hw
    seg[F_L_DOWN_BIAS] = lo_beg
    seg[F_L_DOWN_SKEW] = (lo_end - lo_beg)
    seg[F_L_UPP_BIAS] = hi_beg
    seg[F_L_UPP_SKEW] = (hi_end - hi_beg)
    seg[F_BEG_L] = np.minimum(lo_beg, lo_end) - 0.2
    seg[F_END_L] = np.maximum(hi_beg, hi_end) + 0.2
    seg[F_DS_LO] = 0.0
    seg[F_DS_HI] = 50.0
    # FormulateProblem (solve_3d.cc:1159-1166): ref sampled at knots 10k, 10k+1
    seg[F_X_SKEW] = (s_star[:, k0 + 1] - s_star[:, k0]) / sh.delta
    seg[F_X_BIAS] = s_star[:, k0]
    seg[F_Y_SKEW] = (l_star[:, k0 + 1] - l_star[:, k0]) / sh.delta
    seg[F_Y_BIAS] = l_star[:, k0]
    init = np.zeros((B, 6), dtype=dtype)
    init[:, 1] = v[:, 0]
    init[:, 3] = l_star[:, 0]
    ref_end = np.stack([s_star[:, -1], l_star[:, -1]], axis=1).astype(dtype)
    dlb = np.tile(np.array([-2.0, 2.0] * 5, dtype=dtype), (B, 1))
    return Batch(B=B, S=S, seg=seg, init=init, ref_end=ref_end, dl_bounds=dlb), sh
