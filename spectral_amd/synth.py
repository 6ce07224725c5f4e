"""Synthetic candidate-corridor batches (the workloads of BASELINE.json configs 2-5).

The reference ships no generator (its corridors come from the CommonRoad harness,
cart_frenet.py:833-1030, 384-453).  This module draws corridors of the same shape
as the bundled scenario_1 inputs (src/c1.txt: lane corridors l in (1,3)/(3,4.5),
s_hi obstacle ramps at 3 m/s, limits of src/c_road_s1_2.txt) directly at the level
of the hot path's input record -- one `Cube` (include/btrapz/cube_type.h:2-24) per
segment -- so that a batch is feasible by construction.  Generator spec: SURVEY 8(d).
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
