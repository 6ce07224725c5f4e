"""Test-side glue between the product's batch layout and the CPU oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from spectral_amd import layout as L  # noqa: E402


class _Src:
    pass


def oracle_qp_from_batch(batch, sh, b):
    """Candidate b of a Batch -> oracle AssembledQp (general CSC P,q,A,l,u).

    Rebuilds the per-knot arrays the reference's assembly reads (N = 10*S+1 knots) so
    that orc_assemble() -- the restatement of solve_3d.cc:70-321,779-1129 -- sees the
    same numbers the batch record carries."""
    S = batch.S
    N = 10 * S + 1
    g = lambda f: batch.seg[f, b]
    cubes = []
    for k in range(S):
        c = O.Cube()
        c.beg_t, c.end_t, c.t = 10 * k, 10 * k + 10, float(g(L.F_T)[k])
        c.beg_l, c.end_l = float(g(L.F_BEG_L)[k]), float(g(L.F_END_L)[k])
        c.upp_skew, c.upp_bias = float(g(L.F_UPP_SKEW)[k]), float(g(L.F_UPP_BIAS)[k])
        c.down_skew, c.down_bias = float(g(L.F_DOWN_SKEW)[k]), float(g(L.F_DOWN_BIAS)[k])
        c.l_upp_skew, c.l_upp_bias = float(g(L.F_L_UPP_SKEW)[k]), float(g(L.F_L_UPP_BIAS)[k])
        c.l_down_skew, c.l_down_bias = float(g(L.F_L_DOWN_SKEW)[k]), float(g(L.F_L_DOWN_BIAS)[k])
        cubes.append(c)
    src = _Src()
    src.N, src.delta = N, sh.delta
    x_ref = np.zeros(N); y_ref = np.zeros(N)
    for k in range(S):
        x_ref[10 * k] = g(L.F_X_BIAS)[k]; x_ref[10 * k + 1] = g(L.F_X_BIAS)[k] + g(L.F_X_SKEW)[k] * sh.delta
        y_ref[10 * k] = g(L.F_Y_BIAS)[k]; y_ref[10 * k + 1] = g(L.F_Y_BIAS)[k] + g(L.F_Y_SKEW)[k] * sh.delta
    x_ref[N - 1], y_ref[N - 1] = batch.ref_end[b]
    src.x_ref, src.y_ref = x_ref, y_ref
    dxb = np.zeros((N, 2)); dxb[:, 0] = -1e10; dxb[:, 1] = 1e10
    for k in range(S):  # interior knots carry the segment's bounds; shared boundary knots stay loose
        dxb[10 * k + 1:10 * k + 10, 0] = g(L.F_DS_LO)[k]; dxb[10 * k + 1:10 * k + 10, 1] = g(L.F_DS_HI)[k]
    src.dx_bounds = dxb
    dyb = np.zeros((N, 2)); dyb[:, 0] = -1e10; dyb[:, 1] = 1e10
    dyb[:5] = batch.dl_bounds[b].reshape(5, 2)
    src.dy_bounds = dyb
    src.ds_ref, src.dl_ref = sh.ds_ref, sh.dl_ref
    src.dds, src.ddds, src.ddl, src.dddl = sh.dds, sh.ddds, sh.ddl, sh.dddl
    src.init_s, src.init_l = batch.init[b, :3], batch.init[b, 3:]
    p = O.Params(sh.w_s[2], sh.w_s[3], sh.w_l[2], sh.w_l[3], sh.w_s[0], sh.w_s[1], sh.w_l[0], sh.w_l[1],
                 sh.weight_end_s, sh.weight_end_l, 0)
    return O.AssembledQp(sh.variant, cubes, p, src)


def fuzz_knot_batch(seed, B=16, N=None, num_obs=None):
    """A random corridor-stage input: horizon and obstacle count from the edges of the device stage's range, s bounds
    with slope changes in runs of random length (some below, some above the 0.2 threshold), now and then an upper bound
    with breaks of its own, moving l bounds, collapsed bounds, nan / inf entries, a nan reference knot.  The shapes that
    found three disagreements between the device stage and the reference's arithmetic (twins with a NaN field survive
    the reference's de-dup; fused multiply-adds flip the sign of a degenerate edge function; an infinite slope voids
    the device's 'own knots only' shortcut)."""
    from spectral_amd import synth
    from spectral_amd.knots import KnotBatch
    rng = np.random.default_rng(seed)
    N_ = int(rng.choice([3, 4, 5, 11, 21, 64, 65, 66, 71, 101, 128, 129, 130, 201, 257, 300, 512]))
    O_ = int(rng.choice([1, 2, 3, 5, 8, 13, 64])) if N_ <= 130 else int(rng.choice([1, 2, 3, 5]))
    N = N_ if N is None else int(N)                        # (overrides: shapes beyond the wave-wide kernels, round 6)
    num_obs = O_ if num_obs is None else int(num_obs)
    tt = np.arange(N) * 0.1
    sb = np.zeros((B, num_obs, N, 2)); lb = np.zeros((B, num_obs, N, 2))
    for b in range(B):
        for o in range(num_obs):
            lo_run, hi_run = rng.choice([(1, 3), (3, 12), (10, 40), (40, 200)])
            steps = np.repeat(rng.choice([0.0, 0.0, 0.01, 0.019, 0.021, 0.3, -0.2, 0.6], size=N), rng.integers(lo_run, hi_run + 1, size=N))[:N]
            lo = np.round(rng.uniform(0, 10) + np.cumsum(steps), 3)
            width = np.round(rng.uniform(1, 60), 2)
            if rng.random() < 0.3:
                steps2 = np.repeat(rng.choice([0.0, 0.05, 0.3, -0.1], size=N), rng.integers(lo_run, hi_run + 1, size=N))[:N]
                hi = lo + width + np.round(np.cumsum(steps2), 3)
            else:
                hi = lo + width
            sb[b, o, :, 0] = lo; sb[b, o, :, 1] = hi
            l0 = np.round(rng.uniform(-4, 2), 1)
            lb[b, o, :, 0] = l0; lb[b, o, :, 1] = l0 + np.round(rng.uniform(0.5, 4), 1)
            if rng.random() < 0.2:
                lb[b, o, :, 0] += np.round(0.05 * np.arange(N) * rng.choice([0, 1, -1]), 2)
            if rng.random() < 0.05:
                i0 = int(rng.integers(0, N)); sb[b, o, i0:i0 + 5, 1] = sb[b, o, i0:i0 + 5, 0]
            if rng.random() < 0.03:
                sb[b, o, int(rng.integers(0, N)), int(rng.integers(0, 2))] = rng.choice([np.nan, np.inf, -np.inf])
    s_ref = np.tile(rng.uniform(2, 12) + rng.uniform(0, 6) * tt, (B, 1)) + rng.uniform(-2, 2, (B, 1))
    l_ref = rng.uniform(-3, 3, (B, 1)) + np.where(tt < tt[N // 2], 0.0, rng.uniform(-2, 2))[None, :]
    if rng.random() < 0.1:
        s_ref[rng.integers(0, B), rng.integers(0, N)] = np.nan
    return KnotBatch(B, N, num_obs, 0.1, sb, lb, np.tile(np.array([0.0, 20.0]), (B, N, 1)) + rng.uniform(0, 1, (B, N, 2)),
                     np.tile(np.array([-3.0, 3.0]), (B, N, 1)), s_ref, l_ref, np.tile(np.array([0.0, 6.0, 0.0, 0.0, 0.0, 0.0]), (B, 1)),
                     dict(synth.C1_HEADER))


def oracle_corridor(kb, b, variant, per_obstacle_cap=None):
    """The oracle's corridor stage on candidate b: (n, cubes), or (None, None) when an obstacle's list exceeds
    per_obstacle_cap (the device reports such a candidate as unusable)."""
    try:
        lists = [O.corridor_generation(variant, kb.N, kb.delta, kb.s_bounds[b, o], kb.l_bounds[b, o]) for o in range(kb.num_obs)]
    except RuntimeError:
        return None, None
    if per_obstacle_cap is not None and max(len(l) for l in lists) > per_obstacle_cap:
        return None, None
    return O.collision_check(variant, kb.N, kb.delta, lists, kb.s_ref[b], kb.l_ref[b])
