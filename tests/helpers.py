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
