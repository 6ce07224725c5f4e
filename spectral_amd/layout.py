"""HBM data layout of a candidate batch (mirrors include/btrapz_hip.h).

One candidate = S segment records (the reference's `Cube`, cube_type.h:2-24, plus the
per-segment quantities its assembly derives: ds bounds from solve_3d.cc:835-845 and the
piecewise-linear reference of solve_3d.cc:1159-1166).  Stored field-major (SoA):

    seg[f][b][k]   f = field, b = candidate, k = segment        float64

so that the lanes of a wavefront (lane = segment, consecutive candidates packed into
one wave) read consecutive addresses for every field.
"""
from dataclasses import dataclass, field

import numpy as np

(F_T, F_DOWN_BIAS, F_DOWN_SKEW, F_UPP_BIAS, F_UPP_SKEW, F_L_DOWN_BIAS, F_L_DOWN_SKEW,
 F_L_UPP_BIAS, F_L_UPP_SKEW, F_BEG_L, F_END_L, F_DS_LO, F_DS_HI, F_X_SKEW, F_X_BIAS,
 F_Y_SKEW, F_Y_BIAS) = range(17)
NUM_SEG_FIELDS = 17
MAX_SEGMENTS = 64

# OSQP status_val vocabulary (the reference accepts 1 and 2: solve_3d.cc:1253)
STATUS_SOLVED = 1
STATUS_SOLVED_INACCURATE = 2
STATUS_MAX_ITER = -2
STATUS_PRIMAL_INFEASIBLE = -3
FAIL_SENTINEL = 100000000000.0  # trp_wrapper.cpp:199


@dataclass
class Shared:
    """Weights and limits shared by every candidate of a batch (Params + file header)."""
    w_s: tuple  # weight_s_ref, weight_ds_ref, s_acc_weight, s_jerk_weight
    w_l: tuple
    weight_end_s: float
    weight_end_l: float
    ds_ref: float
    dl_ref: float
    dds: tuple
    ddds: tuple
    ddl: tuple
    dddl: tuple
    delta: float = 0.1
    variant: int = 0  # 0 trapezoid (solve_3d.cc), 1 cuboid (cuboid_3d.cc)

    def as_array(self):
        return np.array([*self.w_s, *self.w_l, self.weight_end_s, self.weight_end_l, self.ds_ref,
                         self.dl_ref, *self.dds, *self.ddds, *self.ddl, *self.dddl, self.delta],
                        dtype=np.float64)


@dataclass
class Batch:
    B: int
    S: int
    seg: np.ndarray       # [NUM_SEG_FIELDS][B][S]
    init: np.ndarray      # [B][6]  s0 ds0 dds0 l0 dl0 ddl0
    ref_end: np.ndarray   # [B][2]  x_ref[N-1], y_ref[N-1]   (solve_3d.cc:268,315)
    dl_bounds: np.ndarray  # [B][10] dy_bounds_[i], i=0..4 as lo,hi (solve_3d.cc:1003-1004)

    def slice(self, lo, hi):
        return Batch(B=hi - lo, S=self.S, seg=np.ascontiguousarray(self.seg[:, lo:hi]),
                     init=np.ascontiguousarray(self.init[lo:hi]),
                     ref_end=np.ascontiguousarray(self.ref_end[lo:hi]),
                     dl_bounds=np.ascontiguousarray(self.dl_bounds[lo:hi]))

    def algorithmic_bytes(self):
        """Bytes one solve must move: inputs + 12S control points + cost + status."""
        S = self.S
        return (NUM_SEG_FIELDS * S + 6 + 2 + 10) * 8 + 12 * S * 8 + 8 + 4
