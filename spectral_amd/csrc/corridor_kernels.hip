// corridor_kernels.hip -- the corridor stage batched on the device and the
// bucketing that lets the QP kernel take candidates with different segment counts.
//
// corridor_batch_kernel: one wavefront per candidate.  Per-knot bounds of every obstacle ->
// CorridorGeneration + CorridorSplit (one lane per obstacle) -> CollisionCheck: reference knots are
// tested against every segment with the lanes spread over the knots (ballot + popcount gives the hit
// count; the reference's running counter becomes a carry, corridor_core.h) -> de-dup / order / overlap
// resolution (lane 0, the lists live in LDS) -> the batch record of the QP kernel (one lane per
// selected segment, coalesced field-major stores).  This is the one stage of the path that streams
// HBM: num_obs * N * 4 doubles per candidate (11 KB at N = 71, 5 obstacles).
// References: src/solve_3d.cc:323-486,488-714,729-772,835-845,1159-1166 ; src/cuboid_3d.cc:301-573.
#include <hip/hip_runtime.h>

#include "btrapz_device.h"
#include "corridor_core.h"

namespace btrapz {

enum { MAX_ALL = 160, MAX_SEL = 64 };

// Dynamic LDS: [s_ref N][l_ref N][s_bounds O*N*2][l_bounds O*N*2] when `staged` (it fits), else only the refs.
__global__ __launch_bounds__(64) void corridor_batch_kernel(const CorridorArgs a, int staged) {
  __shared__ Seg all[MAX_ALL];
  __shared__ Seg sel[MAX_SEL];
  __shared__ int ocount[64];
  __shared__ int nsel_sh;
  extern __shared__ double dyn[];

  const int b = blockIdx.x, lane = threadIdx.x;
  const int N = a.N, O = a.num_obs;
  const int cap_o = MAX_ALL / (O > 0 ? O : 1);
  double *sref = dyn, *lref = dyn + N;
  for (int i = lane; i < N; i += 64) { sref[i] = a.s_ref[(size_t)b * N + i]; lref[i] = a.l_ref[(size_t)b * N + i]; }
  // The per-obstacle scan below is a serial walk over the knots: from HBM that is N dependent round trips
  // per lane.  Stage the candidate's bounds into LDS with coalesced 16-byte loads first.
  const double *gs = a.s_bounds + (size_t)b * O * N * 2, *gl = a.l_bounds + (size_t)b * O * N * 2;
  const double *ss = gs, *sl_ = gl;
  if (staged) {
    double *ls = dyn + 2 * N, *ll = ls + (size_t)O * N * 2;
    const int n2 = O * N;  // pairs
    const double2 *gs2 = reinterpret_cast<const double2 *>(gs), *gl2 = reinterpret_cast<const double2 *>(gl);
    double2 *ls2 = reinterpret_cast<double2 *>(ls), *ll2 = reinterpret_cast<double2 *>(ll);
    for (int i = lane; i < n2; i += 64) { ls2[i] = gs2[i]; ll2[i] = gl2[i]; }
    ss = ls; sl_ = ll;
  }
  __syncthreads();
  // ---- per-obstacle extraction: lane o owns obstacle o ----
  if (lane < O) {
    const BoundsView sb{ss + (size_t)lane * N * 2}, lb{sl_ + (size_t)lane * N * 2};
    ocount[lane] = extract_segments_core(a.variant, N, a.delta, sb, lb, all + lane * cap_o, cap_o);
  }
  __syncthreads();
  // ---- selection along the reference: hits per segment, lanes over knots ----
  int carry = 0, nsel = 0;
  bool overflow = false;
  for (int o = 0; o < O; o++) {
    const int n = ocount[o];
    if (n < 0) { overflow = true; break; }
    for (int j = 0; j < n; j++) {
      const Seg c = all[o * cap_o + j];
      int hits = 0;
      for (int i0 = 0; i0 < N; i0 += 64) {
        const int i = i0 + lane;
        const bool in = i < N && knot_inside(c, sref[i], lref[i], (double)i, a.delta);
        hits += __popcll(__ballot(in));
      }
      if (selection_pushes(hits, carry) >= 1) {
        if (nsel < MAX_SEL) { if (lane == 0) { sel[nsel] = c; sel[nsel].count = 3; } nsel++; }
        else overflow = true;
      }
    }
  }
  __syncthreads();
  if (lane == 0) nsel_sh = (overflow || nsel == 0) ? (overflow ? -1 : 0) : order_segments_core(a.variant, a.delta, sel, nsel);
  __syncthreads();
  int S = nsel_sh;
  // ---- batch record: lane k writes segment k ----
  bool bad = S > a.seg_stride;
  if (S > 0 && !bad && lane < S) {
    const Seg c = sel[lane];
    if (!(c.t > 0.0)) bad = true;
    const size_t BS = (size_t)a.B * a.seg_stride, e = (size_t)b * a.seg_stride + lane;
    double *sg = a.seg;
    sg[BTRAPZ_F_T * BS + e] = c.t;
    sg[BTRAPZ_F_DOWN_BIAS * BS + e] = c.down_bias; sg[BTRAPZ_F_DOWN_SKEW * BS + e] = c.down_skew;
    sg[BTRAPZ_F_UPP_BIAS * BS + e] = c.upp_bias; sg[BTRAPZ_F_UPP_SKEW * BS + e] = c.upp_skew;
    sg[BTRAPZ_F_L_DOWN_BIAS * BS + e] = c.l_down_bias; sg[BTRAPZ_F_L_DOWN_SKEW * BS + e] = c.l_down_skew;
    sg[BTRAPZ_F_L_UPP_BIAS * BS + e] = c.l_upp_bias; sg[BTRAPZ_F_L_UPP_SKEW * BS + e] = c.l_upp_skew;
    sg[BTRAPZ_F_BEG_L * BS + e] = c.beg_l; sg[BTRAPZ_F_END_L * BS + e] = c.end_l;
    double lo = 0.0, hi = 1000.0;  // solve_3d.cc:835-841
    for (int i = c.beg_t; i <= c.end_t; i++) {
      const int ii = i < 0 ? 0 : (i > N - 1 ? N - 1 : i);
      lo = fmax(a.ds_bounds[((size_t)b * N + ii) * 2], lo);
      hi = fmin(a.ds_bounds[((size_t)b * N + ii) * 2 + 1], hi);
    }
    sg[BTRAPZ_F_DS_LO * BS + e] = lo; sg[BTRAPZ_F_DS_HI * BS + e] = hi;
    const int i0 = 10 * lane > N - 1 ? N - 1 : 10 * lane, i1 = 10 * lane + 1 > N - 1 ? N - 1 : 10 * lane + 1;  // :1161-1165, clamped
    sg[BTRAPZ_F_X_SKEW * BS + e] = (sref[i1] - sref[i0]) / a.delta; sg[BTRAPZ_F_X_BIAS * BS + e] = sref[i0];
    sg[BTRAPZ_F_Y_SKEW * BS + e] = (lref[i1] - lref[i0]) / a.delta; sg[BTRAPZ_F_Y_BIAS * BS + e] = lref[i0];
  }
  if (__any(bad)) S = -1;
  if (lane == 0) {
    a.seg_count[b] = S;
    a.ref_end[(size_t)b * 2] = sref[N - 1]; a.ref_end[(size_t)b * 2 + 1] = lref[N - 1];
  }
  if (lane < 10) {
    const int i = lane >> 1, ii = i > N - 1 ? N - 1 : i;
    a.dl10[(size_t)b * 10 + lane] = a.dl_bounds[((size_t)b * N + ii) * 2 + (lane & 1)];
  }
}

// ---- bucketing by segment count ------------------------------------------------------------------
// meta[0..65]: histogram, then cand_prefix ; meta[66..131]: wave_prefix ; meta[132..197]: cursors.
__global__ void bucket_hist_kernel(int B, int seg_stride, const int *seg_count, int *meta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const int s = seg_count[i];
  if (s >= 1 && s <= 64 && s <= seg_stride) atomicAdd(&meta[132 + s], 1);
}
__global__ void bucket_prefix_kernel(int *meta) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int cand = 0, wave = 0;
  meta[0] = 0; meta[66] = 0;
  for (int s = 1; s <= 64; s++) {
    const int cnt = meta[132 + s], gpw = 64 / s;
    meta[s] = cand; meta[66 + s] = wave;
    cand += cnt; wave += (cnt + gpw - 1) / gpw;
    meta[132 + s] = 0;  // becomes the scatter cursor
  }
  meta[65] = cand; meta[66 + 65] = wave;
}
// Candidate order inside a bucket comes from atomics: it only decides which candidates share a
// wavefront, and every group of a wavefront is solved independently, so results do not depend on it.
__global__ void bucket_scatter_kernel(int B, int seg_stride, const int *seg_count, int *meta, int *order,
                                      double *axis_obj, int *axis_status, int *axis_iters) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const int s = seg_count[i];
  if (s >= 1 && s <= 64 && s <= seg_stride) {
    const int pos = atomicAdd(&meta[132 + s], 1);
    order[meta[s] + pos] = i;
  } else {  // no usable corridor: the reference's find_traj fails here (empty selection / CHECK)
    axis_obj[2 * i] = 0.0; axis_obj[2 * i + 1] = 0.0;
    axis_status[2 * i] = BTRAPZ_NO_CORRIDOR; axis_status[2 * i + 1] = BTRAPZ_NO_CORRIDOR;
    axis_iters[2 * i] = 0; axis_iters[2 * i + 1] = 0;
  }
}

}  // namespace btrapz
