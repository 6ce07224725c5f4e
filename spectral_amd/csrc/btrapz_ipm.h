// btrapz_ipm.h -- device helpers shared by the solve kernels (btrapz_kernels.hip: the packed / split / long forms and
// their warm-start, rescue and capped instantiations; btrapz_lean.hip: the two-wavefronts-per-SIMD form): reciprocals,
// DPP lane shifts, the constraint rows of a segment, the null-space maps, 3x3 LDL^T, group reductions.
#ifndef BTRAPZ_IPM_H
#define BTRAPZ_IPM_H
#include <hip/hip_runtime.h>

#include <type_traits>

#include "btrapz_device.h"

namespace btrapz {

#define UNROLL _Pragma("unroll")
#define SYM(i, j) ((j) * ((j) + 1) / 2 + (i))  // i <= j, packed upper triangle, column-wise

// 1/x: v_rcp_f64 seed (measured 4.6e-8 relative on gfx950) + one Newton step -> 2.2e-15.  No
// denormal / overflow fix-ups: every operand here is a positive slack, multiplier or pivot.
__device__ __forceinline__ double rcp(double x) {
  const double r = __builtin_amdgcn_rcp(x);
  return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
}
// seed only: good to 5e-8, enough for step-length ratios (they carry a 0.5 % safety factor).
__device__ __forceinline__ double rcp_fast(double x) { return __builtin_amdgcn_rcp(x); }

// lane i <- lane i-1 / lane i+1 over the whole wavefront (DPP wave_shr:1 / wave_shl:1, bound_ctrl: the lane
// without a source reads 0, so the destination needs no initialisation).
__device__ __forceinline__ double dpp_prev(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_next(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// ---- constraint rows of one segment (solve_3d.cc:823-888): 6 pos, 5 vel, 4 acc, 3 jerk ----
// row r: first column, number of columns, coefficients (position rows carry the runtime t).
__host__ __device__ constexpr int row_col0(int r) { return r < 6 ? r : r < 11 ? r - 6 : r < 15 ? r - 11 : r - 15; }
__host__ __device__ constexpr int row_nnz(int r) { return r < 6 ? 1 : r < 11 ? 2 : r < 15 ? 3 : 4; }
__host__ __device__ constexpr double row_coef(int r, int j) {
  return r < 6 ? 1.0 : r < 11 ? (j == 0 ? -5.0 : 5.0) : r < 15 ? (j == 1 ? -40.0 : 20.0)
                                                              : (j == 0 ? -60.0 : j == 1 ? 180.0 : j == 2 ? -180.0 : 60.0);
}
template <int R> __device__ __forceinline__ double row_dot(const double (&c)[6], double t) {
  if constexpr (R < 6) return t * c[R];
  else if constexpr (R < 11) return 5.0 * (c[R - 5] - c[R - 6]);
  else if constexpr (R < 15) return 20.0 * ((c[R - 11] - 2.0 * c[R - 10]) + c[R - 9]);
  else return 60.0 * ((c[R - 12] - c[R - 15]) + 3.0 * (c[R - 14] - c[R - 13]));
}
template <int R> __device__ __forceinline__ void row_scatter(double v, double t, double (&o)[6]) {  // o += G_r' v
  if constexpr (R < 6) o[R] += t * v;
  else {
    UNROLL for (int j = 0; j < row_nnz(R); j++) o[row_col0(R) + j] += row_coef(R, j) * v;
  }
}
template <int R> __device__ __forceinline__ void row_outer(double w, double t2, double (&H)[21]) {  // H += w G_r' G_r
  if constexpr (R < 6) H[SYM(R, R)] += t2 * w;
  else {
    UNROLL for (int a = 0; a < row_nnz(R); a++)
      UNROLL for (int b = a; b < row_nnz(R); b++)
        H[SYM(row_col0(R) + a, row_col0(R) + b)] += (row_coef(R, a) * row_coef(R, b)) * w;
  }
}
// Stops the optimiser from carrying row-sized temporaries (G c, residuals, LDS reloads) from one
// row loop to the next: recomputing them is cheap, keeping 9 x 18 doubles alive spills to scratch.
__device__ __forceinline__ void opaque6(double (&v)[6]) {
  UNROLL for (int i = 0; i < 6; i++) asm volatile("" : "+v"(v[i]));
}
// First statement of a block guarded by a wave-uniform condition: keeps the block a real (scalar) branch -- the
// optimiser would otherwise turn the rare path into selects executed on every pass.
#define UNIFORM_BLOCK asm volatile("")
#define PHASE_FENCE(...) do { asm volatile("" ::: "memory"); __VA_ARGS__; } while (0)
template <int N, class F> __device__ __forceinline__ void static_for(F &&f) {
  if constexpr (N > 0) { static_for<N - 1>(f); f(std::integral_constant<int, N - 1>{}); }
}
// Rows kept by a solve.  FULL: all 18 rows of a segment, as the reference assembles them.  Otherwise 15: the first
// position, velocity and acceleration row of a segment (rows 0, 6, 11) state, about the joint at its start, what the
// previous segment's last rows (5, 10, 14) state about the same joint from the other side -- t c_{k,5} = t' c_{k+1,0},
// equal end / start velocities and accelerations are the continuity equalities (solve_3d.cc:918-949).  Two bounds on
// one quantity are one bound: the previous lane keeps the row with the intersection, this lane drops it -- same
// feasible set, same optimum, a sixth less row work and state.  (Segment 0's start rows constrain the given initial
// state, a constant: checked once.)  The rescue pass counts violations row by row and keeps all 18.
template <bool FULL> __host__ __device__ constexpr int rows_kept() { return FULL ? 18 : 15; }
template <bool FULL> __host__ __device__ constexpr int row_id(int i) {   // i-th kept row -> row of the segment
  return FULL ? i : (i < 5 ? i + 1 : i < 9 ? i + 2 : i < 12 ? i + 3 : i + 3);
}
template <bool FULL> __host__ __device__ constexpr int state_index(int r) {   // row of the segment -> slot in the state arrays
  return FULL ? r : (r < 6 ? r - 1 : r < 11 ? r - 2 : r < 15 ? r - 3 : r - 3);
}
template <bool FULL> __host__ __device__ constexpr int next_row(int r) {      // the kept row after r, -1 after the last
  return r >= 17 ? -1 : (FULL ? r + 1 : (r + 1 == 6 || r + 1 == 11) ? r + 2 : r + 1);
}
// Split form (SPLIT in ipm_solve_body): which lane group j = 0..2 owns row r of a segment, and in which of its five
// slots.  Slot 0: jerk row 15 + j; 1: acceleration row 12 + j; 2: velocity row 7 + j; 3: position row 1 / 3 / 5;
// 4: position row 2 / 4 for j = 0 / 1, velocity row 10 for j = 2 (rows 0, 6, 11 are not kept: rows_kept).
__host__ __device__ constexpr int split_owner_j(int r) {
  return r >= 15 ? r - 15 : r >= 12 ? r - 12 : (r >= 7 && r <= 9) ? r - 7 : r == 10 ? 2 : (r & 1) ? (r - 1) / 2 : (r - 2) / 2;
}
__host__ __device__ constexpr int split_owner_slot(int r) {
  return r >= 15 ? 0 : r >= 12 ? 1 : (r >= 7 && r <= 9) ? 2 : r == 10 ? 4 : (r & 1) ? 3 : 4;
}
#define FOR_ROWS(r) static_for<NR>([&](auto r##_c) { constexpr int r = row_id<FULL>(decltype(r##_c)::value);
#define END_ROWS });
#define SI(r) state_index<FULL>(r)

// Null-space maps.  X = (p, v, a) physical state at a joint; segment duration t.
//   start of segment:  c0 = p/t, c1 = c0 + v/5, c2 = c0 + 2v/5 + a t/20
//   end of segment:    c5 = p/t, c4 = c5 - v/5, c3 = c5 - 2v/5 + a t/20
// (the inverse of the reference's equality rows solve_3d.cc:896-949).
struct NullMap { double it, t20; };
__device__ __forceinline__ void U_apply(const NullMap m, const double (&X)[3], double &c0, double &c1, double &c2) {
  c0 = m.it * X[0]; c1 = c0 + 0.2 * X[1]; c2 = c0 + 0.4 * X[1] + m.t20 * X[2];
}
__device__ __forceinline__ void V_apply(const NullMap m, const double (&X)[3], double &c3, double &c4, double &c5) {
  c5 = m.it * X[0]; c4 = c5 - 0.2 * X[1]; c3 = c5 - 0.4 * X[1] + m.t20 * X[2];
}
__device__ __forceinline__ void UT_apply(const NullMap m, double h0, double h1, double h2, double (&o)[3]) {
  o[0] = m.it * ((h0 + h1) + h2); o[1] = 0.2 * h1 + 0.4 * h2; o[2] = m.t20 * h2;
}
__device__ __forceinline__ void VT_apply(const NullMap m, double h3, double h4, double h5, double (&o)[3]) {
  o[0] = m.it * ((h3 + h4) + h5); o[1] = -0.4 * h3 - 0.2 * h4; o[2] = m.t20 * h3;
}

// A solve whose step length stays below BTRAPZ_TINY_STEP for BTRAPZ_TINY_STEPS iterations in a row is going nowhere: the
// iterate of a problem without a solution wedges itself against bounds that contradict each other and creeps on with
// steps of 1e-5 and less, the score falling in its tenth digit -- which the "no better score for six iterations" rule
// counts as progress.  (With the unweighted second-order term such iterates blew up within a few iterations and the
// divergence rule caught them at 17 on average; weighted, they crept on to 30.  Solvable candidates: the smallest steps
// seen on the slowest of them are 5e-3, never twice in a row.)
#define BTRAPZ_TINY_STEP 1e-3
#define BTRAPZ_TINY_STEPS 3

// Cold start (all forms): slacks max(gap, BTRAPZ_COLD_SLACK), multipliers BTRAPZ_COLD_LAMBDA at the initial state propagated
// at constant velocity.  Rounds 1-3 had 1 and 1; with the weighted second-order term (below) a start further inside
// pays: tools/ab_variants.py sweep, round 4, 65 536 x 20 two launches / jittered bundled files (sum of six sets' mean
// iterations): (1, 1) 4.30 ms / 81.9; (1, 2) 4.18 / 75.1; (1, 3) 4.15 / 71.9; (2, 4) 4.08 / 72.2; (2, 5) 4.05 / 72.1;
// (3, 4) 4.05 / 75.6; (3, 6) 4.01 / 73.0 -- a broad plateau; generic batch 3.69 -> 3.67, cuboid 4.39 -> 4.04, 10 segments
// 1.59 -> 1.66.  Per-row centred starts (lambda = mu0 / s) and a multiplier from the group's mean slack: no gain.
// (The rescue pass -- the relaxed problem, its own scaling -- keeps 1 and 1.)
#ifndef BTRAPZ_COLD_SLACK
#define BTRAPZ_COLD_SLACK 3.0
#endif
#ifndef BTRAPZ_COLD_LAMBDA
#define BTRAPZ_COLD_LAMBDA 6.0
#endif

// ---- bounds that are no bounds (round 5) ------------------------------------------------------------------------------
// The reference's rows default to +-1e10 (src/piecewise_jerk_problem.cc:9,25-35) and a caller may leave them there: such
// a row can never be active.  Taken literally it wrecks the solve twice over: a cold start with slack 1e10 and multiplier
// 6 puts mu at 1e10 and the first centring step flings every other row after it (found by the round-5 sweep: a quarter
// of the candidates of a batch with default dl bounds ended "stalled" after 4-9 iterations); and 1 / (1 + |bounds|)
// scales the primal residual of every other row by 1e-10.  So:
//   * a FINITE bound of magnitude >= BTRAPZ_FAR is moved to +-BTRAPZ_FAR_FACTOR (1 + the largest other bound of the
//     lane's rows) when the record is read: 2^20 times further out than anything the problem's data says, still "never
//     active", but near enough that (value + slack - bound) rounds at 1e-10 of the data's scale instead of at 1e-6;
//     header limits (acceleration, jerk) beyond BTRAPZ_FAR are moved to +-BTRAPZ_FAR_LIMIT by the host;
//   * such bounds do not enter |bounds|;
//   * a cold start gives a row whose slack is above BTRAPZ_COLD_FAR the centred multiplier COLD_SLACK COLD_LAMBDA / s
//     instead of COLD_LAMBDA (rows of ordinary size keep the start the bench batches were tuned on, bit for bit).
// An INFINITE bound stays what it was: a defect of the input that ends in "no trajectory" (DESIGN section 6).
// (BTRAPZ_FAR, _FAR_FACTOR, _FAR_LIMIT, _COLD_FAR: btrapz_device.h -- the host moves the header limits)
__device__ __forceinline__ bool far_bound(double v) { const double a_ = fabs(v); return a_ >= BTRAPZ_FAR && a_ <= 1.7e308; }
// largest |v| among bounds that are real ones
__device__ __forceinline__ double near_norm(double acc, double v) { const double a_ = fabs(v); return a_ < BTRAPZ_FAR ? fmax(acc, a_) : acc; }
__device__ __forceinline__ double cold_lambda(double slack, double slack0, double lambda0) {
  return slack > BTRAPZ_COLD_FAR ? (slack0 * lambda0) * rcp_fast(slack) : lambda0;   // (a starting value: the seed will do)
}
// Position lines lo0 + i dlo (i = 0..5) and velocity intervals of a lane, far bounds moved in; returns the lane's `big`.
__device__ __forceinline__ double move_far_bounds(double &plo0, double &dplo, double &phi0, double &dphi, double (&vlo)[5], double (&vhi)[5]) {
  double fin = 0.0;
  fin = near_norm(fin, plo0); fin = near_norm(fin, plo0 + 5.0 * dplo); fin = near_norm(fin, phi0); fin = near_norm(fin, phi0 + 5.0 * dphi);
  UNROLL for (int i = 0; i < 5; i++) { fin = near_norm(fin, vlo[i]); fin = near_norm(fin, vhi[i]); }
  const double big = BTRAPZ_FAR_FACTOR * (1.0 + fin);
  const double plo5 = plo0 + 5.0 * dplo, phi5 = phi0 + 5.0 * dphi;
  if (far_bound(plo0) || far_bound(plo5)) { plo0 = copysign(big, far_bound(plo0) ? plo0 : plo5); dplo = 0.0; }
  if (far_bound(phi0) || far_bound(phi5)) { phi0 = copysign(big, far_bound(phi0) ? phi0 : phi5); dphi = 0.0; }
  UNROLL for (int i = 0; i < 5; i++) {
    if (far_bound(vlo[i])) vlo[i] = copysign(big, vlo[i]);
    if (far_bound(vhi[i])) vhi[i] = copysign(big, vhi[i]);
  }
  return big;
}

// Factor of Mehrotra's second-order term ds_aff * dlambda_aff in the corrector's complementarity target (it enters through
// a fused multiply-add that subtracts it: the factor is minus its weight).  The term describes the affine step; where
// that step is cut short by a bound it describes less, and taking it in full then pushes the corrected step against the
// same bound.  Weight min(1, 2 min(ap, ad)), ap / ad the affine step's lengths to the boundary on the primal / dual
// side: in full from half a step on, proportionally below (in the spirit of the weighted correctors of Colombo and
// Gondzio).  Measured, round 4, 4 x 65 536 bench candidates against the unweighted term: same accept sets, mean
// iterations 9.72 -> 8.83 (scenario_1), 8.21 -> 7.76 (generic), 9.91 -> 9.21 (cuboid), 7.55 -> 6.87 (10 segments);
// min(ap, ad): 9.30 but a longer tail; 1.5 / 1.75 / 2.5 / 3 times: 8.87 / 8.82 / 8.89 / 9.01; squared, ap ad, sqrt: worse.
// plain: the second chance of a stalled solve leaves the term out where the affine step is blocked below a tenth.
__device__ __forceinline__ double second_order_factor(double ap, double ad, bool plain) {
  const double m = fmin(ap, ad);
  return (plain && m < 0.1) ? 0.0 : -fmin(1.0, 2.0 * m);
}

// ---- 3x3 SPD helpers: LDL^T factor F = (l10, l20, l21, 1/d0, 1/d1, 1/d2) ----
__device__ __forceinline__ void ldl3(const double (&A)[6] /*00 01 02 11 12 22*/, double (&F)[6]) {
  const double id0 = rcp(A[0]);
  const double l10 = A[1] * id0, l20 = A[2] * id0;
  const double d1 = A[3] - l10 * A[1];
  const double id1 = rcp(d1);
  const double e = A[4] - l20 * A[1];
  const double l21 = e * id1;
  const double d2 = A[5] - l20 * A[2] - l21 * e;
  F[0] = l10; F[1] = l20; F[2] = l21; F[3] = id0; F[4] = id1; F[5] = rcp(d2);
}
__device__ __forceinline__ void ldl3_solve(const double (&F)[6], double b0, double b1, double b2, double &x0, double &x1, double &x2) {
  const double z0 = b0, z1 = b1 - F[0] * z0, z2 = b2 - F[1] * z0 - F[2] * z1;
  x2 = z2 * F[5];
  x1 = z1 * F[4] - F[2] * x2;
  x0 = z0 * F[3] - F[0] * x1 - F[1] * x2;
}

// ---- LDS: one 64-wide row per per-lane scalar; a lane only ever touches its own column ----
// (row offsets: L_LL = 0, L_LU = NR, L_ISL = 2 NR, L_ISU = 3 NR, four reduction rows behind them)
template <bool FULL> __host__ __device__ constexpr int lds_rows() { return 4 * rows_kept<FULL>() + 4; }

// Reductions over the S lanes of a group: every lane publishes 4 values, then reads its
// group's S entries in batches of RB (reads issued back to back, one wait per batch; fixed order
// -> bit-reproducible).  OPn: 0 sum, 1 max, 2 min.
struct Red4 { double a, b, c, d; };
enum { RB = 10 };   // entries read per batch (pairs of lanes, see group_reduce): 20 segments = 1 batch
template <int OP> __device__ __forceinline__ double red_init() { return OP == 0 ? 0.0 : OP == 1 ? -1e300 : 1e300; }
// (max / min of values that come back from LDS or a DPP move carry a v_max_f64 x, x, x each -- IEEE-mode
// canonicalisation the compiler cannot prove away, ~90 instructions per iteration.  Writing the instruction as inline
// asm removes them and costs far more: the "v" constraints pin operands that now live in AGPRs and the allocator
// answers with 476 B of scratch per lane, 5.57 -> 8.27 ms.  Measured, rejected.)
// (Inline-asm v_max_f64 / v_min_f64 to get rid of the canonicalising v_max_f64 x, x, x in front of maxima of LDS / DPP
//  values: tried in the lean form too, round 4 -- the allocator answers with 708 B of scratch per lane.  Rejected again.)
template <int OP> __device__ __forceinline__ double red_op(double acc, double v, bool in_range) {
  // (padded slots repeat entry S-1: harmless for max / min; a sum takes them with weight 0 -- one fused multiply-add with
  //  a scalar 0 / 1 instead of a two-instruction select and an add; fma(v, 1, acc) rounds as acc + v does)
  if constexpr (OP == 0) return __builtin_fma(v, in_range ? 1.0 : 0.0, acc);
  else if constexpr (OP == 1) return fmax(acc, v);
  else return fmin(acc, v);
}
// One wavefront per workgroup and LDS operations of a wavefront complete in issue order, so publishing and
// reading need no s_barrier and no wait in between: a wavefront-scope fence keeps the compiler from reordering.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int OP> __device__ __forceinline__ double pair_op(double v, bool has_next) {
  const double n = dpp_next(v);
  if constexpr (OP == 0) return v + (has_next ? n : 0.0);
  else if constexpr (OP == 1) return has_next ? fmax(v, n) : v;
  else return has_next ? fmin(v, n) : v;
}
// Every lane first combines its value with its right neighbour's (DPP, if that lane belongs to the same group), so
// only the even lanes' entries have to be read back: half the LDS reads and half the combining operations.
template <int O0, int O1, int O2, int O3>
__device__ __forceinline__ Red4 group_reduce(double (*red)[64], int lane, int gbase, int k, int S, double v0, double v1,
                                             double v2, double v3) {
  const bool has_next = k + 1 < S;
  v0 = pair_op<O0>(v0, has_next); v1 = pair_op<O1>(v1, has_next); v2 = pair_op<O2>(v2, has_next); v3 = pair_op<O3>(v3, has_next);
  wave_lds_sync();
  red[0][lane] = v0; red[1][lane] = v1; red[2][lane] = v2; red[3][lane] = v3;
  wave_lds_sync();
  Red4 r = {red_init<O0>(), red_init<O1>(), red_init<O2>(), red_init<O3>()};
  const int n2 = (S + 1) >> 1;   // pairs (the last one may be a single lane)
  for (int j0 = 0; j0 < n2; j0 += RB) {
    double a[RB], b[RB], c[RB], d[RB];
    UNROLL for (int u = 0; u < RB; u++) {
      const int j = gbase + 2 * (j0 + u < n2 ? j0 + u : n2 - 1);
      a[u] = red[0][j]; b[u] = red[1][j]; c[u] = red[2][j]; d[u] = red[3][j];
    }
    UNROLL for (int u = 0; u < RB; u++) {
      const bool in = j0 + u < n2;
      r.a = red_op<O0>(r.a, a[u], in); r.b = red_op<O1>(r.b, b[u], in);
      r.c = red_op<O2>(r.c, c[u], in); r.d = red_op<O3>(r.d, d[u], in);
    }
  }
  return r;
}

// Two values instead of four (the step ratios of the lean form: half the LDS traffic and combining operations).
template <int O0, int O1>
__device__ __forceinline__ Red4 group_reduce2(double (*red)[64], int lane, int gbase, int k, int S, double v0, double v1) {
  const bool has_next = k + 1 < S;
  v0 = pair_op<O0>(v0, has_next); v1 = pair_op<O1>(v1, has_next);
  wave_lds_sync();
  red[0][lane] = v0; red[1][lane] = v1;
  wave_lds_sync();
  Red4 r = {red_init<O0>(), red_init<O1>(), 0.0, 0.0};
  const int n2 = (S + 1) >> 1;
  for (int j0 = 0; j0 < n2; j0 += RB) {
    double a[RB], b[RB];
    UNROLL for (int u = 0; u < RB; u++) {
      const int j = gbase + 2 * (j0 + u < n2 ? j0 + u : n2 - 1);
      a[u] = red[0][j]; b[u] = red[1][j];
    }
    UNROLL for (int u = 0; u < RB; u++) {
      const bool in = j0 + u < n2;
      r.a = red_op<O0>(r.a, a[u], in); r.b = red_op<O1>(r.b, b[u], in);
    }
  }
  return r;
}

// The same reduction with the maxima and minima through LDS atomics (ds_max_f64 / ds_min_f64 on the group's own entry:
// order-independent, so the result is the one of the ordered loop, bit for bit) and only the sums through the ordered
// reads -- a sum's rounding depends on its order.  OPn: 0 sum, 1 max, 2 min, -1 unused (returns 0).  Lanes outside
// every group (the idle tail of a wavefront) take no part: their values are not those of a real segment.
template <int OP> __device__ __forceinline__ void red_atomic(double *slot, double v) {
  if constexpr (OP == 1) __hip_atomic_fetch_max(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  else if constexpr (OP == 2) __hip_atomic_fetch_min(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
template <int O0, int O1, int O2, int O3>
__device__ __forceinline__ Red4 group_reduce_mixed(double (*red)[64], int lane, int gbase, int k, int S, bool in_group,
                                                   double v0, double v1, double v2, double v3) {
  const bool has_next = k + 1 < S;
  if constexpr (O0 == 0) v0 = pair_op<0>(v0, has_next);
  if constexpr (O1 == 0) v1 = pair_op<0>(v1, has_next);
  if constexpr (O2 == 0) v2 = pair_op<0>(v2, has_next);
  if constexpr (O3 == 0) v3 = pair_op<0>(v3, has_next);
  wave_lds_sync();
  // sums: every lane's entry; maxima / minima: the entry of the group's first lane, set to the identity by that lane
  if constexpr (O0 == 0) red[0][lane] = v0; else if constexpr (O0 > 0) { if (k == 0) red[0][lane] = red_init<O0>(); }
  if constexpr (O1 == 0) red[1][lane] = v1; else if constexpr (O1 > 0) { if (k == 0) red[1][lane] = red_init<O1>(); }
  if constexpr (O2 == 0) red[2][lane] = v2; else if constexpr (O2 > 0) { if (k == 0) red[2][lane] = red_init<O2>(); }
  if constexpr (O3 == 0) red[3][lane] = v3; else if constexpr (O3 > 0) { if (k == 0) red[3][lane] = red_init<O3>(); }
  wave_lds_sync();   // (LDS operations of a wavefront complete in issue order: the atomics see the identities)
  if (in_group) {
    if constexpr (O0 > 0) red_atomic<O0>(&red[0][gbase], v0);
    if constexpr (O1 > 0) red_atomic<O1>(&red[1][gbase], v1);
    if constexpr (O2 > 0) red_atomic<O2>(&red[2][gbase], v2);
    if constexpr (O3 > 0) red_atomic<O3>(&red[3][gbase], v3);
  }
  wave_lds_sync();
  Red4 r = {0.0, 0.0, 0.0, 0.0};
  if constexpr (O0 > 0) r.a = red[0][gbase];
  if constexpr (O1 > 0) r.b = red[1][gbase];
  if constexpr (O2 > 0) r.c = red[2][gbase];
  if constexpr (O3 > 0) r.d = red[3][gbase];
  if constexpr (O0 == 0 || O1 == 0 || O2 == 0 || O3 == 0) {
    const int n2 = (S + 1) >> 1;
    for (int j0 = 0; j0 < n2; j0 += RB) {
      double a[RB], b[RB], c[RB], d[RB];
      UNROLL for (int u = 0; u < RB; u++) {
        const int j = gbase + 2 * (j0 + u < n2 ? j0 + u : n2 - 1);
        if constexpr (O0 == 0) a[u] = red[0][j];
        if constexpr (O1 == 0) b[u] = red[1][j];
        if constexpr (O2 == 0) c[u] = red[2][j];
        if constexpr (O3 == 0) d[u] = red[3][j];
      }
      UNROLL for (int u = 0; u < RB; u++) {
        const bool in = j0 + u < n2;
        if constexpr (O0 == 0) r.a = red_op<0>(r.a, a[u], in);
        if constexpr (O1 == 0) r.b = red_op<0>(r.b, b[u], in);
        if constexpr (O2 == 0) r.c = red_op<0>(r.c, c[u], in);
        if constexpr (O3 == 0) r.d = red_op<0>(r.d, d[u], in);
      }
    }
  }
  return r;
}

// Long form (MULTI in ipm_solve_body: more than 64 segments, one axis problem per WORKGROUP of up to four wavefronts):
// the reduction goes over all the workgroup's lanes through a shared block rm[4][256] behind the 8 seam slots of wgs.
enum { WGS_SEAM = 8, WGS_LANES = 256 };
template <bool MULTI, int O0, int O1, int O2, int O3>
__device__ __forceinline__ Red4 reduce4(double (*red)[64], double *wgs, int lane, int wv, int gbase, int k, int S, double v0,
                                        double v1, double v2, double v3) {
  if constexpr (!MULTI) {
    return group_reduce<O0, O1, O2, O3>(red, lane, gbase, k, S, v0, v1, v2, v3);
  } else {
    double *rm = wgs + WGS_SEAM;
    const int K = wv * 64 + lane;
    __syncthreads();
    rm[K] = v0; rm[WGS_LANES + K] = v1; rm[2 * WGS_LANES + K] = v2; rm[3 * WGS_LANES + K] = v3;
    __syncthreads();
    Red4 r = {red_init<O0>(), red_init<O1>(), red_init<O2>(), red_init<O3>()};
    for (int j = 0; j < S; j++) {   // (fixed order: bit-reproducible; every lane reads the same addresses: broadcasts)
      r.a = red_op<O0>(r.a, rm[j], true); r.b = red_op<O1>(r.b, rm[WGS_LANES + j], true);
      r.c = red_op<O2>(r.c, rm[2 * WGS_LANES + j], true); r.d = red_op<O3>(r.d, rm[3 * WGS_LANES + j], true);
    }
    return r;
  }
}

}  // namespace btrapz
#endif
