// btrapz_kernels.hip -- gfx950 (CDNA4) kernels of the batched Bezier-in-corridor QP path.
//
// What one launch computes, per candidate corridor (reference call sites in /root/reference):
//   FormulateProblem   src/solve_3d.cc:1143-1229   (P,q,A,l,u from S Cube records)
//   CalculateKernel    src/solve_3d.cc:70-224      P_k = 2(t^3 MQM0 + t MQM1 + MQM2/t + MQM3/t^3)
//   CalculateOffset    src/solve_3d.cc:226-321     q_k
//   CalculateAffineConstraint  src/solve_3d.cc:779-1129 (trapezoid), src/cuboid_3d.cc:632-988
//   osqp_setup + osqp_solve    src/solve_3d.cc:1246,1249  (external OSQP) -> replaced, see below
//   acceptance         src/solve_3d.cc:1251-1277
//
// The reference hands the assembled sparse QP to OSQP (ADMM).  On these problems ADMM needs
// hundreds to thousands of iterations and stops ~1e-3 away from the optimum; the QP is
// strictly convex (unique x*), so any exact method returns the same answer.  This kernel
// uses the QP's structure instead:
//   * the two axes (s, l) share no row and no P entry -> 2B independent axis problems;
//   * every equality row is C2-continuity at a joint or the initial state
//     (solve_3d.cc:896-949): the feasible set of the equalities is parametrised EXACTLY by
//     the joint states X_j = (p, v, a), j = 1..S (null-space form, no equality multipliers);
//   * every inequality row touches one segment only -> the Newton matrix of a primal-dual
//     (Mehrotra predictor-corrector) interior-point method in X is block tridiagonal, 3x3 blocks.
// Mapping (packed form, the throughput kernel): one lane per segment, floor(64/S) axis problems per wavefront.  Per lane
// and kept row (15 of a segment's 18: rows_kept) the interior-point state is a slack pair in VGPRs and a multiplier
// pair plus a reciprocal-slack pair in LDS columns private to the lane (30 + 30 + 30 doubles); a wavefront holds problems
// of ONE axis so the batch-invariant M'QM table comes through scalar loads; neighbour exchange is a DPP wave shift; the
// block LDL^T and its sweeps run two-sided (upper and lower half towards the middle block) in S/2+1 sequential steps in
// which lanes s and S-1-s own the pivots.  HBM is touched once to load the Cube records and once to store results.
// Split form (few candidates, at most 21 segments; SPLIT in ipm_solve_body): ONE axis problem per wavefront, the rows
// of every segment spread over three lanes, state in registers only, one scalar per row and phase exchanged through LDS.
// Long form (65..256 segments; MULTI): one axis problem per workgroup of up to four wavefronts.  Large cold batches run
// as two launches (CAPPED / RESUME): groups left alone in their wavefront hand their iterate over and are re-packed.
// Batches of three and more wavefronts per SIMD (cold or warm-started, up to 64 segments) run the LEAN form of the same
// iteration instead -- btrapz_lean_body.h: two wavefronts per SIMD on half the per-lane state; this file's packed form
// serves the smaller batches, the rescue pass, the candidate queue and btrapz_options.start = 1.
#include <hip/hip_runtime.h>
#include "btrapz_ipm.h"

namespace btrapz {

// -----------------------------------------------------------------------------------------
// WARM = false: the cold-start kernel (bench path).  WARM = true adds the warm start and the cold restart of a
// group whose guess did not pay off; a separate instantiation, so the cold kernel keeps its register footprint.
// ELASTIC = true: every inequality row l <= g'x <= u becomes l <= g'x - d <= u with the penalty d^2 / (2 delta) in the
// objective (the rescue pass of btrapz_options.elastic, see the header).  Eliminating d (d = delta * (lambda_u -
// lambda_l)) leaves the same iteration with three changes per row: the row value is g'c - delta (lambda_u - lambda_l),
// the row's weight in the Newton matrix is w / (1 + delta w), and the step of the row value is
// (g'dc - delta b) / (1 + delta w), b the row's entry of the right-hand side.  delta = 0 is the plain method.
// The relaxation of a row is measured in the row's own norm: delta_r = elastic_delta * |g_r|^2 (position rows t^2,
// velocity 50, acceleration 2400, jerk 72000), i.e. the penalty is on d_r / |g_r|, the distance the control points
// would have to move -- a violation of 0.01 in that unit is 0.01 t metres on a position row, 0.07 on a velocity row,
// 0.49 on an acceleration row (what the reference's own status-2 iterate on src/c7.txt shows), 2.7 on a jerk row.
// wave_id: the wavefront's number in the launch (wavefront w solves axis w & 1 of the candidates of pair w >> 1);
// lds: this wavefront's private [L_ROWS][64] block of LDS; lane: 0..63.
// QUEUE = true (uniform cold batches): the wavefront is persistent.  Its group slots draw candidates from a counter in
// device memory; a slot whose candidate has finished writes it back and takes the next one at the top of the loop,
// while the other slots keep iterating -- a wavefront no longer runs as long as its slowest group for every group it
// holds (on the scenario_1 batch the slowest of three needs 11 iterations against a mean of 9.7).  The counter is read
// one candidate ahead, so its latency is hidden behind ten iterations; which slot solves a candidate has no influence
// on its result.
// SPLIT = true (uniform cold batches, S <= 21): the latency form.  ONE candidate per wavefront; its three lane groups
// hold the same axis problem and share its row work -- lane (j, k), j = 0..2, owns five of segment k's fifteen rows
// (slots: jerk row j, acceleration row j + 1, velocity row j + 1, and two of the remaining position / velocity rows, see
// split_owner_*), keeps their slacks and multipliers in registers and publishes one scalar per row and phase through
// LDS; every lane then gathers the fifteen scalars of its segment and forms gradient, Newton block and right-hand
// sides exactly as the one-lane-per-segment form does, so the block LDL^T, its sweeps, the reductions and every
// decision run (redundantly, bit for bit alike) in all three groups.  Row passes cost a third; everything sequential
// costs the same per wavefront but serves one problem instead of three -- lower latency for few candidates, lower
// throughput for many (measured: DESIGN.md 3.6).
// MULTI = true (uniform cold batches of 65..256 segments): the long form.  ONE axis problem per workgroup of ceil(S / 64)
// wavefronts, lane l of wavefront wv = segment 64 wv + l.  Same iteration; what crosses a wavefront's edge goes through
// LDS behind a workgroup barrier: the neighbour exchange at the seams (lane 63 -> lane 0 of the next wavefront and
// back: two barriers per shifted value) and the reductions (reduce4).  Every lane of the workgroup follows the problem's
// termination decisions (lanes beyond S compute garbage nobody reads and write nothing), so all wavefronts leave the
// loop together.  Correctness over speed: the reference has no limit on the segment count (std::vector,
// solve_3d.cc:323-486), its bundled inputs have at most 14.
// CAPPED / RESUME (uniform cold batches much larger than the machine; btrapz_options.cap_iter): two launches instead of
// one.  A wavefront runs as long as its slowest group, and a launch as long as its last wavefront: on the scenario_1
// batch the same candidates sorted by iteration count (hard ones first) take 5.67 ms instead of 7.02.  The order cannot
// be known in advance (neither the cold start's nor the unconstrained optimum's violations predict the count), but it
// can be produced: the CAPPED launch stops every group at cap_iter iterations -- a group that has not converged by then
// SUSPENDS: it writes its iterate (joint states, slacks, multipliers, best iterate and the termination bookkeeping:
// SUSP_FIELDS doubles per lane) to a slot of a.susp_state and reports BTRAPZ_SUSPENDED -- and the RESUME launch picks
// the suspended axis problems up from per-axis lists, bucketed by how far from convergence they were (the machinery of
// the rescue pass; ragged batches: by segment count), restores the iterate and carries on: the same iterates, bit for
// bit, as the one-launch solve.
enum { SUSP_FIELDS = 3 + 4 * 15 + 3 + 8 };
template <bool WARM, bool ORDERED, bool ELASTIC = false, bool QUEUE = false, bool SPLIT = false, bool MULTI = false,
          bool CAPPED = false, bool RESUME = false>
__device__ __forceinline__ void ipm_solve_body(const KernelArgs &a, const double *__restrict__ mqm, double (*lds)[64],
                                               const int wave_id, const int lane, double *wgs = nullptr, const int wv = 0) {
  static_assert(!QUEUE || (!WARM && !ORDERED && !ELASTIC), "the queue serves uniform cold batches");
  static_assert(!SPLIT || (!ORDERED && !ELASTIC && !QUEUE), "the split form serves uniform batches");
  static_assert(!MULTI || (!WARM && !ORDERED && !QUEUE && !SPLIT), "the long form serves uniform cold batches (and their rescue pass) -- and, launched once per segment count with a candidate list in a.order of a.bucket_S entries, the long candidates of a ragged batch");
  static_assert(!(CAPPED || RESUME) || (!WARM && !ELASTIC && !QUEUE && !SPLIT && !MULTI && !(CAPPED && RESUME)), "capped / resume: packed cold form");
  static_assert(!RESUME || ORDERED, "the resume pass reads its problems from per-axis lists");
  constexpr bool PERAXIS = ELASTIC || RESUME;   // one candidate list and one set of bucket tables per axis
  // value of the previous / next segment's lane (0 beyond the ends of the wavefront -- or, long form, of the workgroup)
  auto from_prev = [&](double x) -> double {
    double r = dpp_prev(x);
    if constexpr (MULTI) {
      __syncthreads();
      if (lane == 63) wgs[wv] = x;
      __syncthreads();
      if (lane == 0 && wv > 0) r = wgs[wv - 1];
    }
    return r;
  };
  auto from_next = [&](double x) -> double {
    double r = dpp_next(x);
    if constexpr (MULTI) {
      __syncthreads();
      if (lane == 0) wgs[4 + wv] = x;
      __syncthreads();
      if (lane == 63 && wv < 3) r = ((wv + 1) * 64 < (int)blockDim.x) ? wgs[4 + wv + 1] : 0.0;
    }
    return r;
  };
  constexpr bool CACHE_RP = !ELASTIC;   // see the main loop
  constexpr bool FULL = ELASTIC;                 // rows kept: see rows_kept()
  constexpr int NR = rows_kept<FULL>();
  constexpr int L_LL = 0, L_LU = NR, L_ISL = 2 * NR, L_ISU = 3 * NR, L_RED = SPLIT ? 17 : 4 * NR;
  [[maybe_unused]] constexpr int L_XCH = 0;   // split form: 15 row scalars + 2 statistics per lane, rewritten phase by phase
  // the axis is wave-uniform
  const int axis = __builtin_amdgcn_readfirstlane(wave_id & 1);
  int S, pair = wave_id >> 1, ncand = a.B, cand0 = 0;
  if constexpr (ORDERED) {
    // ragged batch: candidates are bucketed by segment count; find this wave's bucket (wave-uniform)
    // (rescue pass: one set of tables and one candidate list per axis, the stalled axis problems only)
    const int *wave_prefix = a.wave_prefix + (PERAXIS ? axis * 198 : 0), *cand_prefix = a.cand_prefix + (PERAXIS ? axis * 198 : 0);
    if (pair >= wave_prefix[65]) return;
    // slot s with wave_prefix[s] <= pair < wave_prefix[s + 1] (empty buckets repeat their prefix): binary search,
    // six dependent scalar loads instead of up to 63
    int s = 1, hi = 65;
    while (hi - s > 1) {
      const int mid = (s + hi) >> 1;
      if (wave_prefix[mid] <= pair) s = mid; else hi = mid;
    }
    S = a.bucket_S ? a.bucket_S : 65 - s;   // slot s holds key 65 - s (longest first); hint mode: classes of a uniform batch
    pair -= wave_prefix[s]; cand0 = cand_prefix[s] + (PERAXIS ? axis * a.B : 0); ncand = cand_prefix[s + 1] - cand_prefix[s];
  } else {
    S = a.S;
  }
  S = __builtin_amdgcn_readfirstlane(S);
  const int gpw = SPLIT ? 3 : MULTI ? 1 : 64 / S;   // split form: three copies of one problem (3 S <= 64, the host checks)
  const int kk = MULTI ? wv * 64 + lane : lane;     // long form: the workgroup's lanes are the segments
  const int g = kk / S;
  const int k = kk - g * S;
  const bool lane_in_group = g < gpw;
  const int gl = lane_in_group ? g : gpw - 1;
  const int gbase = gl * S;
  const bool first = (k == 0), last = (k == S - 1);
  const int m = S >> 1;                               // root block of the two-sided elimination
  const bool top = k < m, bot = k > m, mid = k == m;
  const int my_step = top ? k : (bot ? S - 1 - k : m);  // the step at which this lane owns a pivot
  const size_t BS = (size_t)a.B * a.seg_stride;
  const double *sg = a.seg;
  const Shared &sh = a.sh;
  const int variant = sh.variant;
  // P block (solve_3d.cc:159-171) from the batch-invariant MQM_d = M' pQp_d M: the table is
  // wave-uniform (one axis per wave) -> scalar loads.
  const double *__restrict__ mq = mqm + axis * 84;
  const size_t lam_row = (size_t)a.B * a.seg_stride;
  const double eps = a.eps;
  [[maybe_unused]] const double edelta0 = ELASTIC ? a.elastic_delta : 0.0;
  // |g_r|^2 of row r (solve_3d.cc:823-888: t c_i ; 5 (c_i+1 - c_i) ; 20 (1, -2, 1) ; 60 (-1, 3, -3, 1))
#define ROW_N2(r) ((r) < 6 ? t2 : (r) < 11 ? 50.0 : (r) < 15 ? 2400.0 : 72000.0)
#define ED(r) (edelta0 * ROW_N2(r))
  const double inv_m = 1.0 / ((double)(2 * NR) * (double)S);

  // ---- everything that belongs to the candidate a group is solving (set by begin_candidate) ----
  int b = 0;                       // candidate
  bool valid = false;              // this lane holds a segment of a real candidate
  size_t lam_e = 0;                // its slot in the warm-start arrays
  double t = 1.0, it = 1.0, t2 = 1.0, t3 = 1.0, it3 = 1.0, pend = 0.0;
  NullMap nm = {1.0, 0.05};
  double plo0 = 0.0, dplo = 0.0, phi0 = 0.0, dphi = 0.0, vlo[5], vhi[5], alo = 0.0, ahi = 0.0, jlo = 0.0, jhi = 0.0;
  double mplo = 0.0, mphi = 0.0, mvlo = 0.0, mvhi = 0.0;
  double q[6], Xinit[3], Pk[21], qn = 0.0, bnorm = 0.0;
  bool infeasible_bounds = false, no_solution = false;
  double X[3] = {0.0, 0.0, 0.0}, Xcold0 = 0.0, sl[NR], su[NR];
  // split form: this lane's five rows -- slacks, multipliers, bounds -- and where the lanes of its segment sit
  [[maybe_unused]] double sl5[5], su5[5], ll5[5], lu5[5], lo5[5], up5[5];
  [[maybe_unused]] const int sj = gl;                         // which third of the rows
  [[maybe_unused]] const int Lj[3] = {k, S + k, 2 * S + k};   // lanes (0, k), (1, k), (2, k)
  // (two flags the optimiser cannot trace back to one index: it otherwise turns every three-way choice into an array
  //  on the stack indexed by j -- scratch loads in the middle of every row pass)
  [[maybe_unused]] int sjf0 = sj == 0, sjf1 = sj == 1;
  asm volatile("" : "+v"(sjf0), "+v"(sjf1));
  [[maybe_unused]] const bool sj0 = sjf0 != 0, sj1 = sjf1 != 0;
#define SEL3(x0, x1, x2) (sj0 ? (x0) : sj1 ? (x1) : (x2))
  // rows of the lane's five slots (split_owner_*), for the warm-start arrays
  [[maybe_unused]] const int slot_row[5] = {15 + sj, 12 + sj, 7 + sj, 1 + 2 * sj, sj == 2 ? 10 : 2 + 2 * sj};
  // values of the lane's five rows for control points c (the expressions of row_dot, on the lane's window c[j..j+3])
  [[maybe_unused]] auto slot_vals = [&](const double (&cc_)[6], double (&v)[5]) {
    // (values pinned in registers first: a choice between elements of an array in memory becomes a load from a chosen
    //  address, and the array then lives in scratch)
    double c_[6];
    UNROLL for (int i = 0; i < 6; i++) c_[i] = cc_[i];
    opaque6(c_);
    const double w0 = SEL3(c_[0], c_[1], c_[2]), w1 = SEL3(c_[1], c_[2], c_[3]), w2 = SEL3(c_[2], c_[3], c_[4]), w3 = SEL3(c_[3], c_[4], c_[5]);
    v[0] = 60.0 * ((w3 - w0) + 3.0 * (w1 - w2));            // jerk row 15 + j
    v[1] = 20.0 * ((w1 - 2.0 * w2) + w3);                   // acceleration row 12 + j
    v[2] = 5.0 * (w2 - w1);                                 // velocity row 7 + j
    v[3] = t * SEL3(c_[1], c_[3], c_[5]);                   // position rows 1, 3, 5
    const double pv = t * (sj0 ? c_[2] : c_[4]), vv = 5.0 * (c_[5] - c_[4]);
    v[4] = SEL3(pv, pv, vv);                                // position rows 2, 4; velocity row 10
  };
  // scalar `base + slot` of row r of this lane's segment, as published by the lane that owns the row
#define XR(base, r) lds[L_XCH + (base) + split_owner_slot(r)][Lj[split_owner_j(r)]]
  bool warm_started = false, restarted = true;   // a warm-started group that stalls gets ONE cold restart
  int it0 = 0;                                   // iteration at which the current start was made
  double best_score = 1e300, Xb[3] = {0.0, 0.0, 0.0};
  int best_it = 0, iters = 0, tiny_steps = 0;   // tiny_steps: consecutive steps shorter than BTRAPZ_TINY_STEP (btrapz_ipm.h)
  float best_res = 3e38f;          // smallest residual part of the score so far and when (the stall test)
  int res_it = 0;
  bool plain = false;              // second chance of a solve whose complementarity is stuck: see the corrector
  [[maybe_unused]] bool lone_start = false;   // capped launch: the group was the only live one of its wavefront at the first iteration
  [[maybe_unused]] bool slot_refused = false; // capped launch: the workspace had no hand-over slot left for this group
  // btrapz_options.start = 1: before the first iteration one Newton step of the UNCONSTRAINED problem (all row weights
  // zero: the block system is Phi' P Phi, its solution the optimum without the inequality rows), slacks re-initialised
  // there.  The pass is the loop body up to the predictor's sweep; it is not counted as an iteration.
  [[maybe_unused]] bool unc_pass = false;
  bool done = true;
  [[maybe_unused]] bool suspended = false;   // capped launch: the group has handed its iterate over to the resume launch
  [[maybe_unused]] bool handed_over = RESUME;   // resume launch, first pass: the iterate was evaluated by the capped launch
  [[maybe_unused]] int cand_live = 0;   // long form: the workgroup has a candidate
  // The group's own iteration count.  (Warm-start instantiations: when one group of the wavefront restarts cold the
  // others lose that pass of the loop; queue: the groups of a wavefront are at different iterations.  A candidate's
  // stall / step-rule / iteration bookkeeping must not depend on which candidates share its wavefront.)
  int eit = 0;

  // bounds of the rows as the reference assembles them ...
#define LO0(r) ((r) < 6 ? plo0 + (double)(r) * dplo : (r) < 11 ? vlo[((r) >= 6 && (r) < 11) ? (r) - 6 : 0] : (r) < 15 ? alo : jlo)
#define UP0(r) ((r) < 6 ? phi0 + (double)(r) * dphi : (r) < 11 ? vhi[((r) >= 6 && (r) < 11) ? (r) - 6 : 0] : (r) < 15 ? ahi : jhi)
  // ... and of the rows this solve keeps (rows_kept): the last position and velocity row of a segment (5, 10) carry the
  // intersection with the next segment's first ones (0, 6), which state the same joint quantity; the acceleration rows
  // 14 / 11 are one and the same interval for a, scaled by the two durations.
#define LO(r) ((!FULL && (r) == 5) ? mplo : (!FULL && (r) == 10) ? mvlo : LO0(r))
#define UP(r) ((!FULL && (r) == 5) ? mphi : (!FULL && (r) == 10) ? mvhi : UP0(r))
#define LOAD_P(H)                                                                                     \
  UNROLL for (int i_ = 0; i_ < 21; i_++)                                                              \
    H[i_] = 2.0 * (t3 * mq[i_] + t * mq[21 + i_] + it * mq[42 + i_] + it3 * mq[63 + i_]);             \
  H[SYM(5, 5)] += pend;
#define HSYM(H, i, j) ((i) <= (j) ? H[SYM(i, j)] : H[SYM(j, i)])
#define LL(r) lds[L_LL + SI(r)][lane]
#define LU(r) lds[L_LU + SI(r)][lane]
  // cold start of this lane: slacks max(gap, BTRAPZ_COLD_SLACK), multipliers BTRAPZ_COLD_LAMBDA at the current X
  // (btrapz_ipm.h; the rescue pass: 1 and 1)
  auto init_slacks = [&]() {
    constexpr double smin0 = ELASTIC ? 1.0 : BTRAPZ_COLD_SLACK, lam0 = ELASTIC ? 1.0 : BTRAPZ_COLD_LAMBDA;
    double Xp[3], c[6];
    UNROLL for (int i = 0; i < 3; i++) { const double v = from_prev(X[i]); Xp[i] = first ? Xinit[i] : v; }
    U_apply(nm, Xp, c[0], c[1], c[2]);
    V_apply(nm, X, c[3], c[4], c[5]);
    if constexpr (SPLIT) {
      double v5[5];
      slot_vals(c, v5);
      UNROLL for (int i = 0; i < 5; i++) {
        sl5[i] = fmax(v5[i] - lo5[i], smin0); su5[i] = fmax(up5[i] - v5[i], smin0);
        ll5[i] = lam0; lu5[i] = lam0;
      }
      // rows without a real bound (btrapz_ipm.h, "bounds that are no bounds"): the centred multiplier.  A pass of its own
      // behind a wave-uniform branch that is all but never taken -- selects inside the loop above cost the warm-start
      // instantiations, which call this from their restart path too, 36 B of scratch per lane.
      bool far_row = false;
      UNROLL for (int i = 0; i < 5; i++) far_row |= sl5[i] > BTRAPZ_COLD_FAR || su5[i] > BTRAPZ_COLD_FAR;
      if (__any(far_row)) {
        UNIFORM_BLOCK;
        UNROLL for (int i = 0; i < 5; i++) { ll5[i] = cold_lambda(sl5[i], smin0, lam0); lu5[i] = cold_lambda(su5[i], smin0, lam0); }
      }
    } else {
      FOR_ROWS(r)
        const double gc_r = row_dot<r>(c, t);
        sl[SI(r)] = fmax(gc_r - LO(r), smin0); su[SI(r)] = fmax(UP(r) - gc_r, smin0);
        LL(r) = lam0; LU(r) = lam0;
      END_ROWS
      bool far_row = false;
      FOR_ROWS(r)
        far_row |= sl[SI(r)] > BTRAPZ_COLD_FAR || su[SI(r)] > BTRAPZ_COLD_FAR;
      END_ROWS
      if (__any(far_row)) {   // (as above)
        UNIFORM_BLOCK;
        FOR_ROWS(r)
          LL(r) = cold_lambda(sl[SI(r)], smin0, lam0); LU(r) = cold_lambda(su[SI(r)], smin0, lam0);
        END_ROWS
      }
    }
  };
  auto cold_start = [&]() {
    X[0] = Xcold0; X[1] = Xinit[1]; X[2] = 0.0;
    init_slacks();
  };

  // ---- capped / resume: a group's iterate in a.susp_state, slot-major, field i of lane k at [slot][i][k] ----
  [[maybe_unused]] auto state_io = [&](const bool store, const long long slot) {
    double *base = a.susp_state + (size_t)slot * SUSP_FIELDS * a.seg_stride + k;
    const size_t fs = a.seg_stride;
    int f = 0;
    auto io = [&](double &v) { if (store) base[(size_t)f * fs] = v; else v = base[(size_t)f * fs]; ++f; };
    UNROLL for (int i = 0; i < 3; i++) io(X[i]);
    UNROLL for (int i = 0; i < 3; i++) io(Xb[i]);
    FOR_ROWS(r)
      io(sl[SI(r)]); io(su[SI(r)]);
      double l_ = LL(r), u_ = LU(r);
      io(l_); io(u_);
      if (!store) { LL(r) = l_; LU(r) = u_; }
    END_ROWS
    double g8[8] = {best_score, (double)best_it, (double)best_res, (double)res_it, (plain ? 1.0 : 0.0) + 2.0 * (double)tiny_steps, (double)eit, (double)it0, (double)iters};
    UNROLL for (int i = 0; i < 8; i++) io(g8[i]);
    if (!store) {
      best_score = g8[0]; best_it = (int)g8[1]; best_res = (float)g8[2]; res_it = (int)g8[3]; plain = ((int)g8[4] & 1) != 0; tiny_steps = (int)g8[4] >> 1;
      eit = (int)g8[5]; it0 = (int)g8[6]; iters = (int)g8[7];
    }
  };

  // What a lane reads of its candidate: its segment's fields for the wavefront's axis and the candidate's per-axis
  // records (coalesced: lanes -> consecutive (b, k)).  A plain value so that the queue can hold the NEXT candidate's
  // record in registers while the current one iterates (its loads are issued a whole solve ahead of their use).
  struct Record { double t, lb, ls, ub, us, begl, endl, v[10], skew, bias, ref_end, init[3]; };
  auto load_record = [&](const int b_) {
    // (only the fields the wavefront's axis and variant use are read: loading all of them in one basic block -- one
    //  memory round trip instead of one per branch -- was measured on the single-candidate launch, whose record comes
    //  over PCIe: no difference, 162.6 against 162.9 us, and 1.31 instead of 1.15 times the algorithmic HBM bytes)
    Record r;
    const size_t e_ = (size_t)b_ * a.seg_stride + k;
    r.t = sg[BTRAPZ_F_T * BS + e_];
    r.begl = 0.0; r.endl = 0.0;
    if (axis == 0) {
      r.lb = sg[BTRAPZ_F_DOWN_BIAS * BS + e_]; r.ls = sg[BTRAPZ_F_DOWN_SKEW * BS + e_];
      r.ub = sg[BTRAPZ_F_UPP_BIAS * BS + e_];  r.us = sg[BTRAPZ_F_UPP_SKEW * BS + e_];
      r.v[0] = sg[BTRAPZ_F_DS_LO * BS + e_]; r.v[1] = sg[BTRAPZ_F_DS_HI * BS + e_];
      UNROLL for (int i = 2; i < 10; i++) r.v[i] = 0.0;
    } else {
      r.lb = sg[BTRAPZ_F_L_DOWN_BIAS * BS + e_]; r.ls = sg[BTRAPZ_F_L_DOWN_SKEW * BS + e_];
      r.ub = sg[BTRAPZ_F_L_UPP_BIAS * BS + e_];  r.us = sg[BTRAPZ_F_L_UPP_SKEW * BS + e_];
      if (variant == BTRAPZ_CUBOID) { r.begl = sg[BTRAPZ_F_BEG_L * BS + e_]; r.endl = sg[BTRAPZ_F_END_L * BS + e_]; }
      UNROLL for (int i = 0; i < 10; i++) r.v[i] = a.dl_bounds[(size_t)b_ * 10 + i];
    }
    r.skew = sg[(axis == 0 ? BTRAPZ_F_X_SKEW : BTRAPZ_F_Y_SKEW) * BS + e_];
    r.bias = sg[(axis == 0 ? BTRAPZ_F_X_BIAS : BTRAPZ_F_Y_BIAS) * BS + e_];
    r.ref_end = a.ref_end[(size_t)b_ * 2 + axis];
    UNROLL for (int i = 0; i < 3; i++) r.init[i] = a.init[(size_t)b_ * 6 + axis * 3 + i];
    return r;
  };

  // Sets up the solve of candidate b_new from its record in the lanes of the calling group(s).  (Called once for the
  // whole wavefront, or -- queue -- under the mask of the groups that take a new candidate: everything in here is per
  // lane or per group; the DPP reads across a group's last lane are masked by `last`.)
  auto begin_candidate = [&](const int b_new, const bool valid_new, const Record &rec) {
    b = b_new; valid = valid_new;
    t = rec.t;
    it = 1.0 / t;
    nm = NullMap{it, t * 0.05};
    // position rows: lo_i = plo0 + i*dplo , up_i = phi0 + i*dphi  (solve_3d.cc:827-828,965-966)
    {
      const double lb = rec.lb, ls = rec.ls, ub = rec.ub, us = rec.us;
      plo0 = lb; dplo = ls * 0.2 * t; phi0 = ub; dphi = us * 0.2 * t;
      if (variant == BTRAPZ_CUBOID) {
        if (axis == 0) {  // cuboid_3d.cc:677-689: inscribed interval, clamped to [0,100].  Its max / min run over
          // bias + skew * (i / 5) * t, i = 0..5, and skip a NaN term: with an infinite slope the i = 0 term is NaN
          // (inf * 0), not the bias -- hence `skew * 0.0` spelled out; the terms between the ends add nothing
          const double lo = fmax(0.0, fmax(ls * 0.0 + lb, lb + ls * t));
          const double hi = fmin(100.0, fmin(us * 0.0 + ub, ub + us * t));
          plo0 = lo; phi0 = hi;
        } else {  // cuboid_3d.cc:826-827
          plo0 = rec.begl; phi0 = rec.endl;
        }
        dplo = 0.0; dphi = 0.0;
      }
    }
    // velocity rows (solve_3d.cc:835-859 s axis; :1003-1004 l axis: dy_bounds_[i], i = row index)
    if (axis == 0) {
      UNROLL for (int i = 0; i < 5; i++) { vlo[i] = rec.v[0]; vhi[i] = rec.v[1]; }
    } else {
      UNROLL for (int i = 0; i < 5; i++) { vlo[i] = rec.v[2 * i]; vhi[i] = rec.v[2 * i + 1]; }
    }
    // acceleration / jerk rows (solve_3d.cc:862-888, 1010-1037)
    alo = (axis == 0 ? sh.acc_s[0] : sh.acc_l[0]) * t; ahi = (axis == 0 ? sh.acc_s[1] : sh.acc_l[1]) * t;
    jlo = (axis == 0 ? sh.jerk_s[0] : sh.jerk_l[0]) * t * t; jhi = (axis == 0 ? sh.jerk_s[1] : sh.jerk_l[1]) * t * t;
    const double far_cut = move_far_bounds(plo0, dplo, phi0, dphi, vlo, vhi);   // bounds that are none (btrapz_ipm.h)
    mplo = plo0 + 5.0 * dplo; mphi = phi0 + 5.0 * dphi; mvlo = vlo[4]; mvhi = vhi[4];
    bool joint_empty = false;
    if constexpr (!FULL) {
      const double nplo = from_next(plo0), nphi = from_next(phi0), nvlo = from_next(vlo[0]), nvhi = from_next(vhi[0]);
      if (!last) {
        const bool own_ok = mplo <= mphi && mvlo <= mvhi && nplo <= nphi && nvlo <= nvhi;
        mplo = fmax(mplo, nplo); mphi = fmin(mphi, nphi); mvlo = fmax(mvlo, nvlo); mvhi = fmin(mvhi, nvhi);
        // Two consistent rows that contradict each other -- by more than round-off: where the corridor changes lane the
        // two sides often just touch, and the same boundary computed as bias + skew * t on one side and as a bias on the
        // other differs in the last digit ([-1.2, 1.6] | [1.6000000000000001, 3.8]: "empty" by 2e-16; the reference's
        // solver and the oracle put the joint on the boundary).  Such a joint is pinned to the common point.
        const double ptol = 1e-9 * (1.0 + fmax(fabs(mplo), fabs(mphi))), vtol = 1e-9 * (1.0 + fmax(fabs(mvlo), fabs(mvhi)));
        joint_empty = own_ok && (mplo > mphi + ptol || mvlo > mvhi + vtol);
        if (mplo > mphi && !(mplo > mphi + ptol)) { mplo = 0.5 * (mplo + mphi); mphi = mplo; }
        if (mvlo > mvhi && !(mvlo > mvhi + vtol)) { mvlo = 0.5 * (mvlo + mvhi); mvhi = mvlo; }
      }
    }
    if constexpr (SPLIT) {   // bounds of this lane's five rows (split_owner_*)
      lo5[0] = LO(15); up5[0] = UP(15); lo5[1] = LO(12); up5[1] = UP(12);
      lo5[2] = SEL3(LO(7), LO(8), LO(9)); up5[2] = SEL3(UP(7), UP(8), UP(9));
      lo5[3] = SEL3(LO(1), LO(3), LO(5)); up5[3] = SEL3(UP(1), UP(3), UP(5));
      lo5[4] = SEL3(LO(2), LO(4), LO(10)); up5[4] = SEL3(UP(2), UP(4), UP(10));
    }
    t3 = t * t * t; it3 = it * it * it; t2 = t * t;
    pend = last ? 2.0 * (axis == 0 ? sh.weight_end_s : sh.weight_end_l) * t2 : 0.0;  // :164-168
    // q block (solve_3d.cc:248-268): q_p in the monomial basis, then q_p * M
    {
      const double skew = rec.skew, bias = rec.bias;
      const double wr = axis == 0 ? sh.w_s[0] : sh.w_l[0], wd = axis == 0 ? sh.w_s[1] : sh.w_l[1];
      const double dref = axis == 0 ? sh.ds_ref : sh.dl_ref;
      double qp[6];
      UNROLL for (int i = 0; i < 6; i++) {
        qp[i] = -2.0 * (t * t * t) * wr * skew / (double)(i + 2) - 2.0 * (t * t) * wr * bias / (double)(i + 1);
        if (i > 0) qp[i] += -2.0 * wd * dref * t;
      }
      // M (Bernstein -> monomial, solve_3d.cc:122-127), row = power, col = control point
      q[0] = qp[0] - 5.0 * qp[1] + 10.0 * qp[2] - 10.0 * qp[3] + 5.0 * qp[4] - qp[5];
      q[1] = 5.0 * qp[1] - 20.0 * qp[2] + 30.0 * qp[3] - 20.0 * qp[4] + 5.0 * qp[5];
      q[2] = 10.0 * qp[2] - 30.0 * qp[3] + 30.0 * qp[4] - 10.0 * qp[5];
      q[3] = 10.0 * qp[3] - 20.0 * qp[4] + 10.0 * qp[5];
      q[4] = 5.0 * qp[4] - 5.0 * qp[5];
      q[5] = qp[5];
      if (last) q[5] -= dref * 2.0 * rec.ref_end * t;  // :268/:315 (multiplies by d_ref: bug-compatible)
    }
    UNROLL for (int i = 0; i < 3; i++) Xinit[i] = rec.init[i];

    // ---------------- consistency of the bounds -------------------------------------------
    double gapmin = 1e300;
    bnorm = 0.0; qn = 0.0;
    static_for<18>([&](auto r_c) {          // (all rows of the reference: a row with l > u is reported as such)
      constexpr int r = decltype(r_c)::value;
      // ... by more than 1e-12 (the tolerance of the test suite's exact solver, so that both decide alike): a lower line that reaches the
      // upper bound exactly at a control point (0.4 + 0.5 * 0.6 against 0.7) gives l = 0.7000000000000004 >
      // u = 0.7000000000000001, an equality in all but the last bit (round-3 fuzz campaign: 1 call in 32 000 was refused
      // for it)
      // |bounds|: the real ones (a moved bound sits at far_cut exactly; header limits at BTRAPZ_FAR_LIMIT x t, x t^2)
      const double cut = r < 11 ? far_cut : r < 15 ? BTRAPZ_FAR_LIMIT * t : BTRAPZ_FAR_LIMIT * t * t;
      const double al = fabs(LO0(r)), au = fabs(UP0(r));
      const double rb = fmax(al < cut ? al : 0.0, au < cut ? au : 0.0);
      gapmin = fmin(gapmin, (UP0(r) - LO0(r)) + 1e-12);
      bnorm = fmax(bnorm, rb);
    });
    // Rows that no iterate can change: segment 0's first position / velocity / acceleration row state the given
    // initial state (c0, c1, c2 of segment 0 follow from it alone), and a joint whose two sides leave no common value.
    // If they cannot be met the problem has no solution: the solve stops before its first iteration with the status of
    // a stalled one (which the rescue pass, btrapz_options.elastic, then takes over).
    bool no_solution_lane = false;
    if constexpr (!FULL) {
      double c0, c1, c2;
      U_apply(nm, Xinit, c0, c1, c2);
      auto outside = [&](double g_, double lo, double hi) {
        const double tol = 1e-7 * (1.0 + fmax(fabs(lo), fabs(hi)));
        return !(g_ >= lo - tol && g_ <= hi + tol);
      };
      no_solution_lane = (first && (outside(t * c0, LO0(0), UP0(0)) || outside(5.0 * (c1 - c0), LO0(6), UP0(6)) ||
                                    outside(20.0 * ((c0 - 2.0 * c1) + c2), LO0(11), UP0(11)))) || joint_empty;
    }
    UNROLL for (int i = 0; i < 6; i++) qn = fmax(qn, fabs(q[i]));
    // ---------------- starting point: constant-velocity propagation of the initial state, or the caller's
    // joint states (warm start; a lane whose values are not finite keeps the cold start) -----
    {
      double tsum = 0.0;
      if constexpr (MULTI) {
        double *rm = wgs + WGS_SEAM;
        __syncthreads();
        rm[kk] = t;
        __syncthreads();
        for (int j = 0; j < S; j++) tsum += (j <= k) ? rm[j] : 0.0;
      } else {
        wave_lds_sync();
        lds[L_RED][lane] = t;
        wave_lds_sync();
        for (int j = 0; j < S; j++) tsum += (j <= k) ? lds[L_RED][gbase + j] : 0.0;
      }
      Xcold0 = Xinit[0] + Xinit[1] * tsum;
    }
    {
      const Red4 r0 = reduce4<MULTI, 0, 1, 1, 2>(lds + L_RED, wgs, lane, wv, gbase, k, S, no_solution_lane ? 1.0 : 0.0, bnorm, qn, gapmin);
      bnorm = r0.b; qn = r0.c; gapmin = r0.d;
      no_solution = r0.a > 0.0;
    }
    infeasible_bounds = !(gapmin >= 0.0) || !(t > 0.0);
    if constexpr (MULTI) {   // one decision for the whole workgroup (a segment without duration counts like an empty row)
      const Red4 rt = reduce4<MULTI, 0, 1, 1, 1>(lds + L_RED, wgs, lane, wv, gbase, k, S, 0.0, (valid && !(t > 0.0)) ? 1.0 : 0.0, 0.0, 0.0);
      infeasible_bounds = !(gapmin >= 0.0) || rt.b > 0.0;
    }

    // multipliers of this lane in the warm-start arrays: [axis][row 0..35][b][k]
    lam_e = (size_t)axis * 36 * lam_row + (size_t)b * a.seg_stride + k;
    warm_started = WARM && (a.x0 || a.lam0);
    if (WARM && warm_started) {
      // warm: slacks floored at smin, multipliers = earlier multipliers (sanitised) + mu0 / s so that every
      // complementarity product is at least mu0
      X[0] = Xcold0; X[1] = Xinit[1]; X[2] = 0.0;
      if (a.x0) {
        const double *xw = a.x0 + (((size_t)b * 2 + axis) * a.seg_stride + k) * 3;
        const double w0 = xw[0], w1 = xw[1], w2 = xw[2];
        if (fabs(w0) < 1e300 && fabs(w1) < 1e300 && fabs(w2) < 1e300) { X[0] = w0; X[1] = w1; X[2] = w2; }
      }
      double Xp[3], c[6];
      UNROLL for (int i = 0; i < 3; i++) { const double v = from_prev(X[i]); Xp[i] = first ? Xinit[i] : v; }
      U_apply(nm, Xp, c[0], c[1], c[2]);
      V_apply(nm, X, c[3], c[4], c[5]);
      const double smin = a.smin, mu0 = a.mu0;
      if constexpr (SPLIT) {
        double v5[5];
        slot_vals(c, v5);
        UNROLL for (int i = 0; i < 5; i++) {
          const double s_l = fmax(v5[i] - lo5[i], smin), s_u = fmax(up5[i] - v5[i], smin);
          double l_l = 0.0, l_u = 0.0;
          if (a.lam0) {
            const double p_l = a.lam0[lam_e + (size_t)slot_row[i] * lam_row], p_u = a.lam0[lam_e + (size_t)(18 + slot_row[i]) * lam_row];
            l_l = (p_l >= 0.0 && p_l < 1e300) ? p_l : 0.0; l_u = (p_u >= 0.0 && p_u < 1e300) ? p_u : 0.0;
          }
          sl5[i] = s_l; su5[i] = s_u;
          ll5[i] = l_l + mu0 * rcp(s_l); lu5[i] = l_u + mu0 * rcp(s_u);
        }
      } else {
      FOR_ROWS(r)
        const double gc_r = row_dot<r>(c, t);
        const double s_l = fmax(gc_r - LO(r), smin), s_u = fmax(UP(r) - gc_r, smin);
        double l_l = 0.0, l_u = 0.0;
        if (a.lam0) {
          const double p_l = a.lam0[lam_e + (size_t)r * lam_row], p_u = a.lam0[lam_e + (size_t)(18 + r) * lam_row];   // (layout of the header: 18 + 18 rows)
          l_l = (p_l >= 0.0 && p_l < 1e300) ? p_l : 0.0; l_u = (p_u >= 0.0 && p_u < 1e300) ? p_u : 0.0;
        }
        sl[SI(r)] = s_l; su[SI(r)] = s_u;
        LL(r) = l_l + mu0 * rcp(s_l); LU(r) = l_u + mu0 * rcp(s_u);
      END_ROWS
      }
    } else {
      cold_start();
    }
    restarted = !warm_started;
    it0 = 0; eit = 0;
    // P block of this lane's segment, once per solve: rebuilding it from the scalar table where it is needed (twice
    // per iteration: 168 FMAs and the spill traffic of 84 scalar doubles) was the price of an earlier, tighter
    // register budget; kept live it costs the allocator nothing measurable (6.67 -> 6.32 ms).
    LOAD_P(Pk)
    best_score = 1e300; Xb[0] = X[0]; Xb[1] = X[1]; Xb[2] = X[2];
    best_it = 0; iters = 0; best_res = 3e38f; res_it = 0; plain = false;
    done = !valid || infeasible_bounds || no_solution;
    if constexpr (MULTI) done = (cand_live == 0) || infeasible_bounds || no_solution;   // lanes beyond S follow the problem
    if constexpr (RESUME) {   // carry on where the capped launch stopped
      if (valid) state_io(false, a.susp_slot[2LL * b + axis]);
    }
#ifdef BTRAPZ_EXPERIMENTS   // (btrapz_options.start = 1: a measured loss, DESIGN 3.2 -- compiled into experiment builds only)
    unc_pass = !ELASTIC && !QUEUE && !RESUME && a.unc_start != 0 && !warm_started;
#endif
  };
  auto write_back = [&]() {
    if constexpr (CAPPED) {
      if (suspended) {
        if (valid && first) { a.axis_status[2LL * b + axis] = BTRAPZ_SUSPENDED; a.axis_iters[2LL * b + axis] = iters; a.axis_obj[2LL * b + axis] = 0.0; }
      }
    }
    // ---------------- write back: multipliers (warm start of a later solve), control points, objective/status ----
    if (WARM && SPLIT && a.lam_out && valid) {   // every lane of a segment writes its own five rows
      UNROLL for (int i = 0; i < 5; i++) {
        a.lam_out[lam_e + (size_t)slot_row[i] * lam_row] = ll5[i]; a.lam_out[lam_e + (size_t)(18 + slot_row[i]) * lam_row] = lu5[i];
      }
    }
    if (WARM && !SPLIT && a.lam_out && valid) {
      FOR_ROWS(r)
        a.lam_out[lam_e + (size_t)r * lam_row] = LL(r); a.lam_out[lam_e + (size_t)(18 + r) * lam_row] = LU(r);
      END_ROWS
    }
    if (WARM && a.lam_out && valid && (!SPLIT || g == 0)) {
      if constexpr (!FULL) {   // the rows this lane does not keep (their bounds live in the previous segment's last rows)
        UNROLL for (int r0 = 0; r0 < 3; r0++) {
          const int rr_ = r0 == 0 ? 0 : r0 == 1 ? 6 : 11;
          a.lam_out[lam_e + (size_t)rr_ * lam_row] = 0.0; a.lam_out[lam_e + (size_t)(18 + rr_) * lam_row] = 0.0;
        }
      }
    }
    if (WARM && a.x_out && valid && (!SPLIT || g == 0)) {   // joint states of the returned iterate: x0 of a later solve of a nearby problem
      double *xo = a.x_out + (((size_t)b * 2 + axis) * a.seg_stride + k) * 3;
      xo[0] = Xb[0]; xo[1] = Xb[1]; xo[2] = Xb[2];
    }
    {
      double Xp[3], c[6];
      UNROLL for (int i = 0; i < 3; i++) { const double v = from_prev(Xb[i]); Xp[i] = first ? Xinit[i] : v; }
      U_apply(nm, Xp, c[0], c[1], c[2]);
      V_apply(nm, Xb, c[3], c[4], c[5]);
      double obj = 0.0;
      double Pm[21];
      UNROLL for (int i_ = 0; i_ < 21; i_++) Pm[i_] = Pk[i_];
      UNROLL for (int i = 0; i < 6; i++) {
        double s = 0.0;
        UNROLL for (int j = 0; j < 6; j++) s += HSYM(Pm, i, j) * c[j];
        obj += c[i] * (0.5 * s + q[i]);
      }
      // rescue pass: largest violation of an original row by the returned control points -- in the row's own unit
      // (viol) and per class of rows (position, velocity, acceleration, jerk: vcls), and in the unit of the penalty,
      // divided by |g_r| (vnorm): the acceptance test below is on the latter
      double viol = 0.0, vnorm = 0.0;
      [[maybe_unused]] double vcls[4] = {0.0, 0.0, 0.0, 0.0};
      if constexpr (ELASTIC) {
        const double inorm[4] = {it, 0.1414213562373095, 0.02041241452319315, 0.003726779962499649};   // 1 / |g_r|
        FOR_ROWS(r)
          constexpr int cls = r < 6 ? 0 : r < 11 ? 1 : r < 15 ? 2 : 3;
          const double gcr = row_dot<r>(c, t);
          const double v = fmax(LO(r) - gcr, gcr - UP(r));
          viol = fmax(viol, v); vcls[cls] = fmax(vcls[cls], v); vnorm = fmax(vnorm, v * inorm[cls]);
        END_ROWS
      }
      const Red4 ro = reduce4<MULTI, 0, 1, 1, 1>(lds + L_RED, wgs, lane, wv, gbase, k, S, obj, viol, vnorm, 0.0);
      [[maybe_unused]] Red4 rv = {0.0, 0.0, 0.0, 0.0};
      if constexpr (ELASTIC) rv = reduce4<MULTI, 1, 1, 1, 1>(lds + L_RED, wgs, lane, wv, gbase, k, S, vcls[0], vcls[1], vcls[2], vcls[3]);
      if (valid && (!SPLIT || g == 0) && !(CAPPED && suspended)) {
        // control points in the reference's order: s axis (6 S), then l axis (6 S); rows are 12*seg_stride apart
        double *dst = a.ctrl + (size_t)b * 12 * a.seg_stride + (size_t)axis * 6 * S + (size_t)k * 6;
        UNROLL for (int i = 0; i < 6; i++) dst[i] = c[i];
        if (first) {
          int st;
          if (infeasible_bounds) st = BTRAPZ_PRIMAL_INFEASIBLE;
          else if (best_score < 1e-7) st = BTRAPZ_SOLVED;
          else if (best_score < 1e-5 || (res_it < 0 && best_score < 1e-4)) st = BTRAPZ_SOLVED_INACCURATE;   // res_it < 0: ended at the dual floor
          else st = BTRAPZ_MAX_ITER_REACHED;
          if constexpr (ELASTIC) {
            // converged on the relaxed problem: feasible after all (rows kept to 1e-7) -> as solved; least violation
            // within the caller's tolerance -> the reference's "solved inaccurate"; beyond it -> infeasible
            if (st > 0 && ro.b > 1e-7 * (1.0 + bnorm)) st = ro.c <= a.elastic_tol ? BTRAPZ_SOLVED_INACCURATE : BTRAPZ_PRIMAL_INFEASIBLE;
          }
          // an objective that is not finite is no solution, whatever the score says (a corridor with an infinite bound
          // assembles to rows of inf / NaN; the reference refuses a solve whose objective is NaN, solve_3d.cc:1251-1253)
          if (st > 0 && !(fabs(ro.a) < 1e300)) st = BTRAPZ_MAX_ITER_REACHED;
          const long long prob = 2LL * b + axis;
          if constexpr (ELASTIC) {
            if (a.axis_viol) {   // per class of rows, in the rows' own units (0 where the rows hold)
              double *vo = a.axis_viol + prob * 4;
              vo[0] = fmax(rv.a, 0.0); vo[1] = fmax(rv.b, 0.0); vo[2] = fmax(rv.c, 0.0); vo[3] = fmax(rv.d, 0.0);
            }
          }
          a.axis_obj[prob] = ro.a;
          a.axis_status[prob] = st;
          a.axis_iters[prob] = ELASTIC ? iters + a.axis_iters[prob] + 1 : iters;   // rescue: on top of the first attempt's
        }
      }
    }
  };

  // ---- which candidate(s) this wavefront solves ----
  // queue: this slot has drawn an index beyond the batch (the idle tail lanes of a wavefront never draw one)
  [[maybe_unused]] bool retired = QUEUE && !lane_in_group;
  [[maybe_unused]] bool has_cand = false;    // queue: the slot holds a candidate whose result is still to be written
  // queue: the index the slot takes at its next refill, drawn one solve ahead (held by the group's first lane, so the
  // counter's latency is hidden).  (Holding the next candidate's whole record in registers as well was tried: the
  // loads have to be parked in AGPRs, which waits for them on the spot -- slower.)
  [[maybe_unused]] int next_cand = 0;
  if constexpr (QUEUE) {
    if (first && lane_in_group) next_cand = atomicAdd(a.queue + axis, 1);
  } else {
    long long cand = (SPLIT || MULTI) ? (long long)pair : (long long)pair * gpw + gl;
    // (long form with a list: the candidates of a ragged batch that have a.S > 64 segments -- one launch per count,
    //  btrapz_host.hip; the list's length travels in a.bucket_S, which this form has no other use for)
    const bool listed = MULTI && a.order != nullptr;
    if (listed) ncand = a.bucket_S;
    const bool valid0 = lane_in_group && cand < ncand;
    cand_live = cand < ncand ? 1 : 0;
    if (cand >= ncand) cand = ncand - 1;
    const int b0 = ORDERED ? a.order[cand0 + (int)cand] : listed ? a.order[(int)cand] : (int)cand;
    begin_candidate(b0, valid0, load_record(b0));
  }

  for (;;) {
    if constexpr (QUEUE) {
      // A slot whose candidate is finished (or that has none yet) writes it back and takes the next one; the other
      // slots of the wavefront wait for the ~0.2 iterations this takes and go on where they were.
      const bool refill = done && !retired;
      if (__any(refill)) {
        if (refill) {
          if (has_cand) write_back();
          wave_lds_sync();
          if (first && lane_in_group) lds[L_RED][lane] = (double)next_cand;
          wave_lds_sync();
          const int cnew = (int)lds[L_RED][gbase];
          if (cnew < a.B) {
            if (first && lane_in_group) next_cand = atomicAdd(a.queue + axis, 1);   // (needed a whole solve from now)
            begin_candidate(cnew, lane_in_group, load_record(cnew)); has_cand = true;
          } else { retired = true; has_cand = false; valid = false; done = true; }
        }
        if (__all(retired)) break;
      }
    }
    // ---- 1. control points of this segment, rows, residuals, gradient ----
    // Row residuals r_l = G c - s_l - l, r_u = G c + s_u - u: constant within an iteration and needed by six row
    // loops.  Cached (36 doubles; the allocator parks them in AGPRs) they save ~100 instructions per row loop:
    // 6.29 -> 6.04 ms.  The warm-start instantiations carry more state and would spill, so they recompute.
    double c[6], gc[6], rpl_[(CACHE_RP && !SPLIT) ? NR : 1], rpu_[(CACHE_RP && !SPLIT) ? NR : 1];
    [[maybe_unused]] double rpl5[5], rpu5[5], isl5[5], isu5[5];   // split form: residuals and reciprocal slacks of the own rows
    double mu_part = 0.0, rp_part = 0.0, dscale = 0.0;
    {
      double Xp[3];
      UNROLL for (int i = 0; i < 3; i++) { const double v = from_prev(X[i]); Xp[i] = first ? Xinit[i] : v; }
      U_apply(nm, Xp, c[0], c[1], c[2]);
      V_apply(nm, X, c[3], c[4], c[5]);
      UNROLL for (int i = 0; i < 6; i++) gc[i] = 0.0;
      if constexpr (SPLIT) {
        // own rows: residuals, reciprocal slacks, and the three scalars per row the segment's gradient, Newton block and
        // predictor right-hand side are made of -- published, then gathered by every lane of the segment
        double v5[5];
        slot_vals(c, v5);
        wave_lds_sync();
        UNROLL for (int i = 0; i < 5; i++) {
          rpl5[i] = v5[i] - sl5[i] - lo5[i]; rpu5[i] = v5[i] + su5[i] - up5[i];
          rp_part = fmax(rp_part, fmax(fabs(rpl5[i]), fabs(rpu5[i])));
          mu_part += sl5[i] * ll5[i] + su5[i] * lu5[i];
          isl5[i] = rcp(sl5[i]); isu5[i] = rcp(su5[i]);
          const double wl = ll5[i] * isl5[i], wu = lu5[i] * isu5[i];
          lds[L_XCH + i][lane] = lu5[i] - ll5[i];
          lds[L_XCH + 5 + i][lane] = wl + wu;
          lds[L_XCH + 10 + i][lane] = wl * (sl5[i] + rpl5[i]) - wu * (su5[i] - rpu5[i]);
        }
        lds[L_XCH + 15][lane] = mu_part; lds[L_XCH + 16][lane] = rp_part;
        wave_lds_sync();
        FOR_ROWS(r)
          row_scatter<r>(XR(0, r), t, gc);
        END_ROWS
        mu_part = (lds[L_XCH + 15][Lj[0]] + lds[L_XCH + 15][Lj[1]]) + lds[L_XCH + 15][Lj[2]];
        rp_part = fmax(fmax(lds[L_XCH + 16][Lj[0]], lds[L_XCH + 16][Lj[1]]), lds[L_XCH + 16][Lj[2]]);
      } else {
      FOR_ROWS(r)
        const double ll = LL(r), lu = LU(r);
        double gcr = row_dot<r>(c, t);
        if constexpr (ELASTIC) gcr -= ED(r) * (lu - ll);
        const double rpl = gcr - sl[SI(r)] - LO(r), rpu = gcr + su[SI(r)] - UP(r);
        if constexpr (CACHE_RP) { rpl_[CACHE_RP ? SI(r) : 0] = rpl; rpu_[CACHE_RP ? SI(r) : 0] = rpu; }
        rp_part = fmax(rp_part, fmax(fabs(rpl), fabs(rpu)));
        mu_part += sl[SI(r)] * ll + su[SI(r)] * lu;
        row_scatter<r>(lu - ll, t, gc);
      END_ROWS
      }
      UNROLL for (int i = 0; i < 6; i++) dscale = fmax(dscale, fabs(gc[i]));
      double Pm[21];
      UNROLL for (int i_ = 0; i_ < 21; i_++) Pm[i_] = Pk[i_];
      UNROLL for (int i = 0; i < 6; i++) {
        double s = 0.0;
        UNROLL for (int j = 0; j < 6; j++) s += HSYM(Pm, i, j) * c[j];
        dscale = fmax(dscale, fabs(s));
        gc[i] += s + q[i];
      }
    }
    // reduced gradient for X_{k+1}: V' gc[3..5] + (lane k+1) U' gc[0..2]
    double rd_part;
    {
      double rd[3], un[3];
      VT_apply(nm, gc[3], gc[4], gc[5], rd);
      UT_apply(nm, gc[0], gc[1], gc[2], un);
      UNROLL for (int i = 0; i < 3; i++) { const double v = from_next(un[i]); rd[i] += last ? 0.0 : v; }
      rd_part = fmax(fabs(rd[0]), fmax(fabs(rd[1]), fabs(rd[2])));
    }
    // fmax/fmin ignore NaN: a lane whose residuals are not finite must poison its group's score, or a NaN
    // iterate would be ranked by mu alone and could be kept as the best one.
    if (!(mu_part == mu_part) || !(rd_part == rd_part) || !(rp_part == rp_part) || !(dscale == dscale) ||
        !(fabs(rd_part) < 1e300) || !(fabs(mu_part) < 1e300))
      rp_part = 1e300;
    const Red4 rr = reduce4<MULTI, 0, 1, 1, 1>(lds + L_RED, wgs, lane, wv, gbase, k, S, mu_part, rd_part, rp_part, dscale);
    const double mu = rr.a * inv_m;
    // KKT score: dual residual relative to (1+|q|) with a round-off floor, primal residual
    // relative to the bound scale, complementarity absolute.
    const double rd_eff = fmax(rr.b - 2e-13 * rr.d, 0.0);
    const double res = fmax(rd_eff / (1.0 + qn), rr.c / (1.0 + bnorm));
    const double score = (mu == mu) ? fmax(res, mu) : 1e300;   // (fmax would drop a NaN mu: ADVICE r4)
    const double mu_primal = fmax(mu, rr.c / (1.0 + bnorm));
    const bool feasible_and_complementary = mu_primal < 1e-7;   // (only the dual residual is left: see the dual floor below)
#ifdef BTRAPZ_TRACE   // debugging aid: one line per iteration and axis problem (tools: build with -DBTRAPZ_TRACE)
    if (first && lane_in_group && valid && !done && gl == 0) printf("trace axis %d b %d eit %d score %.3e res %.3e (dual %.3e primal %.3e) mu %.3e best %.3e@%d res_it %d | rd %.3e dscale %.3e qn %.3e bnorm %.3e\n", axis, b, eit, score, res, rd_eff / (1.0 + qn), rr.c / (1.0 + bnorm), mu, best_score, best_it, res_it, rr.b, rr.d, qn, bnorm);
#endif
    bool restart_now = false;
    if (!done && !unc_pass && !(RESUME && handed_over)) {
      iters = eit;
      if (score < best_score) { best_score = score; best_it = eit; Xb[0] = X[0]; Xb[1] = X[1]; Xb[2] = X[2]; }
      if ((float)res < a.stall_factor * best_res) { best_res = (float)res; res_it = eit; }
#ifndef ABL_FIXED
      // (the divergence test is for the plain problem: in the relaxed one the multipliers of rows that end up violated
      //  grow to violation / delta, 1e7 and more, and mu with them, on the way to the solution)
      // stop: converged; at the round-off floor (best < 1e-5, 3 iterations without progress); diverging or
      // infeasible (stall_len iterations without progress after the first stall_start); not finite.  "Progress" is a
      // smaller score OR residuals smaller by stall_factor: on a hard but solvable corridor (a run of 0.2-0.5 s segments)
      // the complementarity part of the score climbs for ten iterations while the residuals fall by a factor of 40 -- the
      // score alone called that a stall; an infeasible one leaves both where they are.  A warm-started group
      // that ends in the last two ways, or is still far from converged after 12 iterations (a useful guess
      // needs about 5, a cold start 8-14; below 1e-4 the method is in its fast final phase), or is still running
      // after 24, is restarted once from the cold start: a bad guess must neither turn a solvable candidate
      // into a failure nor cost more than a bounded number of iterations.
      const bool stalled = (eit - it0 >= a.stall_start && eit - best_it >= a.stall_len && eit - res_it >= a.stall_len) ||
                           (!ELASTIC && mu > (double)a.diverge_factor * best_score) || !(score < 1e299) ||
                           (!ELASTIC && tiny_steps >= BTRAPZ_TINY_STEPS);
      // (the dual floor: feasible to 1e-7 and complementary to 1e-7, the dual residual alone stuck between 1e-5 and
      //  1e-4 for three iterations -- the accuracy of the block elimination on a badly scaled corridor (a 0.1 s segment
      //  among 1 s ones), not of the iterate's position: "solved inaccurate", instead of iterating on until the slacks
      //  underflow.  Found by the round-3 fuzz campaign: 1 call in 16 000.)
      // (the Newton floor: below 1e-7 -- "solved" whatever follows -- with complementarity and feasibility a hundred
      //  times smaller than the score, the score IS the dual residual and the iteration a pure Newton step, which
      //  squares the residual or, at the accuracy of the block elimination, leaves it where it is.  An iterate that
      //  does not improve on such a best one ends the solve at once instead of after three: the typical scenario_1
      //  solve reached 2e-9 at iteration 6 and spent iterations 7-9 finding that out.)
      // (not in the rescue pass: the relaxed problem's x moves by 1e-3 while its score moves from 5e-9 to 1e-9)
      const int patience = (!ELASTIC && best_score < 1e-7 && mu_primal < 1e-2 * best_score) ? 1 : 3;   // iterations without a better score
      const bool at_floor = best_score < (feasible_and_complementary ? 1e-4 : 1e-5) && eit - best_it >= patience;
      if (at_floor && feasible_and_complementary) res_it = -1;   // (the mark write_back reads: the stall bookkeeping is over)
      if (score < eps || at_floor) done = true;
      else if (WARM && !restarted && (stalled || (eit - it0 >= 12 && best_score > 1e-4) || eit - it0 >= 24)) restart_now = true;
      else if (stalled) {
        // stalled with residuals at round-off level: only the complementarity is stuck (the two-cycle described at the
        // corrector).  One second chance without the second-order term on blocked iterations; an infeasible or
        // diverging solve (residuals stuck, score not finite) ends here.
        if (!plain && res < 1e-6 && score < 1e299) { plain = true; best_it = eit; res_it = eit; }
        else done = true;
      }
#endif
      // the iteration budget: the iterate just evaluated was the last one (a step nobody evaluates is not taken)
      if (!done && !restart_now && eit + 1 >= a.max_iter) done = true;
    }
    handed_over = false;
    if constexpr (CAPPED) {
      // the cap: a group still iterating after cap_iter iterations hands its iterate over to the resume launch.  (The
      // iterate has just been evaluated, its bookkeeping -- best iterate, stall marks, the second chance -- goes along;
      // the resume launch forms the Newton step of its first pass without evaluating the iterate a second time: a
      // second evaluation of a solve whose second chance was granted on this very iterate would end it.)
      // Who hands over: a group that is the only one of its wavefront still iterating after cap_iter iterations (the
      // wavefront would run at a third of its width for it), and any group still iterating cap_hi iterations in (a long
      // runner belongs at the front of a launch, not wherever the batch order put it).
      const int nact = __popcll(__ballot(first && lane_in_group && valid && !done));
      if (eit == 0) lone_start = nact <= a.cap_alone;   // (alone from the start: hands over after its first iteration, see the lean form)
      const bool want = !done && !unc_pass && valid && !slot_refused && ((eit >= (lone_start ? 1 : a.cap_iter) && nact <= a.cap_alone && score >= a.cap_score) || eit >= a.cap_hi);
      if (__any(want)) {
        UNIFORM_BLOCK;
        wave_lds_sync();
        if (want && first) lds[L_RED][lane] = (double)atomicAdd(a.susp_count, 1);
        wave_lds_sync();
        const long long slot = want ? (long long)lds[L_RED][gbase] : -1;
        if (want && slot < (long long)a.susp_cap) {   // (no room: the group simply goes on)
          state_io(true, slot);
          if (first) {
            // how far from convergence: the class the resume launch buckets by (far ones first, like with like)
            const double sc = fmax(fmin(score, 1e3), 1e-12);
            int cls = 1 + (int)(4.0 * (log10(sc) + 12.0));
            a.susp_slot[2LL * b + axis] = (int)slot;
            // (ragged batches: the resume lists are bucketed by segment count, as the batch itself is)
            a.susp_key[(size_t)axis * a.B + b] = (ORDERED && !a.bucket_S) ? S : (cls < 1 ? 1 : cls > 64 ? 64 : cls);
          }
          suspended = true; done = true;
        } else if (want) {
          slot_refused = true;   // no room: the group goes on to the end, and does not ask again
        }
      }
    }
    if constexpr (!QUEUE) { if (__all(done)) break; }
    if (WARM && __any(restart_now)) {
      // wave-uniform branch; the other groups of the wavefront only lose this iteration's Newton step.  A group
      // restarts as a whole (the score is group-uniform), so the DPP reads inside cold_start stay in the group.
      if (restart_now) {
        cold_start();
        best_score = 1e300; best_it = eit + 1; it0 = eit + 1; restarted = true; best_res = 3e38f; res_it = eit + 1; tiny_steps = 0;
        Xb[0] = X[0]; Xb[1] = X[1]; Xb[2] = X[2];
        ++eit;       // (the other groups did not take a step in this pass: their count stands)
      }
      continue;
    }

    // ---- 2. Newton matrix: H = P + G'WG ; M = Phi' H Phi ; block tridiagonal T, M01 ----
    // right-hand side in control-point space -> reduced to X space, sign flipped: u = -Phi' h
    auto reduce_rhs = [&](const double (&h)[6], double (&u)[3]) {
      double hn[3];
      VT_apply(nm, h[3], h[4], h[5], u);
      UT_apply(nm, h[0], h[1], h[2], hn);
      UNROLL for (int i = 0; i < 3; i++) { const double v = from_next(hn[i]); u[i] = -(u[i] + (last ? 0.0 : v)); }
    };
    double M01[9], T[6], up[3];   // up: the predictor's reduced right-hand side
    {
      double H[21], hp[6];
      UNROLL for (int i_ = 0; i_ < 21; i_++) H[i_] = Pk[i_];
      PHASE_FENCE(opaque6(c));
      // The predictor's right-hand side (rc = s*lambda -> tv = lambda_l (s_l + rp_l)/s_l - lambda_u (s_u - rp_u)/s_u)
      // needs exactly what this loop has in its hands -- the fresh reciprocals, the multipliers, G c -- so it is
      // accumulated here instead of in a row loop of its own (one pass over the rows and 72 LDS reads less).
      UNROLL for (int i = 0; i < 6; i++) hp[i] = gc[i];
      const double wscale = unc_pass ? 0.0 : 1.0;   // (finite operands: slacks and multipliers are those of the cold start)
      if constexpr (SPLIT) {
        FOR_ROWS(r)
          row_outer<r>(XR(5, r) * wscale, t2, H);
          row_scatter<r>(XR(10, r) * wscale, t, hp);
        END_ROWS
      } else {
      FOR_ROWS(r)
        const double isl = rcp(sl[SI(r)]), isu = rcp(su[SI(r)]);
        const double ll = LL(r), lu = LU(r);
        lds[L_ISL + SI(r)][lane] = isl; lds[L_ISU + SI(r)][lane] = isu;
        const double wl = ll * isl, wu = lu * isu;
        double rpl, rpu;
        if constexpr (CACHE_RP) { rpl = rpl_[CACHE_RP ? SI(r) : 0]; rpu = rpu_[CACHE_RP ? SI(r) : 0]; }
        else {
          double gcr = row_dot<r>(c, t);
          if constexpr (ELASTIC) gcr -= ED(r) * (lu - ll);
          rpl = gcr - sl[SI(r)] - LO(r); rpu = gcr + su[SI(r)] - UP(r);
        }
        if constexpr (ELASTIC) {
          const double ef = rcp(1.0 + ED(r) * (wl + wu));
          row_outer<r>((wl + wu) * ef, t2, H);
          row_scatter<r>((wl * (sl[SI(r)] + rpl) - wu * (su[SI(r)] - rpu)) * ef, t, hp);
        } else {
          row_outer<r>((wl + wu) * wscale, t2, H);
          row_scatter<r>((wl * (sl[SI(r)] + rpl) - wu * (su[SI(r)] - rpu)) * wscale, t, hp);
        }
      END_ROWS
      }
      reduce_rhs(hp, up);
      double w0[3], w1[3], w2[3], col[3], M00[6];
      // M00 = U' H00 U
      UT_apply(nm, H[SYM(0, 0)], H[SYM(0, 1)], H[SYM(0, 2)], w0);
      UT_apply(nm, H[SYM(0, 1)], H[SYM(1, 1)], H[SYM(1, 2)], w1);
      UT_apply(nm, H[SYM(0, 2)], H[SYM(1, 2)], H[SYM(2, 2)], w2);
      UT_apply(nm, w0[0], w1[0], w2[0], col); M00[0] = col[0]; M00[1] = col[1]; M00[2] = col[2];
      UT_apply(nm, w0[1], w1[1], w2[1], col); M00[3] = col[1]; M00[4] = col[2];
      UT_apply(nm, w0[2], w1[2], w2[2], col); M00[5] = col[2];
      // M01 = U' H01 V  (rows: X_k components, cols: X_{k+1} components)
      VT_apply(nm, H[SYM(0, 3)], H[SYM(0, 4)], H[SYM(0, 5)], w0);
      VT_apply(nm, H[SYM(1, 3)], H[SYM(1, 4)], H[SYM(1, 5)], w1);
      VT_apply(nm, H[SYM(2, 3)], H[SYM(2, 4)], H[SYM(2, 5)], w2);
      UNROLL for (int j = 0; j < 3; j++) {
        UT_apply(nm, w0[j], w1[j], w2[j], col);
        M01[0 * 3 + j] = col[0]; M01[1 * 3 + j] = col[1]; M01[2 * 3 + j] = col[2];
      }
      // M11 = V' H11 V
      VT_apply(nm, H[SYM(3, 3)], H[SYM(3, 4)], H[SYM(3, 5)], w0);
      VT_apply(nm, H[SYM(3, 4)], H[SYM(4, 4)], H[SYM(4, 5)], w1);
      VT_apply(nm, H[SYM(3, 5)], H[SYM(4, 5)], H[SYM(5, 5)], w2);
      VT_apply(nm, w0[0], w1[0], w2[0], col); T[0] = col[0]; T[1] = col[1]; T[2] = col[2];
      VT_apply(nm, w0[1], w1[1], w2[1], col); T[3] = col[1]; T[4] = col[2];
      VT_apply(nm, w0[2], w1[2], w2[2], col); T[5] = col[2];
      // diagonal block of X_{k+1}: M11_k + M00_{k+1}
      UNROLL for (int i = 0; i < 6; i++) { const double v = from_next(M00[i]); T[i] += last ? 0.0 : v; }
    }
    // ---- 3. two-sided block LDL^T.  Blocks 0..m-1 are eliminated downwards, blocks S-1..m+1 upwards, in the
    // same instruction stream (step s: lanes s and S-1-s); block m = S/2 is the root and takes both Schur
    // updates.  A lane that has factored its pivot S_k = T_k - Z_in forms K = S_k^{-1} Mc and Z_out = Mc' K for
    // its neighbour towards the middle (Mc = M01_{k+1} for the upper half, M01_k' for the lower half).
    double F[6] = {0.0, 0.0, 0.0, 1.0, 1.0, 1.0}, K[9], Z[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    double wp[3] = {0.0, 0.0, 0.0};   // forward-sweep vector of the predictor (see the loop)
    {
      double Mc[9];
      UNROLL for (int i = 0; i < 3; i++)
        UNROLL for (int j = 0; j < 3; j++) {
          const double nM = from_next(M01[i * 3 + j]);
          Mc[i * 3 + j] = top ? nM : M01[j * 3 + i];
          K[i * 3 + j] = 0.0;
        }
#if defined(ABL_NOSEQ) || defined(ABL_NOFACT)
      for (int step = m; step <= m; ++step) {
#else
      for (int step = 0; step <= m; ++step) {
#endif
        // No selects: a lane's Z is written at its own step and read (here, at the top of a step) only by the
        // neighbour towards the root one step later; the neighbour on the other side -- and, across a group
        // boundary, the neighbouring group's end lane, whose step is 0 -- still holds 0 when this lane's step
        // comes.  Only S <= 2 breaks that (the root is an end lane): a wave-uniform fix-up.
        double pZ[6], nZ[6], pw[3], nw[3];
        UNROLL for (int i = 0; i < 6; i++) { pZ[i] = from_prev(Z[i]); nZ[i] = from_next(Z[i]); }
        UNROLL for (int i = 0; i < 3; i++) { pw[i] = from_prev(wp[i]); nw[i] = from_next(wp[i]); }
        if (S <= 2) {
          UNIFORM_BLOCK;
          UNROLL for (int i = 0; i < 6; i++) { pZ[i] = first ? 0.0 : pZ[i]; nZ[i] = last ? 0.0 : nZ[i]; }
          UNROLL for (int i = 0; i < 3; i++) { pw[i] = first ? 0.0 : pw[i]; nw[i] = last ? 0.0 : nw[i]; }
        }
        if (step == my_step) {
          double Sk[6];
          UNROLL for (int i = 0; i < 6; i++) Sk[i] = T[i] - (pZ[i] + nZ[i]);
          UNROLL for (int i = 0; i < 3; i++) up[i] -= pw[i] + nw[i];   // the predictor's forward sweep rides along
          ldl3(Sk, F);
          if (!mid) {
            UNROLL for (int j = 0; j < 3; j++) ldl3_solve(F, Mc[j], Mc[3 + j], Mc[6 + j], K[j], K[3 + j], K[6 + j]);
            Z[0] = Mc[0] * K[0] + Mc[3] * K[3] + Mc[6] * K[6];
            Z[1] = Mc[0] * K[1] + Mc[3] * K[4] + Mc[6] * K[7];
            Z[2] = Mc[0] * K[2] + Mc[3] * K[5] + Mc[6] * K[8];
            Z[3] = Mc[1] * K[1] + Mc[4] * K[4] + Mc[7] * K[7];
            Z[4] = Mc[1] * K[2] + Mc[4] * K[5] + Mc[7] * K[8];
            Z[5] = Mc[2] * K[2] + Mc[5] * K[5] + Mc[8] * K[8];
            wp[0] = K[0] * up[0] + K[3] * up[1] + K[6] * up[2];
            wp[1] = K[1] * up[0] + K[4] * up[1] + K[7] * up[2];
            wp[2] = K[2] * up[0] + K[5] * up[1] + K[8] * up[2];
          }
        }
      }
    }

    // ---- 4. predictor (sigma = 0), then corrector ----
    // One solve: rhs u (already reduced to X space) -> dX by the block LDL^T sweeps -> dc.
    auto forward_u = [&](double (&u)[3]) {
      // forward, both halves towards the root: a lane whose u is final sends w = K' u inwards
      double w[3] = {0.0, 0.0, 0.0};
#if defined(ABL_NOSEQ) || defined(ABL_NOSWEEP)
      for (int step = m; step <= m; ++step) {
#else
      for (int step = 0; step <= m; ++step) {
#endif
        double pw[3], nw[3];
        UNROLL for (int i = 0; i < 3; i++) { pw[i] = from_prev(w[i]); nw[i] = from_next(w[i]); }
        if (S <= 2) { UNIFORM_BLOCK; UNROLL for (int i = 0; i < 3; i++) { pw[i] = first ? 0.0 : pw[i]; nw[i] = last ? 0.0 : nw[i]; } }
        if (step == my_step) {   // (same argument as in the factorisation: the unwanted neighbour still holds 0)
          UNROLL for (int i = 0; i < 3; i++) u[i] -= pw[i] + nw[i];
          if (!mid) {
            w[0] = K[0] * u[0] + K[3] * u[1] + K[6] * u[2];
            w[1] = K[1] * u[0] + K[4] * u[1] + K[7] * u[2];
            w[2] = K[2] * u[0] + K[5] * u[1] + K[8] * u[2];
          }
        }
      }
    };
    auto backward_u = [&](const double (&u)[3], double (&dX)[3], double (&dc)[6]) {
      // backward, from the root outwards: dX_k = S_k^{-1} u_k - K_k dX_(neighbour towards the root).  y carries
      // the FINAL dX of a lane (0 until then), so the neighbour away from the root contributes 0: an add, no select.
      ldl3_solve(F, u[0], u[1], u[2], dX[0], dX[1], dX[2]);
      double y[3];
      UNROLL for (int i = 0; i < 3; i++) y[i] = mid ? dX[i] : 0.0;
#if defined(ABL_NOSEQ) || defined(ABL_NOSWEEP)
      for (int step = 0; step >= 0 && m > 0; --step) {
#else
      for (int step = m - 1; step >= 0; --step) {
#endif
        double xin[3];
        double py[3], ny[3];
        UNROLL for (int i = 0; i < 3; i++) { py[i] = from_prev(y[i]); ny[i] = from_next(y[i]); }
        if (S <= 2) { UNIFORM_BLOCK; UNROLL for (int i = 0; i < 3; i++) { py[i] = first ? 0.0 : py[i]; ny[i] = last ? 0.0 : ny[i]; } }
        UNROLL for (int i = 0; i < 3; i++) xin[i] = py[i] + ny[i];
        if (step == my_step) {  // the root (my_step == m) is final already
          UNROLL for (int i = 0; i < 3; i++) {
            dX[i] -= K[3 * i] * xin[0] + K[3 * i + 1] * xin[1] + K[3 * i + 2] * xin[2];
            y[i] = dX[i];
          }
        }
      }
      double dXp[3];
      UNROLL for (int i = 0; i < 3; i++) { const double vv = from_prev(dX[i]); dXp[i] = first ? 0.0 : vv; }
      U_apply(nm, dXp, dc[0], dc[1], dc[2]);
      V_apply(nm, dX, dc[3], dc[4], dc[5]);
    };
    auto solve_dc = [&](const double (&h)[6], double (&dX)[3], double (&dc)[6]) {
      double u[3];
      reduce_rhs(h, u);
      forward_u(u);
      backward_u(u, dX, dc);
    };
    // per row: reciprocal slacks and multipliers from this lane's LDS column, residuals from c
    // The four LDS values of row r+1 are requested while row r is computed (one wavefront per SIMD: nobody else
    // hides the LDS latency); a scheduling barrier that only LDS reads may not cross keeps the compiler from
    // sinking the loads back to their uses (measured: 7.12 -> 6.98 ms; two rows ahead costs registers: 7.11; the
    // same in the residual and Newton-matrix loops, which read only the multipliers: 7.17).
    double pf_isl, pf_isu, pf_ll, pf_lu;
#define ROW_PREFETCH() do { pf_isl = lds[L_ISL][lane]; pf_isu = lds[L_ISU][lane]; pf_ll = lds[L_LL][lane]; pf_lu = lds[L_LU][lane]; } while (0)
#define ROW_BASE(r)                                                                               \
      const double isl = pf_isl, isu = pf_isu, ll = pf_ll, lu = pf_lu;                              \
      if constexpr (next_row<FULL>(r) >= 0) {                                                       \
        constexpr int rn_ = next_row<FULL>(r) >= 0 ? next_row<FULL>(r) : r;                         \
        pf_isl = lds[L_ISL + SI(rn_)][lane]; pf_isu = lds[L_ISU + SI(rn_)][lane];                   \
        pf_ll = LL(rn_); pf_lu = LU(rn_);                                                           \
        __builtin_amdgcn_sched_barrier(0x067F); /* anything but LDS reads may cross */               \
      }                                                                                             \
      double rpl, rpu;                                                                              \
      if constexpr (CACHE_RP) { rpl = rpl_[CACHE_RP ? SI(r) : 0]; rpu = rpu_[CACHE_RP ? SI(r) : 0]; }                  \
      else {                                                                                        \
        double gcr = row_dot<r>(c, t);                                                              \
        if constexpr (ELASTIC) gcr -= ED(r) * (lu - ll);                                           \
        rpl = gcr - sl[SI(r)] - LO(r); rpu = gcr + su[SI(r)] - UP(r);                                       \
      }
    // elastic rows: the step of the row value, g' dc -> (g' dc - delta b) / (1 + delta w)
#define ROW_STEP(r, gd, b) (ELASTIC ? ((gd) - ED(r) * (b)) * rcp(1.0 + ED(r) * (ll * isl + lu * isu)) : (gd))

    double dca[6], dX[3];
    double sigma_mu, second_order;   // second_order: minus the weight of the second-order term (second_order_factor, btrapz_ipm.h)
    if (__any(unc_pass)) {   // (the groups of a wavefront start together: a wave-uniform branch)
      UNIFORM_BLOCK;
      backward_u(up, dX, dca);
      if (unc_pass && !done && fabs(dX[0]) < 1e300 && fabs(dX[1]) < 1e300 && fabs(dX[2]) < 1e300) {
        UNROLL for (int i = 0; i < 3; i++) X[i] += dX[i];
      }
      // (a group takes the step as a whole or not at all: dX of a singular block system is not finite in every lane)
      {
        const Red4 rf = reduce4<MULTI, 0, 1, 1, 1>(lds + L_RED, wgs, lane, wv, gbase, k, S, 0.0,
                                                 (fabs(dX[0]) < 1e300 && fabs(dX[1]) < 1e300 && fabs(dX[2]) < 1e300) ? 0.0 : 1.0, 0.0, 0.0);
        if (rf.b > 0.0 && unc_pass && !done) { X[0] = Xcold0; X[1] = Xinit[1]; X[2] = 0.0; }
      }
      if (unc_pass) { init_slacks(); Xb[0] = X[0]; Xb[1] = X[1]; Xb[2] = X[2]; }
      unc_pass = false;
      continue;
    }
    if constexpr (SPLIT) {
      // The same predictor / corrector / step as below (see the comments there) on the lane's own five rows; what a
      // row contributes to a right-hand side, and the statistics of the three lanes of a segment, go through LDS.
      backward_u(up, dX, dca);
      double ga5[5];
      slot_vals(dca, ga5);
      double qmin = 0.0, qmax = -1.0, S1 = 0.0, S4 = 0.0;
      UNROLL for (int i = 0; i < 5; i++) {
        const double dsl = ga5[i] + rpl5[i], dsu = -ga5[i] - rpu5[i];
        const double ql = dsl * isl5[i], qu = dsu * isu5[i];
        qmin = fmin(qmin, fmin(ql, qu)); qmax = fmax(qmax, fmax(ql, qu));
        const double al = ll5[i] * dsl, au = lu5[i] * dsu;
        S1 += al + au;
        S4 += al * ql + au * qu;
      }
      wave_lds_sync();
      lds[L_XCH + 0][lane] = S1; lds[L_XCH + 1][lane] = S4; lds[L_XCH + 2][lane] = qmax; lds[L_XCH + 3][lane] = qmin;
      wave_lds_sync();
      S1 = (lds[L_XCH + 0][Lj[0]] + lds[L_XCH + 0][Lj[1]]) + lds[L_XCH + 0][Lj[2]];
      S4 = (lds[L_XCH + 1][Lj[0]] + lds[L_XCH + 1][Lj[1]]) + lds[L_XCH + 1][Lj[2]];
      qmax = fmax(fmax(lds[L_XCH + 2][Lj[0]], lds[L_XCH + 2][Lj[1]]), lds[L_XCH + 2][Lj[2]]);
      qmin = fmin(fmin(lds[L_XCH + 3][Lj[0]], lds[L_XCH + 3][Lj[1]]), lds[L_XCH + 3][Lj[2]]);
      const Red4 ra = reduce4<MULTI, 0, 0, 1, 2>(lds + L_RED, wgs, lane, wv, gbase, k, S, S1, S4, qmax, qmin);
      const double ap = 1.0 / fmax(-ra.d, 1.0), ad = 1.0 / fmax(1.0 + ra.c, 1.0);
      const double mua = ((1.0 - ad) * rr.a + (ap - ad - ap * ad) * ra.a - ap * ad * ra.b) * inv_m;
      const double sr = mua / mu;
      sigma_mu = sr * sr * sr * mu;
      second_order = second_order_factor(ap, ad, plain);
      // corrector
      double el5[5], eu5[5], h[6], dc[6];
      wave_lds_sync();
      UNROLL for (int i = 0; i < 5; i++) {
        const double dsa = ga5[i] + rpl5[i], dua = -ga5[i] - rpu5[i];
        const double rcl = __builtin_fma(second_order, (ll5[i] * dsa) * (1.0 + dsa * isl5[i]), __builtin_fma(sl5[i], ll5[i], -sigma_mu));
        const double rcu = __builtin_fma(second_order, (lu5[i] * dua) * (1.0 + dua * isu5[i]), __builtin_fma(su5[i], lu5[i], -sigma_mu));
        el5[i] = rcl * isl5[i]; eu5[i] = rcu * isu5[i];
        lds[L_XCH + i][lane] = (el5[i] - eu5[i]) + ((ll5[i] * isl5[i]) * rpl5[i] + (lu5[i] * isu5[i]) * rpu5[i]);
      }
      wave_lds_sync();
      UNROLL for (int i = 0; i < 6; i++) h[i] = gc[i];
      FOR_ROWS(r)
        row_scatter<r>(XR(0, r), t, h);
      END_ROWS
      solve_dc(h, dX, dc);
      // step to the boundary
      double gd5[5], dsl5[5], dsu5[5], dll5[5], dlu5[5], pr = 0.0, dr = 0.0;
      slot_vals(dc, gd5);
      UNROLL for (int i = 0; i < 5; i++) {
        dsl5[i] = gd5[i] + rpl5[i]; dsu5[i] = -gd5[i] - rpu5[i];
        dll5[i] = -el5[i] - (ll5[i] * isl5[i]) * dsl5[i]; dlu5[i] = -eu5[i] - (lu5[i] * isu5[i]) * dsu5[i];
        pr = fmax(pr, fmax(-dsl5[i] * isl5[i], -dsu5[i] * isu5[i]));
        dr = fmax(dr, fmax(-dll5[i] * rcp_fast(ll5[i]), -dlu5[i] * rcp_fast(lu5[i])));
      }
      wave_lds_sync();
      lds[L_XCH + 0][lane] = pr; lds[L_XCH + 1][lane] = dr;
      wave_lds_sync();
      pr = fmax(fmax(lds[L_XCH + 0][Lj[0]], lds[L_XCH + 0][Lj[1]]), lds[L_XCH + 0][Lj[2]]);
      dr = fmax(fmax(lds[L_XCH + 1][Lj[0]], lds[L_XCH + 1][Lj[1]]), lds[L_XCH + 1][Lj[2]]);
      const Red4 rb = reduce4<MULTI, 0, 1, 1, 1>(lds + L_RED, wgs, lane, wv, gbase, k, S, 0.0, pr, dr, 0.0);
      const double m_ = fmax(rb.b, rb.c);
      const double tau = (m_ * a.tau_thr <= 1.0 && eit - it0 < a.tau_iters) ? a.tau : fmin(a.tau, 0.995);
      const double alpha = fmin(1.0, tau / fmax(m_, tau));
      if (!done) tiny_steps = alpha < BTRAPZ_TINY_STEP ? (tiny_steps < 3 ? tiny_steps + 1 : 3) : 0;
      if (!done && alpha == alpha) {
        UNROLL for (int i = 0; i < 3; i++) X[i] += alpha * dX[i];
        UNROLL for (int i = 0; i < 5; i++) {
          sl5[i] += alpha * dsl5[i]; su5[i] += alpha * dsu5[i];
          ll5[i] += alpha * dll5[i]; lu5[i] += alpha * dlu5[i];
        }
      }
    } else {
    {
      // predictor.  rc = s*lambda  ->  tv = lambda_l (s_l + rp_l)/s_l - lambda_u (s_u - rp_u)/s_u
      backward_u(up, dX, dca);   // right-hand side from the Newton-matrix loop, forward sweep done in the factorisation loop
      // With rc = s*lambda:  dlambda/lambda = -(1 + ds/s).  Step lengths come from q = ds/s alone, and
      //   m*mu_aff = (1-ad) S0 + (ap - ad - ap*ad) S1 - ap*ad S4,   S0 = sum s*lambda (= m*mu),
      //   S1 = sum lambda*ds,  S4 = sum lambda*ds*q        -- no second pass over the rows.
      double qmin = 0.0, qmax = -1.0, S1 = 0.0, S4 = 0.0;
      PHASE_FENCE(opaque6(c); opaque6(dca));
      ROW_PREFETCH();
      FOR_ROWS(r)
        ROW_BASE(r)
        const double gd = ROW_STEP(r, row_dot<r>(dca, t), (ll - lu) + (ll * isl) * rpl + (lu * isu) * rpu);
        const double dsl = gd + rpl, dsu = -gd - rpu;
        const double ql = dsl * isl, qu = dsu * isu;
        qmin = fmin(qmin, fmin(ql, qu)); qmax = fmax(qmax, fmax(ql, qu));
        const double al = ll * dsl, au = lu * dsu;
        S1 += al + au;
        S4 += al * ql + au * qu;
      END_ROWS
      const Red4 ra = reduce4<MULTI, 0, 0, 1, 2>(lds + L_RED, wgs, lane, wv, gbase, k, S, S1, S4, qmax, qmin);
      const double ap = 1.0 / fmax(-ra.d, 1.0), ad = 1.0 / fmax(1.0 + ra.c, 1.0);
      const double mua = ((1.0 - ad) * rr.a + (ap - ad - ap * ad) * ra.a - ap * ad * ra.b) * inv_m;
      const double sr = mua / mu;
      sigma_mu = sr * sr * sr * mu;
      // Mehrotra's second-order term ds_aff * dlambda_aff describes the affine step; where that step is blocked at
      // less than a tenth of its length it describes nothing, and late in a solve the method can settle into a
      // two-cycle with it (blocked predictor / long corrector step, mu going 5e-4 <-> 1.5e-3 while the residuals reach
      // 1e-11): 2 of the 262 144 candidates of the four bench batches ended that way where the oracle finds x*.
      // A solve that stalls that way (termination test above) gets a second chance in which the corrector leaves the
      // term out on blocked iterations (the factor of the fused multiply-add that subtracts it): both converge within
      // ten more iterations.  (Dropping it on every blocked step costs 0.5-2.5 % more iterations on average; weighting
      // it by how far the affine step gets -- second_order_factor, round 4 -- saves 5-9 %.)
      second_order = second_order_factor(ap, ad, plain);
    }
    {
      // corrector.  rc = s*lambda + [ds_aff*dlambda_aff] - sigma*mu , dlambda_aff = -lambda (1 + ds_aff/s)
#define ROW_CORR(r)                                                                               \
      const double ga = ROW_STEP(r, row_dot<r>(dca, t), (ll - lu) + (ll * isl) * rpl + (lu * isu) * rpu); \
      const double dsa = ga + rpl, dua = -ga - rpu;                                                 \
      const double rcl = __builtin_fma(second_order, (ll * dsa) * (1.0 + dsa * isl), __builtin_fma(sl[SI(r)], ll, -sigma_mu)); \
      const double rcu = __builtin_fma(second_order, (lu * dua) * (1.0 + dua * isu), __builtin_fma(su[SI(r)], lu, -sigma_mu)); \
      const double el = rcl * isl, eu = rcu * isu, wl = ll * isl, wu = lu * isu;
      double h[6], dc[6];
          double el_[NR], eu_[NR];   // rc/s of the corrected complementarity targets, reused by the two loops below
      UNROLL for (int i = 0; i < 6; i++) h[i] = gc[i];
      PHASE_FENCE(opaque6(c); opaque6(dca));
      ROW_PREFETCH();
      FOR_ROWS(r)
        ROW_BASE(r)
        ROW_CORR(r)
        el_[SI(r)] = el; eu_[SI(r)] = eu;
        if constexpr (ELASTIC) row_scatter<r>(((el - eu) + (wl * rpl + wu * rpu)) * rcp(1.0 + ED(r) * (wl + wu)), t, h);
        else row_scatter<r>((el - eu) + (wl * rpl + wu * rpu), t, h);
      END_ROWS
      solve_dc(h, dX, dc);
      // step to the boundary: ratios -ds/s and -dlambda/lambda (seed reciprocal is enough here)
      double pr = 0.0, dr = 0.0;
      PHASE_FENCE(opaque6(c); opaque6(dc));
      ROW_PREFETCH();
      FOR_ROWS(r)
        ROW_BASE(r)
        const double gd = ROW_STEP(r, row_dot<r>(dc, t), (el_[SI(r)] - eu_[SI(r)]) + (ll * isl) * rpl + (lu * isu) * rpu);
        const double dsl = gd + rpl, dsu = -gd - rpu;
        const double dll = -el_[SI(r)] - (ll * isl) * dsl, dlu = -eu_[SI(r)] - (lu * isu) * dsu;
        pr = fmax(pr, fmax(-dsl * isl, -dsu * isu));
        dr = fmax(dr, fmax(-dll * rcp_fast(ll), -dlu * rcp_fast(lu)));
      END_ROWS
      const Red4 ra = reduce4<MULTI, 0, 1, 1, 1>(lds + L_RED, wgs, lane, wv, gbase, k, S, 0.0, pr, dr, 0.0);
      // m = largest ratio -ds/s, -dlambda/lambda: the boundary is 1/m away.  A long step may go almost all the way
      // (fewer iterations); a blocked one keeps 0.5 % distance, or the iterates lose centrality and crawl.
      const double m_ = fmax(ra.b, ra.c);
      const double tau = (m_ * a.tau_thr <= 1.0 && eit - it0 < a.tau_iters) ? a.tau : fmin(a.tau, 0.995);
      const double alpha = fmin(1.0, tau / fmax(m_, tau));
      if (!done) tiny_steps = alpha < BTRAPZ_TINY_STEP ? (tiny_steps < 3 ? tiny_steps + 1 : 3) : 0;
      // a finished group keeps its state (a branch, not alpha = 0: 0 * inf would poison it); a step that is
      // not finite is not taken either -- the score of the unchanged iterate then stalls and the group stops
      if (!done && alpha == alpha) {
        UNROLL for (int i = 0; i < 3; i++) X[i] += alpha * dX[i];
        PHASE_FENCE(opaque6(c); opaque6(dc));
        ROW_PREFETCH();
        FOR_ROWS(r)
          ROW_BASE(r)
          const double gd = ROW_STEP(r, row_dot<r>(dc, t), (el_[SI(r)] - eu_[SI(r)]) + (ll * isl) * rpl + (lu * isu) * rpu);
          const double dsl = gd + rpl, dsu = -gd - rpu;
          sl[SI(r)] += alpha * dsl; su[SI(r)] += alpha * dsu;
          LL(r) = ll + alpha * (-el_[SI(r)] - (ll * isl) * dsl); LU(r) = lu + alpha * (-eu_[SI(r)] - (lu * isu) * dsu);
        END_ROWS
      }
#undef ROW_CORR
    }
    }   // (!SPLIT)
#undef ROW_BASE
#undef ROW_STEP
#undef ROW_PREFETCH
    if (!done) ++eit;
  }
  if constexpr (!QUEUE) write_back();

}

// Four instantiations: {cold, warm start} x {candidates in memory order, candidates through a.order (ragged batches
// and scheduling hints)}.  The bench path is the first; keeping the others out of it keeps its register allocation.
__global__ __launch_bounds__(64) void ipm_solve_kernel(const KernelArgs a, const double *__restrict__ mqm) {
  __shared__ double lds[lds_rows<false>()][64];
  ipm_solve_body<false, false>(a, mqm, lds, (int)blockIdx.x, (int)threadIdx.x);
}
__global__ __launch_bounds__(64) void ipm_solve_ordered_kernel(const KernelArgs a, const double *__restrict__ mqm) {
  __shared__ double lds[lds_rows<false>()][64];
  ipm_solve_body<false, true>(a, mqm, lds, (int)blockIdx.x, (int)threadIdx.x);
}
__global__ __launch_bounds__(64) void ipm_solve_warm_kernel(const KernelArgs a, const double *__restrict__ mqm) {
  __shared__ double lds[lds_rows<false>()][64];
  ipm_solve_body<true, false>(a, mqm, lds, (int)blockIdx.x, (int)threadIdx.x);
}
__global__ __launch_bounds__(64) void ipm_solve_warm_ordered_kernel(const KernelArgs a, const double *__restrict__ mqm) {
  __shared__ double lds[lds_rows<false>()][64];
  ipm_solve_body<true, true>(a, mqm, lds, (int)blockIdx.x, (int)threadIdx.x);
}
// Uniform cold batches much larger than the machine, two launches (CAPPED / RESUME above): every group stops at
// cap_iter iterations; the unfinished ones are carried on by the resume launch, like with like, far ones first.
__global__ __launch_bounds__(64) void ipm_solve_capped_kernel(const KernelArgs a, const double *__restrict__ mqm) {
  __shared__ double lds[lds_rows<false>()][64];
  ipm_solve_body<false, false, false, false, false, false, true, false>(a, mqm, lds, (int)blockIdx.x, (int)threadIdx.x);
}
// ... the first launch of a ragged batch (candidates bucketed by segment count, as in ipm_solve_ordered_kernel)
__global__ __launch_bounds__(64) void ipm_solve_capped_ordered_kernel(const KernelArgs a, const double *__restrict__ mqm) {
  __shared__ double lds[lds_rows<false>()][64];
  ipm_solve_body<false, true, false, false, false, false, true, false>(a, mqm, lds, (int)blockIdx.x, (int)threadIdx.x);
}
__global__ __launch_bounds__(64) void ipm_solve_resume_kernel(const KernelArgs a, const double *__restrict__ mqm) {
  __shared__ double lds[lds_rows<false>()][64];
  ipm_solve_body<false, true, false, false, false, false, false, true>(a, mqm, lds, (int)blockIdx.x, (int)threadIdx.x);
}
#ifdef BTRAPZ_EXPERIMENTS
// Uniform cold batches much larger than the machine: persistent wavefronts over a candidate queue (see QUEUE above).
// A measured loss against the two-launch solve on the bench batches: experiment builds only.
__global__ __launch_bounds__(64) void ipm_solve_queue_kernel(const KernelArgs a, const double *__restrict__ mqm) {
  __shared__ double lds[lds_rows<false>()][64];
  ipm_solve_body<false, false, false, true>(a, mqm, lds, (int)blockIdx.x, (int)threadIdx.x);
}
#endif
// Few candidates (fewer axis problems than SIMDs), at most 21 segments: one candidate per wavefront, its rows split over
// three lane groups (SPLIT above).  Wavefront w: axis w & 1 of candidate w >> 1.
__global__ __launch_bounds__(64) void ipm_solve_split_kernel(const KernelArgs a, const double *__restrict__ mqm) {
  __shared__ double lds[21][64];
  ipm_solve_body<false, false, false, false, true>(a, mqm, lds, (int)blockIdx.x, (int)threadIdx.x);
}
// More than 64 segments (up to 256): one axis problem per workgroup of ceil(S / 64) wavefronts (MULTI above).
// Workgroup w: axis w & 1 of candidate w >> 1.
__global__ __launch_bounds__(256) void ipm_solve_long_kernel(const KernelArgs a, const double *__restrict__ mqm) {
  __shared__ double lds[4][lds_rows<false>()][64];
  __shared__ double wgs[WGS_SEAM + 4 * WGS_LANES];
  const int wv = (int)threadIdx.x >> 6;
  ipm_solve_body<false, false, false, false, false, true>(a, mqm, lds[wv], (int)blockIdx.x, (int)threadIdx.x & 63, wgs, wv);
}
// ... and its rescue pass (btrapz_options.elastic): the workgroups of the axis problems that stalled solve them again with
// elastic rows, the others leave at once.  Up to three wavefronts (192 segments): the 18-row LDS columns of a fourth do
// not fit the CU.
__global__ __launch_bounds__(192) void ipm_solve_long_elastic_kernel(const KernelArgs a, const double *__restrict__ mqm) {
  __shared__ double lds[3][lds_rows<true>()][64];
  __shared__ double wgs[WGS_SEAM + 4 * WGS_LANES];
  if (a.axis_status[blockIdx.x] != BTRAPZ_MAX_ITER_REACHED) return;   // (the same for every thread of the workgroup)
  const int wv = (int)threadIdx.x >> 6;
  ipm_solve_body<false, false, true, false, false, true>(a, mqm, lds[wv], (int)blockIdx.x, (int)threadIdx.x & 63, wgs, wv);
}
// Rescue pass (btrapz_options.elastic): the stalled axis problems, listed per axis, with elastic rows.
__global__ __launch_bounds__(64) void ipm_solve_elastic_kernel(const KernelArgs a, const double *__restrict__ mqm) {
  __shared__ double lds[lds_rows<true>()][64];
  ipm_solve_body<false, true, true>(a, mqm, lds, (int)blockIdx.x, (int)threadIdx.x);
}
// ---- candidates that cannot start (btrapz_options.compact) ------------------------------------------------------------
// A quarter of the cuboid bench batch and of the knot-level pipeline's candidates end BEFORE the first iteration: an
// empty inscribed interval (cuboid_3d.cc:677-689), an initial state outside segment 0's rows, a joint whose two sides
// share no value -- what the set-up of the solve kernels checks.  Left in the launch they cost twice: their groups idle
// in wavefronts that wait for the live ones, and the OTHER axis of such a candidate is solved in full although the
// candidate is lost (cuboid: the s axis is dead, the l axis runs its eight iterations).  This pre-pass -- one lane per
// segment slot, both axes -- lists the live candidates (keys for the bucket kernels: 0 = dead) and
// writes the dead ones' records itself, so that the solve launch holds live groups only.
// It applies the kernels' own tests with DOUBLED tolerances (a row with l > u by more than 2e-12 relative to nothing, a
// joint empty by more than 2e-9, an initial state outside by more than 2e-7): a candidate it passes as live that the
// kernel then finds dead is handled there as before, and one it calls dead is dead in the kernel by a clear margin --
// the two can never disagree about a solvable candidate, whatever the compiler contracts into fused multiply-adds.
// Status: BTRAPZ_PRIMAL_INFEASIBLE / BTRAPZ_MAX_ITER_REACHED for the axis that is dead, exactly what its solve would
// report; 0 for an axis that was not looked at further (the finalize kernel takes the most severe code).
__global__ __launch_bounds__(64) void prestart_kernel(const KernelArgs a, int S_uniform, const int *seg_count, int *keys) {
  // lane = (candidate of the wavefront, slot): floor(64 / seg_stride) candidates per wavefront, every field load 8 contiguous
  // bytes per lane as in the solve kernels; the joint's other side comes from the next lane, the verdict of a candidate's
  // lanes through a ballot
  const int lane = threadIdx.x, stride = a.seg_stride, gpw = 64 / stride;
  const int g = lane / stride, k = lane - g * stride;
  const long long cand = (long long)blockIdx.x * gpw + g;
  const bool in_wave = g < gpw && cand < a.B;
  const int b = in_wave ? (int)cand : a.B - 1;
  int S = S_uniform;
  bool usable = true;
  if (seg_count) { S = seg_count[b]; usable = S >= 1 && S <= 64 && S <= stride; }
  const bool act = in_wave && usable && k < S;
  const bool first = k == 0, last = k == S - 1;
  const size_t BS = (size_t)a.B * stride;
  const double *sg = a.seg;
  const Shared &sh = a.sh;
  const size_t e_ = (size_t)b * stride + (act ? k : 0);
  const double t = sg[BTRAPZ_F_T * BS + e_];
  int st_axis[2] = {0, 0};
  const unsigned long long gmask = (stride >= 64 ? ~0ull : ((1ull << stride) - 1ull)) << (g < gpw ? g * stride : 0);
  for (int axis = 0; axis < 2; axis++) {
    double lb, ls, ub, us, vlo[5], vhi[5];
    if (axis == 0) {
      lb = sg[BTRAPZ_F_DOWN_BIAS * BS + e_]; ls = sg[BTRAPZ_F_DOWN_SKEW * BS + e_];
      ub = sg[BTRAPZ_F_UPP_BIAS * BS + e_];  us = sg[BTRAPZ_F_UPP_SKEW * BS + e_];
      const double lo = sg[BTRAPZ_F_DS_LO * BS + e_], hi = sg[BTRAPZ_F_DS_HI * BS + e_];
      UNROLL for (int i = 0; i < 5; i++) { vlo[i] = lo; vhi[i] = hi; }
    } else {
      lb = sg[BTRAPZ_F_L_DOWN_BIAS * BS + e_]; ls = sg[BTRAPZ_F_L_DOWN_SKEW * BS + e_];
      ub = sg[BTRAPZ_F_L_UPP_BIAS * BS + e_];  us = sg[BTRAPZ_F_L_UPP_SKEW * BS + e_];
      UNROLL for (int i = 0; i < 5; i++) { vlo[i] = a.dl_bounds[(size_t)b * 10 + 2 * i]; vhi[i] = a.dl_bounds[(size_t)b * 10 + 2 * i + 1]; }
    }
    double plo0 = lb, dplo = ls * 0.2 * t, phi0 = ub, dphi = us * 0.2 * t;
    if (sh.variant == BTRAPZ_CUBOID) {
      if (axis == 0) {
        plo0 = fmax(0.0, fmax(ls * 0.0 + lb, lb + ls * t));
        phi0 = fmin(100.0, fmin(us * 0.0 + ub, ub + us * t));
      } else {
        plo0 = sg[BTRAPZ_F_BEG_L * BS + e_]; phi0 = sg[BTRAPZ_F_END_L * BS + e_];
      }
      dplo = 0.0; dphi = 0.0;
    }
    (void)move_far_bounds(plo0, dplo, phi0, dphi, vlo, vhi);
    const double alo = (axis == 0 ? sh.acc_s[0] : sh.acc_l[0]) * t, ahi = (axis == 0 ? sh.acc_s[1] : sh.acc_l[1]) * t;
    const double jlo = (axis == 0 ? sh.jerk_s[0] : sh.jerk_l[0]) * t * t, jhi = (axis == 0 ? sh.jerk_s[1] : sh.jerk_l[1]) * t * t;
    // rows with l > u (all 18 of the reference): the kernels allow 1e-12, this pass 2e-12 and one part in 1e12
    double gapmin = 1e300;
    UNROLL for (int r = 0; r < 6; r++) {
      const double lo = plo0 + (double)r * dplo, up = phi0 + (double)r * dphi;
      gapmin = fmin(gapmin, (up - lo) + 2e-12 + 1e-12 * fmax(fabs(lo), fabs(up)));
    }
    UNROLL for (int r = 0; r < 5; r++) gapmin = fmin(gapmin, (vhi[r] - vlo[r]) + 2e-12 + 1e-12 * fmax(fabs(vlo[r]), fabs(vhi[r])));
    gapmin = fmin(gapmin, (ahi - alo) + 2e-12 + 1e-12 * fmax(fabs(alo), fabs(ahi)));
    gapmin = fmin(gapmin, (jhi - jlo) + 2e-12 + 1e-12 * fmax(fabs(jlo), fabs(jhi)));
    bool dead3 = gapmin < 0.0 || (first && !(t > 0.0));   // (the record's first lane decides on t: see the write-back of the kernels)
    bool dead2 = false;
    // the joint at the end of this segment: its last rows against the next segment's first ones
    const double mplo = plo0 + 5.0 * dplo, mphi = phi0 + 5.0 * dphi, mvlo = vlo[4], mvhi = vhi[4];
    {
      const double nplo = dpp_next(plo0), nphi = dpp_next(phi0), nvlo = dpp_next(vlo[0]), nvhi = dpp_next(vhi[0]);
      if (!last) {
        const bool own_ok = mplo <= mphi && mvlo <= mvhi && nplo <= nphi && nvlo <= nvhi;
        const double jl = fmax(mplo, nplo), jh = fmin(mphi, nphi), jvl = fmax(mvlo, nvlo), jvh = fmin(mvhi, nvhi);
        const double ptol = 2e-9 * (1.0 + fmax(fabs(jl), fabs(jh))), vtol = 2e-9 * (1.0 + fmax(fabs(jvl), fabs(jvh)));
        dead2 = own_ok && (jl > jh + ptol || jvl > jvh + vtol);
      }
    }
    if (first && t > 0.0) {   // segment 0's first rows state the given initial state
      const double it = 1.0 / t, t20 = t * 0.05;
      const double X0 = a.init[(size_t)b * 6 + axis * 3], X1 = a.init[(size_t)b * 6 + axis * 3 + 1], X2 = a.init[(size_t)b * 6 + axis * 3 + 2];
      const double c0 = it * X0, c1 = c0 + 0.2 * X1, c2 = c0 + 0.4 * X1 + t20 * X2;
      auto outside = [&](double g_, double lo, double hi) {
        const double tol = 2e-7 * (1.0 + fmax(fabs(lo), fabs(hi)));
        return g_ < lo - tol || g_ > hi + tol;   // (a value that is not finite is left to the kernel)
      };
      if (outside(t * c0, plo0, phi0) || outside(5.0 * (c1 - c0), vlo[0], vhi[0]) || outside(20.0 * ((c0 - 2.0 * c1) + c2), alo, ahi)) dead2 = true;
    }
    const bool any3 = (__ballot(act && dead3) & gmask) != 0, any2 = (__ballot(act && dead2) & gmask) != 0;
    st_axis[axis] = any3 ? BTRAPZ_PRIMAL_INFEASIBLE : any2 ? BTRAPZ_MAX_ITER_REACHED : 0;
  }
  // A dropped candidate never reaches a solve kernel: its control points are written HERE, as NaN -- a caller that reads
  // every row (eval_states feeding x0 of a later warm start, say) meets a value the warm start refuses instead of whatever
  // the buffer held.  12 doubles per lane of the candidate's slot of 12 * seg_stride.
  if (in_wave && a.ctrl && (!usable || st_axis[0] != 0 || st_axis[1] != 0)) {
    double *dst = a.ctrl + ((size_t)b * stride + k) * 12;
    UNROLL for (int i = 0; i < 12; i++) dst[i] = __longlong_as_double(0x7ff8000000000000LL);
  }
  if (!in_wave || !first) return;
  if (!usable) {   // no usable corridor (what bucket_scatter_kernel writes for these)
    keys[b] = 0;
    a.axis_obj[2 * b] = 0.0; a.axis_obj[2 * b + 1] = 0.0;
    a.axis_status[2 * b] = BTRAPZ_NO_CORRIDOR; a.axis_status[2 * b + 1] = BTRAPZ_NO_CORRIDOR;
    a.axis_iters[2 * b] = 0; a.axis_iters[2 * b + 1] = 0;
    return;
  }
  const bool dead = st_axis[0] != 0 || st_axis[1] != 0;
  keys[b] = dead ? 0 : (seg_count ? S : 1);
  if (dead) {
    UNROLL for (int axis = 0; axis < 2; axis++) {
      a.axis_obj[2 * b + axis] = 0.0; a.axis_status[2 * b + axis] = st_axis[axis]; a.axis_iters[2 * b + axis] = 0;
    }
  }
}

// keys of the rescue lists: key[axis][b] = segment count of candidate b when that axis problem stalled, else 0
__global__ void rescue_keys_kernel(int B, int S, const int *seg_count, const int *axis_status, int *keys, int all) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int s = seg_count ? seg_count[b] : S;
  const bool usable = s >= 1 && s <= 64 && s <= S;
  keys[b] = usable && (all || axis_status[2 * b] == BTRAPZ_MAX_ITER_REACHED) ? s : 0;
  keys[B + b] = usable && (all || axis_status[2 * b + 1] == BTRAPZ_MAX_ITER_REACHED) ? s : 0;
}
// elastic solve of every candidate without a first attempt: per-axis records start empty (iters -1: the rescue kernel
// adds its own count + 1), candidates without a usable corridor are marked as the bucket kernel of a ragged solve does
__global__ void rescue_init_kernel(int B, int S, const int *seg_count, double *axis_obj, int *axis_status, int *axis_iters) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int s = seg_count ? seg_count[b] : S;
  const bool usable = s >= 1 && s <= 64 && s <= S;
  for (int ax = 0; ax < 2; ax++) {
    axis_obj[2 * b + ax] = 0.0; axis_status[2 * b + ax] = usable ? BTRAPZ_MAX_ITER_REACHED : BTRAPZ_NO_CORRIDOR;
    axis_iters[2 * b + ax] = usable ? -1 : 0;
  }
}

// ---- batch-invariant table M' pQp_d M (solve_3d.cc:87-143), built in stream order whenever the weights change --------
// thread = (axis, derivative d, packed upper-triangle entry): out[axis][d][SYM(i, j)].  pQp_d(a, b) = w_d * prod_{r<d}
// (a - r)(b - r) / (a + b - 2d + 1) for a, b >= d (:87-113); M = Bernstein -> monomial (:122-127).
__global__ void mqm_table_kernel(MqmWeights w, double *out) {
#pragma clang fp contract(off)   // same expressions, same rounding as btrapz_mqm_table_host (find_traj's table)
  const int id = threadIdx.x;
  if (id >= 168) return;
  const int axis = id / 84, d = (id % 84) / 21, e = id % 21;
  int j = 0;
  while ((j + 1) * (j + 2) / 2 <= e) ++j;
  const int i = e - j * (j + 1) / 2;
  const double M[6][6] = {{1, 0, 0, 0, 0, 0},      {-5, 5, 0, 0, 0, 0},      {10, -20, 10, 0, 0, 0},
                          {-10, 30, -30, 10, 0, 0}, {5, -20, 30, -20, 5, 0}, {-1, 5, -10, 10, -5, 1}};
  const double wd = w.w[axis][d];
  double acc = 0.0;
  for (int a = d; a < 6; a++) {
    double t = 0.0;   // (M' pQp)(i, b) summed against M(b, j)
    for (int b = d; b < 6; b++) {
      double num = wd;
      for (int r = 0; r < d; r++) num *= (double)((a - r) * (b - r));
      t += num / (double)(a + b - 2 * d + 1) * M[b][j];
    }
    acc += M[a][i] * t;
  }
  out[id] = acc;
}

// viol[b][c] = the larger of the two axes' class-c violation (btrapz_rescue_violations_device)
__global__ void rescue_violations_kernel(int B, const double *axis_viol, double *viol) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 4 * B) return;
  const int b = i >> 2, c = i & 3;
  viol[i] = fmax(axis_viol[(size_t)(2 * b) * 4 + c], axis_viol[(size_t)(2 * b + 1) * 4 + c]);
}

__global__ void finalize_kernel(int B, const double *axis_obj, const int *axis_status, const int *axis_iters,
                                double *cost, int *status, int *iters) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int s0 = axis_status[2 * b], s1 = axis_status[2 * b + 1];
  int st;
  if (s0 == BTRAPZ_SOLVED && s1 == BTRAPZ_SOLVED) st = BTRAPZ_SOLVED;
  else if (s0 > 0 && s1 > 0) st = BTRAPZ_SOLVED_INACCURATE;
  else st = s0 < s1 ? s0 : s1;  // most severe failure code
  const double c = axis_obj[2 * b] + axis_obj[2 * b + 1];
  status[b] = st;
  cost[b] = (st > 0 && c == c) ? c : __builtin_huge_val();
  if (iters) iters[b] = axis_iters[2 * b] > axis_iters[2 * b + 1] ? axis_iters[2 * b] : axis_iters[2 * b + 1];
}

// ---- arg-min over contiguous groups, ties -> lowest index ----------------------------------------------
// One block per (group, chunk of the group): a group of 65 536 candidates read by ONE block is latency-bound (68 us);
// split over 64 blocks and finished by argmin_final_kernel it takes a few.  The result does not depend on the split:
// the comparison (cost, then lowest index) is a total order.
__device__ __forceinline__ bool argmin_better(double c2, long long i2, double c1, long long i1) {
  return c2 < c1 || (c2 == c1 && i2 >= 0 && (i1 < 0 || i2 < i1));
}
__device__ __forceinline__ void argmin_block_reduce(double &bc, long long &bi, double *sc, long long *si) {
  sc[threadIdx.x] = bc; si[threadIdx.x] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s && argmin_better(sc[threadIdx.x + s], si[threadIdx.x + s], sc[threadIdx.x], si[threadIdx.x])) {
      sc[threadIdx.x] = sc[threadIdx.x + s]; si[threadIdx.x] = si[threadIdx.x + s];
    }
    __syncthreads();
  }
  bc = sc[0]; bi = si[0];
}
// grid = (groups, chunks) -- groups in x: there may be millions of them, chunks are at most 256; chunks == 1: writes
// the result; else partial results part_cost / part_idx [groups][chunks]
__global__ __launch_bounds__(256) void argmin_kernel(int group, long long index_base, const double *cost,
                                                      long long *best_idx, double *best_cost, double *part_cost,
                                                      long long *part_idx) {
  __shared__ double sc[256];
  __shared__ long long si[256];
  const long long g = blockIdx.x;
  const int chunks = gridDim.y, chunk = blockIdx.y, per = (group + chunks - 1) / chunks;
  const int lo = chunk * per, hi = lo + per < group ? lo + per : group;
  double bc = __builtin_huge_val();
  long long bi = -1;
  for (int j = lo + threadIdx.x; j < hi; j += blockDim.x) {
    const double c = cost[g * group + j];
    if (c < bc) { bc = c; bi = g * group + j; }
  }
  argmin_block_reduce(bc, bi, sc, si);
  if (threadIdx.x == 0) {
    if (chunks == 1) { best_idx[g] = bi >= 0 ? bi + index_base : -1; best_cost[g] = bc; }
    else { part_cost[g * chunks + chunk] = bc; part_idx[g * chunks + chunk] = bi; }
  }
}
__global__ __launch_bounds__(256) void argmin_final_kernel(int chunks, long long index_base, const double *part_cost,
                                                            const long long *part_idx, long long *best_idx, double *best_cost) {
  __shared__ double sc[256];
  __shared__ long long si[256];
  const long long g = blockIdx.x;
  double bc = __builtin_huge_val();
  long long bi = -1;
  for (int j = threadIdx.x; j < chunks; j += blockDim.x) {
    const double c = part_cost[g * chunks + j]; const long long i = part_idx[g * chunks + j];
    if (argmin_better(c, i, bc, bi)) { bc = c; bi = i; }
  }
  argmin_block_reduce(bc, bi, sc, si);
  if (threadIdx.x == 0) { best_idx[g] = bi >= 0 ? bi + index_base : -1; best_cost[g] = bc; }
}

// ---- the global winner from the ranks' winners (multi-GPU arg-min, after the all-gather) --------------------------
// pairs [world][n][2] int64: (bit pattern of the float64 cost, global candidate index or -1) of every rank's local
// winner of group g.  Lexicographic min over the ranks: cost first (a NaN never wins), then the lowest index.
__global__ void argmin_pairs_kernel(int world, int n, const long long *pairs, double *best_cost, long long *best_idx) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  double bc = __builtin_huge_val();
  long long bi = -1;
  for (int r = 0; r < world; r++) {
    double c = __longlong_as_double(pairs[((size_t)r * n + g) * 2]);
    const long long i = pairs[((size_t)r * n + g) * 2 + 1];
    if (!(c == c)) c = __builtin_huge_val();
    if (argmin_better(c, i, bc, bi)) { bc = c; bi = i; }
  }
  best_cost[g] = bc; best_idx[g] = bi;
}

// ---- state of solved trajectories at arbitrary times (warm start of the next replanning step) ----------
// thread = (candidate, time index).  x[b][axis][j] = (p, v, a) at times[b][j] seconds from the start of the
// candidate's horizon; the Bezier evaluation is the one of solve_3d.cc:1366-1388.  Past the last segment the end
// state is extrapolated at constant velocity (a = 0); before 0 the start of the first segment is used.
__global__ void eval_states_kernel(int B, int seg_stride, const int *seg_count, const double *seg, const double *ctrl,
                                   int n_times, const double *times, double *x) {
  const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long long)B * n_times) return;
  const int b = (int)(id / n_times), j = (int)(id - (long long)b * n_times);
  const int S = seg_count ? seg_count[b] : seg_stride;
  double *xs = x + (((size_t)b * 2 + 0) * n_times + j) * 3, *xl = x + (((size_t)b * 2 + 1) * n_times + j) * 3;
  if (S < 1 || S > seg_stride) {
    UNROLL for (int i = 0; i < 3; i++) { xs[i] = __longlong_as_double(0x7ff8000000000000LL); xl[i] = xs[i]; }
    return;
  }
  const double *tt = seg + (size_t)BTRAPZ_F_T * B * seg_stride + (size_t)b * seg_stride;
  double rem = times[(size_t)b * n_times + j];
  if (!(rem > 0.0)) rem = 0.0;
  int k = 0;
  while (k < S - 1 && rem > tt[k]) { rem -= tt[k]; ++k; }
  const double t = tt[k];
  const double over = rem > t ? rem - t : 0.0;     // beyond the horizon
  const double tau = over > 0.0 ? 1.0 : rem / t, om = 1.0 - tau;
  const double bc0[6] = {1, 5, 10, 10, 5, 1}, bc1[5] = {1, 4, 6, 4, 1}, bc2[4] = {1, 3, 3, 1};
  double pw[6], qw[6];
  pw[0] = 1.0; qw[0] = 1.0;
  UNROLL for (int i = 1; i < 6; i++) { pw[i] = pw[i - 1] * tau; qw[i] = qw[i - 1] * om; }
  UNROLL for (int ax = 0; ax < 2; ax++) {
    const double *c = ctrl + (size_t)b * 12 * seg_stride + (size_t)ax * 6 * S + (size_t)k * 6;
    double p = 0, v = 0, acc = 0;
    UNROLL for (int i = 0; i < 6; i++) p += c[i] * bc0[i] * pw[i] * qw[5 - i];
    UNROLL for (int i = 0; i < 5; i++) v += 5.0 * (c[i + 1] - c[i]) * bc1[i] * pw[i] * qw[4 - i];
    UNROLL for (int i = 0; i < 4; i++) acc += 20.0 * (c[i + 2] - 2.0 * c[i + 1] + c[i]) * bc2[i] * pw[i] * qw[3 - i];
    double *o = ax == 0 ? xs : xl;
    o[0] = p * t + v * over; o[1] = v; o[2] = over > 0.0 ? 0.0 : acc / t;
  }
}

// ---- Bernstein sampling of selected candidates (solve_3d.cc:1279-1392) ---------------------
// Samples of candidate b, written by threads tid, tid + nthreads, ...; returns the sample count.
// tt: the candidate's S segment durations (the single-candidate launch passes a copy in LDS: its seg lives in host
// memory mapped into the device, and the loops below would cross PCIe once per segment -- 40 us of a 170 us call).
__device__ __forceinline__ int sample_candidate(int seg_stride, int S, double delta, const double *tt,
                                                const double *init, const double *ctrl, long long b, int max_points,
                                                double *o, int tid, int nthreads) {
  // num_of_points_: int accumulated with += double (solve_3d.cc:1279-1282)
  int np = 1;
  for (int k = 0; k < S; k++) np = (int)((double)np + tt[k] / delta);
  if (tid == 0 && max_points > 0) {
    UNROLL for (int a = 0; a < 6; a++) o[(size_t)a * max_points] = init[b * 6 + a];
  }
  const double bc0[6] = {1, 5, 10, 10, 5, 1}, bc1[5] = {1, 4, 6, 4, 1}, bc2[4] = {1, 3, 3, 1};
  // one sample per thread and pass: sample idx of the trajectory lies in the segment k whose samples start at `base`
  // (the reference's loop over segments and their samples, flattened: a block that walks the segments one after the
  // other has a tenth of its threads at work and pays one memory round trip per segment)
  int total = 0;
  for (int k = 0; k < S; k++) total += (int)(tt[k] / delta);
  for (int idx = tid; idx < total; idx += nthreads) {
    int k = 0, base = 1, linter = (int)(tt[0] / delta);  // :1351
    while (idx >= base - 1 + linter) { base += linter; ++k; linter = (int)(tt[k] / delta); }
    const double t = tt[k];
    const int l = idx - (base - 1) + 1;
    const int vi = base + l - 1;
    if (vi >= max_points) continue;
    const double tau = (double)l / (double)linter, om = 1.0 - tau;
    double pw[6], qw[6];
    pw[0] = 1.0; qw[0] = 1.0;
    UNROLL for (int i = 1; i < 6; i++) { pw[i] = pw[i - 1] * tau; qw[i] = qw[i - 1] * om; }
    UNROLL for (int ax = 0; ax < 2; ax++) {
      const double *c = ctrl + (size_t)b * 12 * seg_stride + (size_t)ax * 6 * S + (size_t)k * 6;
      double x = 0, dx = 0, ddx = 0;
      UNROLL for (int i = 0; i < 6; i++) x += c[i] * bc0[i] * pw[i] * qw[5 - i];
      UNROLL for (int i = 0; i < 5; i++) dx += 5.0 * (c[i + 1] - c[i]) * bc1[i] * pw[i] * qw[4 - i];
      UNROLL for (int i = 0; i < 4; i++) ddx += 20.0 * (c[i + 2] - 2.0 * c[i + 1] + c[i]) * bc2[i] * pw[i] * qw[3 - i];
      o[(size_t)(3 * ax + 0) * max_points + vi] = x * t;
      o[(size_t)(3 * ax + 1) * max_points + vi] = dx;
      o[(size_t)(3 * ax + 2) * max_points + vi] = ddx / t;
    }
  }
  return np;
}
// one block per selected candidate; thread = sample point.
__global__ void sample_kernel(int B, int seg_stride, const int *seg_count, double delta, const double *seg,
                              const double *init, const double *ctrl, int nsel, const long long *sel, int max_points,
                              double *out, int *npoints) {
  const int j = blockIdx.x;
  if (j >= nsel) return;
  const long long b = sel[j];
  double *o = out + (size_t)j * 6 * max_points;
  if (b < 0 || b >= B) { if (threadIdx.x == 0) npoints[j] = 0; return; }
  const int S = seg_count ? seg_count[b] : seg_stride;
  if (S < 1 || S > seg_stride) { if (threadIdx.x == 0) npoints[j] = 0; return; }
  const int np = sample_candidate(seg_stride, S, delta, seg + (size_t)BTRAPZ_F_T * B * seg_stride + (size_t)b * seg_stride, init, ctrl, b,
                                  max_points, o, (int)threadIdx.x, (int)blockDim.x);
  if (threadIdx.x == 0) npoints[j] = np;
}

// ---- ONE candidate, ONE launch (find_traj: the reference's call pattern, cart_frenet.py:1567) ---------------------
// Two wavefronts solve the two axes side by side, then the workgroup merges their status, evaluates the acceptance
// test and samples the trajectory: what btrapz_solve_batch_device + finalize_kernel + sample_kernel do in three
// launches.  The inputs (a.seg, a.init, ..., mqm) and out may be host memory mapped into the device -- inputs are read
// once, results written once -- so the call needs no copy either; what the kernel reads back (control points, per-axis
// records) lives in device memory (a.ctrl, a.axis_*).  out: [0] cost, [1] status and iterations (two ints), [2]
// sample count (int) and the completion word (int: set to 1, system scope, when everything else is in place),
// [3 .. 3 + 12 S) control points, then traj [6][max_points].
template <bool WARM, bool SPLIT = false>
__device__ __forceinline__ void single_candidate_body(const KernelArgs &a, const double *__restrict__ mqm, double delta,
                                                      int max_points, double *out) {
  double *res = out, *traj = out + 3 + 12 * a.S;
  __shared__ double lds[2][SPLIT ? 21 : lds_rows<false>()][64];
  __shared__ double tseg[64], init6[6];   // what the sampling reads of the inputs (see sample_candidate)
  if ((int)threadIdx.x < a.S) tseg[threadIdx.x] = a.seg[(size_t)BTRAPZ_F_T * a.seg_stride + threadIdx.x];
  if ((int)threadIdx.x >= 64 && (int)threadIdx.x < 70) init6[threadIdx.x - 64] = a.init[threadIdx.x - 64];
  const int w = (int)threadIdx.x >> 6;
  ipm_solve_body<WARM, false, false, false, SPLIT>(a, mqm, lds[w], w, (int)threadIdx.x & 63);
  __threadfence_block();
  __syncthreads();
  const int s0 = a.axis_status[0], s1 = a.axis_status[1];
  int st;
  if (s0 == BTRAPZ_SOLVED && s1 == BTRAPZ_SOLVED) st = BTRAPZ_SOLVED;
  else if (s0 > 0 && s1 > 0) st = BTRAPZ_SOLVED_INACCURATE;
  else st = s0 < s1 ? s0 : s1;
  const double c = a.axis_obj[0] + a.axis_obj[1];
  int np = 0;
  if (st > 0) np = sample_candidate(a.seg_stride, a.S, delta, tseg, init6, a.ctrl, 0, max_points, traj, (int)threadIdx.x, 128);
  for (int i = (int)threadIdx.x; i < 12 * a.S; i += 128) out[3 + i] = a.ctrl[i];
  if (threadIdx.x == 0) {
    res[0] = (st > 0 && c == c) ? c : __builtin_huge_val();
    int *ri = reinterpret_cast<int *>(res + 1);
    ri[0] = st; ri[1] = a.axis_iters[0] > a.axis_iters[1] ? a.axis_iters[0] : a.axis_iters[1];
    reinterpret_cast<int *>(res + 2)[0] = np;
  }
  // "results complete", for a host that polls the mapped block instead of waiting for the end-of-kernel signal: every
  // thread's result stores are released to the system before one thread sets the word
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(reinterpret_cast<int *>(res + 2) + 1, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ __launch_bounds__(128) void single_candidate_kernel(const KernelArgs a, const double *__restrict__ mqm,
                                                               double delta, int max_points, double *out) {
  single_candidate_body<false>(a, mqm, delta, max_points, out);
}
// ... at most 21 segments: the rows of every segment spread over three lanes (SPLIT in ipm_solve_body)
__global__ __launch_bounds__(128) void single_candidate_split_kernel(const KernelArgs a, const double *__restrict__ mqm,
                                                                     double delta, int max_points, double *out) {
  single_candidate_body<false, true>(a, mqm, delta, max_points, out);
}
__global__ __launch_bounds__(128) void single_candidate_warm_split_kernel(const KernelArgs a, const double *__restrict__ mqm,
                                                                          double delta, int max_points, double *out) {
  single_candidate_body<true, true>(a, mqm, delta, max_points, out);
}
// ... starting from the joint states and multipliers the previous call left on the device (find_traj in a replanning
// loop, BTRAPZ_WARM=1), and leaving its own for the next one
__global__ __launch_bounds__(128) void single_candidate_warm_kernel(const KernelArgs a, const double *__restrict__ mqm,
                                                                    double delta, int max_points, double *out) {
  single_candidate_body<true>(a, mqm, delta, max_points, out);
}

}  // namespace btrapz
