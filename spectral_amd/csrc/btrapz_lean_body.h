// btrapz_lean_body.h -- the packed solve at TWO wavefronts per SIMD (gfx950): the body of the kernels that btrapz_lean.hip
// (cold solves) and btrapz_lean_warm.hip (warm starts) instantiate.
//
// Same problem, same method, same mapping as the packed form of btrapz_kernels.hip (one lane per segment, floor(64 / S)
// axis problems per wavefront, Mehrotra predictor-corrector in the joint states, two-sided block LDL^T; reference call
// sites: FormulateProblem src/solve_3d.cc:1143-1229, CalculateKernel :70-224, CalculateOffset :226-321,
// CalculateAffineConstraint :779-1129 and src/cuboid_3d.cc:632-988, osqp_setup + osqp_solve :1246,1249, acceptance
// :1251-1277) -- on a state diet.  The packed form keeps ~290 doubles per lane alive (256 VGPR + ~200 AGPR, 32 KB of
// LDS): one wavefront per SIMD, nobody hides the dependent chains of the block elimination, and a sixth of its vector
// instructions are v_accvgpr moves.  This form keeps, per lane,
//   registers: the 30 slacks, the joint state, 17 doubles of problem data, the termination bookkeeping (~120 VGPRs);
//   LDS:       the 30 multipliers, the best iterate and the initial state in columns private to the lane, 4 reduction
//              rows: 40 rows x 64 lanes x 8 B = 20 KB per wavefront -> 8 wavefronts per CU;
// and recomputes the rest where it is used: reciprocal slacks (v_rcp_f64 + one Newton step in the passes whose result
// enters the iterate, the bare seed where only step-length statistics come out), row values and residuals from the six
// control points, row bounds from the record fields, the P block once per iteration from the scalar M'QM table (it
// serves the gradient and then becomes the Newton block in place), q from two numbers (the Bernstein moments of a
// linear reference are q_j = B / 6 + A (j + 1) / 42), the corrector's complementarity targets in each of the three
// passes that need them.  The first two row passes of the packed form (residuals / gradient, Newton block / predictor
// right-hand side) are one here.  amdgpu_waves_per_eu(2, 2) holds the allocator to 256 registers, all architectural:
// no AGPR traffic; a second resident wavefront fills the issue slots the dependent steps leave.
// Instantiations: uniform batches in memory order, ragged batches / hint classes through a.order, and the capped first
// launch + resume launch of btrapz_options.cap_iter (hand-over record: SUSP_FIELDS doubles per lane, as in the packed form).
// Not here (the packed form serves them): warm starts, the rescue pass, btrapz_options.start = 1, the candidate queue.
#ifndef BTRAPZ_LEAN_BODY_H
#define BTRAPZ_LEAN_BODY_H
#include <hip/hip_runtime.h>
#include "btrapz_ipm.h"

namespace btrapz {

enum { LN_LL = 0, LN_LU = 15, LN_RED = 30, LN_XB = 34, LN_XI = 37, LN_ROWS = 40 };
enum { LEAN_SUSP_FIELDS = 3 + 4 * 15 + 3 + 8 };   // slot stride: SUSP_FIELDS of the packed form (the host sizes one workspace);
                                                  // the lean record itself is 69 doubles per lane (3 + 3 + 60 + 3 of bookkeeping)

template <bool ORDERED, bool CAPPED, bool RESUME, bool WARM = false>
__device__ __forceinline__ void lean_solve_body(const KernelArgs &a, const double *__restrict__ mqm, double (*lds)[64],
                                                const int wave_id, const int lane) {
  static_assert(!(CAPPED && RESUME), "one launch is the first or the second");
  static_assert(!RESUME || ORDERED, "the resume pass reads its problems from per-axis lists");
  static_assert(!WARM || (!CAPPED && !RESUME), "warm starts: one launch");
  constexpr bool FULL = false;
  constexpr int NR = 15;
  constexpr bool PERAXIS = RESUME;
  // One or two segments (ragged batches may hold such buckets; a uniform batch of fewer than three takes the packed form,
  // btrapz_host.hip): the root of the two-sided elimination is then a group's last lane, and what its right neighbour --
  // another group's first lane -- holds is not the root's to add: one select at the root's step, carried by the
  // instantiations that read their candidates through a.order (ragged batches, hint classes, the pre-pass's lists, resume
  // lists).  With three or more segments the select passes the value through, so a scheduling hint, the pre-pass or a
  // second launch cannot change a result's bits.  (Until round 5 a wave-uniform fix-up inside every step of the sequential
  // loops, +2 % on every solve of an instantiation that carried it, and a template parameter of its own.)
  constexpr bool SMALL_S = ORDERED;
  const int axis = __builtin_amdgcn_readfirstlane(wave_id & 1);
  int S, pair = wave_id >> 1, ncand = a.B, cand0 = 0;
  if constexpr (ORDERED) {
    const int *wave_prefix = a.wave_prefix + (PERAXIS ? axis * 198 : 0), *cand_prefix = a.cand_prefix + (PERAXIS ? axis * 198 : 0);
    if (pair >= wave_prefix[65]) return;
    int s = 1, hi = 65;
    while (hi - s > 1) {
      const int mid = (s + hi) >> 1;
      if (wave_prefix[mid] <= pair) s = mid; else hi = mid;
    }
    S = a.bucket_S ? a.bucket_S : 65 - s;
    pair -= wave_prefix[s]; cand0 = cand_prefix[s] + (PERAXIS ? axis * a.B : 0); ncand = cand_prefix[s + 1] - cand_prefix[s];
  } else {
    S = a.S;
  }
  S = __builtin_amdgcn_readfirstlane(S);
  const int gpw = 64 / S;
  const int g = lane / S;
  const int k = lane - g * S;
  const bool lane_in_group = g < gpw;
  const int gl = lane_in_group ? g : gpw - 1;
  const int gbase = gl * S;
  const bool first = (k == 0), last = (k == S - 1);
  const int m = S >> 1;
  const bool top = k < m, mid = k == m;
  const int my_step = top ? k : (k > m ? S - 1 - k : m);
  // Round 5, three attempts on the sequential loops -- 40 % of the kernel's vector instructions at 2 of a group's S lanes
  // useful -- measured and REMOVED (kill criterion of the round's brief; rows in DESIGN 3.3, records in profiles/r05_ab.txt,
  // the code in the history: commits "Lean form: neighbour exchange ... ds_bpermute" and "EXPERIMENT ... cyclic reduction"):
  //  * neighbour exchange through ds_bpermute (one fetch per double with a per-lane address instead of a DPP shift in each
  //    direction and an addition; 18 / 6 LDS-pipe instructions for 45 / 15 vector ones per step; results bit-identical):
  //    SLOWER, scenario_1 x 20 two launches 4.05 -> 4.21 ms (sweeps only 4.18, factorisation only 4.17) -- a crossbar fetch
  //    in every step's dependent chain;
  //  * an upper bound: every sequential loop at HALF its steps (timing build, results garbage, 8 iterations fixed):
  //    3.91 -> 3.28 ms -- what an elimination of half the depth could gain at most: 16 %;
  //  * that elimination: odd-even lane layout + one level of cyclic reduction (the odd joints eliminated at once, every
  //    odd lane for itself; the even joints' chain of half the length on adjacent lanes; segment neighbours through
  //    ds_bpermute outside the chains).  CORRECT on the first run (46 lean-form tests, accept sets, control points within
  //    1.5e-6 of this arrangement's) and 48 % SLOWER, 3.97 -> 5.86 ms: an odd joint keeps TWO 3 x 3 blocks (S^-1 C_a,
  //    S^-1 C_b) where a chain joint keeps one -- +18 registers in a kernel that has none: scratch 92 -> 316-412 B per
  //    lane, and no longer read-only data: 112 scratch loads and 21 stores per iteration instead of 15 and 4.
  const Shared &sh = a.sh;
  const int variant = sh.variant;
  // The kernel's arguments as the loop and the write-back see them: through a pointer into the kernarg segment that is
  // laundered once per iteration -- scalar loads where the value is used.  Left to itself the optimiser loads every
  // argument at the top of the kernel and carries ~100 SGPRs of them through the loop as spills in VGPR lanes
  // (180 v_readlane per iteration and three VGPRs).  (KernelArgs is the kernel's first parameter: offset 0.)
  typedef const KernelArgs __attribute__((address_space(4))) kargs_t;
  kargs_t *ka = (kargs_t *)__builtin_amdgcn_kernarg_segment_ptr();
#define KA_FENCE() asm volatile("" : "+s"(ka))

  // (constant address space: the table is read through scalar loads INSIDE the loop -- the pointer is laundered there, or
  //  the optimiser hoists all 168 dwords of it out of the loop and spills them to VGPR lanes: 180 v_readlane per iteration)
  typedef const double __attribute__((address_space(4))) ctable_t;
  ctable_t *mq = (ctable_t *)(mqm + axis * 84);
  const double inv_m = 1.0 / ((double)(2 * NR) * (double)S);
  auto from_prev = [&](double x) -> double { return dpp_prev(x); };
  auto from_next = [&](double x) -> double { return dpp_next(x); };

  // ---- which candidate this group solves ----
  long long cand = (long long)pair * gpw + gl;
  const bool valid = lane_in_group && cand < ncand;
  if (cand >= ncand) cand = ncand - 1;
  const int b = ORDERED ? a.order[cand0 + (int)cand] : (int)cand;

  // ---- problem data kept for the whole solve ----
  // LEAN_RELOAD: the position lines of rows 1..4 and the velocity intervals of rows 7..9 (10 doubles) are NOT kept: every
  // row pass reads the record fields they come from again (L2 hits, 512 contiguous bytes per wave instruction) and forms
  // them where its first rows use them -- the registers they held are what the allocator used to evict to scratch.
#ifndef LEAN_RELOAD
#define LEAN_RELOAD 0
#endif
  double t, it;
#if LEAN_RELOAD
  double mplo, mphi, mvlo, mvhi;               // rows 5 and 10: the joint's common intervals
  unsigned eo8;                                // byte offset of this lane's entry in a field plane of the record
#else
  double plo0, dplo, phi0, dphi, mplo, mphi;   // position rows 1..4: plo0 + r dplo; row 5: the joint's common interval
  double vl[3], vh[3], mvlo, mvhi;             // velocity rows 7..9; row 10: the joint's common interval
#endif
  double qA, qB, qend;                         // q_j = qB + qA (j + 1) - qC [j = 0] + qC [j = 5] + qend [j = 5]
  double iqn, ibn;                             // 1 / (1 + |q|), 1 / (1 + |bounds|)
  double sl[NR], su[NR], X[3];
  bool infeasible_bounds, no_solution;
  // acceleration / jerk rows: limits x t, x t^2 (solve_3d.cc:862-888, 1010-1037); the reference's reference-tracking
  // weights of this axis
  const double w_end = axis == 0 ? sh.weight_end_s : sh.weight_end_l;
  const double qC0 = -2.0 * (axis == 0 ? sh.w_s[1] : sh.w_l[1]) * (axis == 0 ? sh.ds_ref : sh.dl_ref);   // x t: the d_ref term of q
#define LLO(r) ((r) < 5 ? plo0 + (double)(r) * dplo : (r) == 5 ? mplo : (r) < 10 ? vl[((r) >= 7 && (r) <= 9) ? (r) - 7 : 0] : (r) == 10 ? mvlo : (r) < 15 ? alo : jlo)
#define LUP(r) ((r) < 5 ? phi0 + (double)(r) * dphi : (r) == 5 ? mphi : (r) < 10 ? vh[((r) >= 7 && (r) <= 9) ? (r) - 7 : 0] : (r) == 10 ? mvhi : (r) < 15 ? ahi : jhi)
// (read where they are used, through the laundered kernarg pointer: four scalars less to carry through the loop)
#define ROW_LIMITS_ACC()                                                                            \
  const double __attribute__((address_space(4))) *lim_ = &ka->sh.acc_s[0] + 2 * axis;               \
  asm volatile("" : "+s"(lim_));   /* (a copy: some passes run under a divergent condition) */      \
  const double alo = lim_[0] * t, ahi = lim_[1] * t, jlo = lim_[4] * t * t, jhi = lim_[5] * t * t
#if LEAN_RELOAD
#define ROW_LIMITS()                                                                                \
  ROW_LIMITS_ACC();                                                                                  \
  double plo0, dplo, phi0, dphi, vl[3], vh[3];                                                       \
  load_limits(plo0, dplo, phi0, dphi, vl, vh)
#else
#define ROW_LIMITS() ROW_LIMITS_ACC()
#endif
#define LL(r) lds[LN_LL + SI(r)][lane]
#define LU(r) lds[LN_LU + SI(r)][lane]
#define HSYM(H, i, j) ((i) <= (j) ? H[SYM(i, j)] : H[SYM(j, i)])
  // P block of the lane's segment (solve_3d.cc:159-171) from the scalar table
#ifndef LEAN_P_CHUNK
#define LEAN_P_CHUNK 2
#endif
#define LEAN_P_PART(H, i0_, n_)                                                                       \
  {                                                                                                   \
    ctable_t *q_ = mq;                                                                                \
    asm volatile("" : "+s"(q_));                                                                      \
    UNROLL for (int i_ = (i0_); i_ < (i0_) + (n_); i_++)                                              \
      H[i_] = 2.0 * (t3_ * q_[i_] + t * q_[21 + i_] + it * q_[42 + i_] + it3_ * q_[63 + i_]);         \
  }
  // (in parts, each behind its own laundered pointer and a scheduling barrier: the scalar loads of one part are in
  //  flight while the previous one is consumed, and no more than 2 x 4 x LEAN_P_CHUNK doubles of the table are in SGPRs)
#define LEAN_LOAD_P(H)                                                                               \
  {                                                                                                   \
    const double t3_ = t * t * t, it3_ = it * it * it;                                                \
    static_for<(21 + LEAN_P_CHUNK - 1) / LEAN_P_CHUNK>([&](auto c_c) {                                \
      constexpr int c0_ = decltype(c_c)::value * LEAN_P_CHUNK;                                        \
      constexpr int cn_ = 21 - c0_ < LEAN_P_CHUNK ? 21 - c0_ : LEAN_P_CHUNK;                          \
      LEAN_P_PART(H, c0_, cn_)                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                              \
    });                                                                                               \
    H[SYM(5, 5)] += last ? 2.0 * w_end * (t * t) : 0.0;                                               \
  }
  // control points of the lane's segment from the joint states at its two ends
  auto control_points = [&](const double (&Xe)[3], double (&c)[6]) {
    const NullMap nm = {it, t * 0.05};
    double Xp[3];
    UNROLL for (int i = 0; i < 3; i++) { const double v = from_prev(Xe[i]); const double x0 = lds[LN_XI + i][lane]; Xp[i] = first ? x0 : v; }
    U_apply(nm, Xp, c[0], c[1], c[2]);
    V_apply(nm, Xe, c[3], c[4], c[5]);
  };

  // Every row pass recomputes what it needs from the slacks: without this fence the optimiser recognises the common
  // subexpressions of two passes (reciprocals, residuals, complementarity targets) and keeps 30-entry arrays of them
  // alive from one pass to the next -- the state this form exists to drop.
  // (The same goes for everything derived from the problem data -- t^3, the row limits, q, the P block itself: loop
  //  invariants in the optimiser's eyes, which it would compute once, before the loop, and keep.)
  auto fence_slacks = [&]() {
    UNROLL for (int i = 0; i < NR; i++) { asm volatile("" : "+v"(sl[i])); asm volatile("" : "+v"(su[i])); }
#if LEAN_RELOAD
    asm volatile("" : "+v"(t), "+v"(it), "+v"(qA), "+v"(qB));
#else
    asm volatile("" : "+v"(t), "+v"(it), "+v"(plo0), "+v"(dplo), "+v"(phi0), "+v"(dphi), "+v"(qA), "+v"(qB));
#endif
  };
  // The scheduler may not move anything across the boundary between two rows: left alone it starts the reciprocals and
  // LDS reads of all fifteen rows at the top of a pass (latency it cannot know the second wavefront hides) and the
  // register allocator pays with scratch.
#define ROW_SEP() __builtin_amdgcn_sched_barrier(0)
// (a barrier in front of every second row: pairs of rows give the scheduler room at no extra scratch -- measured: every
//  row 5.77 ms, every second 5.65, every third 5.76 with 8-16 B more scratch)
// LEAN_PRIO: wave priority raised inside the sequential loops (dependent chains: few instructions, each waiting for the
// last) so that the SIMD's other wavefront -- in a row pass, say, with instructions to spare -- fills the gaps instead of
// getting in front.
// (measured, 65 536 x 20: scenario_1 two launches 5.35 -> 5.27 ms, one launch 5.63 -> 5.60; generic +-0)
// -DLEAN_MARKS: the phases named in the ISA (tools: per-phase instruction counts of a -S build)
#ifdef LEAN_MARKS
#define LEAN_MARK(x) asm volatile("; ==PHASE " x)
#else
#define LEAN_MARK(x)
#endif
#ifndef LEAN_PRIO
#define LEAN_PRIO 3
#endif
#if LEAN_PRIO
#define SEQ_BEGIN() __builtin_amdgcn_s_setprio(LEAN_PRIO)
#define SEQ_END() __builtin_amdgcn_s_setprio(0)
#else
#define SEQ_BEGIN()
#define SEQ_END()
#endif
#ifndef LEAN_ROW_GROUP
#define LEAN_ROW_GROUP 2
#endif
#ifndef LEAN_PROJ_SEP
#define LEAN_PROJ_SEP 1
#endif
#if LEAN_PROJ_SEP
#define PROJ_SEP() __builtin_amdgcn_sched_barrier(0)
#else
#define PROJ_SEP()
#endif
#define ROW_SEP_R(r) do { if constexpr (SI(r) % LEAN_ROW_GROUP == 0) __builtin_amdgcn_sched_barrier(0); } while (0)
  // Reciprocal slacks of the passes that form the step (Newton block, corrector, ratios, update).  The bare v_rcp_f64
  // seed (5e-8 relative) perturbs the Newton direction by as much -- an inexact Newton step; the iterate is EVALUATED
  // (residuals, complementarity, score: pass A1) without any reciprocal, so what the solve converges to and when it
  // stops do not depend on it.  LEAN_SEED=0: one Newton step on every reciprocal, as the packed form has it.
#ifndef LEAN_SEED
#define LEAN_SEED 1
#endif
#if LEAN_SEED
#define LEAN_RCP rcp_fast
#else
#define LEAN_RCP rcp
#endif
  // 1 / a and 1 / b from ONE reciprocal (v_rcp_f64 issues at a quarter of the rate of a multiplication): r = 1 / (a b),
  // 1 / a = b r, 1 / b = a r.  Slacks and multipliers lie between 1e-13 and 1e4: the product is far from the ends of
  // the exponent range.
#define RCP_PAIR(RCP, a_, b_, ia_, ib_) const double rab_ = RCP((a_) * (b_)); const double ia_ = (b_) * rab_, ib_ = (a_) * rab_
  // termination bookkeeping (group-uniform)
  double best_score = 1e300;
  float best_res = 3e38f;
  int best_it = 0, res_it = 0, iters = 0, eit = 0, tiny_steps = 0;
  bool plain = false, done, suspended = false;
  // warm-start instantiations: a group whose guess does not pay off gets ONE cold restart (the packed form's rule)
  [[maybe_unused]] bool warm_started = false, restarted = true;
  [[maybe_unused]] int it0 = 0;          // iteration at which the current start was made
  [[maybe_unused]] size_t lam_e = 0;
#if LEAN_RELOAD
  // position lines and velocity intervals from the record, as the set-up forms them (solve_3d.cc:827-859, 965-966,
  // 1003-1004; cuboid_3d.cc:677-689, 826-827)
  auto load_limits = [&](double &plo0, double &dplo, double &phi0, double &dphi, double (&vl)[3], double (&vh)[3]) {
    kargs_t *kb = ka;
    asm volatile("" : "+s"(kb));   // (copies: some passes run under a divergent condition)
    unsigned eo = eo8;
    asm volatile("" : "+v"(eo));
    const char *sg = (const char *)kb->seg;
    const size_t fs8 = (size_t)kb->B * (size_t)kb->seg_stride * 8;
    auto fld = [&](int f) -> double { return *(const double *)(sg + (size_t)f * fs8 + eo); };
    if (variant != BTRAPZ_CUBOID) {
      const int f0 = axis == 0 ? (int)BTRAPZ_F_DOWN_BIAS : (int)BTRAPZ_F_L_DOWN_BIAS;
      const double lb = fld(f0), ls = fld(f0 + 1), ub = fld(f0 + 2), us = fld(f0 + 3);
      plo0 = lb; dplo = ls * 0.2 * t; phi0 = ub; dphi = us * 0.2 * t;
    } else {
      if (axis == 0) {
        const double lb = fld(BTRAPZ_F_DOWN_BIAS), ls = fld(BTRAPZ_F_DOWN_SKEW), ub = fld(BTRAPZ_F_UPP_BIAS), us = fld(BTRAPZ_F_UPP_SKEW);
        plo0 = fmax(0.0, fmax(ls * 0.0 + lb, lb + ls * t));
        phi0 = fmin(100.0, fmin(us * 0.0 + ub, ub + us * t));
      } else {
        plo0 = fld(BTRAPZ_F_BEG_L); phi0 = fld(BTRAPZ_F_END_L);
      }
      dplo = 0.0; dphi = 0.0;
    }
    if (axis == 0) {
      const double lo = fld(BTRAPZ_F_DS_LO), hi = fld(BTRAPZ_F_DS_HI);
      UNROLL for (int i = 0; i < 3; i++) { vl[i] = lo; vh[i] = hi; }
    } else {
      const double *dl = (const double *)((const char *)kb->dl_bounds + (unsigned)b * 80u);
      UNROLL for (int i = 0; i < 3; i++) { vl[i] = dl[2 * (i + 1)]; vh[i] = dl[2 * (i + 1) + 1]; }
    }
  };
#endif
  // cold start at the current X: slacks max(gap, BTRAPZ_COLD_SLACK), multipliers BTRAPZ_COLD_LAMBDA (btrapz_ipm.h)
  auto cold_rows = [&]() {
    double c[6];
    control_points(X, c);
    ROW_LIMITS();
    FOR_ROWS(r)
      const double gc_r = row_dot<r>(c, t);
      sl[SI(r)] = fmax(gc_r - LLO(r), BTRAPZ_COLD_SLACK); su[SI(r)] = fmax(LUP(r) - gc_r, BTRAPZ_COLD_SLACK);
      LL(r) = BTRAPZ_COLD_LAMBDA; LU(r) = BTRAPZ_COLD_LAMBDA;
    END_ROWS
    // rows without a real bound (btrapz_ipm.h, "bounds that are no bounds"): the centred multiplier, in a pass of its
    // own behind a wave-uniform branch that is all but never taken (the packed form's arrangement)
    bool far_row = false;
    FOR_ROWS(r)
      far_row |= sl[SI(r)] > BTRAPZ_COLD_FAR || su[SI(r)] > BTRAPZ_COLD_FAR;
    END_ROWS
    if (__any(far_row)) {
      UNIFORM_BLOCK;
      FOR_ROWS(r)
        LL(r) = cold_lambda(sl[SI(r)], BTRAPZ_COLD_SLACK, BTRAPZ_COLD_LAMBDA); LU(r) = cold_lambda(su[SI(r)], BTRAPZ_COLD_SLACK, BTRAPZ_COLD_LAMBDA);
      END_ROWS
    }
  };

  // ---- set-up: the candidate's record, bounds, consistency, cold start ------------------------------------------
  {
    const size_t BS = (size_t)a.B * a.seg_stride;
    const double *sg = a.seg;
    const size_t e_ = (size_t)b * a.seg_stride + k;
    t = sg[BTRAPZ_F_T * BS + e_];
#if LEAN_RELOAD
    double plo0, dplo, phi0, dphi, vl[3], vh[3];   // (this block's own: the passes form theirs)
    eo8 = (unsigned)e_ * 8u;
#endif
    double lb, ls, ub, us, begl = 0.0, endl = 0.0, rv[10];
    if (axis == 0) {
      lb = sg[BTRAPZ_F_DOWN_BIAS * BS + e_]; ls = sg[BTRAPZ_F_DOWN_SKEW * BS + e_];
      ub = sg[BTRAPZ_F_UPP_BIAS * BS + e_];  us = sg[BTRAPZ_F_UPP_SKEW * BS + e_];
      rv[0] = sg[BTRAPZ_F_DS_LO * BS + e_]; rv[1] = sg[BTRAPZ_F_DS_HI * BS + e_];
      UNROLL for (int i = 2; i < 10; i++) rv[i] = 0.0;
    } else {
      lb = sg[BTRAPZ_F_L_DOWN_BIAS * BS + e_]; ls = sg[BTRAPZ_F_L_DOWN_SKEW * BS + e_];
      ub = sg[BTRAPZ_F_L_UPP_BIAS * BS + e_];  us = sg[BTRAPZ_F_L_UPP_SKEW * BS + e_];
      if (variant == BTRAPZ_CUBOID) { begl = sg[BTRAPZ_F_BEG_L * BS + e_]; endl = sg[BTRAPZ_F_END_L * BS + e_]; }
      UNROLL for (int i = 0; i < 10; i++) rv[i] = a.dl_bounds[(size_t)b * 10 + i];
    }
    const double skew = sg[(axis == 0 ? BTRAPZ_F_X_SKEW : BTRAPZ_F_Y_SKEW) * BS + e_];
    const double bias = sg[(axis == 0 ? BTRAPZ_F_X_BIAS : BTRAPZ_F_Y_BIAS) * BS + e_];
    const double ref_end = a.ref_end[(size_t)b * 2 + axis];
    double Xinit[3];
    UNROLL for (int i = 0; i < 3; i++) Xinit[i] = a.init[(size_t)b * 6 + axis * 3 + i];
    it = 1.0 / t;
    // position rows: lo_i = plo0 + i dplo, up_i = phi0 + i dphi (solve_3d.cc:827-828,965-966); cuboid: one interval
    plo0 = lb; dplo = ls * 0.2 * t; phi0 = ub; dphi = us * 0.2 * t;
    if (variant == BTRAPZ_CUBOID) {
      if (axis == 0) {   // cuboid_3d.cc:677-689 (the NaN-skipping max / min of the reference: see the packed form)
        const double lo = fmax(0.0, fmax(ls * 0.0 + lb, lb + ls * t));
        const double hi = fmin(100.0, fmin(us * 0.0 + ub, ub + us * t));
        plo0 = lo; phi0 = hi;
      } else {           // cuboid_3d.cc:826-827
        plo0 = begl; phi0 = endl;
      }
      dplo = 0.0; dphi = 0.0;
    }
    // velocity rows (solve_3d.cc:835-859 s axis; :1003-1004 l axis: dy_bounds_[i], i = row index)
    double vlo[5], vhi[5];
    if (axis == 0) {
      UNROLL for (int i = 0; i < 5; i++) { vlo[i] = rv[0]; vhi[i] = rv[1]; }
    } else {
      UNROLL for (int i = 0; i < 5; i++) { vlo[i] = rv[2 * i]; vhi[i] = rv[2 * i + 1]; }
    }
    ROW_LIMITS_ACC();
    const double far_cut = move_far_bounds(plo0, dplo, phi0, dphi, vlo, vhi);   // bounds that are none (btrapz_ipm.h)
    mplo = plo0 + 5.0 * dplo; mphi = phi0 + 5.0 * dphi; mvlo = vlo[4]; mvhi = vhi[4];
    // rows of the reference (all 18) for the consistency checks
#define LO0(r) ((r) < 6 ? plo0 + (double)(r) * dplo : (r) < 11 ? vlo[((r) >= 6 && (r) < 11) ? (r) - 6 : 0] : (r) < 15 ? alo : jlo)
#define UP0(r) ((r) < 6 ? phi0 + (double)(r) * dphi : (r) < 11 ? vhi[((r) >= 6 && (r) < 11) ? (r) - 6 : 0] : (r) < 15 ? ahi : jhi)
    double gapmin = 1e300, bnorm = 0.0;
    static_for<18>([&](auto r_c) {
      constexpr int r = decltype(r_c)::value;
      // |bounds|: the real ones (a moved bound sits at far_cut exactly; header limits at BTRAPZ_FAR_LIMIT x t, x t^2)
      const double cut = r < 11 ? far_cut : r < 15 ? BTRAPZ_FAR_LIMIT * t : BTRAPZ_FAR_LIMIT * t * t;
      const double al = fabs(LO0(r)), au = fabs(UP0(r));
      const double rb = fmax(al < cut ? al : 0.0, au < cut ? au : 0.0);
      gapmin = fmin(gapmin, (UP0(r) - LO0(r)) + 1e-12);   // (1e-12: the tolerance of the test suite's exact solver)
      bnorm = fmax(bnorm, rb);
    });
    // the joint at the end of the segment: this lane's last rows and the next lane's first ones bound the same
    // quantities (rows kept once, see rows_kept); two sides that just touch are pinned to the common point
    bool joint_empty = false;
    {
      const double nplo = from_next(plo0), nphi = from_next(phi0), nvlo = from_next(vlo[0]), nvhi = from_next(vhi[0]);
      if (!last) {
        const bool own_ok = mplo <= mphi && mvlo <= mvhi && nplo <= nphi && nvlo <= nvhi;
        mplo = fmax(mplo, nplo); mphi = fmin(mphi, nphi); mvlo = fmax(mvlo, nvlo); mvhi = fmin(mvhi, nvhi);
        const double ptol = 1e-9 * (1.0 + fmax(fabs(mplo), fabs(mphi))), vtol = 1e-9 * (1.0 + fmax(fabs(mvlo), fabs(mvhi)));
        joint_empty = own_ok && (mplo > mphi + ptol || mvlo > mvhi + vtol);
        if (mplo > mphi && !(mplo > mphi + ptol)) { mplo = 0.5 * (mplo + mphi); mphi = mplo; }
        if (mvlo > mvhi && !(mvlo > mvhi + vtol)) { mvlo = 0.5 * (mvlo + mvhi); mvhi = mvlo; }
      }
    }
    // segment 0's first position / velocity / acceleration row state the given initial state
    bool no_solution_lane;
    {
      const NullMap nm = {it, t * 0.05};
      double c0, c1, c2;
      U_apply(nm, Xinit, c0, c1, c2);
      auto outside = [&](double g_, double lo, double hi) {
        const double tol = 1e-7 * (1.0 + fmax(fabs(lo), fabs(hi)));
        return !(g_ >= lo - tol && g_ <= hi + tol);
      };
      no_solution_lane = (first && (outside(t * c0, LO0(0), UP0(0)) || outside(5.0 * (c1 - c0), LO0(6), UP0(6)) ||
                                    outside(20.0 * ((c0 - 2.0 * c1) + c2), LO0(11), UP0(11)))) || joint_empty;
    }
#undef LO0
#undef UP0
    UNROLL for (int i = 0; i < 3; i++) { vl[i] = vlo[i + 1]; vh[i] = vhi[i + 1]; }
    // q (solve_3d.cc:248-268): the monomial coefficients -2 t^3 w skew / (i + 2) - 2 t^2 w bias / (i + 1) [- 2 w_d d_ref t,
    // i > 0] times M.  Column j of M holds the monomial coefficients of the Bernstein polynomial B_j: sum_i M_ij / (i + 1)
    // is its integral, 1 / 6; sum_i M_ij / (i + 2) its first moment, (j + 1) / 42; sum_{i > 0} M_ij = B_j(1) - B_j(0).
    const double wr = axis == 0 ? sh.w_s[0] : sh.w_l[0];
    qA = (-2.0 * (t * t * t) * wr * skew) * (1.0 / 42.0);
    qB = (-2.0 * (t * t) * wr * bias) * (1.0 / 6.0);
    qend = last ? -(axis == 0 ? sh.ds_ref : sh.dl_ref) * 2.0 * ref_end * t : 0.0;   // :268/:315 (multiplies by d_ref: bug-compatible)
    double qn = 0.0;
    {
      const double qC = qC0 * t;
      UNROLL for (int j = 0; j < 6; j++) {
        const double qj = (qB + qA * (double)(j + 1)) + (j == 0 ? -qC : j == 5 ? qC + qend : 0.0);
        qn = fmax(qn, fabs(qj));
      }
    }
    // starting point: constant-velocity propagation of the initial state
    double Xcold0;
    {
      double tsum = 0.0;
      wave_lds_sync();
      lds[LN_RED][lane] = t;
      wave_lds_sync();
      for (int j = 0; j < S; j++) tsum += (j <= k) ? lds[LN_RED][gbase + j] : 0.0;
      Xcold0 = Xinit[0] + Xinit[1] * tsum;
    }
    {
      const Red4 r0 = group_reduce<0, 1, 1, 2>(lds + LN_RED, lane, gbase, k, S, no_solution_lane ? 1.0 : 0.0, bnorm, qn, gapmin);
      bnorm = r0.b; qn = r0.c; gapmin = r0.d;
      no_solution = r0.a > 0.0;
    }
    infeasible_bounds = !(gapmin >= 0.0) || !(t > 0.0);
    iqn = 1.0 / (1.0 + qn); ibn = 1.0 / (1.0 + bnorm);
    // The initial state lives in LDS rows of which only a group's FIRST lane reads its own entry (control_points): the
    // other lanes' entries of row 0 hold their cold-start position instead (the warm-start instantiations restart from
    // it), row 1 holds the candidate's initial velocity in every lane.
    UNROLL for (int i = 0; i < 3; i++) lds[LN_XI + i][lane] = (i == 0 && !first) ? Xcold0 : Xinit[i];
    X[0] = Xcold0; X[1] = Xinit[1]; X[2] = 0.0;
    lam_e = (size_t)axis * 36 * BS + (size_t)b * a.seg_stride + k;   // this lane in the warm-start arrays [axis][row 0..35][b][k]
    warm_started = WARM && (a.x0 || a.lam0);
    if (WARM && warm_started) {
      // warm: slacks floored at smin, multipliers = earlier multipliers (sanitised) + mu0 / s, so that every
      // complementarity product is at least mu0 (the packed form's rule)
      if (a.x0) {
        const double *xw = a.x0 + (((size_t)b * 2 + axis) * a.seg_stride + k) * 3;
        const double w0 = xw[0], w1 = xw[1], w2 = xw[2];
        if (fabs(w0) < 1e300 && fabs(w1) < 1e300 && fabs(w2) < 1e300) { X[0] = w0; X[1] = w1; X[2] = w2; }
      }
      double c[6];
      control_points(X, c);
      const double smin = a.smin, mu0 = a.mu0;
      FOR_ROWS(r)
        const double gc_r = row_dot<r>(c, t);
        const double s_l = fmax(gc_r - LLO(r), smin), s_u = fmax(LUP(r) - gc_r, smin);
        double l_l = 0.0, l_u = 0.0;
        if (a.lam0) {
          const double p_l = a.lam0[lam_e + (size_t)r * BS], p_u = a.lam0[lam_e + (size_t)(18 + r) * BS];
          l_l = (p_l >= 0.0 && p_l < 1e300) ? p_l : 0.0; l_u = (p_u >= 0.0 && p_u < 1e300) ? p_u : 0.0;
        }
        sl[SI(r)] = s_l; su[SI(r)] = s_u;
        LL(r) = l_l + mu0 * rcp(s_l); LU(r) = l_u + mu0 * rcp(s_u);
      END_ROWS
    } else {
      cold_rows();
    }
    UNROLL for (int i = 0; i < 3; i++) lds[LN_XB + i][lane] = X[i];
    restarted = !warm_started;
    done = !valid || infeasible_bounds || no_solution;
  }

  // ---- capped / resume: a group's iterate in a.susp_state, slot-major, field i of lane k at [slot][i][k] ----------
  [[maybe_unused]] auto state_io = [&](const bool store, const long long slot) {
    double *base = ka->susp_state + (size_t)slot * LEAN_SUSP_FIELDS * ka->seg_stride + k;
    const size_t fs = ka->seg_stride;
    int f = 0;
    auto io = [&](double &v) { if (store) base[(size_t)f * fs] = v; else v = base[(size_t)f * fs]; ++f; };
    UNROLL for (int i = 0; i < 3; i++) io(X[i]);
    UNROLL for (int i = 0; i < 3; i++) { double xb = lds[LN_XB + i][lane]; io(xb); if (!store) lds[LN_XB + i][lane] = xb; }
    FOR_ROWS(r)
      io(sl[SI(r)]); io(su[SI(r)]);
      double l_ = LL(r), u_ = LU(r);
      io(l_); io(u_);
      if (!store) { LL(r) = l_; LU(r) = u_; }
    END_ROWS
    // the group's bookkeeping in three doubles: the four counters (12 bits each -- a hand-over happens within cap_hi
    // iterations, and the host keeps that below 4000; res_it may be -1), a flag and the tiny-step count share one, exactly
    double g3[3] = {best_score, (double)best_res,
                    (double)best_it + 4096.0 * (double)(res_it + 1) + 16777216.0 * (double)eit + 68719476736.0 * (double)iters +
                        (plain ? 281474976710656.0 : 0.0) + 562949953421312.0 * (double)tiny_steps};
    UNROLL for (int i = 0; i < 3; i++) io(g3[i]);
    if (!store) {
      best_score = g3[0]; best_res = (float)g3[1];
      const long long code = (long long)g3[2];
      best_it = (int)(code & 4095); res_it = (int)((code >> 12) & 4095) - 1; eit = (int)((code >> 24) & 4095); iters = (int)((code >> 36) & 4095);
      plain = ((code >> 48) & 1) != 0; tiny_steps = (int)((code >> 49) & 3);
    }
  };
  // resume: carry on where the capped launch stopped.  The iterate handed over has been evaluated there (best iterate,
  // stall marks, second chance): the first pass here forms its Newton step without evaluating it a second time.
  [[maybe_unused]] bool handed_over = false;
  [[maybe_unused]] int susp_slot_ = -1;
  [[maybe_unused]] bool lone_start = false;   // (capped launch: the group was the only live one of its wavefront at the first iteration)
  [[maybe_unused]] float susp_score_ = 1.0f;
  if constexpr (RESUME) {
    if (valid) state_io(false, a.susp_slot[2LL * b + axis]);
    handed_over = true;
  }

#ifdef LEAN_TRACE   // (debug build: per-iteration score, primal / dual residual, mu, step, centring instead of the control points)
  double tr_score = 0.0, tr_res = 0.0, tr_mu = 0.0, tr_sr = 0.0, tr_pr = 0.0, tr_dr = 0.0;
#endif
  for (;;) {
    KA_FENCE();
    LEAN_MARK("A1");
    // ---- A1. control points; gradient P c + q + G' (lambda_u - lambda_l), residuals, complementarity ----
    double rd[3];
    double mu_part = 0.0, rp_part = 0.0, dscale = 0.0, rd_part;
    PHASE_FENCE(fence_slacks());
    double c[6], H[21];   // control points and P block: A1's gradient, then (H += G' W G) A2's Newton block
    control_points(X, c);
    {
      double gr[6];
      {
        LEAN_LOAD_P(H)
        const double qC = qC0 * t;
        UNROLL for (int i = 0; i < 6; i++) {
          double s = 0.0;
          UNROLL for (int j = 0; j < 6; j++) s += HSYM(H, i, j) * c[j];
          dscale = fmax(dscale, fabs(s));
          const double qi = (qB + qA * (double)(i + 1)) + (i == 0 ? -qC : i == 5 ? qC + qend : 0.0);
          gr[i] = s + qi;
        }
      }
      {
        ROW_LIMITS();
        FOR_ROWS(r)
          ROW_SEP_R(r);
          const double ll = LL(r), lu = LU(r);
          const double gcr = row_dot<r>(c, t);
          const double s_l = sl[SI(r)], s_u = su[SI(r)];
          const double rpl = gcr - s_l - LLO(r), rpu = gcr + s_u - LUP(r);
          rp_part = fmax(rp_part, fmax(fabs(rpl), fabs(rpu)));
          mu_part += s_l * ll + s_u * lu;
          // (the scale of the gradient's terms, for the round-off floor of the dual residual: the row's largest
          //  coefficient times its multiplier difference)
          dscale = fmax(dscale, fabs(lu - ll) * (r < 6 ? t : r < 11 ? 5.0 : r < 15 ? 40.0 : 180.0));
          row_scatter<r>(lu - ll, t, gr);
        END_ROWS
      }
      ROW_SEP();
      // reduced to the joint states: X_{k+1} collects V' (.)[3..5] of this lane and U' (.)[0..2] of the next
      const NullMap nm = {it, t * 0.05};
      double un[3];
      VT_apply(nm, gr[3], gr[4], gr[5], rd);
      UT_apply(nm, gr[0], gr[1], gr[2], un);
      UNROLL for (int i = 0; i < 3; i++) { const double v = from_next(un[i]); rd[i] += last ? 0.0 : v; }
      rd_part = fmax(fabs(rd[0]), fmax(fabs(rd[1]), fabs(rd[2])));
    }
    LEAN_MARK("TERM");
    // a lane whose residuals are not finite must poison its group's score (fmax / fmin ignore NaN)
    if (!(mu_part == mu_part) || !(rd_part == rd_part) || !(rp_part == rp_part) || !(dscale == dscale) ||
        !(fabs(rd_part) < 1e300) || !(fabs(mu_part) < 1e300))
      rp_part = 1e300;
    const Red4 rr = group_reduce_mixed<0, 1, 1, 1>(lds + LN_RED, lane, gbase, k, S, lane_in_group, mu_part, rd_part, rp_part, dscale);
    const double mu = rr.a * inv_m;
    [[maybe_unused]] bool restart_now = false;
    // ---- termination (the packed form's rules: DESIGN.md 3.4) ----
    {
      const double rd_eff = fmax(rr.b - 2e-13 * rr.d, 0.0);
      const double rprim = rr.c * ibn;
      const double res = fmax(rd_eff * iqn, rprim);
      const double score = (mu == mu) ? fmax(res, mu) : 1e300;   // (fmax would drop a NaN mu: ADVICE r4)
#ifdef LEAN_TRACE
      tr_score = score; tr_res = rd_eff * iqn; tr_mu = mu; tr_pr = rprim;
#endif
      const double mu_primal = fmax(mu, rprim);
      const bool feasible_and_complementary = mu_primal < 1e-7;
      if (!done && !(RESUME && handed_over)) {
        iters = eit;
        if (score < best_score) {
          best_score = score; best_it = eit;
          UNROLL for (int i = 0; i < 3; i++) lds[LN_XB + i][lane] = X[i];
        }
        if ((float)res < ka->stall_factor * best_res) { best_res = (float)res; res_it = eit; }
        const bool stalled = (eit - it0 >= ka->stall_start && eit - best_it >= ka->stall_len && eit - res_it >= ka->stall_len) ||
                             (mu > (double)ka->diverge_factor * best_score) || !(score < 1e299) || tiny_steps >= BTRAPZ_TINY_STEPS;
        const int patience = (best_score < 1e-7 && mu_primal < 1e-2 * best_score) ? 1 : 3;
        const bool at_floor = best_score < (feasible_and_complementary ? 1e-4 : 1e-5) && eit - best_it >= patience;
        if (at_floor && feasible_and_complementary) res_it = -1;
        if (score < ka->eps || at_floor) done = true;
        // a warm-started group that stalls, or is still far from converged after 12 iterations (a useful guess needs
        // about 5, a cold start 8-14), or is still running after 24, restarts once from the cold start
        else if (WARM && !restarted && (stalled || (eit - it0 >= 12 && best_score > 1e-4) || eit - it0 >= 24)) restart_now = true;
        else if (stalled) {
          if (!plain && res < 1e-6 && score < 1e299) { plain = true; best_it = eit; res_it = eit; }
          else done = true;
        }
        if (!done && !restart_now && eit + 1 >= ka->max_iter) done = true;
      }
      handed_over = false;
      if constexpr (CAPPED) {
        // the cap: who hands over is the packed form's rule (a group alone in its wavefront after cap_iter iterations,
        // any group after cap_hi)
        const int nact = __popcll(__ballot(first && lane_in_group && valid && !done));
        // (a group that is alone from the first iteration on -- its neighbours have no solution, a quarter of the cuboid
        //  bench batch -- does not wait for cap_iter: it hands over after its first iteration; cuboid batch 4.08 -> 3.96 ms)
        if (eit == 0) lone_start = nact <= ka->cap_alone;
        const bool want = !done && valid && susp_slot_ != -2 && ((eit >= (lone_start ? 1 : ka->cap_iter) && nact <= ka->cap_alone && score >= ka->cap_score) || eit >= ka->cap_hi);
        if (__builtin_expect(__any(want), 0)) {   // (rare: at most once per group -- a group that finds no slot does not ask again)
          UNIFORM_BLOCK;
          wave_lds_sync();
          if (want && first) lds[LN_RED][lane] = (double)atomicAdd(ka->susp_count, 1);
          wave_lds_sync();
          const long long slot = want ? (long long)lds[LN_RED][gbase] : -1;
          if (want && slot < (long long)ka->susp_cap) {   // (no room: the group simply goes on)
            // The group is done as far as this launch goes: nothing of its iterate or bookkeeping changes any more, so
            // the record is written after the loop (state_io in here costs the allocator 84 B of scratch per lane).
            susp_slot_ = (int)slot; susp_score_ = (float)fmax(fmin(score, 1e3), 1e-12);
            suspended = true; done = true;
          } else if (want) {
            susp_slot_ = -2;   // no room: the group goes on to the end
          }
        }
      }
    }
    if (__all(done)) break;
    if constexpr (WARM) {
      if (__any(restart_now)) {
        // wave-uniform branch; the other groups of the wavefront only lose this pass's Newton step.  A group restarts as a
        // whole (the score is group-uniform), so the DPP reads of control_points stay inside it.
        UNIFORM_BLOCK;
        if (restart_now) {
          const double xc = lds[LN_XI][lane], v0 = lds[LN_XI + 1][lane];
          X[0] = first ? xc + v0 * t : xc; X[1] = v0; X[2] = 0.0;     // (a first lane's row 0 is the initial position itself)
          cold_rows();
          best_score = 1e300; best_it = eit + 1; it0 = eit + 1; restarted = true; best_res = 3e38f; res_it = eit + 1; tiny_steps = 0;
          UNROLL for (int i = 0; i < 3; i++) lds[LN_XB + i][lane] = X[i];
          ++eit;       // (the other groups did not take a step in this pass: their count stands)
        }
        continue;
      }
    }

    LEAN_MARK("A2");
    // ---- A2. Newton block H = P + G' W G and the predictor's right-hand side (rc = s lambda: tv = lambda_l (s_l +
    // rp_l) / s_l - lambda_u (s_u - rp_u) / s_u), then M = Phi' H Phi: block tridiagonal T, M01 ----
    double M01[9], T[6], up[3];
    PHASE_FENCE(opaque6(c); fence_slacks());
    {
      double hr[6];
      UNROLL for (int i = 0; i < 6; i++) hr[i] = 0.0;
      {
        ROW_LIMITS();
        const double t2 = t * t;
        FOR_ROWS(r)
          ROW_SEP_R(r);
          const double ll = LL(r), lu = LU(r);
          const double gcr = row_dot<r>(c, t);
          const double s_l = sl[SI(r)], s_u = su[SI(r)];
          RCP_PAIR(LEAN_RCP, s_l, s_u, isl, isu);
          const double wl = ll * isl, wu = lu * isu;
          row_outer<r>(wl + wu, t2, H);
          // s_l + rp_l = G c - l and s_u - rp_u = u - G c: the slacks drop out of the predictor's right-hand side (round 5:
          // formed directly -- four additions per row less than through the residuals, and one rounding less each)
          row_scatter<r>(wl * (gcr - LLO(r)) - wu * (LUP(r) - gcr), t, hr);
        END_ROWS
      }
      ROW_SEP();
      const NullMap nm = {it, t * 0.05};
      {
        double hn[3];
        VT_apply(nm, hr[3], hr[4], hr[5], up);
        UT_apply(nm, hr[0], hr[1], hr[2], hn);
        UNROLL for (int i = 0; i < 3; i++) { const double w = from_next(hn[i]); up[i] = -((up[i] + (last ? 0.0 : w)) + rd[i]); }
      }
      ROW_SEP();
      double w0[3], w1[3], w2[3], col[3], M00[6];
      UT_apply(nm, H[SYM(0, 0)], H[SYM(0, 1)], H[SYM(0, 2)], w0);
      UT_apply(nm, H[SYM(0, 1)], H[SYM(1, 1)], H[SYM(1, 2)], w1);
      UT_apply(nm, H[SYM(0, 2)], H[SYM(1, 2)], H[SYM(2, 2)], w2);
      UT_apply(nm, w0[0], w1[0], w2[0], col); M00[0] = col[0]; M00[1] = col[1]; M00[2] = col[2];
      UT_apply(nm, w0[1], w1[1], w2[1], col); M00[3] = col[1]; M00[4] = col[2];
      UT_apply(nm, w0[2], w1[2], w2[2], col); M00[5] = col[2];
      PROJ_SEP();
      VT_apply(nm, H[SYM(0, 3)], H[SYM(0, 4)], H[SYM(0, 5)], w0);
      VT_apply(nm, H[SYM(1, 3)], H[SYM(1, 4)], H[SYM(1, 5)], w1);
      VT_apply(nm, H[SYM(2, 3)], H[SYM(2, 4)], H[SYM(2, 5)], w2);
      UNROLL for (int j = 0; j < 3; j++) {
        UT_apply(nm, w0[j], w1[j], w2[j], col);
        M01[0 * 3 + j] = col[0]; M01[1 * 3 + j] = col[1]; M01[2 * 3 + j] = col[2];
      }
      PROJ_SEP();
      VT_apply(nm, H[SYM(3, 3)], H[SYM(3, 4)], H[SYM(3, 5)], w0);
      VT_apply(nm, H[SYM(3, 4)], H[SYM(4, 4)], H[SYM(4, 5)], w1);
      VT_apply(nm, H[SYM(3, 5)], H[SYM(4, 5)], H[SYM(5, 5)], w2);
      VT_apply(nm, w0[0], w1[0], w2[0], col); T[0] = col[0]; T[1] = col[1]; T[2] = col[2];
      VT_apply(nm, w0[1], w1[1], w2[1], col); T[3] = col[1]; T[4] = col[2];
      VT_apply(nm, w0[2], w1[2], w2[2], col); T[5] = col[2];
      UNROLL for (int i = 0; i < 6; i++) { const double v = from_next(M00[i]); T[i] += last ? 0.0 : v; }
    }

    LEAN_MARK("B");
    // ---- B. two-sided block LDL^T (lanes s and S-1-s own the pivots of step s; block m = S/2 is the root); the
    // predictor's forward sweep rides along.  MK: M01 of the neighbour towards the root until the lane's step, then
    // K = S_k^{-1} Mc; TF: the diagonal block until then, then its factor ----
    double MK[9], TF[6], Z[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    {
      double wp[3] = {0.0, 0.0, 0.0};
      UNROLL for (int i = 0; i < 3; i++)
        UNROLL for (int j = 0; j < 3; j++) {
          const double nM = from_next(M01[i * 3 + j]);
          MK[i * 3 + j] = top ? nM : M01[j * 3 + i];
        }
      UNROLL for (int i = 0; i < 6; i++) TF[i] = T[i];
      SEQ_BEGIN();
      // Steps 0 .. m-1, the two chains.  Every lane but the root wants ONE neighbour, the one away from the root: the shift
      // from the left for all lanes, then the shift from the right under the BOTTOM lanes' execution mask into the same
      // registers (a branch, not a select: a source lane outside the mask -- the first lane of the next group, right of
      // a group's last -- reads as 0, which is what that lane's step 0 wants).  Four moves per double and no addition
      // (round 5: until then both shifts for every lane and p + n, the unwanted side still 0 -- the same numbers;
      // scenario_1 x 20 two launches 4.00 -> 3.90 ms, generic 3.68 -> 3.63, cuboid 3.38 -> 3.32, 10 segments 1.63 -> 1.61).
      for (int step = 0; step < m; ++step) {
        double zin[6], win[3];
        UNROLL for (int i = 0; i < 6; i++) zin[i] = from_prev(Z[i]);
        UNROLL for (int i = 0; i < 3; i++) win[i] = from_prev(wp[i]);
        if (k > m) {
          UNIFORM_BLOCK;
          UNROLL for (int i = 0; i < 6; i++) zin[i] = from_next(Z[i]);
          UNROLL for (int i = 0; i < 3; i++) win[i] = from_next(wp[i]);
        }
        if (step == my_step) {
          double Sk[6], F[6];
          UNROLL for (int i = 0; i < 6; i++) Sk[i] = TF[i] - zin[i];
          UNROLL for (int i = 0; i < 3; i++) up[i] -= win[i];
          ldl3(Sk, F);
          UNROLL for (int i = 0; i < 6; i++) TF[i] = F[i];
          {   // (steps below m: never the root)
            // K = S_k^{-1} Mc and Z = Mc' K through Y = L^{-1} Mc (S_k = L D L'): Z = Y' D^{-1} Y, K = L^{-T} D^{-1} Y
            double Y[9], K[9];
            UNROLL for (int j = 0; j < 3; j++) {
              Y[j] = MK[j];
              Y[3 + j] = MK[3 + j] - F[0] * Y[j];
              Y[6 + j] = MK[6 + j] - F[1] * Y[j] - F[2] * Y[3 + j];
            }
            UNROLL for (int j = 0; j < 3; j++) {
              K[6 + j] = Y[6 + j] * F[5];
              K[3 + j] = Y[3 + j] * F[4];
              K[j] = Y[j] * F[3];
            }
            Z[0] = Y[0] * K[0] + Y[3] * K[3] + Y[6] * K[6];
            Z[1] = Y[0] * K[1] + Y[3] * K[4] + Y[6] * K[7];
            Z[2] = Y[0] * K[2] + Y[3] * K[5] + Y[6] * K[8];
            Z[3] = Y[1] * K[1] + Y[4] * K[4] + Y[7] * K[7];
            Z[4] = Y[1] * K[2] + Y[4] * K[5] + Y[7] * K[8];
            Z[5] = Y[2] * K[2] + Y[5] * K[5] + Y[8] * K[8];
            UNROLL for (int j = 0; j < 3; j++) {
              K[3 + j] -= F[2] * K[6 + j];
              K[j] -= F[0] * K[3 + j] + F[1] * K[6 + j];
            }
            wp[0] = K[0] * up[0] + K[3] * up[1] + K[6] * up[2];
            wp[1] = K[1] * up[0] + K[4] * up[1] + K[7] * up[2];
            wp[2] = K[2] * up[0] + K[5] * up[1] + K[8] * up[2];
            UNROLL for (int i = 0; i < 9; i++) MK[i] = K[i];
          }
        }
      }
      {   // step m, the root: both neighbours (one or two segments: the root is a group's last lane and has no right one)
        double zin[6], win[3];
        UNROLL for (int i = 0; i < 6; i++) { const double p = from_prev(Z[i]); double n = from_next(Z[i]); if constexpr (SMALL_S) n = last ? 0.0 : n; zin[i] = p + n; }
        UNROLL for (int i = 0; i < 3; i++) { const double p = from_prev(wp[i]); double n = from_next(wp[i]); if constexpr (SMALL_S) n = last ? 0.0 : n; win[i] = p + n; }
        if (mid) {
          double Sk[6], F[6];
          UNROLL for (int i = 0; i < 6; i++) Sk[i] = TF[i] - zin[i];
          UNROLL for (int i = 0; i < 3; i++) up[i] -= win[i];
          ldl3(Sk, F);
          UNROLL for (int i = 0; i < 6; i++) TF[i] = F[i];
        }
      }
      SEQ_END();
    }
    // One solve with the factor: u (reduced to the joint states) -> dX by the sweeps -> dc.
    auto forward_u = [&](double (&u)[3]) {
      double w[3] = {0.0, 0.0, 0.0};
      SEQ_BEGIN();
      for (int step = 0; step < m; ++step) {
        double win[3];
        UNROLL for (int i = 0; i < 3; i++) win[i] = from_prev(w[i]);
        if (k > m) { UNIFORM_BLOCK; UNROLL for (int i = 0; i < 3; i++) win[i] = from_next(w[i]); }
        if (step == my_step) {
          UNROLL for (int i = 0; i < 3; i++) u[i] -= win[i];
          w[0] = MK[0] * u[0] + MK[3] * u[1] + MK[6] * u[2];
          w[1] = MK[1] * u[0] + MK[4] * u[1] + MK[7] * u[2];
          w[2] = MK[2] * u[0] + MK[5] * u[1] + MK[8] * u[2];
        }
      }
      {
        double win[3];
        UNROLL for (int i = 0; i < 3; i++) { const double p = from_prev(w[i]); double n = from_next(w[i]); if constexpr (SMALL_S) n = last ? 0.0 : n; win[i] = p + n; }
        if (mid) { UNROLL for (int i = 0; i < 3; i++) u[i] -= win[i]; }
      }
      SEQ_END();
    };
    auto backward_u = [&](const double (&u)[3], double (&dX)[3], double (&dc)[6]) {
      ldl3_solve(TF, u[0], u[1], u[2], dX[0], dX[1], dX[2]);
      SEQ_BEGIN();
      for (int step = m - 1; step >= 0; --step) {
        // towards the root: from the right for the top lanes, from the left under the mask of the root and the bottom lanes.
        // The neighbour towards the root has run the step before (the root needs none), so its dX is final when it is read:
        // no second copy of the finished steps (until round 5 a lane published y = its final dX, 0 before, because both sides were added)
        double xin[3];
        UNROLL for (int i = 0; i < 3; i++) xin[i] = from_next(dX[i]);
        if (k >= m) { UNIFORM_BLOCK; UNROLL for (int i = 0; i < 3; i++) xin[i] = from_prev(dX[i]); }
        if (step == my_step) {   // dX -= K xin, three fused multiply-adds per entry (until round 5 the product first, then the difference: 12 instructions for 9)
          UNROLL for (int i = 0; i < 3; i++)
            dX[i] = __builtin_fma(-MK[3 * i + 2], xin[2], __builtin_fma(-MK[3 * i + 1], xin[1], __builtin_fma(-MK[3 * i], xin[0], dX[i])));
        }
      }
      SEQ_END();
      const NullMap nm = {it, t * 0.05};
      double dXp[3];
      UNROLL for (int i = 0; i < 3; i++) { const double vv = from_prev(dX[i]); dXp[i] = first ? 0.0 : vv; }
      U_apply(nm, dXp, dc[0], dc[1], dc[2]);
      V_apply(nm, dX, dc[3], dc[4], dc[5]);
    };

    LEAN_MARK("C");
    // ---- C. predictor (sigma = 0): statistics of the affine step ----
    // per row: multipliers from the lane's LDS column; slack, residuals and reciprocals recomputed
    // (Round 5, built and REMOVED: the passes from here on working on the MOVED control points -- ds_l = (G (c + dc) - l) -
    //  s_l, one row product where the residual and the step's row product are two: -240 instructions per iteration and a
    //  worse method.  The slack then follows G (c + dc), rounded along a different path than the G c the next iteration
    //  evaluates from the joint states, and the difference -- 1e-11 on a jerk row -- comes back every iteration as a fresh
    //  primal residual the size of the active rows' slacks: +1.5 iterations, a third of the solves end at the round-off
    //  floor, control points 1e-4 off.  As written here the residual is ONE expression of the current control points in
    //  every pass, the step's row product an accurate small number, and the slacks absorb the expression's rounding.)
#define LROW(r, RCP)                                                                                 \
      ROW_SEP_R(r);                                                                                   \
      const double ll = LL(r), lu = LU(r);                                                            \
      const double s_l = sl[SI(r)], s_u = su[SI(r)];                                                  \
      const double gcr = row_dot<r>(c, t);                                                            \
      const double rpl = gcr - s_l - LLO(r), rpu = gcr + s_u - LUP(r);                                \
      RCP_PAIR(RCP, s_l, s_u, isl, isu);
    double dca[6], dX[3];
    double sigma_mu, second_order;
    backward_u(up, dX, dca);
    control_points(X, c);
    {
      double qmin = 0.0, qmax = -1.0, S1 = 0.0, S4 = 0.0;
      PHASE_FENCE(opaque6(c); opaque6(dca); fence_slacks());
      ROW_LIMITS();
      FOR_ROWS(r)
        LROW(r, rcp_fast)
        const double gd = row_dot<r>(dca, t);
        const double dsl = gd + rpl, dsu = -gd - rpu;
        const double ql = dsl * isl, qu = dsu * isu;
        qmin = fmin(qmin, fmin(ql, qu)); qmax = fmax(qmax, fmax(ql, qu));
        const double al = ll * dsl, au = lu * dsu;
        S1 += al + au;
        S4 += al * ql + au * qu;
      END_ROWS
      const Red4 ra = group_reduce_mixed<0, 0, 1, 2>(lds + LN_RED, lane, gbase, k, S, lane_in_group, S1, S4, qmax, qmin);
      const double ap = rcp(fmax(-ra.d, 1.0)), ad = rcp(fmax(1.0 + ra.c, 1.0));
      const double mua = ((1.0 - ad) * rr.a + (ap - ad - ap * ad) * ra.a - ap * ad * ra.b) * inv_m;
      const double sr = mua * rcp(mu);
#ifdef LEAN_TRACE
      tr_sr = sr; tr_dr = fmin(ap, ad);
#endif
      sigma_mu = sr * sr * sr * mu;
      second_order = second_order_factor(ap, ad, plain);   // (weighted by how far the affine step gets; the second chance: btrapz_ipm.h)
    }
    LEAN_MARK("D");
    // ---- D. corrector: rc = s lambda + ds_aff dlambda_aff - sigma mu, dlambda_aff = -lambda (1 + ds_aff / s) ----
#define LROW_CORR(r)                                                                              \
      const double ga = row_dot<r>(dca, t);                                                         \
      const double dsa = ga + rpl, dua = -ga - rpu;                                                 \
      const double rcl = __builtin_fma(second_order, (ll * dsa) * (1.0 + dsa * isl), __builtin_fma(s_l, ll, -sigma_mu)); \
      const double rcu = __builtin_fma(second_order, (lu * dua) * (1.0 + dua * isu), __builtin_fma(s_u, lu, -sigma_mu));
    // The corrected complementarity residuals rc are the same numbers in the corrector's right-hand side (pass D) and in
    // the step ratios (E1).  They are formed in both and kept in neither: a cache through the corrector's solve makes the
    // allocator spill ITERATE state inside the sequential sweeps (round 4: -210 instructions, +108 B of scratch per lane,
    // 4.94 -> 5.33 ms).  What IS kept, from E1 to the update E2 in the 60 registers the factorisation's blocks have just
    // left, is the multipliers' steps dlambda -- E1 forms them for the dual step ratio anyway -- so that the update needs no
    // reciprocal slack, no weight and no target (round 5: -15 v_rcp_f64, ~-110 multiplications per iteration; 4.05 ->
    // 3.97 ms on the scenario_1 batch).  Round 5, later: both passes apply ONE reciprocal slack per side to
    // (rc + lambda rp) and (rc + lambda ds) instead of forming the targets rc / s and the weights lambda / s first, and
    // the predictor's right-hand side (A2) takes s + rp = G c - l directly: -150 instructions per iteration, 3.86 -> 3.78 ms.
    double dc[6];
    {
      double h[6], u[3];
      UNROLL for (int i = 0; i < 6; i++) h[i] = 0.0;
      PHASE_FENCE(opaque6(c); opaque6(dca); fence_slacks());
      {
        ROW_LIMITS();
        FOR_ROWS(r)
          LROW(r, LEAN_RCP)
          LROW_CORR(r)
          // (rc_l + lambda_l rp_l) / s_l - (rc_u - lambda_u rp_u) / s_u
          row_scatter<r>(isl * __builtin_fma(ll, rpl, rcl) - isu * __builtin_fma(-lu, rpu, rcu), t, h);
        END_ROWS
      }
      {
        const NullMap nm = {it, t * 0.05};
        double hn[3];
        VT_apply(nm, h[3], h[4], h[5], u);
        UT_apply(nm, h[0], h[1], h[2], hn);
        UNROLL for (int i = 0; i < 3; i++) { const double v = from_next(hn[i]); u[i] = -((u[i] + (last ? 0.0 : v)) + rd[i]); }
      }
      forward_u(u);
      backward_u(u, dX, dc);
    }
    LEAN_MARK("E1");
    // ---- E. step to the boundary, then the step ----
    {
      double pr = 0.0, dr = 0.0;
      double dll_[NR], dlu_[NR];
      PHASE_FENCE(opaque6(c); opaque6(dca); opaque6(dc); fence_slacks());
      {
        ROW_LIMITS();
        FOR_ROWS(r)
          LROW(r, LEAN_RCP)
          LROW_CORR(r)
          const double gd = row_dot<r>(dc, t);
          const double dsl = gd + rpl, dsu = -gd - rpu;
          const double dll = -isl * __builtin_fma(ll, dsl, rcl), dlu = -isu * __builtin_fma(lu, dsu, rcu);   // dlambda = -(rc + lambda ds) / s
          dll_[SI(r)] = dll; dlu_[SI(r)] = dlu;
          pr = fmax(pr, fmax(-dsl * isl, -dsu * isu));
          const double rll_ = rcp_fast(ll * lu);
          dr = fmax(dr, fmax(-dll * (lu * rll_), -dlu * (ll * rll_)));
        END_ROWS
      }
      const Red4 rs2 = group_reduce2<1, 1>(lds + LN_RED, lane, gbase, k, S, pr, dr);
      const Red4 rs = {0.0, rs2.a, rs2.b, 0.0};
    LEAN_MARK("E2");
      const double m_ = fmax(rs.b, rs.c);
      const double tau = (m_ * ka->tau_thr <= 1.0 && eit - it0 < ka->tau_iters) ? ka->tau : fmin(ka->tau, 0.995);
      const double alpha = fmin(1.0, tau * rcp(fmax(m_, tau)));
      const double alpha_p = alpha, alpha_d = alpha;
      if (!done) tiny_steps = alpha < BTRAPZ_TINY_STEP ? (tiny_steps < 3 ? tiny_steps + 1 : 3) : 0;   // (btrapz_ipm.h)
#ifdef LEAN_TRACE
      if (first && valid && !done && (eit + 1) * 4 <= 6 * ka->seg_stride) {   // 4 values per iteration into the axis's half of the candidate's slot
        double *tdst = ka->ctrl + (size_t)b * 12 * ka->seg_stride + (size_t)axis * 6 * ka->seg_stride + (size_t)eit * 4;
        tdst[0] = tr_score; tdst[1] = tr_mu; tdst[2] = rs.b > rs.c ? alpha_p : -alpha_d;   // (negative: the dual ratio limits the step)
        tdst[3] = tr_pr > tr_res ? -tr_sr : tr_sr;                                     // (negative: the primal residual is the larger one)
      }
#endif
      if (!done && alpha == alpha) {
        UNROLL for (int i = 0; i < 3; i++) X[i] += alpha_p * dX[i];
        PHASE_FENCE(opaque6(c); opaque6(dc); fence_slacks());
        ROW_LIMITS();
        FOR_ROWS(r)
          ROW_SEP_R(r);
          const double ll = LL(r), lu = LU(r);
          const double s_l = sl[SI(r)], s_u = su[SI(r)];
          const double gcr = row_dot<r>(c, t);
          const double rpl = gcr - s_l - LLO(r), rpu = gcr + s_u - LUP(r);
          const double gd = row_dot<r>(dc, t);
          const double dsl = gd + rpl, dsu = -gd - rpu;
          sl[SI(r)] = s_l + alpha_p * dsl; su[SI(r)] = s_u + alpha_p * dsu;
          LL(r) = ll + alpha_d * dll_[SI(r)]; LU(r) = lu + alpha_d * dlu_[SI(r)];
        END_ROWS
      }
    }
#undef LROW
#undef LROW_CORR
    LEAN_MARK("END");
    if (!done) ++eit;
  }

  // ---- write back: control points of the best iterate, objective, status ----
  KA_FENCE();
  if constexpr (CAPPED) {
    if (suspended) {
      state_io(true, susp_slot_);
      if (first) {
        const int cls = 1 + (int)(4.0f * (log10f(susp_score_) + 12.0f));   // how far from convergence: key of the resume lists
        ka->susp_slot[2LL * b + axis] = susp_slot_;
        ka->susp_key[(size_t)axis * ka->B + b] = (ORDERED && !ka->bucket_S) ? S : (cls < 1 ? 1 : cls > 64 ? 64 : cls);
      }
      if (valid && first) { ka->axis_status[2LL * b + axis] = BTRAPZ_SUSPENDED; ka->axis_iters[2LL * b + axis] = iters; ka->axis_obj[2LL * b + axis] = 0.0; }
    }
  }
  if constexpr (WARM) {   // multipliers and joint states of the returned iterate: lam0 / x0 of a later solve of a nearby problem
    if (ka->lam_out && valid) {
      const size_t BS = (size_t)ka->B * ka->seg_stride;
      FOR_ROWS(r)
        ka->lam_out[lam_e + (size_t)r * BS] = LL(r); ka->lam_out[lam_e + (size_t)(18 + r) * BS] = LU(r);
      END_ROWS
      UNROLL for (int r0 = 0; r0 < 3; r0++) {   // the rows this lane does not keep (their bounds live in the previous segment's last rows)
        const int rr_ = r0 == 0 ? 0 : r0 == 1 ? 6 : 11;
        ka->lam_out[lam_e + (size_t)rr_ * BS] = 0.0; ka->lam_out[lam_e + (size_t)(18 + rr_) * BS] = 0.0;
      }
    }
    if (ka->x_out && valid) {
      double *xo = ka->x_out + (((size_t)b * 2 + axis) * ka->seg_stride + k) * 3;
      UNROLL for (int i = 0; i < 3; i++) xo[i] = lds[LN_XB + i][lane];
    }
  }
  {
    double Xb[3], c[6], Pm[21];
    UNROLL for (int i = 0; i < 3; i++) Xb[i] = lds[LN_XB + i][lane];
    control_points(Xb, c);
    LEAN_LOAD_P(Pm)
    double obj = 0.0;
    const double qC = qC0 * t;
    UNROLL for (int i = 0; i < 6; i++) {
      double s = 0.0;
      UNROLL for (int j = 0; j < 6; j++) s += HSYM(Pm, i, j) * c[j];
      const double qi = (qB + qA * (double)(i + 1)) + (i == 0 ? -qC : i == 5 ? qC + qend : 0.0);
      obj += c[i] * (0.5 * s + qi);
    }
    const Red4 ro = group_reduce_mixed<0, -1, -1, -1>(lds + LN_RED, lane, gbase, k, S, lane_in_group, obj, 0.0, 0.0, 0.0);
    if (valid && !(CAPPED && suspended)) {
      double *dst = ka->ctrl + (size_t)b * 12 * ka->seg_stride + (size_t)axis * 6 * S + (size_t)k * 6;
#ifndef LEAN_TRACE
      UNROLL for (int i = 0; i < 6; i++) dst[i] = c[i];
#else
      (void)dst;
#endif
      if (first) {
        int st;
        if (infeasible_bounds) st = BTRAPZ_PRIMAL_INFEASIBLE;
        else if (best_score < 1e-7) st = BTRAPZ_SOLVED;
        else if (best_score < 1e-5 || (res_it < 0 && best_score < 1e-4)) st = BTRAPZ_SOLVED_INACCURATE;
        else st = BTRAPZ_MAX_ITER_REACHED;
        if (st > 0 && !(fabs(ro.a) < 1e300)) st = BTRAPZ_MAX_ITER_REACHED;   // (an objective that is not finite: see the packed form)
        const long long prob = 2LL * b + axis;
        ka->axis_obj[prob] = ro.a;
        ka->axis_status[prob] = st;
        ka->axis_iters[prob] = iters;
      }
    }
  }
#undef KA_FENCE
#undef ROW_SEP
#undef LEAN_RCP
#undef RCP_PAIR
#undef LLO
#undef LUP
#undef ROW_LIMITS
#undef ROW_LIMITS_ACC
#undef LL
#undef LU
#undef HSYM
#undef LEAN_LOAD_P
#undef LEAN_P_PART
}

// 256 registers per lane, 20 KB of LDS per wavefront: two wavefronts per SIMD, eight per CU.
#define LEAN_KERNEL __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
#define LEAN_INSTANCE(name, ...)                                                              \
  LEAN_KERNEL void name(const KernelArgs a, const double *__restrict__ mqm) {                \
    __shared__ double lds[LN_ROWS][64];                                                      \
    lean_solve_body<__VA_ARGS__>(a, mqm, lds, (int)blockIdx.x, (int)threadIdx.x);            \
  }

}  // namespace btrapz
#endif
