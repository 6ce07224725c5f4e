// btrapz_host.hip -- host side of the batched C-ABI (include/btrapz_hip.h): context,
// workspace, kernel launches.  No CPU fallback: without a HIP device every entry point
// fails with BTRAPZ_ENODEVICE.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <cstddef>
#include <type_traits>
#include "btrapz_device.h"
#include "prism_core.h"

using namespace btrapz;

// Environment overrides of btrapz_options for experiments (BTRAPZ_CAP, _QUEUE, _LEAN, _SPLIT, _START, _STALL_FACTOR: A/B
// runs of tools/ without touching the caller).  Only in a -DBTRAPZ_EXPERIMENTS build: in the shipped library this file
// reads no environment variable at all -- the batched entry points and btrapz_launch_single take every choice from
// btrapz_options.  (find_traj, whose only configuration channel IS the environment, translates its documented
// variables -- BTRAPZ_EPS, _ELASTIC, _ELASTIC_TOL, _WARM, _SPLIT, _SPIN, _DEVICE, _INPUT, _OUTPUT_PREFIX, _VERBOSE --
// into options in find_traj.hip; btrapz_multi.hip reads BTRAPZ_RCCL_LIB, a path.)
static inline const char *experiment_env(const char *name) {
#ifdef BTRAPZ_EXPERIMENTS
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

// Step-length rule of the interior-point iterations (btrapz_options).  Measured on MI355X, 65 536 x 20 synthetic
// candidates (mean iterations / kernel ms) and 65 536 jittered copies of c_road_s1_3.txt, a quarter of them infeasible
// and many close to it (candidates the 0.995 rule solves and the setting loses):
//   fraction 0.995 everywhere                     9.47 / 8.06    lost  0   (the classic conservative choice)
//   fraction 0.9999 everywhere                    7.81 / 6.85    lost 91   (blocked steps end too close to the
//                                                                           boundary, centrality is lost, mu stalls)
//   0.9999 when the step to the boundary >= 0.9,
//   0.995 otherwise and after 12 iterations       8.24 / 7.11    lost  0   <- default
// and over 262 144 jittered candidates of all bundled scenarios, both variants: lost 0 (1 lost / 17 gained with the
// jitter raised to 1.5 m).  tests/test_gpu_properties.py keeps that comparison as a regression test.
#define BTRAPZ_DEFAULT_STEP_FRACTION 0.9999
#define BTRAPZ_DEFAULT_STEP_THRESHOLD 0.9
#define BTRAPZ_AGGRESSIVE_ITERATIONS 12
// Divergence / infeasibility test: after BTRAPZ_STALL_START iterations, BTRAPZ_STALL_LENGTH iterations without a better
// score end the solve.  On the same 2 x 262 144 jittered candidates: (12, 8) 0 lost, infeasible candidates end after
// 21.5 iterations on average; (10, 6) 0 lost, 19.5; (8, 5) 3 lost; (8, 4) 18 lost; (6, 3) 87 lost.
#define BTRAPZ_STALL_START 10
#define BTRAPZ_STALL_LENGTH 6
// "Without progress" = no better score AND no residual (the score without its complementarity part) below
// BTRAPZ_STALL_FACTOR times the smallest one so far.  The score alone lost solvable candidates on corridors with runs
// of short segments, where mu climbs for ten iterations while the residuals fall 40-fold (found by a fuzz of find_traj
// against the oracle): 49 of the 48 821 solvable ones of the cuboid bench batch, 1 of 16 384 jittered c6.txt.  Lost
// with factor 0.3 / 0.5 / 0.7..1.0: 17 / 5 / 0; infeasible candidates of the jittered c_road_s1_3.txt end after
// 19.4 / 20.1 / 22.2-24.1 iterations (18.4 with the score alone).
#define BTRAPZ_STALL_FACTOR 0.8f
// An infeasible solve does not just stall: a few iterations after its residuals stop falling mu explodes (1 -> 45 ->
// 500 -> 1e5 -> ... on the jittered c_road_s1_3.txt).  mu above BTRAPZ_DIVERGE_FACTOR times the best score so far ends
// the solve at once.  The largest climb seen on a candidate that then converged is 10x; with 1e4 / 1e3 / 1e2 / 30 no
// solvable candidate of the sets above is lost and the infeasible ones of c_road_s1_3.txt end after 17.2 / 16.9 / 16.0 /
// 15.5 iterations (22.7 without the test).
#define BTRAPZ_DIVERGE_FACTOR 1e3f
// Rescue pass (btrapz_options.elastic): penalty parameter of the relaxed rows and the violation still accepted.
// delta: the relaxed solution is within delta * |multipliers| (1e2..1e4 here) of the least-violation limit; 1e-8 keeps
// that below 1e-4 and the interior-point method still converges in 25-40 iterations (1e-10: 40+, scores near 1e-7).
// tol, in the unit of the penalty (violation / |g_r|): the iterate the reference accepted on src/c7.txt (oracle OSQP
// port, status 2) violates acceleration rows by 0.492 = 0.01004 |g|, velocity rows by 0.013, position rows by 1e-4 m;
// the least-squares solution of that input in the same norm violates them by 0.477 / 0.027 / 3e-4 m (0.0097 |g|; cuboid:
// 0.0019 |g|); a grossly infeasible input such as src/c_road_s1_2.txt is at 4.9 |g|.  Default 0.0125: a quarter above
// what the reference is seen accepting; at most 0.0125 t metres outside a position row.  (Round 2 penalised every row in
// its own unit and accepted 0.5 of it: the least-squares solution then leaves c7's lateral corridor by 0.09 m.)
#define BTRAPZ_DEFAULT_ELASTIC_DELTA 1e-8
#define BTRAPZ_DEFAULT_ELASTIC_TOL 0.0125
// Two-launch solve (btrapz_options.cap_iter): who hands over -- a group that is the only one of its wavefront still
// iterating after cap_iter iterations, and any group still iterating BTRAPZ_CAP_HI iterations later; hand-over slots for
// BTRAPZ_SUSP_PERCENT of the axis problems, at most BTRAPZ_SUSP_BYTES_MAX of workspace.  (Tuned on the bench batches,
// cap_alone 1 / 2 / 3: 6.39 / 6.47 / 6.58 ms; cap_hi 2 / 4 / 8: 6.44 / 6.39 / 6.41; 12-21 % of the problems
// hand over.)  Compile-time constants: a stray environment variable must not change what a drop-in library does.
#ifndef BTRAPZ_CAP_ALONE
#define BTRAPZ_CAP_ALONE 1
#endif
#ifndef BTRAPZ_CAP_HI
#define BTRAPZ_CAP_HI 4
#endif
// BTRAPZ_CAP_SCORE > 0: a lone group whose score is already below it is NOT handed over (quadratic convergence: one or
// two iterations left; half of what scenario_1 hands over and three quarters of the generic batch is below 1e-5).
// Measured, round 4 (65 536 x 20, lean form): it halves the hand-over traffic and costs time -- scenario_1 4.85 ms without,
// 4.94 / 4.99 / 5.01 / 5.05 with 1e-6 / 1e-5 / 1e-4 / 1e-3; generic 3.78 -> 3.90 ... 4.01: a group alone in its wavefront
// runs its last iteration at a third of the machine's width, and that costs more than 74 doubles per segment out and
// back.  Off.
#ifndef BTRAPZ_CAP_SCORE
#define BTRAPZ_CAP_SCORE 0.0
#endif
// (round 5: 25 -> 15 % -- 8.5-12 % of the axis problems of the bench batches hand over; a group that finds no slot goes on
//  to the end where it is, which costs time, never a result)
#ifndef BTRAPZ_SUSP_PERCENT
#define BTRAPZ_SUSP_PERCENT 15
#endif
#define BTRAPZ_SUSP_BYTES_MAX (1ull << 30)

// Layout the lean kernels rely on (btrapz_lean_body.h reads the row limits of its axis as (&sh.acc_s[0])[2 axis + i],
// (&sh.acc_s[0])[4 + 2 axis + i], and its arguments through the kernarg segment pointer: KernelArgs is the first parameter).
static_assert(offsetof(btrapz::Shared, acc_l) == offsetof(btrapz::Shared, acc_s) + 2 * sizeof(double) &&
              offsetof(btrapz::Shared, jerk_s) == offsetof(btrapz::Shared, acc_s) + 4 * sizeof(double) &&
              offsetof(btrapz::Shared, jerk_l) == offsetof(btrapz::Shared, acc_s) + 6 * sizeof(double), "Shared: acc_s, acc_l, jerk_s, jerk_l contiguous");
static_assert(std::is_trivially_copyable<btrapz::KernelArgs>::value && alignof(btrapz::KernelArgs) <= 8, "KernelArgs is passed by value at kernarg offset 0");

struct btrapz_ctx {
  int device = 0;
  std::string err;
  // workspace (grown on demand)
  double *d_axis_obj = nullptr;
  int *d_axis_status = nullptr, *d_axis_iters = nullptr;
  double *d_axis_viol = nullptr;    // [2B][4] rescue pass: row violations per class
  size_t viol_valid = 0;            // candidates of the last solve with a rescue pass (0: none)
  size_t axis_cap = 0;
  double *d_mqm = nullptr;          // [2][4][21]
  double h_mqm_w[8] = {NAN, NAN, NAN, NAN, NAN, NAN, NAN, NAN};  // weights the table was built for
  // ragged batches: candidate order by segment count + bucket tables
  int *d_order = nullptr; size_t order_cap = 0;
  int *d_meta = nullptr;            // [198] histogram/cand_prefix, wave_prefix, cursors
  int *d_retry = nullptr; size_t retry_cap = 0;   // corridor stage: [0] count, [1..] candidates of the retry pass
  void *d_strips = nullptr; size_t strip_cap = 0; // btrapz_prism_corridor_batch_device's two-launch path: the strips
  void *d_corr_ws = nullptr; size_t corr_ws_cap = 0;   // corridor_serial_kernel's segment lists (horizons beyond the wave-wide kernels)
  int *d_long_list = nullptr; size_t long_list_cap = 0;   // ragged batches: the candidates of more than 64 segments, by count
  // rescue pass (btrapz_options.elastic): keys [2][B], per-axis candidate lists [2][B], bucket tables [2][198]
  // The workspaces above serve one launch sequence at a time.  Launches of one context issued on DIFFERENT streams are
  // ordered behind each other with this event (recorded after every sequence, waited for when the stream changes).
  hipEvent_t ws_free = nullptr; hipStream_t ws_stream = nullptr; bool ws_used = false;
  double *d_argmin_cost = nullptr; long long *d_argmin_idx = nullptr; size_t argmin_cap = 0;   // partial arg-mins
  int *d_rescue = nullptr; size_t rescue_cap = 0;
  int *d_rescue_meta = nullptr;
  // staging for the host-pointer wrapper
  double *d_stage = nullptr; size_t stage_cap = 0;
  double *d_single = nullptr;       // control points of the single-candidate launch (find_traj)
  // what a warm single-candidate launch leaves for the next one: joint states [2][64][3], multipliers [2][36][64] (two
  // sets, read / written alternately), and the problem shape they belong to
  double *d_single_warm = nullptr; int single_S = 0, single_variant = -1, single_flip = 0;
  int *d_queue = nullptr;           // [2] candidate counters of the persistent launch (ipm_solve_queue_kernel)
  // capped solve (btrapz_options.cap_iter): iterates of the suspended axis problems, their slots and list keys
  double *d_susp_state = nullptr; size_t susp_state_doubles = 0;
  int *d_susp_ints = nullptr; size_t susp_ints = 0;      // [count, 3 pad] [bucket tables 2 x 198] [2B keys] [2B slots]
  int resident_waves = 1024;        // wavefronts the device holds at one per SIMD
  int last_form = -1;               // btrapz_last_solve_form
  int *d_istage = nullptr; size_t istage_cap = 0;
};

#define HIPCHK(ctx, call)                                                                     \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                         \
      return BTRAPZ_EHIP;                                                                     \
    }                                                                                         \
  } while (0)

int btrapz_ctx_device(const btrapz_ctx *c) { return c ? c->device : 0; }

BTRAPZ_EXPORT int btrapz_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

BTRAPZ_EXPORT int btrapz_build_has_experiments(void) {
#ifdef BTRAPZ_EXPERIMENTS
  return 1;
#else
  return 0;
#endif
}

BTRAPZ_EXPORT int btrapz_create(btrapz_ctx **out, int device) {
  if (!out) return BTRAPZ_EINVAL;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return BTRAPZ_ENODEVICE;
  if (hipSetDevice(device) != hipSuccess) return BTRAPZ_ENODEVICE;
  btrapz_ctx *c = new btrapz_ctx();
  c->device = device;
  if (hipMalloc(&c->d_mqm, sizeof(double) * 168) != hipSuccess) { delete c; return BTRAPZ_ENOMEM; }
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->resident_waves = 4 * cus;
  }
  if (hipMalloc(&c->d_queue, sizeof(int) * 2) != hipSuccess) { (void)hipFree(c->d_mqm); delete c; return BTRAPZ_ENOMEM; }
  if (hipEventCreateWithFlags(&c->ws_free, hipEventDisableTiming) != hipSuccess) { (void)hipFree(c->d_mqm); (void)hipFree(c->d_queue); delete c; return BTRAPZ_ENOMEM; }
  *out = c;
  return BTRAPZ_OK;
}

BTRAPZ_EXPORT int btrapz_destroy(btrapz_ctx *c) {
  if (!c) return BTRAPZ_EINVAL;
  (void)hipSetDevice(c->device);
  (void)hipFree(c->d_axis_obj); (void)hipFree(c->d_axis_status); (void)hipFree(c->d_axis_iters); (void)hipFree(c->d_axis_viol);
  (void)hipFree(c->d_mqm); (void)hipFree(c->d_stage); (void)hipFree(c->d_istage); (void)hipFree(c->d_single);
  (void)hipFree(c->d_queue); (void)hipFree(c->d_single_warm); (void)hipFree(c->d_susp_state); (void)hipFree(c->d_susp_ints);
  (void)hipFree(c->d_order); (void)hipFree(c->d_meta); (void)hipFree(c->d_retry); (void)hipFree(c->d_strips); (void)hipFree(c->d_corr_ws); (void)hipFree(c->d_long_list);
  (void)hipFree(c->d_rescue); (void)hipFree(c->d_rescue_meta); (void)hipFree(c->d_argmin_cost); (void)hipFree(c->d_argmin_idx);
  if (c->ws_free) (void)hipEventDestroy(c->ws_free);
  delete c;
  return BTRAPZ_OK;
}

BTRAPZ_EXPORT const char *btrapz_last_error(const btrapz_ctx *c) { return c ? c->err.c_str() : "null context"; }
// Device memory the context holds for its launches right now (it grows on demand and is kept): per-axis records, bucket
// tables and lists, the hand-over workspace of the two-launch solve, staging of the host-pointer wrapper.
BTRAPZ_EXPORT long long btrapz_workspace_bytes(const btrapz_ctx *c) {
  if (!c) return 0;
  size_t n = 0;
  n += c->axis_cap * (sizeof(double) * 5 + sizeof(int) * 2);
  n += sizeof(double) * 168 + sizeof(int) * 2;
  n += c->order_cap * sizeof(int) + (c->d_meta ? 198 * sizeof(int) : 0) + c->retry_cap * sizeof(int) + c->strip_cap;
  n += c->argmin_cap * (sizeof(double) + sizeof(long long));
  n += c->rescue_cap * 2 * sizeof(int) + (c->d_rescue_meta ? 2 * 198 * sizeof(int) : 0);
  n += c->stage_cap * sizeof(double) + c->istage_cap * sizeof(int);
  n += (c->d_single ? sizeof(double) * 12 * BTRAPZ_MAX_SEGMENTS : 0) + (c->d_single_warm ? sizeof(double) * 2 * (2 * 64 * 3 + 2 * 36 * 64) : 0);
  n += c->susp_state_doubles * sizeof(double) + c->susp_ints * sizeof(int);
  return (long long)n;
}
BTRAPZ_EXPORT int btrapz_last_solve_form(const btrapz_ctx *c) { return c ? c->last_form : -1; }

static int ensure_axis_ws(btrapz_ctx *c, size_t nprob) {
  if (nprob <= c->axis_cap) return BTRAPZ_OK;
  (void)hipFree(c->d_axis_obj); (void)hipFree(c->d_axis_status); (void)hipFree(c->d_axis_iters); (void)hipFree(c->d_axis_viol);
  c->d_axis_obj = nullptr; c->d_axis_status = nullptr; c->d_axis_iters = nullptr; c->d_axis_viol = nullptr; c->axis_cap = 0;
  c->viol_valid = 0;
  HIPCHK(c, hipMalloc(&c->d_axis_obj, sizeof(double) * nprob));
  HIPCHK(c, hipMalloc(&c->d_axis_viol, sizeof(double) * 4 * nprob));
  HIPCHK(c, hipMalloc(&c->d_axis_status, sizeof(int) * nprob));
  HIPCHK(c, hipMalloc(&c->d_axis_iters, sizeof(int) * nprob));
  c->axis_cap = nprob;
  return BTRAPZ_OK;
}

BTRAPZ_EXPORT void btrapz_options_init(btrapz_options *opt) {
  if (!opt) return;
  memset(opt, 0, sizeof(*opt));
  opt->struct_size = (int)sizeof(btrapz_options);
}
// A caller built against another layout of btrapz_options would make the library read fields that are not there.
static bool options_ok(btrapz_ctx *c, const btrapz_options *opt) {
  if (!opt || opt->struct_size == (int)sizeof(btrapz_options)) return true;
  if (c) c->err = "invalid argument: btrapz_options.struct_size (call btrapz_options_init first)";
  return false;
}

// Weights, limits and iteration parameters of a launch (everything of KernelArgs that is not a pointer or a size).
static void fill_parameters(KernelArgs &a, const btrapz_shared *sh, const btrapz_options *opt, const btrapz_warm *warm) {
  memcpy(a.sh.w_s, sh->w_s, sizeof(a.sh.w_s)); memcpy(a.sh.w_l, sh->w_l, sizeof(a.sh.w_l));
  a.sh.weight_end_s = sh->weight_end_s; a.sh.weight_end_l = sh->weight_end_l;
  a.sh.ds_ref = sh->ds_ref; a.sh.dl_ref = sh->dl_ref;
  // a header limit left at the reference's default of +-1e10 (piecewise_jerk_problem.cc:9,25-35) is no limit: moved to
  // +-BTRAPZ_FAR_LIMIT, where the kernels know it for one (an infinite or NaN limit stays a defect of the input)
  auto limit = [](double v) { const double m = fabs(v); return (m >= BTRAPZ_FAR && m <= 1.7e308) ? copysign(BTRAPZ_FAR_LIMIT, v) : v; };
  a.sh.acc_s[0] = fmax(sh->dds[0], -1000.0); a.sh.acc_s[1] = fmin(sh->dds[1], 1000.0);  // solve_3d.cc:836,843-844
  a.sh.acc_l[0] = limit(sh->ddl[0]); a.sh.acc_l[1] = limit(sh->ddl[1]);
  a.sh.jerk_s[0] = limit(sh->ddds[0]); a.sh.jerk_s[1] = limit(sh->ddds[1]);
  a.sh.jerk_l[0] = limit(sh->dddl[0]); a.sh.jerk_l[1] = limit(sh->dddl[1]);
  a.sh.variant = sh->variant;
  a.eps = (opt && opt->eps > 0) ? opt->eps : 1e-9;
  a.max_iter = (opt && opt->max_iter > 0) ? opt->max_iter : 60;
  a.tau = (opt && opt->step_fraction > 0 && opt->step_fraction < 1) ? opt->step_fraction : BTRAPZ_DEFAULT_STEP_FRACTION;
  a.tau_iters = BTRAPZ_AGGRESSIVE_ITERATIONS;
  a.stall_start = BTRAPZ_STALL_START; a.stall_len = BTRAPZ_STALL_LENGTH;
  static const float stall_factor_env = [] { const char *v = experiment_env("BTRAPZ_STALL_FACTOR"); return v ? (float)atof(v) : 0.0f; }();   // (experiments)
  a.stall_factor = stall_factor_env > 0.0f ? stall_factor_env : BTRAPZ_STALL_FACTOR;
  a.diverge_factor = BTRAPZ_DIVERGE_FACTOR;
  a.tau_thr = (opt && opt->step_threshold > 0) ? opt->step_threshold : BTRAPZ_DEFAULT_STEP_THRESHOLD;
  a.x0 = warm ? warm->x0 : nullptr; a.lam0 = warm ? warm->lam0 : nullptr; a.lam_out = warm ? warm->lam_out : nullptr;
  a.mu0 = (warm && warm->mu0 > 0) ? warm->mu0 : 1e-4;
  a.smin = (warm && warm->smin > 0) ? warm->smin : 1e-2;
  a.elastic_delta = (opt && opt->elastic_delta > 0) ? opt->elastic_delta : BTRAPZ_DEFAULT_ELASTIC_DELTA;
  a.elastic_tol = (opt && opt->elastic_tol > 0) ? opt->elastic_tol : BTRAPZ_DEFAULT_ELASTIC_TOL;
  a.bucket_S = 0;
  a.cap_iter = 0; a.cap_alone = 0; a.cap_hi = 0; a.cap_score = 0.0; a.susp_cap = 0; a.susp_state = nullptr; a.susp_count = nullptr; a.susp_slot = nullptr; a.susp_key = nullptr;
#ifdef BTRAPZ_EXPERIMENTS
  static const int start_env = [] { const char *e = experiment_env("BTRAPZ_START"); return e ? atoi(e) : -1; }();
  a.unc_start = start_env >= 0 ? start_env : (opt ? opt->start : 0);
#else
  a.unc_start = 0;   // (btrapz_options.start: the optimum does not depend on it, the iteration count does -- for the worse; experiment builds only)
#endif
}

// M' pQp_d M on the host (solve_3d.cc:87-143): the single-candidate path hands the table over with its inputs.
void btrapz_mqm_table_host(const btrapz_shared *sh, double *table) {
#pragma clang fp contract(off)   // the device builder (mqm_table_kernel) evaluates the same expressions: same bits
  static const double M[6][6] = {{1, 0, 0, 0, 0, 0},      {-5, 5, 0, 0, 0, 0},      {10, -20, 10, 0, 0, 0},
                                 {-10, 30, -30, 10, 0, 0}, {5, -20, 30, -20, 5, 0}, {-1, 5, -10, 10, -5, 1}};
  for (int axis = 0; axis < 2; axis++) {
    const double *w = axis == 0 ? sh->w_s : sh->w_l;
    for (int d = 0; d < 4; d++)
      for (int j = 0; j < 6; j++)
        for (int i = 0; i <= j; i++) {
          double acc = 0.0;
          for (int a = d; a < 6; a++) {
            double t = 0.0;
            for (int b = d; b < 6; b++) {
              double num = w[d];
              for (int r = 0; r < d; r++) num *= double((a - r) * (b - r));
              t += num / double(a + b - 2 * d + 1) * M[b][j];
            }
            acc += M[a][i] * t;
          }
          table[(axis * 4 + d) * 21 + j * (j + 1) / 2 + i] = acc;
        }
  }
}

void btrapz_single_forget(btrapz_ctx *c) { if (c) c->single_S = 0; }

// Test hook: the M'QM table as find_traj's single launch receives it (host builder) and as the batched entry points
// use it (device builder), [2][4][21] each -- the two must agree bit for bit (tests/test_gpu_properties.py).
BTRAPZ_EXPORT int btrapz_debug_mqm_tables(btrapz_ctx *c, const btrapz_shared *sh, double *host_table, double *device_table) {
  if (!c || !sh || !host_table || !device_table) return BTRAPZ_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  btrapz_mqm_table_host(sh, host_table);
  double *d = nullptr;
  HIPCHK(c, hipMalloc(&d, sizeof(double) * 168));
  MqmWeights mw;
  memcpy(mw.w[0], sh->w_s, sizeof(double) * 4); memcpy(mw.w[1], sh->w_l, sizeof(double) * 4);
  hipLaunchKernelGGL(mqm_table_kernel, dim3(1), dim3(192), 0, nullptr, mw, d);
  hipError_t e = hipMemcpy(device_table, d, sizeof(double) * 168, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  HIPCHK(c, e);
  return BTRAPZ_OK;
}

// Test / analysis hook: iterations and status of the 2 B axis problems of the context's last batched solve
// ([b][axis], host arrays; synchronises the device).  tools/resume_model.py reads them.
BTRAPZ_EXPORT int btrapz_debug_axis_records(btrapz_ctx *c, int B, int *iters, int *status) {
  if (!c || B < 1 || (size_t)2 * B > c->axis_cap || !iters || !status) return BTRAPZ_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipDeviceSynchronize());
  HIPCHK(c, hipMemcpy(iters, c->d_axis_iters, sizeof(int) * 2 * B, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(status, c->d_axis_status, sizeof(int) * 2 * B, hipMemcpyDeviceToHost));
  return BTRAPZ_OK;
}
// ... and the keys [2][B] the last capped solve's resume launch was bucketed by (0: the axis problem was not handed over)
BTRAPZ_EXPORT int btrapz_debug_resume_keys(btrapz_ctx *c, int B, int *keys) {
  if (!c || B < 1 || !keys || c->susp_ints < 400 + 4 * (size_t)B) return BTRAPZ_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipDeviceSynchronize());
  HIPCHK(c, hipMemcpy(keys, c->d_susp_ints + 400, sizeof(int) * 2 * B, hipMemcpyDeviceToHost));
  return BTRAPZ_OK;
}

int btrapz_launch_single(btrapz_ctx *c, const btrapz_shared *sh, const btrapz_options *opt, int S, const double *in,
                         double *out, int max_points, int warm, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!c || !sh || S < 1 || S > BTRAPZ_MAX_SEGMENTS || !in || !out || max_points < 1 || !options_ok(c, opt)) return BTRAPZ_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  KernelArgs a;
  fill_parameters(a, sh, opt, nullptr);
  a.B = 1; a.S = S; a.seg_stride = S; a.order = nullptr; a.seg_count = nullptr; a.cand_prefix = nullptr; a.wave_prefix = nullptr;
  const double *mqm = in + (size_t)BTRAPZ_NUM_SEG_FIELDS * S + 18;
  a.seg = in; a.init = in + (size_t)BTRAPZ_NUM_SEG_FIELDS * S; a.ref_end = a.init + 6; a.dl_bounds = a.ref_end + 2; a.mqm = mqm;
  // what the kernel reads back stays in device memory: the context's per-axis records and a control-point block
  int rc = ensure_axis_ws(c, 2);
  if (rc != BTRAPZ_OK) return rc;
  if (!c->d_single) HIPCHK(c, hipMalloc(&c->d_single, sizeof(double) * 12 * BTRAPZ_MAX_SEGMENTS));
  if (c->ws_used && c->ws_stream != stream) HIPCHK(c, hipStreamWaitEvent(stream, c->ws_free, 0));
  a.axis_obj = c->d_axis_obj; a.axis_status = c->d_axis_status; a.axis_iters = c->d_axis_iters;
  a.ctrl = c->d_single; a.queue = nullptr; a.x_out = nullptr; a.axis_viol = nullptr;
  // at most 21 segments: the split form (rows of a segment over three lanes), ~0.8 of the time per iteration
  const bool split_form = S <= 21 && !(opt && opt->split < 0);   // (find_traj maps BTRAPZ_SPLIT onto opt->split)
  if (warm) {
    const size_t nx = 2 * 64 * 3, nl = 2 * 36 * 64, set = nx + nl;
    if (!c->d_single_warm) HIPCHK(c, hipMalloc(&c->d_single_warm, sizeof(double) * 2 * set));
    double *rd = c->d_single_warm + (size_t)c->single_flip * set, *wr = c->d_single_warm + (size_t)(1 - c->single_flip) * set;
    const bool have = c->single_S == S && c->single_variant == sh->variant;
    // layouts for B = 1, seg_stride = S: x [2][S][3], lam [2][36][1][S]
    a.x0 = have ? rd : nullptr; a.lam0 = have ? rd + nx : nullptr;
    a.x_out = wr; a.lam_out = wr + nx;
    c->single_flip = 1 - c->single_flip; c->single_S = S; c->single_variant = sh->variant;
    if (split_form)
      hipLaunchKernelGGL(single_candidate_warm_split_kernel, dim3(1), dim3(128), 0, stream, a, mqm, sh->delta, max_points, out);
    else
      hipLaunchKernelGGL(single_candidate_warm_kernel, dim3(1), dim3(128), 0, stream, a, mqm, sh->delta, max_points, out);
  } else {
    if (split_form)
      hipLaunchKernelGGL(single_candidate_split_kernel, dim3(1), dim3(128), 0, stream, a, mqm, sh->delta, max_points, out);
    else
      hipLaunchKernelGGL(single_candidate_kernel, dim3(1), dim3(128), 0, stream, a, mqm, sh->delta, max_points, out);
  }
  c->ws_stream = stream; c->ws_used = true;
  HIPCHK(c, hipEventRecord(c->ws_free, stream));
  HIPCHK(c, hipGetLastError());
  return BTRAPZ_OK;
}

// Uniform batches: seg_count == nullptr, S segments each.  Ragged: S is the slot stride, seg_count[b] the use.
static int solve_common(btrapz_ctx *c, const btrapz_shared *sh, const btrapz_options *opt, const btrapz_warm *warm,
                        int B, int S, const int *seg_count, const double *seg, const double *init, const double *ref_end,
                        const double *dl_bounds, double *ctrl, double *cost, int *status, int *iters, void *stream_) {
  if (!c) return BTRAPZ_EINVAL;
  if (!sh || B < 1 || S < 1 || (!seg_count && S > BTRAPZ_MAX_SEGMENTS_LONG) || !seg || !init || !ref_end || !dl_bounds ||
      !ctrl || !cost || !status) {
    c->err = "invalid argument";
    return BTRAPZ_EINVAL;
  }
  if (!options_ok(c, opt)) return BTRAPZ_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->ws_used && stream != c->ws_stream) HIPCHK(c, hipStreamWaitEvent(stream, c->ws_free, 0));
  int rc = ensure_axis_ws(c, 2 * (size_t)B);
  if (rc != BTRAPZ_OK) return rc;
  // batch-invariant M'QM table: rebuilt only when the weights change
  double wkey[8];
  memcpy(wkey, sh->w_s, sizeof(double) * 4); memcpy(wkey + 4, sh->w_l, sizeof(double) * 4);
  if (memcmp(wkey, c->h_mqm_w, sizeof(wkey)) != 0) {
    // built on the device, in stream order: no host buffer to keep alive, no synchronisation, and a launch that is
    // still reading the previous table (same stream, or ordered by the event above) finishes first
    MqmWeights mw;
    memcpy(mw.w[0], sh->w_s, sizeof(double) * 4); memcpy(mw.w[1], sh->w_l, sizeof(double) * 4);
    hipLaunchKernelGGL(mqm_table_kernel, dim3(1), dim3(192), 0, stream, mw, c->d_mqm);
    HIPCHK(c, hipGetLastError());
    memcpy(c->h_mqm_w, wkey, sizeof(wkey));
  }
  KernelArgs a;
  fill_parameters(a, sh, opt, warm);
  a.B = B; a.S = S; a.seg_stride = S; a.order = nullptr; a.seg_count = nullptr; a.cand_prefix = nullptr; a.wave_prefix = nullptr;
  a.seg = seg; a.init = init; a.ref_end = ref_end; a.dl_bounds = dl_bounds; a.mqm = c->d_mqm;
  a.ctrl = ctrl; a.axis_obj = c->d_axis_obj; a.axis_status = c->d_axis_status; a.axis_iters = c->d_axis_iters;
  a.axis_viol = c->d_axis_viol; c->viol_valid = 0;
  a.x_out = nullptr;
  const int elastic = opt ? opt->elastic : 0;
  if (elastic < 0 || elastic > 2) { c->err = "invalid argument: btrapz_options.elastic"; return BTRAPZ_EINVAL; }
#ifndef BTRAPZ_EXPERIMENTS
  // the two experimental schedules exist in -DBTRAPZ_EXPERIMENTS builds only: asking this build for one is an error, not a
  // silent default (btrapz_build_has_experiments() tells a caller which build it holds)
  if (opt && (opt->queue > 0 || opt->start != 0)) {
    c->err = "invalid argument: btrapz_options.queue / .start are honoured by -DBTRAPZ_EXPERIMENTS builds only (this build has neither kernel)";
    return BTRAPZ_EINVAL;
  }
#endif
  unsigned blocks;
  const int *hint = (warm && !seg_count) ? warm->hint : nullptr;   // uniform batches only
  // Candidates that cannot start (btrapz_options.compact): a pre-pass lists the live ones, the launches below take them
  // through a.order.  Cold solves without a rescue pass only (the rescue pass solves what stalled, axis by axis: it needs
  // every axis's record); automatic for large ragged batches and large batches of the cuboid variant.
  const bool warm_args = a.x0 || a.lam0 || a.lam_out;
  const int compact_opt = opt ? opt->compact : 0;
  const unsigned waves_needed = 2u * (unsigned)((size_t)B / (size_t)(64 / (S < 64 ? S : 64)) + 1);
  const bool compact = elastic == 0 && !warm_args && !hint && S <= BTRAPZ_MAX_SEGMENTS && a.unc_start == 0 &&
                       (compact_opt > 0 || (compact_opt == 0 && (seg_count || sh->variant == BTRAPZ_CUBOID) && waves_needed >= 3u * (unsigned)c->resident_waves));
  int *compact_keys = nullptr;
  if (compact) {
    if (2 * (size_t)B > c->rescue_cap) {   // (keys [B] at the front of the rescue pass's lists: no rescue pass in this solve)
      (void)hipFree(c->d_rescue); c->d_rescue = nullptr; c->rescue_cap = 0;
      HIPCHK(c, hipMalloc(&c->d_rescue, sizeof(int) * 4 * (size_t)B));
      c->rescue_cap = 2 * (size_t)B;
    }
    compact_keys = c->d_rescue;
    KernelArgs pa = a;
    pa.seg_stride = S;
    const int pre_gpw = 64 / S;   // (S: the slot stride, at most 64 here)
    hipLaunchKernelGGL(prestart_kernel, dim3((unsigned)((B + pre_gpw - 1) / pre_gpw)), dim3(64), 0, stream, pa, seg_count ? 0 : S, seg_count, compact_keys);
    HIPCHK(c, hipGetLastError());
  }
  if (seg_count || hint || compact) {
    if ((size_t)B > c->order_cap) {
      (void)hipFree(c->d_order); c->d_order = nullptr; c->order_cap = 0;
      HIPCHK(c, hipMalloc(&c->d_order, sizeof(int) * (size_t)B));
      c->order_cap = B;
    }
    if (!c->d_meta) HIPCHK(c, hipMalloc(&c->d_meta, sizeof(int) * 198));
    HIPCHK(c, hipMemsetAsync(c->d_meta, 0, sizeof(int) * 198, stream));
    const unsigned nb = (unsigned)((B + 255) / 256);
    // keys: segment counts (ragged), hint classes (uniform, fixed_S > 0), or -- compact -- the pre-pass's list keys: the
    // count / 1 of a live candidate, 0 for one that is not listed (uniform: fixed_S < 0, "a key <= 0 is not listed"); the
    // pre-pass has written the records of what it dropped, so the scatter kernel writes none
    const int fixed_S = seg_count ? 0 : (compact ? -S : S);
    const int *keys = compact ? compact_keys : seg_count ? seg_count : hint;
    hipLaunchKernelGGL(bucket_hist_kernel, dim3(nb), dim3(256), 0, stream, B, S, keys, c->d_meta, fixed_S);
    hipLaunchKernelGGL(bucket_prefix_kernel, dim3(1), dim3(64), 0, stream, c->d_meta, fixed_S);
    hipLaunchKernelGGL(bucket_scatter_kernel, dim3(nb), dim3(256), 0, stream, B, S, keys, c->d_meta, c->d_order,
                       compact ? (double *)nullptr : c->d_axis_obj, compact ? (int *)nullptr : c->d_axis_status,
                       compact ? (int *)nullptr : c->d_axis_iters, fixed_S);
    HIPCHK(c, hipGetLastError());
    a.order = c->d_order; a.seg_count = seg_count; a.cand_prefix = c->d_meta; a.wave_prefix = c->d_meta + 66;
    a.bucket_S = seg_count ? 0 : S;
    // upper bound on wavefront pairs without a host round trip: every bucket wastes less than one pair
    const int gpw_min = seg_count ? 1 : 64 / S;
    blocks = 2u * (unsigned)(B / gpw_min + 65);
  } else {
    const int gpw = S <= 64 ? 64 / S : 1;  // axis problems per wavefront; wave w: axis w&1 of candidates (w>>1)*gpw + [0,gpw)
    blocks = 2u * (unsigned)((B + gpw - 1) / gpw);
  }
  const bool warm_kernel = a.x0 || a.lam0 || a.lam_out;   // (a hint alone only reorders the candidates)
  const bool long_form = !seg_count && S > BTRAPZ_MAX_SEGMENTS;   // one axis problem per workgroup (ipm_solve_long_kernel)
  if (long_form && (a.order || warm_kernel || elastic == 2 || (elastic == 1 && S > BTRAPZ_MAX_SEGMENTS_LONG_RESCUE))) {
    c->err = "more than 64 segments: uniform cold solve only, rescue pass (elastic = 1) up to 192 segments";
    return BTRAPZ_EINVAL;
  }
  auto kernel = warm_kernel ? (a.order ? ipm_solve_warm_ordered_kernel : ipm_solve_warm_kernel)
                            : (a.order ? ipm_solve_ordered_kernel : ipm_solve_kernel);
  if (elastic != 2) {
    // Uniform cold batches with several candidates per resident wavefront slot may run on persistent wavefronts that
    // draw candidates from a queue (btrapz_options.queue / BTRAPZ_QUEUE=1): a wavefront's groups then do not wait for
    // its slowest one, at the price of ~half an iteration per hand-over for all of its groups.  Measured (65 536
    // candidates, kernel ms off -> on): scenario_1 x 20 6.93 -> 6.59, its cuboid variant (16 % stalling) 6.57 -> 5.71,
    // generic x 20 5.12 -> 5.33, scenario_1 x 10 2.13 -> 2.34: it pays where iteration counts spread widely, so it is
    // the caller's choice and off by default.
#ifdef BTRAPZ_EXPERIMENTS
    static const int queue_env = [] { const char *q = experiment_env("BTRAPZ_QUEUE"); return q ? (*q == '0' ? -1 : 1) : 0; }();
    const bool queue_on = queue_env ? queue_env > 0 : (opt && opt->queue > 0);
#else
    const bool queue_on = false;   // (btrapz_options.queue: scheduling only, a measured loss -- honoured by experiment builds, ignored here)
#endif
    a.queue = c->d_queue;
    // Few candidates: one per wavefront, rows over three lanes (btrapz_options.split; ipm_solve_split_kernel)
    const char *split_q = experiment_env("BTRAPZ_SPLIT");
    const int split_env = split_q ? (*split_q == '0' ? -1 : 1) : 0;
    const int split_opt = split_env ? split_env : (opt ? opt->split : 0);
    const bool split_on = !a.order && !warm_kernel && S <= 21 &&
                          (split_opt > 0 || (split_opt == 0 && 2u * (unsigned)B <= (unsigned)c->resident_waves));
    // Two launches for uniform cold batches much larger than the machine (btrapz_options.cap_iter / BTRAPZ_CAP): every
    // group stops at cap_iter iterations, the unfinished ones are carried on by a second launch, like with like, the
    // far-from-converged first (CAPPED / RESUME in btrapz_kernels.hip).  Same iterates, same results.
    static const int cap_env = [] { const char *q = experiment_env("BTRAPZ_CAP"); return q ? atoi(q) : -1; }();
    // Measured (tools/cap_bench.py, 65 536 candidates, one launch -> two): scenario_1 x 20 7.00 -> 6.39 ms, its cuboid
    // variant 6.70 -> 6.35, generic x 20 5.21 -> 5.18, scenario_1 x 10 2.24 -> 2.46 (six groups per wavefront: what a
    // lone straggler wastes is less than what the second launch costs).  Automatic (cap_iter = 0): 6 for uniform cold
    // batches of 16 to 64 segments that fill the device at least eight times over; -1: never.
    // Ragged batches (round 3, late): the same two launches on request (cap_iter > 0), the resume lists bucketed by
    // segment count instead of by convergence class.  Not automatic: measured on 65 536 candidates, knots -> control
    // points (tools/pipeline_bench.py, one launch -> cap 8): jittered c_road_s1_3.txt, 8 segments, a quarter of the
    // candidates infeasible (16-19 iterations against 9-12) 3.20 -> 2.90 ms; scenario_1 at knot level, 18-24 segments
    // 9.56 -> 9.79; two-car scenes, 7-9 segments, all feasible 3.21 -> 3.54 -- it pays where iteration counts spread
    // widely, which the library does not know.
    int cap_iter = cap_env > 0 ? cap_env : cap_env == 0 ? -1 : (opt ? opt->cap_iter : 0);   // BTRAPZ_CAP=0: never
    // Two wavefronts per SIMD (btrapz_options.lean / BTRAPZ_LEAN; btrapz_lean.hip): cold solves of at most 64 segments
    static const int lean_env = [] { const char *q = experiment_env("BTRAPZ_LEAN"); return q ? (*q == '0' ? -1 : 1) : 0; }();
    const int lean_opt = lean_env ? lean_env : (opt ? opt->lean : 0);
    const bool lean_ok = !long_form && !split_on && !queue_on && (S >= 3 || seg_count) && S <= BTRAPZ_MAX_SEGMENTS && a.unc_start == 0;   // (uniform batches of one or two segments: the packed form -- only the lean kernels of ragged batches carry the end-lane fix-up)
    // Automatic: batches that give every SIMD its two wavefronts several times over.  Measured (tools/lean_bench.py,
    // scenario_1 x 20, packed -> lean, one launch): 512 candidates 0.375 -> 0.435 ms, 2 048 0.447 -> 0.517, 8 192 1.117 ->
    // 1.032, 16 384 1.964 -> 1.733, 65 536 7.05 -> 5.61: a lone wavefront per SIMD runs the packed form's shorter
    // instruction stream faster; from about three wavefronts per SIMD on the second resident one pays.
    const unsigned est_waves = 2u * (unsigned)((size_t)B / (size_t)(64 / (S < 64 ? S : 64)) + 1);
    // (round 5, re-measured with the final kernels, packed -> lean in one launch, ms: scenario_1 x 20 -- 512 candidates
    //  0.30 -> 0.34, 2 048 0.40 -> 0.38, 4 096 0.63 -> 0.52, 8 192 0.91 -> 0.78; generic x 10 -- 2 048 0.14 -> 0.17, 4 096
    //  0.22 -> 0.19, 8 192 0.33 -> 0.29: the crossover sits at about 1.25 wavefronts per SIMD, not at 3)
    const bool lean_on = lean_ok && (lean_opt > 0 || (lean_opt == 0 && 4u * est_waves >= 5u * (unsigned)c->resident_waves));
    const bool ragged = seg_count != nullptr;
    const bool cap_requested = cap_iter > 0;   // (by the caller; the automatic choice below falls back to one launch when the workspace cannot be had)
    // (S <= 32: a wavefront that holds ONE group has nobody to wait for -- with 33..64 segments every group would be
    //  "alone" at once and a quarter of the batch be written out and read back for nothing: ADVICE r4)
    if (cap_iter == 0 && !ragged && S >= 16 && S <= 32 && blocks >= 8u * (unsigned)c->resident_waves) cap_iter = 6;
    const bool capped = cap_iter > 0 && !long_form && !split_on && (!a.order || ragged || compact) && !warm_kernel && !queue_on && S <= BTRAPZ_MAX_SEGMENTS &&
                        cap_iter < a.max_iter && cap_iter + BTRAPZ_CAP_HI < 4000 && elastic != 2;   // (4000: the lean record's 12-bit counters)
    // Workspace of the two launches: hand-over slots for BTRAPZ_SUSP_PERCENT of the axis problems (a group that finds none simply
    // goes on), 74 doubles per segment each -- 19 KB per slot at 64 segments, 388 MB for 65 536 candidates of 20 -- but
    // never more than BTRAPZ_SUSP_BYTES_MAX.  When the device cannot give it, a solve that chose the two launches by
    // itself (cap_iter = 0) runs as one launch; one that was asked for them (cap_iter > 0) fails with BTRAPZ_ENOMEM.
    size_t slots = 0;
    bool capped_ws = false;
    if (capped) {
      slots = 2 * (size_t)B * (size_t)BTRAPZ_SUSP_PERCENT / 100;
      if (slots < 1024) slots = 1024;
      const size_t slots_max = (size_t)BTRAPZ_SUSP_BYTES_MAX / (sizeof(double) * 74 * (size_t)S);
      if (slots > slots_max) slots = slots_max;
      if (slots > (size_t)0x7fffffff) slots = (size_t)0x7fffffff;   // (susp_cap is an int)
      const size_t need_state = slots * 74 * (size_t)S, need_ints = 400 + 4 * (size_t)B;   // count[4] tables[2][198] keys[2][B] slots[2B]
      bool ok = slots >= 64;
      if (ok && need_state > c->susp_state_doubles) {
        (void)hipFree(c->d_susp_state); c->d_susp_state = nullptr; c->susp_state_doubles = 0;
        if (hipMalloc(&c->d_susp_state, sizeof(double) * need_state) == hipSuccess) c->susp_state_doubles = need_state;
        else { c->d_susp_state = nullptr; ok = false; }
      }
      if (ok && need_ints > c->susp_ints) {
        (void)hipFree(c->d_susp_ints); c->d_susp_ints = nullptr; c->susp_ints = 0;
        if (hipMalloc(&c->d_susp_ints, sizeof(int) * need_ints) == hipSuccess) c->susp_ints = need_ints;
        else { c->d_susp_ints = nullptr; ok = false; }
      }
      if (ok && 2 * (size_t)B > c->rescue_cap) {
        (void)hipFree(c->d_rescue); c->d_rescue = nullptr; c->rescue_cap = 0;
        if (hipMalloc(&c->d_rescue, sizeof(int) * 4 * (size_t)B) == hipSuccess) c->rescue_cap = 2 * (size_t)B;
        else { c->d_rescue = nullptr; ok = false; }
      }
      if (ok && !c->d_rescue_meta && hipMalloc(&c->d_rescue_meta, sizeof(int) * 2 * 198) != hipSuccess) { c->d_rescue_meta = nullptr; ok = false; }
      if (!ok) {
        (void)hipGetLastError();   // (the failed allocation's error is handled here)
        if (cap_requested) { c->err = "btrapz_options.cap_iter: no memory for the hand-over workspace"; return BTRAPZ_ENOMEM; }
      }
      capped_ws = ok;
    }
    if (capped && capped_ws) {
      // (one workspace, one memset: the counter, the bucket tables of both axes and the keys are zeroed together)
      int *count = c->d_susp_ints, *tables = count + 4, *keys = tables + 2 * 198, *slot_of = keys + 2 * (size_t)B;
      HIPCHK(c, hipMemsetAsync(count, 0, sizeof(int) * (400 + 2 * (size_t)B), stream));
      KernelArgs p1 = a;
      p1.cap_iter = cap_iter; p1.cap_alone = BTRAPZ_CAP_ALONE; p1.cap_hi = cap_iter + BTRAPZ_CAP_HI; p1.cap_score = BTRAPZ_CAP_SCORE; p1.susp_cap = (int)slots; p1.susp_state = c->d_susp_state; p1.susp_count = count;
      p1.susp_slot = slot_of; p1.susp_key = keys;
      // (the ordered instantiation serves ragged batches and uniform ones that go through a.order -- the pre-pass -- and
      //  gives a uniform batch the bits of the memory-order one: btrapz_lean_body.h)
      if (lean_on) {
        if (p1.order) hipLaunchKernelGGL(ipm_solve_lean_capped_ordered_kernel, dim3(blocks), dim3(64), 0, stream, p1, (const double *)c->d_mqm);
        else hipLaunchKernelGGL(ipm_solve_lean_capped_kernel, dim3(blocks), dim3(64), 0, stream, p1, (const double *)c->d_mqm);
      } else {
        if (p1.order) hipLaunchKernelGGL(ipm_solve_capped_ordered_kernel, dim3(blocks), dim3(64), 0, stream, p1, (const double *)c->d_mqm);
        else hipLaunchKernelGGL(ipm_solve_capped_kernel, dim3(blocks), dim3(64), 0, stream, p1, (const double *)c->d_mqm);
      }
      int *lists = c->d_rescue + 2 * (size_t)B;
      const unsigned nb = (unsigned)((B + 255) / 256);
      const int list_S = ragged ? 0 : -S;   // ragged: keys are segment counts; uniform: convergence classes, 0 = not listed
      // both axes in one launch each (gridDim.y = 2: keys [2][B], tables [2][198], lists [2][B])
      hipLaunchKernelGGL(bucket_hist_kernel, dim3(nb, 2), dim3(256), 0, stream, B, S, (const int *)keys, tables, list_S);
      hipLaunchKernelGGL(bucket_prefix_kernel, dim3(1, 2), dim3(64), 0, stream, tables, list_S);
      hipLaunchKernelGGL(bucket_scatter_kernel, dim3(nb, 2), dim3(256), 0, stream, B, S, (const int *)keys, tables, lists,
                         (double *)nullptr, (int *)nullptr, (int *)nullptr, list_S);
      KernelArgs p2 = p1;
      p2.cap_iter = 0; p2.order = lists; p2.seg_count = nullptr; p2.cand_prefix = tables; p2.wave_prefix = tables + 66;
      p2.bucket_S = ragged ? 0 : S;
      // (ragged: no candidate has more than min(S, 64) segments, so no wavefront holds fewer groups than that allows)
      const unsigned rblocks = 2u * (unsigned)(slots / (size_t)(64 / (S < 64 ? S : 64)) + 65);
      if (lean_on) hipLaunchKernelGGL(ipm_solve_lean_resume_kernel, dim3(rblocks), dim3(64), 0, stream, p2, (const double *)c->d_mqm);
      else hipLaunchKernelGGL(ipm_solve_resume_kernel, dim3(rblocks), dim3(64), 0, stream, p2, (const double *)c->d_mqm);
      c->last_form = lean_on ? 11 : 3;
    } else if (long_form) {
      c->last_form = 2;
      hipLaunchKernelGGL(ipm_solve_long_kernel, dim3(2u * (unsigned)B), dim3(64u * (unsigned)((S + 63) / 64)), 0, stream, a,
                         (const double *)c->d_mqm);
    } else if (split_on) {
      c->last_form = 1;
      hipLaunchKernelGGL(ipm_solve_split_kernel, dim3(2u * (unsigned)B), dim3(64), 0, stream, a, (const double *)c->d_mqm);
#ifdef BTRAPZ_EXPERIMENTS
    } else if (queue_on && !a.order && !warm_kernel && blocks >= 3u * (unsigned)c->resident_waves) {
      HIPCHK(c, hipMemsetAsync(c->d_queue, 0, sizeof(int) * 2, stream));
      c->last_form = 4;
      hipLaunchKernelGGL(ipm_solve_queue_kernel, dim3((unsigned)c->resident_waves & ~1u), dim3(64), 0, stream, a,
                         (const double *)c->d_mqm);
#endif
    } else if (lean_on) {
      c->last_form = 8;
      // (a.order: the buckets of a ragged batch, or the hint classes / the pre-pass's list of a uniform one)
      auto lean_kernel = warm_kernel ? (a.order ? ipm_solve_lean_warm_ordered_kernel : ipm_solve_lean_warm_kernel)
                                     : (a.order ? ipm_solve_lean_ordered_kernel : ipm_solve_lean_kernel);
      hipLaunchKernelGGL(lean_kernel, dim3(blocks), dim3(64), 0, stream, a, (const double *)c->d_mqm);
    } else {
      c->last_form = 0;
      hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64), 0, stream, a, (const double *)c->d_mqm);
    }
    HIPCHK(c, hipGetLastError());
  }
  if (elastic && long_form) {
    // Rescue pass of the long form: one workgroup per axis problem again; those whose problem did not stall leave at once
    HIPCHK(c, hipMemsetAsync(c->d_axis_viol, 0, sizeof(double) * 8 * (size_t)B, stream));
    c->viol_valid = (size_t)B;
    KernelArgs e = a;
    e.order = nullptr; e.x0 = nullptr; e.lam0 = nullptr; e.lam_out = nullptr;
    e.max_iter = a.max_iter < 120 ? 120 : a.max_iter;   // (as below)
    e.tau_iters = 0;
    e.stall_start = 4 * BTRAPZ_STALL_START; e.stall_len = 4 * BTRAPZ_STALL_LENGTH;
    hipLaunchKernelGGL(ipm_solve_long_elastic_kernel, dim3(2u * (unsigned)B), dim3(64u * (unsigned)((S + 63) / 64)), 0, stream, e,
                       (const double *)c->d_mqm);
    HIPCHK(c, hipGetLastError());
  } else if (elastic) {
    // Rescue pass: the axis problems that stalled (elastic == 2: all of them) are listed per axis, bucketed by
    // segment count with the machinery of the ragged batches, and solved again with elastic rows.  The lists are
    // built on the device; the launch is sized for the worst case and its unused wavefronts leave at once.
    if (2 * (size_t)B > c->rescue_cap) {
      (void)hipFree(c->d_rescue); c->d_rescue = nullptr; c->rescue_cap = 0;
      HIPCHK(c, hipMalloc(&c->d_rescue, sizeof(int) * 4 * (size_t)B));
      c->rescue_cap = 2 * (size_t)B;
    }
    if (!c->d_rescue_meta) HIPCHK(c, hipMalloc(&c->d_rescue_meta, sizeof(int) * 2 * 198));
    HIPCHK(c, hipMemsetAsync(c->d_axis_viol, 0, sizeof(double) * 8 * (size_t)B, stream));
    c->viol_valid = (size_t)B;
    int *keys = c->d_rescue, *lists = c->d_rescue + 2 * (size_t)B;
    const unsigned nb = (unsigned)((B + 255) / 256);
    if (elastic == 2)
      hipLaunchKernelGGL(rescue_init_kernel, dim3(nb), dim3(256), 0, stream, B, S, seg_count, c->d_axis_obj,
                         c->d_axis_status, c->d_axis_iters);
    hipLaunchKernelGGL(rescue_keys_kernel, dim3(nb), dim3(256), 0, stream, B, S, seg_count, (const int *)c->d_axis_status,
                       keys, elastic == 2 ? 1 : 0);
    HIPCHK(c, hipMemsetAsync(c->d_rescue_meta, 0, sizeof(int) * 2 * 198, stream));
    for (int ax = 0; ax < 2; ax++) {
      int *meta = c->d_rescue_meta + ax * 198;
      const int *k = keys + (size_t)ax * B;
      hipLaunchKernelGGL(bucket_hist_kernel, dim3(nb), dim3(256), 0, stream, B, S, k, meta, 0);
      hipLaunchKernelGGL(bucket_prefix_kernel, dim3(1), dim3(64), 0, stream, meta, 0);
      hipLaunchKernelGGL(bucket_scatter_kernel, dim3(nb), dim3(256), 0, stream, B, S, k, meta, lists + (size_t)ax * B,
                         (double *)nullptr, (int *)nullptr, (int *)nullptr, 0);
    }
    HIPCHK(c, hipGetLastError());
    KernelArgs e = a;
    e.order = lists; e.seg_count = seg_count; e.cand_prefix = c->d_rescue_meta; e.wave_prefix = c->d_rescue_meta + 66;
    e.bucket_S = 0; e.x0 = nullptr; e.lam0 = nullptr; e.lam_out = nullptr;
    // Least-violation problems take 25-40 iterations, and up to 60 where a row's violation is fixed by the data (an
    // initial state outside segment 0's rows): its multiplier has to grow to violation / delta before the score moves.
    // The pass is rare and not on the throughput path: be patient (the oracle's relaxed solve is; with these limits every
    // one of 6 000 fuzzed find_traj calls decides as it does, with 80 / 2x / 2x six of 4 800 did not).
    e.max_iter = a.max_iter < 120 ? 120 : a.max_iter;
    e.tau_iters = 0;                                  // conservative step rule throughout
    e.stall_start = 4 * BTRAPZ_STALL_START; e.stall_len = 4 * BTRAPZ_STALL_LENGTH;
    const int smax = seg_count ? (S < 64 ? S : 64) : S;
    const unsigned eblocks = 2u * (unsigned)(B / (64 / smax) + 65);
    hipLaunchKernelGGL(ipm_solve_elastic_kernel, dim3(eblocks), dim3(64), 0, stream, e, (const double *)c->d_mqm);
    HIPCHK(c, hipGetLastError());
  }
  // Ragged batch with slots for more than 64 segments: the candidates that HAVE more than 64 (the bucket kernels have
  // marked them "no usable corridor") are solved by the long form, one launch per segment count with the candidates of
  // that count listed.  The counts come to the host for this (one synchronisation: horizons beyond 64 s are not the
  // throughput path); cold solves without a rescue pass -- what the long form serves.  The reference has no limit on the
  // segment count (std::vector, src/solve_3d.cc:323-486,729-772).
  if (seg_count && S > BTRAPZ_MAX_SEGMENTS && !warm_kernel && elastic == 0) {
    std::vector<int> counts((size_t)B);
    HIPCHK(c, hipMemcpyAsync(counts.data(), seg_count, sizeof(int) * (size_t)B, hipMemcpyDeviceToHost, stream));
    HIPCHK(c, hipStreamSynchronize(stream));
    const int smax = S < BTRAPZ_MAX_SEGMENTS_LONG ? S : BTRAPZ_MAX_SEGMENTS_LONG;
    std::vector<std::vector<int>> by_count((size_t)smax + 1);
    size_t n_long = 0;
    for (int b = 0; b < B; b++)
      if (counts[b] > BTRAPZ_MAX_SEGMENTS && counts[b] <= smax) { by_count[(size_t)counts[b]].push_back(b); ++n_long; }
    if (n_long) {
      if (n_long > c->long_list_cap) {
        (void)hipFree(c->d_long_list); c->d_long_list = nullptr; c->long_list_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_long_list, sizeof(int) * n_long));
        c->long_list_cap = n_long;
      }
      std::vector<int> flat;
      flat.reserve(n_long);
      for (const auto &l : by_count) flat.insert(flat.end(), l.begin(), l.end());
      HIPCHK(c, hipMemcpyAsync(c->d_long_list, flat.data(), sizeof(int) * n_long, hipMemcpyHostToDevice, stream));
      HIPCHK(c, hipStreamSynchronize(stream));   // (flat goes out of scope)
      size_t off = 0;
      for (int s = BTRAPZ_MAX_SEGMENTS + 1; s <= smax; s++) {
        const size_t n = by_count[(size_t)s].size();
        if (!n) continue;
        KernelArgs l = a;
        l.S = s; l.order = c->d_long_list + off; l.bucket_S = (int)n; l.seg_count = nullptr; l.cand_prefix = nullptr; l.wave_prefix = nullptr;
        l.x0 = nullptr; l.lam0 = nullptr; l.lam_out = nullptr; l.x_out = nullptr;
        hipLaunchKernelGGL(ipm_solve_long_kernel, dim3(2u * (unsigned)n), dim3(64u * (unsigned)((s + 63) / 64)), 0, stream, l, (const double *)c->d_mqm);
        HIPCHK(c, hipGetLastError());
        off += n;
      }
      c->last_form |= 16;   // (+ 16: the long form served part of a ragged batch)
    }
  }
  hipLaunchKernelGGL(finalize_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, B, c->d_axis_obj, c->d_axis_status,
                     c->d_axis_iters, cost, status, iters);
  HIPCHK(c, hipGetLastError());
  c->ws_stream = stream; c->ws_used = true;
  HIPCHK(c, hipEventRecord(c->ws_free, stream));
  return BTRAPZ_OK;
}

BTRAPZ_EXPORT int btrapz_solve_batch_device(btrapz_ctx *c, const btrapz_shared *sh, const btrapz_options *opt, int B,
                                         int S, const double *seg, const double *init, const double *ref_end,
                                         const double *dl_bounds, double *ctrl, double *cost, int *status, int *iters,
                                         void *stream) {
  return solve_common(c, sh, opt, nullptr, B, S, nullptr, seg, init, ref_end, dl_bounds, ctrl, cost, status, iters, stream);
}

BTRAPZ_EXPORT int btrapz_solve_ragged_device(btrapz_ctx *c, const btrapz_shared *sh, const btrapz_options *opt, int B,
                                          int seg_stride, const double *seg, const int *seg_count, const double *init,
                                          const double *ref_end, const double *dl_bounds, double *ctrl, double *cost,
                                          int *status, int *iters, void *stream) {
  if (c && !seg_count) { c->err = "seg_count is null"; return BTRAPZ_EINVAL; }
  return solve_common(c, sh, opt, nullptr, B, seg_stride, seg_count, seg, init, ref_end, dl_bounds, ctrl, cost, status,
                      iters, stream);
}

BTRAPZ_EXPORT int btrapz_solve_warm_device(btrapz_ctx *c, const btrapz_shared *sh, const btrapz_options *opt,
                                        const btrapz_warm *warm, int B, int seg_stride, const double *seg,
                                        const int *seg_count, const double *init, const double *ref_end,
                                        const double *dl_bounds, double *ctrl, double *cost, int *status, int *iters,
                                        void *stream) {
  return solve_common(c, sh, opt, warm, B, seg_stride, seg_count, seg, init, ref_end, dl_bounds, ctrl, cost, status,
                      iters, stream);
}

BTRAPZ_EXPORT int btrapz_rescue_violations_device(btrapz_ctx *c, int B, double *viol, void *stream_) {
  if (!c) return BTRAPZ_EINVAL;
  if (B < 1 || !viol) { c->err = "invalid argument"; return BTRAPZ_EINVAL; }
  if ((size_t)B != c->viol_valid) { c->err = "the context's last solve of B candidates had no rescue pass (btrapz_options.elastic)"; return BTRAPZ_EINVAL; }
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t stream = (hipStream_t)stream_;
  if (c->ws_used && stream != c->ws_stream) HIPCHK(c, hipStreamWaitEvent(stream, c->ws_free, 0));
  hipLaunchKernelGGL(rescue_violations_kernel, dim3((4 * B + 255) / 256), dim3(256), 0, stream, B, (const double *)c->d_axis_viol, viol);
  HIPCHK(c, hipGetLastError());
  return BTRAPZ_OK;
}

BTRAPZ_EXPORT int btrapz_eval_states_device(btrapz_ctx *c, int B, int seg_stride, const int *seg_count, const double *seg,
                                         const double *ctrl, int n_times, const double *times, double *x,
                                         void *stream_) {
  if (!c) return BTRAPZ_EINVAL;
  if (B < 1 || seg_stride < 1 || n_times < 1 || !seg || !ctrl || !times || !x) { c->err = "invalid argument"; return BTRAPZ_EINVAL; }
  HIPCHK(c, hipSetDevice(c->device));
  const long long n = (long long)B * n_times;
  hipLaunchKernelGGL(eval_states_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, B,
                     seg_stride, seg_count, seg, ctrl, n_times, times, x);
  HIPCHK(c, hipGetLastError());
  return BTRAPZ_OK;
}

// The launches of the corridor stage: `a` holds the shapes and every pointer but the retry list.  prisms: the fused
// prism + corridor kernels (the strips are evaluated from a.prisms instead of read from a.s_bounds / a.l_bounds).
// Beyond the wave-wide kernels' shapes (more than 512 knots or 64 obstacles): one lane per candidate on
// lists in a workspace (corridor_serial_kernel).  Capacities: what the host driver of find_traj allows itself (2 N + 16
// segments per obstacle) while the workspace stays below BTRAPZ_CORRIDOR_WS_MAX, else what a horizon of N knots needs when
// its bounds break at most 64 times per obstacle; as many selected segments (before the de-dup) as there are segments, under
// the same budget.
#ifndef BTRAPZ_CORRIDOR_WS_MAX
#define BTRAPZ_CORRIDOR_WS_MAX ((size_t)2 << 30)
#endif
static int launch_corridor_serial(btrapz_ctx *c, CorridorArgs a, hipStream_t stream) {
  const size_t seg_bytes = 104;   // sizeof(Seg): corridor_kernels.hip asserts it
  const size_t B = (size_t)a.B, O = (size_t)a.num_obs;
  size_t cap_o = 2 * (size_t)a.N + 16;
  const size_t cap_min = (size_t)(a.N - 1) / 10 + 1 + 64;
  if (B * O * cap_o * seg_bytes > BTRAPZ_CORRIDOR_WS_MAX) {
    cap_o = BTRAPZ_CORRIDOR_WS_MAX / (B * O * seg_bytes);
    if (cap_o < cap_min) cap_o = cap_min;
  }
  // selected segments before the de-dup: every segment of every obstacle can be one (its copies: only for NaN segments)
  size_t cap_sel = O * cap_o;
  if (B * cap_sel * seg_bytes > BTRAPZ_CORRIDOR_WS_MAX) {
    cap_sel = BTRAPZ_CORRIDOR_WS_MAX / (B * seg_bytes);
    if (cap_sel < 2 * (size_t)a.seg_stride + 16) cap_sel = 2 * (size_t)a.seg_stride + 16;
  }
  const size_t need = (B * O * cap_o + B * cap_sel) * seg_bytes;
  if (need > c->corr_ws_cap) {
    (void)hipFree(c->d_corr_ws); c->d_corr_ws = nullptr; c->corr_ws_cap = 0;
    if (hipMalloc(&c->d_corr_ws, need) != hipSuccess) { (void)hipGetLastError(); c->err = "corridor stage: no memory for the segment lists of this horizon"; return BTRAPZ_ENOMEM; }
    c->corr_ws_cap = need;
  }
  if (c->ws_used && stream != c->ws_stream) HIPCHK(c, hipStreamWaitEvent(stream, c->ws_free, 0));
  a.cap_o = (int)cap_o; a.cap_sel = (int)cap_sel; a.pass = 0; a.retry_list = nullptr; a.retry_count = nullptr;
  Seg *all = reinterpret_cast<Seg *>(c->d_corr_ws);
  Seg *sel = reinterpret_cast<Seg *>(reinterpret_cast<char *>(c->d_corr_ws) + B * O * cap_o * seg_bytes);
  hipLaunchKernelGGL(corridor_serial_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, stream, a, all, sel);
  HIPCHK(c, hipGetLastError());
  c->ws_stream = stream; c->ws_used = true;
  HIPCHK(c, hipEventRecord(c->ws_free, stream));
  return BTRAPZ_OK;
}

static int launch_corridor_stage(btrapz_ctx *c, CorridorArgs a, bool prisms, hipStream_t stream) {
  const int B = a.B, N = a.N, num_obs = a.num_obs, seg_stride = a.seg_stride;
  if (!prisms && (N > 512 || num_obs > 64)) return launch_corridor_serial(c, a, stream);
  // Two passes (corridor_kernels.hip): lists sized for the usual case first -- LDS per workgroup is what limits the
  // wavefronts per CU -- then the candidates that overflowed them, with the full capacities.
  if ((size_t)B + 1 > c->retry_cap) {
    (void)hipFree(c->d_retry); c->d_retry = nullptr; c->retry_cap = 0;
    HIPCHK(c, hipMalloc(&c->d_retry, sizeof(int) * ((size_t)B + 1)));
    c->retry_cap = (size_t)B + 1;
  }
  HIPCHK(c, hipMemsetAsync(c->d_retry, 0, sizeof(int), stream));
  a.retry_count = c->d_retry; a.retry_list = c->d_retry + 1;
  const int cap_o_big = 160 / num_obs, cap_sel_big = 64;        // MAX_ALL / O, MAX_SEL of corridor_kernels.hip
  // first-pass lists: one slot per second of horizon (CorridorSplit's pieces) plus a few for the slope breaks.  Every
  // slot is 104 bytes of LDS per obstacle and LDS bounds the wavefronts per CU; a candidate with more breaks only takes
  // the retry pass.  Measured on 65 536 candidates (+9 was round 2's margin): N = 71, 3 obstacles 0.305 -> 0.272 ms with
  // +3 and five wavefronts per SIMD; N = 201, 2 obstacles 0.686 -> 0.637 with +5 (0.663 with +3: more retries).
#ifndef CABL_CAPO_EXTRA
#define CABL_CAPO_EXTRA (N <= 128 ? 3 : 5)
#endif
  // (round 6, LDS once more: ocount[] cut to the obstacle count and 1.5 x seg_stride selected segments -- 7.4 -> 6.9 KB per
  //  wavefront -- N = 71 x 3 obstacles 0.240 -> 0.246 ms, N = 201 x 2 0.569 -> 0.542; with +2 instead of +3 slots per obstacle
  //  the retry pass takes over (0.84 ms at N = 201): LDS is not what holds this kernel any more)
  int cap_o_small = (N - 1) / 10 + CABL_CAPO_EXTRA, cap_sel_small = 2 * seg_stride < 16 ? 16 : 2 * seg_stride;
  if (cap_o_small > cap_o_big) cap_o_small = cap_o_big;
  if (cap_sel_small > cap_sel_big) cap_sel_small = cap_sel_big;
  const size_t slope_bytes = sizeof(double) * 2 * (size_t)N * num_obs;
  const int staged = slope_bytes <= 24 * 1024 ? 1 : 0;
  if (prisms && !staged) { c->err = "internal: fused corridor stage without a staged slope table"; return BTRAPZ_EINVAL; }
  auto lds_bytes = [&](int cap_o, int cap_sel) {
    const size_t cap_all = (size_t)cap_o * num_obs;
    return 104 * (cap_all > (size_t)cap_sel ? cap_all : (size_t)cap_sel) + (staged && slope_bytes > sizeof(double) * 4 * (size_t)N ? slope_bytes : sizeof(double) * 4 * (size_t)N) +
           sizeof(int) * (cap_all + 64 + cap_sel) + sizeof(short) * (cap_all + cap_sel) + 16 +
           (prisms ? prism_tab_bytes(a.P) + 8 : 0)
#ifdef CABL_PAD   // occupancy experiments: scratch/build_variant.sh X -DCABL_PAD=bytes
           + CABL_PAD
#endif
        ;
  };
  const bool two_pass = cap_o_small < cap_o_big || cap_sel_small < cap_sel_big;
  a.pass = 0; a.cap_o = cap_o_small; a.cap_sel = cap_sel_small;
  if (!two_pass) { a.retry_list = nullptr; a.retry_count = nullptr; }
  // (horizons of at most 128 knots: the instantiation that holds half the prefetch registers; the first pass of two
  //  runs without the serial statement of the extraction -- a candidate that needs it takes the retry pass)
  auto kernel = prisms ? (N <= 128 ? prism_corridor_batch_short_kernel : prism_corridor_batch_kernel)
                       : (N <= 128 ? corridor_batch_short_kernel : corridor_batch_kernel);
  auto first = !(two_pass && staged) ? kernel
               : prisms ? (N <= 128 ? prism_corridor_first_short_kernel : prism_corridor_first_kernel)
                        : (N <= 128 ? corridor_first_short_kernel : corridor_first_kernel);
  hipLaunchKernelGGL(first, dim3(B), dim3(64), lds_bytes(a.cap_o, a.cap_sel), stream, a, staged);
  HIPCHK(c, hipGetLastError());
  if (two_pass) {
    a.pass = 1; a.cap_o = cap_o_big; a.cap_sel = cap_sel_big;
    const unsigned blocks = B < 1024 ? (unsigned)B : 1024u;
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64), lds_bytes(a.cap_o, a.cap_sel), stream, a, staged);
  }
  HIPCHK(c, hipGetLastError());
  return BTRAPZ_OK;
}

BTRAPZ_EXPORT int btrapz_corridor_batch_device(btrapz_ctx *c, int variant, int B, int N, int num_obs, double delta,
                                            const double *s_bounds, const double *l_bounds, const double *ds_bounds,
                                            const double *dl_bounds_knots, const double *s_ref, const double *l_ref,
                                            int seg_stride, double *seg, int *seg_count, double *ref_end,
                                            double *dl_bounds, void *stream_) {
  if (!c) return BTRAPZ_EINVAL;
  // (N, num_obs: the reference has no limit -- std::vector --; the bounds here are those of find_traj's parser, corridor.cpp.
  //  More than 512 knots or 64 obstacles: the one-lane-per-candidate kernel instead of the wave-wide ones)
  if (variant < 0 || variant > 1 || B < 1 || N < 3 || N > 100000 || num_obs < 1 || num_obs > 1000 || !(delta > 0) ||
      seg_stride < 1 || seg_stride > BTRAPZ_MAX_SEGMENTS_LONG || !s_bounds || !l_bounds || !ds_bounds || !dl_bounds_knots || !s_ref || !l_ref || !seg ||
      !seg_count || !ref_end || !dl_bounds) {
    c->err = "invalid argument";
    return BTRAPZ_EINVAL;
  }
  HIPCHK(c, hipSetDevice(c->device));
  CorridorArgs a;
  memset(&a, 0, sizeof a);
  a.B = B; a.N = N; a.num_obs = num_obs; a.variant = variant; a.seg_stride = seg_stride; a.delta = delta;
  a.s_bounds = s_bounds; a.l_bounds = l_bounds; a.ds_bounds = ds_bounds; a.dl_bounds = dl_bounds_knots;
  a.s_ref = s_ref; a.l_ref = l_ref; a.seg = seg; a.seg_count = seg_count; a.ref_end = ref_end; a.dl10 = dl_bounds;
  return launch_corridor_stage(c, a, false, (hipStream_t)stream_);
}

// Obstacle prisms -> strips -> corridors: btrapz_prism_bounds_device + btrapz_corridor_batch_device (num_obs = O) with
// the strips evaluated where the corridor stage reads them instead of written to memory and read back.  Scenes whose
// slope table does not fit the kernel's LDS (O * N * 16 bytes > 24 KB) take the two launches through a workspace of
// the context: the results are the same either way.
BTRAPZ_EXPORT int btrapz_prism_corridor_batch_device(btrapz_ctx *c, int variant, int B, int P, int N, const btrapz_road *road,
                                                  const double *prisms, int O, double delta, const double *ds_bounds,
                                                  const double *dl_bounds_knots, const double *s_ref, const double *l_ref,
                                                  int seg_stride, double *seg, int *seg_count, double *ref_end,
                                                  double *dl_bounds, int *n_strips, void *stream_) {
  if (!c) return BTRAPZ_EINVAL;
  if (variant < 0 || variant > 1 || B < 1 || P < 1 || P > 16 || N < 3 || N > 512 || O < 1 || O > 64 || !(delta > 0) || !road ||
      !(road->knots_per_second > 0) || !prisms || seg_stride < 1 || !ds_bounds || !dl_bounds_knots || !s_ref || !l_ref ||
      !seg || !seg_count || !ref_end || !dl_bounds) {
    c->err = "invalid argument";
    return BTRAPZ_EINVAL;
  }
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t stream = (hipStream_t)stream_;
  CorridorArgs a;
  memset(&a, 0, sizeof a);
  a.B = B; a.N = N; a.num_obs = O; a.variant = variant; a.seg_stride = seg_stride; a.delta = delta;
  a.ds_bounds = ds_bounds; a.dl_bounds = dl_bounds_knots;
  a.s_ref = s_ref; a.l_ref = l_ref; a.seg = seg; a.seg_count = seg_count; a.ref_end = ref_end; a.dl10 = dl_bounds;
  if (sizeof(double) * 2 * (size_t)N * O <= 24 * 1024) {
    a.prisms = prisms; a.P = P; a.road = prism_road(road); a.n_strips = n_strips;
    return launch_corridor_stage(c, a, true, stream);
  }
  const size_t need = sizeof(double) * 4 * (size_t)B * O * N + sizeof(int) * (size_t)B;
  if (need > c->strip_cap) {
    (void)hipFree(c->d_strips); c->d_strips = nullptr; c->strip_cap = 0;
    HIPCHK(c, hipMalloc(&c->d_strips, need));
    c->strip_cap = need;
  }
  if (c->ws_used && stream != c->ws_stream) HIPCHK(c, hipStreamWaitEvent(stream, c->ws_free, 0));
  double *sb = reinterpret_cast<double *>(c->d_strips), *lb = sb + 2 * (size_t)B * O * N;
  int *ns = n_strips ? n_strips : reinterpret_cast<int *>(lb + 2 * (size_t)B * O * N);
  const int rc = btrapz_prism_bounds_device(c, B, P, N, road, prisms, O, sb, lb, ns, stream_);
  if (rc != BTRAPZ_OK) return rc;
  a.s_bounds = sb; a.l_bounds = lb;
  const int rc2 = launch_corridor_stage(c, a, false, stream);
  c->ws_stream = stream; c->ws_used = true;
  HIPCHK(c, hipEventRecord(c->ws_free, stream));
  return rc2;
}

BTRAPZ_EXPORT int btrapz_argmin_device(btrapz_ctx *c, int B, int group, long long index_base, const double *cost,
                                    long long *best_idx, double *best_cost, void *stream_) {
  if (!c) return BTRAPZ_EINVAL;
  if (B < 1 || group < 1 || B % group != 0 || !cost || !best_idx || !best_cost) { c->err = "invalid argument"; return BTRAPZ_EINVAL; }
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t stream = (hipStream_t)stream_;
  const int groups = B / group;
  // large groups are split over several blocks (a single block over 65 536 costs is latency-bound)
  int chunks = group >= 8192 ? (group + 1023) / 1024 : 1;
  if (chunks > 256) chunks = 256;
  if (chunks > 1) {
    const size_t need = (size_t)groups * chunks;
    if (need > c->argmin_cap) {
      (void)hipFree(c->d_argmin_cost); (void)hipFree(c->d_argmin_idx);
      c->d_argmin_cost = nullptr; c->d_argmin_idx = nullptr; c->argmin_cap = 0;
      HIPCHK(c, hipMalloc(&c->d_argmin_cost, sizeof(double) * need));
      HIPCHK(c, hipMalloc(&c->d_argmin_idx, sizeof(long long) * need));
      c->argmin_cap = need;
    }
    if (c->ws_used && stream != c->ws_stream) HIPCHK(c, hipStreamWaitEvent(stream, c->ws_free, 0));
  }
  hipLaunchKernelGGL(argmin_kernel, dim3(groups, chunks), dim3(256), 0, stream, group, index_base, cost, best_idx, best_cost,
                     c->d_argmin_cost, c->d_argmin_idx);
  if (chunks > 1) {
    hipLaunchKernelGGL(argmin_final_kernel, dim3(groups), dim3(256), 0, stream, chunks, index_base,
                       (const double *)c->d_argmin_cost, (const long long *)c->d_argmin_idx, best_idx, best_cost);
    c->ws_stream = stream; c->ws_used = true;
    HIPCHK(c, hipEventRecord(c->ws_free, stream));
  }
  HIPCHK(c, hipGetLastError());
  return BTRAPZ_OK;
}

BTRAPZ_EXPORT int btrapz_argmin_pairs_device(btrapz_ctx *c, int world, int n, const long long *pairs, double *best_cost,
                                          long long *best_idx, void *stream_) {
  if (!c) return BTRAPZ_EINVAL;
  if (world < 1 || n < 1 || !pairs || !best_cost || !best_idx) { c->err = "invalid argument"; return BTRAPZ_EINVAL; }
  HIPCHK(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(argmin_pairs_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream_, world, n, pairs, best_cost,
                     best_idx);
  HIPCHK(c, hipGetLastError());
  return BTRAPZ_OK;
}

BTRAPZ_EXPORT int btrapz_sample_device(btrapz_ctx *c, int B, int S, double delta, const double *seg, const double *init,
                                    const double *ctrl, int nsel, const long long *sel, int max_points, double *out,
                                    int *npoints, void *stream_) {
  if (!c) return BTRAPZ_EINVAL;
  if (B < 1 || S < 1 || nsel < 1 || !(delta > 0) || !seg || !init || !ctrl || !sel || !out || !npoints || max_points < 1) {
    c->err = "invalid argument"; return BTRAPZ_EINVAL;
  }
  HIPCHK(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(sample_kernel, dim3(nsel), dim3(64), 0, (hipStream_t)stream_, B, S, (const int *)nullptr, delta, seg,
                     init, ctrl, nsel, sel, max_points, out, npoints);
  HIPCHK(c, hipGetLastError());
  return BTRAPZ_OK;
}

BTRAPZ_EXPORT int btrapz_sample_ragged_device(btrapz_ctx *c, int B, int seg_stride, const int *seg_count, double delta,
                                           const double *seg, const double *init, const double *ctrl, int nsel,
                                           const long long *sel, int max_points, double *out, int *npoints,
                                           void *stream_) {
  if (!c) return BTRAPZ_EINVAL;
  if (B < 1 || seg_stride < 1 || nsel < 1 || !(delta > 0) || !seg || !init || !ctrl || !sel || !out || !npoints ||
      max_points < 1) {
    c->err = "invalid argument"; return BTRAPZ_EINVAL;
  }
  HIPCHK(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(sample_kernel, dim3(nsel), dim3(64), 0, (hipStream_t)stream_, B, seg_stride, seg_count, delta, seg,
                     init, ctrl, nsel, sel, max_points, out, npoints);
  HIPCHK(c, hipGetLastError());
  return BTRAPZ_OK;
}

BTRAPZ_EXPORT int btrapz_solve_batch_host(btrapz_ctx *c, const btrapz_shared *sh, const btrapz_options *opt, int B, int S,
                                       const double *seg, const double *init, const double *ref_end,
                                       const double *dl_bounds, double *ctrl, double *cost, int *status, int *iters) {
  if (!c) return BTRAPZ_EINVAL;
  if (B < 1 || S < 1 || S > BTRAPZ_MAX_SEGMENTS_LONG) { c->err = "invalid argument"; return BTRAPZ_EINVAL; }
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n_seg = (size_t)BTRAPZ_NUM_SEG_FIELDS * B * S, n_init = (size_t)B * 6, n_re = (size_t)B * 2, n_dl = (size_t)B * 10;
  const size_t n_ctrl = (size_t)B * 12 * S, n_cost = B;
  const size_t need = n_seg + n_init + n_re + n_dl + n_ctrl + n_cost;
  if (need > c->stage_cap) {
    (void)hipFree(c->d_stage); c->d_stage = nullptr; c->stage_cap = 0;
    HIPCHK(c, hipMalloc(&c->d_stage, sizeof(double) * need));
    c->stage_cap = need;
  }
  if ((size_t)2 * B > c->istage_cap) {
    (void)hipFree(c->d_istage); c->d_istage = nullptr; c->istage_cap = 0;
    HIPCHK(c, hipMalloc(&c->d_istage, sizeof(int) * 2 * (size_t)B));
    c->istage_cap = 2 * (size_t)B;
  }
  double *d_seg = c->d_stage, *d_init = d_seg + n_seg, *d_re = d_init + n_init, *d_dl = d_re + n_re;
  double *d_ctrl = d_dl + n_dl, *d_cost = d_ctrl + n_ctrl;
  int *d_status = c->d_istage, *d_iters = c->d_istage + B;
  HIPCHK(c, hipMemcpy(d_seg, seg, sizeof(double) * n_seg, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(d_init, init, sizeof(double) * n_init, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(d_re, ref_end, sizeof(double) * n_re, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(d_dl, dl_bounds, sizeof(double) * n_dl, hipMemcpyHostToDevice));
  int rc = btrapz_solve_batch_device(c, sh, opt, B, S, d_seg, d_init, d_re, d_dl, d_ctrl, d_cost, d_status, d_iters, nullptr);
  if (rc != BTRAPZ_OK) return rc;
  HIPCHK(c, hipDeviceSynchronize());
  HIPCHK(c, hipMemcpy(ctrl, d_ctrl, sizeof(double) * n_ctrl, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(cost, d_cost, sizeof(double) * n_cost, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(status, d_status, sizeof(int) * B, hipMemcpyDeviceToHost));
  if (iters) HIPCHK(c, hipMemcpy(iters, d_iters, sizeof(int) * B, hipMemcpyDeviceToHost));
  return BTRAPZ_OK;
}
