// find_traj.hip -- the reference's C-ABI driver, re-hosted on the HIP path.
//
// Replaces src/trp_wrapper.cpp:20-305 (trapezoid) and src/cub_wrapper.cpp:20-283 (cuboid):
// parse the corridor text file, extract/select the corridor segments on the host, solve the
// QP on the GPU through the batched entry point (B = 1), sample the trajectory on the GPU,
// compute the scalar cost the Python harness receives and write the trajectory file.
// There is no CPU solve: without a HIP device the call returns the failure sentinel 1e11
// (and says why on stderr), exactly what the harness treats as "optimizer failed".
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <mutex>
#include <string>
#include <vector>

#include <sched.h>

#include "btrapz_device.h"
#include "corridor.hpp"
#include "traj_cost.h"

using namespace btrapz;

namespace {

// Paths compiled into the reference's libraries (trp_wrapper.cpp:23,288 ; cub_wrapper.cpp:22,268 ;
// the older libbtrapz.so passes its own pair, see shim_btrapz.c).  Kept as defaults so the harness drops in unchanged.
const char *kDefaultInput[2] = {"/home/srujan_d/RISS/code/btrapz/src/c_road_s1_2.txt",
                                "/home/srujan_d/RISS/code/btrapz/src/c_road_s1_3.txt"};
const char *kDefaultOutputPrefix[2] = {"/home/srujan_d/RISS/code/btrapz/src/s1_slt_3d_",
                                       "/home/srujan_d/RISS/code/btrapz/src/s1_cub_3d_"};

// find_traj is re-entrant and concurrent: every calling thread holds, while it lives, a context, a stream and buffers
// of its own (a btrapz_ctx serves one launch sequence at a time), so concurrent callers -- e.g. parallel Optuna trials of
// the harness -- neither queue behind a lock nor behind each other on the device's null stream.  A thread that exits
// hands its set back to a pool (no HIP call at thread exit, when the runtime may already be gone) and the next new
// thread takes it over: a harness that starts a thread per trial holds as many sets as it has threads alive at once,
// not one per thread it ever started.
struct Caller {
  btrapz_ctx *ctx = nullptr;
  hipStream_t stream = nullptr;
  void *pinned = nullptr, *pinned_dev = nullptr;   // pinned host block of the single-candidate launch and its device mapping
  size_t pinned_bytes = 0;
  void *scratch = nullptr;                         // device block of the rescue attempt
  size_t scratch_bytes = 0;
  int device = 0;
};
// What btrapz_find_traj_last_iterations / _last_status report: per thread, and no reason to create a context.
struct LastCall { int iters = -1; int status = 0; double viol[4] = {0.0, 0.0, 0.0, 0.0}; };
thread_local LastCall t_last;
// The pool of sets whose thread has exited, and its lock: heap objects that are never destroyed -- a worker thread may
// exit (and hand its set back) during or after the destruction of this library's static objects at process exit.
struct CallerPool { std::mutex mutex; std::vector<Caller *> idle; };
CallerPool &caller_pool() { static CallerPool *p = new CallerPool(); return *p; }

struct CallerHolder {
  Caller *c = nullptr;
  ~CallerHolder() {
    if (!c) return;
    CallerPool &pool = caller_pool();
    std::lock_guard<std::mutex> lk(pool.mutex);
    pool.idle.push_back(c);
  }
};

Caller *this_caller() {
  thread_local CallerHolder me;
  if (!me.c) {
    const char *dev = getenv("BTRAPZ_DEVICE");
    const int device = dev ? atoi(dev) : 0;
    {
      CallerPool &pool = caller_pool();
      std::lock_guard<std::mutex> lk(pool.mutex);
      for (size_t i = 0; i < pool.idle.size(); i++)
        if (pool.idle[i]->device == device) {
          me.c = pool.idle[i];
          pool.idle.erase(pool.idle.begin() + (long)i);
          break;
        }
    }
    if (me.c) {
      // the previous owner may have returned from its last call on the completion word alone (wait_for_results): its
      // last launch must have retired before this thread rewrites the pinned block; its warm-start state is not ours
      (void)hipSetDevice(device);
      (void)hipStreamSynchronize(me.c->stream);
      btrapz_single_forget(me.c->ctx);
      return me.c;
    }
    Caller *c = new Caller();
    c->device = device;
    if (btrapz_create(&c->ctx, device) != BTRAPZ_OK) { delete c; return nullptr; }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { btrapz_destroy(c->ctx); delete c; return nullptr; }
    me.c = c;
  }
  return me.c;
}

bool verbose() { const char *v = getenv("BTRAPZ_VERBOSE"); return v && *v && *v != '0'; }
// BTRAPZ_VERBOSE=2: where a call's time goes (host stages and the launch), on stderr
bool timing() { const char *v = getenv("BTRAPZ_VERBOSE"); return v && *v == '2'; }
double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// The single launch writes its results through the mapping (a coherent, fine-grained pinned block: visible to the host
// before the kernel ends) and sets a completion word last (system-scope release): polling that word sees the results a
// few microseconds before hipStreamSynchronize returns (end-of-kernel cache release, completion signal, the runtime's
// wake-up).  The poll burns the calling core for at most 300 us -- a usual solve answers within 200 -- then yields the
// core between looks, and gives up after 2 ms (a long or faulted kernel): the stream wait takes over.  Solves of more
// than 64 segments (slow: several wavefronts per problem) do not poll at all.  BTRAPZ_SPIN=0: wait only.
bool wait_for_results(const double *h_out, hipStream_t stream, bool poll) {
  static const bool spin = [] { const char *v = getenv("BTRAPZ_SPIN"); return !(v && *v == '0'); }();
  if (spin && poll) {
    const volatile int *done = reinterpret_cast<const volatile int *>(h_out + 2) + 1;
    const double t0 = now_us();
    bool yielding = false;
    for (int i = 0; !*done; ++i) {
      if (yielding) sched_yield(); else __builtin_ia32_pause();
      if (yielding || (i & 255) == 255) {
        const double dt = now_us() - t0;
        if (dt > 2000.0) break;
        if (dt > 300.0) yielding = true;
      }
    }
    if (*done) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return true; }
  }
  return hipStreamSynchronize(stream) == hipSuccess;
}

// BTRAPZ_ACCEPT=reference (or its older spelling BTRAPZ_ELASTIC=0) turns the rescue pass of find_traj off -- the strict
// mode: a QP without a solution is a failure, as it is for the reference whenever OSQP declares infeasibility
// (trp_wrapper.cpp:191-200; INTEGRATION.md section 3 lists the bundled inputs on which the DEFAULT differs);
// BTRAPZ_ACCEPT=rescue is the default.  BTRAPZ_ELASTIC_TOL overrides btrapz_options.elastic_tol (largest accepted row
// violation / |g|, default 0.0125).
struct ElasticEnv { bool on; double tol; };
ElasticEnv elastic_env() {
  const char *e = getenv("BTRAPZ_ELASTIC"), *t = getenv("BTRAPZ_ELASTIC_TOL"), *acc = getenv("BTRAPZ_ACCEPT");
  ElasticEnv r = {!(e && *e == '0'), 0.0};
  if (acc && *acc) r.on = !(acc[0] == 'r' && acc[1] == 'e' && acc[2] == 'f');   // "reference": strict; anything else ("rescue"): the default
  if (t) { const double v = atof(t); if (v > 0) r.tol = v; }
  return r;
}

}  // namespace

BTRAPZ_EXPORT int btrapz_corridor_from_file(int variant, const char *input_path, btrapz_segment *out, int cap) {
  if (!input_path || !out || cap < 1 || variant < 0 || variant > 1) return BTRAPZ_EINVAL;
  TrajInput in;
  if (!read_traj_input(input_path, in)) return BTRAPZ_EINVAL;
  std::vector<std::vector<Segment>> lists;
  for (int o = 0; o < in.num_obs; o++) lists.push_back(extract_segments(variant, in.N, in.delta, in.s_bounds[o], in.l_bounds[o]));
  std::vector<Segment> seg;
  if (!select_segments(variant, in.delta, lists, in.s_ref, in.l_ref, seg)) return 0;
  const int n = (int)seg.size() < cap ? (int)seg.size() : cap;
  for (int k = 0; k < n; k++) {
    const Segment &c = seg[k];
    btrapz_segment &o = out[k];
    o.beg_t = c.beg_t; o.end_t = c.end_t; o.t = c.t; o.beg_l = c.beg_l; o.end_l = c.end_l;
    o.upp_skew = c.upp_skew; o.upp_bias = c.upp_bias; o.down_skew = c.down_skew; o.down_bias = c.down_bias;
    o.l_upp_skew = c.l_upp_skew; o.l_upp_bias = c.l_upp_bias; o.l_down_skew = c.l_down_skew; o.l_down_bias = c.l_down_bias;
    o.count = c.count;
  }
  return (int)seg.size();
}

namespace {

struct TrajResult {
  int S = 0, np = 0;
  std::vector<double> out;    // [6][np]: s, ds, dds, l, dl, ddl
  std::vector<double> ctrl;   // [12 S]
};

// Everything of find_traj between the parser and the output file.  Returns a_cost or the failure sentinel.
double run_find_traj(int variant, const TrajInput &in, const Params *p, TrajResult &res) {
  const double FAIL = BTRAPZ_FAIL_SENTINEL;
  const bool tm = timing();
  const double t_begin = tm ? now_us() : 0.0;
  t_last = LastCall();
  t_last.status = BTRAPZ_NO_CORRIDOR;   // until the solve says otherwise
  // corridor stage (host): trp_wrapper.cpp:176-188
  std::vector<std::vector<Segment>> lists;
  for (int o = 0; o < in.num_obs; o++) lists.push_back(extract_segments(variant, in.N, in.delta, in.s_bounds[o], in.l_bounds[o]));
  std::vector<Segment> seg;
  if (!select_segments(variant, in.delta, lists, in.s_ref, in.l_ref, seg)) return FAIL;
  const int S = (int)seg.size();
  const double t_corr = tm ? now_us() : 0.0;
  if (S < 1 || S > BTRAPZ_MAX_SEGMENTS_LONG) { fprintf(stderr, "btrapz: %d segments not supported (1..%d)\n", S, BTRAPZ_MAX_SEGMENTS_LONG); return FAIL; }
  const bool long_form = S > BTRAPZ_MAX_SEGMENTS;   // solved through the batched entry point, one workgroup per axis
  for (const Segment &c : seg) if (!(c.t > 0)) return FAIL;

  // batch record, B = 1 (layout: include/btrapz_hip.h)
  const int N = in.N;
  std::vector<double> h_seg((size_t)BTRAPZ_NUM_SEG_FIELDS * S), h_init(6), h_ref_end(2), h_dl(10);
  auto F = [&](int f, int k) -> double & { return h_seg[(size_t)f * S + k]; };
  for (int k = 0; k < S; k++) {
    const Segment &c = seg[k];
    F(BTRAPZ_F_T, k) = c.t;
    F(BTRAPZ_F_DOWN_BIAS, k) = c.down_bias; F(BTRAPZ_F_DOWN_SKEW, k) = c.down_skew;
    F(BTRAPZ_F_UPP_BIAS, k) = c.upp_bias; F(BTRAPZ_F_UPP_SKEW, k) = c.upp_skew;
    F(BTRAPZ_F_L_DOWN_BIAS, k) = c.l_down_bias; F(BTRAPZ_F_L_DOWN_SKEW, k) = c.l_down_skew;
    F(BTRAPZ_F_L_UPP_BIAS, k) = c.l_upp_bias; F(BTRAPZ_F_L_UPP_SKEW, k) = c.l_upp_skew;
    F(BTRAPZ_F_BEG_L, k) = c.beg_l; F(BTRAPZ_F_END_L, k) = c.end_l;
    double lo = 0.0, hi = 1000.0;  // solve_3d.cc:835-841
    for (int i = c.beg_t; i <= c.end_t; i++) {
      lo = std::fmax(in.ds_bounds[clampi(i, N - 1)].first, lo);
      hi = std::fmin(in.ds_bounds[clampi(i, N - 1)].second, hi);
    }
    F(BTRAPZ_F_DS_LO, k) = lo; F(BTRAPZ_F_DS_HI, k) = hi;
    const int i0 = clampi(10 * k, N - 1), i1 = clampi(10 * k + 1, N - 1);  // solve_3d.cc:1161-1165 (clamped)
    F(BTRAPZ_F_X_SKEW, k) = (in.s_ref[i1] - in.s_ref[i0]) / in.delta; F(BTRAPZ_F_X_BIAS, k) = in.s_ref[i0];
    F(BTRAPZ_F_Y_SKEW, k) = (in.l_ref[i1] - in.l_ref[i0]) / in.delta; F(BTRAPZ_F_Y_BIAS, k) = in.l_ref[i0];
  }
  for (int i = 0; i < 3; i++) { h_init[i] = in.init_s[i]; h_init[3 + i] = in.init_l[i]; }
  h_ref_end[0] = in.s_ref[N - 1]; h_ref_end[1] = in.l_ref[N - 1];
  for (int i = 0; i < 5; i++) { h_dl[2 * i] = in.dl_bounds[clampi(i, N - 1)].first; h_dl[2 * i + 1] = in.dl_bounds[clampi(i, N - 1)].second; }

  btrapz_shared sh = {};
  sh.w_s[0] = p->weight_s_ref; sh.w_s[1] = p->weight_ds_ref; sh.w_s[2] = p->s_acc_weight; sh.w_s[3] = p->s_jerk_weight;
  sh.w_l[0] = p->weight_l_ref; sh.w_l[1] = p->weight_dl_ref; sh.w_l[2] = p->l_acc_weight; sh.w_l[3] = p->l_jerk_weight;
  sh.weight_end_s = p->weight_end_s; sh.weight_end_l = p->weight_end_l;
  sh.ds_ref = in.ds_ref; sh.dl_ref = in.dl_ref;
  for (int i = 0; i < 2; i++) { sh.dds[i] = in.dds[i]; sh.ddds[i] = in.ddds[i]; sh.ddl[i] = in.ddl[i]; sh.dddl[i] = in.dddl[i]; }
  sh.delta = in.delta; sh.variant = variant;

  Caller *me = this_caller();
  if (!me) { fprintf(stderr, "btrapz: no HIP device available (this library has no CPU path)\n"); t_last.status = BTRAPZ_ENODEVICE; return FAIL; }
  btrapz_ctx *ctx = me->ctx;

  // expected sample count and the reference's CHECK_EQ(var_index, num_of_points_) (solve_3d.cc:1407)
  int np_expected = 1, var_index = 1;
  for (const Segment &c : seg) { np_expected = (int)((double)np_expected + c.t / in.delta); var_index += (int)(c.t / in.delta); }
  if (np_expected != var_index || np_expected < 1) return FAIL;
  const int max_points = np_expected;

  // the buffers below must belong to the context's device whatever the calling thread's current device is
  if (hipSetDevice(btrapz_ctx_device(ctx)) != hipSuccess) return FAIL;
  // One pinned host block, mapped into the device, laid out as doubles:
  //   in : seg[17 S] init[6] ref_end[2] dl[10] mqm[168]
  //   out: cost[1] status,iters[1] np[1] ctrl[12 S] traj[6 max_points]
  // First attempt: ONE launch that reads the inputs and writes the results through the mapping -- no copy calls, no
  // second and third launch (btrapz_launch_single).  Only a stalled solve takes the batched entry points below.
  const size_t n_in = (size_t)BTRAPZ_NUM_SEG_FIELDS * S + 6 + 2 + 10 + 168;
  const size_t n_out = 3 + (size_t)12 * S + (size_t)6 * max_points;
  if ((n_in + n_out) * 8 > me->pinned_bytes) {
    if (me->pinned) (void)hipHostFree(me->pinned);
    me->pinned = nullptr; me->pinned_dev = nullptr; me->pinned_bytes = 0;
    const size_t want = (n_in + n_out) * 8 * 2;
    if (hipHostMalloc(&me->pinned, want, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return FAIL;
    if (hipHostGetDevicePointer(&me->pinned_dev, me->pinned, 0) != hipSuccess) { (void)hipHostFree(me->pinned); me->pinned = nullptr; return FAIL; }
    me->pinned_bytes = want;
  }
  double *h_in = static_cast<double *>(me->pinned), *h_out = h_in + n_in;
  double *d_in = static_cast<double *>(me->pinned_dev), *d_out = d_in + n_in;
  std::copy(h_seg.begin(), h_seg.end(), h_in);
  double *p_init = h_in + (size_t)BTRAPZ_NUM_SEG_FIELDS * S;
  std::copy(h_init.begin(), h_init.end(), p_init);
  std::copy(h_ref_end.begin(), h_ref_end.end(), p_init + 6);
  std::copy(h_dl.begin(), h_dl.end(), p_init + 8);
  btrapz_mqm_table_host(&sh, p_init + 18);
  reinterpret_cast<volatile int *>(h_out + 2)[1] = 0;   // the kernel's completion word (wait_for_results)
  int h_status[2] = {0, 0}, h_np = 0;
  double h_cost = 0.0;
  const ElasticEnv el = elastic_env();
  // BTRAPZ_WARM=1 (a replanning loop: every call solves a problem close to the previous one's): start from the joint
  // states and multipliers the previous call of this thread left on the device.  The optimum does not depend on the
  // start -- the result is x* to solver accuracy either way -- but its last digits and the iteration count do, so it is
  // off by default: a call's output then depends on its inputs alone.
  const char *warm_env = getenv("BTRAPZ_WARM");
  const bool warm_on = warm_env && *warm_env && *warm_env != '0';
  t_last.status = BTRAPZ_EHIP;
  const double t_launch = tm ? now_us() : 0.0;
  btrapz_options opt1;
  btrapz_options_init(&opt1);
#ifdef BTRAPZ_EXPERIMENTS
  if (const char *mi = getenv("BTRAPZ_MAX_ITER")) opt1.max_iter = atoi(mi);   // (cost per iteration)
#endif
  // BTRAPZ_EPS: the solve's tolerance (btrapz_options.eps; default 1e-9, control points within ~2e-6 of x*).  The
  // reference's OSQP runs at 1e-5; 1e-6 saves about one iteration per call ("Tolerance").
  if (const char *ep = getenv("BTRAPZ_EPS")) { const double v = atof(ep); if (v > 0.0 && v < 1e-2) opt1.eps = v; }
  // BTRAPZ_SPLIT=0: the single launch in the three-candidates-per-wavefront form instead of the split one (find_traj's
  // only configuration channel is the environment; the library below it reads btrapz_options alone)
  if (const char *sp = getenv("BTRAPZ_SPLIT")) opt1.split = *sp == '0' ? -1 : 1;
  if (long_form) {
    h_status[0] = BTRAPZ_MAX_ITER_REACHED;   // (nothing solved yet: the block below does it, without the rescue rows)
  } else
  if (btrapz_launch_single(ctx, &sh, &opt1, S, d_in, d_out, max_points, warm_on ? 1 : 0, me->stream) != BTRAPZ_OK ||
      !wait_for_results(h_out, me->stream, S <= BTRAPZ_MAX_SEGMENTS)) {
    fprintf(stderr, "btrapz: %s\n", btrapz_last_error(ctx));
    btrapz_single_forget(ctx);   // (whatever the failed launch left is no start for the next call)
    return FAIL;
  }
  const double t_done = tm ? now_us() : 0.0;
  if (!long_form) {
  h_cost = h_out[0];
  memcpy(h_status, &h_out[1], 8);
  memcpy(&h_np, &h_out[2], 4);
  }
  t_last.iters = h_status[1]; t_last.status = h_status[0];
  if (tm) fprintf(stderr, "btrapz: timing [us]: corridor stage %.1f, record + table %.1f, launch to results %.1f (%d iterations)\n",
                  t_corr - t_begin, t_launch - t_corr, t_done - t_launch, h_status[1]);
  if (h_status[0] != BTRAPZ_SOLVED && h_status[0] != BTRAPZ_SOLVED_INACCURATE) btrapz_single_forget(ctx);
  if (long_form || (h_status[0] == BTRAPZ_MAX_ITER_REACHED && el.on)) {
    // Second attempt (no solution to converge to: a marginally infeasible corridor): the rescue pass of
    // btrapz_options.elastic, the counterpart of the reference accepting OSQP's status 2.  Rare, so it simply goes
    // through the batched entry points on a device copy of the inputs.
    if (verbose() && !long_form) fprintf(stderr, "btrapz: S=%d stalled after %d iterations, rescue pass\n", S, h_status[1]);
    const size_t n_dev = n_in + n_out + 1 + 4;   // (+ the four class violations behind the results)
    if (n_dev * 8 > me->scratch_bytes) {
      if (me->scratch) (void)hipFree(me->scratch);
      me->scratch = nullptr; me->scratch_bytes = 0;
      if (hipMalloc(&me->scratch, n_dev * 8 * 2) != hipSuccess) return FAIL;
      me->scratch_bytes = n_dev * 8 * 2;
    }
    double *s_in = static_cast<double *>(me->scratch), *s_out = s_in + n_in + 1;
    double *s_init = s_in + (size_t)BTRAPZ_NUM_SEG_FIELDS * S;
    long long *s_sel = reinterpret_cast<long long *>(s_in + n_in);
    int *s_status = reinterpret_cast<int *>(s_out + 1), *s_np = reinterpret_cast<int *>(s_out + 2);
    const long long sel0 = 0;
    memcpy(&h_in[n_in], &sel0, 8);   // (h_in[n_in] is h_out[0]: it stages the selection index until the results overwrite it)
    if (hipMemcpyAsync(s_in, h_in, (n_in + 1) * 8, hipMemcpyHostToDevice, me->stream) != hipSuccess) return FAIL;
    btrapz_options opt;
    btrapz_options_init(&opt);
    // (the long form has its rescue pass up to 192 segments; beyond, the plain solve decides)
    const bool rescue = el.on && (!long_form || S <= BTRAPZ_MAX_SEGMENTS_LONG_RESCUE);
    opt.elastic = rescue ? 1 : 0; opt.elastic_tol = el.tol;
    double *s_viol = s_out + n_out;
    double h_viol[4] = {0.0, 0.0, 0.0, 0.0};
    if (btrapz_solve_batch_device(ctx, &sh, &opt, 1, S, s_in, s_init, s_init + 6, s_init + 8, s_out + 3, s_out, s_status,
                                  s_status + 1, me->stream) != BTRAPZ_OK ||
        btrapz_sample_device(ctx, 1, S, in.delta, s_in, s_init, s_out + 3, 1, s_sel, max_points, s_out + 3 + 12 * S, s_np,
                             me->stream) != BTRAPZ_OK ||
        (rescue && btrapz_rescue_violations_device(ctx, 1, s_viol, me->stream) != BTRAPZ_OK)) {
      fprintf(stderr, "btrapz: %s\n", btrapz_last_error(ctx));
      t_last.status = BTRAPZ_EHIP;
      btrapz_single_forget(ctx);
      return FAIL;
    }
    if (hipMemcpyAsync(h_out, s_out, n_out * 8, hipMemcpyDeviceToHost, me->stream) != hipSuccess ||
        (rescue && hipMemcpyAsync(h_viol, s_viol, 4 * 8, hipMemcpyDeviceToHost, me->stream) != hipSuccess) ||
        hipStreamSynchronize(me->stream) != hipSuccess) { t_last.status = BTRAPZ_EHIP; btrapz_single_forget(ctx); return FAIL; }
    h_cost = h_out[0];
    memcpy(h_status, &h_out[1], 8);
    memcpy(&h_np, &h_out[2], 4);
    t_last.iters = h_status[1]; t_last.status = h_status[0];
    for (int i = 0; i < 4; i++) t_last.viol[i] = h_viol[i];
    if (verbose()) fprintf(stderr, "btrapz: rescue pass: status %d, row violations pos %.3g vel %.3g acc %.3g jerk %.3g\n", h_status[0],
                           h_viol[0], h_viol[1], h_viol[2], h_viol[3]);
  }
  std::vector<double> out(h_out + 3 + 12 * S, h_out + n_out);
  if (verbose()) fprintf(stderr, "btrapz: S=%d status=%d iters=%d obj=%.9g\n", S, h_status[0], h_status[1], h_cost);
  // acceptance: solve_3d.cc:1251-1277
  if (h_status[0] != BTRAPZ_SOLVED && h_status[0] != BTRAPZ_SOLVED_INACCURATE) return FAIL;
  if (h_np != max_points) { t_last.status = BTRAPZ_NO_CORRIDOR; return FAIL; }

  const double *s = &out[0], *ds = &out[(size_t)max_points], *dds = &out[(size_t)2 * max_points];
  const double *l = &out[(size_t)3 * max_points], *dl = &out[(size_t)4 * max_points], *ddl = &out[(size_t)5 * max_points];
  const double cost = trajectory_cost(variant, *p, in, max_points, s, ds, dds, l, dl, ddl);
  // A trajectory that is not finite is no trajectory (the reference refuses a solve whose objective is NaN,
  // solve_3d.cc:1251-1253): a corridor with infinite bounds assembles to rows and an objective of inf / NaN, and the solve
  // of such a problem may still end with a small score.  (Fuzzed: tests/fuzz/find_traj_vs_oracle.py, seeds 911 / 914.)
  // (The trajectory and the objective decide, not a_cost: with a reference line that is not finite a_cost is NaN for a
  //  perfectly good trajectory, and the reference returns that NaN -- trp_wrapper.cpp:217-286.)
  bool finite = std::isfinite(h_cost);
  for (size_t i = 0; finite && i < out.size(); i++) finite = std::isfinite(out[i]);
  if (!finite) { t_last.status = BTRAPZ_MAX_ITER_REACHED; return FAIL; }

  res.S = S; res.np = max_points; res.out = std::move(out);
  res.ctrl.assign(h_out + 3, h_out + 3 + 12 * S);
  return cost;
}

}  // namespace

BTRAPZ_EXPORT double btrapz_find_traj(int variant, const char *input_path, const char *output_path, const Params *p) {
  const double FAIL = BTRAPZ_FAIL_SENTINEL;
  if (!p || variant < 0 || variant > 1) return FAIL;
  std::string in_path = input_path ? input_path : "";
  if (in_path.empty()) { const char *e = getenv("BTRAPZ_INPUT"); in_path = e ? e : kDefaultInput[variant]; }
  std::string out_path = output_path ? output_path : "";
  if (out_path.empty()) {
    const char *e = getenv("BTRAPZ_OUTPUT_PREFIX");
    out_path = std::string(e ? e : kDefaultOutputPrefix[variant]) + std::to_string(p->iteration) + ".txt";
  }

  TrajInput in;
  if (!read_traj_input(in_path, in)) {
    fprintf(stderr, "btrapz: cannot read corridor file '%s'\n", in_path.c_str());
    return FAIL;
  }
  TrajResult res;
  const double cost = run_find_traj(variant, in, p, res);
  if (cost == FAIL) return FAIL;
  const int max_points = res.np;
  const double *s = &res.out[0], *ds = &res.out[(size_t)max_points], *dds = &res.out[(size_t)2 * max_points];
  const double *l = &res.out[(size_t)3 * max_points], *dl = &res.out[(size_t)4 * max_points], *ddl = &res.out[(size_t)5 * max_points];

  // trajectory file: trp_wrapper.cpp:288-301 (fixed, 3 decimals)
  if (!write_trajectory_file(out_path, max_points, in.delta, s, l, ds, dl, dds, ddl) && verbose())
    fprintf(stderr, "btrapz: cannot write '%s'\n", out_path.c_str());
  return cost;
}

// Test hooks (no device needed): the text scanner and writer of the file-based find_traj against strtod / printf
BTRAPZ_EXPORT double btrapz_debug_parse_double(const char *text, int *consumed) {
  const char *end = text;
  const double v = parse_double(text, &end);
  if (consumed) *consumed = (int)(end - text);
  return v;
}
BTRAPZ_EXPORT int btrapz_debug_format_fixed(double v, char *out336) { return format_3(out336, v); }

BTRAPZ_EXPORT int btrapz_find_traj_last_iterations(void) { return t_last.iters; }

BTRAPZ_EXPORT int btrapz_find_traj_last_status(double *viol) {
  if (viol) for (int i = 0; i < 4; i++) viol[i] = t_last.status == BTRAPZ_SOLVED_INACCURATE ? t_last.viol[i] : 0.0;
  return t_last.status;
}

// find_traj without the file side channel: the parsed content of the corridor file comes in as
// arrays, the trajectory goes out in full precision.
BTRAPZ_EXPORT double btrapz_find_traj_mem(int variant, const btrapz_traj_input *ti, const Params *p, int cap, double *traj,
                                       int *n_points, double *ctrl, int *n_segments) {
  return btrapz_find_traj_mem_cap(variant, ti, p, cap, traj, n_points, ctrl, ctrl ? 12 * BTRAPZ_MAX_SEGMENTS : 0, n_segments);
}

BTRAPZ_EXPORT double btrapz_find_traj_mem_cap(int variant, const btrapz_traj_input *ti, const Params *p, int cap, double *traj,
                                           int *n_points, double *ctrl, int ctrl_cap, int *n_segments) {
  const double FAIL = BTRAPZ_FAIL_SENTINEL;
  if (ctrl_cap < 0 || (ctrl_cap > 0 && !ctrl)) return FAIL;
  if (n_points) *n_points = 0;
  if (n_segments) *n_segments = 0;
  if (!ti || !p || variant < 0 || variant > 1 || ti->N < 3 || ti->N > 100000 || ti->num_obs < 0 || ti->num_obs > 1000 ||
      !(ti->delta > 0) || (ti->num_obs > 0 && (!ti->s_bounds || !ti->l_bounds)) || !ti->ds_bounds || !ti->dl_bounds ||
      !ti->s_ref || !ti->l_ref || cap < 0 || (cap > 0 && !traj))
    return FAIL;
  TrajInput in;
  in.N = ti->N; in.delta = ti->delta; in.num_obs = ti->num_obs;
  for (int i = 0; i < 3; i++) { in.init_s[i] = ti->init_s[i]; in.init_l[i] = ti->init_l[i]; }
  in.ds_ref = ti->ds_ref; in.dl_ref = ti->dl_ref;
  for (int i = 0; i < 2; i++) { in.dds[i] = ti->dds[i]; in.ddds[i] = ti->ddds[i]; in.ddl[i] = ti->ddl[i]; in.dddl[i] = ti->dddl[i]; }
  const int N = in.N;
  auto pairs = [N](const double *src, Bounds &b) { b.resize(N); for (int i = 0; i < N; i++) b[i] = {src[2 * i], src[2 * i + 1]}; };
  in.s_bounds.resize(in.num_obs); in.l_bounds.resize(in.num_obs);
  for (int o = 0; o < in.num_obs; o++) {
    pairs(ti->s_bounds + (size_t)o * N * 2, in.s_bounds[o]);
    pairs(ti->l_bounds + (size_t)o * N * 2, in.l_bounds[o]);
  }
  pairs(ti->ds_bounds, in.ds_bounds); pairs(ti->dl_bounds, in.dl_bounds);
  in.s_ref.assign(ti->s_ref, ti->s_ref + N); in.l_ref.assign(ti->l_ref, ti->l_ref + N);
  TrajResult res;
  const double cost = run_find_traj(variant, in, p, res);
  if (cost == FAIL) return FAIL;
  // A control-point buffer too small for 12 S values is filled as far as it goes -- which the caller can only notice
  // through *n_segments: without that pointer a truncated block would pass for the whole trajectory, so the
  // call fails instead.
  if (ctrl && (size_t)ctrl_cap < res.ctrl.size() && !n_segments) { t_last.status = BTRAPZ_EINVAL; return FAIL; }
  if (n_points) *n_points = res.np;
  if (n_segments) *n_segments = res.S;
  const int n = res.np < cap ? res.np : cap;
  static const int src_row[6] = {0, 3, 1, 4, 2, 5};   // file columns s l ds dl dds ddl <- rows s ds dds l dl ddl
  for (int i = 0; i < n; i++) {
    traj[i] = i * in.delta;
    for (int c = 0; c < 6; c++) traj[(size_t)(c + 1) * cap + i] = res.out[(size_t)src_row[c] * res.np + i];
  }
  if (ctrl) {   // s axis then l axis, 6 S each; a buffer too small for 12 S receives the first ctrl_cap of them
    const size_t n_ctrl = res.ctrl.size() < (size_t)ctrl_cap ? res.ctrl.size() : (size_t)ctrl_cap;
    for (size_t i = 0; i < n_ctrl; i++) ctrl[i] = res.ctrl[i];
  }
  return cost;
}
