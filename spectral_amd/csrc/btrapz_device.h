// btrapz_device.h -- types shared by the kernels and the host side of libbtrapz_hip.so.
#ifndef BTRAPZ_DEVICE_H
#define BTRAPZ_DEVICE_H

#include "../../include/btrapz_hip.h"

// The library is built with -fvisibility=hidden: only the C-ABI of include/btrapz_hip.h is exported.
#define BTRAPZ_EXPORT extern "C" __attribute__((visibility("default")))

// Bounds that are no bounds (btrapz_ipm.h, "bounds that are no bounds"): a finite bound of at least BTRAPZ_FAR in
// magnitude can never be active; the kernels move such row bounds to BTRAPZ_FAR_FACTOR (1 + the lane's largest real
// bound), the host moves such header limits (acceleration, jerk) to +-BTRAPZ_FAR_LIMIT; neither enters |bounds|.
#define BTRAPZ_FAR 1e9
#define BTRAPZ_FAR_FACTOR 1048576.0
#define BTRAPZ_FAR_LIMIT 1e6
#define BTRAPZ_COLD_FAR 1e4

namespace btrapz {

// Device view of btrapz_shared: limits already in the form the rows use them.
struct Shared {
  double w_s[4], w_l[4];
  double weight_end_s, weight_end_l;
  double ds_ref, dl_ref;
  double acc_s[2];   // dds clamped to [-1000,1000] (solve_3d.cc:836,843-844; ddx_bounds_ is uniform)
  double acc_l[2];   // ddy_bounds_[i] (uniform, solve_3d.cc:1019-1020)
  double jerk_s[2], jerk_l[2];
  int variant;
};

struct KernelArgs {
  int B, S;                 // S: segments per candidate (uniform mode) / unused (ragged mode)
  int seg_stride;           // k-stride owner: seg[f][b][k] has seg_stride slots per candidate (== S when uniform)
  // ragged mode (order != nullptr): candidates bucketed by segment count s = 1..64
  const int *order;         // [B] candidate ids, stable counting sort by s
  const int *seg_count;     // [B] s of every candidate (<= 0 or > seg_stride: skipped, status set by the bucket kernel)
  const int *cand_prefix;   // [66] candidates with count < s
  const int *wave_prefix;   // [66] wavefront pairs needed by buckets < s
  const double *seg;        // [NUM_SEG_FIELDS][B][S]
  const double *init;       // [B][6]
  const double *ref_end;    // [B][2]
  const double *dl_bounds;  // [B][10]
  const double *mqm;        // [2][4][21]  M' pQp_d M, packed upper triangle (solve_3d.cc:87-143)
  double *ctrl;             // [B][12 S]
  double *axis_obj;         // [2B]
  int *axis_status;         // [2B]
  int *axis_iters;          // [2B]
  double *axis_viol;        // [2B][4] rescue pass: row violations per class (may be null)
  Shared sh;
  double eps;
  double tau;               // fraction of the step to the boundary ...
  double tau_thr;           // ... when that step is at least this long; 0.995 of it otherwise
  int max_iter;
  int tau_iters;            // iterations (since the start) during which tau may be used
  int stall_start, stall_len; // divergence test: after stall_start iterations, stall_len without a better score
  float stall_factor;         // ... and without residuals smaller by this factor
  float diverge_factor;       // mu above this multiple of the best score: the iterate has left for good
  // warm start (btrapz_warm): all optional
  const double *x0;         // [B][2][seg_stride][3] joint states at the end of every segment
  const double *lam0;       // [2][36][B][seg_stride] multipliers of an earlier solve
  double *lam_out;          // same layout, multipliers at the end of this solve
  double mu0, smin;         // lambda = lam0 + mu0 / s , s = max(gap, smin)
  // capped first launch + resume launch (btrapz_options.cap_iter): see CAPPED / RESUME in btrapz_kernels.hip
  int cap_iter;             // iterations after which a group of the capped launch may suspend (0: no cap) ...
  int cap_alone;            // ... when at most this many groups of its wavefront are still iterating,
  int cap_hi;               // and after which it suspends in any case
  double cap_score;         // a lone group whose KKT score is already below this stays: it is one or two iterations from done
  int susp_cap;             // slots of susp_state
  double *susp_state;       // [susp_cap][SUSP_FIELDS][seg_stride]
  int *susp_count;          // [1] slots handed out
  int *susp_slot;           // [2B] slot of a suspended axis problem
  int *susp_key;            // [2][B] convergence class of a suspended axis problem (key of the resume lists)
  int unc_start;            // btrapz_options.start: 1 = first one Newton step of the problem without its inequality rows
  int bucket_S;             // 0: the bucket id IS the segment count (ragged); else buckets are hint classes, S = bucket_S
  // rescue pass (btrapz_options.elastic): order = [2][B] per-axis lists, cand_prefix / wave_prefix = [2][198] tables
  double elastic_delta;     // penalty d^2 / (2 delta) on the relaxation d of every inequality row
  double elastic_tol;       // largest row violation still reported as BTRAPZ_SOLVED_INACCURATE
  int *queue;               // ipm_solve_queue_kernel: [2] next candidate per axis (zeroed before the launch)
  double *x_out;            // warm-start instantiations: joint states of the returned iterate, layout of x0 (may be null)
};

struct PrismRoad {              // btrapz_road in the form the prism stage uses it (prism_core.h)
  double rate;                  // knots per second (the reference hard-codes 10: `i/10`, `t0*10`)
  double s_lo, s_hi, l_lo, l_hi, l_safe, w_safe;
};

struct CorridorArgs {
  int B, N, num_obs, variant, seg_stride;
  double delta;
  const double *s_bounds, *l_bounds;   // [B][num_obs][N][2]; unused by the fused prism + corridor kernels
  // fused prism + corridor kernels (btrapz_prism_corridor_batch_device): the strips are evaluated from the prisms
  const double *prisms;                // [B][P][8]
  int P;
  PrismRoad road;
  int *n_strips;                       // [B]
  const double *ds_bounds, *dl_bounds; // [B][N][2]
  const double *s_ref, *l_ref;         // [B][N]
  double *seg;                         // [NUM_SEG_FIELDS][B][seg_stride]
  int *seg_count;                      // [B]
  double *ref_end;                     // [B][2]
  double *dl10;                        // [B][10]
  // segment-list capacities of this launch and the retry list (see corridor_kernels.hip)
  int cap_o, cap_sel;                  // segments per obstacle, selected segments
  int pass;                            // 0: one workgroup per candidate; 1: workgroups loop over retry_list
  int *retry_list, *retry_count;       // candidates whose lists overflowed in pass 0
};

__global__ void corridor_batch_kernel(const CorridorArgs a, int staged);
__global__ void corridor_batch_short_kernel(const CorridorArgs a, int staged);   // N <= 128
__global__ void corridor_first_kernel(const CorridorArgs a, int staged);         // first pass of a two-pass launch: no serial statement
__global__ void corridor_first_short_kernel(const CorridorArgs a, int staged);
__global__ void prism_corridor_batch_kernel(const CorridorArgs a, int staged);   // prisms -> strips -> corridors in one launch
__global__ void prism_corridor_batch_short_kernel(const CorridorArgs a, int staged);
__global__ void prism_corridor_first_kernel(const CorridorArgs a, int staged);
__global__ void prism_corridor_first_short_kernel(const CorridorArgs a, int staged);
struct Seg;
__global__ void corridor_serial_kernel(const CorridorArgs a, Seg *ws_all, Seg *ws_sel);   // beyond the wave-wide kernels' limits: one lane per candidate
// fixed_S = 0: bucket by segment count (ragged batches); > 0: uniform batch of fixed_S segments, bucket by hint class
// candidates that cannot start: keys[b] = 0 and their records written here (btrapz_options.compact; btrapz_kernels.hip)
__global__ void prestart_kernel(const KernelArgs a, int S_uniform, const int *seg_count, int *keys);
__global__ void bucket_hist_kernel(int B, int seg_stride, const int *seg_count, int *meta, int fixed_S);
__global__ void bucket_prefix_kernel(int *meta, int fixed_S);
__global__ void bucket_scatter_kernel(int B, int seg_stride, const int *seg_count, int *meta, int *order,
                                      double *axis_obj, int *axis_status, int *axis_iters, int fixed_S);
__global__ void ipm_solve_kernel(const KernelArgs a, const double *__restrict__ mqm);
__global__ void ipm_solve_ordered_kernel(const KernelArgs a, const double *__restrict__ mqm);       // through a.order
__global__ void ipm_solve_warm_kernel(const KernelArgs a, const double *__restrict__ mqm);          // + btrapz_warm
__global__ void ipm_solve_warm_ordered_kernel(const KernelArgs a, const double *__restrict__ mqm);
#ifdef BTRAPZ_EXPERIMENTS
__global__ void ipm_solve_queue_kernel(const KernelArgs a, const double *__restrict__ mqm);         // persistent, candidate queue
#endif
__global__ void ipm_solve_elastic_kernel(const KernelArgs a, const double *__restrict__ mqm);       // rescue pass
__global__ void ipm_solve_split_kernel(const KernelArgs a, const double *__restrict__ mqm);         // one candidate per wavefront, rows over 3 lanes
__global__ void ipm_solve_capped_kernel(const KernelArgs a, const double *__restrict__ mqm);        // first launch of a capped solve
__global__ void ipm_solve_capped_ordered_kernel(const KernelArgs a, const double *__restrict__ mqm);  // ... of a ragged batch
__global__ void ipm_solve_resume_kernel(const KernelArgs a, const double *__restrict__ mqm);        // ... and the launch that carries the suspended problems on
// the solve at two wavefronts per SIMD (btrapz_lean_body.h; instantiated in btrapz_lean.hip and btrapz_lean_warm.hip)
__global__ void ipm_solve_lean_kernel(const KernelArgs a, const double *__restrict__ mqm);
__global__ void ipm_solve_lean_ordered_kernel(const KernelArgs a, const double *__restrict__ mqm);
__global__ void ipm_solve_lean_capped_kernel(const KernelArgs a, const double *__restrict__ mqm);
__global__ void ipm_solve_lean_capped_ordered_kernel(const KernelArgs a, const double *__restrict__ mqm);
__global__ void ipm_solve_lean_resume_kernel(const KernelArgs a, const double *__restrict__ mqm);
__global__ void ipm_solve_lean_warm_kernel(const KernelArgs a, const double *__restrict__ mqm);
__global__ void ipm_solve_lean_warm_ordered_kernel(const KernelArgs a, const double *__restrict__ mqm);
#define BTRAPZ_SUSPENDED (-7)   // internal: an axis problem the capped launch handed over (never leaves the library)
__global__ void ipm_solve_long_kernel(const KernelArgs a, const double *__restrict__ mqm);          // 65..256 segments: one axis problem per workgroup
__global__ void ipm_solve_long_elastic_kernel(const KernelArgs a, const double *__restrict__ mqm);  // rescue pass of the long form (<= 192 segments)
__global__ void rescue_keys_kernel(int B, int S, const int *seg_count, const int *axis_status, int *keys, int all);
__global__ void rescue_init_kernel(int B, int S, const int *seg_count, double *axis_obj, int *axis_status, int *axis_iters);
struct MqmWeights { double w[2][4]; };   // [axis][ref, dref, acc, jerk]
__global__ void mqm_table_kernel(MqmWeights w, double *out);
__global__ void rescue_violations_kernel(int B, const double *axis_viol, double *viol);
__global__ void finalize_kernel(int B, const double *axis_obj, const int *axis_status, const int *axis_iters,
                                double *cost, int *status, int *iters);
__global__ void argmin_kernel(int group, long long index_base, const double *cost, long long *best_idx,
                              double *best_cost, double *part_cost, long long *part_idx);
__global__ void argmin_final_kernel(int chunks, long long index_base, const double *part_cost, const long long *part_idx,
                                    long long *best_idx, double *best_cost);
__global__ void argmin_pairs_kernel(int world, int n, const long long *pairs, double *best_cost, long long *best_idx);
__global__ void eval_states_kernel(int B, int seg_stride, const int *seg_count, const double *seg, const double *ctrl,
                                   int n_times, const double *times, double *x);
__global__ void sample_kernel(int B, int seg_stride, const int *seg_count, double delta, const double *seg,
                              const double *init, const double *ctrl, int nsel, const long long *sel, int max_points,
                              double *out, int *npoints);

__global__ void single_candidate_kernel(const KernelArgs a, const double *__restrict__ mqm, double delta, int max_points,
                                        double *out);
__global__ void single_candidate_split_kernel(const KernelArgs a, const double *__restrict__ mqm, double delta, int max_points,
                                              double *out);
__global__ void single_candidate_warm_split_kernel(const KernelArgs a, const double *__restrict__ mqm, double delta, int max_points,
                                                   double *out);
__global__ void single_candidate_warm_kernel(const KernelArgs a, const double *__restrict__ mqm, double delta, int max_points,
                                             double *out);

}  // namespace btrapz

// library-internal: the HIP device a context lives on
int btrapz_ctx_device(const btrapz_ctx *ctx);
// library-internal: find_traj's single-candidate path.  One launch; in and out may be host memory
// mapped into the device.  in: seg[17 S] init[6] ref_end[2] dl[10] mqm[168]; out: cost, status|iters, np, ctrl[12 S],
// traj[6 max_points].  The M'QM table comes from the caller (btrapz_mqm_table_host).
// warm: 0 cold; 1 start from what the previous warm call of this context left (same variant and segment count, else
// cold) and leave this call's joint states and multipliers for the next one.
int btrapz_launch_single(btrapz_ctx *ctx, const btrapz_shared *shared, const btrapz_options *opt, int S, const double *in,
                         double *out, int max_points, int warm, void *stream);
void btrapz_single_forget(btrapz_ctx *ctx);   // the next warm call starts cold
void btrapz_mqm_table_host(const btrapz_shared *shared, double *table /* [2][4][21] */);
#endif
