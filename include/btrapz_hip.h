/*
 * btrapz_hip.h -- C-ABI of the MI355X-native Bezier-in-corridor trajectory QP path.
 *
 * Two groups of entry points:
 *
 *  (1) `find_traj` -- the reference's ONLY exported symbol, unchanged:
 *        extern "C" double find_traj(Params *p);
 *      replaces  src/trp_wrapper.cpp:16-20 (libtrp.so),
 *                src/cub_wrapper.cpp:16-19 (libcub.so),
 *                the older libbtrapz.so build (same symbol; strings: c1.txt -> slt_3d.txt).
 *      `Params` is include/btrapz/py_cpp_.h:6-21 == src/trp_wrapper.py:19-32 (88 bytes).
 *      libtrp.so / libcub.so / libbtrapz.so built from this repo export exactly that
 *      symbol and forward to btrapz_find_traj() below with the library's variant and
 *      default paths.  The reference's hard-coded file paths are kept as defaults and
 *      can be redirected with BTRAPZ_INPUT / BTRAPZ_OUTPUT_PREFIX.
 *
 *  (2) the batched entry points (no reference counterpart: the reference solves one
 *      QP per call, src/solve_3d.cc:1231-1414).  They replace, for B candidates at
 *      once, FormulateProblem + osqp_setup + osqp_solve (src/solve_3d.cc:1143-1229,
 *      1246, 1249; src/cuboid_3d.cc:632-988) and the acceptance test (:1251-1277).
 *
 * Plain C: opaque handle, POD structs, raw pointers and sizes.  No torch types.
 * Every function returns 0 on success or a negative BTRAPZ_E* code; it never aborts
 * the host process (the reference's CHECK_* do: include/btrapz/logging.h:256-258).
 */
#ifndef BTRAPZ_HIP_H
#define BTRAPZ_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- reference ABI: include/btrapz/py_cpp_.h:6-21 ------------------------------ */
typedef struct Params {
  double s_acc_weight;
  double s_jerk_weight;
  double l_acc_weight;
  double l_jerk_weight;
  double weight_s_ref;
  double weight_ds_ref;
  double weight_l_ref;
  double weight_dl_ref;
  double weight_end_s;
  double weight_end_l;
  int iteration;
} Params;

#define BTRAPZ_FAIL_SENTINEL 100000000000.0 /* src/trp_wrapper.cpp:199 */

enum { BTRAPZ_TRAPEZOID = 0, /* src/solve_3d.cc   */
       BTRAPZ_CUBOID = 1 /* src/cuboid_3d.cc  */ };

/* find_traj with the library-specific constants made explicit.
 * variant: BTRAPZ_TRAPEZOID (trp_wrapper.cpp) or BTRAPZ_CUBOID (cub_wrapper.cpp).
 * input_path / output_path: NULL -> environment override -> reference default.
 * Returns a_cost (trp_wrapper.cpp:217-286 / cub_wrapper.cpp:201-262) or 1e11. */
double btrapz_find_traj(int variant, const char *input_path, const char *output_path,
                        const Params *p);

/* find_traj without the file side channel.  The reference hands its input over as a text
 * file whose path is compiled into the library (trp_wrapper.cpp:23, written by cart_frenet.py:384-453) and
 * returns the trajectory as a 3-decimal text file (trp_wrapper.cpp:288-301).  Here the same content comes in
 * as arrays -- the grammar of trp_wrapper.cpp:39-144, field for field -- and the trajectory goes out in full
 * precision.  Same computation, same return value (a_cost or 1e11) as btrapz_find_traj.
 *   traj [7][cap]: rows t, s, l, ds, dl, dds, ddl (the columns of the reference's output file); at most cap
 *                  samples are written, *n_points receives the trajectory's sample count.
 *   ctrl [12*BTRAPZ_MAX_SEGMENTS] (may be NULL): control points, s axis (6 S) then l axis (6 S); *n_segments receives
 *                  S.  A horizon of more than 64 segments (solved since round 3, up to BTRAPZ_MAX_SEGMENTS_LONG) does
 *                  not fit this buffer: the first 12*64 values are written, S is reported, and the caller that wants
 *                  them all calls btrapz_find_traj_mem_cap with a buffer of 12 S (ctrl_cap: its size in doubles).
 *                  CHECK *n_segments: 12 * S above the buffer's size means the block is truncated.  A call that passes
 *                  ctrl but no n_segments cannot notice, so it FAILS (1e11) when the block does not fit. */
typedef struct btrapz_traj_input {
  int N, num_obs;
  double delta;
  double init_s[3], init_l[3];
  double ds_ref, dl_ref;
  double dds[2], ddds[2], ddl[2], dddl[2];
  const double *s_bounds;  /* [num_obs][N][2] (lower, upper) per knot */
  const double *l_bounds;  /* [num_obs][N][2] */
  const double *ds_bounds; /* [N][2] */
  const double *dl_bounds; /* [N][2] */
  const double *s_ref;     /* [N] */
  const double *l_ref;     /* [N] */
} btrapz_traj_input;
double btrapz_find_traj_mem(int variant, const btrapz_traj_input *in, const Params *p, int cap,
                            double *traj, int *n_points, double *ctrl, int *n_segments);
double btrapz_find_traj_mem_cap(int variant, const btrapz_traj_input *in, const Params *p, int cap,
                                double *traj, int *n_points, double *ctrl, int ctrl_cap, int *n_segments);

/* Interior-point iterations of the calling thread's last find_traj / btrapz_find_traj_mem call (-1: none yet).
 * Diagnostics: with BTRAPZ_WARM=1 a call starts from the joint states and multipliers the thread's previous call
 * left on the device (the replanning loop cart_frenet.py:1516-1571 solves a problem close to the previous one at every
 * step), which shows here as fewer iterations; the result is x* to solver accuracy either way. */
int btrapz_find_traj_last_iterations(void);
/* Status of the calling thread's last find_traj / btrapz_find_traj_mem call, and what tells a rescued result from an
 * exact one: viol[4] (if not NULL) receives the largest violation of a position / velocity / acceleration / jerk row by
 * the returned trajectory, in the rows' own units.
 *   BTRAPZ_SOLVED (1), or BTRAPZ_SOLVED_INACCURATE (2) with viol all zero: the trajectory is the QP's optimum (2: the
 *     solve ended at its round-off floor, KKT score between 1e-7 and 1e-5);
 *   BTRAPZ_SOLVED_INACCURATE (2) with a non-zero viol: the QP has no solution, the trajectory is the rescue pass's
 *     least-violation one (btrapz_options.elastic) and leaves its rows by viol;
 *   a negative status: the call returned 1e11.  0: no call yet.
 * The reference's find_traj cannot tell these apart (it returns a_cost for OSQP's status 1 and 2 alike). */
int btrapz_find_traj_last_status(double *viol);

/* Host-side corridor stage of find_traj alone (CorridorGeneration + CorridorSplit per obstacle,
 * then CollisionCheck: src/solve_3d.cc:323-486,729-772,488-714 ; src/cuboid_3d.cc:301-573).
 * Parses input_path, writes up to cap segments.  Returns the segment count S >= 1, 0 when no
 * segment survives the selection, or a negative BTRAPZ_E* code.  Needs no GPU. */
typedef struct btrapz_segment { /* == Cube, include/btrapz/cube_type.h:2-24 */
  int beg_t, end_t;
  double t;
  double beg_l, end_l;
  double upp_skew, upp_bias, down_skew, down_bias;
  double l_upp_skew, l_upp_bias, l_down_skew, l_down_bias;
  int count;
} btrapz_segment;
int btrapz_corridor_from_file(int variant, const char *input_path, btrapz_segment *out, int cap);

/* ---- error codes ----------------------------------------------------------------- */
enum {
  BTRAPZ_OK = 0,
  BTRAPZ_EINVAL = -1,   /* bad argument (B<1, S<1 or S beyond the limits above, null pointer) */
  BTRAPZ_ENODEVICE = -2, /* no HIP device / HIP runtime error at create */
  BTRAPZ_EHIP = -3,     /* HIP runtime error during a call (see btrapz_last_error) */
  BTRAPZ_ENOMEM = -4
};

/* ---- per-candidate status: OSQP status_val vocabulary (solve_3d.cc:1251-1253) ------ */
enum {
  BTRAPZ_SOLVED = 1,
  BTRAPZ_SOLVED_INACCURATE = 2,
  BTRAPZ_MAX_ITER_REACHED = -2,  /* no optimum reached: iteration limit; a solve that stalls or diverges (what an
                                    infeasible corridor does); an initial state outside segment 0's rows or a
                                    joint whose two sides share no value (reported before the first iteration);
                                    the rescue pass (btrapz_options.elastic) takes these over */
  BTRAPZ_PRIMAL_INFEASIBLE = -3, /* a row with l > u (e.g. an empty inscribed interval of the cuboid variant), or a
                                    segment with t <= 0; after a rescue pass: violation above elastic_tol */
  BTRAPZ_NO_CORRIDOR = -5 /* ragged batches: the corridor stage selected no segment, more than
                             seg_stride / 64 segments, or a segment with t <= 0 (the reference's
                             find_traj fails or aborts on these: solve_3d.cc:617, :1407) */
};

/* ---- batch layout ------------------------------------------------------------------
 * seg[f][b][k]: field-major SoA, float64, f in [0,BTRAPZ_NUM_SEG_FIELDS), b candidate,
 * k segment.  Fields 0-10 are the reference's Cube (include/btrapz/cube_type.h:2-24);
 * 11-12 the per-segment ds bounds of solve_3d.cc:835-845; 13-16 the piecewise-linear
 * reference of solve_3d.cc:1159-1166.                                               */
enum {
  BTRAPZ_F_T = 0,
  BTRAPZ_F_DOWN_BIAS, BTRAPZ_F_DOWN_SKEW, BTRAPZ_F_UPP_BIAS, BTRAPZ_F_UPP_SKEW,
  BTRAPZ_F_L_DOWN_BIAS, BTRAPZ_F_L_DOWN_SKEW, BTRAPZ_F_L_UPP_BIAS, BTRAPZ_F_L_UPP_SKEW,
  BTRAPZ_F_BEG_L, BTRAPZ_F_END_L,
  BTRAPZ_F_DS_LO, BTRAPZ_F_DS_HI,
  BTRAPZ_F_X_SKEW, BTRAPZ_F_X_BIAS, BTRAPZ_F_Y_SKEW, BTRAPZ_F_Y_BIAS,
  BTRAPZ_NUM_SEG_FIELDS
};
#define BTRAPZ_MAX_SEGMENTS 64
/* Uniform cold batches (btrapz_solve_batch_device / _host) and find_traj take up to this many segments per candidate:
 * beyond 64 a candidate is solved by a workgroup of several wavefronts (same method, same result, much slower per
 * segment -- the reference has no limit, its bundled inputs have at most 14).  The rescue pass (btrapz_options.elastic
 * = 1) follows up to BTRAPZ_MAX_SEGMENTS_LONG_RESCUE segments (three wavefronts: a fourth's LDS does not fit);
 * ragged batches, warm starts and elastic = 2 stay at BTRAPZ_MAX_SEGMENTS. */
#define BTRAPZ_MAX_SEGMENTS_LONG 256
#define BTRAPZ_MAX_SEGMENTS_LONG_RESCUE 192

/* Weights (Params) and limits (input-file header, trp_wrapper.cpp:59-64) shared by all
 * candidates of a batch. */
typedef struct btrapz_shared {
  double w_s[4]; /* weight_s_ref, weight_ds_ref, s_acc_weight, s_jerk_weight */
  double w_l[4]; /* weight_l_ref, weight_dl_ref, l_acc_weight, l_jerk_weight */
  double weight_end_s, weight_end_l;
  double ds_ref, dl_ref;
  double dds[2], ddds[2], ddl[2], dddl[2];
  double delta;
  int variant;
  int reserved;
} btrapz_shared;

typedef struct btrapz_options {
  int struct_size;       /* sizeof(btrapz_options) of the caller's build: set by btrapz_options_init().  A value the
                            library does not know is rejected (BTRAPZ_EINVAL) instead of being read past its end. */
  int max_iter;          /* interior-point iterations; 0 -> default (60) */
  double eps;            /* KKT score target; 0 -> default (1e-9) */
  double step_fraction;  /* fraction of the step to the boundary taken per iteration, in (0,1); 0 -> default (0.9999) */
  double step_threshold; /* the fraction above is used only when the step to the boundary is at least this long
                            (blocked steps, and every step after the 12th iteration, take 0.995 of it, which
                            keeps the iterates centred); 0 -> default (0.9) */
  /* Rescue of stalled candidates (acceptance, solve_3d.cc:1251-1277 / trp_wrapper.cpp:191-200).  The reference
   * accepts OSQP's status 2: on a marginally infeasible corridor (src/c7.txt) that is an ADMM iterate which stays
   * within 1e-4 m of its position rows and violates acceleration rows by 0.49, and find_traj returns a trajectory.
   * An exact method has no such iterate -- the QP has no solution -- so the equivalent here is explicit: an axis
   * problem whose interior-point solve stalls is solved again with every inequality row relaxed, l <= g'x - d <= u,
   * and sum (d / |g|)^2 / (2 elastic_delta) added to the objective (equalities -- continuity, initial state -- stay
   * exact): the least-squares violation of the rows, each measured in its own norm |g| (t for a position row t c_i,
   * sqrt(50) for a velocity row, sqrt(2400) for an acceleration row, sqrt(72000) for a jerk row: the distance the
   * control points would have to move), the reference's objective deciding among its minimisers.
   *   elastic        0: off (stalled candidates keep status -2, cost +inf) -- default of the batched entry points;
   *                  1: rescue pass over the stalled axis problems after the solve; 2: every candidate is solved
   *                  with elastic rows straight away.  find_traj uses 1 (BTRAPZ_ACCEPT=reference -- older spelling BTRAPZ_ELASTIC=0 -- turns it off: the strict mode).
   *   elastic_tol    a rescued problem whose largest row violation / |g| is at most this is reported as
   *                  BTRAPZ_SOLVED_INACCURATE (2), beyond it as BTRAPZ_PRIMAL_INFEASIBLE (-3); 0 -> default (0.0125:
   *                  at most 0.0125 t metres outside a position row, 0.088 on a velocity row, 0.61 on an acceleration
   *                  row -- the reference's accepted iterate on src/c7.txt is at 0.492 = 0.01004 |g| -- 3.4 on a jerk row)
   *   elastic_delta  0 -> default (1e-8) */
  int elastic;
  double elastic_tol;
  double elastic_delta;
  /* EXPERIMENTAL, honoured by a -DBTRAPZ_EXPERIMENTS build only; the shipped library has no such kernel and answers
   * queue > 0 with BTRAPZ_EINVAL (btrapz_build_has_experiments() says which build a caller holds): persistent wavefronts
   * that draw candidates from a queue.  Scheduling only.  Measured against the two launches of cap_iter: a loss on every
   * bench batch.  The field stays so that the struct's layout does. */
  int queue;
  /* Few candidates: the split form of the solve kernel -- ONE candidate per wavefront, every segment's rows spread over
   * three lanes (uniform cold batches of at most 21 segments).  Per solve it takes ~0.7 of the time of the
   * three-candidates-per-wavefront form and 2-3 times its machine share, so it pays while the batch leaves SIMDs idle.
   * 0 -> automatic (used when 2 B wavefronts fit the device's SIMDs at once); 1 -> whenever the batch qualifies;
   * -1 -> never.  Same problem, same method: results agree to rounding (the row sums are taken in another order). */
  int split;
  /* EXPERIMENTAL, honoured by a -DBTRAPZ_EXPERIMENTS build only (the shipped library answers start != 0 with
   * BTRAPZ_EINVAL): 1 = start a cold solve
   * from the unconstrained optimum (one Newton step of the problem without its inequality rows) instead of the initial
   * state propagated at constant velocity.  The optimum does not depend on it; the iteration count does, for the worse
   * (DESIGN.md 3.2).  The field stays so that the struct's layout does. */
  int start;
  /* Uniform cold batches of many more candidates than the device holds at once: two launches instead of one.  In the
   * first, a candidate that is the only one of its wavefront still iterating after cap_iter interior-point iterations
   * (or still iterating four iterations later) hands its iterate over; the second launch carries those candidates on from
   * exactly where they stopped, like with like and the far-from-converged first -- same iterates, same results bit for
   * bit, but no wavefront runs at a third of its width for one slow candidate and no slow candidate starts last
   * (-9 % on the scenario_1 bench batch).  0 -> automatic (6 for uniform batches of 16..64 segments
   * that fill the device at least eight times over; never for ragged batches); > 0 -> that many iterations, any cold
   * batch -- a ragged one too (btrapz_solve_ragged_device: its second launch re-packs by segment count; it pays where
   * iteration counts spread widely, e.g. a quarter of the candidates infeasible: -9 %, and costs up to 10 % where
   * they do not); -1 -> never. */
  int cap_iter;
  /* Cold solves of at most 64 segments (uniform or ragged batches, one launch or the two of cap_iter): the form of the
   * solve kernel that runs TWO wavefronts per SIMD -- the same iteration on half the per-lane state (slacks in registers,
   * multipliers in LDS, everything else recomputed: btrapz_lean.hip).  Same problem, same method, same termination rules;
   * results agree with the one-wavefront form to rounding.  0 -> automatic (batches of at least 1.25 wavefronts per
   * SIMD: below that the one-wavefront form's shorter instruction stream is faster); 1 -> whenever the solve qualifies;
   * -1 -> never.  (The rescue pass, start = 1 and the candidate queue always run the one-wavefront form; warm starts --
   * btrapz_solve_warm_device -- have their instantiation of this form too, in one launch.) */
  int lean;
  /* Cold solves of at most 64 segments without a rescue pass: a pre-pass lists the candidates that CAN start -- it
   * applies, with doubled tolerances, the tests of the solve kernels' set-up: a row with l > u (e.g. the empty inscribed
   * interval of a cuboid corridor, cuboid_3d.cc:677-689), an initial state outside segment 0's rows, a joint whose two
   * sides share no value -- and the solve launch holds those only: no wavefront waits with an idle group, and the other
   * axis of a candidate that is lost anyway is not solved.  Scheduling only for the candidates that are solved (same
   * kernels, same bits); a candidate the pre-pass drops gets the status its solve would have reported, cost +inf, iters
   * 0, and its control points are left untouched.  0 -> automatic (large ragged batches and large batches of the cuboid
   * variant, where a quarter of the candidates of the bench workloads cannot start: -20 % time; elsewhere the pre-pass
   * costs 1-2 %); 1 -> always; -1 -> never. */
  int compact;
} btrapz_options;
/* Zeroes *opt (every field: "use the default") and sets struct_size.  Call it before filling the struct in. */
void btrapz_options_init(btrapz_options *opt);

/* A context owns the per-launch workspace of one device: it is NOT thread-safe (one context per calling
 * thread; launches of one context must be issued in stream order). */
typedef struct btrapz_ctx btrapz_ctx;

int btrapz_create(btrapz_ctx **ctx, int device);
int btrapz_destroy(btrapz_ctx *ctx);
const char *btrapz_last_error(const btrapz_ctx *ctx);
/* Device memory (bytes) the context holds for its launches right now: it grows on demand and is kept.  The largest part is
 * the hand-over workspace of the two-launch solve (btrapz_options.cap_iter): slots for 15 % of the axis problems, 74 doubles
 * per segment each, at most 1 GiB. */
long long btrapz_workspace_bytes(const btrapz_ctx *ctx);
/* Number of HIP devices visible (0 when the runtime finds none). */
int btrapz_device_count(void);
/* 1 when the library was built with -DBTRAPZ_EXPERIMENTS (btrapz_options.queue / .start honoured, environment overrides of
 * the scheduling choices read), else 0: the shipped build. */
int btrapz_build_has_experiments(void);

/* Solve B candidates of S segments each; all pointers are DEVICE pointers.
 *   seg        [NUM_SEG_FIELDS][B][S]     init     [B][6] (s, ds, dds, l, dl, ddl at t=0)
 *   ref_end    [B][2]  x_ref[N-1], y_ref[N-1]  (solve_3d.cc:268,315)
 *   dl_bounds  [B][10] dy_bounds_[i], i=0..4, as lo,hi pairs (solve_3d.cc:1003-1004)
 * outputs
 *   ctrl   [B][12*S] control points in the reference's variable order (solve_3d.cc:1343-1344)
 *   cost   [B] OSQP-style obj_val 0.5 x'Px + q'x ; +inf when status is not 1/2
 *   status [B] ; iters [B] (may be NULL)
 * stream: a hipStream_t (NULL = default stream).  Asynchronous: returns after launch. */
int btrapz_solve_batch_device(btrapz_ctx *ctx, const btrapz_shared *shared,
                              const btrapz_options *opt, int B, int S, const double *seg,
                              const double *init, const double *ref_end,
                              const double *dl_bounds, double *ctrl, double *cost,
                              int *status, int *iters, void *stream);

/* Row violations of the candidates the context's LAST solve with btrapz_options.elastic != 0 handed to the rescue pass
 * (0 for every other candidate): viol [B][4] = largest violation of a position / velocity / acceleration / jerk row
 * by the returned control points, in the rows' own units (t c_i in metres, 5 (c_i+1 - c_i), 20 (c_i - 2 c_i+1 + c_i+2),
 * 60 (...): solve_3d.cc:823-888), the larger of the two axes.  A caller that must not leave the corridor by more than
 * its own margin tests viol[b][0].  Device pointer; stream order. */
int btrapz_rescue_violations_device(btrapz_ctx *ctx, int B, double *viol, void *stream);

/* Which form of the solve kernel the context's last batched solve ran (scheduling only; results do not depend on it):
 * 0 packed (floor(64/S) candidates per wavefront), 1 split (btrapz_options.split), 2 long (65..256 segments),
 * 3 capped first launch + resume launch (btrapz_options.cap_iter), 4 candidate queue (experiment builds); -1 none yet.
 * + 8 when the launch(es) ran the two-wavefronts-per-SIMD form (btrapz_options.lean); + 16 when the long form solved the
 * candidates of more than 64 segments of a ragged batch beside it. */
int btrapz_last_solve_form(const btrapz_ctx *ctx);

/* Arg-min of cost over contiguous groups of `group` candidates (B % group == 0).
 * best_idx[g] = global candidate index (index_base + local), ties -> lowest index;
 * -1 when no candidate of the group was solved.  Device pointers. */
int btrapz_argmin_device(btrapz_ctx *ctx, int B, int group, long long index_base,
                         const double *cost, long long *best_idx, double *best_cost,
                         void *stream);

/* Multi-GPU arg-min, last step: the winners of all ranks after the all-gather (one process per GPU; RCCL has no
 * MINLOC).  pairs [world][n][2] int64 = (bit pattern of the float64 cost, global index or -1) of every rank's local
 * winner of arg-min group g; best_cost / best_idx [n]: lowest cost, ties -> lowest index, -1 when nobody solved one.
 * Device pointers.  (spectral_amd/dist.py packs, gathers and calls this.) */
int btrapz_argmin_pairs_device(btrapz_ctx *ctx, int world, int n, const long long *pairs, double *best_cost,
                               long long *best_idx, void *stream);

/* ---- the multi-GPU step (one host process, G devices) ------------------------------------------------------------
 * north_star: "batches of candidate corridors shard embarrassingly across the 8 GPUs of one node with RCCL over xGMI
 * only for the final arg-min reduction", the host staying C++ over this C-ABI.  The reference has no counterpart: one
 * corridor per call inside a single-process replanning loop (src/cart_frenet.py:1516-1571, src/solve_3d.cc:1231-1414).
 *
 * A btrapz_multi owns, per entry of `devices`, a context, a stream and the buffers of that device's shard.  A step
 * (btrapz_multi_solve_argmin) is, per device and all asynchronous: solve the shard (btrapz_solve_batch_device), arg-min
 * over it (btrapz_argmin_device), pack the local winner's record -- (cost bits, global index, 12 S control points):
 * 16 + 96 S bytes -- then ONE all-gather of the G records and the same deterministic lexicographic min (cost, then the
 * lowest global index: the order of btrapz_argmin_pairs_device) on every device: each ends with the global winner's
 * index, cost and control points.  No other data crosses devices.
 *   transport   BTRAPZ_MULTI_RCCL: ncclAllGather inside ncclGroupStart / ncclGroupEnd, communicators from
 *               ncclCommInitAll; librccl.so is resolved at run time (dlopen, next to the HIP runtime the process runs
 *               on; BTRAPZ_RCCL_LIB overrides) -- the library has no link dependency on it;
 *               BTRAPZ_MULTI_COPIES: stream-ordered peer copies (hipMemcpyPeerAsync), the fallback without RCCL;
 *               BTRAPZ_MULTI_AUTO: RCCL when it can be had, else copies (btrapz_multi_last_error says why).
 *   logical devices: an ordinal may repeat in `devices` -- every entry still gets its own context, stream, buffers and
 *               shard, so the whole path runs on a box with one GPU (transport: copies).
 * Shards are contiguous, ceil(B / G) candidates per device (spectral_amd/dist.py::shard_bounds; trailing devices may be
 * short or empty).  With group > 0 (B % group == 0, BASELINE config 5: `group` candidates per ego agent) a shard is a
 * whole number of arg-min groups, every group's winner is found on its own device and there is NO collective.
 * Results do not depend on G as long as the form of the solve kernel does not (automatic selection goes by the shard's
 * size: pin btrapz_options.lean / split / cap_iter to compare across G bit for bit).
 * Not thread-safe; one step at a time per handle. */
typedef struct btrapz_multi btrapz_multi;
enum { BTRAPZ_MULTI_AUTO = 0, BTRAPZ_MULTI_COPIES = 1, BTRAPZ_MULTI_RCCL = 2 };
int btrapz_multi_create(btrapz_multi **m, const int *devices, int G, int transport);
int btrapz_multi_destroy(btrapz_multi *m);
const char *btrapz_multi_last_error(const btrapz_multi *m);
int btrapz_multi_transport(const btrapz_multi *m);               /* BTRAPZ_MULTI_COPIES or BTRAPZ_MULTI_RCCL */
const char *btrapz_multi_transport_library(const btrapz_multi *m); /* the librccl.so in use ("" for copies) */
int btrapz_multi_device_count(const btrapz_multi *m);
/* [lo, hi) of device slot g (0 <= g < G) for B candidates; group as above (0: one arg-min group over the whole batch). */
int btrapz_multi_shard_bounds(int B, int G, int g, int group, int *lo, int *hi);

/* The batch, from HOST arrays of the whole batch (layout of btrapz_solve_batch_device): every device receives its shard
 * and keeps it resident until the next upload.  Synchronous (the host arrays may be released on return). */
int btrapz_multi_upload(btrapz_multi *m, int B, int S, int group, const double *seg, const double *init,
                        const double *ref_end, const double *dl_bounds);
/* ... or shards that are on the devices already (e.g. written there by btrapz_corridor_batch_device): shards [G], in
 * device-slot order, contiguous in the global index, pointers valid on that slot's device (layout of
 * btrapz_solve_batch_device with B = the shard's size).  The arrays stay the caller's. */
typedef struct btrapz_multi_shard {
  int B;                 /* candidates of this shard (0: none) */
  long long index_base;  /* global index of its first candidate */
  const double *seg, *init, *ref_end, *dl_bounds;
} btrapz_multi_shard;
int btrapz_multi_set_shards(btrapz_multi *m, int B, int S, int group, const btrapz_multi_shard *shards);

/* One step over the current batch: returns when everything is enqueued. */
int btrapz_multi_solve_argmin(btrapz_multi *m, const btrapz_shared *shared, const btrapz_options *opt);
/* Waits for the step and copies the winner(s) to the host.  One arg-min group (group = 0): best_idx [1], best_cost [1],
 * best_ctrl [12 S] (NaN when no candidate was solved: best_idx -1, best_cost +inf), read from device slot `device_slot`
 * after waiting for that device only, or from slot 0 after waiting for all (-1).  group > 0: the winners of all B / group
 * groups in group order.  Any output pointer may be NULL. */
int btrapz_multi_result(btrapz_multi *m, int device_slot, long long *best_idx, double *best_cost, double *best_ctrl);
int btrapz_multi_wait(btrapz_multi *m);
/* What lives on device slot g after a step (device pointers; valid until the next upload / set_shards / step). */
typedef struct btrapz_multi_view {
  int device, B;
  long long index_base;
  void *stream;               /* the slot's hipStream_t: order further work on this device behind the step */
  btrapz_ctx *ctx;            /* the slot's context (sampling, state evaluation ... on its shard) */
  double *ctrl, *cost;        /* [B][12 S], [B] of the shard */
  int *status, *iters;
  long long *best_idx;        /* the global winner (group = 0: [1], the same on every device) or the shard's winners */
  double *best_cost, *best_ctrl;
} btrapz_multi_view;
int btrapz_multi_shard_view(btrapz_multi *m, int device_slot, btrapz_multi_view *view);
/* The whole batch's results to HOST arrays ctrl [B][12 S], cost [B], status [B], iters [B] (any may be NULL): tests. */
int btrapz_multi_download(btrapz_multi *m, double *ctrl, double *cost, int *status, int *iters);

/* Bernstein sampling (solve_3d.cc:1279-1392) of nsel selected candidates on device.
 *   sel [nsel] candidate indices; t taken from seg; out [nsel][6][max_points]
 *   (s, ds, dds, l, dl, ddl); npoints [nsel].  Device pointers. */
int btrapz_sample_device(btrapz_ctx *ctx, int B, int S, double delta, const double *seg,
                         const double *init, const double *ctrl, int nsel,
                         const long long *sel, int max_points, double *out, int *npoints,
                         void *stream);

/* ---- ragged batches and the device corridor stage -------------------------
 * Candidates may have different segment counts: seg[f][b][k] and ctrl[b][12*seg_stride] have
 * seg_stride slots per candidate, seg_count[b] in 1..min(BTRAPZ_MAX_SEGMENTS_LONG, seg_stride) of them are used
 * (control points of candidate b: s axis at ctrl[b][0 .. 6 S_b), l axis at ctrl[b][6 S_b .. 12 S_b)).  Candidates
 * of at most 64 segments are bucketed by segment count on the device and every bucket is solved by the same kernel
 * in one launch; candidates of 65..256 segments (cold solves without a rescue pass; since round 6) are solved by the
 * long form, one launch per count -- their counts come to the host for that, one stream synchronisation;
 * candidates with an unusable count get status BTRAPZ_NO_CORRIDOR and cost +inf. */
int btrapz_solve_ragged_device(btrapz_ctx *ctx, const btrapz_shared *shared,
                               const btrapz_options *opt, int B, int seg_stride,
                               const double *seg, const int *seg_count, const double *init,
                               const double *ref_end, const double *dl_bounds, double *ctrl,
                               double *cost, int *status, int *iters, void *stream);

/* CorridorGeneration + CorridorSplit + CollisionCheck for B candidates on the device, straight into
 * the batch record above (replaces the host loop trp_wrapper.cpp:176-188 for candidate sets).
 *   s_bounds, l_bounds [B][num_obs][N][2]  per-knot (lower, upper), the order of the input file
 *   ds_bounds, dl_bounds_knots [B][N][2]   s_ref, l_ref [B][N]
 * outputs: seg [NUM_SEG_FIELDS][B][seg_stride], seg_count [B] (0: nothing selected, -1: overflow or a
 * segment with t <= 0), ref_end [B][2], dl_bounds [B][10].  Up to 512 knots and 64 obstacles: the wave-wide kernels
 * (one wavefront per candidate, at most 64 selected segments); beyond (N <= 100 000, num_obs <= 1 000, seg_stride <=
 * BTRAPZ_MAX_SEGMENTS_LONG: the bounds of find_traj's parser and of the solve): one LANE per candidate on lists in a
 * workspace of the context -- the same statements, the same record, two orders of magnitude slower per candidate. */
int btrapz_corridor_batch_device(btrapz_ctx *ctx, int variant, int B, int N, int num_obs,
                                 double delta, const double *s_bounds, const double *l_bounds,
                                 const double *ds_bounds, const double *dl_bounds_knots,
                                 const double *s_ref, const double *l_ref, int seg_stride,
                                 double *seg, int *seg_count, double *ref_end, double *dl_bounds,
                                 void *stream);

/* ---- obstacle prisms -> per-knot bounds -------------------------------------------------------
 * The scene logic in front of the corridor text file, for B scenes at once: Car.getCar + get_bounds of the reference's
 * harness (src/cart_frenet.py:664-1030, lineFromPoints :818-830).  A car is a prism in (s, l, t): centre (s0, l0, t0),
 * constant velocities, duration T, grown by l_safe / w_safe (:694-700); the cars' lateral extents cut the road into
 * strips, every strip becomes one "obstacle corridor" of btrapz_corridor_batch_device: per knot, the rear face of a car
 * that starts at t0 = 0 bounds s from above, the front face of any other car from below, inside the car's window.
 * Where lateral extents overlap the reference's result depends on the order the cars were constructed in; here the
 * geometry has one definition (spectral_amd/csrc/prism_kernels.hip, oracle/prism_oracle.py), equal to the reference's
 * own output on every scene where that output is a proper partition of the road (tests/test_prism_bounds.py).
 *   prisms   [B][P][8]  s0, l0, t0, vel_s, vel_l, T, active (0 = slot unused), reserved;  P <= 16
 *   outputs  s_bounds, l_bounds [B][O][N][2] (lower, upper) -- the layout btrapz_corridor_batch_device reads with
 *            num_obs = O; strips beyond the scene's count are corridors no reference can enter
 *            n_strips [B]: strips of the scene, -1 when it has more than O (at most 2 P + 1)                      */
typedef struct btrapz_road {
  double s_lo, s_hi;        /* s_l_l, s_u_l  (cart_frenet.py:54-55) */
  double l_lo, l_hi;        /* d_l_l, d_u_l  (:57-58) */
  double l_safe, w_safe;    /* 5/3 + 5/3, 2/3 + 2/3 (:698-699) */
  double knots_per_second;  /* 10 (the reference writes `i/10`, `t0*10`) */
} btrapz_road;
int btrapz_prism_bounds_device(btrapz_ctx *ctx, int B, int P, int N, const btrapz_road *road, const double *prisms,
                               int O, double *s_bounds, double *l_bounds, int *n_strips, void *stream);

/* Obstacle prisms -> corridors in ONE launch: btrapz_prism_bounds_device followed by btrapz_corridor_batch_device with
 * num_obs = O, with the strips evaluated inside the corridor kernel where it reads them instead of written to memory
 * and read back (2 x 32 O N bytes per scene less traffic, one launch less).  Same outputs, bit for bit
 * (tests/test_gpu_prism_bounds.py).  Arguments as in the two calls it replaces; n_strips may be NULL.  Scenes with
 * O * N > 1536 take the two launches internally, through a workspace of the context. */
int btrapz_prism_corridor_batch_device(btrapz_ctx *ctx, int variant, int B, int P, int N, const btrapz_road *road,
                                       const double *prisms, int O, double delta, const double *ds_bounds,
                                       const double *dl_bounds_knots, const double *s_ref, const double *l_ref,
                                       int seg_stride, double *seg, int *seg_count, double *ref_end,
                                       double *dl_bounds, int *n_strips, void *stream);

/* btrapz_sample_device for ragged batches (seg_count may be NULL: every candidate has seg_stride). */
int btrapz_sample_ragged_device(btrapz_ctx *ctx, int B, int seg_stride, const int *seg_count,
                                double delta, const double *seg, const double *init,
                                const double *ctrl, int nsel, const long long *sel, int max_points,
                                double *out, int *npoints, void *stream);

/* ---- receding-horizon warm start ----------------------------------------------
 * The reference re-solves cold at every replanning step: a fresh OSQP workspace per call
 * (solve_3d.cc:1246, osqp_cleanup :1256,1410) inside the replan loop cart_frenet.py:1516-1571.  Here a
 * solve may start from the joint states and multipliers of an earlier solve of a nearby problem.  The
 * optimum does not depend on the start (the QP is strictly convex); only the iteration count does.
 *   x0      [B][2][seg_stride][3]  (p, v, a) at the END of every segment, s axis then l axis; entries that
 *           are not finite fall back to the cold start of that segment.  NULL: cold primal start.
 *   lam0    [2][36][B][seg_stride] multipliers (rows 0-17: lower bounds of the 6 position, 5 velocity,
 *           4 acceleration, 3 jerk rows; 18-35: upper bounds), as written to lam_out by an earlier solve;
 *           negative / non-finite entries count as 0.  NULL: none.
 *   lam_out same layout, multipliers at the end of this solve; may be the same array as lam0 (every entry is
 *           read before the iterations and written after them by the same lane).  NULL: not stored.
 *   smin, mu0: slacks start at max(gap, smin), multipliers at lam0 + mu0 / slack (0 -> 1e-2 and 1e-4).
 * A warm-started candidate that stalls, or is still far from converged after 12 iterations, is restarted
 * once from the cold start inside the kernel: a bad guess costs iterations, never the result.
 * With warm == NULL, or x0 == lam0 == NULL, the start is the cold one of btrapz_solve_*_device. */
typedef struct btrapz_warm {
  const double *x0;
  const double *lam0;
  double *lam_out;
  double mu0;
  double smin;
  const int *hint;  /* [B] or NULL (uniform batches): expected difficulty of every candidate, e.g. the iters[] of the
                       previous replanning step.  Candidates of one class (the value clamped to 1..64) share
                       wavefronts, so a few hard or infeasible candidates no longer hold up the wavefronts of the
                       easy ones.  Scheduling only: results do not depend on it. */
} btrapz_warm;

/* btrapz_solve_ragged_device with a warm start (seg_count == NULL: uniform batch of seg_stride segments). */
int btrapz_solve_warm_device(btrapz_ctx *ctx, const btrapz_shared *shared, const btrapz_options *opt,
                             const btrapz_warm *warm, int B, int seg_stride, const double *seg,
                             const int *seg_count, const double *init, const double *ref_end,
                             const double *dl_bounds, double *ctrl, double *cost, int *status,
                             int *iters, void *stream);

/* State (p, v, a) of solved candidates at arbitrary times: x[b][axis][j] at times[b][j] seconds from the
 * start of candidate b's horizon (Bezier evaluation of solve_3d.cc:1366-1388; beyond the last segment the
 * end state is extrapolated at constant velocity).  With times = shift + the cumulative durations of the
 * NEXT step's segments this is x0 of btrapz_warm.  times [B][n_times], x [B][2][n_times][3]; seg_count may
 * be NULL. */
int btrapz_eval_states_device(btrapz_ctx *ctx, int B, int seg_stride, const int *seg_count,
                              const double *seg, const double *ctrl, int n_times, const double *times,
                              double *x, void *stream);

/* Test hook: the batch-invariant M' pQp_d M table (solve_3d.cc:87-143) from the library's host builder (find_traj's
 * single launch) and from its device builder (the batched entry points): [2][4][21] doubles each, host pointers. */
int btrapz_debug_mqm_tables(btrapz_ctx *ctx, const btrapz_shared *shared, double *host_table, double *device_table);
/* Test / analysis hook: iterations [B][2] and status [B][2] (s axis, l axis) of the axis problems of the context's last
 * batched solve of B candidates, into host arrays; synchronises the device. */
/* Test hooks (host only): the scanner of corridor files and the "%.3f" writer of trajectory files -- fast paths around
 * strtod / printf that must give the same value / the same text (tests/test_text_io.py).  out336: 336 bytes. */
double btrapz_debug_parse_double(const char *text, int *consumed);
int btrapz_debug_format_fixed(double v, char *out336);
int btrapz_debug_axis_records(btrapz_ctx *ctx, int B, int *iters, int *status);
/* ... and the keys [2][B] (axis-major) the resume launch of the context's last capped solve of B candidates was bucketed
 * by: 0 where the axis problem was not handed over. */
int btrapz_debug_resume_keys(btrapz_ctx *ctx, int B, int *keys);

/* Host-pointer convenience wrapper: H2D, solve, D2H, synchronous. */
int btrapz_solve_batch_host(btrapz_ctx *ctx, const btrapz_shared *shared,
                            const btrapz_options *opt, int B, int S, const double *seg,
                            const double *init, const double *ref_end,
                            const double *dl_bounds, double *ctrl, double *cost,
                            int *status, int *iters);

#ifdef __cplusplus
}
#endif
#endif
