/*
 * btrapz_oracle.h -- CPU restatement of the reference's Bezier-in-corridor
 * trajectory QP (Srujan-D/spectral).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check
 * in __graft_entry__.py and bench.py's cpu_baseline leg may load it.  The
 * product path (spectral_amd/) never links, imports or calls anything here.
 *
 * HOW PARITY IS PINNED.  The reference has no tests; its sources need Eigen + OSQP (absent
 * from this image, so oracle/_ref cannot be built) and its prebuilt libraries need
 * libosqp.so (absent).  The arithmetic of the solve lives in OSQP (oxfordcontrol/osqp,
 * version unpinned by the reference: only the comment "osqp-0.4.1, 0.5.0" at
 * src/solve_3d.cc:1246), whose published ADMM algorithm is restated in orc_osqp_solve().
 * Pins (tests/test_reference_goldens.py, tests/test_oracle_*.py):
 *  (1) REFERENCE-GENERATED VECTORS: trajectory files the reference itself wrote
 *      (tests/golden/ref_outputs).  orc_find_traj() reproduces s4_slt_3d.txt, s4_cub_3d.txt,
 *      s5_slt_3d.txt (inputs c4.txt / c5.txt, s weights of weights.txt) to print precision
 *      in every column -- including OSQP's unconverged iterate after 5000 iterations -- and
 *      the s columns of s2_slt_3d_{4,5}.txt (input c2.txt) with both the ADMM port and x*.
 *      This pins parser, corridor pipeline, assembly, the ADMM port, sampling and the file
 *      format on those paths.
 *  (2) assembly against closed forms independent of the reference's route (Gauss
 *      quadrature of squared Bernstein derivatives, Bezier derivative control points);
 *  (3) the optimum x* through a KKT certificate, tight ADMM and a third-party QP solver
 *      (HiGHS, shipped inside scipy).
 * Unpinned remainder: where the weights of a saved run are unknown only its row count and
 * first row are checked, and OSQP's wall-clock-dependent adaptive-rho interval is fixed (25).
 *
 * All citations are file:line relative to /root/reference.
 */
#ifndef BTRAPZ_ORACLE_H
#define BTRAPZ_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef long long orc_int; /* OSQP c_int built with DLONG (libtrp.so mangled names) */

/* include/btrapz/cube_type.h:2-24 */
typedef struct {
  int beg_t, end_t;
  double t;
  double beg_l, end_l;
  double upp_skew, upp_bias, down_skew, down_bias;
  double l_upp_skew, l_upp_bias, l_down_skew, l_down_bias;
  int count;
} orc_cube;

/* include/btrapz/py_cpp_.h:6-21 (88 bytes) */
typedef struct {
  double s_acc_weight, s_jerk_weight, l_acc_weight, l_jerk_weight;
  double weight_s_ref, weight_ds_ref, weight_l_ref, weight_dl_ref;
  double weight_end_s, weight_end_l;
  int iteration;
} orc_params;

enum { ORC_TRAPEZOID = 0, ORC_CUBOID = 1 };

/* Parsed input file, grammar of src/trp_wrapper.cpp:39-144. */
typedef struct {
  int N;
  double delta;
  double init_s[3], init_l[3];
  int num_obs;
  double ds_ref, dl_ref;
  double dds[2], ddds[2], ddl[2], dddl[2];
  double *x_bounds; /* [num_obs][N][2] */
  double *y_bounds; /* [num_obs][N][2] */
  double *dx_bounds; /* [N][2] */
  double *dy_bounds; /* [N][2] */
  double *x_ref, *y_ref, *x_kappa, *y_kappa; /* [N] */
} orc_input;

int orc_input_read(const char *path, orc_input *in); /* 0 ok, <0 error */
void orc_input_free(orc_input *in);

/* Corridor pipeline -------------------------------------------------------*/
/* CorridorGeneration + CorridorSplit for one obstacle's per-knot bounds.
 * solve_3d.cc:323-486,729-772 ; cuboid_3d.cc:301-407,588-625.
 * Returns number of cubes written (<= cap) or <0 on error. */
int orc_corridor_generation(int variant, int N, double delta,
                            const double *x_bounds, const double *y_bounds,
                            orc_cube *out, int cap);

/* CollisionCheck: solve_3d.cc:488-714 ; cuboid_3d.cc:409-573.
 * cubes: concatenated per-obstacle lists, counts[num_obs].  Returns S (number
 * of cubes in new_corridor) or <0 (e.g. empty set, where the reference
 * underflows an unsigned loop bound: defined here as failure -2). */
int orc_collision_check(int variant, int N, double delta, const orc_cube *cubes,
                        const int *counts, int num_obs, const double *x_ref,
                        const double *y_ref, orc_cube *out, int cap);

/* Assembly ----------------------------------------------------------------*/
typedef struct {
  /* weights (Params) */
  double w_s[4]; /* weight_x_ref, weight_dx_ref, weight_ddx, weight_dddx */
  double w_l[4];
  double weight_end_s, weight_end_l;
  double ds_ref, dl_ref;
  double dds[2], ddds[2], ddl[2], dddl[2];
  double init_s[3], init_l[3];
  int N;
  double delta;
  const double *dx_bounds; /* [N][2] per-knot */
  const double *dy_bounds; /* [N][2] */
  const double *x_ref, *y_ref; /* [N] */
} orc_qp_params;

/* Sizes: n = 12S, m = 42S, nnzP = 42S, nnzA = 104S-12. */
typedef struct {
  int n, m;
  orc_int *P_p, *P_i; double *P_x; int P_nnz;
  orc_int *A_p, *A_i; double *A_x; int A_nnz;
  double *q, *l, *u;
} orc_qp;

int orc_assemble(int variant, int S, const orc_cube *corridor,
                 const orc_qp_params *pp, orc_qp *qp);
void orc_qp_free(orc_qp *qp);

/* OSQP-style ADMM -----------------------------------------------------------*/
typedef struct {
  double rho, sigma, alpha;
  double eps_abs, eps_rel, eps_prim_inf, eps_dual_inf;
  int max_iter, scaling, scaled_termination, check_termination;
  int adaptive_rho, adaptive_rho_interval;
  double adaptive_rho_tolerance;
  int polish; /* 0: none (reference); 1: oracle-side active-set polish */
} orc_settings;

/* Effective settings of the reference: OSQP defaults + solve_3d.cc:1446-1462
 * + :1236-1243 + max_iter=5000 (trp_wrapper.cpp:191). */
void orc_settings_reference(orc_settings *s);
/* High-accuracy settings used to compute x* for parity. */
void orc_settings_tight(orc_settings *s);

typedef struct {
  int status; /* OSQP status_val: 1 solved, 2 solved inaccurate, -2 max iter,
                 -3 primal infeasible, 3 p.i. inaccurate, -4 dual infeasible,
                 4 d.i. inaccurate, -10 unsolved */
  int iter;
  int rho_updates;
  double obj_val, pri_res, dua_res, rho;
} orc_info;

int orc_osqp_solve(const orc_qp *qp, const orc_settings *s, double *x,
                   double *y, orc_info *info);

/* High-accuracy optimum x* by a dense Mehrotra interior-point method on the
 * general (P,q,A,l,u); independent of orc_osqp_solve and of the product's
 * structured solver.  info->pri_res returns the final KKT score
 * max(|r_dual|/(1+|q|), |r_prim|/(1+|bounds|), mu). */
int orc_ipm_solve(const orc_qp *qp, double eps, int max_iter, double *x,
                  double *y, orc_info *info);

/* The same method on the relaxed problem of the product's rescue pass
 * (include/btrapz_hip.h, btrapz_options.elastic; no reference counterpart -- the
 * reference's behaviour it stands in for is the acceptance of OSQP's status 2,
 * solve_3d.cc:1251-1253): every inequality row elastic, l <= a'x - d <= u, with
 * sum d^2 / (2 delta) added to the objective; equality rows stay exact. */
int orc_elastic_solve(const orc_qp *qp, double delta, double eps, int max_iter,
                      double *x, double *y, orc_info *info);

/* Unscaled KKT residuals of (x,y): stationarity inf-norm, primal violation,
 * complementarity; res[0..2]. */
void orc_kkt_residuals(const orc_qp *qp, const double *x, const double *y,
                       double *res);

/* Post-solve ---------------------------------------------------------------*/
/* Bernstein sampling, solve_3d.cc:1279-1392.  out: 6 arrays of length
 * *npoints (s,ds,dds,l,dl,ddl), cap each.  Returns 0 ok, <0 if the
 * reference's CHECK_EQ(var_index,num_of_points_) (:1407) would abort. */
int orc_sample(int S, const orc_cube *corridor, double delta,
               const double *x /*12S*/, const double init_s[3],
               const double init_l[3], double *s, double *ds, double *dds,
               double *l, double *dl, double *ddl, int cap, int *npoints);

/* a_cost: trp_wrapper.cpp:207-286 ; cub_wrapper.cpp:201-262. */
double orc_acost(int variant, const orc_params *p, const orc_input *in,
                 int npoints, const double *s, const double *ds,
                 const double *dds, const double *l, const double *dl,
                 const double *ddl);

/* Whole find_traj (trp_wrapper.cpp:20-305 / cub_wrapper.cpp:20-283) with the
 * two hard-coded paths made explicit.  Returns a_cost or 1e11. Extra outputs
 * (may be NULL): S, control points (cap 12*64), info. */
double orc_find_traj(int variant, const char *input_path,
                     const char *output_path, const orc_params *p,
                     const orc_settings *settings, int *S_out, double *ctrl_out,
                     orc_cube *corridor_out, orc_info *info_out);

/* Candidates [b0,b1) of a batch record (layout of include/btrapz_hip.h) solved one by
 * one: orc_assemble + orc_osqp_solve (exact=0, the reference's algorithm) or
 * orc_ipm_solve (exact=1, x*).  shared[21]: see btrapz_oracle.c. */
int orc_batch_solve(int variant, int B, int S, const double *seg, const double *init,
                    const double *ref_end, const double *dl_bounds, const double *shared,
                    const orc_settings *settings, int exact, int b0, int b1, double *ctrl,
                    double *obj, int *status, int *iters);

/* The same over `threads` POSIX threads (candidates drawn from a shared counter: their cost is not uniform). */
int orc_batch_solve_mt(int variant, int B, int S, const double *seg, const double *init, const double *ref_end,
                       const double *dl_bounds, const double *shared, const orc_settings *settings, int exact,
                       int b0, int b1, int threads, double *ctrl, double *obj, int *status, int *iters);

#ifdef __cplusplus
}
#endif
#endif
