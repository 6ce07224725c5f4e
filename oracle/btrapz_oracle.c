/*
 * btrapz_oracle.c -- CPU restatement of the reference's trajectory QP path.
 * TEST INFRASTRUCTURE ONLY (see btrapz_oracle.h header: "HOW PARITY IS PINNED").
 *
 * Follows, function by function (paths relative to /root/reference):
 *   corridor pipeline  src/solve_3d.cc:323-486,488-714,729-772
 *                      src/cuboid_3d.cc:301-407,409-573,588-625
 *   assembly           src/solve_3d.cc:70-321 (P,q), :779-1129 (A,l,u),
 *                      src/cuboid_3d.cc:632-988, src/solve_3d.cc:1143-1229
 *   solver             OSQP (external, version unpinned; algorithm restated
 *                      from the OSQP paper / 0.5.0 sources' published
 *                      behaviour), settings src/solve_3d.cc:1236-1243,1446-1462
 *   sampling / cost    src/solve_3d.cc:1279-1392, src/trp_wrapper.cpp:207-286,
 *                      src/cub_wrapper.cpp:201-262
 *
 * Undefined behaviour in the reference is given DEFINED semantics here:
 *   - x_ref_[10k+1] read past the end (solve_3d.cc:1161)      -> index clamped to N-1
 *   - temp.size()-1 underflow on an empty set (:617,639,678)  -> failure (-2)
 *   - x_ref[i], i >= N in a_cost (trp_wrapper.cpp:221,257)     -> index clamped to N-1
 *   - l[num_of_knots-1] past the sampled length (:269)         -> index clamped
 *   - uninitialised l_cost (cub_wrapper.cpp:237)               -> 0.0
 *   - CHECK_* abort()                                          -> failure return
 */
#include "btrapz_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define N_POLY 6
#define TRAJ_ORDER 5

/* ------------------------------------------------------------------------ */
/* Parser: src/trp_wrapper.cpp:39-144.  A failed `ifs >> v` leaves v
 * unchanged and every later read fails too (stream failbit).               */
/* ------------------------------------------------------------------------ */
typedef struct { FILE *f; int failed; } tok_stream;

static void rd_double(tok_stream *ts, double *v) {
  if (ts->failed) return;
  double t;
  if (fscanf(ts->f, "%lf", &t) == 1) *v = t; else ts->failed = 1;
}
static void rd_int(tok_stream *ts, int *v) {
  if (ts->failed) return;
  int t;
  if (fscanf(ts->f, "%d", &t) == 1) *v = t; else ts->failed = 1;
}

int orc_input_read(const char *path, orc_input *in) {
  memset(in, 0, sizeof(*in));
  tok_stream ts; ts.f = fopen(path, "r"); ts.failed = 0;
  if (!ts.f) return -1;
  rd_int(&ts, &in->N); rd_double(&ts, &in->delta);
  for (int i = 0; i < 3; i++) rd_double(&ts, &in->init_s[i]);
  for (int i = 0; i < 3; i++) rd_double(&ts, &in->init_l[i]);
  rd_int(&ts, &in->num_obs);
  rd_double(&ts, &in->ds_ref); rd_double(&ts, &in->dl_ref);
  rd_double(&ts, &in->dds[0]); rd_double(&ts, &in->dds[1]);
  rd_double(&ts, &in->ddds[0]); rd_double(&ts, &in->ddds[1]);
  rd_double(&ts, &in->ddl[0]); rd_double(&ts, &in->ddl[1]);
  rd_double(&ts, &in->dddl[0]); rd_double(&ts, &in->dddl[1]);
  if (ts.failed || in->N < 2 || in->N > 100000 || in->num_obs < 0 || in->num_obs > 1000) {
    fclose(ts.f); return -2;
  }
  int N = in->N, O = in->num_obs;
  in->x_bounds = (double *)calloc((size_t)O * N * 2 + 2, sizeof(double));
  in->y_bounds = (double *)calloc((size_t)O * N * 2 + 2, sizeof(double));
  in->dx_bounds = (double *)calloc((size_t)N * 2, sizeof(double));
  in->dy_bounds = (double *)calloc((size_t)N * 2, sizeof(double));
  in->x_ref = (double *)calloc(N, sizeof(double));
  in->y_ref = (double *)calloc(N, sizeof(double));
  in->x_kappa = (double *)calloc(N, sizeof(double));
  in->y_kappa = (double *)calloc(N, sizeof(double));
  double val[2] = {0, 0};
  for (int o = 0; o < O; o++) {
    for (int i = 0; i < N; i++) {
      rd_double(&ts, &val[0]); rd_double(&ts, &val[1]);
      in->x_bounds[((size_t)o * N + i) * 2] = val[0];
      in->x_bounds[((size_t)o * N + i) * 2 + 1] = val[1];
    }
    for (int i = 0; i < N; i++) {
      rd_double(&ts, &val[0]); rd_double(&ts, &val[1]);
      in->y_bounds[((size_t)o * N + i) * 2] = val[0];
      in->y_bounds[((size_t)o * N + i) * 2 + 1] = val[1];
    }
  }
  for (int i = 0; i < N; i++) {
    rd_double(&ts, &val[0]); rd_double(&ts, &val[1]);
    in->dx_bounds[2 * i] = val[0]; in->dx_bounds[2 * i + 1] = val[1];
  }
  for (int i = 0; i < N; i++) {
    rd_double(&ts, &val[0]); rd_double(&ts, &val[1]);
    in->dy_bounds[2 * i] = val[0]; in->dy_bounds[2 * i + 1] = val[1];
  }
  for (int i = 0; i < N; i++) { rd_double(&ts, &val[0]); in->x_ref[i] = val[0]; }
  for (int i = 0; i < N; i++) { rd_double(&ts, &val[0]); in->y_ref[i] = val[0]; }
  for (int i = 0; i < N; i++) { rd_double(&ts, &val[0]); in->x_kappa[i] = val[0]; }
  for (int i = 0; i < N; i++) { rd_double(&ts, &val[0]); in->y_kappa[i] = val[0]; }
  fclose(ts.f);
  return 0;
}

void orc_input_free(orc_input *in) {
  free(in->x_bounds); free(in->y_bounds); free(in->dx_bounds); free(in->dy_bounds);
  free(in->x_ref); free(in->y_ref); free(in->x_kappa); free(in->y_kappa);
  memset(in, 0, sizeof(*in));
}

/* ------------------------------------------------------------------------ */
/* Corridor pipeline                                                         */
/* ------------------------------------------------------------------------ */
static void cube_default(orc_cube *c) { /* cube_type.h:12-21 */
  memset(c, 0, sizeof(*c));
  c->upp_bias = 1000.0;
  c->l_upp_bias = 1000.0;
}

/* CorridorSplit: solve_3d.cc:729-772 ; cuboid_3d.cc:588-625 (no l_* copy). */
static int corridor_split(int variant, orc_cube *c, int num, int cap) {
  int temp_num = num;
  for (int k = 0; k < temp_num; k++) {
    while (c[k].t > 1) {
      if (temp_num + 1 > cap) return -1;
      c[k].t = c[k].t - 1;
      orc_cube m; cube_default(&m);
      m.beg_t = c[k].beg_t;
      c[k].beg_t = c[k].beg_t + 10;
      m.end_t = m.beg_t + 10;
      m.t = 1.0;
      m.down_skew = c[k].down_skew;
      m.down_bias = c[k].down_bias;
      if (variant == ORC_TRAPEZOID) {
        m.l_down_skew = c[k].l_down_skew;
        m.l_down_bias = c[k].l_down_bias;
      }
      c[k].down_bias = m.down_bias + 1.0 * m.down_skew;
      m.upp_skew = c[k].upp_skew;
      m.upp_bias = c[k].upp_bias;
      if (variant == ORC_TRAPEZOID) {
        m.l_upp_skew = c[k].l_upp_skew;
        m.l_upp_bias = c[k].l_upp_bias;
      }
      m.beg_l = c[k].beg_l;
      m.end_l = c[k].end_l;
      c[k].upp_bias = m.upp_bias + 1.0 * m.upp_skew;
      /* corridor.insert(corridor.begin() + k, mcube) */
      memmove(&c[k + 1], &c[k], (size_t)(temp_num - k) * sizeof(orc_cube));
      c[k] = m;
      temp_num++;
      k++;
    }
  }
  return temp_num;
}

int orc_corridor_generation(int variant, int N, double delta, const double *xb,
                            const double *yb, orc_cube *out, int cap) {
#define XLO(i) xb[2 * (i)]
#define XHI(i) xb[2 * (i) + 1]
#define YLO(i) yb[2 * (i)]
#define YHI(i) yb[2 * (i) + 1]
  if (cap < 1 || N < 3) return -1;
  int j = 0;
  {
    orc_cube m; cube_default(&m);
    m.beg_t = 0;
    m.down_skew = (XLO(1) - XLO(0)) / delta;
    m.down_bias = XLO(0);
    m.upp_skew = (XHI(1) - XHI(0)) / delta;
    m.upp_bias = XHI(0);
    if (variant == ORC_TRAPEZOID) { /* solve_3d.cc:338-341 */
      m.l_down_skew = (YLO(1) - YLO(0)) / delta;
      m.l_down_bias = YLO(0);
      m.l_upp_skew = (YHI(1) - YHI(0)) / delta;
      m.l_upp_bias = YHI(0);
    }
    m.beg_l = YLO(0);
    m.end_l = YHI(0);
    out[j++] = m;
  }
  for (int i = 2; i < N - 1; i++) {
    orc_cube m; cube_default(&m);
    double dskew = (XLO(i) - XLO(i - 1)) / delta;
    double uskew = (XHI(i) - XHI(i - 1)) / delta;
    if (variant == ORC_TRAPEZOID) { /* solve_3d.cc:358-367 */
      double l_dskew = (YLO(i) - YLO(i - 1)) / delta;
      double l_uskew = (YHI(i) - YHI(i - 1)) / delta;
      m.l_down_bias = YLO(i);
      m.l_upp_bias = YHI(i);
      m.l_down_skew = l_dskew;
      m.l_upp_skew = l_uskew;
    }
    const double mthre = 0.2;
    if ((fabs(dskew - out[j - 1].down_skew) > mthre) ||
        (fabs(uskew - out[j - 1].upp_skew) > mthre)) {
      if (j + 1 > cap) return -1;
      out[j - 1].end_t = i;
      m.beg_t = i;
      m.down_skew = (XLO(i + 1) - XLO(i)) / delta;
      m.down_bias = XLO(i);
      m.upp_skew = (XHI(i + 1) - XHI(i)) / delta;
      m.upp_bias = XHI(i);
      m.beg_l = YLO(i);
      m.end_l = YHI(i);
      out[j++] = m;
    }
  }
  out[j - 1].end_t = N - 1;
  for (int i = 0; i < j; i++) out[i].t = (out[i].end_t - out[i].beg_t) * delta;
  return corridor_split(variant, out, j, cap);
#undef XLO
#undef XHI
#undef YLO
#undef YHI
}

static int cube_same(const orc_cube *a, const orc_cube *b) { /* solve_3d.cc:621 */
  return a->beg_t == b->beg_t && a->end_t == b->end_t && a->down_bias == b->down_bias &&
         a->down_skew == b->down_skew && a->upp_bias == b->upp_bias &&
         a->upp_skew == b->upp_skew && a->beg_l == b->beg_l && a->end_l == b->end_l;
}

int orc_collision_check(int variant, int N, double delta, const orc_cube *cubes_in,
                        const int *counts, int num_obs, const double *x_ref,
                        const double *y_ref, orc_cube *temp, int cap) {
  int total = 0;
  for (int j = 0; j < num_obs; j++) total += counts[j];
  orc_cube *cubes = (orc_cube *)malloc((size_t)(total + 1) * sizeof(orc_cube));
  memcpy(cubes, cubes_in, (size_t)total * sizeof(orc_cube));
  int nt = 0;
  int count = 0;
  int base = 0;
  /* solve_3d.cc:525-612.  `count` is NOT reset between cubes; the `k++` at
   * :595-596 is dead code (count was just reset or is <= 2). */
  for (int j = 0; j < num_obs; j++) {
    for (int k = 0; k < counts[j]; k++) {
      orc_cube *c = &cubes[base + k];
      for (int i = 0; i < N; i++) {
        int pos = 0, neg = 0;
        double ts = x_ref[i], tl = y_ref[i], tt = (double)i;
        if (tl <= c->end_l && tl >= c->beg_l) {
          double d;
          d = (ts - c->down_bias) * (c->beg_t - c->beg_t) -
              (tt - c->beg_t) * (c->upp_bias - c->down_bias);
          if (d > 0) pos++;
          if (d < 0) neg++;
          if (pos > 0 && neg > 0) continue;
          d = (ts - c->upp_bias) * (c->end_t - c->beg_t) -
              (tt - c->beg_t) * (c->upp_skew * delta + c->upp_bias - c->upp_bias);
          if (d > 0) pos++;
          if (d < 0) neg++;
          if (pos > 0 && neg > 0) continue;
          d = (ts - c->upp_bias - c->upp_skew * delta) * (c->end_t - c->end_t) -
              (tt - c->end_t) *
                  (c->down_skew * delta + c->down_bias - c->upp_skew * delta - c->upp_bias);
          if (d > 0) pos++;
          if (d < 0) neg++;
          if (pos > 0 && neg > 0) continue;
          d = (ts - c->down_bias - c->down_skew * delta) * (c->beg_t - c->end_t) -
              (tt - c->end_t) * (c->down_bias - c->down_skew * delta - c->down_bias);
          if (d > 0) pos++;
          if (d < 0) neg++;
          if (pos > 0 && neg > 0) continue;
          count++;
          if (count > 2) {
            c->count = count;
            if (nt + 1 > cap) { free(cubes); return -1; }
            temp[nt++] = *c;
            count = 0;
          } else {
            c->count = count;
          }
        }
      }
    }
    base += counts[j];
  }
  free(cubes);
  if (nt == 0) return -2; /* reference: size()-1 underflow -> UB */

  /* de-dup: solve_3d.cc:617-628 */
  for (int i = 0; i < nt - 1; i++) {
    for (int j = i + 1; j < nt; j++) {
      if (cube_same(&temp[i], &temp[j])) {
        memmove(&temp[j], &temp[j + 1], (size_t)(nt - j - 1) * sizeof(orc_cube));
        nt--;
        j--;
      }
    }
  }

  if (variant == ORC_TRAPEZOID) {
    /* std::sort by beg_t (:630).  libstdc++ uses a stable insertion sort for
     * n <= 16; insertion sort is used here for all n (ties keep their order). */
    for (int i = 1; i < nt; i++) {
      orc_cube v = temp[i];
      int j = i - 1;
      while (j >= 0 && v.beg_t < temp[j].beg_t) { temp[j + 1] = temp[j]; j--; }
      temp[j + 1] = v;
    }
    /* reorder for l-continuity: :639-673 */
    for (int i = 0; i < nt - 1; i++) {
      for (int j = i + 1; j < nt; j++) {
        if (temp[i].beg_l == temp[j].beg_l && j - i == 1) {
          break;
        } else {
          for (int k = j + 1; k < nt; k++) {
            if (temp[i].beg_l == temp[k].beg_l && temp[i].end_t == temp[k].beg_t) {
              orc_cube sw = temp[j]; temp[j] = temp[k]; temp[k] = sw;
              break;
            }
          }
        }
      }
    }
    /* overlaps: :678-703 (only j = i+1 because of the trailing break) */
    for (int i = 0; i < nt - 1; i++) {
      for (int j = i + 1; j < nt; j++) {
        if (temp[i].beg_t == temp[j].beg_t && temp[i].end_t == temp[j].end_t) {
          int diff = (temp[i].end_t - temp[i].beg_t) / 2;
          temp[i].end_t = temp[i].end_t - diff;
          temp[i].t = (temp[i].end_t - temp[i].beg_t) * delta;
          temp[j].beg_t = temp[j].beg_t + diff;
          temp[j].t = (temp[j].end_t - temp[j].beg_t) * delta;
        } else if (temp[i].beg_t > temp[j].beg_t && temp[i].end_t <= temp[j].end_t) {
          int diff = (temp[i].end_t - temp[i].beg_t) / 2;
          if (diff > 1) {
            temp[i].end_t = temp[i].end_t - diff;
            temp[i].t = (temp[i].end_t - temp[i].beg_t) * delta;
          }
          temp[j].beg_t = temp[i].end_t;
          temp[j].t = (temp[j].end_t - temp[j].beg_t) * delta;
        }
        break;
      }
    }
  } else {
    /* cuboid_3d.cc:553-567: no sort, no reorder, all j > i, diff = /3 */
    for (int i = 0; i < nt - 1; i++) {
      for (int j = i + 1; j < nt; j++) {
        if (temp[i].beg_t == temp[j].beg_t && temp[i].end_t == temp[j].end_t) {
          int diff = (temp[i].end_t - temp[i].beg_t) / 3;
          temp[i].end_t = temp[i].end_t - diff;
          temp[i].t = (temp[i].end_t - temp[i].beg_t) * delta;
          temp[j].beg_t = temp[j].beg_t + diff;
          temp[j].t = (temp[j].end_t - temp[j].beg_t) * delta;
        }
      }
    }
  }
  (void)N;
  return nt;
}

/* ------------------------------------------------------------------------ */
/* Assembly                                                                  */
/* ------------------------------------------------------------------------ */
/* Bernstein -> monomial, solve_3d.cc:122-127: row = power, col = ctrl pt.   */
static const double M_B2M[6][6] = {
    {1, 0, 0, 0, 0, 0},      {-5, 5, 0, 0, 0, 0},      {10, -20, 10, 0, 0, 0},
    {-10, 30, -30, 10, 0, 0}, {5, -20, 30, -20, 5, 0}, {-1, 5, -10, 10, -5, 1}};

/* MQM[d] = M^T pQp[d] M, solve_3d.cc:87-143 */
static void compute_mqm(const double w[4], double MQM[4][6][6]) {
  double pQp[4][6][6];
  memset(pQp, 0, sizeof(pQp));
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      pQp[0][i][j] = w[0] / (i + j + 1);
      if (i >= 1 && j >= 1) pQp[1][i][j] = (w[1] * i * j) / (i + j - 1);
      if (i >= 2 && j >= 2) pQp[2][i][j] = (w[2] * i * j * (i - 1) * (j - 1)) / (i + j - 3);
      if (i >= 3 && j >= 3)
        pQp[3][i][j] = (w[3] * i * j * (i - 1) * (j - 1) * (i - 2) * (j - 2)) / (i + j - 5);
    }
  for (int d = 0; d < 4; d++) {
    double T[6][6];
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 6; j++) {
        double s = 0;
        for (int k = 0; k < 6; k++) s += M_B2M[k][i] * pQp[d][k][j]; /* M^T pQp */
        T[i][j] = s;
      }
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 6; j++) {
        double s = 0;
        for (int k = 0; k < 6; k++) s += T[i][k] * M_B2M[k][j];
        MQM[d][i][j] = s;
      }
  }
}

typedef struct { orc_int row; double val; } nz_t;
typedef struct { nz_t *e; int n, cap; } col_t;
static void col_push(col_t *c, orc_int row, double val) {
  if (c->n == c->cap) { c->cap = c->cap ? 2 * c->cap : 16; c->e = (nz_t *)realloc(c->e, (size_t)c->cap * sizeof(nz_t)); }
  c->e[c->n].row = row; c->e[c->n].val = val; c->n++;
}

static int clampi(int i, int hi) { return i < 0 ? 0 : (i > hi ? hi : i); }

int orc_assemble(int variant, int S, const orc_cube *nc, const orc_qp_params *pp, orc_qp *qp) {
  memset(qp, 0, sizeof(*qp));
  if (S < 1) return -1;
  const int n_poly = N_POLY, traj_order = TRAJ_ORDER;
  const int nvar = S * n_poly; /* per axis */
  const int n = 2 * nvar;
  const int m = 2 * (S * (3 * n_poly - 3 + n_poly - 3) + 3 + 3 * (S - 1)); /* :785 */
  qp->n = n; qp->m = m;
  const int N = pp->N;

  /* FormulateProblem :1159-1166: ref sampled at knots 10k, 10k+1 (clamped). */
  double *x_skew = (double *)malloc(sizeof(double) * S * 4);
  double *x_bias = x_skew + S, *y_skew = x_skew + 2 * S, *y_bias = x_skew + 3 * S;
  for (int k = 0; k < S; k++) {
    int i0 = clampi(k * 10, N - 1), i1 = clampi(k * 10 + 1, N - 1);
    x_skew[k] = (pp->x_ref[i1] - pp->x_ref[i0]) / pp->delta;
    x_bias[k] = pp->x_ref[i0];
    y_skew[k] = (pp->y_ref[i1] - pp->y_ref[i0]) / pp->delta;
    y_bias[k] = pp->y_ref[i0];
  }

  /* ---- P: CalculateKernel :70-224 ---- */
  double MQM_x[4][6][6], MQM_y[4][6][6];
  compute_mqm(pp->w_s, MQM_x);
  compute_mqm(pp->w_l, MQM_y);
  qp->P_nnz = 2 * S * 21;
  qp->P_p = (orc_int *)malloc(sizeof(orc_int) * (n + 1));
  qp->P_i = (orc_int *)malloc(sizeof(orc_int) * qp->P_nnz);
  qp->P_x = (double *)malloc(sizeof(double) * qp->P_nnz);
  int idx = 0, sub_shift = 0, col = 0;
  for (int axis = 0; axis < 2; axis++) {
    double(*MQM)[6][6] = axis == 0 ? MQM_x : MQM_y;
    double w_end = axis == 0 ? pp->weight_end_s : pp->weight_end_l;
    for (int k = 0; k < S; k++) {
      double t = nc[k].t;
      for (int j = 0; j < n_poly; j++) {
        qp->P_p[col++] = idx;
        for (int i = 0; i < n_poly; i++) {
          if (j >= i) {
            double mval = pow(t, 3) * MQM[0][i][j] + t * MQM[1][i][j] + MQM[2][i][j] / t +
                          MQM[3][i][j] / pow(t, 3);
            qp->P_i[idx] = sub_shift + i;
            if ((k == S - 1) && (i == n_poly - 1) && (j == n_poly - 1)) mval = mval + w_end * t * t;
            qp->P_x[idx] = 2.0 * mval;
            idx++;
          }
        }
      }
      sub_shift += n_poly;
    }
  }
  qp->P_p[col] = idx;

  /* ---- q: CalculateOffset :226-321 ---- */
  qp->q = (double *)calloc(n, sizeof(double));
  for (int axis = 0; axis < 2; axis++) {
    const double *skew = axis == 0 ? x_skew : y_skew, *bias = axis == 0 ? x_bias : y_bias;
    double w_ref = axis == 0 ? pp->w_s[0] : pp->w_l[0];
    double w_d = axis == 0 ? pp->w_s[1] : pp->w_l[1];
    double d_ref = axis == 0 ? pp->ds_ref : pp->dl_ref;
    const double *ref = axis == 0 ? pp->x_ref : pp->y_ref;
    for (int k = 0; k < S; k++) {
      double t = nc[k].t, q_p[6];
      for (int i = 0; i < n_poly; i++) {
        q_p[i] = 0.0;
        q_p[i] += -2.0 * pow(t, 3) * w_ref * skew[k] / (i + 2);
        q_p[i] += -2.0 * pow(t, 2) * w_ref * bias[k] / (i + 1);
        if (i > 0) q_p[i] += -2.0 * w_d * d_ref * t;
      }
      for (int j = 0; j < n_poly; j++) {
        double s = 0;
        for (int i = 0; i < n_poly; i++) s += q_p[i] * M_B2M[i][j];
        qp->q[axis * nvar + k * n_poly + j] = s;
      }
    }
    /* :268 / :315 -- multiplies by d_ref (not a weight): bug-compatible */
    qp->q[axis * nvar + nvar - 1] -= d_ref * 2.0 * ref[N - 1] * nc[S - 1].t;
  }

  /* ---- A,l,u: CalculateAffineConstraint :779-1129 / cuboid :632-988 ---- */
  col_t *vars = (col_t *)calloc(n, sizeof(col_t));
  qp->l = (double *)malloc(sizeof(double) * m);
  qp->u = (double *)malloc(sizeof(double) * m);
  double aval[3] = {1.0 * traj_order * (traj_order - 1), -2.0 * traj_order * (traj_order - 1),
                    1.0 * traj_order * (traj_order - 1)};
  double jv = 1.0 * traj_order * (traj_order - 1) * (traj_order - 2);
  double jval[4] = {-1.0 * jv, 3.0 * jv, -3.0 * jv, 1.0 * jv};
  int ci = 0;
  for (int axis = 0; axis < 2; axis++) {
    const int off = axis * nvar;
    int var_shift = 0;
    for (int k = 0; k < S; k++) {
      const orc_cube *c = &nc[k];
      double t = c->t;
      /* safety rows */
      if (axis == 0) {
        if (variant == ORC_TRAPEZOID) {
          for (int i = 0; i < n_poly; i++) {
            col_push(&vars[off + var_shift + i], ci, 1.0 * t);
            qp->l[ci] = c->down_bias + c->down_skew * (i / 5.0) * t; /* inv_M(i,1) = i/5 */
            qp->u[ci] = c->upp_bias + c->upp_skew * (i / 5.0) * t;
            ++ci;
          }
        } else { /* cuboid_3d.cc:677-697 */
          double lb = 0, ub = 100;
          for (int i = 0; i < n_poly; i++) {
            lb = fmax(lb, c->down_bias + c->down_skew * (i / 5.0) * t);
            ub = fmin(ub, c->upp_bias + c->upp_skew * (i / 5.0) * t);
          }
          for (int i = 0; i < n_poly; i++) {
            col_push(&vars[off + var_shift + i], ci, 1.0 * t);
            qp->l[ci] = lb; qp->u[ci] = ub; ++ci;
          }
        }
      } else {
        for (int i = 0; i < n_poly; i++) {
          col_push(&vars[off + var_shift + i], ci, 1.0 * t);
          if (variant == ORC_TRAPEZOID) { /* :965-966 */
            qp->l[ci] = c->l_down_bias + c->l_down_skew * (i / 5.0) * t;
            qp->u[ci] = c->l_upp_bias + c->l_upp_skew * (i / 5.0) * t;
          } else { /* cuboid :826-827 */
            qp->l[ci] = c->beg_l; qp->u[ci] = c->end_l;
          }
          ++ci;
        }
      }
      /* physical rows */
      double d_lo = 0.0, d_hi = 1000.0, dd_lo = -1000.0, dd_hi = 1000.0;
      if (axis == 0) { /* :835-845 ; ddx_bounds_ is uniform (set_ddx_bounds scalar) */
        for (int i = c->beg_t; i <= c->end_t; i++) {
          int ii = clampi(i, N - 1);
          d_lo = fmax(pp->dx_bounds[2 * ii], d_lo);
          d_hi = fmin(pp->dx_bounds[2 * ii + 1], d_hi);
          dd_lo = fmax(pp->dds[0], dd_lo);
          dd_hi = fmin(pp->dds[1], dd_hi);
        }
      }
      for (int i = 0; i < n_poly - 1; i++) {
        col_push(&vars[off + var_shift + i], ci, -1.0 * traj_order);
        col_push(&vars[off + var_shift + i + 1], ci, 1.0 * traj_order);
        if (axis == 0) { qp->l[ci] = d_lo; qp->u[ci] = d_hi; }
        else { /* :1003-1004: dy_bounds_ indexed by CONTROL-POINT index i */
          int ii = clampi(i, N - 1);
          qp->l[ci] = pp->dy_bounds[2 * ii]; qp->u[ci] = pp->dy_bounds[2 * ii + 1];
        }
        ++ci;
      }
      for (int i = 0; i < n_poly - 2; i++) {
        col_push(&vars[off + var_shift + i], ci, aval[0]);
        col_push(&vars[off + var_shift + i + 1], ci, aval[1]);
        col_push(&vars[off + var_shift + i + 2], ci, aval[2]);
        if (axis == 0) { qp->l[ci] = dd_lo * t; qp->u[ci] = dd_hi * t; }
        else { qp->l[ci] = pp->ddl[0] * t; qp->u[ci] = pp->ddl[1] * t; } /* :1019-1020, uniform */
        ++ci;
      }
      for (int i = 0; i < n_poly - 3; i++) {
        col_push(&vars[off + var_shift + i], ci, jval[0]);
        col_push(&vars[off + var_shift + i + 1], ci, jval[1]);
        col_push(&vars[off + var_shift + i + 2], ci, jval[2]);
        col_push(&vars[off + var_shift + i + 3], ci, jval[3]);
        const double *b3 = axis == 0 ? pp->ddds : pp->dddl;
        qp->l[ci] = b3[0] * t * t; qp->u[ci] = b3[1] * t * t;
        ++ci;
      }
      var_shift += n_poly;
    }
    /* init rows :896-912 */
    const double *init = axis == 0 ? pp->init_s : pp->init_l;
    col_push(&vars[off + 0], ci, 1.0 * nc[0].t);
    qp->l[ci] = init[0]; qp->u[ci] = init[0]; ++ci;
    col_push(&vars[off + 0], ci, -1.0 * traj_order);
    col_push(&vars[off + 1], ci, 1.0 * traj_order);
    qp->l[ci] = init[1]; qp->u[ci] = init[1]; ++ci;
    col_push(&vars[off + 0], ci, aval[0]);
    col_push(&vars[off + 1], ci, aval[1]);
    col_push(&vars[off + 2], ci, aval[2]);
    qp->l[ci] = init[2] * nc[0].t; qp->u[ci] = init[2] * nc[0].t; ++ci;
    /* joints :918-949 */
    for (int k = 0; k < S - 1; k++) {
      int ss = (k + 1) * n_poly;
      double tk = nc[k].t, tk1 = nc[k + 1].t;
      col_push(&vars[off + ss - 1], ci, -1.0 * tk);
      col_push(&vars[off + ss], ci, 1.0 * tk1);
      qp->l[ci] = 0.0; qp->u[ci] = 0.0; ++ci;
      col_push(&vars[off + ss - 2], ci, -1.0);
      col_push(&vars[off + ss - 1], ci, 1.0);
      col_push(&vars[off + ss], ci, 1.0);
      col_push(&vars[off + ss + 1], ci, -1.0);
      qp->l[ci] = 0.0; qp->u[ci] = 0.0; ++ci;
      col_push(&vars[off + ss - 3], ci, 1.0 * tk1);
      col_push(&vars[off + ss - 2], ci, -2.0 * tk1);
      col_push(&vars[off + ss - 1], ci, 1.0 * tk1);
      col_push(&vars[off + ss], ci, -1.0 * tk);
      col_push(&vars[off + ss + 1], ci, 2.0 * tk);
      col_push(&vars[off + ss + 2], ci, -1.0 * tk);
      qp->l[ci] = 0.0; qp->u[ci] = 0.0; ++ci;
    }
  }
  int rc = 0;
  if (ci != m) rc = -3; /* CHECK_EQ :1107 */
  int nnz = 0;
  for (int i = 0; i < n; i++) nnz += vars[i].n;
  qp->A_nnz = nnz;
  qp->A_p = (orc_int *)malloc(sizeof(orc_int) * (n + 1));
  qp->A_i = (orc_int *)malloc(sizeof(orc_int) * (nnz + 1));
  qp->A_x = (double *)malloc(sizeof(double) * (nnz + 1));
  int p = 0;
  for (int i = 0; i < n; i++) {
    qp->A_p[i] = p;
    for (int e = 0; e < vars[i].n; e++) { qp->A_x[p] = vars[i].e[e].val; qp->A_i[p] = vars[i].e[e].row; p++; }
    free(vars[i].e);
  }
  qp->A_p[n] = p;
  free(vars);
  free(x_skew);
  return rc;
}

void orc_qp_free(orc_qp *qp) {
  free(qp->P_p); free(qp->P_i); free(qp->P_x);
  free(qp->A_p); free(qp->A_i); free(qp->A_x);
  free(qp->q); free(qp->l); free(qp->u);
  memset(qp, 0, sizeof(*qp));
}

/* ------------------------------------------------------------------------ */
/* OSQP-style ADMM (restated from the published algorithm; OSQP 0.5.0        */
/* behaviour; constants RHO_MIN 1e-6, RHO_MAX 1e6, RHO_EQ_OVER_RHO_INEQ 1e3,   */
/* RHO_TOL 1e-4, MIN_SCALING 1e-4, MAX_SCALING 1e4, OSQP_INFTY 1e20).          */
/* The KKT system is solved in its reduced form (P+sigma I+A'RA) x = rhs with */
/* a banded Cholesky factor; z~ = A x~ is algebraically identical to OSQP's    */
/* quasi-definite LDL' solve.                                                  */
/* ------------------------------------------------------------------------ */
#define ORC_RHO_MIN 1e-6
#define ORC_RHO_MAX 1e6
#define ORC_RHO_EQ_OVER_INEQ 1e3
#define ORC_RHO_TOL 1e-4
#define ORC_MIN_SCALING 1e-4
#define ORC_MAX_SCALING 1e4
#define ORC_INFTY 1e20

void orc_settings_reference(orc_settings *s) {
  s->rho = 0.1; s->sigma = 1e-6; s->alpha = 1.6;          /* OSQP defaults */
  s->eps_abs = 1e-5; s->eps_rel = 1e-5;                     /* solve_3d.cc:1238-1239 */
  s->eps_prim_inf = 0.000025; s->eps_dual_inf = 0.000025;   /* :1454-1455 */
  s->max_iter = 5000;                                       /* trp_wrapper.cpp:191 */
  s->scaling = 4;                                           /* :1242 */
  s->scaled_termination = 1;                                /* :1459 */
  s->check_termination = 25;                                /* OSQP default */
  s->adaptive_rho = 1;                                      /* OSQP default */
  /* OSQP derives the interval from wall-clock setup time (unpinnable); 25 =
   * one check_termination period, the value it rounds to for small KKTs. */
  s->adaptive_rho_interval = 25;
  s->adaptive_rho_tolerance = 5.0;
  s->polish = 0;                                            /* :1243 */
}

void orc_settings_tight(orc_settings *s) {
  orc_settings_reference(s);
  s->eps_abs = 1e-9; s->eps_rel = 1e-9;
  s->max_iter = 200000;
  s->scaled_termination = 0;
  s->polish = 1;
}

typedef struct {
  int n, m, hb;
  /* scaled data */
  orc_int *Pp, *Pi; double *Px;
  orc_int *Ap, *Ai; double *Ax;
  double *q, *l, *u;
  double *D, *E, *Dinv, *Einv; double c, cinv;
  double *rho_vec, *rho_inv; int *ctype;
  double *band; /* lower band Cholesky: band[i*(hb+1)+d] = L(i, i-d) */
  double *x, *z, *y, *x_prev, *z_prev, *xt, *zt, *dx, *dy;
  double *Axv, *Px_v, *Aty;
  double rho;
} orc_work;

static double vnorm_inf(const double *v, int n) {
  double r = 0; for (int i = 0; i < n; i++) { double a = fabs(v[i]); if (a > r) r = a; } return r;
}
static double limit_scaling(double v) {
  v = v < ORC_MIN_SCALING ? 1.0 : v;
  v = v > ORC_MAX_SCALING ? ORC_MAX_SCALING : v;
  return v;
}
static void mat_vec_A(const orc_work *w, const double *x, double *y) { /* y = A x */
  for (int i = 0; i < w->m; i++) y[i] = 0;
  for (int j = 0; j < w->n; j++)
    for (orc_int p = w->Ap[j]; p < w->Ap[j + 1]; p++) y[w->Ai[p]] += w->Ax[p] * x[j];
}
static void mat_tvec_A(const orc_work *w, const double *y, double *x) { /* x = A' y */
  for (int j = 0; j < w->n; j++) {
    double s = 0;
    for (orc_int p = w->Ap[j]; p < w->Ap[j + 1]; p++) s += w->Ax[p] * y[w->Ai[p]];
    x[j] = s;
  }
}
static void mat_vec_P(const orc_work *w, const double *x, double *y) { /* y = P x, P sym upper */
  for (int i = 0; i < w->n; i++) y[i] = 0;
  for (int j = 0; j < w->n; j++)
    for (orc_int p = w->Pp[j]; p < w->Pp[j + 1]; p++) {
      orc_int i = w->Pi[p];
      y[i] += w->Px[p] * x[j];
      if (i != j) y[j] += w->Px[p] * x[i];
    }
}

static void scale_data(orc_work *w, int iters) {
  int n = w->n, m = w->m;
  double *Dt = (double *)malloc(sizeof(double) * n), *Et = (double *)malloc(sizeof(double) * m);
  for (int i = 0; i < n; i++) w->D[i] = 1.0;
  for (int i = 0; i < m; i++) w->E[i] = 1.0;
  w->c = 1.0;
  for (int it = 0; it < iters; it++) {
    for (int j = 0; j < n; j++) Dt[j] = 0;
    for (int i = 0; i < m; i++) Et[i] = 0;
    for (int j = 0; j < n; j++)
      for (orc_int p = w->Pp[j]; p < w->Pp[j + 1]; p++) {
        double a = fabs(w->Px[p]); orc_int i = w->Pi[p];
        if (a > Dt[j]) Dt[j] = a;
        if (i != j && a > Dt[i]) Dt[i] = a;
      }
    for (int j = 0; j < n; j++)
      for (orc_int p = w->Ap[j]; p < w->Ap[j + 1]; p++) {
        double a = fabs(w->Ax[p]);
        if (a > Dt[j]) Dt[j] = a;
        if (a > Et[w->Ai[p]]) Et[w->Ai[p]] = a;
      }
    for (int j = 0; j < n; j++) Dt[j] = 1.0 / sqrt(limit_scaling(Dt[j]));
    for (int i = 0; i < m; i++) Et[i] = 1.0 / sqrt(limit_scaling(Et[i]));
    for (int j = 0; j < n; j++)
      for (orc_int p = w->Pp[j]; p < w->Pp[j + 1]; p++) w->Px[p] *= Dt[j] * Dt[w->Pi[p]];
    for (int j = 0; j < n; j++)
      for (orc_int p = w->Ap[j]; p < w->Ap[j + 1]; p++) w->Ax[p] *= Dt[j] * Et[w->Ai[p]];
    for (int j = 0; j < n; j++) { w->q[j] *= Dt[j]; w->D[j] *= Dt[j]; }
    for (int i = 0; i < m; i++) w->E[i] *= Et[i];
    /* cost scaling */
    for (int j = 0; j < n; j++) Dt[j] = 0;
    for (int j = 0; j < n; j++)
      for (orc_int p = w->Pp[j]; p < w->Pp[j + 1]; p++) {
        double a = fabs(w->Px[p]); orc_int i = w->Pi[p];
        if (a > Dt[j]) Dt[j] = a;
        if (i != j && a > Dt[i]) Dt[i] = a;
      }
    double c_temp = 0; for (int j = 0; j < n; j++) c_temp += Dt[j]; c_temp /= n;
    double qn = limit_scaling(vnorm_inf(w->q, n));
    c_temp = c_temp > qn ? c_temp : qn;
    c_temp = 1.0 / limit_scaling(c_temp);
    for (orc_int p = 0; p < w->Pp[n]; p++) w->Px[p] *= c_temp;
    for (int j = 0; j < n; j++) w->q[j] *= c_temp;
    w->c *= c_temp;
  }
  for (int j = 0; j < n; j++) w->Dinv[j] = 1.0 / w->D[j];
  for (int i = 0; i < m; i++) { w->Einv[i] = 1.0 / w->E[i]; w->l[i] *= w->E[i]; w->u[i] *= w->E[i]; }
  w->cinv = 1.0 / w->c;
  free(Dt); free(Et);
}

static void set_rho_vec(orc_work *w, double rho) {
  w->rho = rho < ORC_RHO_MIN ? ORC_RHO_MIN : (rho > ORC_RHO_MAX ? ORC_RHO_MAX : rho);
  for (int i = 0; i < w->m; i++) {
    if (w->l[i] < -ORC_INFTY * ORC_MIN_SCALING && w->u[i] > ORC_INFTY * ORC_MIN_SCALING) {
      w->ctype[i] = -1; w->rho_vec[i] = ORC_RHO_MIN;
    } else if (w->u[i] - w->l[i] < ORC_RHO_TOL) {
      w->ctype[i] = 1; w->rho_vec[i] = ORC_RHO_EQ_OVER_INEQ * w->rho;
    } else {
      w->ctype[i] = 0; w->rho_vec[i] = w->rho;
    }
    w->rho_inv[i] = 1.0 / w->rho_vec[i];
  }
}

/* Build K = P + sigma I + A' R A in lower-band storage and Cholesky-factor it. */
static int factor_kkt(orc_work *w, double sigma, const orc_int *Rp, const orc_int *Rj,
                      const double *Rx /* CSR of scaled A */) {
  int n = w->n, hb = w->hb, ld = hb + 1;
  double *B = w->band;
  memset(B, 0, sizeof(double) * (size_t)n * ld);
  for (int j = 0; j < n; j++)
    for (orc_int p = w->Pp[j]; p < w->Pp[j + 1]; p++) {
      int i = (int)w->Pi[p]; /* i <= j: entry (j,i) of lower */
      B[j * ld + (j - i)] += w->Px[p];
    }
  for (int j = 0; j < n; j++) B[j * ld] += sigma;
  for (int r = 0; r < w->m; r++) {
    double rho = w->rho_vec[r];
    for (orc_int a = Rp[r]; a < Rp[r + 1]; a++)
      for (orc_int b = Rp[r]; b <= a; b++) {
        int ia = (int)Rj[a], ib = (int)Rj[b];
        int hi = ia > ib ? ia : ib, lo = ia > ib ? ib : ia;
        B[hi * ld + (hi - lo)] += rho * Rx[a] * Rx[b];
      }
  }
  /* banded Cholesky, in place */
  for (int i = 0; i < n; i++) {
    int j0 = i - hb < 0 ? 0 : i - hb;
    for (int j = j0; j <= i; j++) {
      double s = B[i * ld + (i - j)];
      int k0 = j - hb < 0 ? 0 : j - hb; if (k0 < j0) k0 = j0;
      for (int k = k0; k < j; k++) s -= B[i * ld + (i - k)] * B[j * ld + (j - k)];
      if (i == j) { if (s <= 0) return -1; B[i * ld] = sqrt(s); }
      else B[i * ld + (i - j)] = s / B[j * ld];
    }
  }
  return 0;
}
static void solve_kkt(const orc_work *w, double *b) {
  int n = w->n, hb = w->hb, ld = hb + 1; const double *B = w->band;
  for (int i = 0; i < n; i++) {
    double s = b[i]; int j0 = i - hb < 0 ? 0 : i - hb;
    for (int j = j0; j < i; j++) s -= B[i * ld + (i - j)] * b[j];
    b[i] = s / B[i * ld];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = b[i]; int j1 = i + hb > n - 1 ? n - 1 : i + hb;
    for (int j = i + 1; j <= j1; j++) s -= B[j * ld + (j - i)] * b[j];
    b[i] = s / B[i * ld];
  }
}

static double compute_obj(orc_work *w, const double *x) {
  mat_vec_P(w, x, w->Px_v);
  double o = 0;
  for (int i = 0; i < w->n; i++) o += 0.5 * x[i] * w->Px_v[i] + w->q[i] * x[i];
  return o * w->cinv;
}

/* dense Gaussian elimination with partial pivoting: solves M s = b in place  */
static int dense_solve(double *Mx, double *b, int N) {
  for (int c = 0; c < N; c++) {
    int piv = c; double best = fabs(Mx[(size_t)c * N + c]);
    for (int r = c + 1; r < N; r++) { double a = fabs(Mx[(size_t)r * N + c]); if (a > best) { best = a; piv = r; } }
    if (best == 0) return -1;
    if (piv != c) {
      for (int k = 0; k < N; k++) { double t = Mx[(size_t)c * N + k]; Mx[(size_t)c * N + k] = Mx[(size_t)piv * N + k]; Mx[(size_t)piv * N + k] = t; }
      double t = b[c]; b[c] = b[piv]; b[piv] = t;
    }
    double inv = 1.0 / Mx[(size_t)c * N + c];
    for (int r = c + 1; r < N; r++) {
      double f = Mx[(size_t)r * N + c] * inv;
      if (f == 0) continue;
      for (int k = c; k < N; k++) Mx[(size_t)r * N + k] -= f * Mx[(size_t)c * N + k];
      b[r] -= f * b[c];
    }
  }
  for (int r = N - 1; r >= 0; r--) {
    double s = b[r];
    for (int k = r + 1; k < N; k++) s -= Mx[(size_t)r * N + k] * b[k];
    b[r] = s / Mx[(size_t)r * N + r];
  }
  return 0;
}

/* Oracle-side polish on the UNSCALED problem: guess the active set from the
 * ADMM duals, then solve the equality-constrained QP
 *   [P+dI  Aact'] [x]   [-q ]
 *   [Aact  -dI  ] [nu] = [b  ]   with iterative refinement against d=0
 * (OSQP's polish step; delta=1e-7, 10 refinement passes).  Accepted only if
 * it keeps primal feasibility and the dual signs.                            */
static int polish_unscaled(const orc_qp *qp, double *x, double *y) {
  int n = qp->n, m = qp->m;
  double *Ax = (double *)calloc(m, sizeof(double));
  for (int j = 0; j < n; j++)
    for (orc_int p = qp->A_p[j]; p < qp->A_p[j + 1]; p++) Ax[qp->A_i[p]] += qp->A_x[p] * x[j];
  int *act = (int *)malloc(sizeof(int) * m); double *bnd = (double *)malloc(sizeof(double) * m);
  int *side = (int *)malloc(sizeof(int) * m);
  int na = 0;
  for (int i = 0; i < m; i++) {
    double scale = 1.0 + fabs(qp->l[i]) + fabs(qp->u[i]);
    int lo = (Ax[i] - qp->l[i] < -y[i]) || (qp->u[i] - qp->l[i] < 1e-9 * scale);
    int up = (qp->u[i] - Ax[i] < y[i]);
    if (lo) { act[na] = i; bnd[na] = qp->l[i]; side[na] = -1; na++; }
    else if (up) { act[na] = i; bnd[na] = qp->u[i]; side[na] = 1; na++; }
  }
  int Nk = n + na; const double delta = 1e-7;
  int *rowpos = (int *)malloc(sizeof(int) * m);
  for (int i = 0; i < m; i++) rowpos[i] = -1;
  for (int a = 0; a < na; a++) rowpos[act[a]] = a;
  double *K0 = (double *)calloc((size_t)Nk * Nk, sizeof(double));
  for (int j = 0; j < n; j++)
    for (orc_int p = qp->P_p[j]; p < qp->P_p[j + 1]; p++) {
      int i = (int)qp->P_i[p];
      K0[(size_t)i * Nk + j] += qp->P_x[p];
      if (i != j) K0[(size_t)j * Nk + i] += qp->P_x[p];
    }
  for (int j = 0; j < n; j++)
    for (orc_int p = qp->A_p[j]; p < qp->A_p[j + 1]; p++) {
      int a = rowpos[qp->A_i[p]];
      if (a >= 0) { K0[(size_t)(n + a) * Nk + j] = qp->A_x[p]; K0[(size_t)j * Nk + (n + a)] = qp->A_x[p]; }
    }
  double *rhs = (double *)malloc(sizeof(double) * Nk), *sol = (double *)calloc(Nk, sizeof(double));
  double *res = (double *)malloc(sizeof(double) * Nk), *Kr = (double *)malloc(sizeof(double) * (size_t)Nk * Nk);
  for (int i = 0; i < n; i++) rhs[i] = -qp->q[i];
  for (int a = 0; a < na; a++) rhs[n + a] = bnd[a];
  int ok = 1;
  for (int pass = 0; pass < 10 && ok; pass++) {
    for (int i = 0; i < Nk; i++) {
      double s = rhs[i];
      for (int k = 0; k < Nk; k++) s -= K0[(size_t)i * Nk + k] * sol[k];
      res[i] = s;
    }
    memcpy(Kr, K0, sizeof(double) * (size_t)Nk * Nk);
    for (int i = 0; i < n; i++) Kr[(size_t)i * Nk + i] += delta;
    for (int a = 0; a < na; a++) Kr[(size_t)(n + a) * Nk + (n + a)] -= delta;
    if (dense_solve(Kr, res, Nk) != 0) { ok = 0; break; }
    for (int i = 0; i < Nk; i++) sol[i] += res[i];
  }
  int accepted = 0;
  if (ok) {
    double *Ax2 = (double *)calloc(m, sizeof(double));
    for (int j = 0; j < n; j++)
      for (orc_int p = qp->A_p[j]; p < qp->A_p[j + 1]; p++) Ax2[qp->A_i[p]] += qp->A_x[p] * sol[j];
    double viol = 0, dsign = 0;
    for (int i = 0; i < m; i++) {
      double v = fmax(qp->l[i] - Ax2[i], Ax2[i] - qp->u[i]); if (v > viol) viol = v;
    }
    for (int a = 0; a < na; a++) {
      int i = act[a];
      if (qp->u[i] - qp->l[i] < 1e-12) continue; /* equality: any sign */
      double nu = sol[n + a];
      double bad = side[a] > 0 ? -nu : nu; if (bad > dsign) dsign = bad;
    }
    if (viol < 1e-7 && dsign < 1e-7) {
      memcpy(x, sol, sizeof(double) * n);
      for (int i = 0; i < m; i++) y[i] = 0;
      for (int a = 0; a < na; a++) y[act[a]] = sol[n + a];
      accepted = 1;
    }
    free(Ax2);
  }
  free(Ax); free(act); free(bnd); free(side); free(rowpos); free(K0); free(rhs); free(sol); free(res); free(Kr);
  return accepted;
}

int orc_osqp_solve(const orc_qp *qp, const orc_settings *s, double *x_out, double *y_out,
                   orc_info *info) {
  int n = qp->n, m = qp->m;
  orc_work W; memset(&W, 0, sizeof(W)); orc_work *w = &W;
  w->n = n; w->m = m;
#define DUPL(dst, src, cnt, T) do { dst = (T *)malloc(sizeof(T) * (size_t)((cnt) + 1)); memcpy(dst, src, sizeof(T) * (size_t)(cnt)); } while (0)
  DUPL(w->Pp, qp->P_p, n + 1, orc_int); DUPL(w->Pi, qp->P_i, qp->P_nnz, orc_int); DUPL(w->Px, qp->P_x, qp->P_nnz, double);
  DUPL(w->Ap, qp->A_p, n + 1, orc_int); DUPL(w->Ai, qp->A_i, qp->A_nnz, orc_int); DUPL(w->Ax, qp->A_x, qp->A_nnz, double);
  DUPL(w->q, qp->q, n, double); DUPL(w->l, qp->l, m, double); DUPL(w->u, qp->u, m, double);
#define VEC(cnt) (double *)calloc((size_t)(cnt) + 1, sizeof(double))
  w->D = VEC(n); w->Dinv = VEC(n); w->E = VEC(m); w->Einv = VEC(m);
  w->rho_vec = VEC(m); w->rho_inv = VEC(m); w->ctype = (int *)calloc(m + 1, sizeof(int));
  w->x = VEC(n); w->z = VEC(m); w->y = VEC(m); w->x_prev = VEC(n); w->z_prev = VEC(m);
  w->xt = VEC(n); w->zt = VEC(m); w->dx = VEC(n); w->dy = VEC(m);
  w->Axv = VEC(m); w->Px_v = VEC(n); w->Aty = VEC(n);

  if (s->scaling > 0) scale_data(w, s->scaling);
  else {
    for (int i = 0; i < n; i++) w->D[i] = w->Dinv[i] = 1;
    for (int i = 0; i < m; i++) w->E[i] = w->Einv[i] = 1;
    w->c = w->cinv = 1;
  }

  /* CSR of scaled A + bandwidth of P + A'A */
  orc_int *Rp = (orc_int *)calloc(m + 2, sizeof(orc_int)), *Rj = (orc_int *)malloc(sizeof(orc_int) * (qp->A_nnz + 1));
  double *Rx = (double *)malloc(sizeof(double) * (qp->A_nnz + 1));
  for (orc_int p = 0; p < qp->A_nnz; p++) Rp[w->Ai[p] + 1]++;
  for (int i = 0; i < m; i++) Rp[i + 1] += Rp[i];
  { orc_int *nx = (orc_int *)malloc(sizeof(orc_int) * (m + 1)); memcpy(nx, Rp, sizeof(orc_int) * (m + 1));
    for (int j = 0; j < n; j++) for (orc_int p = w->Ap[j]; p < w->Ap[j + 1]; p++) { orc_int r = w->Ai[p]; Rj[nx[r]] = j; Rx[nx[r]] = w->Ax[p]; nx[r]++; }
    free(nx); }
  int hb = 0;
  for (int j = 0; j < n; j++) for (orc_int p = w->Pp[j]; p < w->Pp[j + 1]; p++) { int d = j - (int)w->Pi[p]; if (d > hb) hb = d; }
  for (int r = 0; r < m; r++) if (Rp[r + 1] > Rp[r]) { int d = (int)(Rj[Rp[r + 1] - 1] - Rj[Rp[r]]); if (d > hb) hb = d; }
  w->hb = hb;
  w->band = (double *)malloc(sizeof(double) * (size_t)n * (hb + 1));

  set_rho_vec(w, s->rho);
  int rc = factor_kkt(w, s->sigma, Rp, Rj, Rx);
  int status = -10, iter = 0, rho_updates = 0;
  double pri_res = 0, dua_res = 0, obj = 0;
  double *rhs = VEC(n), *tmp_m = VEC(m), *dy_prev = VEC(m);
  double eps_abs = s->eps_abs, eps_rel = s->eps_rel, eps_pinf = s->eps_prim_inf, eps_dinf = s->eps_dual_inf;

  if (rc == 0) {
    for (iter = 1; iter <= s->max_iter; iter++) {
      memcpy(w->x_prev, w->x, sizeof(double) * n);
      memcpy(w->z_prev, w->z, sizeof(double) * m);
      /* x~: (P+sigma I+A'RA) x~ = sigma x - q + A'(rho z - y) */
      for (int i = 0; i < m; i++) tmp_m[i] = w->rho_vec[i] * w->z_prev[i] - w->y[i];
      mat_tvec_A(w, tmp_m, rhs);
      for (int i = 0; i < n; i++) rhs[i] += s->sigma * w->x_prev[i] - w->q[i];
      solve_kkt(w, rhs);
      memcpy(w->xt, rhs, sizeof(double) * n);
      mat_vec_A(w, w->xt, w->zt);
      for (int i = 0; i < n; i++) { w->x[i] = s->alpha * w->xt[i] + (1.0 - s->alpha) * w->x_prev[i]; w->dx[i] = w->x[i] - w->x_prev[i]; }
      for (int i = 0; i < m; i++) {
        double zz = s->alpha * w->zt[i] + (1.0 - s->alpha) * w->z_prev[i] + w->rho_inv[i] * w->y[i];
        w->z[i] = zz < w->l[i] ? w->l[i] : (zz > w->u[i] ? w->u[i] : zz);
      }
      for (int i = 0; i < m; i++) {
        w->dy[i] = w->rho_vec[i] * (s->alpha * w->zt[i] + (1.0 - s->alpha) * w->z_prev[i] - w->z[i]);
        w->y[i] += w->dy[i];
      }
      int can_check = s->check_termination && (iter % s->check_termination == 0);
      int do_adapt = s->adaptive_rho && s->adaptive_rho_interval && (iter % s->adaptive_rho_interval == 0);
      int last = (iter == s->max_iter);
      if (can_check || do_adapt || last) {
        /* update_info: residuals (scaled or unscaled) */
        obj = compute_obj(w, w->x);
        mat_vec_A(w, w->x, w->Axv);
        for (int i = 0; i < m; i++) w->z_prev[i] = w->Axv[i] - w->z[i];
        mat_tvec_A(w, w->y, w->Aty);
        for (int i = 0; i < n; i++) w->x_prev[i] = w->q[i] + w->Px_v[i] + w->Aty[i];
        int unsc = (s->scaling > 0 && !s->scaled_termination);
        double nz, nAx, nq, nAty, nPx;
        if (unsc) {
          pri_res = 0; nz = 0; nAx = 0;
          for (int i = 0; i < m; i++) {
            double a = fabs(w->Einv[i] * w->z_prev[i]); if (a > pri_res) pri_res = a;
            a = fabs(w->Einv[i] * w->z[i]); if (a > nz) nz = a;
            a = fabs(w->Einv[i] * w->Axv[i]); if (a > nAx) nAx = a;
          }
          dua_res = 0; nq = 0; nAty = 0; nPx = 0;
          for (int i = 0; i < n; i++) {
            double a = fabs(w->Dinv[i] * w->x_prev[i]); if (a > dua_res) dua_res = a;
            a = fabs(w->Dinv[i] * w->q[i]); if (a > nq) nq = a;
            a = fabs(w->Dinv[i] * w->Aty[i]); if (a > nAty) nAty = a;
            a = fabs(w->Dinv[i] * w->Px_v[i]); if (a > nPx) nPx = a;
          }
          dua_res *= w->cinv; nq *= w->cinv; nAty *= w->cinv; nPx *= w->cinv;
        } else {
          pri_res = vnorm_inf(w->z_prev, m); nz = vnorm_inf(w->z, m); nAx = vnorm_inf(w->Axv, m);
          dua_res = vnorm_inf(w->x_prev, n); nq = vnorm_inf(w->q, n); nAty = vnorm_inf(w->Aty, n); nPx = vnorm_inf(w->Px_v, n);
        }
        if (can_check || last) {
          int approx_pass = 0;
          for (; approx_pass < 2; approx_pass++) {
            double ea = eps_abs, er = eps_rel, epi = eps_pinf, edi = eps_dinf;
            if (approx_pass) { if (!last) break; ea *= 10; er *= 10; epi *= 10; edi *= 10; }
            double eps_prim = ea + er * fmax(nz, nAx);
            double eps_dual = ea + er * fmax(fmax(nq, nAty), nPx);
            int prim_ok = (m == 0) || pri_res < eps_prim, dual_ok = dua_res < eps_dual;
            int prim_inf = 0, dual_inf = 0;
            if (!prim_ok) { /* is_primal_infeasible (all bounds finite here: no cone projection needed beyond INFTY test) */
              double ndy = 0;
              for (int i = 0; i < m; i++) {
                double d = w->dy[i];
                if (w->u[i] > ORC_INFTY * ORC_MIN_SCALING) { if (w->l[i] < -ORC_INFTY * ORC_MIN_SCALING) d = 0; else d = fmin(d, 0); }
                else if (w->l[i] < -ORC_INFTY * ORC_MIN_SCALING) d = fmax(d, 0);
                dy_prev[i] = d;
                double a = unsc ? fabs(w->E[i] * d) : fabs(d); if (a > ndy) ndy = a;
              }
              if (ndy > epi) {
                double lhs = 0;
                for (int i = 0; i < m; i++) lhs += w->u[i] * fmax(dy_prev[i], 0) + w->l[i] * fmin(dy_prev[i], 0);
                if (lhs < -epi * ndy) {
                  mat_tvec_A(w, dy_prev, rhs);
                  double nn = 0; for (int i = 0; i < n; i++) { double a = unsc ? fabs(w->Dinv[i] * rhs[i]) : fabs(rhs[i]); if (a > nn) nn = a; }
                  prim_inf = nn < epi * ndy;
                }
              }
            }
            if (!dual_ok) { /* is_dual_infeasible */
              double ndx = 0; for (int i = 0; i < n; i++) { double a = unsc ? fabs(w->D[i] * w->dx[i]) : fabs(w->dx[i]); if (a > ndx) ndx = a; }
              double csc = unsc ? w->c : 1.0;
              if (ndx > edi) {
                double qdx = 0; for (int i = 0; i < n; i++) qdx += w->q[i] * w->dx[i];
                if (qdx < -csc * edi * ndx) {
                  mat_vec_P(w, w->dx, rhs);
                  double nn = 0; for (int i = 0; i < n; i++) { double a = unsc ? fabs(w->Dinv[i] * rhs[i]) : fabs(rhs[i]); if (a > nn) nn = a; }
                  if (nn < csc * edi * ndx) {
                    mat_vec_A(w, w->dx, tmp_m);
                    dual_inf = 1;
                    for (int i = 0; i < m; i++) {
                      double v = unsc ? w->Einv[i] * tmp_m[i] : tmp_m[i];
                      if ((w->u[i] < ORC_INFTY * ORC_MIN_SCALING && v > edi * ndx) ||
                          (w->l[i] > -ORC_INFTY * ORC_MIN_SCALING && v < -edi * ndx)) { dual_inf = 0; break; }
                    }
                  }
                }
              }
            }
            if (prim_ok && dual_ok) { status = approx_pass ? 2 : 1; break; }
            if (prim_inf) { status = approx_pass ? 3 : -3; break; }
            if (dual_inf) { status = approx_pass ? 4 : -4; break; }
          }
          if (status != -10) break;
        }
        if (do_adapt && !last) {
          double pr = vnorm_inf(w->z_prev, m), dr = vnorm_inf(w->x_prev, n);
          double pn = fmax(vnorm_inf(w->z, m), vnorm_inf(w->Axv, m));
          double dn = fmax(fmax(vnorm_inf(w->q, n), vnorm_inf(w->Aty, n)), vnorm_inf(w->Px_v, n));
          pr /= (pn + 1e-10); dr /= (dn + 1e-10);
          double rho_new = w->rho * sqrt(pr / (dr + 1e-10));
          rho_new = fmin(fmax(rho_new, ORC_RHO_MIN), ORC_RHO_MAX);
          if (rho_new > w->rho * s->adaptive_rho_tolerance || rho_new < w->rho / s->adaptive_rho_tolerance) {
            set_rho_vec(w, rho_new);
            if (factor_kkt(w, s->sigma, Rp, Rj, Rx) != 0) { status = -10; rc = -1; break; }
            rho_updates++;
          }
        }
      }
    }
    if (iter > s->max_iter) iter = s->max_iter;
    if (status == -10 && rc == 0) status = -2; /* max iter reached */
  }

  /* unscale solution: x = D x, y = E y / c */
  for (int i = 0; i < n; i++) x_out[i] = w->D[i] * w->x[i];
  if (y_out) for (int i = 0; i < m; i++) y_out[i] = w->cinv * w->E[i] * w->y[i];
  if (s->polish && (status == 1 || status == 2)) {
    double *yt = y_out ? y_out : (double *)malloc(sizeof(double) * m);
    if (!y_out) for (int i = 0; i < m; i++) yt[i] = w->cinv * w->E[i] * w->y[i];
    if (polish_unscaled(qp, x_out, yt)) {
      double o = 0; /* objective on the unscaled data */
      double *Pv = (double *)calloc(n, sizeof(double));
      for (int j = 0; j < n; j++) for (orc_int p = qp->P_p[j]; p < qp->P_p[j + 1]; p++) { int i = (int)qp->P_i[p]; Pv[i] += qp->P_x[p] * x_out[j]; if (i != j) Pv[j] += qp->P_x[p] * x_out[i]; }
      for (int i = 0; i < n; i++) o += 0.5 * x_out[i] * Pv[i] + qp->q[i] * x_out[i];
      obj = o; free(Pv);
    }
    if (!y_out) free(yt);
  }
  if (info) { info->status = status; info->iter = iter; info->rho_updates = rho_updates; info->obj_val = obj; info->pri_res = pri_res; info->dua_res = dua_res; info->rho = w->rho; }

  free(w->Pp); free(w->Pi); free(w->Px); free(w->Ap); free(w->Ai); free(w->Ax); free(w->q); free(w->l); free(w->u);
  free(w->D); free(w->Dinv); free(w->E); free(w->Einv); free(w->rho_vec); free(w->rho_inv); free(w->ctype);
  free(w->x); free(w->z); free(w->y); free(w->x_prev); free(w->z_prev); free(w->xt); free(w->zt); free(w->dx); free(w->dy);
  free(w->Axv); free(w->Px_v); free(w->Aty); free(w->band); free(Rp); free(Rj); free(Rx); free(rhs); free(tmp_m); free(dy_prev);
  return rc;
}

void orc_kkt_residuals(const orc_qp *qp, const double *x, const double *y, double *res) {
  int n = qp->n, m = qp->m;
  double *g = (double *)calloc(n, sizeof(double)), *Ax = (double *)calloc(m, sizeof(double));
  for (int j = 0; j < n; j++) for (orc_int p = qp->P_p[j]; p < qp->P_p[j + 1]; p++) { int i = (int)qp->P_i[p]; g[i] += qp->P_x[p] * x[j]; if (i != j) g[j] += qp->P_x[p] * x[i]; }
  for (int j = 0; j < n; j++) { double s = 0; for (orc_int p = qp->A_p[j]; p < qp->A_p[j + 1]; p++) { s += qp->A_x[p] * y[qp->A_i[p]]; Ax[qp->A_i[p]] += qp->A_x[p] * x[j]; } g[j] += s + qp->q[j]; }
  res[0] = vnorm_inf(g, n);
  double viol = 0, comp = 0;
  for (int i = 0; i < m; i++) {
    double v = fmax(qp->l[i] - Ax[i], Ax[i] - qp->u[i]); if (v > viol) viol = v;
    double c = fmax(y[i], 0) * fabs(qp->u[i] - Ax[i]) + fmax(-y[i], 0) * fabs(Ax[i] - qp->l[i]); if (c > comp) comp = c;
  }
  res[1] = viol < 0 ? 0 : viol; res[2] = comp;
  free(g); free(Ax);
}

/* ------------------------------------------------------------------------ */
/* Post-solve                                                                */
/* ------------------------------------------------------------------------ */
int orc_sample(int S, const orc_cube *nc, double delta, const double *x, const double init_s[3],
               const double init_l[3], double *xs, double *dxs, double *ddxs, double *ys, double *dys,
               double *ddys, int cap, int *npoints) {
  const int n_poly = N_POLY, traj_order = TRAJ_ORDER;
  int num_of_points = 1; /* solve_3d.h:115 */
  for (int i = 0; i < S; i++) num_of_points += nc[i].t / delta; /* int += double, :1279-1282 */
  *npoints = num_of_points;
  if (num_of_points > cap || num_of_points < 1) return -1;
  double factorial[6]; factorial[0] = 1.0;
  for (int i = 1; i < n_poly; i++) factorial[i] = factorial[i - 1] * i;
  double b_coe[6][3]; memset(b_coe, 0, sizeof(b_coe));
  for (int i = 0; i < n_poly; i++) b_coe[i][0] = factorial[traj_order] / (factorial[i] * factorial[traj_order - i]);
  for (int i = 0; i < n_poly - 1; i++) b_coe[i][1] = factorial[traj_order - 1] / (factorial[i] * factorial[traj_order - 1 - i]);
  for (int i = 0; i < n_poly - 2; i++) b_coe[i][2] = factorial[traj_order - 2] / (factorial[i] * factorial[traj_order - 2 - i]);
  int var_index = 0, sub_shift = 0;
  xs[0] = init_s[0]; dxs[0] = init_s[1]; ddxs[0] = init_s[2];
  ys[0] = init_l[0]; dys[0] = init_l[1]; ddys[0] = init_l[2];
  var_index++;
  for (int k = 0; k < S; ++k) {
    double c[12];
    for (int i = 0; i < n_poly; i++) { c[i] = x[sub_shift + i]; c[i + n_poly] = x[sub_shift + i + S * n_poly]; }
    sub_shift += n_poly;
    int linter = nc[k].t / delta; /* :1351 */
    for (int l = 1; l <= linter; l++) {
      if (var_index >= num_of_points) return -2; /* .at() would throw */
      double X = 0, DX = 0, DDX = 0, Y = 0, DY = 0, DDY = 0;
      double tau = (double)l / linter;
      for (int i = 0; i < n_poly; i++) {
        double b = b_coe[i][0] * pow(tau, i) * pow(1 - tau, traj_order - i);
        X += c[i] * b; Y += c[i + n_poly] * b;
      }
      X = X * nc[k].t; Y = Y * nc[k].t;
      for (int i = 0; i < n_poly - 1; i++) {
        double b = b_coe[i][1] * pow(tau, i) * pow(1 - tau, traj_order - 1 - i);
        DX += traj_order * (c[i + 1] - c[i]) * b;
        DY += traj_order * (c[i + 1 + n_poly] - c[i + n_poly]) * b;
      }
      for (int i = 0; i < n_poly - 2; i++) {
        double b = b_coe[i][2] * pow(tau, i) * pow(1 - tau, traj_order - 2 - i);
        DDX += traj_order * (traj_order - 1) * (c[i + 2] - 2.0 * c[i + 1] + c[i]) * b;
        DDY += traj_order * (traj_order - 1) * (c[i + 2 + n_poly] - 2.0 * c[i + 1 + n_poly] + c[i + n_poly]) * b;
      }
      DDX = DDX / nc[k].t; DDY = DDY / nc[k].t;
      xs[var_index] = X; dxs[var_index] = DX; ddxs[var_index] = DDX;
      ys[var_index] = Y; dys[var_index] = DY; ddys[var_index] = DDY;
      var_index++;
    }
  }
  if (var_index != num_of_points) return -3; /* CHECK_EQ :1407 */
  return 0;
}

double orc_acost(int variant, const orc_params *p, const orc_input *in, int np, const double *s,
                 const double *ds, const double *dds, const double *l, const double *dl,
                 const double *ddl) {
  double dt = in->delta; int N = in->N;
  double s_cost = 0.0, l_cost = 0.0, mmax_a = 0.0;
  for (int i = 0; i < np; ++i) {
    double ddds = (i == 0) ? (dds[np > 1 ? 1 : 0] - dds[0]) / dt : (dds[i] - dds[i - 1]) / dt;
    double xr = in->x_ref[clampi(i, N - 1)];
    if (variant == ORC_TRAPEZOID) { /* trp_wrapper.cpp:221-224 */
      s_cost += p->weight_s_ref * (s[i] - xr) * (s[i] - xr) * dt;
      s_cost += p->weight_ds_ref * ds[i] * ds[i] * dt;
      s_cost += p->s_acc_weight * dds[i] * dds[i] * dt;
      s_cost += p->s_jerk_weight * ddds * ddds * dt;
    } else { /* cub_wrapper.cpp:214-217 */
      s_cost += (s[i] - xr) * (s[i] - xr) * dt;
      s_cost += ds[i] * ds[i] * dt;
      s_cost += dds[i] * dds[i] * dds[i] * dds[i] * dt;
      s_cost += ddds * ddds * ddds * ddds * dt;
    }
    mmax_a = fmax(mmax_a, fabs(dds[i]));
  }
  if (variant == ORC_CUBOID) s_cost += mmax_a * mmax_a * mmax_a * mmax_a; /* :228 */
  mmax_a = 0.0;
  for (int i = 0; i < np; ++i) {
    double dddl = (i == 0) ? (ddl[np > 1 ? 1 : 0] - ddl[0]) / dt : (ddl[i] - ddl[i - 1]) / dt;
    double yr = in->y_ref[clampi(i, N - 1)];
    if (variant == ORC_TRAPEZOID) {
      l_cost += p->weight_l_ref * (l[i] - yr) * (l[i] - yr) * dt;
      l_cost += p->weight_dl_ref * dl[i] * dl[i] * dt;
      l_cost += p->l_acc_weight * ddl[i] * ddl[i] * dt;
      l_cost += p->l_jerk_weight * dddl * dddl * dt;
    } else {
      l_cost += (l[i] - yr) * (l[i] - yr) * dt;
      l_cost += dl[i] * dl[i] * dt;
      l_cost += ddl[i] * ddl[i] * dt;
      l_cost += dddl * dddl * dt;
    }
    mmax_a = fmax(mmax_a, fabs(ddl[i]));
  }
  if (variant == ORC_TRAPEZOID) { /* trp_wrapper.cpp:269 */
    double le = l[clampi(N - 1, np - 1)], yr = in->y_ref[N - 1];
    l_cost += p->weight_end_l * (le - yr) * (le - yr) * dt;
  } else {
    l_cost += mmax_a * mmax_a; /* cub_wrapper.cpp:257 */
  }
  return s_cost + l_cost;
}

double orc_find_traj(int variant, const char *input_path, const char *output_path,
                     const orc_params *p, const orc_settings *settings, int *S_out,
                     double *ctrl_out, orc_cube *corridor_out, orc_info *info_out) {
  const double FAIL = 100000000000.0; /* trp_wrapper.cpp:199 */
  orc_input in;
  if (orc_input_read(input_path, &in) != 0) return FAIL;
  enum { CAP = 4096 };
  orc_cube *all = (orc_cube *)malloc(sizeof(orc_cube) * CAP);
  int *counts = (int *)calloc(in.num_obs + 1, sizeof(int));
  int total = 0; double ret = FAIL;
  orc_cube *nc = (orc_cube *)malloc(sizeof(orc_cube) * CAP);
  orc_qp qp; memset(&qp, 0, sizeof(qp));
  double *x = NULL;
  for (int o = 0; o < in.num_obs; o++) {
    int c = orc_corridor_generation(variant, in.N, in.delta, in.x_bounds + (size_t)o * in.N * 2,
                                    in.y_bounds + (size_t)o * in.N * 2, all + total, CAP - total);
    if (c < 0) goto done;
    counts[o] = c; total += c;
  }
  int S = orc_collision_check(variant, in.N, in.delta, all, counts, in.num_obs, in.x_ref, in.y_ref, nc, CAP);
  if (S_out) *S_out = S;
  if (S < 1 || S > 64) goto done;
  for (int k = 0; k < S; k++) if (!(nc[k].t > 0)) goto done; /* degenerate segment: 1/t^3 */
  if (corridor_out) memcpy(corridor_out, nc, sizeof(orc_cube) * S);
  orc_qp_params pp; memset(&pp, 0, sizeof(pp));
  pp.w_s[0] = p->weight_s_ref; pp.w_s[1] = p->weight_ds_ref; pp.w_s[2] = p->s_acc_weight; pp.w_s[3] = p->s_jerk_weight;
  pp.w_l[0] = p->weight_l_ref; pp.w_l[1] = p->weight_dl_ref; pp.w_l[2] = p->l_acc_weight; pp.w_l[3] = p->l_jerk_weight;
  pp.weight_end_s = p->weight_end_s; pp.weight_end_l = p->weight_end_l;
  pp.ds_ref = in.ds_ref; pp.dl_ref = in.dl_ref;
  memcpy(pp.dds, in.dds, sizeof(pp.dds)); memcpy(pp.ddds, in.ddds, sizeof(pp.ddds));
  memcpy(pp.ddl, in.ddl, sizeof(pp.ddl)); memcpy(pp.dddl, in.dddl, sizeof(pp.dddl));
  memcpy(pp.init_s, in.init_s, sizeof(pp.init_s)); memcpy(pp.init_l, in.init_l, sizeof(pp.init_l));
  pp.N = in.N; pp.delta = in.delta; pp.dx_bounds = in.dx_bounds; pp.dy_bounds = in.dy_bounds;
  pp.x_ref = in.x_ref; pp.y_ref = in.y_ref;
  if (orc_assemble(variant, S, nc, &pp, &qp) != 0) goto done;
  x = (double *)calloc(qp.n, sizeof(double));
  orc_settings sdef; if (!settings) { orc_settings_reference(&sdef); settings = &sdef; }
  orc_info info; memset(&info, 0, sizeof(info));
  orc_osqp_solve(&qp, settings, x, NULL, &info);
  if (info_out) *info_out = info;
  if (ctrl_out) memcpy(ctrl_out, x, sizeof(double) * qp.n);
  /* acceptance solve_3d.cc:1251-1277 (cuboid has no NaN test, cuboid_3d.cc:1110-1128) */
  if (info.status != 1 && info.status != 2) goto done;
  if (variant == ORC_TRAPEZOID && info.obj_val != info.obj_val) goto done;
  {
    int cap = 16 * in.N + 64, np = 0;
    double *buf = (double *)calloc((size_t)cap * 6, sizeof(double));
    double *s = buf, *ds = buf + cap, *dds = buf + 2 * cap, *l = buf + 3 * cap, *dl = buf + 4 * cap, *ddl = buf + 5 * cap;
    if (orc_sample(S, nc, in.delta, x, in.init_s, in.init_l, s, ds, dds, l, dl, ddl, cap, &np) == 0) {
      ret = orc_acost(variant, p, &in, np, s, ds, dds, l, dl, ddl);
      if (output_path) {
        FILE *f = fopen(output_path, "w");
        if (f) { /* trp_wrapper.cpp:298-301: fixed, precision 3 */
          for (int i = 0; i < np; i++)
            fprintf(f, "%.3f %.3f %.3f %.3f %.3f %.3f %.3f\n", i * in.delta, s[i], l[i], ds[i], dl[i], dds[i], ddl[i]);
          fclose(f);
        }
      }
    }
    free(buf);
  }
done:
  free(x); orc_qp_free(&qp); free(all); free(counts); free(nc); orc_input_free(&in);
  return ret;
}

/* ------------------------------------------------------------------------ */
/* High-accuracy optimum x*: dense Mehrotra predictor-corrector interior      */
/* point method on the general (P,q,A,l,u).  Independent of the ADMM above    */
/* and of the product's structured solver; used by tests as the parity        */
/* reference (P > 0 here, so x* is unique).  Rows with u-l <= eq_tol are       */
/* equalities; l > u + eq_tol is reported primal infeasible (-3).             */
/* ------------------------------------------------------------------------ */
static int lu_factor(double *M, int *piv, int N) {
  for (int c = 0; c < N; c++) {
    int p = c; double best = fabs(M[(size_t)c * N + c]);
    for (int r = c + 1; r < N; r++) { double a = fabs(M[(size_t)r * N + c]); if (a > best) { best = a; p = r; } }
    piv[c] = p;
    if (best == 0) return -1;
    if (p != c) for (int k = 0; k < N; k++) { double t = M[(size_t)c * N + k]; M[(size_t)c * N + k] = M[(size_t)p * N + k]; M[(size_t)p * N + k] = t; }
    double inv = 1.0 / M[(size_t)c * N + c];
    for (int r = c + 1; r < N; r++) {
      double f = M[(size_t)r * N + c] * inv; M[(size_t)r * N + c] = f;
      if (f != 0) for (int k = c + 1; k < N; k++) M[(size_t)r * N + k] -= f * M[(size_t)c * N + k];
    }
  }
  return 0;
}
static void lu_solve(const double *M, const int *piv, double *b, int N) {
  for (int c = 0; c < N; c++) { int p = piv[c]; if (p != c) { double t = b[c]; b[c] = b[p]; b[p] = t; } }
  for (int r = 0; r < N; r++) { double s = b[r]; for (int k = 0; k < r; k++) s -= M[(size_t)r * N + k] * b[k]; b[r] = s; }
  for (int r = N - 1; r >= 0; r--) { double s = b[r]; for (int k = r + 1; k < N; k++) s -= M[(size_t)r * N + k] * b[k]; b[r] = s / M[(size_t)r * N + r]; }
}

/* delta > 0: every inequality row is elastic, l <= a'x - d <= u with d^2 / (2 delta) added to the objective (the
 * rescue pass of the product, include/btrapz_hip.h btrapz_options.elastic).  Eliminating d = delta (lambda_u - lambda_l)
 * changes three things per row: its value a'x - delta y, its weight W / (1 + delta W) in the Newton matrix, and the step
 * of its value (a'dx - delta t) / (1 + delta W), t the row's entry of the right-hand side.  delta = 0: the plain method. */
static int ipm_core(const orc_qp *qp, double delta, double eps, int max_iter, double *x_out, double *y_out, orc_info *info) {
  const int n = qp->n, m = qp->m;
  const double eq_tol = 1e-12;
  int *iseq = (int *)malloc(sizeof(int) * m), *rowpos = (int *)malloc(sizeof(int) * m);
  int me = 0, mi = 0, status = -2;
  for (int i = 0; i < m; i++) {
    double g = qp->u[i] - qp->l[i];
    if (g < -eq_tol) status = -3;
    iseq[i] = g <= eq_tol; rowpos[i] = iseq[i] ? me++ : mi++;
  }
  int iter = 0; double obj = 0, best_score = 1e300;
  if (status == -3) { for (int i = 0; i < n; i++) x_out[i] = 0; if (y_out) for (int i = 0; i < m; i++) y_out[i] = 0; goto fin; }
  {
  /* dense A (row major) and P */
  double *Ad = (double *)calloc((size_t)m * n, sizeof(double)), *Pd = (double *)calloc((size_t)n * n, sizeof(double));
  for (int j = 0; j < n; j++) for (orc_int p = qp->A_p[j]; p < qp->A_p[j + 1]; p++) Ad[(size_t)qp->A_i[p] * n + j] = qp->A_x[p];
  for (int j = 0; j < n; j++) for (orc_int p = qp->P_p[j]; p < qp->P_p[j + 1]; p++) { int i = (int)qp->P_i[p]; Pd[(size_t)i * n + j] = qp->P_x[p]; Pd[(size_t)j * n + i] = qp->P_x[p]; }
  int Nk = n + me;
  double *K = (double *)malloc(sizeof(double) * (size_t)Nk * Nk); int *piv = (int *)malloc(sizeof(int) * Nk);
  double *K0 = (double *)malloc(sizeof(double) * (size_t)Nk * Nk), *rhs0 = (double *)malloc(sizeof(double) * Nk), *rres = (double *)malloc(sizeof(double) * Nk), *dsc = (double *)malloc(sizeof(double) * Nk);
  double *x = (double *)calloc(n, sizeof(double)), *nu = (double *)calloc(me + 1, sizeof(double));
  double *sl = (double *)malloc(sizeof(double) * (mi + 1)), *su = (double *)malloc(sizeof(double) * (mi + 1));
  double *ll = (double *)malloc(sizeof(double) * (mi + 1)), *lu_ = (double *)malloc(sizeof(double) * (mi + 1));
  double *Ax = (double *)malloc(sizeof(double) * m), *rd = (double *)malloc(sizeof(double) * n);
  double *rpl = (double *)malloc(sizeof(double) * (mi + 1)), *rpu = (double *)malloc(sizeof(double) * (mi + 1)), *re = (double *)malloc(sizeof(double) * (me + 1));
  double *W = (double *)malloc(sizeof(double) * (mi + 1)), *rhs = (double *)malloc(sizeof(double) * Nk);
  double *dsl = (double *)malloc(sizeof(double) * (mi + 1)), *dsu = (double *)malloc(sizeof(double) * (mi + 1));
  double *dll = (double *)malloc(sizeof(double) * (mi + 1)), *dlu = (double *)malloc(sizeof(double) * (mi + 1));
  double *rcl = (double *)malloc(sizeof(double) * (mi + 1)), *rcu = (double *)malloc(sizeof(double) * (mi + 1));
  double *Adx = (double *)malloc(sizeof(double) * m), *tt = (double *)malloc(sizeof(double) * (mi + 1));
  double *bx = (double *)malloc(sizeof(double) * n), *by = (double *)calloc(m, sizeof(double));
  double qn = vnorm_inf(qp->q, n), bn = 0;
  /* Bounds that are no bounds (the reference leaves rows it does not use at +-1e10, src/trp.cc dl_bounds and the header's
   * limits; OSQP treats them as finite rows that never become active): they do not measure the problem -- a primal residual
   * relative to 1e10 would accept anything -- and a multiplier of 1 on a slack of 1e10 starts the method at mu = 1e9, where
   * every step length is 1e-6 (round 5: candidate 26 of tests/test_gpu_forms.py's far-bounds batch never left the start);
   * and a slack of 1e10 carries 1e-6 of absolute round-off into its row's residual (seen as 6e-6 in x* on the sweep's
   * +-1e10 cases).  Such a side takes no part: no slack, multiplier 0, out of mu and of the residual norms. */
  int *farl = (int *)calloc(mi + 1, sizeof(int)), *faru = (int *)calloc(mi + 1, sizeof(int)), nside = 0;
  for (int i = 0; i < m; i++) { if (fabs(qp->l[i]) < 1e9) bn = fmax(bn, fabs(qp->l[i])); if (fabs(qp->u[i]) < 1e9) bn = fmax(bn, fabs(qp->u[i])); }
  for (int i = 0; i < m; i++) if (!iseq[i]) {
    int r = rowpos[i]; sl[r] = fmax(0.0 - qp->l[i], 1.0); su[r] = fmax(qp->u[i] - 0.0, 1.0);
    farl[r] = isfinite(qp->l[i]) && qp->l[i] <= -1e9; faru[r] = isfinite(qp->u[i]) && qp->u[i] >= 1e9;   /* (an infinite bound is not a far one: such a corridor has no answer here, as in the product) */
    ll[r] = farl[r] ? 0.0 : 1.0; lu_[r] = faru[r] ? 0.0 : 1.0;
    nside += !farl[r] + !faru[r];
  }
  int best_it = 0, safe = 0;
  double score_hist[4] = {1e300, 1e300, 1e300, 1e300};
  for (iter = 0; iter < max_iter; iter++) {
    for (int i = 0; i < m; i++) { double s = 0; const double *a = Ad + (size_t)i * n; for (int j = 0; j < n; j++) s += a[j] * x[j]; Ax[i] = s; }
    for (int j = 0; j < n; j++) { double s = qp->q[j]; const double *pr = Pd + (size_t)j * n; for (int k = 0; k < n; k++) s += pr[k] * x[k]; rd[j] = s; }
    for (int i = 0; i < m; i++) {
      double yi = iseq[i] ? nu[rowpos[i]] : (lu_[rowpos[i]] - ll[rowpos[i]]);
      const double *a = Ad + (size_t)i * n; if (yi != 0) for (int j = 0; j < n; j++) rd[j] += a[j] * yi;
    }
    double mu = 0, rpn = 0;
    for (int i = 0; i < m; i++) {
      int r = rowpos[i];
      if (iseq[i]) { re[r] = Ax[i] - qp->l[i]; rpn = fmax(rpn, fabs(re[r])); }
      else {
        const double v = Ax[i] - delta * (lu_[r] - ll[r]);
        rpl[r] = farl[r] ? 0.0 : v - sl[r] - qp->l[i]; rpu[r] = faru[r] ? 0.0 : v + su[r] - qp->u[i];
        mu += sl[r] * ll[r] + su[r] * lu_[r]; rpn = fmax(rpn, fmax(fabs(rpl[r]), fabs(rpu[r])));
      }
    }
    mu = nside ? mu / nside : 0;
    double score = fmax(fmax(vnorm_inf(rd, n) / (1 + qn), rpn / (1 + bn)), mu);
    {  /* fmax() skips NaN: an iterate or a residual that is not finite must not pass for a small score */
      int finite = isfinite(mu) && isfinite(bn) && isfinite(qn);
      for (int j = 0; j < n && finite; j++) finite = isfinite(rd[j]) && isfinite(x[j]);
      for (int i = 0; i < m && finite; i++) finite = isfinite(Ax[i]);
      for (int r = 0; r < mi && finite; r++) finite = isfinite(rpl[r]) && isfinite(rpu[r]);
      if (!finite) score = 1e300;
    }
    if (getenv("ORC_TRACE")) fprintf(stderr, "orc trace iter %d score %.3e dual %.3e primal %.3e mu %.3e\n", iter, score, vnorm_inf(rd, n) / (1 + qn), rpn / (1 + bn), mu);
    if (iter == 0 || score < best_score) {
      best_score = score; best_it = iter; memcpy(bx, x, sizeof(double) * n);
      for (int i = 0; i < m; i++) by[i] = iseq[i] ? nu[rowpos[i]] : (lu_[rowpos[i]] - ll[rowpos[i]]);
    }
    if (score < eps || (best_score < 1e-5 && iter - best_it >= 3) || !(score < 1e299)) break;
    /* K = [P + G'WG, E'; E, 0] */
    memset(K, 0, sizeof(double) * (size_t)Nk * Nk);
    for (int i = 0; i < n; i++) memcpy(K + (size_t)i * Nk, Pd + (size_t)i * n, sizeof(double) * n);
    for (int i = 0; i < m; i++) {
      const double *a = Ad + (size_t)i * n; int r = rowpos[i];
      if (iseq[i]) { for (int j = 0; j < n; j++) if (a[j] != 0) { K[(size_t)(n + r) * Nk + j] = a[j]; K[(size_t)j * Nk + (n + r)] = a[j]; } }
      else {
        W[r] = ll[r] / sl[r] + lu_[r] / su[r];
        const double We = W[r] / (1.0 + delta * W[r]);
        for (int j = 0; j < n; j++) if (a[j] != 0) { double f = We * a[j]; for (int k = 0; k < n; k++) if (a[k] != 0) K[(size_t)j * Nk + k] += f * a[k]; }
      }
    }
    /* Symmetric equilibration D K D before the factorisation: unit diagonal in the Hessian block, unit largest entry in
     * every equality row.  Partial pivoting compares magnitudes across rows, and rows weighted by lambda / s = 1e15 next
     * to equality rows of size one make it pick badly (same finding as the refinement below). */
    for (int i = 0; i < n; i++) { const double d = K[(size_t)i * Nk + i]; dsc[i] = d > 0 ? 1.0 / sqrt(d) : 1.0; }
    for (int r = 0; r < me; r++) {
      double mx = 0; const double *kr = K + (size_t)(n + r) * Nk;
      for (int j = 0; j < n; j++) mx = fmax(mx, fabs(kr[j]) * dsc[j]);
      dsc[n + r] = mx > 0 ? 1.0 / mx : 1.0;
    }
    for (int i = 0; i < Nk; i++) { double *kr = K + (size_t)i * Nk; const double di = dsc[i]; for (int j = 0; j < Nk; j++) kr[j] *= di * dsc[j]; }
    memcpy(K0, K, sizeof(double) * (size_t)Nk * Nk);   /* (for the refinement of the solves below) */
    if (lu_factor(K, piv, Nk) != 0) break;
    /* Safeguard (round 5, found by the sweep over the reference's weight space): unsafeguarded Mehrotra steps can cycle --
     * mu going round 2e-3, 1e-3, 2e-3, 6e-4 for ever with residuals at 1e-13 (seen on 4 of ~57 000 candidates under
     * weight rows with a near-zero jerk weight or zero end weights; a cycle may still creep down in its fourth digit).
     * Four iterations with less than a halving of the score switch the solve, for three iterations, to plain centred
     * path-following: no second-order term, sigma at least 0.2. */
    /* (only where complementarity is all that is left -- residuals four digits below mu: a slow START, residuals and mu
     *  falling together by 10 % per iteration as on src/c4.txt, is not a cycle and Mehrotra's steps get it going) */
    if (fmax(vnorm_inf(rd, n) / (1 + qn), rpn / (1 + bn)) < 1e-4 * mu &&
        iter >= 8 && score > 0.5 * score_hist[iter & 3]) {   /* (less than a halving in four iterations) */
      safe = 3;   /* three centred steps, then Mehrotra's again from a better-centred iterate (crawling all the way down at
                   * sigma = 0.2 spends many iterations at lambda / s = 1e11 and the multiplier update's round-off, amplified
                   * by that factor each time, ruins the dual residual: src fuzz seed 31, calls 33 and 93) */
      for (int h = 0; h < 4; h++) score_hist[h] = 1e300;
    } else if (safe > 0) --safe;
    score_hist[iter & 3] = score;
    double sigma = 0, alpha = 1;
    for (int pass = 0; pass < 2; pass++) {
      for (int r = 0; r < mi; r++) {
        if (pass == 0) { rcl[r] = sl[r] * ll[r]; rcu[r] = su[r] * lu_[r]; }
        else if (safe) { rcl[r] = sl[r] * ll[r] - sigma * mu; rcu[r] = su[r] * lu_[r] - sigma * mu; }
        else { rcl[r] = sl[r] * ll[r] - sigma * mu + dsl[r] * dll[r]; rcu[r] = su[r] * lu_[r] - sigma * mu + dsu[r] * dlu[r]; }
        if (farl[r]) rcl[r] = 0.0;
        if (faru[r]) rcu[r] = 0.0;
      }
      for (int j = 0; j < n; j++) rhs[j] = -rd[j];
      for (int i = 0; i < m; i++) {
        int r = rowpos[i];
        if (iseq[i]) rhs[n + r] = -re[r];
        else {
          double t = rcl[r] / sl[r] - rcu[r] / su[r] + (ll[r] / sl[r]) * rpl[r] + (lu_[r] / su[r]) * rpu[r];
          tt[r] = t; t /= 1.0 + delta * W[r];
          const double *a = Ad + (size_t)i * n; for (int j = 0; j < n; j++) if (a[j] != 0) rhs[j] -= a[j] * t;
        }
      }
      /* Solve with two steps of iterative refinement, residuals in long double: near the end the weights lambda / s span
       * twenty orders of magnitude and a plain LU solve of the indefinite system loses the dual residual altogether (found
       * by the round-5 sweep over the reference's weight space: src/c2.txt, cuboid, trial rows 5 and 36 of all_weights.txt
       * -- the dual residual went from 4e-8 to 1e+11 within five iterations at mu = 1e-5). */
      for (int i = 0; i < Nk; i++) rhs[i] *= dsc[i];
      memcpy(rhs0, rhs, sizeof(double) * Nk);
      lu_solve(K, piv, rhs, Nk);
      for (int refine = 0; refine < 2; refine++) {
        for (int i = 0; i < Nk; i++) {
          long double acc = rhs0[i]; const double *kr = K0 + (size_t)i * Nk;
          for (int j = 0; j < Nk; j++) acc -= (long double)kr[j] * rhs[j];
          rres[i] = (double)acc;
        }
        lu_solve(K, piv, rres, Nk);
        for (int i = 0; i < Nk; i++) rhs[i] += rres[i];
      }
      for (int i = 0; i < Nk; i++) rhs[i] *= dsc[i];
      for (int i = 0; i < m; i++) { double s = 0; const double *a = Ad + (size_t)i * n; for (int j = 0; j < n; j++) s += a[j] * rhs[j]; Adx[i] = s; }
      double ap = 1, ad = 1;
      for (int i = 0; i < m; i++) if (!iseq[i]) {
        int r = rowpos[i];
        const double g = (Adx[i] - delta * tt[r]) / (1.0 + delta * W[r]);
        dsl[r] = farl[r] ? 0.0 : g + rpl[r]; dsu[r] = faru[r] ? 0.0 : -g - rpu[r];
        dll[r] = (-rcl[r] - ll[r] * dsl[r]) / sl[r]; dlu[r] = (-rcu[r] - lu_[r] * dsu[r]) / su[r];
        if (dsl[r] < 0) ap = fmin(ap, -sl[r] / dsl[r]);
        if (dsu[r] < 0) ap = fmin(ap, -su[r] / dsu[r]);
        if (dll[r] < 0) ad = fmin(ad, -ll[r] / dll[r]);
        if (dlu[r] < 0) ad = fmin(ad, -lu_[r] / dlu[r]);
      }
      if (pass == 0) {
        double mua = 0;
        for (int r = 0; r < mi; r++) mua += (sl[r] + ap * dsl[r]) * (ll[r] + ad * dll[r]) + (su[r] + ap * dsu[r]) * (lu_[r] + ad * dlu[r]);
        mua = nside ? mua / nside : 0;
        sigma = mu > 0 ? pow(mua / mu, 3) : 0;
        if (safe) sigma = fmax(sigma, 0.2);
      } else {
        alpha = fmin(1.0, 0.995 * fmin(ap, ad));
      }
    }
    if (getenv("ORC_TRACE")) fprintf(stderr, "orc trace        step: safe %d sigma %.3e alpha %.6f\n", safe, sigma, alpha);
    for (int j = 0; j < n; j++) x[j] += alpha * rhs[j];
    for (int r = 0; r < me; r++) nu[r] += alpha * rhs[n + r];
    for (int r = 0; r < mi; r++) { sl[r] += alpha * dsl[r]; su[r] += alpha * dsu[r]; ll[r] += alpha * dll[r]; lu_[r] += alpha * dlu[r]; }
  }
  memcpy(x_out, bx, sizeof(double) * n);
  if (y_out) memcpy(y_out, by, sizeof(double) * m);
  status = best_score < 1e-7 ? 1 : (best_score < 1e-5 ? 2 : -2);
  for (int j = 0; j < n; j++) { double s = 0; const double *pr = Pd + (size_t)j * n; for (int k = 0; k < n; k++) s += pr[k] * bx[k]; obj += 0.5 * bx[j] * s + qp->q[j] * bx[j]; }
  free(K0); free(rhs0); free(rres); free(dsc); free(farl); free(faru);
  free(Ad); free(Pd); free(K); free(piv); free(x); free(nu); free(sl); free(su); free(ll); free(lu_); free(Ax); free(rd);
  free(rpl); free(rpu); free(re); free(W); free(rhs); free(dsl); free(dsu); free(dll); free(dlu); free(rcl); free(rcu); free(Adx); free(tt); free(bx); free(by);
  }
fin:
  if (info) { memset(info, 0, sizeof(*info)); info->status = status; info->iter = iter; info->obj_val = obj; info->pri_res = best_score; }
  free(iseq); free(rowpos);
  return 0;
}

int orc_ipm_solve(const orc_qp *qp, double eps, int max_iter, double *x_out, double *y_out, orc_info *info) {
  return ipm_core(qp, 0.0, eps, max_iter, x_out, y_out, info);
}
int orc_elastic_solve(const orc_qp *qp, double delta, double eps, int max_iter, double *x_out, double *y_out, orc_info *info) {
  return ipm_core(qp, delta, eps, max_iter, x_out, y_out, info);
}

/* ------------------------------------------------------------------------ */
/* Batch-record entry point: the same candidates the product's batched C-ABI  */
/* takes (layout of include/btrapz_hip.h: seg[f][b][k], init[b][6],           */
/* ref_end[b][2], dl_bounds[b][10], shared[21]) solved one by one by the       */
/* reference's algorithm: orc_assemble (general CSC) + orc_osqp_solve.  Used   */
/* as the CPU baseline of bench.py and by batch parity tests.  shared[] =      */
/* w_s[4] w_l[4] weight_end_s weight_end_l ds_ref dl_ref dds[2] ddds[2] ddl[2]  */
/* dddl[2] delta.  exact != 0 -> orc_ipm_solve (x*) instead of the ADMM.       */
/* ------------------------------------------------------------------------ */
int orc_batch_solve(int variant, int B, int S, const double *seg, const double *init, const double *ref_end,
                    const double *dl_bounds, const double *shared, const orc_settings *settings, int exact,
                    int b0, int b1, double *ctrl, double *obj, int *status, int *iters) {
  enum { F_T = 0, F_DOWN_BIAS, F_DOWN_SKEW, F_UPP_BIAS, F_UPP_SKEW, F_L_DOWN_BIAS, F_L_DOWN_SKEW, F_L_UPP_BIAS,
         F_L_UPP_SKEW, F_BEG_L, F_END_L, F_DS_LO, F_DS_HI, F_X_SKEW, F_X_BIAS, F_Y_SKEW, F_Y_BIAS };
  if (S < 1 || S > 64 || b0 < 0 || b1 > B) return -1;
  const int N = 10 * S + 1;
  const size_t BS = (size_t)B * S;
  const double delta = shared[20];
  double *x_ref = (double *)calloc(N, sizeof(double)), *y_ref = (double *)calloc(N, sizeof(double));
  double *dxb = (double *)malloc(sizeof(double) * 2 * N), *dyb = (double *)malloc(sizeof(double) * 2 * N);
  orc_cube *cubes = (orc_cube *)malloc(sizeof(orc_cube) * S);
  double *x = (double *)malloc(sizeof(double) * 12 * S);
  orc_settings sdef; if (!settings) { orc_settings_reference(&sdef); settings = &sdef; }
  for (int b = b0; b < b1; b++) {
#define SEG(f, k) seg[(size_t)(f) * BS + (size_t)b * S + (k)]
    for (int i = 0; i < N; i++) { x_ref[i] = 0; y_ref[i] = 0; dxb[2 * i] = -1e10; dxb[2 * i + 1] = 1e10; dyb[2 * i] = -1e10; dyb[2 * i + 1] = 1e10; }
    for (int k = 0; k < S; k++) {
      orc_cube *c = &cubes[k]; cube_default(c);
      c->beg_t = 10 * k; c->end_t = 10 * k + 10; c->t = SEG(F_T, k);
      c->beg_l = SEG(F_BEG_L, k); c->end_l = SEG(F_END_L, k);
      c->upp_skew = SEG(F_UPP_SKEW, k); c->upp_bias = SEG(F_UPP_BIAS, k);
      c->down_skew = SEG(F_DOWN_SKEW, k); c->down_bias = SEG(F_DOWN_BIAS, k);
      c->l_upp_skew = SEG(F_L_UPP_SKEW, k); c->l_upp_bias = SEG(F_L_UPP_BIAS, k);
      c->l_down_skew = SEG(F_L_DOWN_SKEW, k); c->l_down_bias = SEG(F_L_DOWN_BIAS, k);
      x_ref[10 * k] = SEG(F_X_BIAS, k); x_ref[10 * k + 1] = SEG(F_X_BIAS, k) + SEG(F_X_SKEW, k) * delta;
      y_ref[10 * k] = SEG(F_Y_BIAS, k); y_ref[10 * k + 1] = SEG(F_Y_BIAS, k) + SEG(F_Y_SKEW, k) * delta;
      for (int i = 10 * k + 1; i < 10 * k + 10; i++) { dxb[2 * i] = SEG(F_DS_LO, k); dxb[2 * i + 1] = SEG(F_DS_HI, k); }
    }
    x_ref[N - 1] = ref_end[2 * b]; y_ref[N - 1] = ref_end[2 * b + 1];
    for (int i = 0; i < 5 && i < N; i++) { dyb[2 * i] = dl_bounds[10 * b + 2 * i]; dyb[2 * i + 1] = dl_bounds[10 * b + 2 * i + 1]; }
    orc_qp_params pp; memset(&pp, 0, sizeof(pp));
    memcpy(pp.w_s, shared, sizeof(double) * 4); memcpy(pp.w_l, shared + 4, sizeof(double) * 4);
    pp.weight_end_s = shared[8]; pp.weight_end_l = shared[9]; pp.ds_ref = shared[10]; pp.dl_ref = shared[11];
    memcpy(pp.dds, shared + 12, 16); memcpy(pp.ddds, shared + 14, 16); memcpy(pp.ddl, shared + 16, 16); memcpy(pp.dddl, shared + 18, 16);
    memcpy(pp.init_s, init + 6 * b, 24); memcpy(pp.init_l, init + 6 * b + 3, 24);
    pp.N = N; pp.delta = delta; pp.dx_bounds = dxb; pp.dy_bounds = dyb; pp.x_ref = x_ref; pp.y_ref = y_ref;
    orc_qp qp; orc_info info; memset(&info, 0, sizeof(info));
    if (orc_assemble(variant, S, cubes, &pp, &qp) != 0) { status[b] = -10; obj[b] = 0; if (iters) iters[b] = 0; orc_qp_free(&qp); continue; }
    if (exact) orc_ipm_solve(&qp, 1e-9, 80, x, NULL, &info); else orc_osqp_solve(&qp, settings, x, NULL, &info);
    memcpy(ctrl + (size_t)b * 12 * S, x, sizeof(double) * 12 * S);
    obj[b] = info.obj_val; status[b] = info.status; if (iters) iters[b] = info.iter;
    orc_qp_free(&qp);
#undef SEG
  }
  free(x_ref); free(y_ref); free(dxb); free(dyb); free(cubes); free(x);
  return 0;
}

/* ------------------------------------------------------------------------ */
/* orc_batch_solve over [b0,b1) on `threads` POSIX threads.  Candidates are     */
/* independent; their cost is not uniform (ADMM takes 200 .. 5000 iterations), */
/* so the threads draw candidates from a shared counter instead of owning a    */
/* fixed slice; every thread has its own buffers and output rows are disjoint. */
/* bench.py's cpu_baseline leg.                                               */
/* ------------------------------------------------------------------------ */
#include <pthread.h>
#include <stdatomic.h>
typedef struct {
  int variant, B, S, exact, b1, rc;
  atomic_int *next;
  const double *seg, *init, *ref_end, *dl_bounds, *shared; const orc_settings *settings;
  double *ctrl, *obj; int *status, *iters;
} orc_mt_job;
static void *orc_mt_run(void *p) {
  orc_mt_job *j = (orc_mt_job *)p;
  for (;;) {
    const int b = atomic_fetch_add(j->next, 1);
    if (b >= j->b1) break;
    const int rc = orc_batch_solve(j->variant, j->B, j->S, j->seg, j->init, j->ref_end, j->dl_bounds, j->shared,
                                   j->settings, j->exact, b, b + 1, j->ctrl, j->obj, j->status, j->iters);
    if (rc != 0) j->rc = rc;
  }
  return NULL;
}
int orc_batch_solve_mt(int variant, int B, int S, const double *seg, const double *init, const double *ref_end,
                       const double *dl_bounds, const double *shared, const orc_settings *settings, int exact,
                       int b0, int b1, int threads, double *ctrl, double *obj, int *status, int *iters) {
  if (S < 1 || S > 64 || b0 < 0 || b1 > B) return -1;
  if (threads < 1) threads = 1;
  if (threads > b1 - b0) threads = b1 - b0 > 0 ? b1 - b0 : 1;
  if (threads == 1) return orc_batch_solve(variant, B, S, seg, init, ref_end, dl_bounds, shared, settings, exact, b0, b1, ctrl, obj, status, iters);
  orc_mt_job *jobs = (orc_mt_job *)calloc(threads, sizeof(orc_mt_job));
  pthread_t *tid = (pthread_t *)calloc(threads, sizeof(pthread_t));
  atomic_int next; atomic_init(&next, b0);
  int rc = 0;
  for (int t = 0; t < threads; t++) {
    orc_mt_job *j = &jobs[t];
    j->variant = variant; j->B = B; j->S = S; j->exact = exact; j->seg = seg; j->init = init; j->ref_end = ref_end;
    j->dl_bounds = dl_bounds; j->shared = shared; j->settings = settings; j->ctrl = ctrl; j->obj = obj; j->status = status; j->iters = iters;
    j->b1 = b1; j->next = &next;
    if (t + 1 < threads) { if (pthread_create(&tid[t], NULL, orc_mt_run, j) != 0) tid[t] = 0; }
    else orc_mt_run(j);                                   /* the calling thread works too */
  }
  for (int t = 0; t + 1 < threads; t++) if (tid[t]) pthread_join(tid[t], NULL);
  for (int t = 0; t < threads; t++) if (jobs[t].rc != 0) rc = jobs[t].rc;
  free(jobs); free(tid);
  return rc;
}
