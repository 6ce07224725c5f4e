/* multi_host_demo.c -- a host program in plain C over the C-ABI of include/btrapz_hip.h, no Python, no torch: candidate
 * corridors sharded over the devices given on the command line, one step (solve + arg-min + the one gather), the winner
 * printed.  What a C++ planner that links libbtrapz_hip.so does for a candidate set (the reference solves one corridor
 * per call inside src/cart_frenet.py:1516-1571).  tests/test_gpu_multi.py builds and runs it.
 *
 *   multi_host_demo B S transport device [device ...]     transport: 0 auto, 1 copies, 2 RCCL; a device may repeat
 *
 * Output (one line): winner <index> cost <%.17g> transport <1|2> ctrl0 <%.17g> ctrl_last <%.17g>                          */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "btrapz_hip.h"

static double urand(unsigned long long *s) { /* xorshift64*: the same candidates on every run */
  *s ^= *s >> 12; *s ^= *s << 25; *s ^= *s >> 27;
  return (double)((*s * 2685821657736338717ULL) >> 11) / 9007199254740992.0;
}

int main(int argc, char **argv) {
  if (argc < 5) { fprintf(stderr, "usage: %s B S transport device [device ...]\n", argv[0]); return 2; }
  const int B = atoi(argv[1]), S = atoi(argv[2]), transport = atoi(argv[3]), G = argc - 4;
  int devices[64];
  if (B < 1 || S < 1 || S > BTRAPZ_MAX_SEGMENTS || G > 64) return 2;
  for (int g = 0; g < G; g++) devices[g] = atoi(argv[4 + g]);

  /* B candidates: an ego at v0 in a straight lane, an obstacle ramp ahead of it in some segments (layout of
   * btrapz_solve_batch_device: seg[f][b][k], init[b][6], ref_end[b][2], dl_bounds[b][10]) */
  const size_t n = (size_t)B * S;
  double *seg = calloc((size_t)BTRAPZ_NUM_SEG_FIELDS * n, sizeof(double)), *init = calloc((size_t)B * 6, sizeof(double));
  double *ref_end = calloc((size_t)B * 2, sizeof(double)), *dlb = calloc((size_t)B * 10, sizeof(double));
  if (!seg || !init || !ref_end || !dlb) return 3;
  unsigned long long rs = 0x9E3779B97F4A7C15ULL;
  for (int b = 0; b < B; b++) {
    const double v0 = 4.0 + 5.0 * urand(&rs), l0 = -0.5 + urand(&rs), margin = 6.0 + 6.0 * urand(&rs);
    for (int k = 0; k < S; k++) {
      const size_t e = (size_t)b * S + k;
      const double s0 = v0 * k;
      seg[BTRAPZ_F_T * n + e] = 1.0;
      seg[BTRAPZ_F_DOWN_BIAS * n + e] = s0 - margin; seg[BTRAPZ_F_DOWN_SKEW * n + e] = v0;
      seg[BTRAPZ_F_UPP_BIAS * n + e] = s0 + margin;  seg[BTRAPZ_F_UPP_SKEW * n + e] = (urand(&rs) < 0.3) ? 0.5 * v0 : v0;
      seg[BTRAPZ_F_L_DOWN_BIAS * n + e] = -2.0; seg[BTRAPZ_F_L_UPP_BIAS * n + e] = 2.0;
      seg[BTRAPZ_F_BEG_L * n + e] = -2.0; seg[BTRAPZ_F_END_L * n + e] = 2.0;
      seg[BTRAPZ_F_DS_LO * n + e] = 0.0; seg[BTRAPZ_F_DS_HI * n + e] = 30.0;
      seg[BTRAPZ_F_X_SKEW * n + e] = v0; seg[BTRAPZ_F_X_BIAS * n + e] = s0;
      seg[BTRAPZ_F_Y_SKEW * n + e] = 0.0; seg[BTRAPZ_F_Y_BIAS * n + e] = 0.0;
    }
    init[(size_t)b * 6 + 1] = v0; init[(size_t)b * 6 + 3] = l0;
    ref_end[(size_t)b * 2] = v0 * S; ref_end[(size_t)b * 2 + 1] = 0.0;
    for (int i = 0; i < 5; i++) { dlb[(size_t)b * 10 + 2 * i] = -2.0; dlb[(size_t)b * 10 + 2 * i + 1] = 2.0; }
  }
  btrapz_shared sh;
  memset(&sh, 0, sizeof sh);
  /* src/weights.txt in Params order: s_acc s_jerk l_acc l_jerk s_ref ds_ref l_ref dl_ref end_s end_l */
  sh.w_s[0] = 0.12; sh.w_s[1] = 10.04; sh.w_s[2] = 35.73; sh.w_s[3] = 41.61;
  sh.w_l[0] = 0.71; sh.w_l[1] = 14.3;  sh.w_l[2] = 25.57; sh.w_l[3] = 41.59;
  sh.weight_end_s = 7.27; sh.weight_end_l = 32.13;
  sh.ds_ref = 7.0; sh.dl_ref = 0.0;
  sh.dds[0] = -2.0; sh.dds[1] = 2.0; sh.ddds[0] = -30.0; sh.ddds[1] = 30.0;
  sh.ddl[0] = -0.7; sh.ddl[1] = 0.7; sh.dddl[0] = -10.0; sh.dddl[1] = 10.0;
  sh.delta = 0.1; sh.variant = BTRAPZ_TRAPEZOID;
  btrapz_options opt;
  btrapz_options_init(&opt);
  opt.lean = -1; opt.split = -1; opt.cap_iter = -1;   /* one form whatever the shard size: the winner's bits do not depend on G */

  btrapz_multi *m = NULL;
  int rc = btrapz_multi_create(&m, devices, G, transport);
  if (rc != BTRAPZ_OK) { fprintf(stderr, "btrapz_multi_create -> %d\n", rc); return 4; }
  if ((rc = btrapz_multi_upload(m, B, S, 0, seg, init, ref_end, dlb)) != BTRAPZ_OK ||
      (rc = btrapz_multi_solve_argmin(m, &sh, &opt)) != BTRAPZ_OK) {
    fprintf(stderr, "step -> %d: %s\n", rc, btrapz_multi_last_error(m)); return 5;
  }
  long long idx = -2; double cost = 0.0;
  double *ctrl = calloc((size_t)12 * S, sizeof(double));
  if ((rc = btrapz_multi_result(m, -1, &idx, &cost, ctrl)) != BTRAPZ_OK) { fprintf(stderr, "result -> %d: %s\n", rc, btrapz_multi_last_error(m)); return 6; }
  printf("winner %lld cost %.17g transport %d ctrl0 %.17g ctrl_last %.17g\n", idx, cost, btrapz_multi_transport(m), ctrl[1], ctrl[12 * S - 1]);
  btrapz_multi_destroy(m);
  free(seg); free(init); free(ref_end); free(dlb); free(ctrl);
  return 0;
}
