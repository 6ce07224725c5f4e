/* Sanitizer self-test of the oracle (CPU only): runs the whole find_traj restatement and the batch entry
 * on the committed inputs under -fsanitize=address,undefined.  `make -C oracle selftest`. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "btrapz_oracle.h"

int orc_ipm_solve(const orc_qp *qp, double eps, int max_iter, double *x, double *y, orc_info *info);

int main(int argc, char **argv) {
  const char *dir = argc > 1 ? argv[1] : "../tests/golden/inputs";
  const char *names[] = {"c1", "c2", "c3", "c4", "c6", "c7", "c7_7", "c_road_s1", "c_road_s1_2", "c_road_s1_3"};
  orc_params p = {35.73, 41.61, 25.57, 41.59, 0.12, 10.04, 0.71, 14.3, 7.27, 32.13, 3};
  int fails = 0;
  for (unsigned i = 0; i < sizeof(names) / sizeof(names[0]); i++)
    for (int v = 0; v < 2; v++) {
      char path[512]; snprintf(path, sizeof(path), "%s/%s.txt", dir, names[i]);
      int S = 0; double ctrl[12 * 64]; orc_cube cubes[64]; orc_info info;
      orc_settings st; orc_settings_reference(&st); st.max_iter = 600;
      double cost = orc_find_traj(v, path, NULL, &p, &st, &S, ctrl, cubes, &info);
      printf("%-12s var %d S %2d status %3d iter %4d cost %.6g\n", names[i], v, S, info.status, info.iter, cost);
      if (S < 0 && S != -2) fails++;
    }
  /* batch entry + exact solver on a tiny synthetic record */
  enum { B = 3, S = 4, F = 17 };
  double seg[F * B * S], init[B * 6], ref_end[B * 2], dl[B * 10], shared[21] = {0.12, 10.04, 35.73, 41.61, 0.71, 14.3, 25.57, 41.59, 7.27, 32.13, 7.0, 0.0, -2, 2, -30, 30, -0.7, 0.7, -10, 10, 0.1};
  memset(seg, 0, sizeof(seg)); memset(init, 0, sizeof(init));
  for (int b = 0; b < B; b++) {
    for (int k = 0; k < S; k++) {
      double *e = seg + b * S + k;
      e[0 * B * S] = 1.0;                                   /* t */
      e[1 * B * S] = 5.0 * k - 6; e[2 * B * S] = 5.0; e[3 * B * S] = 5.0 * k + 6; e[4 * B * S] = 5.0;   /* s lines */
      e[5 * B * S] = 1.0; e[7 * B * S] = 3.0; e[9 * B * S] = 1.0; e[10 * B * S] = 3.0;                      /* l lines / box */
      e[11 * B * S] = 0.0; e[12 * B * S] = 50.0; e[13 * B * S] = 5.0; e[14 * B * S] = 5.0 * k; e[16 * B * S] = 2.0;
    }
    init[b * 6 + 1] = 5.0; init[b * 6 + 3] = 2.0; ref_end[2 * b] = 5.0 * S; ref_end[2 * b + 1] = 2.0;
    for (int i = 0; i < 5; i++) { dl[10 * b + 2 * i] = -2; dl[10 * b + 2 * i + 1] = 2; }
  }
  double ctrl[B * 12 * S], obj[B]; int status[B], iters[B];
  for (int exact = 0; exact < 2; exact++) {
    orc_batch_solve(0, B, S, seg, init, ref_end, dl, shared, NULL, exact, 0, B, ctrl, obj, status, iters);
    printf("batch exact=%d status %d %d %d obj %.6f\n", exact, status[0], status[1], status[2], obj[0]);
    if (status[0] != 1) fails++;
  }
  printf(fails ? "SELFTEST FAILED\n" : "selftest ok\n");
  return fails;
}
