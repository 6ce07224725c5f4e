// traj_cost.h -- the scalar find_traj returns (plain C++: part of the host code the sanitizer build covers, Makefile
// target host_asan).
#ifndef BTRAPZ_TRAJ_COST_H
#define BTRAPZ_TRAJ_COST_H
#include <cmath>

#include "../../include/btrapz_hip.h"
#include "corridor.hpp"

namespace btrapz {

inline int clampi(int i, int hi) { return i < 0 ? 0 : (i > hi ? hi : i); }

// a_cost of trp_wrapper.cpp:207-286 / cub_wrapper.cpp:201-262.  Reads of x_ref[i] past N
// and of l[N-1] past the sampled length (undefined in the reference) are clamped.
inline double trajectory_cost(int variant, const Params &p, const TrajInput &in, int np, const double *s, const double *ds,
                       const double *dds, const double *l, const double *dl, const double *ddl) {
  const double dt = in.delta;
  const int N = in.N;
  double s_cost = 0.0, l_cost = 0.0, max_a = 0.0;
  for (int i = 0; i < np; ++i) {
    const double jerk = (i == 0) ? (dds[np > 1 ? 1 : 0] - dds[0]) / dt : (dds[i] - dds[i - 1]) / dt;
    const double e = s[i] - in.s_ref[clampi(i, N - 1)];
    if (variant == BTRAPZ_TRAPEZOID) {
      s_cost += p.weight_s_ref * e * e * dt;
      s_cost += p.weight_ds_ref * ds[i] * ds[i] * dt;
      s_cost += p.s_acc_weight * dds[i] * dds[i] * dt;
      s_cost += p.s_jerk_weight * jerk * jerk * dt;
    } else {
      s_cost += e * e * dt;
      s_cost += ds[i] * ds[i] * dt;
      s_cost += dds[i] * dds[i] * dds[i] * dds[i] * dt;
      s_cost += jerk * jerk * jerk * jerk * dt;
    }
    max_a = std::fmax(max_a, std::fabs(dds[i]));
  }
  if (variant == BTRAPZ_CUBOID) s_cost += max_a * max_a * max_a * max_a;
  max_a = 0.0;
  for (int i = 0; i < np; ++i) {
    const double jerk = (i == 0) ? (ddl[np > 1 ? 1 : 0] - ddl[0]) / dt : (ddl[i] - ddl[i - 1]) / dt;
    const double e = l[i] - in.l_ref[clampi(i, N - 1)];
    if (variant == BTRAPZ_TRAPEZOID) {
      l_cost += p.weight_l_ref * e * e * dt;
      l_cost += p.weight_dl_ref * dl[i] * dl[i] * dt;
      l_cost += p.l_acc_weight * ddl[i] * ddl[i] * dt;
      l_cost += p.l_jerk_weight * jerk * jerk * dt;
    } else {
      l_cost += e * e * dt;
      l_cost += dl[i] * dl[i] * dt;
      l_cost += ddl[i] * ddl[i] * dt;
      l_cost += jerk * jerk * dt;
    }
    max_a = std::fmax(max_a, std::fabs(ddl[i]));
  }
  if (variant == BTRAPZ_TRAPEZOID) {
    const double e = l[clampi(N - 1, np - 1)] - in.l_ref[N - 1];
    l_cost += p.weight_end_l * e * e * dt;
  } else {
    l_cost += max_a * max_a;
  }
  return s_cost + l_cost;
}

}  // namespace btrapz
#endif
