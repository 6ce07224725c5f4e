// prism_core.h -- the strip geometry of the prism stage, shared by prism_bounds_kernel (strips
// written to memory) and the fused prism + corridor kernel (strips evaluated where the corridor stage reads them):
// ONE statement of every expression, so the two give the same bits.
// Reference: Car.getCar + get_bounds of the harness, src/cart_frenet.py:664-1030; lineFromPoints :818-830.
#ifndef BTRAPZ_PRISM_CORE_H
#define BTRAPZ_PRISM_CORE_H
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#endif
// (a host compiler -- the sanitizer build, host_check/ -- brings its own definitions of __device__, double2,
//  __syncthreads and __ballot: one thread per lane, a real barrier)

#include "btrapz_device.h"

namespace btrapz {

enum { PRISM_MAX_CARS = 16 };

inline PrismRoad prism_road(const btrapz_road *road) {
  PrismRoad r;
  r.rate = road->knots_per_second;
  r.s_lo = road->s_lo; r.s_hi = road->s_hi; r.l_lo = road->l_lo; r.l_hi = road->l_hi;
  r.l_safe = road->l_safe; r.w_safe = road->w_safe;
  return r;
}

// round(x, 2) of Python (correctly rounded to two decimals, ties to even in exact arithmetic): the reference rounds
// every face value (lineFromPoints, cart_frenet.py:827).  y = x * 100 is rounded once; the exact residual of that
// product decides an apparent tie.
__device__ __forceinline__ double round2(double x) {
#pragma clang fp contract(off)
  const double y = x * 100.0;
  const double e = __builtin_fma(x, 100.0, -y);
  double r = __builtin_rint(y);
  const double d = y - __builtin_trunc(y);
  if (d == 0.5 || d == -0.5) {          // y sits on a tie: the true product is y + e
    const double lo = __builtin_floor(y), hi = lo + 1.0;
    if (e > 0.0) r = hi; else if (e < 0.0) r = lo;   // e == 0: rint's half-even is the answer
  }
  return r / 100.0;
}

// Tables of one scene in LDS, sized by the scene's P cars: car[P][8] | edge[2P+2] | cand[2P+2] | cover[2P+1] |
// flags[P] | cand_on[2P+2].  car: first and last knot of the window (t0 rate, (t0 + T) rate), y1, c (slope of the
// face), l_min, l_max, c t0; flags: bit 0 active, bit 1 "ahead"
// (starts at t0 = 0: its rear face bounds s from above); cover[j]: the cars whose lateral extent contains strip j.
struct PrismTab {
  double *car, *edge, *cand;
  int *cover, *flags, *cand_on;
  int P, strips;
};
__host__ __device__ inline size_t prism_tab_bytes(int P) {
  return sizeof(double) * (8 * (size_t)P + 2 * (2 * (size_t)P + 2)) + sizeof(int) * ((2 * (size_t)P + 1) + P + (2 * (size_t)P + 2)) + 8;
}
__device__ __forceinline__ PrismTab prism_tab_at(void *lds, int P) {   // lds: 8-byte aligned
  PrismTab t;
  t.P = P; t.strips = 0;
  t.car = reinterpret_cast<double *>(lds);
  t.edge = t.car + 8 * P;
  t.cand = t.edge + 2 * P + 2;
  t.cover = reinterpret_cast<int *>(t.cand + 2 * P + 2);
  t.flags = t.cover + 2 * P + 1;
  t.cand_on = t.flags + P;
  return t;
}

// Fills the tables of scene `p` ([P][8]: s0, l0, t0, vel_s, vel_l, T, active, reserved) with one wavefront (or more:
// every thread of the workgroup must call it; lanes beyond 2P + 2 idle).  Returns the number of strips.
__device__ __forceinline__ int prism_tables(PrismTab &t, const PrismRoad &r, const double *p, int lane) {
#pragma clang fp contract(off)   // the reference's expressions, operation by operation (no fused multiply-adds)
  const int P = t.P;
  if (lane < P) {
    const double *q = p + (size_t)lane * 8;
    const double s0 = q[0], l0 = q[1], t0 = q[2], vs = q[3], vl = q[4], T = q[5];
    const bool on = q[6] != 0.0;
    const double fl = l0 + vl * T;                                    // forw_state[1] (:704)
    const double lmin = vl >= 0 ? l0 - r.w_safe : fl - r.w_safe;      // :709-710 / :764-765
    const double lmax = vl >= 0 ? fl + r.w_safe : l0 + r.w_safe;
    const bool ahead = t0 == 0.0;                                     // :905
    const double fs = s0 + vs * T;                                    // forw_state[0]
    const double y1 = ahead ? s0 - r.l_safe : s0 + r.l_safe;          // corner 0 / corner 2 (:727-741)
    const double y2 = ahead ? fs - r.l_safe : fs + r.l_safe;          // corner 4 / corner 6
    const double x1 = t0, x2 = t0 + T;
    double *c = t.car + 8 * lane;
    const double cc = (y2 - y1) / (x2 - x1);                          // lineFromPoints: c = a / b (:820-822)
    c[0] = t0 * r.rate; c[1] = (t0 + T) * r.rate; c[2] = y1; c[3] = cc;
    c[4] = lmin; c[5] = lmax; c[6] = cc * t0;
    t.flags[lane] = (on ? 1 : 0) | (ahead ? 2 : 0);
    t.cand[lane] = lmin; t.cand[P + lane] = lmax; t.cand_on[lane] = on; t.cand_on[P + lane] = on;
  }
  __syncthreads();
  // the road's own edges only where the cars leave room (:881-887, :969-975)
  if (lane == 0) {
    double mn = 1e300, mx = -1e300;
    for (int j = 0; j < 2 * P; j++) if (t.cand_on[j]) { mn = t.cand[j] < mn ? t.cand[j] : mn; mx = t.cand[j] > mx ? t.cand[j] : mx; }
    t.cand[2 * P] = r.l_lo; t.cand_on[2 * P] = mn > r.l_lo;
    t.cand[2 * P + 1] = r.l_hi; t.cand_on[2 * P + 1] = mx < r.l_hi;
  }
  __syncthreads();
  // sorted distinct edges: a candidate counts if no earlier candidate has its value; its slot = distinct values below
  const int nc = 2 * P + 2;
  bool mine = false; double v = 0.0;
  if (lane < nc && t.cand_on[lane]) {
    v = t.cand[lane]; mine = true;
    for (int j = 0; j < lane; j++) if (t.cand_on[j] && t.cand[j] == v) mine = false;
  }
  const unsigned long long firsts = __ballot(mine);   // (one wavefront holds all 2P + 2 <= 34 candidates)
  if (mine) {
    int rank = 0;
    for (int j = 0; j < nc; j++) if (((firsts >> j) & 1ull) && t.cand[j] < v) ++rank;
    t.edge[rank] = v;
  }
  __syncthreads();
  const int E = __popcll(firsts);
  t.strips = E > 0 ? E - 1 : 0;
  if (lane < t.strips) {
    const double e0 = t.edge[lane], e1 = t.edge[lane + 1];
    int m = 0;
    for (int q = 0; q < P; q++)
      if ((t.flags[q] & 1) && t.car[8 * q + 4] <= e0 && e1 <= t.car[8 * q + 5]) m |= 1 << q;
    t.cover[lane] = m;
  }
  __syncthreads();
  return t.strips;
}

// s bounds (lower, upper) of strip j < strips at knot i: the intersection over the cars that cover the strip
// (:905-945, :987-995)
__device__ __forceinline__ double2 prism_strip_s(const PrismTab &t, const PrismRoad &r, int j, int i) {
#pragma clang fp contract(off)
  double lo = r.s_lo, hi = r.s_hi;
  bool first = true;
  for (int m = t.cover[j]; m; m &= m - 1) {
    const int q = __builtin_ctz(m);
    const double *c = t.car + 8 * q;
    const bool ahead = (t.flags[q] & 2) != 0;
    const bool inside = !((double)i < c[0] || (double)i > c[1]);                          // :909-913
    const double y = round2(c[3] * (double)i / r.rate - c[6] + c[2]);                     // :827
    const double c_lo = (inside && !ahead) ? y : r.s_lo;
    const double c_hi = (inside && ahead) ? y : r.s_hi;
    if (first) { lo = c_lo; hi = c_hi; first = false; }
    else { lo = c_lo > lo ? c_lo : lo; hi = c_hi < hi ? c_hi : hi; }                      // :989-992
  }
  return make_double2(lo, hi);
}
// Padding strips (j >= strips) are corridors no reference trajectory can be inside.
__device__ __forceinline__ double2 prism_pair_s(const PrismTab &t, const PrismRoad &r, int j, int i) {
  return j < t.strips ? prism_strip_s(t, r, j, i) : make_double2(0.0, 0.0);
}
__device__ __forceinline__ double2 prism_pair_l(const PrismTab &t, int j) {
  return j < t.strips ? make_double2(t.edge[j], t.edge[j + 1]) : make_double2(1e9, 1e9);
}

}  // namespace btrapz
#endif
