// prism_kernels.hip -- obstacle prisms -> per-knot corridor bounds, batched on the device.
//
// Replaces, for candidate sets, the scene logic in front of the corridor text file: Car.getCar + get_bounds of the
// reference's harness (src/cart_frenet.py:664-1030; helper lineFromPoints :818-830).  Every car is a prism in
// (s, l, t) -- centre (s0, l0, t0), constant velocities, duration T, grown by l_safe / w_safe (:694-700).  Its lateral
// extent cuts the road into strips; a strip covered by a car gets, inside the car's time window, the car's rear face
// as upper s bound when the car starts at t0 = 0 and its front face as lower s bound otherwise (:905-945); strips
// nobody covers are free, several cars over one strip intersect (:987-995).  The reference reaches that through a
// class-level edge list filled pairwise at construction and hash-ordered de-duplication, so its output depends on
// construction order when extents overlap; here the geometry has ONE definition (oracle/prism_oracle.py states it
// and is held to the reference's own output on every scene where that output is well formed):
//   edges = sorted distinct l_min, l_max of the cars (+ the road's edges where the cars leave room);
//   strip j = [e_j, e_j+1]; bounds = intersection over the cars whose lateral extent contains the strip of
//   (s_lo, rear(i)) | (front(i), s_hi) inside the window, (s_lo, s_hi) outside; free when no car covers it.
// Output: the arrays btrapz_corridor_batch_device reads (s_bounds, l_bounds [B][O][N][2]), one "obstacle corridor"
// per strip.  One wavefront per scene; the fill is one 16-byte store per lane and knot, coalesced along the knots --
// the stage is write-bound (O * N * 32 bytes per scene).
#include <hip/hip_runtime.h>

#include "btrapz_device.h"

namespace btrapz {

// round(x, 2) of Python (correctly rounded to two decimals, ties to even in exact arithmetic): the reference rounds
// every face value (lineFromPoints, cart_frenet.py:827).  y = x * 100 is rounded once; the exact residual of that
// product decides an apparent tie.
__device__ __forceinline__ double round2(double x) {
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

struct PrismArgs {
  int B, P, N, O;
  double rate;                 // knots per second (the reference hard-codes 10: `i/10`, `t0*10`)
  double s_lo, s_hi, l_lo, l_hi, l_safe, w_safe;
  const double *prisms;        // [B][P][8]: s0, l0, t0, vel_s, vel_l, T, active, reserved
  double *s_bounds, *l_bounds; // [B][O][N][2]
  int *n_strips;               // [B]
};

__global__ __launch_bounds__(64) void prism_bounds_kernel(const PrismArgs a) {
#pragma clang fp contract(off)   // the reference's expressions, operation by operation (no fused multiply-adds)
  __shared__ double car[16][8];          // s0, t0, vs, T, l_min, l_max, y1, c (slope)
  __shared__ int car_on[16], car_ahead[16];
  __shared__ double cand[34], edge[34];
  __shared__ int cand_on[34];
  const int lane = threadIdx.x, b = blockIdx.x;
  const int P = a.P, N = a.N, O = a.O;
  if (lane < P) {
    const double *p = a.prisms + ((size_t)b * P + lane) * 8;
    const double s0 = p[0], l0 = p[1], t0 = p[2], vs = p[3], vl = p[4], T = p[5];
    const bool on = p[6] != 0.0;
    const double fl = l0 + vl * T;                                    // forw_state[1] (:704)
    const double lmin = vl >= 0 ? l0 - a.w_safe : fl - a.w_safe;      // :709-710 / :764-765
    const double lmax = vl >= 0 ? fl + a.w_safe : l0 + a.w_safe;
    const bool ahead = t0 == 0.0;                                     // :905
    const double fs = s0 + vs * T;                                    // forw_state[0]
    const double y1 = ahead ? s0 - a.l_safe : s0 + a.l_safe;          // corner 0 / corner 2 (:727-741)
    const double y2 = ahead ? fs - a.l_safe : fs + a.l_safe;          // corner 4 / corner 6
    const double x1 = t0, x2 = t0 + T;
    car[lane][0] = s0; car[lane][1] = t0; car[lane][2] = vs; car[lane][3] = T; car[lane][4] = lmin; car[lane][5] = lmax;
    car[lane][6] = y1; car[lane][7] = (y2 - y1) / (x2 - x1);          // lineFromPoints: c = a / b (:820-822)
    car_on[lane] = on; car_ahead[lane] = ahead;
    cand[lane] = lmin; cand[P + lane] = lmax; cand_on[lane] = on; cand_on[P + lane] = on;
  }
  __syncthreads();
  // the road's own edges only where the cars leave room (:881-887, :969-975)
  if (lane == 0) {
    double mn = 1e300, mx = -1e300;
    for (int j = 0; j < 2 * P; j++) if (cand_on[j]) { mn = cand[j] < mn ? cand[j] : mn; mx = cand[j] > mx ? cand[j] : mx; }
    cand[2 * P] = a.l_lo; cand_on[2 * P] = mn > a.l_lo;
    cand[2 * P + 1] = a.l_hi; cand_on[2 * P + 1] = mx < a.l_hi;
  }
  __syncthreads();
  // sorted distinct edges: a candidate counts if no earlier candidate has its value; its slot = distinct values below
  const int nc = 2 * P + 2;
  bool mine = false; double v = 0.0;
  if (lane < nc && cand_on[lane]) {
    v = cand[lane]; mine = true;
    for (int j = 0; j < lane; j++) if (cand_on[j] && cand[j] == v) mine = false;
  }
  const unsigned long long firsts = __ballot(mine);
  if (mine) {
    int rank = 0;
    for (int j = 0; j < nc; j++) if (((firsts >> j) & 1ull) && cand[j] < v) ++rank;
    edge[rank] = v;
  }
  __syncthreads();
  const int E = __popcll(firsts);
  const int strips = E > 0 ? E - 1 : 0;
  if (lane == 0) a.n_strips[b] = strips <= O ? strips : -1;
  double2 *sb = reinterpret_cast<double2 *>(a.s_bounds) + (size_t)b * O * N;
  double2 *lb = reinterpret_cast<double2 *>(a.l_bounds) + (size_t)b * O * N;
  for (int idx = lane; idx < O * N; idx += 64) {
    const int j = idx / N, i = idx - j * N;
    if (j >= strips) {   // padding: a corridor no reference trajectory can be inside
      sb[idx] = make_double2(0.0, 0.0); lb[idx] = make_double2(1e9, 1e9);
      continue;
    }
    const double e0 = edge[j], e1 = edge[j + 1];
    double lo = a.s_lo, hi = a.s_hi;
    bool first = true;
    for (int p = 0; p < P; p++) {
      if (!car_on[p] || !(car[p][4] <= e0 && e1 <= car[p][5])) continue;
      const double t0 = car[p][1], T = car[p][3], c = car[p][7];
      const bool inside = !((double)i < t0 * a.rate || (double)i > (t0 + T) * a.rate);      // :909-913
      const double y = round2(c * (double)i / a.rate - c * t0 + car[p][6]);                 // :827
      const double c_lo = (inside && !car_ahead[p]) ? y : a.s_lo;
      const double c_hi = (inside && car_ahead[p]) ? y : a.s_hi;
      if (first) { lo = c_lo; hi = c_hi; first = false; }
      else { lo = c_lo > lo ? c_lo : lo; hi = c_hi < hi ? c_hi : hi; }                        // :989-992
    }
    sb[idx] = make_double2(lo, hi); lb[idx] = make_double2(e0, e1);
  }
}

}  // namespace btrapz

using namespace btrapz;

BTRAPZ_EXPORT int btrapz_prism_bounds_device(btrapz_ctx *c, int B, int P, int N, const btrapz_road *road, const double *prisms,
                                          int O, double *s_bounds, double *l_bounds, int *n_strips, void *stream) {
  if (!c) return BTRAPZ_EINVAL;
  if (B < 1 || P < 1 || P > 16 || N < 1 || O < 1 || !road || !(road->knots_per_second > 0) || !prisms || !s_bounds ||
      !l_bounds || !n_strips)
    return BTRAPZ_EINVAL;
  if (hipSetDevice(btrapz_ctx_device(c)) != hipSuccess) return BTRAPZ_EHIP;
  PrismArgs a;
  a.B = B; a.P = P; a.N = N; a.O = O; a.rate = road->knots_per_second;
  a.s_lo = road->s_lo; a.s_hi = road->s_hi; a.l_lo = road->l_lo; a.l_hi = road->l_hi;
  a.l_safe = road->l_safe; a.w_safe = road->w_safe;
  a.prisms = prisms; a.s_bounds = s_bounds; a.l_bounds = l_bounds; a.n_strips = n_strips;
  hipLaunchKernelGGL(prism_bounds_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? BTRAPZ_OK : BTRAPZ_EHIP;
}
