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
#include "prism_core.h"

namespace btrapz {

struct PrismArgs {
  int B, P, N, O;
  PrismRoad road;
  const double *prisms;        // [B][P][8]: s0, l0, t0, vel_s, vel_l, T, active, reserved
  double *s_bounds, *l_bounds; // [B][O][N][2]
  int *n_strips;               // [B]
};

__global__ __launch_bounds__(64) void prism_bounds_kernel(const PrismArgs a) {
  __shared__ double tab_mem[(sizeof(double) * 8 * PRISM_MAX_CARS + sizeof(double) * 2 * (2 * PRISM_MAX_CARS + 2) +
                             sizeof(int) * (5 * PRISM_MAX_CARS + 3) + 8 + 7) / 8];
  const int lane = threadIdx.x, b = blockIdx.x;
  const int N = a.N, O = a.O;
  PrismTab t = prism_tab_at(tab_mem, a.P);
  const int strips = prism_tables(t, a.road, a.prisms + (size_t)b * a.P * 8, lane);
  if (lane == 0) a.n_strips[b] = strips <= O ? strips : -1;
  double2 *sb = reinterpret_cast<double2 *>(a.s_bounds) + (size_t)b * O * N;
  double2 *lb = reinterpret_cast<double2 *>(a.l_bounds) + (size_t)b * O * N;
  for (int idx = lane; idx < O * N; idx += 64) {
    const int j = idx / N, i = idx - j * N;
    sb[idx] = prism_pair_s(t, a.road, j, i); lb[idx] = prism_pair_l(t, j);
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
  a.B = B; a.P = P; a.N = N; a.O = O; a.road = prism_road(road);
  a.prisms = prisms; a.s_bounds = s_bounds; a.l_bounds = l_bounds; a.n_strips = n_strips;
  hipLaunchKernelGGL(prism_bounds_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? BTRAPZ_OK : BTRAPZ_EHIP;
}
