// simt_shim.h -- what a host compiler needs to build the product's wavefront code (prism_core.h) for the sanitizer
// run: a "wavefront" is 64 std::threads, __syncthreads a real barrier, __ballot a shared mask between two barriers.
// Test infrastructure (Makefile target host_asan); no product code includes it.
#ifndef BTRAPZ_SIMT_SHIM_H
#define BTRAPZ_SIMT_SHIM_H
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <mutex>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)

struct double2 { double x, y; };
static inline double2 make_double2(double x, double y) { return double2{x, y}; }

namespace simt {
enum { WAVE = 64 };
struct Barrier {
  std::mutex m; std::condition_variable cv; int waiting = 0; unsigned long generation = 0;
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    const unsigned long g = generation;
    if (++waiting == WAVE) { waiting = 0; ++generation; cv.notify_all(); }
    else cv.wait(lk, [&] { return generation != g; });
  }
};
struct Wave { Barrier barrier; std::atomic<unsigned long long> mask{0}; };
extern thread_local Wave *wave;
extern thread_local int lane;
}  // namespace simt

static inline void __syncthreads() { simt::wave->barrier.wait(); }
static inline unsigned long long __ballot(bool pred) {
  if (pred) simt::wave->mask.fetch_or(1ull << simt::lane);
  simt::wave->barrier.wait();
  const unsigned long long m = simt::wave->mask.load();
  simt::wave->barrier.wait();
  if (simt::lane == 0) simt::wave->mask.store(0);
  simt::wave->barrier.wait();
  return m;
}
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
#endif
