// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the solve kernel uses.
//
// MI355X_MICROARCH.md (HBM): FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B per
// lane); other widths are uncalibrated -- "calibrate on a known byte count in your own access pattern".  The solve
// kernel reads its batch record with 8 B per lane (one double per lane, 480-512 contiguous bytes per wave-instruction)
// and stores control points with 8 B per lane at a 48-B lane stride.  This program streams a buffer far larger than the
// 256 MiB Infinity Cache with exactly those patterns so that the byte count per launch is known:
//     ./fetch_calib <mode> [MiB]      mode 8: 8 B/lane reads   16: 16 B/lane reads   48: 8 B stores at 48 B lane stride
// Run under `rocprofv3 --pmc FETCH_SIZE -- ./fetch_calib 8` (and WRITE_SIZE); tools/summarize_profiles.py divides.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

__global__ void read8(const double *__restrict__ p, size_t n, double *out) {
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 123.456) out[0] = acc;   // never true: keeps the loads
}
__global__ void read16(const double2 *__restrict__ p, size_t n, double *out) {
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double2 v = p[i];
    acc += v.x + v.y;
  }
  if (acc == 123.456) out[0] = acc;
}
// lane l of a wavefront writes 6 consecutive doubles at l * 48 B: the control-point store of the solve kernel
__global__ void write48(double *__restrict__ p, size_t nrec) {
  for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrec; r += (size_t)gridDim.x * blockDim.x) {
    double *d = p + r * 6;
    for (int i = 0; i < 6; i++) d[i] = (double)(r + i);
  }
}

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char **argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 8;
  const size_t mib = argc > 2 ? (size_t)atoll(argv[2]) : 2048;
  const size_t bytes = mib << 20;
  double *buf, *out;
  CHK(hipMalloc(&buf, bytes));
  CHK(hipMalloc(&out, 8));
  CHK(hipMemset(buf, 0, bytes));
  CHK(hipDeviceSynchronize());
  const int reps = 3;
  for (int r = 0; r < reps; r++) {
    if (mode == 8) hipLaunchKernelGGL(read8, dim3(256 * 16), dim3(256), 0, 0, buf, bytes / 8, out);
    else if (mode == 16) hipLaunchKernelGGL(read16, dim3(256 * 16), dim3(256), 0, 0, (const double2 *)buf, bytes / 16, out);
    else hipLaunchKernelGGL(write48, dim3(256 * 16), dim3(256), 0, 0, buf, bytes / 48);
    CHK(hipGetLastError());
  }
  CHK(hipDeviceSynchronize());
  printf("{\"mode\": %d, \"bytes_per_launch\": %zu, \"launches\": %d}\n", mode, mode == 48 ? (bytes / 48) * 48 : bytes, reps);
  return 0;
}
