// btrapz_multi.hip -- the multi-GPU step behind the C-ABI (include/btrapz_hip.h, btrapz_multi_*): ONE host process,
// one context + stream per device, the candidates sharded contiguously, per-device solve + local arg-min, ONE gather of
// G x (16 + 96 S) bytes (cost bits, global index, the local winner's control points), a lexicographic min on every
// device.  The reference has no counterpart: it solves one corridor per call inside a single-process replanning loop
// (src/cart_frenet.py:1516-1571, src/solve_3d.cc:1231-1414); north_star asks for candidate sets sharded over the GPUs of
// one node "with RCCL over xGMI only for the final arg-min reduction", the host staying C++ over this thin C-ABI.
//
// Transport of the one gather:
//   RCCL   ncclAllGather inside ncclGroupStart/End, one communicator per device (ncclCommInitAll), resolved from
//          librccl.so at RUN time (dlopen: the library carries no link dependency on RCCL) -- next to the HIP runtime
//          the process already holds, so that a second runtime is never pulled in;
//   copies stream-ordered peer copies (hipMemcpyPeerAsync; same-device copies for logical devices): the fallback when
//          RCCL is absent, and the only transport when a device ordinal repeats.
// Logical devices: the same ordinal may appear several times in `devices` -- every entry still gets its own context,
// stream, buffers and shard, and the whole path (shard, solve, local arg-min, pack, gather, select) runs on a 1-GPU box.
#include <hip/hip_runtime.h>
// RCCL: types and prototypes only -- every function is called through dlsym.  A ROCm installation without the rccl
// development headers still builds the whole library (single-GPU path and drop-in shims included): the handful of
// declarations the step needs are then stated here, as RCCL's public header states them.
#if __has_include(<rccl/rccl.h>) && !defined(BTRAPZ_NO_RCCL_HEADER)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5,
               ncclFloat16 = 6, ncclHalf = 6, ncclFloat32 = 7, ncclFloat = 7, ncclFloat64 = 8, ncclDouble = 8 } ncclDataType_t;
ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
const char *ncclGetErrorString(ncclResult_t result);
ncclResult_t ncclGetVersion(int *version);
}
#endif

#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "btrapz_device.h"

namespace {

// ---- the two small kernels of the step --------------------------------------------------------------------------------
// rec [n][2 + P] int64: (bit pattern of the local winner's cost, its GLOBAL index or -1, its P = 12 S control points)
__global__ void multi_pack_kernel(int n, int P, long long index_base, const long long *loc_idx, const double *loc_cost,
                                  const double *ctrl, long long *rec) {
  const int g = blockIdx.x;
  if (g >= n) return;
  const long long gi = loc_idx[g];
  long long *r = rec + (size_t)g * (2 + P);
  if (threadIdx.x == 0) { r[0] = __double_as_longlong(loc_cost[g]); r[1] = gi; }
  const double *src = gi >= 0 ? ctrl + (size_t)(gi - index_base) * P : nullptr;
  for (int j = threadIdx.x; j < P; j += blockDim.x) r[2 + j] = src ? __double_as_longlong(src[j]) : 0;
}
// gathered [world][n][2 + P]: lexicographic min over the ranks (cost first, a NaN never wins, then the lowest index --
// the order of argmin_pairs_kernel), and the owner's control points; NaN where nobody solved a candidate.
__global__ void multi_select_kernel(int world, int n, int P, const long long *gathered, double *best_cost,
                                    long long *best_idx, double *best_ctrl) {
  const int g = blockIdx.x;
  if (g >= n) return;
  double bc = __builtin_huge_val();
  long long bi = -1;
  int owner = -1;
  for (int r = 0; r < world; r++) {
    const long long *rec = gathered + ((size_t)r * n + g) * (2 + P);
    double c = __longlong_as_double(rec[0]);
    const long long i = rec[1];
    if (!(c == c)) c = __builtin_huge_val();
    if (c < bc || (c == bc && i >= 0 && (bi < 0 || i < bi))) { bc = c; bi = i; owner = r; }
  }
  if (bi < 0) owner = -1;
  if (threadIdx.x == 0) { best_cost[g] = bc; best_idx[g] = bi; }
  const long long *rec = owner >= 0 ? gathered + ((size_t)owner * n + g) * (2 + P) + 2 : nullptr;
  for (int j = threadIdx.x; j < P; j += blockDim.x)
    best_ctrl[(size_t)g * P + j] = rec ? __longlong_as_double(rec[j]) : __longlong_as_double(0x7ff8000000000000LL);
}

struct Rccl {
  void *handle = nullptr;
  std::string path;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  bool ok() const { return handle && CommInitAll && CommDestroy && AllGather && GroupStart && GroupEnd && GetErrorString; }
};

// librccl.so of the ROCm installation whose HIP runtime this process already runs on: BTRAPZ_RCCL_LIB, a copy that is
// loaded already (PyTorch-ROCm brings its own), the directory of the loaded libamdhip64, then the loader's search path.
bool load_rccl(Rccl &r, std::string &why) {
  std::vector<std::string> tries;
  if (const char *e = getenv("BTRAPZ_RCCL_LIB")) if (*e) tries.push_back(e);
  for (const char *name : {"librccl.so", "librccl.so.1"}) {
    if (void *h = dlopen(name, RTLD_NOW | RTLD_NOLOAD)) { r.handle = h; r.path = std::string(name) + " (already loaded)"; break; }
  }
  if (!r.handle) {
    Dl_info info;
    if (dladdr((void *)&hipGetDeviceCount, &info) && info.dli_fname) {
      std::string dir(info.dli_fname);
      const size_t slash = dir.rfind('/');
      if (slash != std::string::npos) { dir.resize(slash); tries.push_back(dir + "/librccl.so"); tries.push_back(dir + "/librccl.so.1"); }
    }
    tries.push_back("librccl.so"); tries.push_back("librccl.so.1");
    for (const std::string &p : tries) {
      if (void *h = dlopen(p.c_str(), RTLD_NOW | RTLD_LOCAL)) { r.handle = h; r.path = p; break; }
      const char *de = dlerror();   // (one call: dlerror clears the message it returns)
      why += p + ": " + (de ? de : "?") + "; ";
    }
  }
  if (!r.handle) return false;
#define BTRAPZ_SYM(n) r.n = reinterpret_cast<decltype(r.n)>(dlsym(r.handle, "nccl" #n))
  BTRAPZ_SYM(CommInitAll); BTRAPZ_SYM(CommDestroy); BTRAPZ_SYM(AllGather); BTRAPZ_SYM(GroupStart); BTRAPZ_SYM(GroupEnd);
  BTRAPZ_SYM(GetErrorString); BTRAPZ_SYM(GetVersion);
#undef BTRAPZ_SYM
  if (!r.ok()) { why += r.path + ": a symbol of the collective API is missing"; return false; }
  return true;
}

struct Shard {
  int device = 0;
  btrapz_ctx *ctx = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t packed = nullptr, done = nullptr;
  bool done_recorded = false;
  int B = 0; long long base = 0;
  const double *seg = nullptr, *init = nullptr, *ref_end = nullptr, *dl_bounds = nullptr;
  double *own_in = nullptr; size_t own_in_cap = 0;                 // inputs uploaded through btrapz_multi_upload
  double *ctrl = nullptr, *cost = nullptr; int *status = nullptr, *iters = nullptr; size_t out_cand = 0, out_ctrl = 0;
  long long *loc_idx = nullptr; double *loc_cost = nullptr; size_t loc_cap = 0;
  long long *rec = nullptr, *gathered = nullptr; size_t rec_cap = 0, gat_cap = 0;   // [n][2 + P], [G][n][2 + P]
  double *best_cost = nullptr; long long *best_idx = nullptr; double *best_ctrl = nullptr; size_t best_cap = 0, best_ctrl_cap = 0;
  ncclComm_t comm = nullptr;
};

}  // namespace

struct btrapz_multi {
  int G = 0;
  std::vector<Shard> sh;
  std::string err;
  int transport = BTRAPZ_MULTI_COPIES;
  Rccl rccl;
  bool have_comms = false;
  int B = 0, S = 0, n_groups = 0, group = 0;
  bool global_groups = true;    // arg-min groups span the devices (one gather) / live on one device each (no collective)
  bool solved = false;
};

// The entry points below walk the devices with hipSetDevice: the calling thread gets its current device back on return
// (a caller that shares the thread with PyTorch, or with its own single-device code, must not find it changed).
struct DeviceGuard {
  int dev = -1;
  DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
  ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
};

#define MCHK(m, call)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      (m)->err = std::string(#call) + ": " + hipGetErrorString(e_);                            \
      return BTRAPZ_EHIP;                                                                      \
    }                                                                                          \
  } while (0)
#define NCHK(m, call)                                                                          \
  do {                                                                                         \
    ncclResult_t r_ = (call);                                                                  \
    if (r_ != ncclSuccess) {                                                                   \
      (m)->err = std::string(#call) + ": " + (m)->rccl.GetErrorString(r_);                     \
      return BTRAPZ_EHIP;                                                                      \
    }                                                                                          \
  } while (0)

BTRAPZ_EXPORT const char *btrapz_multi_last_error(const btrapz_multi *m) { return m ? m->err.c_str() : "null handle"; }
BTRAPZ_EXPORT int btrapz_multi_transport(const btrapz_multi *m) { return m ? m->transport : BTRAPZ_EINVAL; }
BTRAPZ_EXPORT int btrapz_multi_device_count(const btrapz_multi *m) { return m ? m->G : BTRAPZ_EINVAL; }
BTRAPZ_EXPORT const char *btrapz_multi_transport_library(const btrapz_multi *m) {
  return (m && m->transport == BTRAPZ_MULTI_RCCL) ? m->rccl.path.c_str() : "";
}

static void free_shard(Shard &s) {
  (void)hipSetDevice(s.device);
  if (s.stream) (void)hipStreamSynchronize(s.stream);
  (void)hipFree(s.own_in); (void)hipFree(s.ctrl); (void)hipFree(s.cost); (void)hipFree(s.status); (void)hipFree(s.iters);
  (void)hipFree(s.loc_idx); (void)hipFree(s.loc_cost); (void)hipFree(s.rec); (void)hipFree(s.gathered);
  (void)hipFree(s.best_cost); (void)hipFree(s.best_idx); (void)hipFree(s.best_ctrl);
  if (s.packed) (void)hipEventDestroy(s.packed);
  if (s.done) (void)hipEventDestroy(s.done);
  if (s.ctx) (void)btrapz_destroy(s.ctx);
  if (s.stream) (void)hipStreamDestroy(s.stream);
  s = Shard();
}

BTRAPZ_EXPORT int btrapz_multi_destroy(btrapz_multi *m) {
  if (!m) return BTRAPZ_EINVAL;
  DeviceGuard guard_;
  if (m->have_comms)
    for (Shard &s : m->sh)
      if (s.comm) { (void)hipSetDevice(s.device); (void)hipStreamSynchronize(s.stream); (void)m->rccl.CommDestroy(s.comm); s.comm = nullptr; }
  for (Shard &s : m->sh) free_shard(s);
  // (the RCCL handle stays loaded: unloading a library with live device state is not worth the risk)
  delete m;
  return BTRAPZ_OK;
}

BTRAPZ_EXPORT int btrapz_multi_create(btrapz_multi **out, const int *devices, int G, int transport) {
  if (!out) return BTRAPZ_EINVAL;
  *out = nullptr;
  if (!devices || G < 1 || G > 64 || transport < BTRAPZ_MULTI_AUTO || transport > BTRAPZ_MULTI_RCCL) return BTRAPZ_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return BTRAPZ_ENODEVICE;
  bool distinct = true;
  for (int g = 0; g < G; g++) {
    if (devices[g] < 0 || devices[g] >= ndev) return BTRAPZ_ENODEVICE;
    for (int h = 0; h < g; h++) distinct = distinct && devices[h] != devices[g];
  }
  DeviceGuard guard_;
  btrapz_multi *m = new btrapz_multi();
  m->G = G; m->sh.resize(G);
  for (int g = 0; g < G; g++) {
    Shard &s = m->sh[g];
    s.device = devices[g];
    int rc = btrapz_create(&s.ctx, s.device);
    if (rc == BTRAPZ_OK && (hipSetDevice(s.device) != hipSuccess || hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess ||
                            hipEventCreateWithFlags(&s.packed, hipEventDisableTiming) != hipSuccess ||
                            hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess))
      rc = BTRAPZ_EHIP;
    if (rc != BTRAPZ_OK) { (void)btrapz_multi_destroy(m); return rc; }
  }
  // direct peer copies where the devices allow them (the copies work without, through the host)
  for (int g = 0; g < G; g++)
    for (int h = 0; h < G; h++)
      if (devices[g] != devices[h]) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, devices[g], devices[h]) == hipSuccess && can) {
          (void)hipSetDevice(devices[g]);
          (void)hipDeviceEnablePeerAccess(devices[h], 0);   // (hipErrorPeerAccessAlreadyEnabled is fine)
          (void)hipGetLastError();
        }
      }
  m->transport = BTRAPZ_MULTI_COPIES;
  if (transport != BTRAPZ_MULTI_COPIES) {
    std::string why;
    if (!distinct) why = "a device ordinal repeats (logical devices): RCCL wants one communicator rank per GPU";
    else if (load_rccl(m->rccl, why)) {
      std::vector<ncclComm_t> comms(G, nullptr);
      const ncclResult_t r = m->rccl.CommInitAll(comms.data(), G, devices);
      if (r == ncclSuccess) {
        for (int g = 0; g < G; g++) m->sh[g].comm = comms[g];
        m->have_comms = true; m->transport = BTRAPZ_MULTI_RCCL;
      } else {
        why = std::string("ncclCommInitAll: ") + m->rccl.GetErrorString(r);
      }
    }
    if (m->transport != BTRAPZ_MULTI_RCCL && transport == BTRAPZ_MULTI_RCCL) {
      // asked for RCCL and nothing else: say why it cannot be had (the handle cannot carry the message: it is not returned)
      fprintf(stderr, "btrapz_multi_create: RCCL transport unavailable: %s\n", why.c_str());
      (void)btrapz_multi_destroy(m);
      return BTRAPZ_ENODEVICE;
    }
    m->err = why;   // (automatic: the reason for the fallback can be read with btrapz_multi_last_error)
  }
  *out = m;
  return BTRAPZ_OK;
}

// Contiguous shards: the rule of spectral_amd/dist.py::shard_bounds (ceil(B / G) per device, the last ones may be short
// or empty); with arg-min groups that live on one device each, a shard is a whole number of groups.
static void shard_bounds(int B, int G, int g, int unit, int *lo, int *hi) {
  const long long units = (B + unit - 1) / unit, per_units = (units + G - 1) / G;
  long long l = (long long)g * per_units * unit, h = l + per_units * unit;
  if (l > B) l = B;
  if (h > B) h = B;
  *lo = (int)l; *hi = (int)h;
}
BTRAPZ_EXPORT int btrapz_multi_shard_bounds(int B, int G, int g, int group, int *lo, int *hi) {
  if (B < 1 || G < 1 || g < 0 || g >= G || !lo || !hi || group < 0 || (group > 0 && B % group != 0)) return BTRAPZ_EINVAL;
  shard_bounds(B, G, g, (group > 0 && group < B) ? group : 1, lo, hi);
  return BTRAPZ_OK;
}

template <class T> static int grow(btrapz_multi *m, T *&p, size_t &cap, size_t need) {
  if (need <= cap) return BTRAPZ_OK;
  (void)hipFree(p); p = nullptr; cap = 0;
  MCHK(m, hipMalloc(&p, sizeof(T) * need));
  cap = need;
  return BTRAPZ_OK;
}

static int ensure_outputs(btrapz_multi *m, Shard &s, int S, int n_local, int n_gather) {
  MCHK(m, hipSetDevice(s.device));
  const size_t cand = (size_t)(s.B > 0 ? s.B : 1), P = 12 * (size_t)S;
  if (cand > s.out_cand) {
    (void)hipFree(s.cost); (void)hipFree(s.status); (void)hipFree(s.iters); s.cost = nullptr; s.status = nullptr; s.iters = nullptr; s.out_cand = 0;
    MCHK(m, hipMalloc(&s.cost, sizeof(double) * cand)); MCHK(m, hipMalloc(&s.status, sizeof(int) * cand)); MCHK(m, hipMalloc(&s.iters, sizeof(int) * cand));
    s.out_cand = cand;
  }
  int rc = grow(m, s.ctrl, s.out_ctrl, cand * P);
  if (rc != BTRAPZ_OK) return rc;
  const size_t nl = (size_t)(n_local > 0 ? n_local : 1), ng = (size_t)(n_gather > 0 ? n_gather : 1);
  if (nl > s.loc_cap) {
    (void)hipFree(s.loc_idx); (void)hipFree(s.loc_cost); s.loc_idx = nullptr; s.loc_cost = nullptr; s.loc_cap = 0;
    MCHK(m, hipMalloc(&s.loc_idx, sizeof(long long) * nl)); MCHK(m, hipMalloc(&s.loc_cost, sizeof(double) * nl));
    s.loc_cap = nl;
  }
  // one arg-min group over the batch: a record of this device and G gathered ones; groups that live on the device: one
  // record per group, packed straight into `gathered` (the select kernel runs on them with world = 1)
  const size_t rec = (m->global_groups ? ng : nl) * (2 + P), gat = m->global_groups ? rec * (size_t)m->G : rec;
  if (rec > s.rec_cap || gat > s.gat_cap) {
    // rec is READ and gathered WRITTEN by the other devices (peer copies on their streams; RCCL's kernels over xGMI), and
    // hipFree waits for the owning device only: every slot's stream is drained before the two go (ADVICE r5 -- a step
    // issued after btrapz_multi_set_shards with more segments, the last step's gather still in flight on a peer)
    for (Shard &o : m->sh)
      if (o.stream) { MCHK(m, hipSetDevice(o.device)); MCHK(m, hipStreamSynchronize(o.stream)); }
    MCHK(m, hipSetDevice(s.device));
    (void)hipFree(s.rec); (void)hipFree(s.gathered); s.rec = nullptr; s.gathered = nullptr; s.rec_cap = 0; s.gat_cap = 0;
    MCHK(m, hipMalloc(&s.rec, sizeof(long long) * rec)); MCHK(m, hipMalloc(&s.gathered, sizeof(long long) * gat));
    s.rec_cap = rec; s.gat_cap = gat;
  }
  const size_t nb = nl > ng ? nl : ng;
  if (nb > s.best_cap) {
    (void)hipFree(s.best_cost); (void)hipFree(s.best_idx); s.best_cost = nullptr; s.best_idx = nullptr; s.best_cap = 0;
    MCHK(m, hipMalloc(&s.best_cost, sizeof(double) * nb)); MCHK(m, hipMalloc(&s.best_idx, sizeof(long long) * nb));
    s.best_cap = nb;
  }
  return grow(m, s.best_ctrl, s.best_ctrl_cap, nb * P);
}

static int set_shape(btrapz_multi *m, int B, int S, int group) {
  if (B < 1 || S < 1 || S > BTRAPZ_MAX_SEGMENTS_LONG || group < 0 || (group > 0 && B % group != 0)) { m->err = "invalid argument"; return BTRAPZ_EINVAL; }
  m->B = B; m->S = S;
  m->global_groups = group == 0 || group >= B;
  m->group = m->global_groups ? B : group;
  m->n_groups = m->global_groups ? 1 : B / group;
  m->solved = false;
  return BTRAPZ_OK;
}

BTRAPZ_EXPORT int btrapz_multi_upload(btrapz_multi *m, int B, int S, int group, const double *seg, const double *init,
                                      const double *ref_end, const double *dl_bounds) {
  if (!m) return BTRAPZ_EINVAL;
  DeviceGuard guard_;
  if (!seg || !init || !ref_end || !dl_bounds) { m->err = "invalid argument"; return BTRAPZ_EINVAL; }
  int rc = set_shape(m, B, S, group);
  if (rc != BTRAPZ_OK) return rc;
  for (int g = 0; g < m->G; g++) {
    Shard &s = m->sh[g];
    int lo, hi;
    shard_bounds(B, m->G, g, m->global_groups ? 1 : m->group, &lo, &hi);
    s.B = hi - lo; s.base = lo;
    MCHK(m, hipSetDevice(s.device));
    const size_t nb = (size_t)(s.B > 0 ? s.B : 1), n_seg = (size_t)BTRAPZ_NUM_SEG_FIELDS * nb * S;
    rc = grow(m, s.own_in, s.own_in_cap, n_seg + nb * 18);
    if (rc != BTRAPZ_OK) return rc;
    double *d_seg = s.own_in, *d_init = d_seg + n_seg, *d_re = d_init + nb * 6, *d_dl = d_re + nb * 2;
    if (s.B > 0) {
      // (a context's launches of an earlier step may still read these buffers: stream order covers it)
      for (int f = 0; f < BTRAPZ_NUM_SEG_FIELDS; f++)
        MCHK(m, hipMemcpyAsync(d_seg + (size_t)f * s.B * S, seg + ((size_t)f * B + lo) * S, sizeof(double) * (size_t)s.B * S, hipMemcpyHostToDevice, s.stream));
      MCHK(m, hipMemcpyAsync(d_init, init + (size_t)lo * 6, sizeof(double) * 6 * s.B, hipMemcpyHostToDevice, s.stream));
      MCHK(m, hipMemcpyAsync(d_re, ref_end + (size_t)lo * 2, sizeof(double) * 2 * s.B, hipMemcpyHostToDevice, s.stream));
      MCHK(m, hipMemcpyAsync(d_dl, dl_bounds + (size_t)lo * 10, sizeof(double) * 10 * s.B, hipMemcpyHostToDevice, s.stream));
    }
    s.seg = d_seg; s.init = d_init; s.ref_end = d_re; s.dl_bounds = d_dl;
  }
  for (Shard &s : m->sh) { MCHK(m, hipSetDevice(s.device)); MCHK(m, hipStreamSynchronize(s.stream)); }   // the host arrays may go
  return BTRAPZ_OK;
}

BTRAPZ_EXPORT int btrapz_multi_set_shards(btrapz_multi *m, int B, int S, int group, const btrapz_multi_shard *shards) {
  if (!m) return BTRAPZ_EINVAL;
  if (!shards) { m->err = "invalid argument"; return BTRAPZ_EINVAL; }
  int rc = set_shape(m, B, S, group);
  if (rc != BTRAPZ_OK) return rc;
  long long next = 0;
  for (int g = 0; g < m->G; g++) {
    const btrapz_multi_shard &in = shards[g];
    if (in.B < 0 || in.index_base != next || (in.B > 0 && (!in.seg || !in.init || !in.ref_end || !in.dl_bounds)) ||
        (!m->global_groups && in.B % m->group != 0)) {
      m->err = "invalid argument: shards must be contiguous, in device order, and whole arg-min groups"; return BTRAPZ_EINVAL;
    }
    Shard &s = m->sh[g];
    s.B = in.B; s.base = in.index_base; s.seg = in.seg; s.init = in.init; s.ref_end = in.ref_end; s.dl_bounds = in.dl_bounds;
    next += in.B;
  }
  if (next != B) { m->err = "invalid argument: the shards do not add up to B"; return BTRAPZ_EINVAL; }
  return BTRAPZ_OK;
}

BTRAPZ_EXPORT int btrapz_multi_solve_argmin(btrapz_multi *m, const btrapz_shared *shared, const btrapz_options *opt) {
  if (!m) return BTRAPZ_EINVAL;
  DeviceGuard guard_;
  if (!shared || m->B < 1) { m->err = "invalid argument (no batch: call btrapz_multi_upload or btrapz_multi_set_shards first)"; return BTRAPZ_EINVAL; }
  const int G = m->G, S = m->S, P = 12 * S;
  const int n_gather = m->global_groups ? 1 : 0;
  // 1. every device: solve its shard, arg-min over its groups, pack the record of its winner.  All launches are
  //    asynchronous: the host thread walks the devices once and they run side by side.
  for (int g = 0; g < G; g++) {
    Shard &s = m->sh[g];
    const int n_local = m->global_groups ? 1 : s.B / m->group;
    int rc = ensure_outputs(m, s, S, n_local, n_gather);
    if (rc != BTRAPZ_OK) return rc;
    MCHK(m, hipSetDevice(s.device));
    if (m->global_groups)   // this step's pack overwrites the record the other devices copied in the last step
      for (int d = 0; d < G; d++)
        if (d != g && m->sh[d].done_recorded) MCHK(m, hipStreamWaitEvent(s.stream, m->sh[d].done, 0));
    if (s.B > 0) {
      rc = btrapz_solve_batch_device(s.ctx, shared, opt, s.B, S, s.seg, s.init, s.ref_end, s.dl_bounds, s.ctrl, s.cost, s.status, s.iters, s.stream);
      if (rc != BTRAPZ_OK) { m->err = std::string("device ") + std::to_string(s.device) + ": " + btrapz_last_error(s.ctx); return rc; }
      const int grp = m->global_groups ? s.B : m->group;
      rc = btrapz_argmin_device(s.ctx, s.B, grp, s.base, s.cost, m->global_groups ? s.loc_idx : s.best_idx, m->global_groups ? s.loc_cost : s.best_cost, s.stream);
      if (rc != BTRAPZ_OK) { m->err = std::string("device ") + std::to_string(s.device) + ": " + btrapz_last_error(s.ctx); return rc; }
    }
    if (m->global_groups) {
      if (s.B > 0) {
        hipLaunchKernelGGL(multi_pack_kernel, dim3(1), dim3(256), 0, s.stream, 1, P, s.base, (const long long *)s.loc_idx, (const double *)s.loc_cost,
                           (const double *)s.ctrl, s.rec);
      } else {   // an empty shard takes part in the gather with "nobody": +inf, -1
        static const long long none[2] = {0x7ff0000000000000LL, -1};   // (static: an asynchronous copy may read it after this call returns)
        MCHK(m, hipMemsetAsync(s.rec, 0, sizeof(long long) * (2 + (size_t)P), s.stream));
        MCHK(m, hipMemcpyAsync(s.rec, none, sizeof(none), hipMemcpyHostToDevice, s.stream));
      }
      MCHK(m, hipGetLastError());
      MCHK(m, hipEventRecord(s.packed, s.stream));
    } else if (s.B > 0) {
      // groups that live on this device: the local winner IS the winner (BASELINE config 5 sharded by agent) -- its
      // control points through the same pack / select pair, world = 1, no collective
      const int n = s.B / m->group;
      hipLaunchKernelGGL(multi_pack_kernel, dim3(n), dim3(256), 0, s.stream, n, P, s.base, (const long long *)s.best_idx, (const double *)s.best_cost,
                         (const double *)s.ctrl, s.gathered);
      hipLaunchKernelGGL(multi_select_kernel, dim3(n), dim3(256), 0, s.stream, 1, n, P, (const long long *)s.gathered, s.best_cost, s.best_idx, s.best_ctrl);
      MCHK(m, hipGetLastError());
    }
  }
  // 2. the one gather, 3. the same lexicographic min on every device
  if (m->global_groups) {
    const size_t R = 2 + (size_t)P;
    if (m->transport == BTRAPZ_MULTI_RCCL) {
      NCHK(m, m->rccl.GroupStart());
      for (int g = 0; g < G; g++) {
        Shard &s = m->sh[g];
        const ncclResult_t r = m->rccl.AllGather(s.rec, s.gathered, R, ncclInt64, s.comm, s.stream);
        if (r != ncclSuccess) { (void)m->rccl.GroupEnd(); m->err = std::string("ncclAllGather: ") + m->rccl.GetErrorString(r); return BTRAPZ_EHIP; }
      }
      NCHK(m, m->rccl.GroupEnd());
    } else {
      for (int d = 0; d < G; d++) {
        Shard &dst = m->sh[d];
        MCHK(m, hipSetDevice(dst.device));
        for (int g = 0; g < G; g++) {
          Shard &src = m->sh[g];
          if (g != d) MCHK(m, hipStreamWaitEvent(dst.stream, src.packed, 0));
          if (src.device == dst.device)
            MCHK(m, hipMemcpyAsync(dst.gathered + (size_t)g * R, src.rec, sizeof(long long) * R, hipMemcpyDeviceToDevice, dst.stream));
          else
            MCHK(m, hipMemcpyPeerAsync(dst.gathered + (size_t)g * R, dst.device, src.rec, src.device, sizeof(long long) * R, dst.stream));
        }
      }
    }
    for (int d = 0; d < G; d++) {
      Shard &s = m->sh[d];
      MCHK(m, hipSetDevice(s.device));
      hipLaunchKernelGGL(multi_select_kernel, dim3(1), dim3(256), 0, s.stream, G, 1, P, (const long long *)s.gathered, s.best_cost, s.best_idx, s.best_ctrl);
      MCHK(m, hipGetLastError());
      MCHK(m, hipEventRecord(s.done, s.stream));
      s.done_recorded = true;
    }
  }
  m->solved = true;
  return BTRAPZ_OK;
}

BTRAPZ_EXPORT int btrapz_multi_result(btrapz_multi *m, int device_slot, long long *best_idx, double *best_cost, double *best_ctrl) {
  if (!m) return BTRAPZ_EINVAL;
  DeviceGuard guard_;
  if (!m->solved || device_slot < -1 || device_slot >= m->G) { m->err = "invalid argument (or no step issued yet)"; return BTRAPZ_EINVAL; }
  const size_t P = 12 * (size_t)m->S;
  if (m->global_groups) {
    // every device holds the same winner; device_slot = -1 waits for ALL of them (the step is over), else for that one
    for (int g = 0; g < m->G; g++)
      if (device_slot < 0 || g == device_slot) { MCHK(m, hipSetDevice(m->sh[g].device)); MCHK(m, hipStreamSynchronize(m->sh[g].stream)); }
    Shard &s = m->sh[device_slot < 0 ? 0 : device_slot];
    MCHK(m, hipSetDevice(s.device));
    if (best_idx) MCHK(m, hipMemcpy(best_idx, s.best_idx, sizeof(long long), hipMemcpyDeviceToHost));
    if (best_cost) MCHK(m, hipMemcpy(best_cost, s.best_cost, sizeof(double), hipMemcpyDeviceToHost));
    if (best_ctrl) MCHK(m, hipMemcpy(best_ctrl, s.best_ctrl, sizeof(double) * P, hipMemcpyDeviceToHost));
    return BTRAPZ_OK;
  }
  // groups per device: the winners of all groups, in group order
  for (int g = 0; g < m->G; g++) {
    Shard &s = m->sh[g];
    if (s.B <= 0) continue;
    const size_t n = (size_t)(s.B / m->group), first = (size_t)(s.base / m->group);
    MCHK(m, hipSetDevice(s.device));
    MCHK(m, hipStreamSynchronize(s.stream));
    if (best_idx) MCHK(m, hipMemcpy(best_idx + first, s.best_idx, sizeof(long long) * n, hipMemcpyDeviceToHost));
    if (best_cost) MCHK(m, hipMemcpy(best_cost + first, s.best_cost, sizeof(double) * n, hipMemcpyDeviceToHost));
    if (best_ctrl) MCHK(m, hipMemcpy(best_ctrl + first * P, s.best_ctrl, sizeof(double) * n * P, hipMemcpyDeviceToHost));
  }
  return BTRAPZ_OK;
}

BTRAPZ_EXPORT int btrapz_multi_wait(btrapz_multi *m) {
  if (!m) return BTRAPZ_EINVAL;
  DeviceGuard guard_;
  for (Shard &s : m->sh) { MCHK(m, hipSetDevice(s.device)); MCHK(m, hipStreamSynchronize(s.stream)); }
  return BTRAPZ_OK;
}

BTRAPZ_EXPORT int btrapz_multi_shard_view(btrapz_multi *m, int device_slot, btrapz_multi_view *v) {
  if (!m) return BTRAPZ_EINVAL;
  if (device_slot < 0 || device_slot >= m->G || !v) { m->err = "invalid argument"; return BTRAPZ_EINVAL; }
  const Shard &s = m->sh[device_slot];
  v->device = s.device; v->B = s.B; v->index_base = s.base; v->stream = (void *)s.stream; v->ctx = s.ctx;
  v->ctrl = s.ctrl; v->cost = s.cost; v->status = s.status; v->iters = s.iters;
  v->best_idx = s.best_idx; v->best_cost = s.best_cost; v->best_ctrl = s.best_ctrl;
  return BTRAPZ_OK;
}

BTRAPZ_EXPORT int btrapz_multi_download(btrapz_multi *m, double *ctrl, double *cost, int *status, int *iters) {
  if (!m) return BTRAPZ_EINVAL;
  DeviceGuard guard_;
  if (!m->solved) { m->err = "no step issued yet"; return BTRAPZ_EINVAL; }
  const size_t P = 12 * (size_t)m->S;
  for (Shard &s : m->sh) {
    if (s.B <= 0) continue;
    MCHK(m, hipSetDevice(s.device));
    MCHK(m, hipStreamSynchronize(s.stream));
    if (ctrl) MCHK(m, hipMemcpy(ctrl + (size_t)s.base * P, s.ctrl, sizeof(double) * P * s.B, hipMemcpyDeviceToHost));
    if (cost) MCHK(m, hipMemcpy(cost + s.base, s.cost, sizeof(double) * s.B, hipMemcpyDeviceToHost));
    if (status) MCHK(m, hipMemcpy(status + s.base, s.status, sizeof(int) * s.B, hipMemcpyDeviceToHost));
    if (iters) MCHK(m, hipMemcpy(iters + s.base, s.iters, sizeof(int) * s.B, hipMemcpyDeviceToHost));
  }
  return BTRAPZ_OK;
}
