// corridor_kernels.hip -- the corridor stage batched on the device and the
// bucketing that lets the QP kernel take candidates with different segment counts.
//
// corridor_batch_kernel: one wavefront per candidate, every phase across the lanes.  Per-knot bounds of every
// obstacle -> slopes (all lanes, loads issued in blocks) -> CorridorGeneration + CorridorSplit (ballot search for the
// next slope break, then one lane per base segment; extract_segments_wave) -> CollisionCheck: one lane per segment
// counts the reference knots inside it, a scan turns the reference's running counter into a per-segment decision
// -> de-dup (keys compared through readlane), stable rank sort, the "continues this lane" reorder (ballot search) and
// the neighbour-overlap walk, all on keys held in the lanes (only the cuboid variant's all-pairs overlap pass runs
// serially) -> the batch record of the QP kernel (one lane per selected segment, coalesced field-major stores).
// The serial statements of corridor_core.h (what the host driver runs) remain the fallback for the shapes the
// wave-wide code does not take (slope table larger than 24 KB, more than 64 base segments) and give the same
// segments bit for bit.  This is the one stage of the path that streams HBM: num_obs * N * 4 doubles per candidate
// (11 KB at N = 71, 5 obstacles); it runs on instruction issue and dependent latency (~1 700 vector and ~1 300 scalar
// instructions per candidate, profiles/r03_corridor_pmc.json), so resident wavefronts pay: hence the LDS overlays, the
// first-pass instantiations without the serial statement (52-60 registers) and the list sizes the host chooses.
// The prism_ instantiations evaluate the bounds from the scene's obstacle prisms instead of reading them
// (btrapz_prism_corridor_batch_device: prism_bounds_kernel + this kernel in one launch, same bits).
// References: src/solve_3d.cc:323-486,488-714,729-772,835-845,1159-1166 ; src/cuboid_3d.cc:301-573.
#include <hip/hip_runtime.h>

#include <cstring>

#include "btrapz_device.h"
#include "corridor_core.h"
#include "prism_core.h"

namespace btrapz {

#define UNIFORM_BLOCK_C asm volatile("")
#ifdef CABL_MARKS
#define CABL_MARK(i_, x) asm volatile("; ==PHASE " x)
#elif defined(CABL_TIMING)
// -DCABL_TIMING (a measuring build, tools/corridor_bench.py --timing): the wall-clock cycles a wavefront spends between two
// phase marks, summed over all wavefronts in a device array the host reads through btrapz_debug_corridor_timing()
// (1 024 sets of slots, a wavefront adds to set blockIdx & 1023: 65 536 atomics on ONE address cost more than the kernel)
__device__ unsigned long long g_corridor_timing[1024][16];
extern "C" __attribute__((visibility("default"))) int btrapz_debug_corridor_timing(unsigned long long *out, int reset) {
  static unsigned long long h[1024][16];
  if (out) {
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_corridor_timing), sizeof h) != hipSuccess) return -1;
    for (int q = 0; q < 16; q++) { out[q] = 0; for (int i = 0; i < 1024; i++) out[q] += h[i][q]; }
  }
  if (reset) { memset(h, 0, sizeof h); if (hipMemcpyToSymbol(HIP_SYMBOL(g_corridor_timing), h, sizeof h) != hipSuccess) return -1; }
  return 0;
}
#define CABL_PHASES "SLOPES REFS BREAKS REFSTORE SELECT DEDUP RANK REORDER OVERLAP RECORD END"
// (the deltas stay in the wavefront's LDS until the last mark: an atomic per mark would put its own latency into the next
//  phase; LDS, not registers, so that a mark can sit inside any of the kernel's functions)
__device__ __forceinline__ void cabl_mark(int i) {
  __shared__ unsigned long long sh[18];
  const unsigned long long t = __builtin_readcyclecounter();
  if (threadIdx.x == 0) {
    if (i >= 0) sh[i] = t - sh[17];
    sh[17] = __builtin_readcyclecounter();
    if (i == 10) for (int q = 0; q < 16; q++) atomicAdd(&g_corridor_timing[blockIdx.x & 1023][q], q < 11 || (q >= 11 && q <= 14) ? sh[q] : 0ull);
  }
}
#define CABL_MARK(i_, name_) cabl_mark(i_)
#else
#define CABL_MARK(i_, x)
#endif
enum { MAX_ALL = 160, MAX_SEL = 64 };   // capacities of the retry pass (see btrapz_corridor_batch_device)

// LDS is what limits the wavefronts per CU here, and the serial phases of this kernel live on latency, so the
// segment lists are sized per launch: a first pass with room for the usual case (cap_o segments per obstacle,
// cap_sel selected ones: 14 KB, 11 wavefronts per CU at N = 71 with 3 obstacles) and, for the candidates that
// overflow it, a retry pass with the full MAX_ALL / MAX_SEL (29 KB, 5 per CU).  Same code, same results.
// Dynamic LDS: Seg all[O * cap_o] (later: Seg sel[cap_sel], same storage) | slopes[O][N][2] (when `staged`; later, same
//              storage: s_ref[N] | l_ref[N] | ds_bounds[N][2]) |
//              int hits[O * cap_o] | int ocount[64] | int key[cap_sel] | short slot_of[O * cap_o] | short pick[cap_sel]
// RB: blocks of 64 knots of the reference and the ds bounds held in registers across the extraction (2 serve N <= 128).
// PRISMS: the per-knot bounds are not read but evaluated from the scene's obstacle prisms (prism_core.h), whose tables
// sit behind pick[] -- the fused form of prism_bounds_kernel + this kernel, same segments bit for bit.
// SERIAL: the instantiation carries the serial statement of the extraction for the shapes the wave-wide code does not
// take.  The first pass of a two-pass launch runs without it (a candidate that needs it goes to the retry list): the
// statement is a quarter of the kernel's code and its registers are what the hot path then does not spill.
template <int RB, bool PRISMS, bool SERIAL>
__device__ __forceinline__ void corridor_candidate(const CorridorArgs &a, int staged, int b, unsigned char *lds_raw);

// Registers against wavefronts per SIMD.  The kernel issues ~1 900 vector and ~1 400 scalar instructions per candidate
// and kept the vector unit 68 % busy at four wavefronts per SIMD (profiles/r03_corridor_pmc_before.json): it runs on
// issue slots and dependent latency, so more resident wavefronts pay as long as they do not spill.  The first-pass
// instantiations (no serial statement) need 52-60 registers: LDS alone bounds them, and the host sizes the lists for
// that (btrapz_host.hip, launch_corridor_stage).  The instantiations with the serial statement -- retry pass, and the
// only pass for shapes without a staged slope table -- are held to four wavefronts per SIMD (128 registers; measured
// with the serial statement in the hot path: five 0.272 ms against 0.305 at N = 71, but 132-176 B of scratch).
#ifndef CABL_WAVES
#define CABL_WAVES 4
#endif
#ifndef CABL_WAVES_SHORT
#define CABL_WAVES_SHORT 4
#endif
// (the upper bound 8 holds the first-pass instantiations to 64 registers although LDS admits five wavefronts per SIMD; with
//  5 or 6 -- 102 / 85 registers to schedule in -- nothing moves: 0.231 -> 0.233 / 0.230 ms at N = 71, round 6)
template <int RB, bool PRISMS, bool SERIAL>
__device__ __forceinline__ void corridor_batch_body(const CorridorArgs &a, int staged) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  if (!SERIAL || a.pass == 0) {
    corridor_candidate<RB, PRISMS, SERIAL>(a, staged, (int)blockIdx.x, lds_raw);
  } else {  // retry pass: the candidates the first pass could not hold
    const int n = *a.retry_count;
    // (round 6: two counters used alternately, this launch zeroing the next call's, so that a call needs no memset of its
    //  own: no gain -- 0.2145 -> 0.217 ms, the memset runs behind the previous call's tail anyway)
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
      // (the candidate's number is the same in every lane: say so, or every address derived from it costs vector registers)
      corridor_candidate<RB, PRISMS, SERIAL>(a, staged, __builtin_amdgcn_readfirstlane(a.retry_list[i]), lds_raw);
      __syncthreads();
    }
  }
}
#define CORRIDOR_KERNEL(name, waves, RB, PRISMS, SERIAL)                                                                   \
  __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(waves, 8))) void name(const CorridorArgs a, int staged) { \
    corridor_batch_body<RB, PRISMS, SERIAL>(a, staged);                                                                       \
  }
// (short: horizons of at most 128 knots -- the bundled scenes have 71-121 --, half the prefetch registers; prism_: the
//  fused form, btrapz_prism_corridor_batch_device; first_: first pass of a two-pass launch, without the serial statement)
CORRIDOR_KERNEL(corridor_batch_kernel, CABL_WAVES, 4, false, true)
CORRIDOR_KERNEL(corridor_batch_short_kernel, CABL_WAVES_SHORT, 2, false, true)
CORRIDOR_KERNEL(corridor_first_kernel, CABL_WAVES, 4, false, false)
CORRIDOR_KERNEL(corridor_first_short_kernel, CABL_WAVES_SHORT, 2, false, false)
CORRIDOR_KERNEL(prism_corridor_batch_kernel, CABL_WAVES, 4, true, true)
CORRIDOR_KERNEL(prism_corridor_batch_short_kernel, CABL_WAVES_SHORT, 2, true, true)
CORRIDOR_KERNEL(prism_corridor_first_kernel, CABL_WAVES, 4, true, false)
CORRIDOR_KERNEL(prism_corridor_first_short_kernel, CABL_WAVES_SHORT, 2, true, false)

// Where the per-knot bounds of obstacle corridor o come from: memory ([num_obs][N][2] of the candidate) or the prisms.
struct MemoryBounds {
  const double *gs, *gl;
  int N;
  __device__ __forceinline__ double2 s2(int o, int i) const { const double *p = gs + ((size_t)o * N + i) * 2; return make_double2(p[0], p[1]); }
  __device__ __forceinline__ double2 l2(int o, int i) const { const double *p = gl + ((size_t)o * N + i) * 2; return make_double2(p[0], p[1]); }
};
struct PrismBounds {
  PrismTab t;
  PrismRoad r;
  __device__ __forceinline__ double2 s2(int o, int i) const { return prism_pair_s(t, r, o, i); }
  __device__ __forceinline__ double2 l2(int o, int /*i*/) const { return prism_pair_l(t, o); }
};
struct PrismViewS { const PrismBounds *p; int o; __device__ double lo(int i) const { return p->s2(o, i).x; } __device__ double hi(int i) const { return p->s2(o, i).y; } };
struct PrismViewL { const PrismBounds *p; int o; __device__ double lo(int i) const { return p->l2(o, i).x; } __device__ double hi(int i) const { return p->l2(o, i).y; } };

__device__ __forceinline__ double readlane_f64(double v, int l) {  // l: the same in every lane
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// Inclusive prefix sum over the 64 lanes on the DPP network (no LDS round trips): shifts by 1, 2, 4, 8 inside each
// row of 16 (lanes without a source add 0), then the last lane of row 0 / 2 is added to row 1 / 3 (row_bcast:15) and
// the last lane of row 1 to rows 2 and 3 (row_bcast:31).
__device__ __forceinline__ int wave_inclusive_scan(int v, int /*lane*/) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
  return v;
}

// CorridorGeneration + CorridorSplit of every obstacle with the whole wavefront (extract_segments_core, the serial
// statement, gives the same segments bit for bit and is what the host driver and the fallback below run):
//  1. breaks: a segment ends at the first knot whose slopes differ from the OPEN segment's by more than 0.2 -- a
//     recurrence over the knots, but the open segment's slopes only change at a break.  64 knots are compared at a
//     time against the current pair; ballot + find-first gives the next break, the pair becomes that knot's successor
//     (readlane) and the lanes behind it are compared again: N/64 LDS reads + one ballot per break and obstacle
//     instead of N steps on one lane.
//  2. one lane per (obstacle, base segment): bounds at its first knot (independent loads: one memory round trip for
//     all of them instead of one per segment), duration, number of 1-second pieces; a scan places the pieces.
//  3. every lane writes the pieces of its base segment (the bias recurrence `bias + 1.0 * skew` step by step).
// brk (int[O * cap_o]) and first (short[O]) are scratch.  find_breaks_wave is step 1 and returns false when there are
// more than 64 base segments in total (the caller then runs the serial statement); build_segments_wave is steps 2-3.
__device__ __forceinline__ bool find_breaks_wave(const CorridorArgs &a, int lane, const double *slopes, int cap_o, int *brk,
                                                 int &my_nb) {
  const int N = a.N, O = a.num_obs;
  const double2 *sk2 = reinterpret_cast<const double2 *>(slopes);
  const double threshold = 0.2;  // solve_3d.cc:372
  my_nb = 0;                     // lane o: base segments of obstacle o, -1 when they exceed cap_o
  for (int o = 0; o < O; o++) {
    const double2 *sk = sk2 + (size_t)o * N;
    double2 cur = sk[1];
    int nb = 1;
    if (lane == 0) brk[o * cap_o] = 0;
    for (int start = 2; start < N - 1 && nb > 0; start += 64) {
      // the window's slopes and their successors are read once; the breaks inside it are found in registers
      const int i = start + lane;
      const bool in = i < N - 1;
      const double2 v = sk[in ? i : N - 2], nx = sk[in ? i + 1 : N - 1];
      unsigned long long ahead = ~0ull;          // lanes behind the last break
      for (;;) {
        const unsigned long long m = __ballot(in && (fabs(v.x - cur.x) > threshold || fabs(v.y - cur.y) > threshold)) & ahead;
        if (m == 0) break;
        const int l = __ffsll((long long)m) - 1;
        if (nb + 1 > cap_o) { nb = -1; break; }
        if (lane == 0) brk[o * cap_o + nb] = start + l;
        nb++;
        cur = make_double2(readlane_f64(nx.x, l), readlane_f64(nx.y, l));   // the slopes at knot start + l + 1
        ahead = l == 63 ? 0ull : ~0ull << (l + 1);
      }
    }
    if (lane == o) my_nb = nb;
  }
  __syncthreads();
  return __builtin_amdgcn_readlane(wave_inclusive_scan(my_nb > 0 ? my_nb : 0, lane), 63) <= 64;
}
// after_slopes(): called by every lane once the slope table has been read for the last time (the caller stores the
// prefetched reference over it there: held in registers any longer, the allocator spills the prefetch to scratch)
template <class Src, class F>
__device__ __forceinline__ void build_segments_wave(const CorridorArgs &a, int lane, const Src &src,
                                                    const double *slopes, Seg *all, int cap_o, int *ocount, int *brk_w,
                                                    short *first, int my_nb, F &&after_slopes) {
  const int *brk = brk_w;
  const int N = a.N, O = a.num_obs;
  const double2 *sk2 = reinterpret_cast<const double2 *>(slopes);
  const int nbp = my_nb > 0 ? my_nb : 0;
  const int incl = wave_inclusive_scan(nbp, lane);
  if (lane < O) ocount[lane] = my_nb < 0 ? -1 : 0;
  int mo = -1, mk = 0, mn = 0;
  for (int o = 0; o < O; o++) {
    const int n = __builtin_amdgcn_readlane(nbp, o), q0 = __builtin_amdgcn_readlane(incl, o) - n;
    if (lane >= q0 && lane < q0 + n) { mo = o; mk = lane - q0; mn = n; }
  }
  Seg s = seg_default();
  int h = 0;
  bool bad = false;
  double2 slope = make_double2(0.0, 0.0);
  if (mo >= 0) slope = sk2[(size_t)mo * N + brk[mo * cap_o + mk] + 1];
  CABL_MARK(12, "SEG_MAPPED");
  after_slopes();
  CABL_MARK(13, "SEG_REFS_STORED");
  if (mo >= 0) {
    const int beg = brk[mo * cap_o + mk];
    const double2 sb = src.s2(mo, beg), lb = src.l2(mo, beg);
    s.beg_t = beg;
    s.end_t = mk + 1 < mn ? brk[mo * cap_o + mk + 1] : N - 1;
    s.down_skew = slope.x; s.down_bias = sb.x;
    s.upp_skew = slope.y; s.upp_bias = sb.y;
    s.beg_l = lb.x; s.end_l = lb.y;
    if (a.variant == 0) {  // forward difference for the first segment, backward for the later ones (solve_3d.cc:338-341,358-367)
      const int i1 = beg == 0 ? 1 : beg, i0 = i1 - 1;
      const double2 l1 = src.l2(mo, i1), l0 = src.l2(mo, i0);
      s.l_down_bias = s.beg_l; s.l_upp_bias = s.end_l;
      s.l_down_skew = (l1.x - l0.x) / a.delta; s.l_upp_skew = (l1.y - l0.y) / a.delta;
    }
    s.t = (s.end_t - s.beg_t) * a.delta;
    double t = s.t;
    while (t > 1) { t = t - 1; if (++h > cap_o) { bad = true; break; } }
  }
  const int cnt = mo >= 0 ? h + 1 : 0;
  const int upto = wave_inclusive_scan(cnt, lane);
  if (mo >= 0 && mk == 0) first[mo] = (short)(upto - cnt);
  __syncthreads();
  const int pos = mo >= 0 ? upto - cnt - first[mo] : 0;
  if (bad) ocount[mo] = -1;
  __syncthreads();
  if (mo >= 0 && mk == mn - 1 && ocount[mo] == 0) ocount[mo] = pos + cnt > cap_o ? -1 : pos + cnt;
  __syncthreads();
  // CorridorSplit, solve_3d.cc:735-746 -- one lane per PIECE (round 6; until then the lane of a base segment wrote its pieces one
  // after the other: seven times thirteen LDS stores and a default-initialised Seg each on ONE lane for an obstacle corridor
  // without slope breaks, the other lanes waiting).  A base lane leaves its segment in the slot of its last piece and marks its
  // slots (piece number, distance to that slot) in brk[] -- free now: every break has been read --; then lane l builds the piece
  // of slot base + l from the base segment it finds there: the reference's recurrences step by step (the bias after j pieces is
  // j additions of the slope, the rest's duration cnt - 1 subtractions of 1), so the pieces are the serial statement's bit for bit.
  CABL_MARK(14, "SEG_BASES");
  const int cap_all = cap_o * O;
  const bool okb = mo >= 0 && ocount[mo] > 0;
  const int slot0 = okb ? mo * cap_o + pos : 0;
  for (int i = lane; i < cap_all; i += 64) brk_w[i] = -1;
  __syncthreads();
  if (okb) {
    all[slot0 + cnt - 1] = s;
    for (int j = 0; j < cnt; j++) brk_w[slot0 + j] = j | ((cnt - 1 - j) << 8);
  }
  __syncthreads();
  for (int base = 0; base < cap_all; base += 64) {
    const int slot = base + lane;
    const int code = slot < cap_all ? brk_w[slot] : -1;
    Seg piece = seg_default();
    if (code >= 0) {
      const int j = code & 255, rem = code >> 8;
      const Seg b = all[slot + rem];
      double db = b.down_bias, ub = b.upp_bias, t = b.t;
      for (int u = 0; u < j; u++) { db = db + 1.0 * b.down_skew; ub = ub + 1.0 * b.upp_skew; t = t - 1; }
      if (rem == 0) {          // what is left of the base segment behind its pieces
        piece = b;
        piece.beg_t = b.beg_t + 10 * j; piece.t = t; piece.down_bias = db; piece.upp_bias = ub;
      } else {
        piece.beg_t = b.beg_t + 10 * j; piece.end_t = piece.beg_t + 10; piece.t = 1.0;
        piece.down_skew = b.down_skew; piece.down_bias = db;
        piece.upp_skew = b.upp_skew; piece.upp_bias = ub;
        if (a.variant == 0) {
          piece.l_down_skew = b.l_down_skew; piece.l_down_bias = b.l_down_bias;
          piece.l_upp_skew = b.l_upp_skew; piece.l_upp_bias = b.l_upp_bias;
        }
        piece.beg_l = b.beg_l; piece.end_l = b.end_l;
      }
    }
    __syncthreads();            // every lane has read its base segment: the slots of the rests may be overwritten
    if (code >= 0) all[slot] = piece;
  }
}

__device__ __forceinline__ double from_lane_below(double v) {   // lane l gets lane l - 1's value (lane 0: 0)
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, false),    // wave_shr:1
                          __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, false));
}

template <int RB, bool PRISMS, bool SERIAL>
__device__ __forceinline__ void corridor_candidate(const CorridorArgs &a, int staged, int b, unsigned char *lds_raw) {
  const int lane = threadIdx.x;
#ifdef CABL_TIMING
  cabl_mark(-1);   // slot i: from the mark before it (or this start) to mark i
#endif
  const int N = a.N, O = a.num_obs;
  const int cap_o = a.cap_o, cap_all = cap_o * O, cap_sel = a.cap_sel;
  Seg *all = reinterpret_cast<Seg *>(lds_raw);           // segments of every obstacle, obstacle o at [o * cap_o, ...)
  Seg *sel = all;                                        // the selected, ordered corridor: written (from registers)
                                                         // when all[] has been read for the last time, over it
  // (16-byte aligned: the slopes are written as double2)
  // (as an offset from lds_raw: a pointer that went through an integer is a generic pointer to the compiler, and every
  //  access through it a flat_ instruction -- all lists behind the segments were reached that way until round 3)
  double *dyn = reinterpret_cast<double *>(lds_raw + ((sizeof(Seg) * (size_t)(cap_all > cap_sel ? cap_all : cap_sel) + 15) & ~(size_t)15));
  // the slope table lives until the segments are built; the reference and the ds bounds (needed from the selection
  // on) are loaded meanwhile and stored over it
  double *slopes = dyn;
  double *sref = dyn, *lref = dyn + N;
  double *dsb = dyn + 2 * N;                             // ds bounds of the knots, (lower, upper) pairs
  const size_t shared_doubles = staged && O > 2 ? (size_t)O * N * 2 : (size_t)N * 4;
  int *hits = reinterpret_cast<int *>(dyn + shared_doubles);   // reference knots inside every segment
  int *ocount = hits + cap_all;
  int *key = ocount + 64;                                // beg_t of the survivors of the de-dup, for the rank sort
  short *slot_of = reinterpret_cast<short *>(key + cap_sel);  // flattened segment index -> slot in all[]
  short *pick = slot_of + cap_all;                       // slots of the selected segments, in selection order
  PrismBounds prism;
  if constexpr (PRISMS) {                                // the scene's tables: edges of the strips, faces of the cars
    const size_t used = reinterpret_cast<unsigned char *>(pick + cap_sel) - lds_raw;
    prism.t = prism_tab_at(lds_raw + ((used + 7) & ~(size_t)7), a.P);
    prism.r = a.road;
    const int strips = prism_tables(prism.t, prism.r, a.prisms + (size_t)b * a.P * 8, lane);
    if (lane == 0 && a.n_strips) a.n_strips[b] = strips <= O ? strips : -1;
  }

  // The streaming phase.  A wavefront that waits for each load before it issues the next one pays a memory round
  // trip per 64 knots: the loads are issued in blocks of four per array (indices clamped instead of predicated, so
  // that they stay in one basic block).
  // The extraction compares slopes: all lanes compute them (two divisions per knot) from coalesced 16-byte loads and
  // leave them in LDS; the bounds themselves are read again only where a segment starts.
  const double *gs = PRISMS ? nullptr : a.s_bounds + (size_t)b * O * N * 2, *gl = PRISMS ? nullptr : a.l_bounds + (size_t)b * O * N * 2;
  const MemoryBounds memory{gs, gl, N};
  const int n2 = O * N;  // (lower, upper) pairs
  // idx / N for idx < O * N <= 64 * 512 as a multiplication: exact while idx * N < 2^32 (an integer division costs ~25
  // instructions, and there is one per pair)
  const unsigned n_magic = 0xFFFFFFFFu / (unsigned)N + 1u;
  auto obstacle_of = [&](int idx) { return (int)__umulhi((unsigned)idx, n_magic); };
  const double2 *gs2 = reinterpret_cast<const double2 *>(gs);
  double2 *sk2 = reinterpret_cast<double2 *>(slopes);
  double cur_lo[4], cur_hi[4], prv_lo[4], prv_hi[4];
  double2 below = make_double2(0.0, 0.0);   // PRISMS: the pair in front of lane 0's (the previous block's last)
  auto load_pairs = [&](int base) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = base + u * 64 + lane, ic = i < n2 ? i : n2 - 1;
      if constexpr (PRISMS) {   // every pair is evaluated once; its predecessor comes from the lane below
        const int j = obstacle_of(ic);
        const double2 c2 = prism.s2(j, ic - j * N);
        double2 p2 = make_double2(from_lane_below(c2.x), from_lane_below(c2.y));
        if (lane == 0) p2 = below;
        below = make_double2(readlane_f64(c2.x, 63), readlane_f64(c2.y, 63));
        cur_lo[u] = c2.x; cur_hi[u] = c2.y; prv_lo[u] = p2.x; prv_hi[u] = p2.y;
      } else {
        const double2 c2 = gs2[ic], p2 = gs2[ic > 0 ? ic - 1 : 0];
        cur_lo[u] = c2.x; cur_hi[u] = c2.y; prv_lo[u] = p2.x; prv_hi[u] = p2.y;
      }
    }
  };
  auto store_slopes = [&](int base) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = base + u * 64 + lane;
      if (i < n2 && i - obstacle_of(i) * N > 0)  // (b(i) - b(i-1)) / delta: the expression of SlopesOnTheFly
        sk2[i] = make_double2((cur_lo[u] - prv_lo[u]) / a.delta, (cur_hi[u] - prv_hi[u]) / a.delta);
    }
  };
  CABL_MARK(0, "SLOPES");
  if (staged) {
    load_pairs(0);
    store_slopes(0);
    for (int base = 256; base < n2; base += 256) { load_pairs(base); store_slopes(base); }
  }
  // reference and ds bounds of the first 256 knots: issued behind the break search, in the same memory round trip
  // as the bounds at the segment starts
  CABL_MARK(1, "REFS");
  const double *gsr = a.s_ref + (size_t)b * N, *glr = a.l_ref + (size_t)b * N;
  const double2 *gds = reinterpret_cast<const double2 *>(a.ds_bounds + (size_t)b * N * 2);
  double ref_s[RB], ref_l[RB], ref_dlo[RB], ref_dhi[RB];   // (scalars: an array of double2 behind a lambda stays in scratch)
  auto load_refs = [&](int base) {
#pragma unroll
    for (int u = 0; u < RB; u++) {
      const int i = base + u * 64 + lane, ic = i < N ? i : N - 1;
      const double2 d = gds[ic];
      ref_s[u] = gsr[ic]; ref_l[u] = glr[ic]; ref_dlo[u] = d.x; ref_dhi[u] = d.y;
    }
  };
  // (round 6) the ten dl bounds the record carries: read HERE, with the streaming loads, not at the kernel's end -- a load
  // in front of the last store is a memory round trip added to every wavefront's lifetime
  double dl10_v = 0.0;
  if (lane < 10) {
    const int i = lane >> 1, ii = i > N - 1 ? N - 1 : i;
    dl10_v = a.dl_bounds[((size_t)b * N + ii) * 2 + (lane & 1)];
  }
  bool refs_finite = true;
  auto store_refs = [&](int base) {
    double2 *d2 = reinterpret_cast<double2 *>(dsb);
#pragma unroll
    for (int u = 0; u < RB; u++) {
      const int i = base + u * 64 + lane;
      if (i < N) {
        sref[i] = ref_s[u]; lref[i] = ref_l[u]; d2[i] = make_double2(ref_dlo[u], ref_dhi[u]);
        refs_finite = refs_finite && fabs(ref_s[u]) < 1e300 && fabs(ref_l[u]) < 1e300;
      }
    }
  };
  __syncthreads();
  bool refs_stored = false;
  // ---- per-obstacle extraction: lane o owns obstacle o ----
#ifdef CABL_NOEXTRACT
  if (lane < O) { ocount[lane] = 1; all[lane * cap_o] = seg_default(); all[lane * cap_o].end_t = N - 1; all[lane * cap_o].t = 1.0; }
  load_refs(0);
#else
  // hits / slot_of are free until the selection: scratch of the wave-wide extraction
  CABL_MARK(2, "BREAKS");
  int my_nb = 0;
  const bool wide = staged && find_breaks_wave(a, lane, slopes, cap_o, hits, my_nb);
  CABL_MARK(11, "BREAKS_FOUND");
  // (round 6: issued in FRONT of the break search instead -- so that its round trip runs behind the search's LDS chains --
  //  N = 71 x 3 obstacles 0.231 -> 0.228 ms, N = 201 x 2 0.560 -> 0.574: five resident wavefronts hide it either way)
  load_refs(0);
  if (wide) {
    auto after_slopes = [&]() {
      __syncthreads();                                     // every lane has its slopes: the table may be overwritten
      store_refs(0);
    };
    if constexpr (PRISMS) build_segments_wave(a, lane, prism, slopes, all, cap_o, ocount, hits, slot_of, my_nb, after_slopes);
    else build_segments_wave(a, lane, memory, slopes, all, cap_o, ocount, hits, slot_of, my_nb, after_slopes);
    refs_stored = true;
  } else if constexpr (!SERIAL) {   // not a shape for the wave-wide code: the retry pass has the serial statement
    if (lane == 0) a.retry_list[atomicAdd(a.retry_count, 1)] = b;
    return;
  } else {
    __syncthreads();
    if (lane < O) {  // the serial statement: lane o owns obstacle o
      if constexpr (PRISMS) {   // (always staged: the host takes the two-kernel path when the slope table does not fit)
        ocount[lane] = extract_segments_core(a.variant, N, a.delta, PrismViewS{&prism, lane}, PrismViewL{&prism, lane},
                                             SlopeTable{slopes + (size_t)lane * N * 2}, all + lane * cap_o, cap_o);
      } else {
        const BoundsView sb{gs + (size_t)lane * N * 2}, lb{gl + (size_t)lane * N * 2};
        if (staged)
          ocount[lane] = extract_segments_core(a.variant, N, a.delta, sb, lb, SlopeTable{slopes + (size_t)lane * N * 2}, all + lane * cap_o, cap_o);
        else
          ocount[lane] = extract_segments_core(a.variant, N, a.delta, sb, lb, SlopesOnTheFly{sb, a.delta}, all + lane * cap_o, cap_o);
      }
    }
  }
#endif
  CABL_MARK(3, "REFSTORE");
  __syncthreads();                                         // the slope table has been read for the last time
  if (!refs_stored) store_refs(0);
  for (int base = 64 * RB; base < N; base += 64 * RB) { load_refs(base); store_refs(base); }
  refs_finite = __all(refs_finite);
  __syncthreads();
  // ---- selection along the reference (solve_3d.cc:534-596): knots inside every segment, the lanes spread over
  // (segment, knot) pairs; the reference's running hit counter then reduces to a carry over the segments in order
  CABL_MARK(4, "SELECT");
  int total = 0;
  bool overflow = false;
  for (int o = 0; o < O; o++) {
    const int n = ocount[o];
    if (n < 0) { overflow = true; break; }
    for (int j = lane; j < n; j += 64) slot_of[total + j] = (short)(o * cap_o + j);
    total += n;
  }
#ifdef CABL_NOSELECT
  total = 0;
#endif
  __syncthreads();
  int nsel = 0;
  if (!overflow) {
    // One lane per segment, its fields in registers.  A knot before beg_t or after end_t cannot be inside: with
    // upp_bias > down_bias and the upper end above the lower end, the edge functions d0 and d2 of knot_inside
    // (whose first products are exact zeros for a finite reference and finite bounds: an infinite bias or slope makes
    // them NaN, hence the finite gaps) have opposite signs there -- so the lane only visits its own knots.  A segment
    // that fails those conditions, or a reference that is not finite, takes all.
    // The reference's running hit counter (selection_pushes): its value when it reaches a segment is the number of
    // hits before it, mod 3 -- a scan over the lanes instead of a walk over the segments.
    // (round 6) Up to 32 segments in all -- the usual case: 3 obstacles x 8 pieces -- TWO lanes per segment, lane l and lane
    // l + 32 each walking half of its knots: the walk is the phase's cost (11 knots of a one-second piece, four edge
    // functions each, for 24-30 busy lanes of 64), the counts are integers -- added through one cross-lane fetch.
    int carry = 0;
    const bool halves = total <= 32;                        // wave-uniform
    for (int q0 = 0; q0 < total; q0 += halves ? 32 : 64) {
      const int q = halves ? (lane & 31) : q0 + lane;
      int h = 0;
      bool equals_itself = true;  // false with a NaN among the fields same_segment compares
      if (q < total) {
        const Seg c = all[slot_of[q]];
        equals_itself = same_segment(c, c);
        const double gap0 = c.upp_bias - c.down_bias, gap1 = c.down_skew * a.delta + c.down_bias - c.upp_skew * a.delta - c.upp_bias;
        const bool own_range = refs_finite && gap0 > 0.0 && gap0 < 1e300 && gap1 < 0.0 && gap1 > -1e300 && c.beg_t <= c.end_t;
        int i_lo = own_range ? (c.beg_t > 0 ? c.beg_t : 0) : 0;
        int i_hi = own_range ? (c.end_t < N - 1 ? c.end_t : N - 1) : N - 1;
        if (halves) {
          const int mid_i = i_lo + ((i_hi - i_lo + 1) >> 1);   // [i_lo, mid_i) for lane l, [mid_i, i_hi] for lane l + 32
          if (lane < 32) i_hi = mid_i - 1; else i_lo = mid_i;
        }
        // six knots at most per lane (half of a one-second piece's eleven): a fixed-length, predicated walk whose twelve LDS
        // reads are issued together instead of one dependent pair per step (bit-identical; 0.243 -> 0.239 ms)
        // (and, every segment of the wavefront walking its own span only, without the two edge functions whose signs the
        //  span decides: knot_inside_own_span, corridor_core.h)
        if (__all(own_range && i_hi - i_lo < 6)) {
          UNIFORM_BLOCK_C;
#pragma unroll
          for (int u = 0; u < 6; u++) {
            const int i = i_lo + u, ic = i <= i_hi ? i : i_lo <= i_hi ? i_hi : (i_lo < N ? i_lo : N - 1);   // (a knot of the span, or any valid one for an empty half)
            const bool in = knot_inside_own_span(c, sref[ic], lref[ic], ic, a.delta);
            h += (i <= i_hi && in) ? 1 : 0;
          }
        } else
        for (int i = i_lo; i <= i_hi; i++) h += knot_inside(c, sref[i], lref[i], (double)i, a.delta) ? 1 : 0;
      }
      if (halves) {
        h += __shfl_xor(h, 32);
        if (lane >= 32) { h = 0; }
      }
      const bool owner = !halves || lane < 32;              // the lane that speaks for segment q below
      const int upto = wave_inclusive_scan(h, lane);
      int counter = (carry + upto - h) % 3;
      const int copies = (q < total && owner) ? selection_copies(selection_pushes(h, counter), equals_itself) : 0;   // 1, but for NaN segments
      const int placed = wave_inclusive_scan(copies, lane);
      for (int j = 0, r = nsel + placed - copies; j < copies && r < cap_sel; j++, r++) pick[r] = slot_of[q];
      nsel += __builtin_amdgcn_readlane(placed, 63);
      carry = (carry + __builtin_amdgcn_readlane(upto, 63)) % 3;
    }
    if (nsel > cap_sel) { nsel = cap_sel; overflow = true; }
  }
  __syncthreads();
  if (overflow && a.pass == 0 && a.retry_list && lane == 0) a.retry_list[atomicAdd(a.retry_count, 1)] = b;  // second chance
  CABL_MARK(5, "DEDUP");
  // ---- de-dup (keep first), stable sort by beg_t: lane j holds selected segment j ----
  int S = overflow ? -1 : 0;
  if (!overflow && nsel > 0) {
#ifdef CABL_NOORDER
    Seg mine = seg_default();
    if (lane < nsel) { mine = all[pick[lane]]; mine.count = 3; }
    __syncthreads();
    if (lane < nsel) sel[lane] = mine;
    S = nsel;
#else
    nsel = __builtin_amdgcn_readfirstlane(nsel);
    Seg mine = seg_default();
    bool keep = lane < nsel;
    if (keep) {
      mine = all[pick[lane]];
      mine.count = 3;                                      // the reference pushes a counted copy (solve_3d.cc:590)
    }
    // a later twin of segment i goes (same_segment; equality is transitive, so "a twin of any earlier one" is the
    // reference's "a twin of an earlier kept one"): the spans are compared first, the rest only when some lane matches.
    // (round 6) Twins share their first knot: every lane sets the bit of its beg_t in a bitmap (hits[] is free by now), and
    // unless SOME lane finds its bit already set there is no twin to look for -- the usual corridor, one segment per second.
    bool maybe_twins = true;
#ifndef CABL_NO_TWIN_BITMAP
    if (cap_all >= 16) {   // (16 words: first knots below 512 -- what the wave-wide kernels take)
      unsigned *bits = reinterpret_cast<unsigned *>(hits);
      if (lane < 16) bits[lane] = 0u;
      __syncthreads();
      unsigned seen = 0u;
      const bool valid_bt = keep && mine.beg_t >= 0 && mine.beg_t < 512;
      if (valid_bt) seen = atomicOr(&bits[mine.beg_t >> 5], 1u << (mine.beg_t & 31)) & (1u << (mine.beg_t & 31));
      maybe_twins = __ballot((keep && !valid_bt) || seen != 0u) != 0;
    }
#endif
    if (maybe_twins)
    for (int i = 0; i + 1 < nsel; i++) {
      const bool span = lane > i && lane < nsel && mine.beg_t == __builtin_amdgcn_readlane(mine.beg_t, i) &&
                        mine.end_t == __builtin_amdgcn_readlane(mine.end_t, i);
      if (__ballot(span) == 0) continue;
      if (span && mine.down_bias == readlane_f64(mine.down_bias, i) && mine.down_skew == readlane_f64(mine.down_skew, i) &&
          mine.upp_bias == readlane_f64(mine.upp_bias, i) && mine.upp_skew == readlane_f64(mine.upp_skew, i) &&
          mine.beg_l == readlane_f64(mine.beg_l, i) && mine.end_l == readlane_f64(mine.end_l, i))
        keep = false;
    }
  CABL_MARK(6, "RANK");
    const unsigned long long kept = __ballot(keep);
    const int pos = __popcll(kept & ((1ull << lane) - 1ull)), n = __popcll(kept);
    int rank = pos;
    // (round 6) nothing dropped and the first knots already in non-decreasing order along the lanes -- a corridor selected
    // obstacle by obstacle in the order the reference passes them --: the stable rank IS the lane
    const int bt_below = __builtin_amdgcn_update_dpp(0, mine.beg_t, 0x138, 0xf, 0xf, false);   // wave_shr:1: lane l - 1's
    const bool in_order = n == nsel && __ballot(lane > 0 && lane < nsel && bt_below > mine.beg_t) == 0;
    if (a.variant == 0 && !in_order) {  // stable rank by beg_t among the kept ones
      rank = 0;
      for (int i = 0; i < nsel; i++) {
        if (!((kept >> i) & 1ull)) continue;
        const int bt_i = __builtin_amdgcn_readlane(mine.beg_t, i);
        rank += (bt_i < mine.beg_t || (bt_i == mine.beg_t && i < lane)) ? 1 : 0;
      }
    }
    __syncthreads();                                       // every lane has its copy: all[] may be overwritten
    if (keep) sel[rank] = mine;
    __syncthreads();
    if (a.variant == 0) {
  CABL_MARK(7, "REORDER");
      // lane r holds the keys of position r
      double bl = 0.0;
      int bt = 0, et = 0, src = lane;
      if (lane < n) { bl = sel[lane].beg_l; bt = sel[lane].beg_t; et = sel[lane].end_t; }
      // reorder_segments_core: the search for "the first k > j that continues segment i" is a ballot; a hit swaps
      // the keys (and the source slots) of lanes j and k
      // (positions without a continuation further on -- the usual case -- cost one ballot: the j loop of the serial
      //  statement does nothing once no lane beyond j matches)
      // (round 6: the outer loop visits only the positions i whose neighbour i + 1 has ANOTHER beg_l -- the statement's
      //  `break` at j = i + 1 skips all the others --, found by one ballot over the pairs as they stand: a corridor that
      //  changes lanes once has one such position, not n - 2)
      for (int i_from = 0;;) {
        const double bl_nx = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(bl), 0x130, 0xf, 0xf, false),
                                              __builtin_amdgcn_update_dpp(0, __double2loint(bl), 0x130, 0xf, 0xf, false));   // wave_shl:1: lane l + 1's
        const unsigned long long need = __ballot(lane >= i_from && lane + 2 < n && !(bl == bl_nx));
        if (need == 0) break;
        const int i = __ffsll((long long)need) - 1;
        i_from = i + 1;
        const double bl_i = readlane_f64(bl, i);
        const int et_i = __builtin_amdgcn_readlane(et, i);
        for (int j = i + 1; j + 1 < n; j++) {
          const unsigned long long m = __ballot(lane > j && lane < n && bl == bl_i && bt == et_i);
          if (m == 0) break;
          const int k = __ffsll((long long)m) - 1;
          const int other = lane == j ? k : j;       // only lanes j and k take the exchanged values
          const double bl_x = __shfl(bl, other);
          const int bt_x = __shfl(bt, other), et_x = __shfl(et, other), src_x = __shfl(src, other);
          if (lane == j || lane == k) { bl = bl_x; bt = bt_x; et = et_x; src = src_x; }
        }
      }
      const bool moves = lane < n && src != lane;
      if (__any(moves)) {
        Seg moved = seg_default();
        if (moves) moved = sel[src];
        __syncthreads();
        if (moves) sel[lane] = moved;
      }
#ifndef CABL_NOOVERLAP
  CABL_MARK(8, "OVERLAP");
      // overlap_segments_core, trapezoid: every step sees the spans the previous one left -- a serial walk over the
      // neighbours, on the keys in the lanes; a span the walk assigned gets its duration recomputed, as there.
      // (round 6) The walk changes nothing unless SOME neighbouring pair meets one of its two conditions on the spans as
      // they are now -- a step that assigns nothing leaves the next one what it would have seen anyway -- so one ballot over
      // the pairs decides whether it runs at all (the usual corridor has no such pair).
      bool assigned = false;
      const int nbt = __builtin_amdgcn_update_dpp(0, bt, 0x130, 0xf, 0xf, false), net = __builtin_amdgcn_update_dpp(0, et, 0x130, 0xf, 0xf, false);   // wave_shl:1: lane l + 1's
      const bool pair_overlaps = lane + 1 < n && ((bt == nbt && et == net) || (bt > nbt && et <= net));
      if (__ballot(pair_overlaps) != 0)
      for (int i = 0; i + 1 < n; i++) {
        const int a_bt = __builtin_amdgcn_readlane(bt, i), b_et = __builtin_amdgcn_readlane(et, i + 1);
        int a_et = __builtin_amdgcn_readlane(et, i), b_bt = __builtin_amdgcn_readlane(bt, i + 1);
        bool ta = false, tb = false;
        if (a_bt == b_bt && a_et == b_et) {
          const int half = (a_et - a_bt) / 2;
          a_et -= half; b_bt += half; ta = tb = true;
        } else if (a_bt > b_bt && a_et <= b_et) {
          const int half = (a_et - a_bt) / 2;
          if (half > 1) { a_et -= half; ta = true; }
          b_bt = a_et; tb = true;
        }
        if (lane == i && ta) { et = a_et; assigned = true; }
        if (lane == i + 1 && tb) { bt = b_bt; assigned = true; }
      }
      if (lane < n && assigned) { sel[lane].beg_t = bt; sel[lane].end_t = et; sel[lane].t = (et - bt) * a.delta; }
#endif
    } else {
#ifndef CABL_NOOVERLAP
      if (lane == 0) overlap_segments_core(a.variant, a.delta, sel, n);   // cuboid: every pair of twins, serial
#endif
    }
    S = n;
#endif
  }
  __syncthreads();
  CABL_MARK(9, "RECORD");
  // ---- batch record: lane k writes segment k ----
  bool bad = S > a.seg_stride;
#ifdef CABL_NORECORD
  if (false) {
#else
  if (S > 0 && !bad && lane < S) {
#endif
    const Seg c = sel[lane];
    if (!(c.t > 0.0)) bad = true;
    const size_t BS = (size_t)a.B * a.seg_stride, e = (size_t)b * a.seg_stride + lane;
    double *sg = a.seg;
    sg[BTRAPZ_F_T * BS + e] = c.t;
    sg[BTRAPZ_F_DOWN_BIAS * BS + e] = c.down_bias; sg[BTRAPZ_F_DOWN_SKEW * BS + e] = c.down_skew;
    sg[BTRAPZ_F_UPP_BIAS * BS + e] = c.upp_bias; sg[BTRAPZ_F_UPP_SKEW * BS + e] = c.upp_skew;
    sg[BTRAPZ_F_L_DOWN_BIAS * BS + e] = c.l_down_bias; sg[BTRAPZ_F_L_DOWN_SKEW * BS + e] = c.l_down_skew;
    sg[BTRAPZ_F_L_UPP_BIAS * BS + e] = c.l_upp_bias; sg[BTRAPZ_F_L_UPP_SKEW * BS + e] = c.l_upp_skew;
    sg[BTRAPZ_F_BEG_L * BS + e] = c.beg_l; sg[BTRAPZ_F_END_L * BS + e] = c.end_l;
    double lo = 0.0, hi = 1000.0;  // solve_3d.cc:835-841
    // (round 6: this walk as a fixed-length predicated one with its reads issued together -- the selection's gain -- is no
    //  gain here: 0.2164 -> 0.2189 ms; eight lanes, eleven steps)
    for (int i = c.beg_t; i <= c.end_t; i++) {
      const int ii = i < 0 ? 0 : (i > N - 1 ? N - 1 : i);
      lo = fmax(dsb[2 * ii], lo);
      hi = fmin(dsb[2 * ii + 1], hi);
    }
    sg[BTRAPZ_F_DS_LO * BS + e] = lo; sg[BTRAPZ_F_DS_HI * BS + e] = hi;
    const int i0 = 10 * lane > N - 1 ? N - 1 : 10 * lane, i1 = 10 * lane + 1 > N - 1 ? N - 1 : 10 * lane + 1;  // :1161-1165, clamped
    sg[BTRAPZ_F_X_SKEW * BS + e] = (sref[i1] - sref[i0]) / a.delta; sg[BTRAPZ_F_X_BIAS * BS + e] = sref[i0];
    sg[BTRAPZ_F_Y_SKEW * BS + e] = (lref[i1] - lref[i0]) / a.delta; sg[BTRAPZ_F_Y_BIAS * BS + e] = lref[i0];
  }
  if (__any(bad)) S = -1;
  if (lane == 0) {
    a.seg_count[b] = S;
    a.ref_end[(size_t)b * 2] = sref[N - 1]; a.ref_end[(size_t)b * 2 + 1] = lref[N - 1];
  }
  if (lane < 10) a.dl10[(size_t)b * 10 + lane] = dl10_v;
  CABL_MARK(10, "END");
}

// ---- the corridor stage without the wave-wide kernel's limits (round 6) ---------------------------------------------------
// corridor_serial_kernel: ONE LANE per candidate runs the serial statements of corridor_core.h -- what the host driver of
// find_traj runs (corridor.cpp: extract_segments, select_segments, order_segments_core) -- on segment lists in a workspace
// in global memory.  The reference has no limit on knots, obstacles or segments (std::vector throughout,
// src/solve_3d.cc:323-486,488-714,729-772); the wave-wide kernels above hold their lists in LDS and one selected segment
// per lane: N <= 512 knots, <= 64 obstacles, <= 64 selected segments.  Beyond any of those the stage takes this kernel:
// same statements, same arithmetic, same record -- slow (a lane walks every knot of every segment; the loads of
// neighbouring lanes are a candidate apart) and without limits but the workspace the host sizes (cap_o segments per
// obstacle, cap_sel selected ones: a candidate that needs more gets seg_count = -1, as in the first pass above).
static_assert(sizeof(Seg) == 104, "btrapz_host.hip sizes the serial kernel's workspace with this");
__global__ __launch_bounds__(64) void corridor_serial_kernel(const CorridorArgs a, Seg *ws_all, Seg *ws_sel) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  const int N = a.N, O = a.num_obs, cap_o = a.cap_o, cap_sel = a.cap_sel;
  Seg *all = ws_all + (size_t)b * O * cap_o, *sel = ws_sel + (size_t)b * cap_sel;
  const double *gs = a.s_bounds + (size_t)b * O * N * 2, *gl = a.l_bounds + (size_t)b * O * N * 2;
  const double *sref = a.s_ref + (size_t)b * N, *lref = a.l_ref + (size_t)b * N;
  const double *dsb = a.ds_bounds + (size_t)b * N * 2;
  int S = 0, nsel = 0, carry = 0;
  bool overflow = false;
  for (int o = 0; o < O && !overflow; o++) {
    const BoundsView sb{gs + (size_t)o * N * 2}, lb{gl + (size_t)o * N * 2};
    Seg *list = all + (size_t)o * cap_o;
    const int n = extract_segments_core(a.variant, N, a.delta, sb, lb, SlopesOnTheFly{sb, a.delta}, list, cap_o);
    if (n < 0) { overflow = true; break; }
    for (int j = 0; j < n && !overflow; j++) {   // CollisionCheck's selection, solve_3d.cc:534-596
      const Seg c = list[j];
      int hits = 0;
      for (int i = 0; i < N; i++) hits += knot_inside(c, sref[i], lref[i], (double)i, a.delta) ? 1 : 0;
      for (int copies = selection_copies(selection_pushes(hits, carry), c); copies > 0; copies--) {
        if (nsel >= cap_sel) { overflow = true; break; }
        Seg t = c; t.count = 3; sel[nsel++] = t;
      }
    }
  }
  if (overflow) S = -1;
  else if (nsel > 0) S = order_segments_core(a.variant, a.delta, sel, nsel);
  bool bad = S > a.seg_stride;
  if (S > 0 && !bad) {
    const size_t BS = (size_t)a.B * a.seg_stride;
    double *sg = a.seg;
    for (int k = 0; k < S; k++) {
      const Seg c = sel[k];
      if (!(c.t > 0.0)) bad = true;
      const size_t e = (size_t)b * a.seg_stride + k;
      sg[BTRAPZ_F_T * BS + e] = c.t;
      sg[BTRAPZ_F_DOWN_BIAS * BS + e] = c.down_bias; sg[BTRAPZ_F_DOWN_SKEW * BS + e] = c.down_skew;
      sg[BTRAPZ_F_UPP_BIAS * BS + e] = c.upp_bias; sg[BTRAPZ_F_UPP_SKEW * BS + e] = c.upp_skew;
      sg[BTRAPZ_F_L_DOWN_BIAS * BS + e] = c.l_down_bias; sg[BTRAPZ_F_L_DOWN_SKEW * BS + e] = c.l_down_skew;
      sg[BTRAPZ_F_L_UPP_BIAS * BS + e] = c.l_upp_bias; sg[BTRAPZ_F_L_UPP_SKEW * BS + e] = c.l_upp_skew;
      sg[BTRAPZ_F_BEG_L * BS + e] = c.beg_l; sg[BTRAPZ_F_END_L * BS + e] = c.end_l;
      double lo = 0.0, hi = 1000.0;  // solve_3d.cc:835-841
      for (int i = c.beg_t; i <= c.end_t; i++) {
        const int ii = i < 0 ? 0 : (i > N - 1 ? N - 1 : i);
        lo = fmax(dsb[2 * ii], lo);
        hi = fmin(dsb[2 * ii + 1], hi);
      }
      sg[BTRAPZ_F_DS_LO * BS + e] = lo; sg[BTRAPZ_F_DS_HI * BS + e] = hi;
      const int i0 = 10 * k > N - 1 ? N - 1 : 10 * k, i1 = 10 * k + 1 > N - 1 ? N - 1 : 10 * k + 1;  // :1161-1165, clamped
      sg[BTRAPZ_F_X_SKEW * BS + e] = (sref[i1] - sref[i0]) / a.delta; sg[BTRAPZ_F_X_BIAS * BS + e] = sref[i0];
      sg[BTRAPZ_F_Y_SKEW * BS + e] = (lref[i1] - lref[i0]) / a.delta; sg[BTRAPZ_F_Y_BIAS * BS + e] = lref[i0];
    }
  }
  if (bad) S = -1;
  a.seg_count[b] = S;
  a.ref_end[(size_t)b * 2] = sref[N - 1]; a.ref_end[(size_t)b * 2 + 1] = lref[N - 1];
  for (int j = 0; j < 10; j++) {
    const int i = j >> 1, ii = i > N - 1 ? N - 1 : i;
    a.dl10[(size_t)b * 10 + j] = a.dl_bounds[((size_t)b * N + ii) * 2 + (j & 1)];
  }
}

// ---- bucketing by segment count ------------------------------------------------------------------
// meta[0..65]: histogram, then cand_prefix ; meta[66..131]: wave_prefix ; meta[132..197]: cursors.
// Counts are aggregated per workgroup in LDS first: with one global atomic per candidate a batch whose candidates
// all have the same count (the usual case) serialises on one address -- 0.74 ms per kernel at B = 65 536.
// fixed_S > 0: the keys are difficulty hints of a uniform batch (any int; clamped to a class 1..64) instead of counts.
__device__ __forceinline__ int hint_class(int v) { return v < 1 ? 1 : (v > 64 ? 64 : v); }
// fixed_S < 0: classes of a uniform batch of -fixed_S segments in which a key <= 0 means "not listed" (the resume pass of
// a capped solve lists the suspended axis problems only)
__device__ __forceinline__ int bucket_key(int v, int fixed_S) {
  return fixed_S > 0 ? hint_class(v) : fixed_S < 0 ? (v < 1 ? 0 : hint_class(v)) : v;
}
// (A launch with gridDim.y = 2 builds the lists of both axes at once: keys [2][B], tables [2][198], lists [2][B].)
__global__ void bucket_hist_kernel(int B, int seg_stride, const int *seg_count, int *meta, int fixed_S) {
  seg_count += (size_t)blockIdx.y * B; meta += blockIdx.y * 198;
  __shared__ int h[65];
  for (int j = threadIdx.x; j < 65; j += blockDim.x) h[j] = 0;
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) {
    const int s = bucket_key(seg_count[i], fixed_S);
    if (s >= 1 && s <= 64 && (fixed_S || s <= seg_stride)) atomicAdd(&h[65 - s], 1);   // slot 65 - key: see the prefix kernel
  }
  __syncthreads();
  for (int j = threadIdx.x; j < 65; j += blockDim.x) if (h[j]) atomicAdd(&meta[132 + j], h[j]);
}
// Buckets are laid out by DESCENDING key (slot j holds key 65 - j): wavefronts of long corridors / hard classes are
// launched first, so the longest-running wavefronts do not start at the tail of the launch.
// (round 6: lane j - 1 owns slot j and the two prefixes are wave scans -- until then ONE lane walked the 64 slots, three
//  dependent global accesses per step: 11.6 us per solve of the two-launch form for 64 additions, a third of its small kernels)
__global__ __launch_bounds__(64) void bucket_prefix_kernel(int *meta, int fixed_S) {
  if (blockIdx.x != 0) return;
  meta += blockIdx.y * 198;
  const int lane = threadIdx.x, j = lane + 1;
  const int key = 65 - j;
  const int cnt = meta[132 + j], gpw = 64 / (fixed_S > 0 ? fixed_S : fixed_S < 0 ? -fixed_S : key);
  const int waves = (cnt + gpw - 1) / gpw;
  const int cand_incl = wave_inclusive_scan(cnt, lane), wave_incl = wave_inclusive_scan(waves, lane);
  meta[j] = cand_incl - cnt; meta[66 + j] = wave_incl - waves;
  meta[132 + j] = 0;  // becomes the scatter cursor
  if (lane == 0) { meta[0] = 0; meta[66] = 0; }
  if (lane == 63) { meta[65] = cand_incl; meta[66 + 65] = wave_incl; }
}
// Order inside a bucket: index order within a 256-candidate workgroup (stable ranks from ballots), workgroups in
// the order their atomics land.  It only decides which candidates share a wavefront -- every group of a wavefront
// is solved independently, so results do not depend on it -- but neighbours in a wavefront stay neighbours in
// memory, which keeps the loads of the batch record and of the warm-start arrays coalesced.
__global__ __launch_bounds__(256) void bucket_scatter_kernel(int B, int seg_stride, const int *seg_count, int *meta,
                                                             int *order, double *axis_obj, int *axis_status,
                                                             int *axis_iters, int fixed_S) {
  seg_count += (size_t)blockIdx.y * B; meta += blockIdx.y * 198; order += (size_t)blockIdx.y * B;
  __shared__ int wcount[4][65], base[65];
  for (int j = threadIdx.x; j < 4 * 65; j += blockDim.x) (&wcount[0][0])[j] = 0;
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int s = 0, rank = 0;
  bool usable = false;
  if (i < B) {
    s = bucket_key(seg_count[i], fixed_S);
    usable = s >= 1 && s <= 64 && (fixed_S || s <= seg_stride);
    if (!usable && axis_status) {  // no usable corridor: the reference's find_traj fails here (empty selection / CHECK)
      axis_obj[2 * i] = 0.0; axis_obj[2 * i + 1] = 0.0;
      axis_status[2 * i] = BTRAPZ_NO_CORRIDOR; axis_status[2 * i + 1] = BTRAPZ_NO_CORRIDOR;
      axis_iters[2 * i] = 0; axis_iters[2 * i + 1] = 0;
    }
  }
  // rank among the lanes of this wavefront with the same key, in lane order: one ballot per distinct key
  unsigned long long remaining = __ballot(usable);
  while (remaining) {
    const int leader = __ffsll((long long)remaining) - 1;
    const int k = __builtin_amdgcn_readlane(s, leader);
    const unsigned long long m = __ballot(usable && s == k);
    if (usable && s == k) rank = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == leader) wcount[wave][65 - k] = __popcll(m);
    remaining &= ~m;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < 65; j += blockDim.x) {
    const int n = wcount[0][j] + wcount[1][j] + wcount[2][j] + wcount[3][j];
    if (n) base[j] = atomicAdd(&meta[132 + j], n);   // one range per bucket and workgroup
  }
  __syncthreads();
  if (usable) {
    int before = 0;
    const int j = 65 - s;   // slot
    for (int w = 0; w < wave; w++) before += wcount[w][j];
    order[meta[j] + base[j] + before + rank] = i;
  }
}

}  // namespace btrapz
