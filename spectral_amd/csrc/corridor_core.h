// corridor_core.h -- the corridor stage on fixed-capacity arrays, compiled for host AND device.
//
// One implementation serves both the C++ find_traj driver (corridor.cpp) and the batched device
// pipeline (corridor_kernels.hip).  Restates (file:line in /root/reference)
//   CorridorGeneration  src/solve_3d.cc:323-486   src/cuboid_3d.cc:301-407
//   CorridorSplit       src/solve_3d.cc:729-772   src/cuboid_3d.cc:588-625
//   CollisionCheck      src/solve_3d.cc:488-714   src/cuboid_3d.cc:409-573
// with the reference's arithmetic kept expression by expression (several factors of its edge
// functions are identically zero; `(k*d + b) - b` is NOT simplified: it is not k*d in floating point).
#ifndef BTRAPZ_CORRIDOR_CORE_H
#define BTRAPZ_CORRIDOR_CORE_H

#if defined(__HIPCC__)
#define BTRAPZ_HD __host__ __device__ inline
#else
#define BTRAPZ_HD inline
#endif

#include <math.h>

// Expression by expression means operation by operation: a fused multiply-add rounds once where the reference rounds
// twice, and the signs of the edge functions below decide membership on degenerate quads (collapsed or crossing
// bounds).  hipcc contracts device code by default; from here to the end of the translation unit it does not.
#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace btrapz {

// One corridor segment == the reference's Cube (include/btrapz/cube_type.h:2-24).  POD: usable in LDS.
struct Seg {
  int beg_t, end_t;
  double t;
  double beg_l, end_l;
  double upp_skew, upp_bias, down_skew, down_bias;
  double l_upp_skew, l_upp_bias, l_down_skew, l_down_bias;
  int count;
};

BTRAPZ_HD Seg seg_default() {  // Cube::Cube(), cube_type.h:12-21
  Seg s;
  s.beg_t = 0; s.end_t = 0; s.t = 0.0; s.beg_l = 0.0; s.end_l = 0.0;
  s.upp_skew = 0.0; s.upp_bias = 1000.0; s.down_skew = 0.0; s.down_bias = 0.0;
  s.l_upp_skew = 0.0; s.l_upp_bias = 1000.0; s.l_down_skew = 0.0; s.l_down_bias = 0.0;
  s.count = 0;
  return s;
}

// Per-knot bounds as interleaved (lower, upper) pairs with an element stride, so the same code reads
// a std::vector<pair>, a file-ordered [N][2] block or an LDS staging buffer.
struct BoundsView {
  const double *p;
  BTRAPZ_HD double lo(int i) const { return p[2 * i]; }
  BTRAPZ_HD double hi(int i) const { return p[2 * i + 1]; }
};

// Per-knot slopes of the s bounds, (b(i) - b(i-1)) / delta for i = 1..N-1: computed on the fly on the host, read
// from a table the lanes filled in parallel on the device (the divisions are the expensive part of the scan; the
// expression -- hence the rounding -- is the same).
struct SlopesOnTheFly {
  BoundsView sb;
  double delta;
  BTRAPZ_HD double down(int i) const { return (sb.lo(i) - sb.lo(i - 1)) / delta; }
  BTRAPZ_HD double up(int i) const { return (sb.hi(i) - sb.hi(i - 1)) / delta; }
};
struct SlopeTable {  // interleaved (down, up) pairs, entry i at p[2 i]
  const double *p;
  BTRAPZ_HD double down(int i) const { return p[2 * i]; }
  BTRAPZ_HD double up(int i) const { return p[2 * i + 1]; }
};

// ---- CorridorGeneration + CorridorSplit for one obstacle.  Returns the number of segments written,
// or -1 when `cap` is too small.
// (SB, LB: anything with lo(i) / hi(i) -- BoundsView, or the strips of the fused prism + corridor kernel)
template <class SB, class LB, class Slopes>
BTRAPZ_HD int extract_segments_core(int variant, int N, double delta, SB sb, LB lb, Slopes sk, Seg *v, int cap) {
  if (cap < 1 || N < 3) return -1;
  int n = 0;
  {
    Seg s = seg_default();
    s.beg_t = 0;
    s.down_skew = sk.down(1); s.down_bias = sb.lo(0);
    s.upp_skew = sk.up(1); s.upp_bias = sb.hi(0);
    if (variant == 0) {  // solve_3d.cc:338-341
      s.l_down_skew = (lb.lo(1) - lb.lo(0)) / delta; s.l_down_bias = lb.lo(0);
      s.l_upp_skew = (lb.hi(1) - lb.hi(0)) / delta; s.l_upp_bias = lb.hi(0);
    }
    s.beg_l = lb.lo(0); s.end_l = lb.hi(0);
    v[n++] = s;
  }
  const double threshold = 0.2;  // solve_3d.cc:372
  double cur_down = v[0].down_skew, cur_up = v[0].upp_skew;  // slopes of the open segment
  for (int i = 2; i < N - 1; i++) {
    const double dskew = sk.down(i), uskew = sk.up(i);
    if (fabs(dskew - cur_down) > threshold || fabs(uskew - cur_up) > threshold) {
      if (n + 1 > cap) return -1;
      v[n - 1].end_t = i;
      Seg s = seg_default();
      s.beg_t = i;
      s.down_skew = sk.down(i + 1); s.down_bias = sb.lo(i);
      s.upp_skew = sk.up(i + 1); s.upp_bias = sb.hi(i);
      cur_down = s.down_skew; cur_up = s.upp_skew;
      s.beg_l = lb.lo(i); s.end_l = lb.hi(i);
      if (variant == 0) {  // solve_3d.cc:358-367: the l line of a later segment is the backward difference at i
        s.l_down_bias = lb.lo(i); s.l_upp_bias = lb.hi(i);
        s.l_down_skew = (lb.lo(i) - lb.lo(i - 1)) / delta; s.l_upp_skew = (lb.hi(i) - lb.hi(i - 1)) / delta;
      }
      v[n++] = s;
    }
  }
  v[n - 1].end_t = N - 1;
  for (int i = 0; i < n; i++) v[i].t = (v[i].end_t - v[i].beg_t) * delta;
  // CorridorSplit: peel 1.0 s / 10-knot pieces off the front of every segment while its t > 1 (the reference
  // hard-codes delta = 0.1: solve_3d.cc:735-746; it inserts each piece with vector::insert).  Same pieces, same
  // arithmetic, but written straight to their final slots: count first, then fill from the back (the slots of
  // segment k start at k + the pieces peeled before it, so unread segments are never overwritten).
  int total = 0;
  for (int k = 0; k < n; k++) {
    double t = v[k].t;
    int h = 0;
    while (t > 1) { t = t - 1; if (++h > cap) return -1; }   // bounded: a huge (or infinite) t must not spin
    total += h + 1;
  }
  if (total > cap) return -1;
  int pos = total;
  for (int k = n - 1; k >= 0; k--) {
    Seg rest = v[k];
    int h = 0;
    { double t = rest.t; while (t > 1) { t = t - 1; h++; } }
    pos -= h + 1;
    int w = pos;
    while (rest.t > 1) {
      rest.t = rest.t - 1;
      Seg head = seg_default();
      head.beg_t = rest.beg_t;
      head.end_t = head.beg_t + 10;
      head.t = 1.0;
      head.down_skew = rest.down_skew; head.down_bias = rest.down_bias;
      head.upp_skew = rest.upp_skew; head.upp_bias = rest.upp_bias;
      if (variant == 0) {
        head.l_down_skew = rest.l_down_skew; head.l_down_bias = rest.l_down_bias;
        head.l_upp_skew = rest.l_upp_skew; head.l_upp_bias = rest.l_upp_bias;
      }
      head.beg_l = rest.beg_l; head.end_l = rest.end_l;
      rest.beg_t = rest.beg_t + 10;
      rest.down_bias = head.down_bias + 1.0 * head.down_skew;
      rest.upp_bias = head.upp_bias + 1.0 * head.upp_skew;
      v[w++] = head;
    }
    v[w] = rest;
  }
  return total;
}

BTRAPZ_HD bool same_segment(const Seg &a, const Seg &b) {  // solve_3d.cc:621
  return a.beg_t == b.beg_t && a.end_t == b.end_t && a.down_bias == b.down_bias && a.down_skew == b.down_skew &&
         a.upp_bias == b.upp_bias && a.upp_skew == b.upp_skew && a.beg_l == b.beg_l && a.end_l == b.end_l;
}

// Point-in-quadrilateral by the signs of four edge functions (solve_3d.cc:534-581).
BTRAPZ_HD bool knot_inside(const Seg &c, double s, double l, double knot, double delta) {
  if (!(l <= c.end_l && l >= c.beg_l)) return false;
  const double d0 = (s - c.down_bias) * (c.beg_t - c.beg_t) - (knot - c.beg_t) * (c.upp_bias - c.down_bias);
  const double d1 = (s - c.upp_bias) * (c.end_t - c.beg_t) - (knot - c.beg_t) * (c.upp_skew * delta + c.upp_bias - c.upp_bias);
  const double d2 = (s - c.upp_bias - c.upp_skew * delta) * (c.end_t - c.end_t) -
                    (knot - c.end_t) * (c.down_skew * delta + c.down_bias - c.upp_skew * delta - c.upp_bias);
  const double d3 = (s - c.down_bias - c.down_skew * delta) * (c.beg_t - c.end_t) -
                    (knot - c.end_t) * (c.down_bias - c.down_skew * delta - c.down_bias);
  const bool pos = d0 > 0 || d1 > 0 || d2 > 0 || d3 > 0;
  const bool neg = d0 < 0 || d1 < 0 || d2 < 0 || d3 < 0;
  return !(pos && neg);
}

// knot_inside for a knot of the segment's OWN span, beg_t <= knot <= end_t, of a segment with finite fields and
//   gap0 = upp_bias - down_bias in (0, 1e300),  gap1 = down_skew delta + down_bias - upp_skew delta - upp_bias in (-1e300, 0)
// and a finite reference -- what the device's selection establishes before it restricts a segment's walk to its own knots
// (corridor_kernels.hip).  There the first products of d0 and d2 are (finite) * (t - t) = +-0 exactly, and their second
// ones are (knot - beg_t) * gap0 >= 0 and (knot - end_t) * gap1 >= 0, zero only at the span's ends and never rounded to
// zero elsewhere (an integer of magnitude >= 1 times a non-zero double): d0 and d2 are never positive, d0 < 0 iff
// knot > beg_t, d2 < 0 iff knot < end_t.  Same decision as knot_inside, bit for bit, without the two edge functions.
BTRAPZ_HD bool knot_inside_own_span(const Seg &c, double s, double l, int knot_i, double delta) {
  if (!(l <= c.end_l && l >= c.beg_l)) return false;
  const double knot = (double)knot_i;
  const double d1 = (s - c.upp_bias) * (c.end_t - c.beg_t) - (knot - c.beg_t) * (c.upp_skew * delta + c.upp_bias - c.upp_bias);
  const double d3 = (s - c.down_bias - c.down_skew * delta) * (c.beg_t - c.end_t) -
                    (knot - c.end_t) * (c.down_bias - c.down_skew * delta - c.down_bias);
  const bool pos = d1 > 0 || d3 > 0;
  const bool neg = knot_i > c.beg_t || knot_i < c.end_t || d1 < 0 || d3 < 0;
  return !(pos && neg);
}

// The reference takes a segment each time a running hit counter reaches 3; the counter is shared by all
// segments of all obstacles and reset only when a segment is taken (solve_3d.cc:584-596).  Given the
// number of reference knots inside a segment and the counter carried in, this returns how many copies the
// reference pushes (>= 1 means the segment is selected; later copies are exact duplicates, removed by its
// de-dup pass) and updates the counter.
BTRAPZ_HD int selection_pushes(int hits_inside, int &carry) {
  const int total = carry + hits_inside;
  carry = total % 3;
  return total / 3;
}

// How many of those copies survive the reference's de-dup pass: one -- unless the segment does not compare equal to
// itself (a NaN among the fields same_segment compares: garbage input, but the count is the reference's), then all.
BTRAPZ_HD int selection_copies(int pushes, bool equals_itself) {
  return pushes < 1 ? 0 : (pushes >= 2 && !equals_itself ? pushes : 1);
}
BTRAPZ_HD int selection_copies(int pushes, const Seg &c) { return selection_copies(pushes, same_segment(c, c)); }

// De-dup (keep first), then ordering and time-overlap resolution: solve_3d.cc:617-703 (trapezoid),
// cuboid_3d.cc:538-567 (cuboid: no sort, no reorder, every later twin, a third of the span).  Separate steps so that
// the device can run all but the overlap walk across the lanes (corridor_kernels.hip).
BTRAPZ_HD int dedup_segments_core(Seg *v, int n) {
  for (int i = 0; i + 1 < n; i++)
    for (int j = i + 1; j < n;) {
      if (same_segment(v[i], v[j])) { for (int m = j; m + 1 < n; m++) v[m] = v[m + 1]; n--; } else j++;
    }
  return n;
}
BTRAPZ_HD void sort_segments_core(Seg *v, int n) {  // stable insertion sort by beg_t (libstdc++'s behaviour for n <= 16)
  for (int i = 1; i < n; i++) {
    const Seg x = v[i];
    int j = i - 1;
    while (j >= 0 && x.beg_t < v[j].beg_t) { v[j + 1] = v[j]; j--; }
    v[j + 1] = x;
  }
}
// Trapezoid variant, first half: pull a segment that continues segment i's lane next to it (solve_3d.cc:640-666).
// Only beg_l, beg_t and end_t are compared; the device runs the k search across the lanes (corridor_kernels.hip).
BTRAPZ_HD void reorder_segments_core(Seg *v, int n) {
  for (int i = 0; i + 1 < n; i++)
    for (int j = i + 1; j < n; j++) {
      if (v[i].beg_l == v[j].beg_l && j - i == 1) break;
      for (int k = j + 1; k < n; k++)
        if (v[i].beg_l == v[k].beg_l && v[i].end_t == v[k].beg_t) { const Seg x = v[j]; v[j] = v[k]; v[k] = x; break; }
    }
}
// Time overlaps: between neighbours (trapezoid, solve_3d.cc:678-703: its inner loop breaks after j = i + 1) or between every pair of twins (cuboid).
// Each step sees the spans the previous one left, so this stays a serial walk on the device too.
BTRAPZ_HD void overlap_segments_core(int variant, double delta, Seg *v, int n) {
  if (variant == 0) {
    for (int i = 0; i + 1 < n; i++) {
      Seg &a = v[i], &b = v[i + 1];
      if (a.beg_t == b.beg_t && a.end_t == b.end_t) {
        const int half = (a.end_t - a.beg_t) / 2;
        a.end_t -= half; a.t = (a.end_t - a.beg_t) * delta;
        b.beg_t += half; b.t = (b.end_t - b.beg_t) * delta;
      } else if (a.beg_t > b.beg_t && a.end_t <= b.end_t) {
        const int half = (a.end_t - a.beg_t) / 2;
        if (half > 1) { a.end_t -= half; a.t = (a.end_t - a.beg_t) * delta; }
        b.beg_t = a.end_t; b.t = (b.end_t - b.beg_t) * delta;
      }
    }
  } else {
    for (int i = 0; i + 1 < n; i++)
      for (int j = i + 1; j < n; j++)
        if (v[i].beg_t == v[j].beg_t && v[i].end_t == v[j].end_t) {
          const int third = (v[i].end_t - v[i].beg_t) / 3;
          v[i].end_t -= third; v[i].t = (v[i].end_t - v[i].beg_t) * delta;
          v[j].beg_t += third; v[j].t = (v[j].end_t - v[j].beg_t) * delta;
        }
  }
}
BTRAPZ_HD void resolve_segments_core(int variant, double delta, Seg *v, int n) {
  if (variant == 0) reorder_segments_core(v, n);
  overlap_segments_core(variant, delta, v, n);
}
BTRAPZ_HD int order_segments_core(int variant, double delta, Seg *v, int n) {
  n = dedup_segments_core(v, n);
  if (variant == 0) sort_segments_core(v, n);
  resolve_segments_core(variant, delta, v, n);
  return n;
}

}  // namespace btrapz
#endif
