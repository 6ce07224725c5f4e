// corridor.cpp -- see corridor.hpp.
#include "corridor.hpp"

#include <algorithm>
#include <fstream>

namespace btrapz {

// ---- input file ---------------------------------------------------------------------------
// The reference reads with `ifs >> v` and never checks the stream: after the first failed
// extraction every later one is a no-op that leaves its target untouched
// (src/c_road_s1_2.txt has a short last row and relies on this).  TokenReader mirrors it.
namespace {
class TokenReader {
 public:
  explicit TokenReader(const std::string &path) : in_(path) {}
  bool open() const { return in_.is_open(); }
  void get(double &v) { double t; if (!failed_ && (in_ >> t)) v = t; else failed_ = true; }
  void get(int &v) { int t; if (!failed_ && (in_ >> t)) v = t; else failed_ = true; }
  bool failed() const { return failed_; }
 private:
  std::ifstream in_;
  bool failed_ = false;
};
}  // namespace

bool read_traj_input(const std::string &path, TrajInput &in) {
  TokenReader r(path);
  if (!r.open()) return false;
  r.get(in.N); r.get(in.delta);
  for (double &v : in.init_s) r.get(v);
  for (double &v : in.init_l) r.get(v);
  r.get(in.num_obs);
  r.get(in.ds_ref); r.get(in.dl_ref);
  r.get(in.dds[0]); r.get(in.dds[1]);
  r.get(in.ddds[0]); r.get(in.ddds[1]);
  r.get(in.ddl[0]); r.get(in.ddl[1]);
  r.get(in.dddl[0]); r.get(in.dddl[1]);
  if (r.failed() || in.N < 3 || in.N > 100000 || in.num_obs < 0 || in.num_obs > 1000 || !(in.delta > 0)) return false;
  double lo = 0, hi = 0;
  auto read_pairs = [&](Bounds &b) {
    b.clear(); b.reserve(in.N);
    for (int i = 0; i < in.N; i++) { r.get(lo); r.get(hi); b.emplace_back(lo, hi); }
  };
  auto read_vals = [&](std::vector<double> &v) {
    v.clear(); v.reserve(in.N);
    for (int i = 0; i < in.N; i++) { r.get(lo); v.push_back(lo); }
  };
  in.s_bounds.resize(in.num_obs); in.l_bounds.resize(in.num_obs);
  for (int o = 0; o < in.num_obs; o++) { read_pairs(in.s_bounds[o]); read_pairs(in.l_bounds[o]); }
  read_pairs(in.ds_bounds); read_pairs(in.dl_bounds);
  read_vals(in.s_ref); read_vals(in.l_ref); read_vals(in.s_kappa); read_vals(in.l_kappa);
  return true;
}

// ---- per-obstacle extraction ----------------------------------------------------------------
// Peel 1.0 s / 10-knot pieces off every segment longer than 1 s (the reference hard-codes
// delta = 0.1 here: solve_3d.cc:735-746).
static void split_long_segments(int variant, std::vector<Segment> &v) {
  for (size_t k = 0; k < v.size(); k++) {
    while (v[k].t > 1) {
      Segment &rest = v[k];
      rest.t = rest.t - 1;
      Segment head;
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
      v.insert(v.begin() + k, head);
      k++;
    }
  }
}

std::vector<Segment> extract_segments(int variant, int N, double delta, const Bounds &sb, const Bounds &lb) {
  std::vector<Segment> v;
  auto slope_lo = [&](const Bounds &b, int i) { return (b[i + 1].first - b[i].first) / delta; };
  auto slope_hi = [&](const Bounds &b, int i) { return (b[i + 1].second - b[i].second) / delta; };
  auto open_segment = [&](int i) {
    Segment s;
    s.beg_t = i;
    s.down_skew = slope_lo(sb, i); s.down_bias = sb[i].first;
    s.upp_skew = slope_hi(sb, i); s.upp_bias = sb[i].second;
    s.beg_l = lb[i].first; s.end_l = lb[i].second;
    return s;
  };
  {
    Segment s = open_segment(0);
    if (variant == 0) {
      s.l_down_skew = slope_lo(lb, 0); s.l_down_bias = lb[0].first;
      s.l_upp_skew = slope_hi(lb, 0); s.l_upp_bias = lb[0].second;
    }
    v.push_back(s);
  }
  const double threshold = 0.2;  // solve_3d.cc:372
  for (int i = 2; i < N - 1; i++) {
    const double dskew = slope_lo(sb, i - 1), uskew = slope_hi(sb, i - 1);
    if (std::fabs(dskew - v.back().down_skew) > threshold || std::fabs(uskew - v.back().upp_skew) > threshold) {
      v.back().end_t = i;
      Segment s = open_segment(i);
      if (variant == 0) {  // solve_3d.cc:358-367: l line of a later segment = backward difference at i
        s.l_down_bias = lb[i].first; s.l_upp_bias = lb[i].second;
        s.l_down_skew = slope_lo(lb, i - 1); s.l_upp_skew = slope_hi(lb, i - 1);
      }
      v.push_back(s);
    }
  }
  v.back().end_t = N - 1;
  for (Segment &s : v) s.t = (s.end_t - s.beg_t) * delta;
  split_long_segments(variant, v);
  return v;
}

// ---- selection along the reference trajectory ---------------------------------------------------
static bool same_segment(const Segment &a, const Segment &b) {
  return a.beg_t == b.beg_t && a.end_t == b.end_t && a.down_bias == b.down_bias && a.down_skew == b.down_skew &&
         a.upp_bias == b.upp_bias && a.upp_skew == b.upp_skew && a.beg_l == b.beg_l && a.end_l == b.end_l;
}

// Point-in-quadrilateral by the signs of four edge functions, with the reference's literal
// edge expressions (several factors are identically zero there: solve_3d.cc:536,559).
static bool knot_inside(const Segment &c, double s, double l, double knot, double delta) {
  if (!(l <= c.end_l && l >= c.beg_l)) return false;
  const double d[4] = {
      (s - c.down_bias) * (c.beg_t - c.beg_t) - (knot - c.beg_t) * (c.upp_bias - c.down_bias),
      (s - c.upp_bias) * (c.end_t - c.beg_t) - (knot - c.beg_t) * (c.upp_skew * delta + c.upp_bias - c.upp_bias),
      (s - c.upp_bias - c.upp_skew * delta) * (c.end_t - c.end_t) -
          (knot - c.end_t) * (c.down_skew * delta + c.down_bias - c.upp_skew * delta - c.upp_bias),
      (s - c.down_bias - c.down_skew * delta) * (c.beg_t - c.end_t) -
          (knot - c.end_t) * (c.down_bias - c.down_skew * delta - c.down_bias)};
  bool pos = false, neg = false;
  for (double v : d) { pos = pos || v > 0; neg = neg || v < 0; }
  return !(pos && neg);
}

bool select_segments(int variant, double delta, const std::vector<std::vector<Segment>> &lists,
                     const std::vector<double> &s_ref, const std::vector<double> &l_ref, std::vector<Segment> &out) {
  out.clear();
  // A segment is taken each time the running hit counter reaches 3; the counter is shared by
  // all segments of all obstacles and only reset when a segment is taken (solve_3d.cc:584-596).
  int hits = 0;
  for (const auto &list : lists)
    for (Segment c : list)
      for (size_t i = 0; i < s_ref.size(); i++)
        if (knot_inside(c, s_ref[i], l_ref[i], double(i), delta)) {
          c.count = ++hits;
          if (hits > 2) { out.push_back(c); hits = 0; }
        }
  if (out.empty()) return false;

  for (size_t i = 0; i + 1 < out.size(); i++)  // drop exact duplicates, keep the first
    for (size_t j = i + 1; j < out.size();)
      if (same_segment(out[i], out[j])) out.erase(out.begin() + j); else j++;

  auto retime = [&](Segment &s) { s.t = (s.end_t - s.beg_t) * delta; };
  if (variant == 0) {
    // order by start knot (ties keep their order, as libstdc++'s insertion sort does for n<=16)
    std::stable_sort(out.begin(), out.end(), [](const Segment &a, const Segment &b) { return a.beg_t < b.beg_t; });
    // pull a segment that continues segment i's lane (same beg_l, starts where i ends) next to it
    for (size_t i = 0; i + 1 < out.size(); i++)
      for (size_t j = i + 1; j < out.size(); j++) {
        if (out[i].beg_l == out[j].beg_l && j - i == 1) break;
        for (size_t k = j + 1; k < out.size(); k++)
          if (out[i].beg_l == out[k].beg_l && out[i].end_t == out[k].beg_t) { std::swap(out[j], out[k]); break; }
      }
    // time overlaps between neighbours (solve_3d.cc:678-703)
    for (size_t i = 0; i + 1 < out.size(); i++) {
      Segment &a = out[i], &b = out[i + 1];
      if (a.beg_t == b.beg_t && a.end_t == b.end_t) {
        const int half = (a.end_t - a.beg_t) / 2;
        a.end_t -= half; retime(a);
        b.beg_t += half; retime(b);
      } else if (a.beg_t > b.beg_t && a.end_t <= b.end_t) {
        const int half = (a.end_t - a.beg_t) / 2;
        if (half > 1) { a.end_t -= half; retime(a); }
        b.beg_t = a.end_t; retime(b);
      }
    }
  } else {
    // cuboid_3d.cc:553-567: every later twin, a third of the span
    for (size_t i = 0; i + 1 < out.size(); i++)
      for (size_t j = i + 1; j < out.size(); j++)
        if (out[i].beg_t == out[j].beg_t && out[i].end_t == out[j].end_t) {
          const int third = (out[i].end_t - out[i].beg_t) / 3;
          out[i].end_t -= third; retime(out[i]);
          out[j].beg_t += third; retime(out[j]);
        }
  }
  return true;
}

}  // namespace btrapz
