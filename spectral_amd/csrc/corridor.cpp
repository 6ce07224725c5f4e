// corridor.cpp -- see corridor.hpp.
#include "corridor.hpp"

#include "corridor_core.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace btrapz {

// ---- input file ---------------------------------------------------------------------------
// The reference reads with `ifs >> v` and never checks the stream: after the first failed
// extraction every later one is a no-op that leaves its target untouched
// (src/c_road_s1_2.txt has a short last row and relies on this).  TokenReader mirrors it.
// strtod for the numbers corridor files hold.  Plain decimals ([+-]digits[.digits][e[+-]digits]) of at most 15
// significant digits and a decimal exponent within +-22 are converted exactly as strtod does it, by ONE correctly
// rounded operation on two exactly representable values (Clinger's fast path); everything else -- longer digit
// strings, inf / nan / hex, no number at all -- goes to strtod itself.  Three times faster on the ~3 000 tokens of a
// 20-second scene (the scan was a third of a file-based find_traj call).
double parse_double(const char *p, const char **end) {
  static const double P10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15,
                                 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  const char *q = p;
  while (*q == ' ' || *q == '\n' || *q == '\t' || *q == '\r' || *q == '\f' || *q == '\v') ++q;
  bool neg = false;
  if (*q == '+' || *q == '-') { neg = *q == '-'; ++q; }
  unsigned long long mant = 0;
  int digits = 0, scale = 0;
  bool any = false, slow = false;
  while (*q >= '0' && *q <= '9') {
    any = true;
    if (mant || *q != '0') { if (++digits > 15) slow = true; else mant = mant * 10 + (unsigned)(*q - '0'); }
    ++q;
  }
  if (*q == '.') {
    ++q;
    while (*q >= '0' && *q <= '9') {
      any = true;
      if (mant || *q != '0') { if (++digits > 15) slow = true; else mant = mant * 10 + (unsigned)(*q - '0'); }
      --scale; ++q;
    }
  }
  if (any && !slow && (*q == 'e' || *q == 'E')) {
    const char *r = q + 1;
    bool eneg = false;
    if (*r == '+' || *r == '-') { eneg = *r == '-'; ++r; }
    if (*r >= '0' && *r <= '9') {
      int ex = 0;
      while (*r >= '0' && *r <= '9') { if (ex < 10000) ex = ex * 10 + (*r - '0'); ++r; }
      scale += eneg ? -ex : ex;
      q = r;
    }
  }
  if (!any || slow || scale < -22 || scale > 22 || *q == 'x' || *q == 'X' || *q == 'p' || *q == 'P') {
    char *e = nullptr;
    const double v = strtod(p, &e);
    *end = e;
    return v;
  }
  double v = (double)mant;                       // exact: below 1e15
  v = scale < 0 ? v / P10[-scale] : v * P10[scale];
  *end = q;
  return neg ? -v : v;
}

// printf("%.3f") for the numbers a trajectory file holds: v * 1000 is rounded once, to within 1e-4 of the true
// product, and unless its fraction is within 1e-3 of a tie (the residual is below 1e-4 for |v| < 1e9) the nearest integer is the correctly
// rounded decimal -- what printf computes from the exact binary expansion.  Near a tie, for |v| >= 1e9 and for inf / nan, snprintf decides.  Returns the length.
int format_3(char *out, double v) {
  const double a = fabs(v);
  if (!(a < 1e9)) return snprintf(out, FORMAT_3_MAX, "%.3f", v);
  const double y = a * 1000.0;                   // a * 1000 = y + e exactly, |e| <= ulp(y) / 2
  const double r = floor(y), frac = y - r;       // exact: y < 2^53
  if (fabs(frac - 0.5) < 1e-3) return snprintf(out, FORMAT_3_MAX, "%.3f", v);
  unsigned long long R = (unsigned long long)r + (frac > 0.5 ? 1u : 0u);
  char tmp[24];
  int n = 0;
  const unsigned f3 = (unsigned)(R % 1000u);
  unsigned long long ip = R / 1000u;
  do { tmp[n++] = (char)('0' + ip % 10u); ip /= 10u; } while (ip);
  char *o = out;
  if (signbit(v)) *o++ = '-';
  while (n) *o++ = tmp[--n];
  *o++ = '.';
  *o++ = (char)('0' + f3 / 100u); *o++ = (char)('0' + f3 / 10u % 10u); *o++ = (char)('0' + f3 % 10u);
  *o = 0;
  return (int)(o - out);
}

namespace {
// The whole file is read at once and scanned with strtod / strtol (the stream extraction it replaces spent 0.1 ms on
// the ~1000 tokens of a corridor file -- as long as the solve).
class TokenReader {
 public:
  explicit TokenReader(const std::string &path) {
    if (FILE *f = fopen(path.c_str(), "rb")) {
      open_ = true;
      char chunk[1 << 14];
      size_t n;
      while ((n = fread(chunk, 1, sizeof(chunk), f)) > 0) buf_.append(chunk, n);
      fclose(f);
    }
    p_ = buf_.c_str();
  }
  bool open() const { return open_; }
  void get(double &v) {
    if (failed_) return;
    const char *end = nullptr;
    const double t = parse_double(p_, &end);
    if (end == p_) failed_ = true; else { v = t; p_ = end; }
  }
  void get(int &v) {
    if (failed_) return;
    char *end = nullptr;
    const long t = strtol(p_, &end, 10);
    if (end == p_) failed_ = true; else { v = (int)t; p_ = end; }
  }
  bool failed() const { return failed_; }
 private:
  std::string buf_;
  const char *p_ = nullptr;
  bool open_ = false, failed_ = false;
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

bool write_trajectory_file(const std::string &path, int np, double delta, const double *s, const double *l, const double *ds,
                           const double *dl, const double *dds, const double *ddl) {
  FILE *f = fopen(path.c_str(), "w");
  if (!f) return false;
  std::string text;
  text.reserve((size_t)(np > 0 ? np : 0) * 64);
  char num[FORMAT_3_MAX + 1];
  for (int i = 0; i < np; i++) {
    const double row[7] = {i * delta, s[i], l[i], ds[i], dl[i], dds[i], ddl[i]};
    for (int c = 0; c < 7; c++) { int n = format_3(num, row[c]); num[n++] = c < 6 ? ' ' : '\n'; text.append(num, (size_t)n); }
  }
  const bool ok = fwrite(text.data(), 1, text.size(), f) == text.size();
  return fclose(f) == 0 && ok;
}

// ---- per-obstacle extraction and selection: thin std::vector wrappers over corridor_core.h ----------
static Segment to_segment(const Seg &c) {
  Segment s;
  s.beg_t = c.beg_t; s.end_t = c.end_t; s.t = c.t; s.beg_l = c.beg_l; s.end_l = c.end_l;
  s.upp_skew = c.upp_skew; s.upp_bias = c.upp_bias; s.down_skew = c.down_skew; s.down_bias = c.down_bias;
  s.l_upp_skew = c.l_upp_skew; s.l_upp_bias = c.l_upp_bias; s.l_down_skew = c.l_down_skew; s.l_down_bias = c.l_down_bias;
  s.count = c.count;
  return s;
}
static Seg to_seg(const Segment &c) {
  Seg s;
  s.beg_t = c.beg_t; s.end_t = c.end_t; s.t = c.t; s.beg_l = c.beg_l; s.end_l = c.end_l;
  s.upp_skew = c.upp_skew; s.upp_bias = c.upp_bias; s.down_skew = c.down_skew; s.down_bias = c.down_bias;
  s.l_upp_skew = c.l_upp_skew; s.l_upp_bias = c.l_upp_bias; s.l_down_skew = c.l_down_skew; s.l_down_bias = c.l_down_bias;
  s.count = c.count;
  return s;
}
static_assert(sizeof(std::pair<double, double>) == 2 * sizeof(double), "Bounds must be interleaved (lower, upper) pairs");

std::vector<Segment> extract_segments(int variant, int N, double delta, const Bounds &sb, const Bounds &lb) {
  std::vector<Seg> buf(2 * (size_t)N + 16);
  const BoundsView s{reinterpret_cast<const double *>(sb.data())}, l{reinterpret_cast<const double *>(lb.data())};
  const int n = extract_segments_core(variant, N, delta, s, l, SlopesOnTheFly{s, delta}, buf.data(), (int)buf.size());
  std::vector<Segment> out;
  for (int i = 0; i < n; i++) out.push_back(to_segment(buf[i]));
  return out;
}

bool select_segments(int variant, double delta, const std::vector<std::vector<Segment>> &lists,
                     const std::vector<double> &s_ref, const std::vector<double> &l_ref, std::vector<Segment> &out) {
  out.clear();
  std::vector<Seg> sel;
  int carry = 0;
  for (const auto &list : lists)
    for (const Segment &c0 : list) {
      const Seg c = to_seg(c0);
      int hits = 0;
      for (size_t i = 0; i < s_ref.size(); i++) hits += knot_inside(c, s_ref[i], l_ref[i], double(i), delta) ? 1 : 0;
      for (int copies = selection_copies(selection_pushes(hits, carry), c); copies > 0; copies--) { Seg t = c; t.count = 3; sel.push_back(t); }
    }
  if (sel.empty()) return false;
  const int n = order_segments_core(variant, delta, sel.data(), (int)sel.size());
  for (int i = 0; i < n; i++) out.push_back(to_segment(sel[i]));
  return true;
}

}  // namespace btrapz
