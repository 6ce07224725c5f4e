// host_check.cpp -- driver of the sanitizer build of the product's HOST code (Makefile target host_asan: g++
// -fsanitize=address,undefined, no HIP): the corridor-file scanner and the %.3f trajectory writer (corridor.cpp),
// TokenReader's reference-compatible failure mode, the host instantiation of corridor_core.h (extraction, split,
// selection), trajectory_cost (traj_cost.h) with its clamped reads, and the strip geometry of prism_core.h run as 64
// threads per wavefront (simt_shim.h).  The reference reads its input unchecked (src/trp_wrapper.cpp:39-144, OOB reads
// at :221,257,269 and src/solve_3d.cc:1161); this is where the product is held to not doing so.
// tests/test_host_sanitizers.py runs it over the bundled inputs, damaged copies of them and seeded scenes.
//   host_check corridor <variant> <file> [out.txt]   -> "S <n>" + one line per segment; writes a trajectory file when asked
//   host_check text <seed> <count>                   -> parse_double / format_3 against strtod / snprintf
//   host_check prisms <seed> <scenes>                -> strips of seeded scenes, one line per strip and knot sample
//   host_check knots <seed> <count>                  -> knot_inside_own_span against knot_inside under its preconditions
#include "simt_shim.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../corridor.hpp"
#include "../traj_cost.h"
#include "../prism_core.h"

thread_local simt::Wave *simt::wave = nullptr;
thread_local int simt::lane = 0;

using namespace btrapz;

static int run_corridor(int variant, const char *path, const char *out_path) {
  TrajInput in;
  if (!read_traj_input(path, in)) { printf("S -1\n"); return 0; }
  std::vector<std::vector<Segment>> lists;
  for (int o = 0; o < in.num_obs; o++) lists.push_back(extract_segments(variant, in.N, in.delta, in.s_bounds[o], in.l_bounds[o]));
  std::vector<Segment> seg;
  if (!select_segments(variant, in.delta, lists, in.s_ref, in.l_ref, seg)) { printf("S 0\n"); return 0; }
  printf("S %d\n", (int)seg.size());
  for (const Segment &c : seg)
    printf("%d %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %d\n", c.beg_t, c.end_t, c.t, c.beg_l, c.end_l, c.upp_skew,
           c.upp_bias, c.down_skew, c.down_bias, c.l_upp_skew, c.l_upp_bias, c.l_down_skew, c.l_down_bias, c.count);
  // the cost and the writer on a trajectory made of the references themselves: np beyond N, np = 1 and the exact
  // length (the reference reads x_ref[i] for i < num_of_points whatever N is, trp_wrapper.cpp:221,257)
  Params p;
  memset(&p, 0, sizeof(p));
  p.s_acc_weight = p.s_jerk_weight = p.l_acc_weight = p.l_jerk_weight = 0.3;
  p.weight_s_ref = p.weight_ds_ref = p.weight_l_ref = p.weight_dl_ref = 0.7; p.weight_end_s = p.weight_end_l = 1.1;
  for (int np : {1, 2, in.N - 1, in.N, in.N + 7}) {
    if (np < 1) continue;
    std::vector<double> s(np), l(np), d(np, 0.25), dd(np, -0.5);
    for (int i = 0; i < np; i++) { s[i] = in.s_ref[clampi(i, in.N - 1)] + 0.125 * i; l[i] = in.l_ref[clampi(i, in.N - 1)] - 0.0625; }
    for (int v = 0; v < 2; v++)
      printf("cost np %d variant %d %.17g\n", np, v, trajectory_cost(v, p, in, np, s.data(), d.data(), dd.data(), l.data(), d.data(), dd.data()));
    if (out_path && np == in.N) {
      s[0] = 1e300; l[0] = -0.0005; if (np > 2) { s[1] = std::nan(""); s[2] = -HUGE_VAL; }     // what a failed solve could hand over
      printf("write %d\n", write_trajectory_file(out_path, np, in.delta, s.data(), l.data(), d.data(), d.data(), dd.data(), dd.data()) ? 1 : 0);
    }
  }
  return 0;
}

static int run_text(unsigned seed, int count) {
  std::mt19937_64 rng(seed);
  int bad = 0;
  char a[FORMAT_3_MAX + 8], b[FORMAT_3_MAX + 8];
  for (int i = 0; i < count; i++) {
    double v;
    const unsigned long long bits = rng();
    switch (i % 4) {
      case 0: memcpy(&v, &bits, 8); break;                                          // any bit pattern: inf, nan, denormals
      case 1: v = (double)(long long)(bits % 2000000001ull - 1000000000ll) / 1000.0 + 0.0005; break;   // ties of %.3f
      case 2: v = std::ldexp((double)(bits >> 11), -53) * 200.0 - 100.0; break;      // what a corridor file holds
      default: v = std::ldexp((double)(bits >> 11), (int)(bits % 120) - 60); break;
    }
    const int n = format_3(a, v);
    snprintf(b, sizeof(b), "%.3f", v);
    if (n != (int)strlen(b) || strcmp(a, b) != 0) { if (bad++ < 5) printf("format_3 %a: '%s' != '%s'\n", v, a, b); }
    // the scanner on the 17-digit, the 3-decimal and a damaged rendering of the value
    char txt[3][64];
    snprintf(txt[0], 64, "%.17g", v); snprintf(txt[1], 64, " \t%.3f ", v); snprintf(txt[2], 64, "%.6e", v);
    txt[2][(bits >> 7) % (strlen(txt[2]) + 1)] = "e+-.x"[(bits >> 3) % 5];
    for (int k = 0; k < 3; k++) {
      const char *e1 = nullptr; char *e2 = nullptr;
      const double p1 = parse_double(txt[k], &e1), p2 = strtod(txt[k], &e2);
      if (e1 != e2 || memcmp(&p1, &p2, 8) != 0) {
        if (!(p1 != p1 && p2 != p2 && e1 == e2)) { if (bad++ < 5) printf("parse_double '%s': %a (%d) != %a (%d)\n", txt[k], p1, (int)(e1 - txt[k]), p2, (int)(e2 - txt[k])); }
      }
    }
  }
  printf("text %d values, %d differences\n", count, bad);
  return bad ? 1 : 0;
}

static int run_prisms(unsigned seed, int scenes) {
  std::mt19937_64 rng(seed);
  auto uni = [&](double lo, double hi) { return lo + (hi - lo) * std::ldexp((double)(rng() >> 11), -53); };
  btrapz_road road;
  memset(&road, 0, sizeof(road));
  road.knots_per_second = 10.0; road.s_lo = 0.0; road.s_hi = 50.0; road.l_lo = -2.0; road.l_hi = 8.0;
  road.l_safe = 5.0 / 3 + 5.0 / 3; road.w_safe = 2.0 / 3 + 2.0 / 3;
  const PrismRoad r = prism_road(&road);
  const int N = 71;
  for (int sc = 0; sc < scenes; sc++) {
    const int P = 1 + (int)(rng() % PRISM_MAX_CARS);
    std::vector<double> prisms((size_t)P * 8, 0.0);
    for (int q = 0; q < P; q++) {
      double *c = &prisms[(size_t)q * 8];
      const bool ahead = rng() & 1;
      c[0] = uni(5, 40); c[1] = uni(-3, 9); c[2] = ahead ? 0.0 : uni(0.1, 3.0); c[3] = uni(0, 8);
      c[4] = (double[]){0.0, 0.3, -0.3}[rng() % 3]; c[5] = (rng() & 1) ? 3.0 : 4.0; c[6] = (rng() % 8) ? 1.0 : 0.0;
    }
    // the table block sized exactly as the kernels size it: a write past prism_tab_bytes(P) is the sanitizer's to find
    std::vector<char> mem(prism_tab_bytes(P));
    simt::Wave wave;
    std::vector<int> strips(simt::WAVE, -1);
    std::vector<std::vector<double>> rows(simt::WAVE);
    std::vector<std::thread> th;
    for (int ln = 0; ln < simt::WAVE; ln++)
      th.emplace_back([&, ln] {
        simt::wave = &wave; simt::lane = ln;
        PrismTab t = prism_tab_at(mem.data(), P);
        strips[ln] = prism_tables(t, r, prisms.data(), ln);
        // every lane evaluates "its" strips at every knot, padding strips included (the fill loop of the kernels)
        for (int idx = ln; idx < (2 * P + 1) * N; idx += simt::WAVE) {
          const int j = idx / N, i = idx - j * N;
          const double2 s = prism_pair_s(t, r, j, i), l = prism_pair_l(t, j);
          if (i % 10 == 0) { rows[ln].push_back(j); rows[ln].push_back(i); rows[ln].push_back(s.x); rows[ln].push_back(s.y); rows[ln].push_back(l.x); rows[ln].push_back(l.y); }
        }
      });
    for (auto &t : th) t.join();
    for (int ln = 1; ln < simt::WAVE; ln++) if (strips[ln] != strips[0]) { printf("lanes disagree on the strip count\n"); return 1; }
    printf("scene %d cars %d strips %d\n", sc, P, strips[0]);
    printf("cars");
    for (double v : prisms) printf(" %.17g", v);
    printf("\n");
    for (int ln = 0; ln < simt::WAVE; ln++)
      for (size_t k = 0; k + 5 < rows[ln].size() + 1 && k < rows[ln].size(); k += 6)
        if ((int)rows[ln][k] < strips[0])
          printf("strip %d knot %d s %.17g %.17g l %.17g %.17g\n", (int)rows[ln][k], (int)rows[ln][k + 1], rows[ln][k + 2], rows[ln][k + 3], rows[ln][k + 4], rows[ln][k + 5]);
  }
  return 0;
}

// knot_inside_own_span against knot_inside (corridor_core.h) on seeded segments that meet the former's preconditions --
// finite fields, gap0 in (0, 1e300), gap1 in (-1e300, 0), a finite reference, a knot of the span -- ordinary ones, tiny and
// huge gaps, denormals, spans of one knot, references on and across the edges.
#include "../corridor_core.h"
static int run_knots(unsigned seed, int count) {
  std::mt19937_64 rng(seed);
  auto uni = [&](double lo, double hi) { return lo + (hi - lo) * std::ldexp((double)(rng() >> 11), -53); };
  auto mag = [&]() { return std::ldexp(uni(1.0, 2.0), (int)(rng() % 2000) - 1040); };   // 2^-1040 .. 2^959: denormals to 1e288
  int bad = 0, tried = 0, inside = 0;
  for (int n = 0; n < count; n++) {
    Seg c = seg_default();
    const double delta = (rng() % 4) ? 0.1 : uni(0.01, 1.0);
    c.beg_t = (int)(rng() % 500); c.end_t = c.beg_t + (int)(rng() % 12);
    const int kind = (int)(rng() % 4);
    c.down_bias = kind == 3 ? uni(-1, 1) * mag() : uni(-50, 100);
    c.upp_bias = c.down_bias + (kind == 0 ? uni(0.1, 30) : kind == 1 ? mag() : uni(1e-12, 1e-3));
    c.down_skew = kind == 2 ? uni(-1, 1) * mag() : uni(-8, 8);
    c.upp_skew = c.down_skew + (kind == 2 ? 0.0 : uni(-3, 3));
    c.beg_l = uni(-5, 5); c.end_l = c.beg_l + uni(0, 4);
    const double gap0 = c.upp_bias - c.down_bias, gap1 = c.down_skew * delta + c.down_bias - c.upp_skew * delta - c.upp_bias;
    if (!(gap0 > 0.0 && gap0 < 1e300 && gap1 < 0.0 && gap1 > -1e300)) continue;
    for (int i = c.beg_t; i <= c.end_t; i++) {
      const double edge = (rng() & 1) ? c.down_bias + c.down_skew * delta * (i - c.beg_t) : c.upp_bias + c.upp_skew * delta * (i - c.beg_t);
      const double s = (rng() % 3 == 0) ? edge : (rng() % 3 == 1) ? edge + uni(-1, 1) * std::ldexp(fabs(edge) + 1e-300, -50) : uni(c.down_bias - 5, c.upp_bias + 5);
      const double l = (rng() % 8) ? uni(c.beg_l, c.end_l) : uni(-6, 10);
      const bool a = knot_inside(c, s, l, (double)i, delta), b = knot_inside_own_span(c, s, l, i, delta);
      tried++; inside += a;
      if (a != b && bad++ < 5) printf("knot %d of [%d, %d]: knot_inside %d, own_span %d (s %a, gap0 %a, gap1 %a)\n", i, c.beg_t, c.end_t, a, b, s, gap0, gap1);
    }
  }
  printf("knots %d decisions (%d inside), %d differences\n", tried, inside, bad);
  return bad ? 1 : 0;
}

int main(int argc, char **argv) {
  if (argc >= 4 && !strcmp(argv[1], "knots")) return run_knots((unsigned)atoi(argv[2]), atoi(argv[3]));
  if (argc >= 4 && !strcmp(argv[1], "corridor")) return run_corridor(atoi(argv[2]), argv[3], argc > 4 ? argv[4] : nullptr);
  if (argc >= 4 && !strcmp(argv[1], "text")) return run_text((unsigned)atoi(argv[2]), atoi(argv[3]));
  if (argc >= 4 && !strcmp(argv[1], "prisms")) return run_prisms((unsigned)atoi(argv[2]), atoi(argv[3]));
  fprintf(stderr, "usage: host_check corridor <variant> <file> [out] | text <seed> <count> | prisms <seed> <scenes>\n");
  return 2;
}
