// corridor.hpp -- host-side corridor pipeline that feeds the hot path (product code).
//
// Restates, for the product's find_traj driver, the reference's per-obstacle corridor
// extraction and corridor selection (file:line in /root/reference):
//   CorridorGeneration  src/solve_3d.cc:323-486   src/cuboid_3d.cc:301-407
//   CorridorSplit       src/solve_3d.cc:729-772   src/cuboid_3d.cc:588-625
//   CollisionCheck      src/solve_3d.cc:488-714   src/cuboid_3d.cc:409-573
// It is independent of oracle/ (which holds its own C restatement used only as the
// checker).  Behaviour the reference leaves undefined is defined here the same way the
// oracle documents it: an empty selection is a failure, never an out-of-bounds read.
#ifndef BTRAPZ_CORRIDOR_HPP
#define BTRAPZ_CORRIDOR_HPP

#include <cmath>
#include <string>
#include <utility>
#include <vector>

namespace btrapz {

// One corridor segment == the reference's Cube (include/btrapz/cube_type.h:2-24).
struct Segment {
  int beg_t = 0, end_t = 0;
  double t = 0.0;
  double beg_l = 0.0, end_l = 0.0;
  double upp_skew = 0.0, upp_bias = 1000.0, down_skew = 0.0, down_bias = 0.0;
  double l_upp_skew = 0.0, l_upp_bias = 1000.0, l_down_skew = 0.0, l_down_bias = 0.0;
  int count = 0;
};

using Bounds = std::vector<std::pair<double, double>>;  // (lower, upper) per knot

struct TrajInput {  // grammar of src/trp_wrapper.cpp:39-144
  int N = 0;
  double delta = 0.0;
  double init_s[3] = {0, 0, 0}, init_l[3] = {0, 0, 0};
  int num_obs = 0;
  double ds_ref = 0, dl_ref = 0;
  double dds[2] = {0, 0}, ddds[2] = {0, 0}, ddl[2] = {0, 0}, dddl[2] = {0, 0};
  std::vector<Bounds> s_bounds, l_bounds;  // per obstacle
  Bounds ds_bounds, dl_bounds;
  std::vector<double> s_ref, l_ref, s_kappa, l_kappa;
};

bool read_traj_input(const std::string &path, TrajInput &in);
// strtod / printf("%.3f") with fast paths for the numbers corridor and trajectory files hold (corridor.cpp); same
// values, same text (tests/test_text_io.py)
double parse_double(const char *p, const char **end);
enum { FORMAT_3_MAX = 336 };         // "-" + 309 digits + ".000" + NUL of the largest double
int format_3(char *out, double v);   // out: FORMAT_3_MAX bytes
// The reference's trajectory file (trp_wrapper.cpp:288-301 / cub_wrapper.cpp:268-279): one row per sample, fixed, 3
// decimals: i*delta s l ds dl dds ddl.  false when the file cannot be written.
bool write_trajectory_file(const std::string &path, int np, double delta, const double *s, const double *l, const double *ds,
                           const double *dl, const double *dds, const double *ddl);

// CorridorGeneration + CorridorSplit for one obstacle's bounds.
std::vector<Segment> extract_segments(int variant, int N, double delta, const Bounds &sb, const Bounds &lb);

// CollisionCheck over all obstacles' segment lists.  Returns false when no segment
// survives (the reference's temp.size()-1 underflow).
bool select_segments(int variant, double delta, const std::vector<std::vector<Segment>> &lists,
                     const std::vector<double> &s_ref, const std::vector<double> &l_ref, std::vector<Segment> &out);

}  // namespace btrapz
#endif
