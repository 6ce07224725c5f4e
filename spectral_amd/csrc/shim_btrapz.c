/* libbtrapz.so: the reference's older trapezoid build (strings: .../src/c1.txt ->
 * .../slt_3d.txt, no iteration suffix).  Its sources are not in the reference tree
 * (FormNewCorridors is declared, solve_3d.h:54, and defined nowhere), so this library keeps
 * that build's symbol and file pair and runs the current trapezoid path on them. */
#include <stdlib.h>
#include "../../include/btrapz_hip.h"
double find_traj(Params *p) {
  const char *in = getenv("BTRAPZ_INPUT"), *out = getenv("BTRAPZ_OUTPUT");
  return btrapz_find_traj(BTRAPZ_TRAPEZOID, in ? in : "/home/srujan_d/RISS/code/btrapz/src/c1.txt",
                          out ? out : "/home/srujan_d/RISS/code/riss/src/btrapz/src/slt_3d.txt", p);
}
