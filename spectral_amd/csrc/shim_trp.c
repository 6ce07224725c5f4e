/* libtrp.so: the trapezoid-prism driver's one exported symbol (src/trp_wrapper.cpp:16-20). */
#include <stddef.h>
#include "../../include/btrapz_hip.h"
double find_traj(Params *p) { return btrapz_find_traj(BTRAPZ_TRAPEZOID, NULL, NULL, p); }
