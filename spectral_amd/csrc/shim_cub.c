/* libcub.so: the cuboid driver's one exported symbol (src/cub_wrapper.cpp:16-19). */
#include <stddef.h>
#include "../../include/btrapz_hip.h"
double find_traj(Params *p) { return btrapz_find_traj(BTRAPZ_CUBOID, NULL, NULL, p); }
