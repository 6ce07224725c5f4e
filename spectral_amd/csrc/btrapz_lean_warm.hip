// btrapz_lean_warm.hip -- the warm-start instantiations of the two-wavefronts-per-SIMD solve (btrapz_lean_body.h):
// btrapz_warm's joint states x0 and multipliers lam0 of an earlier solve as the start, one cold restart inside the kernel
// for a group whose guess does not pay off, multipliers and joint states of the result written for the next solve.
#include "btrapz_lean_body.h"

namespace btrapz {

LEAN_INSTANCE(ipm_solve_lean_warm_kernel, false, false, false, true)          // uniform, memory order
LEAN_INSTANCE(ipm_solve_lean_warm_ordered_kernel, true, false, false, true)   // through a.order: ragged batches, hint classes

}  // namespace btrapz
