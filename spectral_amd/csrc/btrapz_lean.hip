// btrapz_lean.hip -- the cold instantiations of the two-wavefronts-per-SIMD solve (btrapz_lean_body.h).
// Template arguments: ORDERED (candidates through a.order: ragged batches, hint classes, resume lists), CAPPED / RESUME
// (the two launches of btrapz_options.cap_iter), SMALL_S (the end-lane fix-up of one and two segments: ragged batches
// only), WARM.  A uniform batch runs kernels WITHOUT the fix-up whether or not it goes through a.order -- a scheduling
// hint or a second launch must not change a result's bits, and the fix-up's code does (other contraction of the sums).
#include "btrapz_lean_body.h"

namespace btrapz {

LEAN_INSTANCE(ipm_solve_lean_kernel, false, false, false, false)                 // uniform, memory order
LEAN_INSTANCE(ipm_solve_lean_hint_kernel, true, false, false, false)             // uniform, hint classes (btrapz_warm.hint)
LEAN_INSTANCE(ipm_solve_lean_ragged_kernel, true, false, false, true)            // ragged (buckets by segment count)
LEAN_INSTANCE(ipm_solve_lean_capped_kernel, false, true, false, false)           // first launch, uniform
LEAN_INSTANCE(ipm_solve_lean_capped_hint_kernel, true, true, false, false)       // first launch, uniform, through a.order (btrapz_options.compact)
LEAN_INSTANCE(ipm_solve_lean_capped_ragged_kernel, true, true, false, true)      // first launch, ragged
LEAN_INSTANCE(ipm_solve_lean_resume_kernel, true, false, true, false)            // second launch, uniform
LEAN_INSTANCE(ipm_solve_lean_resume_ragged_kernel, true, false, true, true)      // second launch, ragged

}  // namespace btrapz
