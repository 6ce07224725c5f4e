// btrapz_lean.hip -- the cold instantiations of the two-wavefronts-per-SIMD solve (btrapz_lean_body.h).
// Template arguments: ORDERED (candidates through a.order: ragged batches, hint classes, the pre-pass's lists, resume
// lists), CAPPED / RESUME (the two launches of btrapz_options.cap_iter), WARM.  The ordered instantiations serve uniform
// and ragged batches alike (the segment count comes from the bucket or from a.bucket_S at run time) and give a uniform
// batch the bits the memory-order ones give it.
#include "btrapz_lean_body.h"

namespace btrapz {

LEAN_INSTANCE(ipm_solve_lean_kernel, false, false, false)                 // uniform, memory order
LEAN_INSTANCE(ipm_solve_lean_ordered_kernel, true, false, false)          // through a.order: ragged batches, hint classes, the pre-pass
LEAN_INSTANCE(ipm_solve_lean_capped_kernel, false, true, false)           // first launch, uniform, memory order
LEAN_INSTANCE(ipm_solve_lean_capped_ordered_kernel, true, true, false)    // first launch through a.order
LEAN_INSTANCE(ipm_solve_lean_resume_kernel, true, false, true)            // second launch (always through its lists)

}  // namespace btrapz
