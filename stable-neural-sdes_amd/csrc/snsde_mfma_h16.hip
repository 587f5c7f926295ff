// MFMA fast-path kernels instantiated for hidden size 16: 16-row tiles (forward) and the adjoint kernels.
#include "snsde_mfma_kernels.h"

namespace snsde_mfma {

int dispatch_fwd_m16_h16(const MfmaPlan& p, const MfmaArgs& a, hipStream_t st) { return dispatch_io<16, 0>(p, a, st); }

int dispatch_rev_h16(const RevPlan& p, const RevArgs& a, hipStream_t st) {
    return p.FL ? dispatch_rev<16, 1>(p, a, st) : dispatch_rev<16, 0>(p, a, st);
}

}  // namespace snsde_mfma
