// MFMA fast-path kernels instantiated for hidden size 128: 16-row tiles (forward) and the adjoint kernels.
#include "snsde_mfma_kernels.h"

namespace snsde_mfma {

int dispatch_fwd_m16_h128(const MfmaPlan& p, const MfmaArgs& a, hipStream_t st) { return dispatch_io<128, 0>(p, a, st); }

int dispatch_rev_h128(const RevPlan& p, const RevArgs& a, hipStream_t st) {
    return p.FL ? dispatch_rev<128, 1>(p, a, st) : dispatch_rev<128, 0>(p, a, st);
}

}  // namespace snsde_mfma
