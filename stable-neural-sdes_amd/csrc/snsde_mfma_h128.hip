// MFMA fast-path kernels instantiated for hidden size 128 (forward and adjoint, both tile flavours).
#include "snsde_m4_kernel.h"

namespace snsde_mfma {

int dispatch_fwd_h128(const MfmaPlan& p, const MfmaArgs& a, hipStream_t st) {
    return p.FL ? dispatch_io<128, 1>(p, a, st) : dispatch_io<128, 0>(p, a, st);
}

int dispatch_rev_h128(const RevPlan& p, const RevArgs& a, hipStream_t st) {
    return p.FL ? dispatch_rev<128, 1>(p, a, st) : dispatch_rev<128, 0>(p, a, st);
}

int dispatch_lean_h128(const MfmaPlan& p, const MfmaArgs& a, hipStream_t st) { return dispatch_lean<128>(p, a, st); }

}  // namespace snsde_mfma
