// MFMA fast-path kernels instantiated for hidden size 64 (forward and adjoint, both tile flavours).
#include "snsde_m4_kernel.h"

namespace snsde_mfma {

int dispatch_fwd_h64(const MfmaPlan& p, const MfmaArgs& a, hipStream_t st) {
    return p.FL ? dispatch_io<64, 1>(p, a, st) : dispatch_io<64, 0>(p, a, st);
}

int dispatch_rev_h64(const RevPlan& p, const RevArgs& a, hipStream_t st) {
    return p.FL ? dispatch_rev<64, 1>(p, a, st) : dispatch_rev<64, 0>(p, a, st);
}

int dispatch_lean_h64(const MfmaPlan& p, const MfmaArgs& a, hipStream_t st) { return dispatch_lean<64>(p, a, st); }

}  // namespace snsde_mfma
