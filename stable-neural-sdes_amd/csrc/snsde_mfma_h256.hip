// MFMA fast-path kernels instantiated for hidden size 256: 16 waves per workgroup, weights streamed from L2 (the matrices exceed a CU's register file): 16-row tiles (forward) and the adjoint kernels.
#include "snsde_mfma_kernels.h"

namespace snsde_mfma {

int dispatch_fwd_m16_h256(const MfmaPlan& p, const MfmaArgs& a, hipStream_t st) { return dispatch_io<256, 0>(p, a, st); }

int dispatch_rev_h256(const RevPlan& p, const RevArgs& a, hipStream_t st) {
    return p.FL ? dispatch_rev<256, 1>(p, a, st) : dispatch_rev<256, 0>(p, a, st);
}

}  // namespace snsde_mfma
