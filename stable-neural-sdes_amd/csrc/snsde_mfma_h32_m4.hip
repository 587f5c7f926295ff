// MFMA fast-path kernels instantiated for hidden size 32: 4-row tiles of the general forward kernel (SRK, diffusion
// nets, exact-order first layer).
#include "snsde_mfma_kernels.h"

namespace snsde_mfma {

int dispatch_fwd_m4_h32(const MfmaPlan& p, const MfmaArgs& a, hipStream_t st) { return dispatch_io<32, 1>(p, a, st); }

}  // namespace snsde_mfma
