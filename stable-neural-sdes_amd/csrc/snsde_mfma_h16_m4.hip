// MFMA fast-path kernels instantiated for hidden size 16: 4-row tiles of the general forward kernel (SRK, diffusion
// nets, exact-order first layer, and every Euler / Milstein configuration: the lean kernel covers H = 32 / 64 / 128 only).
#include "snsde_mfma_kernels.h"

namespace snsde_mfma {

int dispatch_fwd_m4_h16(const MfmaPlan& p, const MfmaArgs& a, hipStream_t st) { return dispatch_io<16, 1>(p, a, st); }

}  // namespace snsde_mfma
