// Two-tiles-per-wave lean kernel (snsde_m4t_kernel.h) for hidden size 128: four waves, one per SIMD.  Instantiations and dispatch.
#include "snsde_m4t_kernel.h"

namespace snsde_mfma {

int dispatch_lean_h128_two_tile(const MfmaPlan& p, const MfmaArgs& a, hipStream_t st) {
    const bool save = a.act_save || a.traj || a.dW_out;
    if (p.IO == 0 || a.act != SNSDE_ACT_RELU || a.acc_col >= 0) return SNSDE_ERR_UNSUPPORTED;
#define SNSDE_LEAN2(NH_, KX_) \
    if (p.NHID == NH_ && p.KUXT == KX_) \
        return save ? launch_lean2<CfgT<128, NH_, KX_, 1>>(a, st) : launch_lean2<CfgT<128, NH_, KX_, 0>>(a, st);
    SNSDE_LEAN2(1, 2) SNSDE_LEAN2(1, 1)
#undef SNSDE_LEAN2
    return SNSDE_ERR_UNSUPPORTED;
}

}  // namespace snsde_mfma
