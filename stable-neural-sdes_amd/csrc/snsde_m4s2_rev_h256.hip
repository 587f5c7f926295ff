// Two-tile adjoint of the H = 256 Euler / Milstein solve (snsde_m4s2_rev_kernel.h): instantiations and dispatch.
#include "snsde_m4s2_rev_kernel.h"

namespace snsde_mfma {

int dispatch_rev_h256_two_tile(const RevPlan& p, const RevArgs& a, hipStream_t st) {
    // the reference's own fields on 4-row tiles, elementwise diffusions, y-dependent drifts (everything else: the general kernel)
    if (!p.FL || p.SRK || p.IO0 || p.NN != 0 || a.act_fn != 0 || a.f_out != 0 || a.g_out != 0 || a.acc_col >= 0) return SNSDE_ERR_UNSUPPORTED;
#ifdef SNSDE_DEV_SUBSET
    if (p.NHID == 1 && !p.GEO) return launch_rev2<CfgS2R<1, 0>>(a, st);
#else
#define SNSDE_R2(NH_) if (p.NHID == NH_) return p.GEO ? launch_rev2<CfgS2R<NH_, 1>>(a, st) : launch_rev2<CfgS2R<NH_, 0>>(a, st);
    SNSDE_R2(0) SNSDE_R2(1) SNSDE_R2(2)
#undef SNSDE_R2
#endif
    return SNSDE_ERR_UNSUPPORTED;
}

}  // namespace snsde_mfma
