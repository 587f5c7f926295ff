// H = 256 on 4-row tiles with a quarter of the weights resident (two tiles per wave): instantiations and dispatch.
// See snsde_m4s2_kernel.h.  relu fields only (the LipSwish / SiLU variants stay on snsde_m4s_kernel).
#include "snsde_m4s2_kernel.h"

namespace snsde_mfma {

int dispatch_lean_h256_two_tile(const MfmaPlan& p, const MfmaArgs& a, hipStream_t st) {
    const bool save = a.act_save || a.traj || a.dW_out;
    if (p.IO == 0 || a.act != SNSDE_ACT_RELU) return SNSDE_ERR_UNSUPPORTED;
    // instantiated where the two tiles' working set + the resident blocks fit 256 registers without scratch
    // (-Rpass-analysis=kernel-resource-usage, profiles/r06_h256_two_tile_resources.txt); everything else: UNSUPPORTED, the caller
    // falls back to the fully streamed kernel (same results)
#define SNSDE_STREAM2(NH_, KX_) \
    if (p.NHID == NH_ && p.KUXT == KX_) \
        return save ? launch_stream2<CfgS2<NH_, KX_, 1>>(a, st) : launch_stream2<CfgS2<NH_, KX_, 0>>(a, st);
#define SNSDE_STREAM2_INFER(NH_, KX_) \
    if (p.NHID == NH_ && p.KUXT == KX_ && !save) return launch_stream2<CfgS2<NH_, KX_, 0>>(a, st);
#ifdef SNSDE_DEV_SUBSET
    SNSDE_STREAM2(1, 1) SNSDE_STREAM2_INFER(1, 2)
#else
    SNSDE_STREAM2(0, 0) SNSDE_STREAM2(1, 0) SNSDE_STREAM2_INFER(2, 0)
    SNSDE_STREAM2(0, 1) SNSDE_STREAM2(1, 1)
    SNSDE_STREAM2(0, 2) SNSDE_STREAM2_INFER(1, 2)
#endif
#undef SNSDE_STREAM2
#undef SNSDE_STREAM2_INFER
    return SNSDE_ERR_UNSUPPORTED;
}

}  // namespace snsde_mfma
