// Streamed-weight lean kernel (H = 256): instantiations and dispatch.  See snsde_m4s_kernel.h.
#include "snsde_m4s_kernel.h"

namespace snsde_mfma {

int dispatch_lean_h256_two_tile(const MfmaPlan& p, const MfmaArgs& a, hipStream_t st);   // snsde_m4s2_h256.hip

int dispatch_lean_h256(const MfmaPlan& p, const MfmaArgs& a, hipStream_t st, bool stream_all) {
    const bool save = a.act_save || a.traj || a.dW_out;
    if (p.IO == 0) return SNSDE_ERR_UNSUPPORTED;
    // round 6: eight waves of two tiles with a quarter of every layer resident (snsde_m4s2_kernel.h, bit-identical results);
    // SNSDE_FLAG_STREAM_ALL keeps the sixteen-wave kernel below (A/B measurements, the bit-identity test)
    if (!stream_all && a.act == SNSDE_ACT_RELU) {
        const int rc = dispatch_lean_h256_two_tile(p, a, st);
        if (rc != SNSDE_ERR_UNSUPPORTED) return rc;
    }
    if (a.act != SNSDE_ACT_RELU) {      // tutorial fields (LipSwish / SiLU): inference only
        if (save) return SNSDE_ERR_UNSUPPORTED;
#define SNSDE_STREAM_ACT(NH_, KX_) if (p.NHID == NH_ && p.KUXT == KX_) return launch_stream<CfgS<NH_, KX_, 0, 1>>(a, st);
#ifndef SNSDE_DEV_SUBSET
        SNSDE_STREAM_ACT(0, 1) SNSDE_STREAM_ACT(1, 1) SNSDE_STREAM_ACT(2, 1)
        SNSDE_STREAM_ACT(0, 2) SNSDE_STREAM_ACT(1, 2) SNSDE_STREAM_ACT(2, 2)
        SNSDE_STREAM_ACT(1, 3) SNSDE_STREAM_ACT(2, 3)
#endif
#undef SNSDE_STREAM_ACT
        return SNSDE_ERR_UNSUPPORTED;
    }
#define SNSDE_STREAM(NH_, KX_) \
    if (p.NHID == NH_ && p.KUXT == KX_) \
        return save ? launch_stream<CfgS<NH_, KX_, 1>>(a, st) : launch_stream<CfgS<NH_, KX_, 0>>(a, st);
#ifdef SNSDE_DEV_SUBSET
    SNSDE_STREAM(1, 1) SNSDE_STREAM(1, 2)
#else
    SNSDE_STREAM(0, 0) SNSDE_STREAM(1, 0) SNSDE_STREAM(2, 0)
    SNSDE_STREAM(0, 1) SNSDE_STREAM(1, 1) SNSDE_STREAM(2, 1)
    SNSDE_STREAM(0, 2) SNSDE_STREAM(1, 2) SNSDE_STREAM(2, 2)
    SNSDE_STREAM(1, 3) SNSDE_STREAM(2, 3)
#endif
#undef SNSDE_STREAM
    return SNSDE_ERR_UNSUPPORTED;
}

}  // namespace snsde_mfma
