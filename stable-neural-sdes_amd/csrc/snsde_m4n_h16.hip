// 4-row-tile MFMA kernels for the diffusion nets under SRK / Milstein (snsde_m4n_kernel.h), hidden size 16.
#include "snsde_m4n_mil_rev_kernel.h"

namespace snsde_mfma {

int dispatch_m4n_h16(const MfmaPlan& p, const MfmaArgs& a, hipStream_t st) { return dispatch_m4n<16>(p, a, st); }
int dispatch_m4n_rev_h16(const RevPlan& p, const RevArgs& a, hipStream_t st) { return dispatch_m4n_rev<16>(p, a, st); }
int dispatch_m4n_mil_rev_h16(const RevPlan& p, const RevArgs& a, hipStream_t st) { return dispatch_m4n_mil_rev<16>(p, a, st); }

}  // namespace snsde_mfma
