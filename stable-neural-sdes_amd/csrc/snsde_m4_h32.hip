// Lean 4-row-tile forward kernel (snsde_m4_kernel.h) instantiated for hidden size 32.
#include "snsde_m4_kernel.h"

namespace snsde_mfma {

int dispatch_lean_h32(const MfmaPlan& p, const MfmaArgs& a, hipStream_t st) { return dispatch_lean<32>(p, a, st); }

}  // namespace snsde_mfma
