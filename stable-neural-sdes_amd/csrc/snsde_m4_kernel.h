// Lean 4-row-tile forward kernel (Euler / Milstein, elementwise diffusions) of the MFMA fast path.
//
// Why a second M4 kernel: on gfx950 v_mfma_f32_4x4x1_16b_f32 (like every f32-input MFMA) issues through the SIMD's
// VALU port and does NOT overlap with other VALU work of either wave of the SIMD (tools/ubench/mfma4_issue*.hip:
// MFMA + k VALU ops cost 8 + ~5k cycles, alone or with two waves per SIMD; 16x16x4 behaves the same).  A solver step of
// the 4-row tile therefore costs  (MFMA count) x 8 + (VALU count) x ~4.5 cycles per SIMD  plus whatever latency of the
// three layer hand-offs (LDS write -> barrier -> LDS read) is left exposed.  This kernel is the general M4 kernel
// (snsde_mfma_kernel<.., FL = 1>) re-cut for exactly that cost model:
//   * the [sin t, cos t] features share the control path's k-block ([X(t) | sin t, cos t | 0..] in LDS) instead of
//     occupying a 16-wide block of their own: 4 MFMAs per wave-step fewer;
//   * layer biases enter through the accumulator operand of each chain's first MFMA (k-slot 0 lanes hold the bias):
//     no bias add;
//   * the diffusion g(t_n, y_n) and y_n + g dW_n are evaluated at the TOP of the step (they need only y_n) while the
//     first layer's B operands are in flight; only  reduce -> tanh -> fma  follows the last GEMM;
//   * tanh = copysign((1 - t) / (1 + t), x), t = 2^(-2 log2e |x|): 7 VALU ops, absolute error <= ~4e-8 (the |x| < 0.25
//     polynomial branch of fast_tanh bought relative accuracy the Euler update cannot see); nan_to_num of the raw
//     diffusion folds into it (NaN -> v_max(NaN, 0) = 0, +-inf -> t = 0 -> +-1 = tanh(sigmoid(theta) FLT_MAX));
//   * the y-independent work of the NEXT step (spline value X(t_{n+1}), time features, Philox block, dW / diffusion
//     table / coefficient fetches) sits in the windows right after the layer barriers, where both waves of a SIMD
//     would otherwise wait for their B operands; the spline items are evaluated by one wave per SIMD only;
//   * the step-table row is consumed as broadcast LDS reads in VGPRs (no v_readfirstlane except the output count).
// Numerics: same MFMA chains (k order, two accumulators) as the general kernel; differences are the tanh form and the
// order of the final update  y + g dW + f h  (was y + f h + g dW): both inside the parity tolerance (tests/helpers.py).
// Reference semantics: benchmark_classification/models_sde/neuralsde.py:295-307 (f, g), SURVEY.md A3-A6 (stepping).
#pragma once
#include "snsde_mfma_kernels.h"

namespace snsde_mfma {

template <int H_, int NHID_, int KUXT_, int YIN_, int SAVE_, int ACT_ = 0, int ACC_ = 0>
struct CfgL {
    static constexpr bool ACC = ACC_ != 0;      // path-integral accumulator column (snsde.h: kl_column1): the LatentSDE mapping's instantiations
    static constexpr bool SWISH = ACT_ != 0;    // hidden activation scale * x * sigmoid(x) (LipSwish / SiLU) instead of relu
    static constexpr int H = H_, NHID = NHID_, KUXT = KUXT_;
    static constexpr bool YIN = YIN_ != 0;      // the first layer reads y (every input_option but 0)
    static constexpr bool SAVE = SAVE_ != 0;    // training / diagnostics outputs: act_save, traj, dW_out
    static constexpr int NW = H / 16;
    static constexpr int NT = NW * 64;
    static constexpr int WPS = NW >= 4 ? NW / 4 : 1;
    static constexpr int KUH = H / 16;
    static constexpr int LDY = ld_for(16 * KUH, 16);
    static constexpr int LDX = ld_for(16 * (KUXT > 0 ? KUXT : 1), 16);
    static constexpr int LDA = LDY;
    static constexpr int NLAYER = NHID + 2;                   // bias rows: first, hidden.., out
    static constexpr int NSAVE = NHID + 2 + (SWISH ? NHID + 1 : 0);   // outputs: first, hidden.., pre-tanh drift [, the pre-activations]
    static constexpr int ZSLOT = NHID + 1;
    static constexpr int PRE0 = NHID + 2;                     // SWISH: slot of the first layer's pre-activation (then hidden..)
    static constexpr int ZB = 4;                              // Philox calls generated together per element
    static constexpr int ZSTASH = 4 * ZB * 64;                // floats per wave
    static constexpr int ROWCH = 128;
    static constexpr int XI = KUXT > 0 ? (4 * 16 * KUXT + NT - 1) / NT : 1;   // [X(t) | sin t, cos t] entries per lane (4 rows, spread over ALL waves)
    static constexpr int RS = 8;                              // floats per step in the kernel's own table: quad A, quad B
    static constexpr int ACCF = 64;                           // per-wave row sums of the path-integral accumulator (4 x NW <= 64)
    static constexpr int LDS_FLOATS = 4 * (LDY + 2 * LDX + 2 * LDA) + (ROWCH + 3) * RS + NW * ZSTASH + ACCF;
};

// tanh(x) = copysign((1 - t) / (1 + t), x), t = 2^(-2 log2(e) |x|) in (0, 1]: no cancellation beyond the rounding of t
// (absolute error <= ~4e-8, saturates to +-1 for large |x| and for +-inf).  NANZ: a NaN argument gives +-0 (the
// reference's nan_to_num(raw) = 0 followed by tanh), through v_max_f32(NaN, 0) = 0.
template <bool NANZ>
__device__ __forceinline__ float lean_tanh(float x) {
    const float t = __builtin_amdgcn_exp2f(fabsf(x) * -2.8853900817779268f);
    float q = (1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t);
    if constexpr (NANZ) q = __builtin_fmaxf(q, 0.0f);
    return __builtin_copysignf(q, x);
}

// The same with the odd Taylor polynomial to x^7 below |x| = 0.125 (truncation < 2e-10 relative): relative accuracy at
// small arguments, where the exponential form only has absolute accuracy.  7 more VALU ops.
template <bool NANZ>
__device__ __forceinline__ float lean_tanh_rel(float x) {
    const float x2 = x * x;
    float p = fmaf(x2, -0.053968253968253971f, 0.13333333333333333f);     // -17/315, 2/15
    p = fmaf(x2, p, -0.33333333333333333f);
    p = fmaf(x * x2, p, x);
    const float q = lean_tanh<NANZ>(x);
    return fabsf(x) < 0.125f ? p : q;
}

// hidden activation of the tutorial fields: scale * x * sigmoid(x)  (LipSwish: scale = 0.909, SiLU: 1)
__device__ __forceinline__ float lean_swish(float x, float scale) {
    const float e = __builtin_amdgcn_exp2f(x * -1.4426950408889634f);
    return (x * scale) * __builtin_amdgcn_rcpf(1.0f + e);
}

template <int KU>
__device__ __forceinline__ void lean_load_w(float (&w)[KU * 4], const float* __restrict__ g, int wave, int lane) {
#pragma unroll
    for (int u = 0; u < KU; ++u) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(g + (((size_t)wave * KU + u) * 64 + lane) * 4);
        w[4 * u] = v[0]; w[4 * u + 1] = v[1]; w[4 * u + 2] = v[2]; w[4 * u + 3] = v[3];
    }
}

// B operands of one layer: lanes 0-15 read (k-slot s, row r) = 16 bytes per 16-wide k-block; the MFMAs broadcast them
// (blgp:4), so the other lanes' registers are never read.  The reads are issued from inline asm with EXEC narrowed to
// lanes 0-15 (no branch), and waited for by lean_wait_b() just before the MFMAs: hipcc's own s_waitcnt insertion loses
// count across the exec-masked branch it would otherwise generate and drains the LDS queue before ANY instruction placed
// between the reads and the MFMAs (the filler work this kernel hides in that window).  Its own counts stay safe: LDS
// returns in order, and an s_waitcnt computed without these reads only waits longer.
// Global loads of the step loop are issued from asm as well (one dword, scalar base + 32-bit lane offset) and waited for
// ONCE per step by lean_vm_wait(): hipcc cannot count vmcnt across the loop's control flow and would otherwise drain the
// vector-memory queue (s_waitcnt vmcnt(0)) right after the prefetches are issued.
// (s_nop 4: a VALU-written SGPR, e.g. v_readfirstlane, needs 5 wait states before a VMEM instruction reads it, and the
// hazard recognizer does not look inside asm.)
__device__ __forceinline__ void lean_gload(float& dst, uint32_t voff, const float* sbase) {
    asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(lean_uniform(sbase)) : "memory");
}
// store through a scalar base + 32-bit lane offset as well: the training-mode saves then need no 64-bit address VGPRs
__device__ __forceinline__ void lean_gstore(float v, uint32_t voff, const float* sbase) {
    // (no "memory" clobber: these buffers are never read back by the kernel, and the clobber would pin every LDS access of the
    //  step on one side of the store)
    asm volatile("s_nop 4\n\tglobal_store_dword %1, %0, %2" :: "v"(v), "v"(voff), "s"(lean_uniform(sbase)));
}
__device__ __forceinline__ void lean_gload4(float& d0, float& d1, float& d2, float& d3, uint32_t v0, uint32_t v1, uint32_t v2,
                                            uint32_t v3, const float* sbase) {
    asm volatile("s_nop 4\n\tglobal_load_dword %0, %4, %8\n\tglobal_load_dword %1, %5, %8\n\tglobal_load_dword %2, %6, %8\n\t"
                 "global_load_dword %3, %7, %8"
                 : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(lean_uniform(sbase)) : "memory");
}
template <int KU> struct LeanB { f32x4 v[KU]; };
__device__ __forceinline__ void lean_read_b(uint32_t a, LeanB<1>& b) {
    asm volatile("s_mov_b64 exec, 0xffff\n\tds_read_b128 %0, %[a]\n\ts_mov_b64 exec, -1" : "=&v"(b.v[0]) : [a] "v"(a));
}
__device__ __forceinline__ void lean_read_b(uint32_t a, LeanB<2>& b) {
    asm volatile("s_mov_b64 exec, 0xffff\n\tds_read_b128 %0, %[a]\n\tds_read_b128 %1, %[a] offset:64\n\ts_mov_b64 exec, -1"
                 : "=&v"(b.v[0]), "=&v"(b.v[1]) : [a] "v"(a));
}
__device__ __forceinline__ void lean_read_b(uint32_t a, LeanB<3>& b) {
    asm volatile("s_mov_b64 exec, 0xffff\n\tds_read_b128 %0, %[a]\n\tds_read_b128 %1, %[a] offset:64\n\tds_read_b128 %2, %[a] offset:128\n\t"
                 "s_mov_b64 exec, -1" : "=&v"(b.v[0]), "=&v"(b.v[1]), "=&v"(b.v[2]) : [a] "v"(a));
}
__device__ __forceinline__ void lean_read_b(uint32_t a, LeanB<4>& b) {
    asm volatile("s_mov_b64 exec, 0xffff\n\tds_read_b128 %0, %[a]\n\tds_read_b128 %1, %[a] offset:64\n\tds_read_b128 %2, %[a] offset:128\n\t"
                 "ds_read_b128 %3, %[a] offset:192\n\ts_mov_b64 exec, -1"
                 : "=&v"(b.v[0]), "=&v"(b.v[1]), "=&v"(b.v[2]), "=&v"(b.v[3]) : [a] "v"(a));
}
__device__ __forceinline__ void lean_read_b(uint32_t a, LeanB<5>& b) {
    asm volatile("s_mov_b64 exec, 0xffff\n\tds_read_b128 %0, %[a]\n\tds_read_b128 %1, %[a] offset:64\n\tds_read_b128 %2, %[a] offset:128\n\t"
                 "ds_read_b128 %3, %[a] offset:192\n\tds_read_b128 %4, %[a] offset:256\n\ts_mov_b64 exec, -1"
                 : "=&v"(b.v[0]), "=&v"(b.v[1]), "=&v"(b.v[2]), "=&v"(b.v[3]), "=&v"(b.v[4]) : [a] "v"(a));
}
__device__ __forceinline__ void lean_read_b(uint32_t a, LeanB<6>& b) {
    asm volatile("s_mov_b64 exec, 0xffff\n\tds_read_b128 %0, %[a]\n\tds_read_b128 %1, %[a] offset:64\n\tds_read_b128 %2, %[a] offset:128\n\t"
                 "ds_read_b128 %3, %[a] offset:192\n\tds_read_b128 %4, %[a] offset:256\n\tds_read_b128 %5, %[a] offset:320\n\t"
                 "s_mov_b64 exec, -1"
                 : "=&v"(b.v[0]), "=&v"(b.v[1]), "=&v"(b.v[2]), "=&v"(b.v[3]), "=&v"(b.v[4]), "=&v"(b.v[5]) : [a] "v"(a));
}
__device__ __forceinline__ void lean_read_b(uint32_t a, LeanB<8>& b) {
    asm volatile("s_mov_b64 exec, 0xffff\n\tds_read_b128 %0, %[a]\n\tds_read_b128 %1, %[a] offset:64\n\tds_read_b128 %2, %[a] offset:128\n\t"
                 "ds_read_b128 %3, %[a] offset:192\n\tds_read_b128 %4, %[a] offset:256\n\tds_read_b128 %5, %[a] offset:320\n\t"
                 "ds_read_b128 %6, %[a] offset:384\n\tds_read_b128 %7, %[a] offset:448\n\ts_mov_b64 exec, -1"
                 : "=&v"(b.v[0]), "=&v"(b.v[1]), "=&v"(b.v[2]), "=&v"(b.v[3]), "=&v"(b.v[4]), "=&v"(b.v[5]), "=&v"(b.v[6]), "=&v"(b.v[7])
                 : [a] "v"(a));
}
// the same into registers that live across the step loop ("+v": read in place, no copy at the back edge)
__device__ __forceinline__ void lean_read_b_carried(uint32_t a, LeanB<1>& b) {
    asm volatile("s_mov_b64 exec, 0xffff\n\tds_read_b128 %0, %[a]\n\ts_mov_b64 exec, -1" : "+v"(b.v[0]) : [a] "v"(a));
}
__device__ __forceinline__ void lean_read_b_carried(uint32_t a, LeanB<2>& b) {
    asm volatile("s_mov_b64 exec, 0xffff\n\tds_read_b128 %0, %[a]\n\tds_read_b128 %1, %[a] offset:64\n\ts_mov_b64 exec, -1"
                 : "+v"(b.v[0]), "+v"(b.v[1]) : [a] "v"(a));
}
__device__ __forceinline__ void lean_read_b_carried(uint32_t a, LeanB<3>& b) {
    asm volatile("s_mov_b64 exec, 0xffff\n\tds_read_b128 %0, %[a]\n\tds_read_b128 %1, %[a] offset:64\n\tds_read_b128 %2, %[a] offset:128\n\t"
                 "s_mov_b64 exec, -1" : "+v"(b.v[0]), "+v"(b.v[1]), "+v"(b.v[2]) : [a] "v"(a));
}
__device__ __forceinline__ void lean_read_b_carried(uint32_t a, LeanB<4>& b) {
    asm volatile("s_mov_b64 exec, 0xffff\n\tds_read_b128 %0, %[a]\n\tds_read_b128 %1, %[a] offset:64\n\tds_read_b128 %2, %[a] offset:128\n\t"
                 "ds_read_b128 %3, %[a] offset:192\n\ts_mov_b64 exec, -1"
                 : "+v"(b.v[0]), "+v"(b.v[1]), "+v"(b.v[2]), "+v"(b.v[3]) : [a] "v"(a));
}
__device__ __forceinline__ void lean_read_b_carried(uint32_t a, LeanB<5>& b) {
    asm volatile("s_mov_b64 exec, 0xffff\n\tds_read_b128 %0, %[a]\n\tds_read_b128 %1, %[a] offset:64\n\tds_read_b128 %2, %[a] offset:128\n\t"
                 "ds_read_b128 %3, %[a] offset:192\n\tds_read_b128 %4, %[a] offset:256\n\ts_mov_b64 exec, -1"
                 : "+v"(b.v[0]), "+v"(b.v[1]), "+v"(b.v[2]), "+v"(b.v[3]), "+v"(b.v[4]) : [a] "v"(a));
}
__device__ __forceinline__ void lean_read_b_carried(uint32_t a, LeanB<6>& b) {
    asm volatile("s_mov_b64 exec, 0xffff\n\tds_read_b128 %0, %[a]\n\tds_read_b128 %1, %[a] offset:64\n\tds_read_b128 %2, %[a] offset:128\n\t"
                 "ds_read_b128 %3, %[a] offset:192\n\tds_read_b128 %4, %[a] offset:256\n\tds_read_b128 %5, %[a] offset:320\n\t"
                 "s_mov_b64 exec, -1"
                 : "+v"(b.v[0]), "+v"(b.v[1]), "+v"(b.v[2]), "+v"(b.v[3]), "+v"(b.v[4]), "+v"(b.v[5]) : [a] "v"(a));
}
__device__ __forceinline__ void lean_read_b_carried(uint32_t a, LeanB<8>& b) {
    asm volatile("s_mov_b64 exec, 0xffff\n\tds_read_b128 %0, %[a]\n\tds_read_b128 %1, %[a] offset:64\n\tds_read_b128 %2, %[a] offset:128\n\t"
                 "ds_read_b128 %3, %[a] offset:192\n\tds_read_b128 %4, %[a] offset:256\n\tds_read_b128 %5, %[a] offset:320\n\t"
                 "ds_read_b128 %6, %[a] offset:384\n\tds_read_b128 %7, %[a] offset:448\n\ts_mov_b64 exec, -1"
                 : "+v"(b.v[0]), "+v"(b.v[1]), "+v"(b.v[2]), "+v"(b.v[3]), "+v"(b.v[4]), "+v"(b.v[5]), "+v"(b.v[6]), "+v"(b.v[7])
                 : [a] "v"(a));
}
#undef SNSDE_RD
// The fragments are tied through the wait ("+v"): nothing that reads them can be scheduled above it.  Y = LDS operations
// this wave issued AFTER the reads of `b` and that may stay in flight (LDS returns in order); pairs of k-blocks are
// released as they land.
template <int CNT> __device__ __forceinline__ void lean_wait2(f32x4& x, f32x4& y) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x), "+v"(y) : "n"(CNT < 15 ? CNT : 15));
}
template <int CNT> __device__ __forceinline__ void lean_wait1(f32x4& x) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "n"(CNT < 15 ? CNT : 15));
}
// c/d += W . b over KU 16-wide k-blocks, two accumulator chains; k-blocks are consumed in pairs as their reads land
// (the sched_barrier keeps hipcc from collecting all the waits in front of the first MFMA)
template <int Y, int KU, int U>
__device__ __forceinline__ void lean_gemm_from(const float (&w)[KU * 4], LeanB<KU>& b, f32x4& c, f32x4& d) {
    if constexpr (U < KU) {
        if constexpr (U + 1 < KU) lean_wait2<Y + KU - 2 - U>(b.v[U], b.v[U + 1]);
        else lean_wait1<Y>(b.v[U]);
#pragma unroll
        for (int uu = U; uu < U + 2 && uu < KU; ++uu) {
            c = __builtin_amdgcn_mfma_f32_4x4x1f32(w[4 * uu], b.v[uu][0], c, 0, 0, 4);
            d = __builtin_amdgcn_mfma_f32_4x4x1f32(w[4 * uu + 1], b.v[uu][1], d, 0, 0, 4);
            c = __builtin_amdgcn_mfma_f32_4x4x1f32(w[4 * uu + 2], b.v[uu][2], c, 0, 0, 4);
            d = __builtin_amdgcn_mfma_f32_4x4x1f32(w[4 * uu + 3], b.v[uu][3], d, 0, 0, 4);
        }
        __builtin_amdgcn_sched_barrier(0);
        lean_gemm_from<Y, KU, U + 2>(w, b, c, d);
    }
}
template <int Y, int KU>
__device__ __forceinline__ void lean_gemm(const float (&w)[KU * 4], LeanB<KU>& b, f32x4& c, f32x4& d) {
    lean_gemm_from<Y, KU, 0>(w, b, c, d);
}

// Cycle timeline (development builds, -DLEAN_TRACE): s_memtime stamps at LT(i) are left in flight (no s_waitcnt) and collected
// in three groups behind the step's barriers (whose own s_waitcnt lgkmcnt(0) has already drained the queue there), so they do
// not drain the LDS queue the way TRACE() does.  The kernel has no registers to spare (251 of 256 VGPRs, SGPRs at the limit, and
// loop-carried scalars of this loop are VGPRs to hipcc), so nothing is carried: each collected stamp's low word is added into a
// per-lane LDS slot with a fire-and-forget ds_add_u32 (conflict-free, 12 x NT words behind the kernel's own LDS); phase i =
// sum_i - sum_{i-1} (mod 2^32); the wrap-around phase 9 -> 0 from the sums of stamp 0 over steps >= 1 and stamp 9 over steps
// <= N - 2 (slots 10 / 11).  Cost: 2 instructions per stamp and step (the traced kernel runs ~15 % slower than the plain one).
#ifdef LEAN_TRACE
#define LT_DECL unsigned long long lt[10]; uint32_t* const ltl = reinterpret_cast<uint32_t*>(lds + CF::LDS_FLOATS) + tid; \
    for (int i_ = 0; i_ < 12; ++i_) ltl[i_ * NT] = 0u;
#define LT(i) asm volatile("s_memtime %0" : "=s"(lt[i]));
#define LT_ADD(slot, i) __hip_atomic_fetch_add(ltl + (slot) * NT, (uint32_t)lt[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#define LT_COLLECT_A { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(lt[0]), "+s"(lt[1]), "+s"(lt[2]), "+s"(lt[3])); \
    LT_ADD(0, 0) LT_ADD(1, 1) LT_ADD(2, 2) LT_ADD(3, 3) if (n > 0) LT_ADD(10, 0) }
#define LT_COLLECT_B { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(lt[4]), "+s"(lt[5]), "+s"(lt[6])); LT_ADD(4, 4) LT_ADD(5, 5) LT_ADD(6, 6) }
#define LT_COLLECT { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(lt[7]), "+s"(lt[8]), "+s"(lt[9])); LT_ADD(7, 7) LT_ADD(8, 8) LT_ADD(9, 9) \
    if (n + 1 < N) LT_ADD(11, 9) }
#define LT_LDS_EXTRA (12 * CF::NT * sizeof(uint32_t))
#else
#define LT_DECL
#define LT(i)
#define LT_COLLECT_A
#define LT_COLLECT_B
#define LT_COLLECT
#define LT_LDS_EXTRA 0
#endif

#ifndef LEAN_TANH_F
#define LEAN_TANH_F lean_tanh_rel<false>
#endif
#ifndef LEAN_TANH_G
#define LEAN_TANH_G lean_tanh_rel<true>
#endif

template <class CF>
__global__ void __launch_bounds__(CF::NT, CF::WPS) snsde_m4_kernel(MfmaArgs a) {
    constexpr int H = CF::H, NT = CF::NT, NHID = CF::NHID, KUH = CF::KUH, KUXT = CF::KUXT;
    constexpr int LDY = CF::LDY, LDX = CF::LDX, LDA = CF::LDA, RS = CF::RS;
    constexpr bool YIN = CF::YIN, SAVE = CF::SAVE;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ybuf = lds;                       // [4][LDY]  y
    float* xbuf = ybuf + 4 * LDY;            // [2][4][LDX]  X(t) (xc) | sin t, cos t | 0..   (step parity)
    float* bufA = xbuf + 8 * LDX;            // [4][LDA]
    float* bufB = bufA + 4 * LDA;            // [4][LDA]
    float* rowtab = bufB + 4 * LDA;          // [ROWCH + 3][RS]  step i: (h_i, sqrt h_{i+1}, -, - | sin t, cos t, frac of step i+1, idx of step i+2)
    float* zstash_all = rowtab + (CF::ROWCH + 3) * RS;
    float* accbuf = zstash_all + CF::NW * CF::ZSTASH;      // [NW][4]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 3, s = (lane >> 2) & 3, q = lane >> 4;
    const int fo = wave * 16 + 4 * q + s;                 // the state / activation element this lane owns (row r)
    const int row0 = blockIdx.x * 4;
    const int B = a.B, C = a.C, N = a.N;
    const int row = row0 + r;
    const bool row_ok = row < B;
    const int rowc = row_ok ? row : B - 1;
    const size_t BH = (size_t)B * H;
    const size_t goff = (size_t)rowc * H + fo;
    float* zstash = zstash_all + wave * CF::ZSTASH;
    const uint32_t fo4 = (uint32_t)(fo * sizeof(float)), goff4 = (uint32_t)(goff * sizeof(float));
    const int xc = a.lean_xc;                              // control channels in the xt block (0: drift without X)
    const bool time_on = a.lean_time != 0, geo = a.lean_geo != 0;
    const float act_scale = a.act == SNSDE_ACT_LIPSWISH ? 0.909f : 1.0f;
    const int f_out = a.f_out;                             // SNSDE_DRIFT_*: tanh(z) | z | z * y
    const bool g_raw = a.g_out == SNSDE_DIFFUSION_RAW;     // g = raw instead of tanh(sigmoid(theta) nan_to_num(raw))

    // ---- resident weights, bias fragments ---------------------------------------------------------------------
    int li = 0;
    float wxt[(KUXT > 0 ? KUXT : 1) * 4], wy[KUH * 4], wh[NHID > 0 ? NHID : 1][KUH * 4], wo[KUH * 4];
    if constexpr (KUXT > 0) lean_load_w<KUXT>(wxt, a.ws + a.w_off[li++], wave, lane);
    if constexpr (YIN) lean_load_w<KUH>(wy, a.ws + a.w_off[li++], wave, lane);
#pragma unroll
    for (int l = 0; l < NHID; ++l) lean_load_w<KUH>(wh[l], a.ws + a.w_off[li++], wave, lane);
    lean_load_w<KUH>(wo, a.ws + a.w_off[li++], wave, lane);
    f32x4 bfr[CF::NLAYER];                                 // accumulator init: the bias in the k-slot 0 lanes, 0 elsewhere
#pragma unroll
    for (int l = 0; l < CF::NLAYER; ++l)
#pragma unroll
        for (int i = 0; i < 4; ++i) bfr[l][i] = (s == 0) ? a.ws[a.bias_off + l * H + wave * 16 + 4 * q + i] : 0.0f;

    // ---- LDS init; the step table is re-cut into the two quads per step this kernel reads -------------------------
    for (int i = tid; i < 4 * (LDY + 2 * LDX + 2 * LDA); i += NT) lds[i] = 0.0f;
    auto fill_rows = [&](int base) {
        for (int i = tid; i < (CF::ROWCH + 3) * RS; i += NT) {
            const int j = i % RS;
            int rr = base + i / RS + (j == 0 ? 0 : (j == 7 ? 2 : 1));
            rr = rr < N - 1 ? rr : N - 1;
            const int src = j == 0 ? 1 : j == 1 ? 6 : j == 4 ? (a.raw_time ? 0 : 2) : j == 5 ? (a.raw_time ? 10 : 3) : j == 6 ? 4 : j == 7 ? 5 : 10;
            rowtab[i] = a.step_tab[(size_t)rr * SNSDE_STEP_STRIDE + src];
        }
    };
    fill_rows(0);
    __syncthreads();

    const float sig_theta = snsde_sigmoid(a.params[a.off_theta]);
    const int no = a.no;
    const bool tab = a.gt_off >= 0;
    const float* gt = a.gt_ext ? a.gt_ext : a.ws + (tab ? a.gt_off : 0);
    const bool mul_y = (no == 13 || no == 17 || no == 3 || no == 6 || no == 11);
    const bool yfun = (no >= 7 && no <= 10);
    const bool mil = a.method == SNSDE_MILSTEIN;
    const bool phx = a.dW == nullptr;
    const uint32_t grow = (uint32_t)(a.row_offset + row);
    const uint64_t seed = a.seed_dev ? *a.seed_dev : a.seed;
    const int rslot = a.row_out ? a.row_out[rowc] : -1;

    float yv = a.y0[goff];
    ybuf[r * LDY + fo] = yv;
    if (row_ok) {
        a.ys[(size_t)row * H + fo] = yv;
        if constexpr (SAVE) { if (a.traj) a.traj[(size_t)row * H + fo] = yv; }
    }

    // ---- the [X(t) | sin t, cos t] entries of the tile (4 rows x W columns), spread evenly over ALL waves: the waves of a
    //      SIMD stay in step (a wave that works alone while its partner waits at the barrier issues one VALU instruction
    //      per ~5.5 cycles and hides nothing).  Branch-free per step: every lane fetches and evaluates (idle lanes
    //      re-read entry 0), only the owners write.  The cubic pieces of X(t_{n+2}) are fetched at the top of step n,
    //      right after those of X(t_{n+1}) were evaluated out of the same registers ------------------------------------------
    const int xw = xc + (time_on ? 2 : 0);                  // columns per row
    const int xquota = (4 * xw + CF::NW - 1) / CF::NW;      // entries per wave
    float ca[CF::XI], cb[CF::XI], cc[CF::XI], cd[CF::XI];
    uint32_t cvo[CF::XI];                                   // byte offset of (tile row, channel) from the tile's first row
    int xdst[CF::XI], xkind[CF::XI];                        // LDS float offset inside an xbuf half (-1: none); 0 spline, 1 sin, 2 cos
    const size_t cstride = (size_t)(a.L - 1) * 4 * C;       // floats per batch row
#pragma unroll
    for (int i = 0; i < CF::XI; ++i) {
        const int li_ = lane + 64 * i, it = wave * xquota + li_;
        const bool ok = KUXT > 0 && li_ < xquota && it < 4 * xw;
        const int rr = ok ? it / xw : 0, col = ok ? it - rr * xw : 0;
        xdst[i] = ok ? rr * LDX + col : -1;
        xkind[i] = col < xc ? 0 : (col == xc ? 1 : 2);
        const int gr = row0 + rr < B ? rr : B - 1 - row0, ch = col < xc ? col : 0;
        cvo[i] = (uint32_t)((gr * cstride + ch) * sizeof(float));
    }
    const bool has_x = KUXT > 0 && xc > 0;
    const float* ctile = a.coeffs + (size_t)row0 * cstride;
    const uint32_t cstep = (uint32_t)(C * sizeof(float));
    const uint32_t cidx = (uint32_t)(4 * C * sizeof(float));       // bytes per spline interval
    auto load_coeffs = [&](int idx) {      // idx may be a (lane-uniform) VGPR value: the interval offset is per-lane arithmetic,
        if (__builtin_expect(has_x, 1)) {  // no v_readfirstlane / scalar round trip on the way to the addresses
            const uint32_t io = (uint32_t)idx * cidx;
#pragma unroll
            for (int i = 0; i < CF::XI; ++i)
                lean_gload4(ca[i], cb[i], cc[i], cd[i], cvo[i] + io, cvo[i] + io + cstep, cvo[i] + io + 2 * cstep,
                            cvo[i] + io + 3 * cstep, ctile);
        }
    };
    // every asm-issued load has landed: ties the registers they fill
    // (round 4, measured and dropped: s_waitcnt vmcnt(k) that leaves the k hidden-layer activation stores issued after the step's last
    //  prefetch in flight - the training-mode forward did not move (203.2 vs 203.1 us at K2), and with supplied increments the states
    //  came out wrong (test_backward_matches_fp64_autograd_through_the_unrolled_loop[0-mfma4]): a store's acknowledgement can overtake
    //  an older load's return on this part, so a counted wait does not cover the loads while stores are outstanding.)
    auto vm_wait = [&](float& dwn, float& gtn) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(dwn), "+v"(gtn));
#pragma unroll
        for (int i = 0; i < CF::XI; ++i) asm volatile("" : "+v"(ca[i]), "+v"(cb[i]), "+v"(cc[i]), "+v"(cd[i]));
    };
    // X(t) = a + (b + (two_c / 2 + three_d frac / 3) frac) frac  (interpolate.py:270-276); the division by 3 as
    // q = x/3 rounded via two fma corrections (correctly rounded for normal results)
    auto store_xt = [&](float* xb, float frac, float sn, float cs) {
        if constexpr (KUXT > 0) {
#pragma unroll
            for (int i = 0; i < CF::XI; ++i) {
                float v = 0.0f;
                if (__builtin_expect(has_x, 1)) {
                    const float x3 = cd[i] * frac;
                    float q3 = x3 * 0.333333343f;
                    q3 = fmaf(fmaf(-3.0f, q3, x3), 0.333333343f, q3);
                    v = ca[i] + (cb[i] + (0.5f * cc[i] + q3) * frac) * frac;
                }
                v = xkind[i] == 0 ? v : (xkind[i] == 1 ? sn : cs);
                if (xdst[i] >= 0) xb[xdst[i]] = v;
            }
        }
    };

    // Brownian increment of step i for the owned element (Philox: ZB blocks of 4 steps generated together and parked
    // in this wave's LDS stash; else the supplied increments)
    auto next_dw = [&](int i, float sqh) -> float {
        if (__builtin_expect(phx, 1)) {
            const int k = i % (4 * CF::ZB);
            if (__builtin_expect(k == 0, 0)) {
                float zq[4 * CF::ZB];
#pragma unroll
                for (int bb = 0; bb < CF::ZB; ++bb)
                    snsde_philox_normal4(seed, grow, (uint32_t)((i >> 2) + bb), (uint32_t)fo, &zq[4 * bb]);
#pragma unroll
                for (int j = 0; j < 4 * CF::ZB; ++j) zstash[j * 64 + lane] = zq[j];
            }
            return zstash[k * 64 + lane] * sqh;
        }
        float v;
        lean_gload(v, goff4, a.dW + (size_t)i * BH);
        return v;
    };

    // diffusion g(t, y) and  y + g dW (+ the Milstein term)  for the owned element
    auto gpart = [&](float y, float gtv, float dwv, float hh) -> float {
        float g = 0.0f, draw = 0.0f;
        if (__builtin_expect(yfun, 0)) {
            float p1, p2;
            const float raw = snsde_phi(no, y, p1, p2);
            g = fast_tanh(sig_theta * snsde_nan_to_num(raw));
            draw = snsde_finite(raw) ? p1 : 0.0f;
        } else
        {                                 // table noise (no table: gtv = 0 and g = tanh(0) = 0)
            const float raw = mul_y ? gtv * y : gtv;
            if (__builtin_expect(g_raw, 0)) {     // tutorial fields: g = raw, dg/dy = the table entry (or 0)
                float yp = fmaf(raw, dwv, y);
                if (mil && mul_y) yp = fmaf(0.5f * raw * gtv, fmaf(dwv, dwv, -hh), yp);
                return yp;
            }
            g = LEAN_TANH_G(sig_theta * raw);
            draw = (mul_y && snsde_finite(raw)) ? gtv : 0.0f;
        }
        float yp = fmaf(g, dwv, y);
        if (__builtin_expect(mil, 0)) yp = fmaf(0.5f * (g * ((1.0f - g * g) * sig_theta * draw)), fmaf(dwv, dwv, -hh), yp);
        return yp;
    };

    // ---- inputs of step 0; pieces of X(t_1) ---------------------------------------------------------------------------
    float dw_cur, gt_cur = 0.0f;
    f32x4 qa, qb;       // the table row of the step about to run: (h_n, sqrt h_{n+1}, -, -), (sin, cos, frac of step n+1, idx of step n+2)
    {
        const float* g0 = a.step_tab;                       // row 0 of the host table: t0, h, sin, cos, frac, idx, sqrt h
        load_coeffs(__float_as_int(g0[5]));
        float dummy = 0.0f;
        vm_wait(dummy, gt_cur);
        store_xt(xbuf, g0[4], a.raw_time ? g0[0] : g0[2], a.raw_time ? 0.0f : g0[3]);
        dw_cur = next_dw(0, g0[6]);
        if (tab) lean_gload(gt_cur, fo4, gt);
        load_coeffs(__float_as_int(a.step_tab[(size_t)(N > 1 ? 1 : 0) * SNSDE_STEP_STRIDE + 5]));
        vm_wait(dw_cur, gt_cur);
        qa = *reinterpret_cast<const f32x4*>(rowtab);
        qb = *reinterpret_cast<const f32x4*>(rowtab + 4);
    }
    __syncthreads();
#ifndef LEAN_PRIO_MODE
#define LEAN_PRIO_MODE 1      // development A/B (build.py variant): 0 = no priorities, 1 = the younger half first (product), 2 = the older half first
#endif
    if constexpr (LEAN_PRIO_MODE == 1) { if (__builtin_amdgcn_readfirstlane(tid) >= NT / 2 && CF::NW >= 8) __builtin_amdgcn_s_setprio(1); }   // younger half
    if constexpr (LEAN_PRIO_MODE == 2) { if (__builtin_amdgcn_readfirstlane(tid) < NT / 2 && CF::NW >= 8) __builtin_amdgcn_s_setprio(1); }

    // B-operand read addresses (LDS byte offsets; only lanes 0-15 read: q = 0 there)
    const uint32_t yrow = lean_lds_addr(ybuf + r * LDY + 4 * s);
    const uint32_t xrow = lean_lds_addr(xbuf + r * LDX + 4 * s);      // + 4 * LDX floats on odd steps
    const uint32_t arow = lean_lds_addr(bufA + r * LDA + 4 * s);
    const uint32_t brow = lean_lds_addr(bufB + r * LDA + 4 * s);
    float* const aown = bufA + r * LDA + fo;
    float* const bown = bufB + r * LDA + fo;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // (round 4, measured and dropped: the trajectory plane stored TILE-WIDE out of ybuf by one wave of the younger half - H lanes x 16
    //  bytes = the tile's 4 H contiguous bytes, instead of one dword per lane from all eight waves: 216.7 -> 217.8 us for the training-
    //  mode forward at K2.  With the counted-vmcnt experiment at vm_wait this rules out both the store instruction count / segment
    //  width and the acknowledgement drain as the source of the ~4 us per plane and 100 steps that every per-step store costs here.)

    LeanB<(KUXT > 0 ? KUXT : 1)> bx{};    // [X(t_n) | tau_n] operands of the step about to start
    if constexpr (KUXT > 0) lean_read_b_carried(xrow, bx);
    LT_DECL
    // Outer loop over the requested outputs, inner loop over the solver steps up to each of them (out_step[k] = the step
    // after which output k + 1 is due): the step loop itself carries no output bookkeeping.
    int n = 0;
    float yold = yv;
    for (int ko = 0; ko < a.T - 1; ++ko) {
    const int n_end = a.out_step[ko];
    for (; n <= n_end; ++n) {
        const int rbase = (n / CF::ROWCH) * CF::ROWCH;
        if (n > 0 && n == rbase) {               // next chunk of step-table rows (every wave is past the closing barrier)
            fill_rows(rbase);
            __syncthreads();
        }
        const bool more = n + 1 < N;
        // training-mode stores: per-step bases as SCALAR values (readfirstlane makes the step number an SGPR to hipcc, uoff's
        // 32 x 32 -> 64-bit products then run on the scalar unit).  Formed from the loop counter directly, every store paid two
        // quarter-rate v_mul_lo_u32, a v_mad_u64_u32 and four more VALU instructions for its address (~40 VALU instructions per
        // wave-step for five stores, on the issue port the MFMAs share).
        [[maybe_unused]] float* act_n = nullptr;
        [[maybe_unused]] uint32_t nu = 0;
        [[maybe_unused]] uint32_t sgn = 0;
        if constexpr (SAVE) {
            nu = (uint32_t)__builtin_amdgcn_readfirstlane(n);
            act_n = a.act_save + uoff((int)nu, (uint32_t)CF::NSAVE * (uint32_t)BH);
        }
        LT(0)
        // ---- top: the first layer's B operands, then the table quads of the coming steps (asm: the compiler never waits
        //      on them; they have landed once the first layer's last s_waitcnt has passed) ------------------------------------
        LeanB<KUH> by;
        if constexpr (YIN) lean_read_b(yrow, by);
        asm volatile("" : "+v"(qa), "+v"(qb));      // (read before the closing barrier of the previous step: landed)
        const float h = qa[0];
        __builtin_amdgcn_sched_barrier(0);
        // the [X(t_n) | tau_n] part of the first layer: its operands were read before the barrier, so these MFMAs start at
        // once and cover the latency of the y reads
        f32x4 c = bfr[0], d = zero4;
        if constexpr (KUXT > 0) lean_gemm<15, KUXT>(wxt, bx, c, d);
        // ---- in the shadow of those reads: diffusion, y + g dW; X(t_{n+1}) and the fetch of X(t_{n+2})'s pieces -----------
        const float ypart = gpart(yv, gt_cur, dw_cur, h);
        store_xt(xbuf + ((n + 1) & 1) * (4 * LDX), qb[2], qb[0], qb[1]);       // (past the last step: a clamped row, never read)
        load_coeffs(__float_as_int(qb[3]));
        __builtin_amdgcn_sched_barrier(0);
        LT(1)
        if constexpr (YIN) lean_gemm<0, KUH>(wy, by, c, d);
        LT(2)
        {
            const float pre = m4_reduce_scatter(c + d);
            const float o = CF::SWISH ? lean_swish(pre, act_scale) : fmaxf(pre, 0.0f);
            *aown = o;
            if constexpr (SAVE) {
                if (a.act_save && row_ok) {
                    lean_gstore(o, goff4, act_n);
                    if constexpr (CF::SWISH) lean_gstore(pre, goff4, act_n + uoff(0, 0, CF::PRE0, (uint32_t)BH));
                }
                if constexpr (!CF::SWISH) sgn = o > 0.0f ? 1u : 0u;      // relu signs of this lane's element (snsde_pack_signs)
            }
        }
        LT(3)
        __syncthreads();
        LT_COLLECT_A
        LT(4)
        // ---- hidden layers; the next step's increment and diffusion-table entry are produced in the first window -------------
        float dw_nxt = 0.0f, gt_nxt = 0.0f;
        auto prep = [&]() {
            {
                const int n1 = more ? n + 1 : n;          // (the last step prefetches its own inputs again: never used)
                dw_nxt = next_dw(n1, qa[1]);
                if (__builtin_expect(tab, 1)) lean_gload(gt_nxt, fo4, gt + (size_t)n1 * H);
            }
        };
        uint32_t cur = arow;
#pragma unroll
        for (int l = 0; l < NHID; ++l) {
            const bool toB = (l % 2 == 0);
            LeanB<KUH> bh;
            lean_read_b(cur, bh);
            __builtin_amdgcn_sched_barrier(0);
            if (l == 0) prep();
            __builtin_amdgcn_sched_barrier(0);
            if (l == 0) { LT(5) }
            c = bfr[1 + l]; d = zero4;
            lean_gemm<0, KUH>(wh[l], bh, c, d);
            const float pre = m4_reduce_scatter(c + d);
            const float o = CF::SWISH ? lean_swish(pre, act_scale) : fmaxf(pre, 0.0f);
            *(toB ? bown : aown) = o;
            if constexpr (SAVE) {
                if (a.act_save && row_ok) {
                    lean_gstore(o, goff4, act_n + uoff(0, 0, 1 + l, (uint32_t)BH));
                    if constexpr (CF::SWISH) lean_gstore(pre, goff4, act_n + uoff(0, 0, CF::PRE0 + 1 + l, (uint32_t)BH));
                }
                if constexpr (!CF::SWISH) sgn |= (o > 0.0f ? 1u : 0u) << (1 + l);
            }
            if (l == NHID - 1) { LT(6) }
            __syncthreads();
            if (l == NHID - 1) { LT_COLLECT_B LT(7) }
            cur = toB ? brow : arow;
        }
        // ---- output layer, f, update -------------------------------------------------------------------------------
        {
            LeanB<KUH> bo;
            lean_read_b(cur, bo);
            __builtin_amdgcn_sched_barrier(0);
            if (NHID == 0) prep();
            __builtin_amdgcn_sched_barrier(0);
            c = bfr[NHID + 1]; d = zero4;
            lean_gemm<0, KUH>(wo, bo, c, d);
        }
        LT(8)
        vm_wait(dw_nxt, gt_nxt);      // this step's prefetches (issued one to three phases ago)
        float z = m4_reduce_scatter(c + d);
        if constexpr (SAVE) {      // the saved pre-tanh drift carries the step's relu signs in its low NHID + 1 bits (the adjoint's masks)
            if (a.act_save && row_ok) lean_gstore(CF::SWISH ? z : snsde_pack_signs(z, sgn, NHID + 1), goff4, act_n + uoff(0, 0, CF::ZSLOT, (uint32_t)BH));
        }
        if (__builtin_expect(geo, 0)) z *= fast_tanh(yv);
        float f;
        if (__builtin_expect(f_out != SNSDE_DRIFT_TANH, 0)) f = f_out == SNSDE_DRIFT_TIMES_Y ? z * yv : z;
        else f = LEAN_TANH_F(z);
        if constexpr (CF::ACC) {
            // path-integral accumulator column (snsde.h: kl_column1; the field variants' LatentSDE mapping): its drift is the KL rate
            // u = 1/2 sum_j ((f_j - a y_j - b) / g_j)^2 over the latent columns j < acc_col of the tile row: per-wave row sums
            // through LDS and one more barrier per step; the owner lane takes u as its drift value
            if (__builtin_expect(a.acc_col >= 0, 0)) {
                float ev = 0.0f;
                if (fo < a.acc_col) {
                    const float qv = (f - fmaf(a.acc_a, yv, a.acc_b)) * snsde_stable_inv(gt_cur);
                    ev = 0.5f * qv * qv;
                }
                ev = m4_row_sum(ev);
                if (lane < 4) accbuf[wave * 4 + lane] = ev;
                __syncthreads();
                if (fo == a.acc_col) {
                    float u = 0.0f;
                    for (int w = 0; w < CF::NW; ++w) u += accbuf[w * 4 + r];
                    f = u;
                }
            }
        }
        const float ynew = fmaf(f, h, ypart);
        yold = yv;
        yv = ynew;
        ybuf[r * LDY + fo] = ynew;
        if constexpr (SAVE) {
            if (row_ok) {
                if (a.traj) lean_gstore(ynew, goff4, a.traj + uoff((int)nu + 1, (uint32_t)BH));
                if (a.dW_out) lean_gstore(dw_cur, goff4, a.dW_out + uoff((int)nu, (uint32_t)BH));
            }
        }
        dw_cur = dw_nxt; gt_cur = gt_nxt;
        // the next step's table row and [X | tau] operands (written before this step's first barrier), in place; they
        // land before the closing barrier releases (its s_waitcnt lgkmcnt(0))
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16"
                     : "+v"(qa), "+v"(qb) : "v"(lean_lds_addr(rowtab + (n + 1 - rbase) * RS)));
        if constexpr (KUXT > 0) lean_read_b_carried(xrow + ((n + 1) & 1) * (4 * LDX * 4), bx);
        LT(9)
        __syncthreads();
        LT_COLLECT
    }
    if (row_ok) {          // output ko + 1 (linear interpolation inside the last step when the output time is not on the grid)
        const float w0 = a.out_w[2 * ko], w1 = a.out_w[2 * ko + 1];
        const float o = (w0 == 0.0f) ? yv : snsde_interp_out(w0, w1, yold, yv);
        if (!a.row_out) a.ys[(size_t)(ko + 1) * BH + goff] = o;
        else if (rslot == ko + 1) a.ys[goff] = o;
    }
    }
#ifdef LEAN_TRACE
    __syncthreads();       // the timeline goes out through row 0 of the last output plane (trace builds only: that row is garbage then)
    if (blockIdx.x == 0 && lane == 0) {
        float* o = a.ys + (a.row_out ? 0 : (size_t)(a.T - 1) * BH);
        o[wave * 16] = (float)(uint32_t)(ltl[10 * NT] - ltl[11 * NT]);
        for (int i = 1; i < 10; ++i) o[wave * 16 + i] = (float)(uint32_t)(ltl[i * NT] - ltl[(i - 1) * NT]);
    }
#endif
}

template <class CF>
int launch_lean(const MfmaArgs& a, hipStream_t stream) {
    const size_t lds_bytes = (size_t)CF::LDS_FLOATS * sizeof(float) + LT_LDS_EXTRA;
    static SnsdeLdsAttr lds_attr;   // per instantiation and device
    if (const int rc = snsde_lds_attr(reinterpret_cast<const void*>(snsde_m4_kernel<CF>), lds_bytes, lds_attr)) return rc;
    const int grid = (a.B + 3) / 4;
    hipLaunchKernelGGL(snsde_m4_kernel<CF>, dim3(grid), dim3(CF::NT), lds_bytes, stream, a);
    return hipGetLastError() == hipSuccess ? SNSDE_OK : SNSDE_ERR_LAUNCH;
}

// KUXT = 16-wide k-blocks of [X(t) | sin t, cos t]; YIN = 0 only for input_option 0
template <int H>
int dispatch_lean(const MfmaPlan& p, const MfmaArgs& a, hipStream_t st) {
    const bool save = a.act_save || a.traj || a.dW_out;
    if (a.acc_col >= 0) {               // the LatentSDE mapping (fields.compose_latent): relu, one control k-block, y-dependent drift
#ifndef SNSDE_DEV_SUBSET
#define SNSDE_LEAN_ACC(NH_) \
    if constexpr (lean_fits(H, NH_, 1, true) && lean_act_save_fits(H, NH_, 1)) { \
        if (p.NHID == NH_ && p.KUXT == 1 && p.IO != 0 && a.act == SNSDE_ACT_RELU) \
            return save ? launch_lean<CfgL<H, NH_, 1, 1, 1, 0, 1>>(a, st) : launch_lean<CfgL<H, NH_, 1, 1, 0, 0, 1>>(a, st); }
        SNSDE_LEAN_ACC(0) SNSDE_LEAN_ACC(1) SNSDE_LEAN_ACC(2) SNSDE_LEAN_ACC(3)
#undef SNSDE_LEAN_ACC
#endif
        return SNSDE_ERR_UNSUPPORTED;
    }
    if (a.act != SNSDE_ACT_RELU) {      // tutorial fields (LipSwish / SiLU): y-dependent drift on [y | X, t], C + 1 <= 48
    // (training mode also stores the pre-activations: one more live register, so the fullest configurations are inference-only)
#define SNSDE_LEAN_ACT(NH_, KX_) \
    if constexpr (lean_fits(H, NH_, KX_, true)) { \
        if (p.NHID == NH_ && p.KUXT == KX_ && p.IO != 0) { \
            if (!save) return launch_lean<CfgL<H, NH_, KX_, 1, 0, 1>>(a, st); \
            if constexpr (lean_act_save_fits(H, NH_, KX_)) return launch_lean<CfgL<H, NH_, KX_, 1, 1, 1>>(a, st); \
            else return SNSDE_ERR_UNSUPPORTED; } }
#ifndef SNSDE_DEV_SUBSET
        SNSDE_LEAN_ACT(0, 1) SNSDE_LEAN_ACT(1, 1) SNSDE_LEAN_ACT(2, 1) SNSDE_LEAN_ACT(3, 1)
        SNSDE_LEAN_ACT(0, 2) SNSDE_LEAN_ACT(1, 2) SNSDE_LEAN_ACT(2, 2) SNSDE_LEAN_ACT(3, 2)
        SNSDE_LEAN_ACT(0, 3) SNSDE_LEAN_ACT(1, 3) SNSDE_LEAN_ACT(2, 3) SNSDE_LEAN_ACT(3, 3)
#endif
#undef SNSDE_LEAN_ACT
        return SNSDE_ERR_UNSUPPORTED;
    }
#ifdef SNSDE_DEV_SUBSET
    if (p.NHID == 1 && p.KUXT == 2 && p.IO != 0)
        return save ? launch_lean<CfgL<H, 1, 2, 1, 1>>(a, st) : launch_lean<CfgL<H, 1, 2, 1, 0>>(a, st);
    if (p.NHID == 1 && p.KUXT == 1 && p.IO != 0)
        return save ? launch_lean<CfgL<H, 1, 1, 1, 1>>(a, st) : launch_lean<CfgL<H, 1, 1, 1, 0>>(a, st);
    return SNSDE_ERR_UNSUPPORTED;
#else
#define SNSDE_LEAN(NH_, KX_, Y_) \
    if constexpr (lean_fits(H, NH_, KX_, Y_ != 0)) { \
        if (p.NHID == NH_ && p.KUXT == KX_ && (p.IO != 0) == (Y_ != 0)) \
            return save ? launch_lean<CfgL<H, NH_, KX_, Y_, 1>>(a, st) : launch_lean<CfgL<H, NH_, KX_, Y_, 0>>(a, st); }
#define SNSDE_LEANS(KX_, Y_) SNSDE_LEAN(0, KX_, Y_) SNSDE_LEAN(1, KX_, Y_) SNSDE_LEAN(2, KX_, Y_) SNSDE_LEAN(3, KX_, Y_)
    SNSDE_LEANS(0, 1) SNSDE_LEANS(1, 1) SNSDE_LEANS(2, 1) SNSDE_LEANS(3, 1) SNSDE_LEANS(6, 1)
    SNSDE_LEANS(1, 0) SNSDE_LEANS(2, 0) SNSDE_LEANS(3, 0) SNSDE_LEANS(6, 0)
#undef SNSDE_LEANS
#undef SNSDE_LEAN
    return SNSDE_ERR_UNSUPPORTED;
#endif
}

int dispatch_lean_h32(const MfmaPlan& p, const MfmaArgs& a, hipStream_t st);
int dispatch_lean_h64(const MfmaPlan& p, const MfmaArgs& a, hipStream_t st);
int dispatch_lean_h128(const MfmaPlan& p, const MfmaArgs& a, hipStream_t st);
int dispatch_lean_h128_two_tile(const MfmaPlan& p, const MfmaArgs& a, hipStream_t st);   // snsde_m4t_kernel.h
int dispatch_lean_h256(const MfmaPlan& p, const MfmaArgs& a, hipStream_t st, bool stream_all);   // snsde_m4s_kernel.h / snsde_m4s2_kernel.h

}  // namespace snsde_mfma
