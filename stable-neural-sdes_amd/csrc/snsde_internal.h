// Internal declarations shared by the HIP translation units of libsnsde.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "snsde.h"

#define SNSDE_MAX_HIDDEN 8  // NL-1 <= 8 (`linears`, neuralsde.py:156-157; the reference sweeps NL 1..4)

// One nn.Linear inside the flat parameter block (src_*) and inside the packed workspace (w/b).
// Packed weight = W^T with K padded to a multiple of 4: wt[k * N + n], zero rows for k >= K.
// `tshift` = number of leading source columns (the [sin t, cos t] features, neuralsde.py:191-201)
// that are rotated to the END of the packed K axis, so the state y sits at k = 0..H-1, 16-byte
// aligned in LDS, and the time features at k = K-2, K-1.
struct SnsdeLayer {
    int32_t src_w;   // float offset of weight (N, K) in params
    int32_t src_b;   // float offset of bias (N)
    int32_t w;       // float offset of packed W^T (Kpad, N) in workspace
    int32_t K;
    int32_t Kpad;
    int32_t N;
    int32_t tshift;
    int32_t present;
};

struct SnsdeNet {
    SnsdeLayer init;  // initial_network   C -> H          (io 0,2,4,6)
    SnsdeLayer in;    // linear_in         H(+2) -> HH     (io != 0)
    SnsdeLayer emb;   // emb               2H -> H         (io 2,4,6)
    SnsdeLayer hid[SNSDE_MAX_HIDDEN];
    SnsdeLayer out;   // linear_out        HH -> H
    SnsdeLayer ny0;   // noise_y / noise_y.0   H+2 -> H    (no 14,15,18,19)
    SnsdeLayer ny1;   // noise_y.2             H -> H      (no 18,19)
    SnsdeLayer nt0;   // noise_t / noise_t.0   2 -> H      (no 12,13,16,17)  (never packed)
    SnsdeLayer nt1;   // noise_t.2             H -> H      (no 16,17)        (never packed)
    int32_t n_hid;
    int32_t off_theta, off_sigma, off_sigma_diag;  // float offsets in params (-1 if absent)
    int32_t gt_tab;      // float offset in workspace of the time-only diffusion table (N, H), -1 if none
    int32_t packed_floats;  // total packed weight floats
};

struct SnsdeDims {
    int32_t B, H, HH, C, L, NL, io, no, N, T, method;
};

// host-side helpers (snsde_api.cpp)
int snsde_build_net(const snsde_model& m, int32_t n_steps, SnsdeNet* net);

// launchers (snsde_generic.hip)
int snsde_generic_workspace_floats(const snsde_solve* s, const SnsdeNet& net, size_t* floats);
int snsde_generic_launch(const snsde_solve* s, const SnsdeNet& net, hipStream_t stream, int eval_mode,
                         const float* eval_y, float* eval_f, float* eval_g, const float* step_row_dev);
bool snsde_generic_backward_supported(const snsde_solve* s);
int snsde_generic_backward_launch(const snsde_backward* b, const SnsdeNet& net, hipStream_t stream);
int snsde_srk_launch(const snsde_solve* s, const SnsdeNet& net, hipStream_t stream);
int snsde_time_table_srk_launch(const float* params, const float* srk_tab, float* gt, const SnsdeNet& net, int H, int no,
                                int n_rows, hipStream_t stream);
// launchers (snsde_mfma.hip)
bool snsde_mfma_supported(const snsde_solve* s, const SnsdeNet& net);
int snsde_mfma_path(const snsde_solve* s, const SnsdeNet& net, int flavor_hint);
size_t snsde_mfma_workspace_floats(const snsde_solve* s, const SnsdeNet& net);
int snsde_mfma_launch(const snsde_solve* s, const SnsdeNet& net, hipStream_t stream, int flavor_hint);
bool snsde_mfma_backward_supported(const snsde_solve* s, const SnsdeNet& net);
size_t snsde_mfma_backward_workspace_floats(const snsde_solve* s, const SnsdeNet& net);
int snsde_mfma_backward_launch(const snsde_backward* b, const SnsdeNet& net, hipStream_t stream);
const float* snsde_mfma_gt_table(const snsde_solve* s, const SnsdeNet& net);
const float* snsde_mfma_srk_pass_table(const snsde_solve* s, const SnsdeNet& net);   // (3N, SNSDE_STEP_STRIDE) or null
bool snsde_mfma_backward_partials(const snsde_solve* s, const SnsdeNet& net, int* nwg, int* waves, size_t* ds_off,
                                  size_t* dth_off);
// launchers (snsde_w4.hip): wave-owns-rows forward kernels (H = 64, diffusion nets, Euler)
bool snsde_w4_supported(const snsde_solve* s, const SnsdeNet& net);
int snsde_w4_launch(const snsde_solve* s, const SnsdeNet& net, hipStream_t stream);
bool snsde_w4_rev_supported(const snsde_solve* s, const SnsdeNet& net);       // adjoint of the Euler solve on the same wave pairs
// gpart != null: the weight gradients are accumulated inside the adjoint (per-tile blocks in gpart, snsde_w4_grad_floats floats) and
// snsde_w4_grad_reduce_launch forms dL/d params from them; no delta planes are written
int snsde_w4_rev_launch(const snsde_backward* b, const SnsdeNet& net, float* dth_part, float* gpart, hipStream_t stream);
size_t snsde_w4_grad_floats(const snsde_solve* s);
int snsde_w4_grad_reduce_launch(const snsde_backward* b, const SnsdeNet& net, float* grad_params, int32_t n_params, float* gpart,
                                const float* dth_part, hipStream_t stream);
// (snsde_mfma.hip) does the backward of this solve take the wave-pair adjoint with fused weight gradients?  -> offsets (floats) of
// the per-tile blocks and of the theta partial sums inside the backward workspace
bool snsde_mfma_w4_fused(const snsde_backward* b, const SnsdeNet& net, size_t* gpart_off, size_t* dth_off);
bool snsde_mfma_w4_fused_solve(const snsde_solve* s, const SnsdeNet& net, size_t* gpart_off, size_t* dth_off);
// launchers (snsde_wgrad.hip)
size_t snsde_wgrad_workspace_floats(const snsde_backward* b, const SnsdeNet& net);
int snsde_wgrad_launch(const snsde_backward* b, const SnsdeNet& net, float* grad_params, int32_t n_params, float* ws,
                       hipStream_t stream);
int snsde_z0_launch(const snsde_solve* s, hipStream_t stream);   // y0 = z0_weight . X(ts[0]) + z0_bias (stand-alone launch)
int snsde_spline_launch(const float* coeffs, int32_t B, int32_t L, int32_t C, int32_t index, float frac,
                        int32_t derivative, float* out, hipStream_t stream);

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
#ifdef __HIPCC__

// ---- SRK tableau: Roessler's SRI2W1 (SIAM J. Numer. Anal. 48(3), 2010, Table 5.2: strong order 1.5, diagonal noise; the
// coefficients torchsde 0.2.5 ships as _core/methods/tableaus/srid2.py and SRK.diagonal_or_scalar_step walks).  The kernels spell the
// stage formulas out, so only the rows with more than one possible transcription are named here; the whole tableau is in
// oracle/sde_oracle.py (SRK_*), checked term by term against an exact-rational evaluation of the published table
// (tests/golden/make_srk_golden.py).  Rounds 1 - 5 carried B(1) / beta(2) of SRI1W1 (srid1) in these two places: a valid order-1.5
// scheme, but not torchsde's trajectory on the same draws.
//   c(1) = (0, 1/4, 1, 1/4)     A(1) = [1/4; 1 0; 0 0 1/4]     B(1) = [-1/2; 1 0; 2 -1 1/2]
//   H1_1 = y + f0 h/4 + B1_10 g0 sqrt h        H1_2 = y + f0 h + B1_20 g0 sqrt h
//   H1_3 = y + f2 h/4 + (B1_30 g0 + B1_31 g1 + B1_32 g2) sqrt h
#define SRK_B1_10 (-0.5f)
#define SRK_B1_20 (1.0f)
#define SRK_B1_30 (2.0f)
#define SRK_B1_31 (-1.0f)
#define SRK_B1_32 (0.5f)
// diffusion weights of the stages: w_s = beta1_s I_k + beta2_s I_kk / sqrt h + beta3_s I_k0 / h + beta4_s I_kkk / h  (a1 .. a4);
//   beta1 = (-1, 4/3, 2/3, 0)  beta2 = (1, -4/3, 1/3, 0)  beta3 = (2, -4/3, -2/3, 0)  beta4 = (-2, 5/3, -2/3, 1)   (w_3 = a4)
__device__ __forceinline__ float srk_w0(float a1, float a2, float a3, float a4) { return -a1 + a2 + 2.0f * a3 - 2.0f * a4; }
__device__ __forceinline__ float srk_w1(float a1, float a2, float a3, float a4) {
    return (4.0f / 3.0f) * a1 - (4.0f / 3.0f) * a2 - (4.0f / 3.0f) * a3 + (5.0f / 3.0f) * a4;
}
__device__ __forceinline__ float srk_w2(float a1, float a2, float a3, float a4) {
    return (2.0f / 3.0f) * a1 + (1.0f / 3.0f) * a2 - (2.0f / 3.0f) * a3 - (2.0f / 3.0f) * a4;
}

__device__ __forceinline__ float snsde_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// torch.nan_to_num defaults: NaN -> 0, +inf -> FLT_MAX, -inf -> -FLT_MAX (neuralsde.py:306)
__device__ __forceinline__ float snsde_nan_to_num(float x) {
    if (x != x) return 0.0f;
    return fminf(fmaxf(x, -3.402823466e+38f), 3.402823466e+38f);
}

// Element `index` (wave-uniform, known only at run time) of an array member of a kernel's BY-VALUE argument struct, read
// straight from the kernarg segment with scalar loads.  `byte_offset` = offset of the array inside the kernarg segment
// (argument offset + offsetof(struct, member)).  Indexing the by-value struct itself (`a.tile[blockIdx.y]`) makes hipcc copy
// the WHOLE struct to scratch in every lane first: kilobytes of private memory per lane, written by every thread of every
// workgroup — and beyond ~3 KB per lane the launches of this library started to fault in the scratch aperture.
template <class T>
__device__ __forceinline__ T snsde_kernarg_element(size_t byte_offset, int index) {
    static_assert(sizeof(T) % 4 == 0, "dword-sized elements");
    typedef const uint32_t __attribute__((address_space(4)))* KP;
    KP p = (KP)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + byte_offset +
                (size_t)index * sizeof(T));
    T out;
    uint32_t* d = reinterpret_cast<uint32_t*>(&out);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 4); ++i) d[i] = p[i];
    return out;
}

// Finiteness test of a diffusion's raw value (the derivative of nan_to_num is taken as 0 at the clipped values).  NOT
// `x - x == 0`: with x = a * b hipcc's default fp-contract fuses that into fma(a, b, -x), the rounding residual of the
// product, which is non-zero for finite x.
__device__ __forceinline__ bool snsde_finite(float x) { return __builtin_fabsf(x) <= 3.402823466e+38f; }

// Cubic piece evaluation with the reference's exact operation order (controldiffeq/interpolate.py:270-283).
// FMA contraction is switched off for these two functions (HIP's __fmul_rn/__fadd_rn are plain
// operators and would be contracted), so the result is bit-identical to the CPU reference.
__device__ __forceinline__ float snsde_spline_eval(float a, float b, float two_c, float three_d, float frac) {
#pragma clang fp contract(off)
    float inner = 0.5f * two_c + three_d * frac / 3.0f;
    inner = b + inner * frac;
    return a + inner * frac;
}

// Initial state rows of the wrapper (NeuralSDE._prepare_initial_state, neuralsde.py:63-69): y0 = W X(ts[0]) + b with the
// control path evaluated in the interval of the first solver step.  Blocks bx of nbx share the (row, feature) pairs.
struct SnsdeZ0Job {
    const float* w;        // (H, C)
    const float* b;        // (H)
    const float* coeffs;   // (B, L-1, 4C)
    const float* step_tab; // row 0: frac at [4], interval index at [5]
    float* y0;             // (B, H) out
    int32_t B, H, C, L;
};

__device__ __forceinline__ void snsde_z0_rows(const SnsdeZ0Job& z, int bx, int nbx) {
    const float frac = z.step_tab[4];
    const int idx = __float_as_int(z.step_tab[5]);
    const int total = z.B * z.H;
    for (int i = bx * blockDim.x + threadIdx.x; i < total; i += nbx * blockDim.x) {
        const int row = i / z.H, j = i - row * z.H;
        const float* cp = z.coeffs + ((size_t)row * (z.L - 1) + idx) * (4 * z.C);
        const float* wp = z.w + (size_t)j * z.C;
        float acc = z.b[j];
#pragma unroll 4
        for (int c = 0; c < z.C; ++c)
            acc = fmaf(wp[c], snsde_spline_eval(cp[c], cp[z.C + c], cp[2 * z.C + c], cp[3 * z.C + c], frac), acc);
        z.y0[i] = acc;
    }
}
// output between two solver states: torchsde's linear_interp on float32 tensors, (t1 - t) / (t1 - t0) * y0 + (t - t0) / (t1 - t0) * y1 -
// two rounded products and a rounded sum (eager tensor ops do not contract into an fma).  One form in every kernel: left to hipcc the
// contraction differs between instantiations (w0 y0 + fma or fma + w1 y1), a last-bit difference between kernel families on the same states.
__device__ __forceinline__ float snsde_interp_out(float w0, float w1, float y_prev, float y_new) {
#pragma clang fp contract(off)
    const float a = w0 * y_prev, b = w1 * y_new;
    return a + b;
}
__device__ __forceinline__ float snsde_spline_deriv(float b, float two_c, float three_d, float frac) {
#pragma clang fp contract(off)
    float inner = two_c + three_d * frac;
    return b + inner * frac;
}

// Philox4x32-10 (Salmon et al., SC'11).  Specification = oracle/sde_oracle.py:philox4x32_10.
__device__ __forceinline__ void snsde_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                    uint32_t k0, uint32_t k1, uint32_t out[4]) {
#ifndef SNSDE_DEV_PHILOX_ROUNDS
#define SNSDE_DEV_PHILOX_ROUNDS 10      // (development knock-out only - build.py variant: fewer rounds are NOT the specified stream)
#endif
#pragma unroll
    for (int r = 0; r < SNSDE_DEV_PHILOX_ROUNDS; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;   // one v_mad_u64_u32 each
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
#ifdef SNSDE_OLD_PHILOX
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
#else
        c0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96);     // three-input xor: one v_bitop3_b32 (gfx950)
        c1 = lo1;
        c2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
        c3 = lo0;
#endif
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Brownian increments: one Philox4x32-10 call per (global row, block of 4 solver steps, column) yields the four
// standard normals of that state element for steps 4b .. 4b+3:
//   counter = (row, step >> 2, col, 0), key = seed;  u = ((x >> 9) + 0.5) * 2^-23 (exact in fp32);
//   (z0, z1) = sqrt(-2 ln u(x0)) * (cos, sin)(2 pi u(x1)),  (z2, z3) likewise from (x2, x3);  dW_n = z[n & 3] * sqrt(h_n).
// Every state element is owned by exactly one lane in every kernel, so no lane recomputes another's call.
// Specification: oracle/sde_oracle.py philox_normals.
__device__ __forceinline__ void snsde_philox_normal4(uint64_t seed, uint32_t row, uint32_t step_block, uint32_t col,
                                                     float z[4], uint32_t stream = 0u) {
    // stream = 4th counter word: 0 = Brownian increments, 1 = the independent normal of the SRK space-time Levy area
    uint32_t x[4];
    snsde_philox4x32_10(row, step_block, col, stream, (uint32_t)seed, (uint32_t)(seed >> 32), x);
    // u = ((x >> 9) + 0.5) * 2^-23 formed without an int->float conversion: v_alignbit_b32 (0x7f : x) >> 9 is the float
    // 1 + (x >> 9) 2^-23 in [1, 2); subtracting 1 - 2^-24 is exact (the result (2k + 1) 2^-24 has 24 significant bits)
    auto u01 = [](uint32_t v) { return __uint_as_float(__builtin_amdgcn_alignbit(0x7Fu, v, 9)) - 0.99999994f; };
    const float ua = u01(x[0]), ub = u01(x[1]), uc = u01(x[2]), ud = u01(x[3]);
    // Box-Muller on the hardware transcendental units: v_log_f32 (log2), v_sqrt_f32, v_sin_f32 / v_cos_f32 (argument in
    // revolutions, so 2*pi*u needs no range reduction).  -2 ln u = -2 ln2 * log2 u.
    const float ra = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(ua));
    const float rb = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(uc));
    z[0] = ra * __builtin_amdgcn_cosf(ub); z[1] = ra * __builtin_amdgcn_sinf(ub);
    z[2] = rb * __builtin_amdgcn_cosf(ud); z[3] = rb * __builtin_amdgcn_sinf(ud);
}

// Time-only part of the diffusion for noise_option 12,13,16,17 (neuralsde.py:266-277), one table row:
// gt[j] = noise_t([sn, cs])[j] (relu applied for 16/17).  Block-cooperative; hbuf = H floats of LDS.
// Row n of the time-only diffusion table: the part of `raw` (neuralsde.py:233-288) that does not depend on y —
// noise_t(tau_n) for 12/13/16/17, and for the closed forms that are (function of t) x {1, y}:
//   1/2/3: exp(sigma) {1, t, 1}   4/5/6: exp(sigma_diag[j]) {1, t, 1}   11: t      (3, 6, 11 multiply by y in the kernel)
__device__ __forceinline__ void snsde_time_table_row(const float* __restrict__ params, float t, float sn, float cs,
                                                     float* __restrict__ gt, const SnsdeLayer& nt0, const SnsdeLayer& nt1,
                                                     int H, int no, float* hbuf, int off_sigma, int off_sigma_diag) {
    if (no <= 11) {
        for (int j = threadIdx.x; j < H; j += blockDim.x) {
            float v = 1.0f;
            if (no >= 1 && no <= 3) v = expf(params[off_sigma]);
            else if (no >= 4 && no <= 6) v = expf(params[off_sigma_diag + j]);
            if (no == 2 || no == 5 || no == 11) v *= t;
            gt[j] = v;
        }
        return;
    }
    const bool two = (no == 16 || no == 17);
    for (int j = threadIdx.x; j < H; j += blockDim.x) {
        const float v = fmaf(cs, params[nt0.src_w + 2 * j + 1], sn * params[nt0.src_w + 2 * j]) + params[nt0.src_b + j];
        if (two) hbuf[j] = fmaxf(v, 0.0f);
        else gt[j] = v;
    }
    if (!two) return;
    __syncthreads();
    for (int j = threadIdx.x; j < H; j += blockDim.x) {
        float acc = 0.0f;
        const float* w = params + nt1.src_w + (size_t)j * H;
        int k = 0;
        for (; k + 15 < H; k += 16) {      // 16 loads in flight per round trip; the fmaf chain keeps its order
            float wv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) wv[i] = w[k + i];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc = fmaf(hbuf[k + i], wv[i], acc);
        }
        for (; k < H; ++k) acc = fmaf(hbuf[k], w[k], acc);
        gt[j] = fmaxf(acc + params[nt1.src_b + j], 0.0f);
    }
}

// Dynamic LDS above 64 KiB must be enabled per kernel function AND per device (hipFuncSetAttribute acts on the current device's
// copy of the function): remembered per device, so a process that drives several GPUs does not launch the second device's kernel
// without it.  One hipGetDevice per launch (~50 ns); `bytes` may grow between launches of the same function.
struct SnsdeLdsAttr { size_t enabled[16] = {}; };
inline int snsde_lds_attr(const void* fn, size_t bytes, SnsdeLdsAttr& seen) {
    if (bytes <= 64 * 1024) return SNSDE_OK;
    if (bytes > 160 * 1024) return SNSDE_ERR_LDS;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = -1;
    if (dev >= 0 && seen.enabled[dev] >= bytes) return SNSDE_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return SNSDE_ERR_LDS;
    if (dev >= 0) seen.enabled[dev] = bytes;
    return SNSDE_OK;
}

#endif  // __HIPCC__
