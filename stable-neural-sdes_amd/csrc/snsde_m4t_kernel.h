// Lean 4-row-tile forward kernel, TWO tiles per wave (round 6 experiment for the K2 headline): the same step as
// snsde_m4_kernel.h (register-stationary weights, B operands broadcast from LDS, reduce-scatter epilogues, three hand-offs per
// step), but a workgroup is H / 32 waves instead of H / 16 - at H = 128 FOUR waves, one per SIMD, each with the whole 512-register
// file of its SIMD (256 VGPRs + 256 AccVGPRs: the unified file of gfx950) and 32 features:
//   * no second wave competes for the issue port a SIMD's MFMAs and VALU instructions share: the 1664 MFMA cycles per
//     SIMD and step are one wave's, back to back on four independent accumulator chains;
//   * a B-operand read feeds two tiles' MFMAs, the barriers have four participants, the per-wave bookkeeping (table quads, loop
//     control, addresses) is paid four times per step instead of eight.
// What it gives up: the other wave's work in a wave's latency shadows.  Same k order, accumulator chains (c: fragments 0, 2; d: 1, 3,
// bias in the k-slot-0 lanes of the initial accumulator), tanh forms and Philox counters as the lean kernel => bit-identical results;
// SNSDE_FLAG_ONE_TILE keeps the eight-wave kernel for the A/B.  relu fields, y-dependent drifts.
// Reference semantics: benchmark_classification/models_sde/neuralsde.py:295-307 (f, g), SURVEY.md A3-A6 (stepping).
#pragma once
#include "snsde_m4_kernel.h"

namespace snsde_mfma {

template <int H_, int NHID_, int KUXT_, int SAVE_>
struct CfgT {
    static constexpr int H = H_, NHID = NHID_, KUXT = KUXT_;
    static constexpr bool SAVE = SAVE_ != 0;
    static constexpr int NW = H / 32, NT = NW * 64, KUH = H / 16;
    static constexpr int LDY = ld_for(16 * KUH, 16);
    static constexpr int LDX = ld_for(16 * (KUXT > 0 ? KUXT : 1), 16);
    static constexpr int LDA = LDY;
    static constexpr int NLAYER = NHID + 2, NSAVE = NHID + 2, ZSLOT = NHID + 1;
    static constexpr int ROWCH = 128, RS = 8;
    static constexpr int ZB = 4;                              // Philox calls generated together per element
    static constexpr int ZSTASH = 2 * 4 * ZB * 64;            // floats per wave: two tiles
    static constexpr int XI = KUXT > 0 ? (4 * 16 * KUXT + NT - 1) / NT : 1;   // [X(t) | sin t, cos t] entries per lane
    static constexpr int LDS_FLOATS = 4 * (LDY + 2 * LDX + 2 * LDA) + (ROWCH + 3) * RS + NW * ZSTASH;
};

// ---- MFMAs with their A operands in AccVGPRs ----------------------------------------------------------------------------------
// One wave per SIMD owns 512 registers: 256 VGPRs + 256 AccVGPRs.  v_mfma accepts AccVGPR A / B operands on gfx950, but hipcc keeps
// the weight arrays in VGPRs and shuttles the overflow through v_accvgpr_read / write (measured: ~480 copies per step, +28 % kernel
// time).  So the hidden and output layers' fragments are pinned in AccVGPRs by hand ("a" constraints) and their MFMAs issued from
// inline asm; the first layer's stay in VGPRs (same asm, "v" constraints).  Per 16-wide k-block both tiles' eight MFMAs, the four
// accumulator chains interleaved (c0, d0, c1, d1: a chain's next MFMA is three instructions away) in the lean kernel's order per
// chain (c: fragments 0, 2; d: 1, 3).
#define SNSDE_T2_BLOCK(CW)                                                                                                            \
    asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %4, %12, %0 blgp:4\n\tv_mfma_f32_4x4x1_16b_f32 %1, %5, %13, %1 blgp:4\n\t"             \
                 "v_mfma_f32_4x4x1_16b_f32 %2, %8, %12, %2 blgp:4\n\tv_mfma_f32_4x4x1_16b_f32 %3, %9, %13, %3 blgp:4\n\t"             \
                 "v_mfma_f32_4x4x1_16b_f32 %0, %6, %14, %0 blgp:4\n\tv_mfma_f32_4x4x1_16b_f32 %1, %7, %15, %1 blgp:4\n\t"             \
                 "v_mfma_f32_4x4x1_16b_f32 %2, %10, %14, %2 blgp:4\n\tv_mfma_f32_4x4x1_16b_f32 %3, %11, %15, %3 blgp:4"                \
                 : "+v"(c[0]), "+v"(d[0]), "+v"(c[1]), "+v"(d[1])                                                                      \
                 : CW(w0[4 * u]), CW(w0[4 * u + 1]), CW(w0[4 * u + 2]), CW(w0[4 * u + 3]), CW(w1[4 * u]), CW(w1[4 * u + 1]),           \
                   CW(w1[4 * u + 2]), CW(w1[4 * u + 3]), "v"(bb[0]), "v"(bb[1]), "v"(bb[2]), "v"(bb[3]))
#define SNSDE_T2_A(x) "a"(x)
#define SNSDE_T2_V(x) "v"(x)
template <bool AG, int KU>
__device__ __forceinline__ void t2_block(const float (&w0)[KU * 4], const float (&w1)[KU * 4], int u, const f32x4& bb, f32x4 (&c)[2], f32x4 (&d)[2]) {
    if constexpr (AG) { SNSDE_T2_BLOCK(SNSDE_T2_A); } else { SNSDE_T2_BLOCK(SNSDE_T2_V); }
}
// c/d += W . b over KU k-blocks for both tiles; k-blocks consumed in pairs as their reads land (lean_gemm's waits: Y = LDS operations
// issued after the reads of b that may stay in flight)
template <bool AG, int Y, int KU, int U>
__device__ __forceinline__ void t2_gemm_from(const float (&w0)[KU * 4], const float (&w1)[KU * 4], LeanB<KU>& b, f32x4 (&c)[2], f32x4 (&d)[2]) {
    if constexpr (U < KU) {
        if constexpr (U + 1 < KU) lean_wait2<Y + KU - 2 - U>(b.v[U], b.v[U + 1]);
        else lean_wait1<Y>(b.v[U]);
        t2_block<AG, KU>(w0, w1, U, b.v[U], c, d);
        if constexpr (U + 1 < KU) t2_block<AG, KU>(w0, w1, U + 1, b.v[U + 1], c, d);
        __builtin_amdgcn_sched_barrier(0);
        t2_gemm_from<AG, Y, KU, U + 2>(w0, w1, b, c, d);
    }
}
template <bool AG, int Y, int KU>
__device__ __forceinline__ void t2_gemm(const float (&w0)[KU * 4], const float (&w1)[KU * 4], LeanB<KU>& b, f32x4 (&c)[2], f32x4 (&d)[2]) {
    // (VALU-written accumulator init -> first MFMA reads it as SrcC: the hazard recognizer does not see inside asm)
    asm volatile("s_nop 1" : "+v"(c[0]), "+v"(d[0]), "+v"(c[1]), "+v"(d[1]));
    t2_gemm_from<AG, Y, KU, 0>(w0, w1, b, c, d);
}
// the MFMA results are about to be read by VALU instructions: cover the XDL-write -> VALU-read wait states the assembler does not
// insert for asm-issued MFMAs (2-pass 4x4x1: well under 8 cycles)
__device__ __forceinline__ void t2_settle(f32x4 (&c)[2], f32x4 (&d)[2]) {
    asm volatile("s_nop 7" : "+v"(c[0]), "+v"(d[0]), "+v"(c[1]), "+v"(d[1]));
}
// park a weight array in AccVGPRs
template <int K> __device__ __forceinline__ void t2_to_agpr(float (&w)[K]) {
#pragma unroll
    for (int i = 0; i < K; ++i) { float v = w[i], o; asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(o) : "v"(v)); w[i] = o; }
}

template <class CF>
__global__ void __launch_bounds__(CF::NT, 1) snsde_m4t_kernel(MfmaArgs a) {
    constexpr int H = CF::H, NT = CF::NT, NHID = CF::NHID, KUH = CF::KUH, KUXT = CF::KUXT;
    constexpr int LDY = CF::LDY, LDX = CF::LDX, LDA = CF::LDA, RS = CF::RS;
    constexpr bool SAVE = CF::SAVE;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ybuf = lds;                       // [4][LDY]  y
    float* xbuf = ybuf + 4 * LDY;            // [2][4][LDX]  X(t) (xc) | sin t, cos t | 0..   (step parity)
    float* bufA = xbuf + 8 * LDX;            // [4][LDA]
    float* bufB = bufA + 4 * LDA;            // [4][LDA]
    float* rowtab = bufB + 4 * LDA;          // [ROWCH + 3][RS]
    float* zstash_all = rowtab + (CF::ROWCH + 3) * RS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 3, s = (lane >> 2) & 3, q = lane >> 4;
    const int row0 = blockIdx.x * 4;
    const int B = a.B, C = a.C, N = a.N;
    const int row = row0 + r;
    const bool row_ok = row < B;
    const int rowc = row_ok ? row : B - 1;
    const size_t BH = (size_t)B * H;
    int fo[2];
    uint32_t fo4[2], goff4[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        fo[j] = (2 * wave + j) * 16 + 4 * q + s;
        fo4[j] = (uint32_t)(fo[j] * sizeof(float));
        goff4[j] = (uint32_t)(((size_t)rowc * H + fo[j]) * sizeof(float));
    }
    float* const zst = zstash_all + wave * CF::ZSTASH + lane;      // [tile][4 ZB][64]
    const int xc = a.lean_xc;
    const bool time_on = a.lean_time != 0, geo = a.lean_geo != 0;
    const int f_out = a.f_out;
    const bool g_raw = a.g_out == SNSDE_DIFFUSION_RAW;

    // ---- resident weights (both tiles), bias fragments --------------------------------------------------------------------------
    int li = 0;
    float wxt[2][(KUXT > 0 ? KUXT : 1) * 4], wy[2][KUH * 4], wh[NHID > 0 ? NHID : 1][2][KUH * 4], wo[2][KUH * 4];
    if constexpr (KUXT > 0) {
        lean_load_w<KUXT>(wxt[0], a.ws + a.w_off[li], 2 * wave, lane); lean_load_w<KUXT>(wxt[1], a.ws + a.w_off[li], 2 * wave + 1, lane); ++li;
    }
    lean_load_w<KUH>(wy[0], a.ws + a.w_off[li], 2 * wave, lane); lean_load_w<KUH>(wy[1], a.ws + a.w_off[li], 2 * wave + 1, lane); ++li;
#pragma unroll
    for (int l = 0; l < NHID; ++l) {
        lean_load_w<KUH>(wh[l][0], a.ws + a.w_off[li], 2 * wave, lane); lean_load_w<KUH>(wh[l][1], a.ws + a.w_off[li], 2 * wave + 1, lane); ++li;
    }
    lean_load_w<KUH>(wo[0], a.ws + a.w_off[li], 2 * wave, lane); lean_load_w<KUH>(wo[1], a.ws + a.w_off[li], 2 * wave + 1, lane); ++li;
    // hidden and output layers' fragments live in AccVGPRs (t2_block<true>), the first layer's in VGPRs
#pragma unroll
    for (int l = 0; l < NHID; ++l) { t2_to_agpr(wh[l][0]); t2_to_agpr(wh[l][1]); }
    t2_to_agpr(wo[0]); t2_to_agpr(wo[1]);
    f32x4 bfr[CF::NLAYER][2];                              // accumulator init: the bias in the k-slot 0 lanes, 0 elsewhere
#pragma unroll
    for (int l = 0; l < CF::NLAYER; ++l)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) bfr[l][j][i] = (s == 0) ? a.ws[a.bias_off + l * H + (2 * wave + j) * 16 + 4 * q + i] : 0.0f;

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

    float yv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        yv[j] = a.y0[(size_t)rowc * H + fo[j]];
        ybuf[r * LDY + fo[j]] = yv[j];
        if (row_ok) {
            a.ys[(size_t)row * H + fo[j]] = yv[j];
            if constexpr (SAVE) { if (a.traj) a.traj[(size_t)row * H + fo[j]] = yv[j]; }
        }
    }

    // ---- the [X(t) | sin t, cos t] entries of the tile (4 rows x W columns), spread evenly over all waves ----------------------
    const int xw = xc + (time_on ? 2 : 0);
    const int xquota = (4 * xw + CF::NW - 1) / CF::NW;
    float ca[CF::XI], cb[CF::XI], cc[CF::XI], cd[CF::XI];
    uint32_t cvo[CF::XI];
    int xdst[CF::XI], xkind[CF::XI];
    const size_t cstride = (size_t)(a.L - 1) * 4 * C;
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
    const uint32_t cidx = (uint32_t)(4 * C * sizeof(float));
    auto load_coeffs = [&](int idx) {
        if (__builtin_expect(has_x, 1)) {
            const uint32_t io = (uint32_t)idx * cidx;
#pragma unroll
            for (int i = 0; i < CF::XI; ++i)
                lean_gload4(ca[i], cb[i], cc[i], cd[i], cvo[i] + io, cvo[i] + io + cstep, cvo[i] + io + 2 * cstep,
                            cvo[i] + io + 3 * cstep, ctile);
        }
    };
    auto vm_wait = [&](float& d0, float& d1, float& g0, float& g1) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(d0), "+v"(d1), "+v"(g0), "+v"(g1));
#pragma unroll
        for (int i = 0; i < CF::XI; ++i) asm volatile("" : "+v"(ca[i]), "+v"(cb[i]), "+v"(cc[i]), "+v"(cd[i]));
    };
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

    // Brownian increments of step i for the two owned elements (Philox: ZB blocks of 4 steps generated together and parked in this
    // wave's LDS stash; else the supplied increments)
    auto next_dw = [&](int i, float sqh, float& out0, float& out1) {
        if (__builtin_expect(phx, 1)) {
            const int k = i % (4 * CF::ZB);
            if (__builtin_expect(k == 0, 0)) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float zq[4 * CF::ZB];
#pragma unroll
                    for (int bb = 0; bb < CF::ZB; ++bb)
                        snsde_philox_normal4(seed, grow, (uint32_t)((i >> 2) + bb), (uint32_t)fo[j], &zq[4 * bb]);
#pragma unroll
                    for (int m = 0; m < 4 * CF::ZB; ++m) zst[(j * 4 * CF::ZB + m) * 64] = zq[m];
                }
            }
            out0 = zst[k * 64] * sqh;
            out1 = zst[(4 * CF::ZB + k) * 64] * sqh;
            return;
        }
        lean_gload(out0, goff4[0], a.dW + (size_t)i * BH);
        lean_gload(out1, goff4[1], a.dW + (size_t)i * BH);
    };

    auto gpart = [&](float y, float gtv, float dwv, float hh) -> float {
        float g = 0.0f, draw = 0.0f;
        if (__builtin_expect(yfun, 0)) {
            float p1, p2;
            const float raw = snsde_phi(no, y, p1, p2);
            g = fast_tanh(sig_theta * snsde_nan_to_num(raw));
            draw = snsde_finite(raw) ? p1 : 0.0f;
        } else {
            const float raw = mul_y ? gtv * y : gtv;
            if (__builtin_expect(g_raw, 0)) {
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
    float dw_cur[2] = {0.f, 0.f}, gt_cur[2] = {0.f, 0.f};
    f32x4 qa, qb;
#pragma unroll
    for (int i = 0; i < CF::XI; ++i) ca[i] = cb[i] = cc[i] = cd[i] = 0.0f;
    {
        const float* g0 = a.step_tab;
        load_coeffs(__float_as_int(g0[5]));
        float du0 = 0.0f, du1 = 0.0f;
        vm_wait(du0, du1, gt_cur[0], gt_cur[1]);
        store_xt(xbuf, g0[4], a.raw_time ? g0[0] : g0[2], a.raw_time ? 0.0f : g0[3]);
        next_dw(0, g0[6], dw_cur[0], dw_cur[1]);
        if (tab) { lean_gload(gt_cur[0], fo4[0], gt); lean_gload(gt_cur[1], fo4[1], gt); }
        load_coeffs(__float_as_int(a.step_tab[(size_t)(N > 1 ? 1 : 0) * SNSDE_STEP_STRIDE + 5]));
        vm_wait(dw_cur[0], dw_cur[1], gt_cur[0], gt_cur[1]);
        qa = *reinterpret_cast<const f32x4*>(rowtab);
        qb = *reinterpret_cast<const f32x4*>(rowtab + 4);
    }
    __syncthreads();

    const uint32_t yrow = lean_lds_addr(ybuf + r * LDY + 4 * s);
    const uint32_t xrow = lean_lds_addr(xbuf + r * LDX + 4 * s);
    const uint32_t arow = lean_lds_addr(bufA + r * LDA + 4 * s);
    const uint32_t brow = lean_lds_addr(bufB + r * LDA + 4 * s);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    LeanB<(KUXT > 0 ? KUXT : 1)> bx{};
    if constexpr (KUXT > 0) lean_read_b_carried(xrow, bx);
    int n = 0;
    float yold[2] = {yv[0], yv[1]};
    for (int ko = 0; ko < a.T - 1; ++ko) {
    const int n_end = a.out_step[ko];
    for (; n <= n_end; ++n) {
        const int rbase = (n / CF::ROWCH) * CF::ROWCH;
        if (n > 0 && n == rbase) {
            fill_rows(rbase);
            __syncthreads();
        }
        const bool more = n + 1 < N;
        [[maybe_unused]] float* act_n = nullptr;
        [[maybe_unused]] uint32_t nu = 0;
        [[maybe_unused]] uint32_t sgn[2] = {0u, 0u};
        if constexpr (SAVE) {
            nu = (uint32_t)__builtin_amdgcn_readfirstlane(n);
            act_n = a.act_save + uoff((int)nu, (uint32_t)CF::NSAVE * (uint32_t)BH);
        }
        // ---- top: the first layer's B operands; the [X(t_n) | tau_n] part of both tiles covers their latency ----------------------
        LeanB<KUH> by;
        lean_read_b(yrow, by);
        asm volatile("" : "+v"(qa), "+v"(qb));
        const float h = qa[0];
        __builtin_amdgcn_sched_barrier(0);
        f32x4 c[2] = {bfr[0][0], bfr[0][1]}, d[2] = {zero4, zero4};
        if constexpr (KUXT > 0) t2_gemm<false, 15, KUXT>(wxt[0], wxt[1], bx, c, d);
        float ypart[2];
        ypart[0] = gpart(yv[0], gt_cur[0], dw_cur[0], h);
        ypart[1] = gpart(yv[1], gt_cur[1], dw_cur[1], h);
        store_xt(xbuf + ((n + 1) & 1) * (4 * LDX), qb[2], qb[0], qb[1]);
        load_coeffs(__float_as_int(qb[3]));
        __builtin_amdgcn_sched_barrier(0);
        t2_gemm<false, 0, KUH>(wy[0], wy[1], by, c, d);
        t2_settle(c, d);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float pre = m4_reduce_scatter(c[j] + d[j]);
            const float o = fmaxf(pre, 0.0f);
            bufA[r * LDA + fo[j]] = o;
            if constexpr (SAVE) {
                if (a.act_save && row_ok) lean_gstore(o, goff4[j], act_n);
                sgn[j] = o > 0.0f ? 1u : 0u;
            }
        }
        __syncthreads();
        // ---- hidden layers; the next step's increments and diffusion-table entries are produced in the first window ---------------
        float dw_nxt[2] = {0.f, 0.f}, gt_nxt[2] = {0.f, 0.f};
        auto prep = [&]() {
            const int n1 = more ? n + 1 : n;
            next_dw(n1, qa[1], dw_nxt[0], dw_nxt[1]);
            if (__builtin_expect(tab, 1)) { lean_gload(gt_nxt[0], fo4[0], gt + (size_t)n1 * H); lean_gload(gt_nxt[1], fo4[1], gt + (size_t)n1 * H); }
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
            c[0] = bfr[1 + l][0]; c[1] = bfr[1 + l][1]; d[0] = zero4; d[1] = zero4;
            t2_gemm<true, 0, KUH>(wh[l][0], wh[l][1], bh, c, d);
            t2_settle(c, d);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float pre = m4_reduce_scatter(c[j] + d[j]);
                const float o = fmaxf(pre, 0.0f);
                (toB ? bufB : bufA)[r * LDA + fo[j]] = o;
                if constexpr (SAVE) {
                    if (a.act_save && row_ok) lean_gstore(o, goff4[j], act_n + uoff(0, 0, 1 + l, (uint32_t)BH));
                    sgn[j] |= (o > 0.0f ? 1u : 0u) << (1 + l);
                }
            }
            __syncthreads();
            cur = toB ? brow : arow;
        }
        // ---- output layer, f, update ------------------------------------------------------------------------------------------
        {
            LeanB<KUH> bo;
            lean_read_b(cur, bo);
            __builtin_amdgcn_sched_barrier(0);
            if (NHID == 0) prep();
            __builtin_amdgcn_sched_barrier(0);
            c[0] = bfr[NHID + 1][0]; c[1] = bfr[NHID + 1][1]; d[0] = zero4; d[1] = zero4;
            t2_gemm<true, 0, KUH>(wo[0], wo[1], bo, c, d);
        }
        vm_wait(dw_nxt[0], dw_nxt[1], gt_nxt[0], gt_nxt[1]);
        t2_settle(c, d);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float z = m4_reduce_scatter(c[j] + d[j]);
            if constexpr (SAVE) {
                if (a.act_save && row_ok) lean_gstore(snsde_pack_signs(z, sgn[j], NHID + 1), goff4[j], act_n + uoff(0, 0, CF::ZSLOT, (uint32_t)BH));
            }
            if (__builtin_expect(geo, 0)) z *= fast_tanh(yv[j]);
            float f;
            if (__builtin_expect(f_out != SNSDE_DRIFT_TANH, 0)) f = f_out == SNSDE_DRIFT_TIMES_Y ? z * yv[j] : z;
            else f = LEAN_TANH_F(z);
            const float ynew = fmaf(f, h, ypart[j]);
            yold[j] = yv[j];
            yv[j] = ynew;
            ybuf[r * LDY + fo[j]] = ynew;
            if constexpr (SAVE) {
                if (row_ok) {
                    if (a.traj) lean_gstore(ynew, goff4[j], a.traj + uoff((int)nu + 1, (uint32_t)BH));
                    if (a.dW_out) lean_gstore(dw_cur[j], goff4[j], a.dW_out + uoff((int)nu, (uint32_t)BH));
                }
            }
            dw_cur[j] = dw_nxt[j]; gt_cur[j] = gt_nxt[j];
        }
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16"
                     : "+v"(qa), "+v"(qb) : "v"(lean_lds_addr(rowtab + (n + 1 - rbase) * RS)));
        if constexpr (KUXT > 0) lean_read_b_carried(xrow + ((n + 1) & 1) * (4 * LDX * 4), bx);
        __syncthreads();
    }
    if (row_ok) {
        const float w0 = a.out_w[2 * ko], w1 = a.out_w[2 * ko + 1];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float o = (w0 == 0.0f) ? yv[j] : snsde_interp_out(w0, w1, yold[j], yv[j]);
            const size_t go = (size_t)rowc * H + fo[j];
            if (!a.row_out) a.ys[(size_t)(ko + 1) * BH + go] = o;
            else if (rslot == ko + 1) a.ys[go] = o;
        }
    }
    }
}

template <class CF>
int launch_lean2(const MfmaArgs& a, hipStream_t stream) {
    const size_t lds_bytes = (size_t)CF::LDS_FLOATS * sizeof(float);
    static SnsdeLdsAttr lds_attr;   // per instantiation and device
    if (const int rc = snsde_lds_attr(reinterpret_cast<const void*>(snsde_m4t_kernel<CF>), lds_bytes, lds_attr)) return rc;
    const int grid = (a.B + 3) / 4;
    hipLaunchKernelGGL(snsde_m4t_kernel<CF>, dim3(grid), dim3(CF::NT), lds_bytes, stream, a);
    return hipGetLastError() == hipSuccess ? SNSDE_OK : SNSDE_ERR_LAUNCH;
}

}  // namespace snsde_mfma
