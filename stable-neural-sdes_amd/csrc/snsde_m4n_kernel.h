// 4-row-tile MFMA forward kernel for the DIFFUSION NETS (noise_option 14 / 15 / 18 / 19: g = tanh(sigmoid(theta) raw),
// raw = noise_y([sin t, cos t, y]) {x y}, neuralsde.py:270-281) under torchsde's `srk` (SRID2) and `milstein` steppers.
// Included by one translation unit per hidden size (snsde_m4n_h*.hip).
//
// The reference's headline "Neural SDE" (neuralsde_3_18, README.md:32) runs under torch_ists' default method `srk`
// (torch-ists/torch_ists/diff_module/NSDE/nsde_model.py:63-74); dg/dy of these nets is dense, so SRID2's four diffusion
// evaluations per step and Milstein's  1/2 J_g^T (g (dW^2 - h))  are GEMM chains of their own:
//
//   SRK      : per solver step three pseudo-steps (one register-stationary drift pass each at the stage times t0, t0 + h,
//              t0 + h/2, as the elementwise-diffusion SRK variant of snsde_mfma_kernel) and FOUR net evaluations at
//              (t0, y), (t0 + h/4, H1_1), (t0 + h, H1_2), (t0 + h/4, H1_3).  The first three run BESIDE the drift pass of
//              the same pseudo-step (independent chain on its own input rows `gybuf`, same barriers); the fourth needs
//              F2 and is a tail of the third pass.
//   Milstein : per step the drift chain beside  net forward -> [q > 0] W2^T -> [h > 0] W1_y^T  (the VJP of g with
//              cotangent g (dW^2 - h) / 2), four GEMMs of the net's chain against NL + 1 of the drift's.
//
// Skeleton as snsde_mfma_kernel's M4 flavour: persistent 4-row tiles, one 16-feature tile of every layer per wave,
// v_mfma_f32_4x4x1 with the activations as the B operand (read by lanes 0-15, BLGP broadcast), k-slot reduce-scatter by
// DPP, state and SRID2 stage values in registers.  What differs is WHERE THE WEIGHTS LIVE: drift + net (+ the net's
// transposed matrices for Milstein) are up to 232 registers per lane at H = 128 — more than the 256-register budget of
// two waves per SIMD leaves.  The matrices that do not fit are parked in the wave's PRIVATE slice of LDS (each lane
// reads back exactly the 16 bytes it wrote at start-up: no barrier, no bank conflict — LDS as an extension of the
// register file; up to 136 of the 160 KB), read as the A operand with one ds_read_b128 per 16-wide k-block.
#pragma once
#include "snsde_mfma_kernels.h"

namespace snsde_mfma {

template <bool INLDS, int KU> struct WN;
template <int KU> struct WN<false, KU> {
    float v[1][KU * 4];
    __device__ __forceinline__ void load(const float* __restrict__ g, int wave, int lane, float*) { load_weights<KU, 1>(v, g, wave, lane); }
};
template <int KU> struct WN<true, KU> {
    const float* p;   // this lane's 16 bytes of k-block 0 inside the wave's private LDS slice
    __device__ __forceinline__ void load(const float* __restrict__ g, int wave, int lane, float* slice) {
        float* q = slice + lane * 4;
#pragma unroll
        for (int u = 0; u < KU; ++u)
            *reinterpret_cast<f32x4*>(q + u * 256) = *reinterpret_cast<const f32x4*>(g + (((size_t)wave * KU + u) * 64 + lane) * 4);
        p = q;
    }
};

// acc += W_tile . in  (M4 flavour, one tile per wave);  `in` = this lane's LDS row pointer + 4 s
template <int KU>
__device__ __forceinline__ void gemm4(const WN<false, KU>& w, const float* in, f32x4& c, f32x4& d) {
    f32x4 acc[1] = {c}, acc2[1] = {d};
    gemm<1, KU, 1>(w.v, in, acc, acc2);
    c = acc[0]; d = acc2[0];
}
template <int KU>
__device__ __forceinline__ void gemm4(const WN<true, KU>& w, const float* in, f32x4& c, f32x4& d) {
    f32x4 b[KU];
#pragma unroll
    for (int u = 0; u < KU; ++u) asm volatile("" : "=v"(b[u]));     // lanes 16-63: never read (BLGP broadcast of lanes 0-15)
    if ((int)(threadIdx.x & 63) < 16) {
#pragma unroll
        for (int u = 0; u < KU; ++u) b[u] = *reinterpret_cast<const f32x4*>(in + 16 * u);
    }
#pragma unroll
    for (int u = 0; u < KU; ++u) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(w.p + u * 256);
        c = mfma<1>(av[0], b[u][0], c);
        d = mfma<1>(av[1], b[u][1], d);
        c = mfma<1>(av[2], b[u][2], c);
        d = mfma<1>(av[3], b[u][3], d);
    }
}

// Weight placement.  Matrix list of a configuration (the kernel's load order = the plan's pack order):
//     [init piece (KUX)], in piece (KUH + 1), hidden.. (KUH), out (KUH), ny0 (KUH + 1), [ny1 (KUH)], {Milstein: [ny1^T], ny0_y^T}
// Matrices are moved to LDS from the END of the list until the register-resident ones fit the budget (the kernel's working
// set is ~116 registers: B operands of one layer, SRID2 state, Philox block, step-table rows).  -1: does not fit.
__host__ __device__ constexpr int m4n_net_count(int NN, int METHOD) { return METHOD == SNSDE_MILSTEIN ? 2 * NN : NN; }
__host__ __device__ constexpr int m4n_nmat(int KUX, int NHID, int NN, int METHOD) { return (KUX > 0 ? 1 : 0) + 2 + NHID + m4n_net_count(NN, METHOD); }
__host__ __device__ constexpr int m4n_mat_ku(int H, int KUX, int NHID, int i) {
    int j = i;
    if (KUX > 0) { if (j == 0) return KUX; --j; }
    if (j == 0) return H / 16 + 1;          // in piece: [y | sin t, cos t]
    --j;
    if (j < NHID + 1) return H / 16;        // hidden.., out
    j -= NHID + 1;
    return j == 0 ? H / 16 + 1 : H / 16;    // ny0 on [y | sin t, cos t]; ny1 and the transposed matrices
}
// Weight registers per lane.  H = 128 under SRK: the SRID2 state, the two reciprocals and the sign word of the step push the working
// set past what 136 weight registers leave (13 - 31 spilled registers per instantiation, scratch traffic inside the pass loop), so the
// placement first tries a tighter budget (SNSDE_M4N_SRK_BUDGET: one more matrix parked in LDS) and keeps 136 where that does not fit the
// LDS cap - no configuration loses its instantiation.
#ifndef SNSDE_M4N_SRK_BUDGET
#define SNSDE_M4N_SRK_BUDGET 120
#endif
__host__ __device__ constexpr int m4n_reg_budget(int H) { return H >= 128 ? 136 : 144; }
__host__ __device__ constexpr int m4n_lds_cap_blocks(int H) { return (H >= 128 ? 136 : 56) / (H / 16); }   // 1 KB blocks per wave
__host__ __device__ constexpr int m4n_nlds_for(int H, int KUX, int NHID, int NN, int METHOD, int budget) {
    const int nm = m4n_nmat(KUX, NHID, NN, METHOD);
    int regs = 0;
    for (int i = 0; i < nm; ++i) regs += 4 * m4n_mat_ku(H, KUX, NHID, i);
    int nl = 0, blocks = 0;
    while (regs > budget && nl < nm) {
        const int ku = m4n_mat_ku(H, KUX, NHID, nm - 1 - nl);
        if (blocks + ku > m4n_lds_cap_blocks(H)) break;
        regs -= 4 * ku; blocks += ku; ++nl;
    }
    return regs > budget ? -1 : nl;
}
__host__ __device__ constexpr int m4n_nlds(int H, int KUX, int NHID, int NN, int METHOD) {
    if (H >= 128 && METHOD == SNSDE_SRK) {
        const int tight = m4n_nlds_for(H, KUX, NHID, NN, METHOD, SNSDE_M4N_SRK_BUDGET);
        if (tight >= 0) return tight;
    }
    return m4n_nlds_for(H, KUX, NHID, NN, METHOD, m4n_reg_budget(H));
}

template <int H_, int KUX_, int NHID_, int NN_, int METHOD_>
struct CfgN {
    static constexpr int H = H_, KUX = KUX_, NHID = NHID_, NN = NN_, METHOD = METHOD_;
    static constexpr bool SRK = METHOD == SNSDE_SRK, MIL = METHOD == SNSDE_MILSTEIN;
    static constexpr bool EMB = KUX > 0;            // folded first layer on [y, tau | X(t)] (input_option 2 / 4 / 6)
    static constexpr int NW = H / 16, NT = NW * 64, WPS = NW >= 8 ? NW / 4 : 2, M = 4;
    static constexpr int KUH = H / 16, KUY = KUH + 1, KUN = KUH + 1;
    static constexpr int PAD = 16;
    static constexpr int LDY = ld_for(16 * KUY, PAD), LDX = EMB ? ld_for(16 * KUX, PAD) : 0, LDA = ld_for(16 * KUH, PAD);
    static constexpr int NLAYER = NHID + 2 + NN;                      // bias rows: first, hidden.., out, net..
    static constexpr int XI = EMB ? (M * 16 * KUX + NT - 1) / NT : 1;
    static constexpr int ZSLOT = NHID + 1;                            // act_save slot of the pre-tanh drift
    static constexpr int NSAVE = NHID + 2 + (SRK ? 2 * NN : NN);      // + net hidden / output (SRK: and the tail evaluation's)
    static constexpr int NPLANE = SRK ? 3 : 1;                        // stage_save planes per pass: H0 | H1 | H1_3
    static constexpr int ROWCH = 48;                                  // step-table rows staged in LDS (multiple of 3)
    static constexpr int NNET = m4n_net_count(NN, METHOD);
    static constexpr int NMAT = m4n_nmat(KUX, NHID, NN, METHOD);
    static constexpr int NLDS = m4n_nlds(H, KUX, NHID, NN, METHOD);
    static constexpr bool FITS = NLDS >= 0;
    // indices into the matrix list
    static constexpr int IX_Y = EMB ? 1 : 0, IX_H0 = IX_Y + 1, IX_O = IX_H0 + NHID, IX_N0 = IX_O + 1, IX_N1 = IX_N0 + 1,
                         IX_N1T = IX_N0 + NN, IX_N0T = IX_N0 + NNET - 1;
    static constexpr bool in_lds(int i) { return i >= 0 && i < NMAT && i >= NMAT - NLDS; }
    static constexpr int ku_of(int i) { return m4n_mat_ku(H, KUX, NHID, i); }
    static constexpr int lds_w_off(int i) {                           // float offset of matrix i's region (all waves)
        int o = 0;
        for (int j = 0; j < i && j < NMAT; ++j) if (in_lds(j)) o += ku_of(j) * 256 * NW;
        return o;
    }
    static constexpr int NBUF = 3 + (MIL ? 2 : 0);
    static constexpr int LDS_ACT = M * ((SRK ? 2 : 1) * LDY + LDX + NBUF * LDA) + (ROWCH + 1) * SNSDE_STEP_STRIDE;
    static constexpr int LDS_FLOATS = LDS_ACT + lds_w_off(NMAT);
};

template <class CF>
__global__ void __launch_bounds__(CF::NT, CF::WPS) snsde_m4n_kernel(MfmaArgs a) {
    constexpr int H = CF::H, M = 4, NT = CF::NT, NHID = CF::NHID, NN = CF::NN;
    constexpr int KUX = CF::KUX, KUY = CF::KUY, KUH = CF::KUH, KUN = CF::KUN;
    constexpr int LDY = CF::LDY, LDX = CF::LDX, LDA = CF::LDA, NSAVE = CF::NSAVE, NP = CF::NPLANE;
    constexpr bool SRK = CF::SRK, MIL = CF::MIL;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ybuf = lds;                                   // [4][LDY]  drift input state | sin t, cos t | 0..
    float* gybuf = SRK ? ybuf + M * LDY : ybuf;          // [4][LDY]  net input state | ITS sin t, cos t  (SRK: own rows)
    float* xbuf = ybuf + (SRK ? 2 : 1) * M * LDY;        // [4][LDX]  X(t) | 0..
    float* bufA = xbuf + M * LDX;
    float* bufB = bufA + M * LDA;
    float* nbuf = bufB + M * LDA;                        // hidden layer of the diffusion net (18 / 19)
    float* tb0 = nbuf + M * LDA;                         // Milstein: inputs of the net's transposed chain
    float* tb1 = tb0 + (MIL ? M * LDA : 0);
    float* rowtab = nbuf + (CF::NBUF - 2) * M * LDA;     // [ROWCH + 1][SNSDE_STEP_STRIDE]
    float* wlds = rowtab + (CF::ROWCH + 1) * SNSDE_STEP_STRIDE;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 3, s = (lane >> 2) & 3, fsub = 4 * (lane >> 4);
    const int row0 = blockIdx.x * M, B = a.B, C = a.C;
    const int row = row0 + r, rowc = row < B ? row : B - 1;
    const bool row_ok = row < B;
    const size_t BH = (size_t)B * H;
    const int fcol = wave * 16 + fsub + s;               // the state element this lane owns after a reduce-scatter
    const size_t goff = (size_t)rowc * H + fcol;

    // ---- weights (registers or this wave's private LDS slice) ----
    WN<CF::EMB && CF::in_lds(0), CF::EMB ? KUX : 1> wx;
    WN<CF::in_lds(CF::IX_Y), KUY> wy;
    WN<(NHID > 0) && CF::in_lds(CF::IX_H0), KUH> wh0;
    WN<(NHID > 1) && CF::in_lds(CF::IX_H0 + 1), KUH> wh1;
    WN<(NHID > 2) && CF::in_lds(CF::IX_H0 + 2), KUH> wh2;
    WN<CF::in_lds(CF::IX_O), KUH> wo;
    WN<CF::in_lds(CF::IX_N0), KUN> wn0;
    WN<(NN > 1) && CF::in_lds(CF::IX_N1), (NN > 1) ? KUH : 1> wn1;
    WN<MIL && (NN > 1) && CF::in_lds(CF::IX_N1T), (MIL && NN > 1) ? KUH : 1> wn1t;     // Milstein: ny1^T
    WN<MIL && CF::in_lds(CF::IX_N0T), MIL ? KUH : 1> wn0t;                              // Milstein: ny0[:, y columns]^T
    {
        int li = 0;
        auto slice = [&](int ix, int ku) { return wlds + CF::lds_w_off(ix) + wave * ku * 256; };
        if constexpr (CF::EMB) { wx.load(a.ws + a.w_off[li], wave, lane, slice(0, KUX)); ++li; }
        wy.load(a.ws + a.w_off[li++], wave, lane, slice(CF::IX_Y, KUY));
        if constexpr (NHID > 0) wh0.load(a.ws + a.w_off[li++], wave, lane, slice(CF::IX_H0, KUH));
        if constexpr (NHID > 1) wh1.load(a.ws + a.w_off[li++], wave, lane, slice(CF::IX_H0 + 1, KUH));
        if constexpr (NHID > 2) wh2.load(a.ws + a.w_off[li++], wave, lane, slice(CF::IX_H0 + 2, KUH));
        wo.load(a.ws + a.w_off[li++], wave, lane, slice(CF::IX_O, KUH));
        wn0.load(a.ws + a.w_off[li++], wave, lane, slice(CF::IX_N0, KUN));
        if constexpr (NN > 1) wn1.load(a.ws + a.w_off[li++], wave, lane, slice(CF::IX_N1, KUH));
        if constexpr (MIL && NN > 1) wn1t.load(a.ws + a.w_off[li++], wave, lane, slice(CF::IX_N1T, KUH));
        if constexpr (MIL) wn0t.load(a.ws + a.w_off[li++], wave, lane, slice(CF::IX_N0T, KUH));
    }

    for (int i = tid; i < CF::LDS_ACT - (CF::ROWCH + 1) * SNSDE_STEP_STRIDE; i += NT) lds[i] = 0.0f;
    __syncthreads();

    const float sig_theta = snsde_sigmoid(a.params[a.off_theta]);
    const int no = a.no;
    const bool mul_y = (no == 15 || no == 19);
    // field variants (tutorial-style NeuralSDEFunc, fields.py): LipSwish / SiLU instead of relu, f = z, g = the net's LINEAR output
    // (no rectifier on its last layer, no tanh(sigmoid(theta) .)), raw time feature [t, 0]
    const int act_fn = a.act;
    const bool f_lin = a.f_out != 0, g_raw = a.g_out != 0, net_lin = a.g_out == SNSDE_DIFFUSION_RAW_NET, raw_time = a.raw_time != 0;
    const float act_scale = act_fn == SNSDE_ACT_LIPSWISH ? 0.909f : 1.0f;
    auto actf = [&](float x) {
        if (__builtin_expect(act_fn != 0, 0)) return act_scale * x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
        return fmaxf(x, 0.0f);
    };
    // d act(x) / dx from the PRE-activation (relu: from the sign)
    auto dactf = [&](float x) { return __builtin_expect(act_fn != 0, 0) ? swish_grad(x, act_scale) : (x > 0.0f ? 1.0f : 0.0f); };
    const bool geo = a.lean_geo != 0;
    const uint32_t grow = (uint32_t)(a.row_offset + row);
    const uint64_t seed = a.seed_dev ? *a.seed_dev : a.seed;
    const int rslot = a.row_out ? a.row_out[rowc] : -1;
    const bool phx = a.dW == nullptr;

    float bias_own[CF::NLAYER];
#pragma unroll
    for (int l = 0; l < CF::NLAYER; ++l) bias_own[l] = a.ws[a.bias_off + l * H + fcol];
    constexpr int NROW = NHID + 2;      // bias row of the net's first layer

    float yv = a.y0[goff];
    ybuf[r * LDY + fcol] = yv;
    if constexpr (SRK) gybuf[r * LDY + fcol] = yv;
    if (row_ok) {
        a.ys[(size_t)row * H + fcol] = yv;
        if (a.traj) a.traj[(size_t)row * H + fcol] = yv;
        if constexpr (SRK) {
            if (a.stage_save) { a.stage_save[(size_t)row * H + fcol] = yv; a.stage_save[BH + (size_t)row * H + fcol] = yv; }
        }
    }

    // spline items of this thread
    int xr[CF::XI], xc[CF::XI];
    bool xok[CF::XI];
    float ca[CF::XI], cb[CF::XI], cc[CF::XI], cd[CF::XI];
#pragma unroll
    for (int i = 0; i < CF::XI; ++i) {
        const int it = tid + i * NT;
        xr[i] = it / C; xc[i] = it - xr[i] * C;
        xok[i] = CF::EMB && it < M * C;
        if (!xok[i]) { xr[i] = 0; xc[i] = 0; }
    }
    auto load_coeffs = [&](int idx) {
#pragma unroll
        for (int i = 0; i < CF::XI; ++i)
            if (xok[i]) {
                const int rr = row0 + xr[i] < B ? row0 + xr[i] : B - 1;
                const float* cp = a.coeffs + ((size_t)rr * (a.L - 1) + idx) * (4 * C) + xc[i];
                ca[i] = cp[0]; cb[i] = cp[C]; cc[i] = cp[2 * C]; cd[i] = cp[3 * C];
            }
    };
    auto store_x = [&](float frac) {
#pragma unroll
        for (int i = 0; i < CF::XI; ++i)
            if (xok[i]) xbuf[xr[i] * LDX + xc[i]] = snsde_spline_eval(ca[i], cb[i], cc[i], cd[i], frac);
    };
    {   // pass 0 inputs
        const float* st = a.step_tab;
        if constexpr (CF::EMB) { load_coeffs(__float_as_int(st[5])); store_x(st[4]); }
        if (tid < M) {
            // (Euler / Milstein rows carry sin / cos of the host grid: the raw feature is [t0, 0]; the SRK pass table is built
            //  with the right feature already)
            ybuf[tid * LDY + H] = (raw_time && !SRK) ? st[0] : st[2]; ybuf[tid * LDY + H + 1] = (raw_time && !SRK) ? 0.0f : st[3];
            if constexpr (SRK) { gybuf[tid * LDY + H] = st[10]; gybuf[tid * LDY + H + 1] = st[11]; }
        }
    }

    const float* yrow = ybuf + r * LDY + 4 * s;
    const float* gyrow = gybuf + r * LDY + 4 * s;
    const float* xrow = xbuf + r * LDX + 4 * s;
    const float* arow = bufA + r * LDA + 4 * s;
    const float* brow = bufB + r * LDA + 4 * s;
    const float* nrow = nbuf + r * LDA + 4 * s;

    struct Row { float h, sn, cs, frac, sqh, nsn, ncs; int idx, nout, kfirst; };      // (sn, cs: the pass's time features)
    const int n_loop = SRK ? 3 * a.N : a.N;
    auto fill_rows = [&](int base) {
        for (int i = tid; i < (CF::ROWCH + 1) * SNSDE_STEP_STRIDE; i += NT) {
            const int rr = base + i / SNSDE_STEP_STRIDE;
            rowtab[i] = a.step_tab[(size_t)(rr < n_loop - 1 ? rr : n_loop - 1) * SNSDE_STEP_STRIDE + i % SNSDE_STEP_STRIDE];
        }
    };
    auto get_row = [&](int i, int base) {
        const float* st = rowtab + (i - base) * SNSDE_STEP_STRIDE;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(st), v1 = *reinterpret_cast<const f32x4*>(st + 4),
                    v2 = *reinterpret_cast<const f32x4*>(st + 8);
        Row q;
        q.h = v0[1]; q.sn = v0[2]; q.cs = v0[3]; q.frac = v1[0]; q.sqh = v1[2];
        if (!SRK && raw_time) { q.sn = v0[0]; q.cs = 0.0f; }
        q.idx = __float_as_int(v1[1]); q.nout = __float_as_int(v2[0]); q.kfirst = __float_as_int(v2[1]);
        q.nsn = v2[2]; q.ncs = v2[3];
        return q;
    };
    fill_rows(0);
    __syncthreads();

    // g = tanh(sigmoid(theta) nan_to_num(raw)), raw = q or q * (the state the net was evaluated at)
    auto gfun = [&](float q, float yy) {
        const float raw = mul_y ? q * yy : q;
        if (__builtin_expect(g_raw, 0)) return raw;
        return fast_tanh(sig_theta * snsde_nan_to_num(raw));
    };
    // smooth activations (field variants): the NL drift pre-activations and the net's hidden pre-activation follow the regular slots
    // (snsde_act_slots); SRK: per pass, + the hidden pre-activation of the step's fourth evaluation (pass 3n + 2) in one more slot
    const int nsave_rt = act_fn != 0 ? NSAVE + NHID + 1 + (NN == 2 ? (SRK ? 2 : 1) : 0) : NSAVE;
    auto save_act = [&](int pass, int slot, float v) {
        if (a.act_save && row_ok) (a.act_save + uoff(pass, (uint32_t)nsave_rt * (uint32_t)BH, slot, (uint32_t)BH))[(uint32_t)(row * H + fcol)] = v;
    };
    auto save_pre = [&](int pass, int idx, float v) {      // idx: drift layer 0 .. NHID, NHID + 1 = the net's hidden layer (SRK: + 2 = the tail's)
        if (act_fn != 0) save_act(pass, NSAVE + idx, v);
    };
    // one net evaluation's layer 1 (reads gyrow): NN == 2 -> relu'd hidden into nbuf (returned); NN == 1 -> the output q
    auto net_l1 = [&](int pass, int slot0, float& q) {
        f32x4 c = {0.f, 0.f, 0.f, 0.f}, d = c;
        gemm4<KUN>(wn0, gyrow, c, d);
        float o = m4_reduce_scatter(c + d) + bias_own[NROW];
        const float pre = o;
        if constexpr (NN == 2) {
            o = actf(o);
            nbuf[r * LDA + fcol] = o;
            save_pre(pass, slot0 == CF::ZSLOT + 1 ? NHID + 1 : NHID + 2, pre);
        } else {
            q = o;
        }
        save_act(pass, slot0, o);
        return pre;          // (NN == 2: the hidden PRE-activation, for the derivative of the activation)
    };
    auto net_l2 = [&](int pass, int slot1, float& q) {      // NN == 2: q = relu(W2 hidden + b2)
        f32x4 c = {0.f, 0.f, 0.f, 0.f}, d = c;
        gemm4<(NN > 1) ? KUH : 1>(wn1, nrow, c, d);
        q = m4_reduce_scatter(c + d) + bias_own[NN > 1 ? NROW + 1 : NROW];
        if (!net_lin) q = fmaxf(q, 0.0f);        // (SNSDE_DIFFUSION_RAW_NET: the net ends in its linear layer)
        save_act(pass, slot1, q);
    };

    // SRID2 state of the step (own element)
    float sk_y = yv, sk_f0 = 0.f, sk_f1 = 0.f, sk_g0 = 0.f, sk_g1 = 0.f, sk_g2 = 0.f, sk_dw = 0.f, sk_du = 0.f, sk_h1 = yv;
    float sk_z[4] = {0.f, 0.f, 0.f, 0.f}, sk_x[4] = {0.f, 0.f, 0.f, 0.f};
    float tail_sn = 0.f, tail_cs = 0.f;      // time features of t0 + h/4 (the tail evaluation's)
    // 1 / h and 1 / sqrt h of the step in hand (two IEEE divisions per step at stage 0; the stage formulas below multiply by them:
    // written with `/ h` and `/ sqh` they cost seven divisions of ~11 VALU instructions each per step on the issue port the MFMAs share)
    float sk_rh = 0.f, sk_rsqh = 0.f;

    Row row_next = get_row(0, 0);
    for (int n = 0; n < n_loop; ++n) {
        const bool more = n + 1 < n_loop;
        const int stage = SRK ? n % 3 : 0;
        const int ns = SRK ? n / 3 : n;
        const int rbase = (n / CF::ROWCH) * CF::ROWCH;
        if (n > 0 && n == rbase) {
            fill_rows(rbase);
            __syncthreads();
        }
        // (the row of pass n was read as `nxt` by the previous pass: one LDS fetch + scalarisation per pass instead of two - its
        //  ds_read -> s_waitcnt -> v_readfirstlane chain sits at the top of every pass, in front of the first layers' operand reads)
        const Row cur_row = row_next, nxt = get_row(more ? n + 1 : n, rbase);
        row_next = nxt;
        if constexpr (CF::EMB) { if (more) load_coeffs(nxt.idx); }
        const float h = cur_row.h, sqh = cur_row.sqh;
        if constexpr (SRK) { if (stage == 1) { tail_sn = cur_row.nsn; tail_cs = cur_row.ncs; } }

        // ---- phase 0: first layers of the drift (on yrow [+ xrow]) and of the net (on gyrow) ----
        float q = 0.0f, z = 0.0f, dw = 0.0f;
        uint32_t sgn = 0;
        {
            f32x4 c = {0.f, 0.f, 0.f, 0.f}, d = c;
            gemm4<KUY>(wy, yrow, c, d);
            if constexpr (CF::EMB) gemm4<CF::EMB ? KUX : 1>(wx, xrow, c, d);
            const float pre = m4_reduce_scatter(c + d) + bias_own[0];
            const float o = actf(pre);
            bufA[r * LDA + fcol] = o;
            save_act(n, 0, o);
            save_pre(n, 0, pre);
            sgn = o > 0.0f ? 1u : 0u;      // relu signs of this lane's element (Euler: folded into the saved z, snsde_pack_signs)
        }
        const float nhid_own = net_l1(n, CF::ZSLOT + 1, q);      // (NN == 2: this lane's hidden pre-activation of the net)
        if constexpr ((SRK || MIL) && NN == 2) sgn |= (nhid_own > 0.0f ? 1u : 0u) << (NHID + 1);      // SRK / Milstein: the net's hidden sign rides in z as well
        __syncthreads();
        // ---- phase 1: net output; increments; next pass's control values / time features ----
        if constexpr (NN == 2) net_l2(n, CF::ZSLOT + 2, q);
        if constexpr (SRK) {
            if (stage == 0) {
                if (phx) {
                    if ((ns & 3) == 0) {
                        snsde_philox_normal4(seed, grow, (uint32_t)(ns >> 2), (uint32_t)fcol, sk_z, 0u);
                        snsde_philox_normal4(seed, grow, (uint32_t)(ns >> 2), (uint32_t)fcol, sk_x, 1u);
                    }
                    const int k = ns & 3;
                    const float zz = k == 0 ? sk_z[0] : (k == 1 ? sk_z[1] : (k == 2 ? sk_z[2] : sk_z[3]));
                    const float xi = k == 0 ? sk_x[0] : (k == 1 ? sk_x[1] : (k == 2 ? sk_x[2] : sk_x[3]));
                    sk_dw = zz * sqh;
                    sk_du = h * fmaf(sqrtf(h / 12.0f), xi, 0.5f * sk_dw);
                } else {
                    sk_dw = a.dW[(size_t)ns * BH + goff];
                    sk_du = a.dU[(size_t)ns * BH + goff];
                }
            }
            dw = sk_dw;
        } else {
            if (phx) {
                if ((n & 3) == 0) snsde_philox_normal4(seed, grow, (uint32_t)(n >> 2), (uint32_t)fcol, sk_z, 0u);
                const int k = n & 3;
                dw = (k == 0 ? sk_z[0] : (k == 1 ? sk_z[1] : (k == 2 ? sk_z[2] : sk_z[3]))) * sqh;
            } else {
                dw = a.dW[(size_t)n * BH + goff];
            }
        }
        if (more || SRK) {
            // the first layers have read this pass's xbuf / time columns: write the next pass's.  SRK stage 2: the net rows
            // get the TAIL's time (t0 + h/4) here and the next step's after the tail's first layer.
            if constexpr (CF::EMB) { if (more) store_x(nxt.frac); }
            if (tid < M) {
                if (more) { ybuf[tid * LDY + H] = nxt.sn; ybuf[tid * LDY + H + 1] = nxt.cs; }
                if constexpr (SRK) {
                    const bool tl = stage == 2;
                    gybuf[tid * LDY + H] = tl ? tail_sn : nxt.nsn;
                    gybuf[tid * LDY + H + 1] = tl ? tail_cs : nxt.ncs;
                }
            }
        }
        // Milstein: cotangent of the net's transposed chain,  u = (g (dW^2 - h) / 2) (1 - g^2) sigmoid(theta) {y} [q > 0]
        float mg = 0.0f, mdirect = 0.0f, mv = 0.0f;
        if constexpr (MIL) {
            const float raw = mul_y ? q * yv : q;
            mg = g_raw ? raw : fast_tanh(sig_theta * snsde_nan_to_num(raw));
            const float dgr = g_raw ? 1.0f : (1.0f - mg * mg) * sig_theta;          // dg / d raw
            const float cr = snsde_finite(raw) ? 0.5f * mg * fmaf(dw, dw, -h) * dgr : 0.0f;
            mdirect = mul_y ? cr * q : 0.0f;
            float u = mul_y ? cr * yv : cr;
            if constexpr (NN == 2) { if (!net_lin) u = q > 0.0f ? u : 0.0f; }
            tb0[r * LDA + fcol] = u;
        }
        // ---- drift hidden layers / output layer, the net's transposed chain beside them (Milstein) ----
        const float* cur = arow;
        constexpr int NTR = MIL ? NN : 0;          // transposed GEMMs of the net still to run
        int tdone = 0;
        bool tsync = true;                         // the next transposed GEMM's input buffer is behind a barrier
        auto net_t = [&]() {
            f32x4 c = {0.f, 0.f, 0.f, 0.f}, d = c;
            if (NN == 2 && tdone == 0) {
                gemm4<(MIL && NN > 1) ? KUH : 1>(wn1t, tb0 + r * LDA + 4 * s, c, d);
                const float o = m4_reduce_scatter(c + d);
                tb1[r * LDA + fcol] = o * dactf(nhid_own);
            } else {
                gemm4<MIL ? KUH : 1>(wn0t, (NN == 2 ? tb1 : tb0) + r * LDA + 4 * s, c, d);
                mv = m4_reduce_scatter(c + d) + mdirect;
            }
            ++tdone;
            tsync = false;
        };
        if constexpr (MIL) __syncthreads();        // tb0 complete
#pragma unroll
        for (int l = 0; l < NHID; ++l) {
            const bool toB = (l % 2 == 0);
            f32x4 c = {0.f, 0.f, 0.f, 0.f}, d = c;
            if (l == 0) gemm4<KUH>(wh0, cur, c, d);
            else if (l == 1) gemm4<KUH>(wh1, cur, c, d);
            else gemm4<KUH>(wh2, cur, c, d);
            const float pre = m4_reduce_scatter(c + d) + bias_own[1 + l];
            const float o = actf(pre);
            (toB ? bufB : bufA)[r * LDA + fcol] = o;
            save_act(n, 1 + l, o);
            save_pre(n, 1 + l, pre);
            sgn |= (o > 0.0f ? 1u : 0u) << (1 + l);
            if constexpr (MIL) { if (tdone < NTR) net_t(); }
            __syncthreads();
            tsync = true;
            cur = toB ? brow : arow;
        }
        {
            f32x4 c = {0.f, 0.f, 0.f, 0.f}, d = c;
            gemm4<KUH>(wo, cur, c, d);
            z = m4_reduce_scatter(c + d) + bias_own[NHID + 1];
        }
        if constexpr (MIL) {
#pragma unroll
            for (int i = 0; i < NTR; ++i) {        // what the drift's barriers did not cover
                if (tdone < NTR) {
                    if (!tsync) __syncthreads();
                    net_t();
                }
            }
        }
        // Euler on these kernels is differentiated by snsde_mfma_reverse_kernel, which takes the relu masks of the drift chain from z.
        // SRK (snsde_m4n_rev_kernel.h): every pass's z carries the pass's drift signs, the hidden sign of the net evaluation beside it
        // (bit NHID + 1) and - pass 3n + 2 - of the step's fourth evaluation (bit NHID + 2, known after the tail: stored there)
        constexpr int SRK_BITS = NHID + 1 + (NN == 2 ? 2 : 0);
        if (CF::METHOD == SNSDE_EULER) save_act(n, CF::ZSLOT, act_fn == 0 ? snsde_pack_signs(z, sgn, NHID + 1) : z);
        else if (MIL) save_act(n, CF::ZSLOT, act_fn == 0 ? snsde_pack_signs(z, sgn, NHID + 1 + (NN == 2 ? 1 : 0)) : z);      // (snsde_m4n_mil_rev_kernel.h)
        else if (stage != 2) save_act(n, CF::ZSLOT, act_fn == 0 ? snsde_pack_signs(z, sgn, SRK_BITS) : z);

        // ---- f, g and the update (own element) ----
        const float yin = SRK ? sk_y : yv;         // (the drift pass's input state is yv)
        const float f = f_lin ? z : fast_tanh(geo ? z * fast_tanh(yv) : z);
        if constexpr (!SRK) {
            float g, yn;
            if constexpr (MIL) { g = mg; yn = fmaf(g, dw, fmaf(f, h, yv)) + mv; }
            else { g = gfun(q, yv); yn = fmaf(g, dw, fmaf(f, h, yv)); }
            const float yold = yv;
            yv = yn;
            ybuf[r * LDY + fcol] = yn;
            if (row_ok) {
                if (a.traj) a.traj[(size_t)(n + 1) * BH + goff] = yn;
#ifdef SNSDE_DBG_M4N
                if (a.dW_out) a.dW_out[(size_t)n * BH + goff] = (n & 1) ? mdirect : mv;
#else
                if (a.dW_out) a.dW_out[(size_t)n * BH + goff] = dw;
#endif
                for (int k = cur_row.kfirst; k < cur_row.kfirst + cur_row.nout; ++k) {
                    const float w0 = a.out_w[2 * k], w1 = a.out_w[2 * k + 1];
                    const float o = (w0 == 0.0f) ? yn : snsde_interp_out(w0, w1, yold, yn);
                    if (!a.row_out) a.ys[(size_t)(k + 1) * BH + goff] = o;
                    else if (rslot == k + 1) a.ys[goff] = o;
                }
            }
            __syncthreads();
        } else {
            const float yb = yin, f0 = sk_f0, g0 = sk_g0;
            float h0n, h1n;       // next drift input / next net input
            if (stage == 0) {            // F0, G0 at (t0, y)
                sk_rh = 1.0f / h; sk_rsqh = 1.0f / sqh;
                sk_f0 = f;
                sk_g0 = gfun(q, yb);
                h0n = yb + f * h;
                h1n = yb + 0.25f * f * h + SRK_B1_10 * sk_g0 * sqh;
            } else if (stage == 1) {     // F1 at (t0 + h, H0_1), G1 at (t0 + h/4, H1_1)
                sk_f1 = f;
                const float g1 = gfun(q, sk_h1);
                sk_g1 = g1;
                const float du = sk_du;
                h0n = yb + 0.25f * f0 * h + 0.25f * f * h + (g0 + 0.5f * g1) * (du * sk_rh);
                h1n = yb + f0 * h + SRK_B1_20 * g0 * sqh;
            } else {                     // F2 at (t0 + h/2, H0_2), G2 at (t0 + h, H1_2); H1_3 for the tail
                sk_g2 = gfun(q, sk_h1);
                h1n = yb + 0.25f * f * h + (SRK_B1_30 * g0 + SRK_B1_31 * sk_g1 + SRK_B1_32 * sk_g2) * sqh;
                h0n = 0.0f;
            }
            sk_h1 = h1n;
            gybuf[r * LDY + fcol] = h1n;
            if (stage < 2) {
                yv = h0n;
                ybuf[r * LDY + fcol] = h0n;
                if (a.stage_save && row_ok) {
                    a.stage_save[((size_t)(n + 1) * NP) * BH + goff] = h0n;
                    a.stage_save[((size_t)(n + 1) * NP + 1) * BH + goff] = h1n;
                }
                __syncthreads();
            } else {
                if (a.stage_save && row_ok) a.stage_save[((size_t)n * NP + 2) * BH + goff] = h1n;
                __syncthreads();
                // ---- tail: G3 = g(t0 + h/4, H1_3) ----
                float q3 = 0.0f;
                const float nh3 = net_l1(n, CF::ZSLOT + NN + 1, q3);
                if constexpr (NN == 2) sgn |= (nh3 > 0.0f ? 1u : 0u) << (NHID + 2);
                save_act(n, CF::ZSLOT, act_fn == 0 ? snsde_pack_signs(z, sgn, SRK_BITS) : z);
                __syncthreads();
                if constexpr (NN == 2) net_l2(n, CF::ZSLOT + NN + 2, q3);
                if (tid < M) { gybuf[tid * LDY + H] = nxt.nsn; gybuf[tid * LDY + H + 1] = nxt.ncs; }
                const float g3 = gfun(q3, h1n);
                const float f1 = sk_f1, g1 = sk_g1, g2 = sk_g2, ik = sk_dw, ik0 = sk_du;
                const float ikk = 0.5f * (ik * ik - h);
                const float ikkk = (ik * ik * ik - 3.0f * h * ik) * (1.0f / 6.0f);
                const float a1 = ik, a2 = ikk * sk_rsqh, a3 = ik0 * sk_rh, a4 = ikkk * sk_rh;
                const float w0 = srk_w0(a1, a2, a3, a4);
                const float w1 = srk_w1(a1, a2, a3, a4);
                const float w2 = srk_w2(a1, a2, a3, a4);
                float yn1 = yb + (f0 + f1) * (h * (1.0f / 6.0f)) + f * (h * (2.0f / 3.0f));
                yn1 += w0 * g0 + w1 * g1 + w2 * g2 + a4 * g3;
                sk_y = yn1; sk_h1 = yn1; yv = yn1;
                ybuf[r * LDY + fcol] = yn1;
                gybuf[r * LDY + fcol] = yn1;
                if (row_ok) {
                    if (a.stage_save) {
                        a.stage_save[((size_t)(n + 1) * NP) * BH + goff] = yn1;
                        a.stage_save[((size_t)(n + 1) * NP + 1) * BH + goff] = yn1;
                    }
                    if (a.traj) a.traj[(size_t)(ns + 1) * BH + goff] = yn1;
                    if (a.dW_out) a.dW_out[(size_t)ns * BH + goff] = ik;
                    if (a.dU_out) a.dU_out[(size_t)ns * BH + goff] = ik0;
                    for (int k = cur_row.kfirst; k < cur_row.kfirst + cur_row.nout; ++k) {
                        const float v0 = a.out_w[2 * k], v1 = a.out_w[2 * k + 1];
                        const float o = (v0 == 0.0f) ? yn1 : v0 * yb + v1 * yn1;
                        if (!a.row_out) a.ys[(size_t)(k + 1) * BH + goff] = o;
                        else if (rslot == k + 1) a.ys[goff] = o;
                    }
                }
                __syncthreads();
            }
        }
    }
}

template <class CF>
int launch_m4n(const MfmaArgs& a, hipStream_t stream) {
    if constexpr (!CF::FITS) return SNSDE_ERR_UNSUPPORTED;
    else {
        const size_t lds_bytes = (size_t)CF::LDS_FLOATS * sizeof(float);
        static SnsdeLdsAttr lds_attr;   // per instantiation and device
        if (const int rc = snsde_lds_attr(reinterpret_cast<const void*>(snsde_m4n_kernel<CF>), lds_bytes, lds_attr)) return rc;
        hipLaunchKernelGGL(snsde_m4n_kernel<CF>, dim3((a.B + 3) / 4), dim3(CF::NT), lds_bytes, stream, a);
        return hipGetLastError() == hipSuccess ? SNSDE_OK : SNSDE_ERR_LAUNCH;
    }
}

// host-side mirror of the instantiation rule (make_plan): does a 4-row-tile net kernel exist for this shape?
inline bool m4n_instantiated(int H, int KUX, int NHID, int NN, int METHOD) {
    if (!(H == 16 || H == 32 || H == 64 || H == 128)) return false;
    if (!(KUX == 0 || KUX == 2 || KUX == 5) || NHID < 0 || NHID > 3 || NN < 1 || NN > 2) return false;
    if (METHOD != SNSDE_SRK && METHOD != SNSDE_MILSTEIN && METHOD != SNSDE_EULER) return false;
    return m4n_nlds(H, KUX, NHID, NN, METHOD) >= 0;
}

template <int H>
int dispatch_m4n(const MfmaPlan& p, const MfmaArgs& a, hipStream_t st) {
#define SNSDE_N1(KUX_, NHID_, NN_, METH_) \
    if (p.KUXN == KUX_ && p.NHID == NHID_ && p.NN == NN_ && a.method == METH_) return launch_m4n<CfgN<H, KUX_, NHID_, NN_, METH_>>(a, st);
#define SNSDE_N2(KUX_, NHID_) SNSDE_N1(KUX_, NHID_, 1, SNSDE_SRK) SNSDE_N1(KUX_, NHID_, 2, SNSDE_SRK) \
                              SNSDE_N1(KUX_, NHID_, 1, SNSDE_MILSTEIN) SNSDE_N1(KUX_, NHID_, 2, SNSDE_MILSTEIN) \
                              SNSDE_N1(KUX_, NHID_, 1, SNSDE_EULER) SNSDE_N1(KUX_, NHID_, 2, SNSDE_EULER)
#define SNSDE_N3(KUX_) SNSDE_N2(KUX_, 0) SNSDE_N2(KUX_, 1) SNSDE_N2(KUX_, 2) SNSDE_N2(KUX_, 3)
    SNSDE_N3(0) SNSDE_N3(2) SNSDE_N3(5)
#undef SNSDE_N3
#undef SNSDE_N2
#undef SNSDE_N1
    return SNSDE_ERR_UNSUPPORTED;
}

int dispatch_m4n_h16(const MfmaPlan& p, const MfmaArgs& a, hipStream_t st);
int dispatch_m4n_h32(const MfmaPlan& p, const MfmaArgs& a, hipStream_t st);
int dispatch_m4n_h64(const MfmaPlan& p, const MfmaArgs& a, hipStream_t st);
int dispatch_m4n_h128(const MfmaPlan& p, const MfmaArgs& a, hipStream_t st);

}  // namespace snsde_mfma
