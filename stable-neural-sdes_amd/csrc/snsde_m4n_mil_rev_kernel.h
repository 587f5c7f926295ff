// Adjoint of the Milstein step through a DIFFUSION NET (noise_option 14 / 15 / 18 / 19) on the MFMA path, 4-row tiles:
// the backward of snsde_m4n_kernel<.., SNSDE_MILSTEIN>.
//
//   y' = y + f h + g dW + m,   m = 1/2 J_g^T (g v),  v = dW^2 - h        (torchsde's diagonal-noise Milstein: the VJP of g)
//
// With a = dL/dy' held constant,  a . m = 1/2 sum_j v_j g_j p_j =: Psi(y, params)  with the TANGENT  p = J_g a.  So the step's
// cotangent is  a + h J_f^T a + grad_y[(a dW) . g] + grad_y Psi, a reverse pass over (raw, rawdot):
//     tangent      hdot = [h1 > 0] (W1_y a),  qdot = [q > 0] (W2 hdot),  rawdot = qdot {y} {+ q a}          (forward weights)
//     alpha = dPsi/d rawdot = v u / 2,   beta = dPsi/d raw = v u' rawdot / 2,   u = g g' (as a function of raw),
//     u' = sigma^2 (1 - g^2)(1 - 3 g^2),   rho = a dW g' + beta
//     cot q = rho {y} {+ alpha a},  cot qdot = alpha {y},  direct dy = {rho q + alpha qdot}                  ({..}: raw = q y)
//     reverse      [q > 0] cot q -> W2^T -> [h1 > 0] -> W1_y^T -> dy                                          (transposed weights)
//     second-order parameter terms (the tangent depends on W1_y, W2):  eps2 = [q > 0] cot qdot,  eps1 = [h1 > 0] W2^T eps2,
//         d W2 += eps2 hdot^T,   d W1_y += eps1 a^T            (left for snsde_param_gradients in extra delta slots)
// beside the drift's transposed chain (same barriers).  Up to ND + 3 NN - 1 GEMMs per step; weights that do not fit the
// register budget are parked in the wave's private LDS slice (snsde_m4n_kernel.h).
#pragma once
#include "snsde_m4n_rev_kernel.h"

namespace snsde_mfma {

__host__ __device__ constexpr int m4nm_nmat(int NHID, int NN) { return NHID + 2 + 2 * NN; }
__host__ __device__ constexpr int m4nm_nlds(int H, int NHID, int NN) {
    const int KUH = H / 16, nm = m4nm_nmat(NHID, NN);
    int regs = 4 * KUH * nm, nl = 0, blocks = 0;
    while (regs > m4nr_reg_budget(H) && nl < nm) {
        if (blocks + KUH > m4n_lds_cap_blocks(H)) break;
        regs -= 4 * KUH; blocks += KUH; ++nl;
    }
    return regs > m4nr_reg_budget(H) + 32 ? -1 : nl;
}

// VAR: the field-variant switches (NeuralSDEFunc-shaped fields, fields.py; see CfgNR): smooth activations - first AND second
// derivative at the pre-activations the forward saved behind the regular slots: the tangent hdot = act'(p1) (W1_y a) depends on the
// hidden pre-activation p1, so its cotangent reaches p1 through act''(p1) -, linear drift output, the net's linear output as g
template <int H_, int NHID_, int NN_, bool VAR_ = false>
struct CfgNM {
    static constexpr int H = H_, NHID = NHID_, NN = NN_;
    static constexpr bool VAR = VAR_;
    static constexpr int NW = H / 16, NT = NW * 64, WPS = NW >= 8 ? NW / 4 : 2, M = 4;
    static constexpr int KUH = H / 16, LDA = ld_for(16 * KUH, 16);
    static constexpr int ND = NHID + 2, NMAT = m4nm_nmat(NHID, NN);
    static constexpr int NSAVE = NHID + 2 + NN;                       // act_save slots per step
    static constexpr int NEXTRA = NN == 2 ? 3 : 1;                    // eps2, eps1, hdot | eps
    static constexpr int NDELTA = NSAVE + NEXTRA;                     // delta_save slots per step
    static constexpr int ZSLOT = NHID + 1, NB0 = NHID + 2;
    static constexpr int NLDS = m4nm_nlds(H, NHID, NN);
    static constexpr bool FITS = NLDS >= 0;
    static constexpr bool in_lds(int i) { return i >= 0 && i < NMAT && i >= NMAT - NLDS; }
    static constexpr int lds_w_off(int i) {
        int o = 0;
        for (int j = 0; j < i && j < NMAT; ++j) if (in_lds(j)) o += KUH * 256 * NW;
        return o;
    }
    static constexpr int NBUF = ND + 5;                               // drift buffers | a | hdot | delta2 | eps2 | delta1
    static constexpr int LDS_ACT = NBUF * M * LDA;
    static constexpr int LDS_FLOATS = LDS_ACT + lds_w_off(NMAT);
};

template <class CF>
__global__ void __launch_bounds__(CF::NT, CF::WPS) snsde_m4n_mil_reverse_kernel(RevArgs a) {
    constexpr int H = CF::H, M = 4, NT = CF::NT, NHID = CF::NHID, NN = CF::NN, ND = CF::ND, KUH = CF::KUH, LDA = CF::LDA;
    constexpr int NSAVE = CF::NSAVE, NDEL = CF::NDELTA, ZSLOT = CF::ZSLOT, NB0 = CF::NB0;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* bTA = lds + ND * M * LDA;          // a (input of the tangent chain)
    float* bTB = bTA + M * LDA;               // hdot
    float* bRA = bTB + M * LDA;               // delta2 = [q > 0] cot q
    float* bRB = bRA + M * LDA;               // eps2   = [q > 0] cot qdot
    float* bRC = bRB + M * LDA;               // delta1 = [h1 > 0] W2^T delta2
    float* wlds = lds + CF::LDS_ACT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 3, s = (lane >> 2) & 3, fsub = 4 * (lane >> 4);
    const int row0 = blockIdx.x * M, B = a.B;
    const int row = row0 + r, rowc = row < B ? row : B - 1;
    const bool row_ok = row < B;
    const size_t BH = (size_t)B * H;
    const int fcol = wave * 16 + fsub + s;
    const uint32_t goff = (uint32_t)(rowc * H + fcol);
    const uint32_t BH32 = (uint32_t)BH;      // uniform strides as 32-bit factors (uoff: scalar-unit products)
    const int lrow = r * LDA + fcol, brow = r * LDA + 4 * s;

    // matrices: drift chain (out^T, hid^T.., first_y^T) | W1_y | [W2 | W2^T] | W1_y^T
    WN<CF::in_lds(0), KUH> w0;
    WN<CF::in_lds(1), KUH> w1;
    WN<(ND > 2) && CF::in_lds(2), KUH> w2;
    WN<(ND > 3) && CF::in_lds(3), KUH> w3;
    WN<(ND > 4) && CF::in_lds(4), KUH> w4;
    constexpr int IX_1F = ND, IX_2F = ND + 1, IX_2T = ND + 2, IX_1T = ND + 2 * NN - 1;
    WN<CF::in_lds(IX_1F), KUH> n1f;
    WN<(NN > 1) && CF::in_lds(IX_2F), KUH> n2f;
    WN<(NN > 1) && CF::in_lds(IX_2T), KUH> n2t;
    WN<CF::in_lds(IX_1T), KUH> n1t;
    {
        auto slice = [&](int ix) { return wlds + CF::lds_w_off(ix) + wave * KUH * 256; };
        w0.load(a.ws + a.w_off[0], wave, lane, slice(0));
        w1.load(a.ws + a.w_off[1], wave, lane, slice(1));
        if constexpr (ND > 2) w2.load(a.ws + a.w_off[2], wave, lane, slice(2));
        if constexpr (ND > 3) w3.load(a.ws + a.w_off[3], wave, lane, slice(3));
        if constexpr (ND > 4) w4.load(a.ws + a.w_off[4], wave, lane, slice(4));
        n1f.load(a.ws + a.w_off[IX_1F], wave, lane, slice(IX_1F));
        if constexpr (NN > 1) n2f.load(a.ws + a.w_off[IX_2F], wave, lane, slice(IX_2F));
        if constexpr (NN > 1) n2t.load(a.ws + a.w_off[IX_2T], wave, lane, slice(IX_2T));
        n1t.load(a.ws + a.w_off[IX_1T], wave, lane, slice(IX_1T));
    }
    for (int i = tid; i < CF::LDS_ACT; i += NT) lds[i] = 0.0f;
    __syncthreads();

    const float sig = snsde_sigmoid(a.params[a.off_theta]);
    const bool mul_y = (a.no == 15 || a.no == 19);
    const bool geo = a.geo != 0;
    constexpr bool VAR = CF::VAR;
    const bool smooth = VAR && a.act_fn != 0, f_lin = VAR && a.f_out != 0, g_raw = VAR && a.g_out != 0;
    const bool net_lin = VAR && a.g_out == SNSDE_DIFFUSION_RAW_NET;
    const float act_scale = a.act_fn == SNSDE_ACT_LIPSWISH ? 0.909f : 1.0f;
    const uint32_t ASL = VAR ? (uint32_t)a.nsave * BH32 : (uint32_t)NSAVE * BH32;      // act_save stride per step (smooth: + the pre-activations)
    const float rowf = row_ok ? 1.0f : 0.0f;
    const int rslot = a.row_out ? a.row_out[rowc] : -1;
    const float gfin = a.row_out ? a.grad_ys[goff] : 0.0f;
    float adj = 0.0f, th_acc = 0.0f;

    auto drift_gemm = [&](int g, const float* in, f32x4& c, f32x4& d) {
        if (g == 0) gemm4<KUH>(w0, in, c, d);
        else if (g == 1) gemm4<KUH>(w1, in, c, d);
        else if (g == 2) gemm4<KUH>(w2, in, c, d);
        else if (g == 3) gemm4<KUH>(w3, in, c, d);
        else gemm4<KUH>(w4, in, c, d);
    };
    auto put_delta = [&](int n, int slot, float v) {
        if (a.delta && row_ok) (a.delta + uoff(n, (uint32_t)NDEL * BH32, slot, BH32))[goff] = v;
    };

    // (round 4: the relu masks - drift chain, the net's hidden layer - are sign bits in the low mantissa bits of the saved z)
    struct StepIn { float y, z, dw, q; float h; int nout, kfirst; };
    struct PreIn { float dpre[NHID + 1], npre; };      // VAR, smooth activations: pre-activations of the drift layers / the net's hidden layer
    constexpr int MIL_BITS = NHID + 1 + (NN == 2 ? 1 : 0);
    auto fetch = [&](int n, StepIn& p, PreIn& pp) {
        const size_t so = uoff(n, BH32) + goff;
        const float* ap = a.act + uoff(n, ASL) + goff;
        if constexpr (VAR) {
            if (smooth) {      // (wave-uniform)
#pragma unroll
                for (int k = 0; k <= NHID; ++k) pp.dpre[k] = ap[(size_t)(NSAVE + k) * BH];
                pp.npre = NN == 2 ? ap[(size_t)(NSAVE + NHID + 1) * BH] : 0.0f;
            }
        }
        p.y = a.traj[so]; p.dw = a.dW[so];
        p.z = ap[(size_t)ZSLOT * BH];
        p.q = ap[(size_t)(ZSLOT + NN) * BH];
        const float* stp = a.step_tab + uoff(n, SNSDE_STEP_STRIDE);
        p.h = stp[1]; p.nout = __float_as_int(stp[8]); p.kfirst = __float_as_int(stp[9]);
    };

    StepIn cur, nxt;
    PreIn curp, nxtp;
    fetch(a.N - 1, cur, curp);
    for (int n = a.N - 1; n >= 0; --n) {
        nxt = cur;
        if (n > 0) fetch(n - 1, nxt, nxtp);
        const float h = cur.h;
        float carry = 0.0f;
        for (int k = cur.kfirst; k < cur.kfirst + cur.nout; ++k) {
            const float w0o = a.out_w[2 * k], w1o = a.out_w[2 * k + 1];
            const float gk = a.row_out ? (rslot == k + 1 ? gfin : 0.0f) : (a.grad_ys + uoff(k + 1, BH32))[goff];
            if (w0o == 0.0f) adj += gk;
            else { adj = fmaf(w1o, gk, adj); carry = fmaf(w0o, gk, carry); }
        }
        if (row_ok) (a.adj + uoff(n + 1, BH32))[goff] = adj;
        const float av = adj, y = cur.y, dw = cur.dw, q = cur.q;
        const uint32_t zb = __builtin_bit_cast(uint32_t, cur.z);
        const float zc = smooth ? cur.z : __builtin_bit_cast(float, zb & ~((1u << MIL_BITS) - 1u));      // z with its sign bits cleared (smooth: saved as it is)
        const bool h1pos = NN == 2 && ((zb >> (NHID + 1)) & 1u) != 0;                       // [h1 > 0]: the net's hidden mask
        // VAR, smooth: act' of the drift layers, act' and act'' of the net's hidden layer (x sigma(x) family: act'' = c s (1 - s)(2 + x (1 - 2 s)))
        float dfac[NHID + 1], nf1 = 0.0f, nf2 = 0.0f;
#pragma unroll
        for (int k = 0; k <= NHID; ++k) dfac[k] = 0.0f;
        if constexpr (VAR) {
            if (smooth) {
#pragma unroll
                for (int k = 0; k <= NHID; ++k) dfac[k] = swish_grad(curp.dpre[k], act_scale);
                if constexpr (NN == 2) {
                    const float x = curp.npre;
                    const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
                    nf1 = act_scale * sg * fmaf(x, 1.0f - sg, 1.0f);
                    nf2 = act_scale * sg * (1.0f - sg) * fmaf(x, 1.0f - 2.0f * sg, 2.0f);
                }
            }
        }

        // drift: F = tanh(z gate(y)); cotangent a h
        const float ty = geo ? fast_tanh(y) : 1.0f;
        const float F = f_lin ? zc : fast_tanh(zc * ty);
        const float dzt = f_lin ? av * h : av * h * (1.0f - F * F);
        const float dz = dzt * ty;
        const float direct_d = geo ? dzt * zc * (1.0f - ty * ty) : 0.0f;
        // diffusion value and its derivatives in raw
        const float raw = mul_y ? q * y : q;
        const bool fin = snsde_finite(raw);
        const float rc = snsde_nan_to_num(raw);
        const float g = g_raw ? raw : fast_tanh(sig * rc), om = g_raw ? 0.0f : 1.0f - g * g;
        const float gp = g_raw ? 1.0f : (fin ? om * sig : 0.0f);      // dg / d raw (VAR: g = the raw value)
        const float v = fmaf(dw, dw, -h);

        // ---- phase 0: inputs of the drift chain (dL/d zout) and of the tangent chain (a) ----
        lds[lrow] = dz;
        put_delta(n, 0, dz);
        bTA[lrow] = av;
        __syncthreads();

        float d_res = 0.0f, n_res = 0.0f, direct_y = 0.0f, qdot = 0.0f;
        [[maybe_unused]] float s_raw = 0.0f;      // W1_y a before the activation's derivative (VAR)
        auto drift_phase = [&](int k) {       // transposed GEMM k of the drift chain; returns true when it wrote a buffer
            if (k >= ND) return false;
            f32x4 c = {0.f, 0.f, 0.f, 0.f}, d = c;
            drift_gemm(k, lds + k * M * LDA + brow, c, d);
            const float o = m4_reduce_scatter(c + d);
            if (k < ND - 1) {
                float dv = ((zb >> (NHID - k)) & 1u) ? o : 0.0f;
                if constexpr (VAR) { if (smooth) dv = o * dfac[NHID - k]; }
                lds[(k + 1) * M * LDA + lrow] = dv;
                put_delta(n, k + 1, dv);
                return true;
            }
            d_res = o;
            return false;
        };
        // after the tangent: cotangents of (q, qdot), theta's share, inputs of the reverse chains
        auto after_tangent = [&]() {
            const float rawdot = mul_y ? fmaf(qdot, y, q * av) : qdot;
            const float alpha = 0.5f * v * g * gp;
            const float beta = g_raw ? 0.5f * v * rawdot : (fin ? 0.5f * v * sig * sig * om * fmaf(-3.0f * g, g, 1.0f) * rawdot : 0.0f);      // (g = raw: u = g g' = raw, u' = 1)
            const float rho = fmaf(av * dw, gp, beta);
            float cq = mul_y ? fmaf(rho, y, alpha * av) : rho;
            float cqd = mul_y ? alpha * y : alpha;
            direct_y = mul_y ? fmaf(rho, q, alpha * qdot) : 0.0f;
            if (!g_raw) {
                th_acc = fmaf(av * dw * om * rowf, rc, th_acc);
                if (fin) th_acc = fmaf(0.5f * v * rawdot * om * rowf, fmaf(sig * rc, fmaf(-3.0f * g, g, 1.0f), g), th_acc);
            }
            if constexpr (NN == 2) { if (!net_lin) { cq = q > 0.0f ? cq : 0.0f; cqd = q > 0.0f ? cqd : 0.0f; } }      // (VAR: the net may end in its linear layer)
            bRA[lrow] = cq;
            put_delta(n, NB0, cq);
            put_delta(n, NSAVE, cqd);                 // eps2 (NN = 2) / eps (NN = 1): left factor of the second-order term
            if constexpr (NN == 2) bRB[lrow] = cqd;
        };
        if constexpr (NN == 2) {
            // phase 1: drift 0 || hdot = [h1 > 0] W1_y a
            drift_phase(0);
            {
                f32x4 c = {0.f, 0.f, 0.f, 0.f}, d = c;
                gemm4<KUH>(n1f, bTA + brow, c, d);
                const float hd_all = m4_reduce_scatter(c + d);      // (DPP: every lane takes part - never inside a select's branch)
                s_raw = hd_all;
                float hd = h1pos ? hd_all : 0.0f;
                if constexpr (VAR) { if (smooth) hd = nf1 * hd_all; }
                bTB[lrow] = hd;
                put_delta(n, NSAVE + 2, hd);
            }
            __syncthreads();
            // phase 2: drift 1 || qdot = [q > 0] W2 hdot, then the cotangents
            drift_phase(1);
            {
                f32x4 c = {0.f, 0.f, 0.f, 0.f}, d = c;
                gemm4<KUH>(n2f, bTB + brow, c, d);
                const float qd_all = m4_reduce_scatter(c + d);
                qdot = (net_lin || q > 0.0f) ? qd_all : 0.0f;
            }
            after_tangent();
            __syncthreads();
            // phase 3: drift 2 || delta1 = [h1 > 0] W2^T delta2 || eps1 = [h1 > 0] W2^T eps2
            drift_phase(2);
            {
                f32x4 c = {0.f, 0.f, 0.f, 0.f}, d = c;
                gemm4<KUH>(n2t, bRA + brow, c, d);
                const float d1_all = m4_reduce_scatter(c + d);
                f32x4 c2 = {0.f, 0.f, 0.f, 0.f}, d2 = c2;
                gemm4<KUH>(n2t, bRB + brow, c2, d2);
                const float e1_all = m4_reduce_scatter(c2 + d2);
                float d1 = h1pos ? d1_all : 0.0f, e1 = h1pos ? e1_all : 0.0f;
                if constexpr (VAR) {
                    if (smooth) {      // the tangent's cotangent reaches the hidden pre-activation through act''
                        d1 = fmaf(nf1, d1_all, nf2 * s_raw * e1_all);
                        e1 = nf1 * e1_all;
                    }
                }
                bRC[lrow] = d1;
                put_delta(n, NB0 + 1, d1);
                put_delta(n, NSAVE + 1, e1);
            }
            __syncthreads();
            // phase 4: drift 3 || dy = W1_y^T delta1
            bool more = drift_phase(3);
            {
                f32x4 c = {0.f, 0.f, 0.f, 0.f}, d = c;
                gemm4<KUH>(n1t, bRC + brow, c, d);
                n_res = m4_reduce_scatter(c + d);
            }
#pragma unroll
            for (int k = 4; k < ND; ++k) {            // (deeper drifts: the rest of their chain)
                if (more) __syncthreads();
                more = drift_phase(k);
            }
        } else {
            // phase 1: drift 0 || qdot = W1_y a, then the cotangents
            drift_phase(0);
            {
                f32x4 c = {0.f, 0.f, 0.f, 0.f}, d = c;
                gemm4<KUH>(n1f, bTA + brow, c, d);
                qdot = m4_reduce_scatter(c + d);
            }
            after_tangent();
            __syncthreads();
            // phase 2: drift 1 || dy = W1_y^T cot q
            bool more = drift_phase(1);
            {
                f32x4 c = {0.f, 0.f, 0.f, 0.f}, d = c;
                gemm4<KUH>(n1t, bRA + brow, c, d);
                n_res = m4_reduce_scatter(c + d);
            }
#pragma unroll
            for (int k = 2; k < ND; ++k) {
                if (more) __syncthreads();
                more = drift_phase(k);
            }
        }
        adj = av + carry + (d_res + direct_d) + (n_res + direct_y);
        cur = nxt;
        if constexpr (VAR) { if (smooth) curp = nxtp; }
    }
    if (row_ok) a.adj[goff] = adj + (a.row_out ? (rslot == 0 ? gfin : 0.0f) : a.grad_ys[goff]);
    if (a.dth_part) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) th_acc += __shfl_down(th_acc, off, 64);
        if (lane == 0) a.dth_part[blockIdx.x * CF::NW + wave] = th_acc;
    }
}

template <class CF>
int launch_m4n_mil_rev(const RevArgs& a, hipStream_t stream) {
    if constexpr (!CF::FITS) return SNSDE_ERR_UNSUPPORTED;
    else {
        const size_t lds_bytes = (size_t)CF::LDS_FLOATS * sizeof(float);
        static SnsdeLdsAttr lds_attr;   // per instantiation and device
        if (const int rc = snsde_lds_attr(reinterpret_cast<const void*>(snsde_m4n_mil_reverse_kernel<CF>), lds_bytes, lds_attr)) return rc;
        hipLaunchKernelGGL(snsde_m4n_mil_reverse_kernel<CF>, dim3((a.B + 3) / 4), dim3(CF::NT), lds_bytes, stream, a);
        return hipGetLastError() == hipSuccess ? SNSDE_OK : SNSDE_ERR_LAUNCH;
    }
}

inline bool m4n_mil_rev_instantiated(int H, int NHID, int NN) {
    if (!(H == 16 || H == 32 || H == 64 || H == 128) || NHID < 0 || NHID > 3 || NN < 1 || NN > 2) return false;
    return m4nm_nlds(H, NHID, NN) >= 0;
}

template <int H>
int dispatch_m4n_mil_rev(const RevPlan& p, const RevArgs& a, hipStream_t st) {
    if (a.act_fn != 0 || a.f_out != 0 || a.g_out != 0) {      // field variants: two-layer nets (NeuralSDEFunc)
#define SNSDE_NMV(NHID_) if (p.NHID == NHID_ && p.NN == 2) return launch_m4n_mil_rev<CfgNM<H, NHID_, 2, true>>(a, st);
        SNSDE_NMV(0) SNSDE_NMV(1) SNSDE_NMV(2)
#undef SNSDE_NMV
        return SNSDE_ERR_UNSUPPORTED;
    }
#define SNSDE_NMR(NHID_, NN_) if (p.NHID == NHID_ && p.NN == NN_) return launch_m4n_mil_rev<CfgNM<H, NHID_, NN_>>(a, st);
    SNSDE_NMR(0, 1) SNSDE_NMR(0, 2) SNSDE_NMR(1, 1) SNSDE_NMR(1, 2) SNSDE_NMR(2, 1) SNSDE_NMR(2, 2) SNSDE_NMR(3, 1) SNSDE_NMR(3, 2)
#undef SNSDE_NMR
    return SNSDE_ERR_UNSUPPORTED;
}

int dispatch_m4n_mil_rev_h16(const RevPlan& p, const RevArgs& a, hipStream_t st);
int dispatch_m4n_mil_rev_h32(const RevPlan& p, const RevArgs& a, hipStream_t st);
int dispatch_m4n_mil_rev_h64(const RevPlan& p, const RevArgs& a, hipStream_t st);
int dispatch_m4n_mil_rev_h128(const RevPlan& p, const RevArgs& a, hipStream_t st);

}  // namespace snsde_mfma
