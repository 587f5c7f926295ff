// Adjoint of the SRID2 step through a DIFFUSION NET (noise_option 14 / 15 / 18 / 19) on the MFMA path, 4-row tiles:
// the backward of snsde_m4n_kernel<.., SNSDE_SRK>.  Per solver step (walked backwards) SEVEN transposed chains:
//
//     G3 | G2 || drift pass 2 | G1 || drift pass 1 | G0 || drift pass 0
//
// `G_e` = the net's chain  dL/dg_e -> (1 - g^2) sigmoid(theta) {y, direct term} -> [q > 0] -> ny1^T -> [h > 0] -> ny0_y^T
// (the cotangent of the state the e-th diffusion evaluation saw), `drift pass s` = the Euler adjoint's chain
// out^T -> [relu] -> hid^T .. -> first_y^T on pass 3n + s.  A net chain and the drift chain written beside it are
// independent (SRID2's stage dependencies, see the comments in the step loop) and share their barriers.  F_s, G_e, H0_s,
// H1_e are recomputed per lane from y_n, the saved pre-tanh drifts / net outputs and (I_k, I_k0); relu masks come from the
// forward's act_save.  Left for snsde_param_gradients: the per-layer deltas of every pass (delta_save: drift slots 0 ..
// NHID + 1, net slots NHID + 2 .., the step's fourth evaluation in a second set of net slots at pass 3n + 2) and the
// per-workgroup sums of dL/d sigmoid(theta).
// Weights as in the forward: what does not fit the register budget is parked in the wave's private LDS slice.
#pragma once
#include "snsde_m4n_kernel.h"

namespace snsde_mfma {

__host__ __device__ constexpr int m4nr_reg_budget(int H) { return H >= 128 ? 104 : 144; }
__host__ __device__ constexpr int m4nr_nlds(int H, int NHID, int NN) {
    const int KUH = H / 16, nm = NHID + 2 + NN;
    int regs = 4 * KUH * nm, nl = 0, blocks = 0;
    while (regs > m4nr_reg_budget(H) && nl < nm) {
        if (blocks + KUH > m4n_lds_cap_blocks(H)) break;
        regs -= 4 * KUH; blocks += KUH; ++nl;
    }
    return regs > m4nr_reg_budget(H) + 32 ? -1 : nl;
}

// VAR: the field-variant switches (tutorial NeuralSDEFunc, fields.py: smooth activations - their derivative from the pre-activations
// the forward saved behind the regular slots -, linear drift output, the net's linear output as the diffusion); a template
// parameter because the pre-activations of a step are 3 (NHID + 1) + 4 more prefetched registers per lane
template <int H_, int NHID_, int NN_, bool VAR_ = false>
struct CfgNR {
    static constexpr int H = H_, NHID = NHID_, NN = NN_;
    static constexpr bool VAR = VAR_;
    static constexpr int NW = H / 16, NT = NW * 64, WPS = NW >= 8 ? NW / 4 : 2, M = 4;
    static constexpr int KUH = H / 16, LDA = ld_for(16 * KUH, 16);
    static constexpr int ND = NHID + 2, NMAT = ND + NN;
    static constexpr int NSLOT = NHID + 2 + 2 * NN;       // act_save / delta_save slots per pass
    static constexpr int ZSLOT = NHID + 1, NB0 = NHID + 2;
    static constexpr int NLDS = m4nr_nlds(H, NHID, NN);
    static constexpr bool FITS = NLDS >= 0;
    static constexpr bool in_lds(int i) { return i >= 0 && i < NMAT && i >= NMAT - NLDS; }
    static constexpr int lds_w_off(int i) {
        int o = 0;
        for (int j = 0; j < i && j < NMAT; ++j) if (in_lds(j)) o += KUH * 256 * NW;
        return o;
    }
    static constexpr int NBUF = ND + 2;                   // drift buffers 0 .. ND - 1, two net buffers
    static constexpr int LDS_ACT = NBUF * M * LDA;
    static constexpr int LDS_FLOATS = LDS_ACT + lds_w_off(NMAT);
};

template <class CF>
__global__ void __launch_bounds__(CF::NT, CF::WPS) snsde_m4n_srk_reverse_kernel(RevArgs a) {
    constexpr int H = CF::H, M = 4, NT = CF::NT, NHID = CF::NHID, NN = CF::NN, ND = CF::ND, KUH = CF::KUH, LDA = CF::LDA;
    constexpr int NSLOT = CF::NSLOT, ZSLOT = CF::ZSLOT, NB0 = CF::NB0;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* nb0 = lds + ND * M * LDA;
    float* nb1 = nb0 + M * LDA;
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
    const uint32_t BH32 = (uint32_t)BH, SLBH = (uint32_t)NSLOT * BH32;      // uniform strides as 32-bit factors (uoff: scalar-unit products)
    const int lrow = r * LDA + fcol;               // own element inside an LDS buffer
    const int brow = r * LDA + 4 * s;              // this lane's B-operand row inside an LDS buffer

    // transposed matrices in chain order: out^T, hid^T (last first), first_y^T, [ny1^T], ny0_y^T
    WN<CF::in_lds(0), KUH> w0;
    WN<CF::in_lds(1), KUH> w1;
    WN<(ND > 2) && CF::in_lds(2), KUH> w2;
    WN<(ND > 3) && CF::in_lds(3), KUH> w3;
    WN<(ND > 4) && CF::in_lds(4), KUH> w4;
    WN<CF::in_lds(ND), KUH> wnA;
    WN<(NN > 1) && CF::in_lds(ND + 1), KUH> wnB;
    {
        auto slice = [&](int ix) { return wlds + CF::lds_w_off(ix) + wave * KUH * 256; };
        w0.load(a.ws + a.w_off[0], wave, lane, slice(0));
        w1.load(a.ws + a.w_off[1], wave, lane, slice(1));
        if constexpr (ND > 2) w2.load(a.ws + a.w_off[2], wave, lane, slice(2));
        if constexpr (ND > 3) w3.load(a.ws + a.w_off[3], wave, lane, slice(3));
        if constexpr (ND > 4) w4.load(a.ws + a.w_off[4], wave, lane, slice(4));
        wnA.load(a.ws + a.w_off[ND], wave, lane, slice(ND));
        if constexpr (NN > 1) wnB.load(a.ws + a.w_off[ND + 1], wave, lane, slice(ND + 1));
    }
    for (int i = tid; i < CF::LDS_ACT; i += NT) lds[i] = 0.0f;
    __syncthreads();

    const float sig_theta = snsde_sigmoid(a.params[a.off_theta]);
    const bool mul_y = (a.no == 15 || a.no == 19);
    const bool geo = a.geo != 0;
    constexpr bool VAR = CF::VAR;
    const bool smooth = VAR && a.act_fn != 0, f_lin = VAR && a.f_out != 0, g_raw = VAR && a.g_out != 0;
    const bool net_lin = VAR && a.g_out == SNSDE_DIFFUSION_RAW_NET;
    const float act_scale = a.act_fn == SNSDE_ACT_LIPSWISH ? 0.909f : 1.0f;
    const uint32_t ASL = VAR ? (uint32_t)a.nsave * BH32 : SLBH;      // act_save stride per pass (smooth: + the pre-activation slots)
    const float rowf = row_ok ? 1.0f : 0.0f;
    const int rslot = a.row_out ? a.row_out[rowc] : -1;
    const float gfin = a.row_out ? a.grad_ys[goff] : 0.0f;
    float adj = 0.0f, th_acc = 0.0f;
    int nsel = 0;                                  // NN == 1: consecutive net chains alternate their LDS buffer

    auto drift_gemm = [&](int g, const float* in, f32x4& c, f32x4& d) {
        if (g == 0) gemm4<KUH>(w0, in, c, d);
        else if (g == 1) gemm4<KUH>(w1, in, c, d);
        else if (g == 2) gemm4<KUH>(w2, in, c, d);
        else if (g == 3) gemm4<KUH>(w3, in, c, d);
        else gemm4<KUH>(w4, in, c, d);
    };

    // (round 4: the relu masks - drift chain of every pass, hidden layer of every net evaluation - are sign bits in the low mantissa
    //  bits of the pass's saved z, snsde_m4n_kernel.h; they were ten of the twenty (N, B, H) planes this kernel read per step)
    struct StepIn {
        float y, ik, ik0, z[3], q[4];
        float h, rdt; int nout, kfirst;
    };
    // VAR, smooth activations: pre-activations of the drift passes / the net's hidden layers (their own struct: members a flavour never
    // touches would travel through scratch with every cur = nxt, see snsde_mfma_reverse_kernel)
    struct PreIn { float dpre[3][NHID + 1], npre[4]; };
    constexpr int SRK_BITS = NHID + 1 + (NN == 2 ? 2 : 0);
    auto fetch = [&](int n, StepIn& p, PreIn& pp) {
        const size_t so = uoff(n, BH32) + goff;
        p.y = a.traj[so]; p.ik = a.dW[so]; p.ik0 = a.dU[so];
#pragma unroll
        for (int st = 0; st < 3; ++st) {
            const float* ap = a.act + uoff(3 * n + st, ASL) + goff;
            p.z[st] = ap[(size_t)ZSLOT * BH];
            p.q[st] = ap[(size_t)(ZSLOT + NN) * BH];
            if (st == 2) p.q[3] = ap[(size_t)(ZSLOT + 2 * NN) * BH];
            if constexpr (VAR) {
                if (smooth) {      // (wave-uniform)
#pragma unroll
                    for (int k = 0; k <= NHID; ++k) pp.dpre[st][k] = ap[(size_t)(NSLOT + k) * BH];
                    if constexpr (NN == 2) {
                        pp.npre[st] = ap[(size_t)(NSLOT + NHID + 1) * BH];
                        if (st == 2) pp.npre[3] = ap[(size_t)(NSLOT + NHID + 2) * BH];
                    }
                }
            }
        }
        const float* stp = a.step_tab + uoff(n, SNSDE_STEP_STRIDE);
        p.h = stp[1]; p.rdt = stp[6]; p.nout = __float_as_int(stp[8]); p.kfirst = __float_as_int(stp[9]);
    };

    // Two independent chains under common barriers.  Drift: input cotangent dz at the pre-tanh output of pass p (delta slot
    // 0), relu masks dm[]; returns first_y^T(..) for the own element.  Net: input cotangent qb at the net's output
    // pre-activation of an evaluation whose deltas go to pass pn, slots ns0 / ns0 + 1; hidden mask nm; returns ny0_y^T(..).
    auto chains = [&](bool do_d, int p, float dz, uint32_t dmbits, bool do_n, int pn, int ns0, float qb, bool nm,
                      float& d_res, float& n_res, const float* dfac = nullptr, float nfac = 0.0f) {      // dmbits: bit (NHID - k) = relu mask behind transposed GEMM k; nm: the net's hidden mask; VAR: dfac[layer] / nfac = the activations' derivatives instead
        float* nbA = (NN == 1 && nsel) ? nb1 : nb0;
        float* nbB = nb1;
        if (do_d) {
            lds[lrow] = dz;
            if (a.delta && row_ok) (a.delta + uoff(p, SLBH))[goff] = dz;
        }
        if (do_n) {
            nbA[lrow] = qb;
            if (a.delta && row_ok) (a.delta + uoff(pn, SLBH, ns0, BH32))[goff] = qb;
            if constexpr (NN == 1) nsel ^= 1;
        }
        __syncthreads();
        constexpr int PMAX = ND > NN ? ND : NN;
#pragma unroll
        for (int k = 0; k < PMAX; ++k) {
            bool more = false;
            if (do_d && k < ND) {
                f32x4 c = {0.f, 0.f, 0.f, 0.f}, d = c;
                drift_gemm(k, lds + k * M * LDA + brow, c, d);
                const float o = m4_reduce_scatter(c + d);
                if (k < ND - 1) {
                    float dv = ((dmbits >> (NHID - k)) & 1u) ? o : 0.0f;
                    if constexpr (VAR) { if (smooth) dv = o * dfac[NHID - k]; }
                    lds[(k + 1) * M * LDA + lrow] = dv;
                    if (a.delta && row_ok) (a.delta + uoff(p, SLBH, k + 1, BH32))[goff] = dv;
                    more = true;
                } else {
                    d_res = o;
                }
            }
            if (do_n && k < NN) {
                f32x4 c = {0.f, 0.f, 0.f, 0.f}, d = c;
                if (k == 0) gemm4<KUH>(wnA, nbA + brow, c, d);
                else gemm4<KUH>(wnB, nbB + brow, c, d);
                const float o = m4_reduce_scatter(c + d);
                if (k < NN - 1) {
                    float dv = nm ? o : 0.0f;
                    if constexpr (VAR) { if (smooth) dv = o * nfac; }
                    nbB[lrow] = dv;
                    if (a.delta && row_ok) (a.delta + uoff(pn, SLBH, ns0 + 1, BH32))[goff] = dv;
                    more = true;
                } else {
                    n_res = o;
                }
            }
            if (more) __syncthreads();
        }
    };

    StepIn cur, nxt;
    PreIn curp, nxtp;
    fetch(a.N - 1, cur, curp);
    for (int n = a.N - 1; n >= 0; --n) {
        nxt = cur;
        if (n > 0) fetch(n - 1, nxt, nxtp);
        const float h = cur.h, rdt = cur.rdt;
        float carry = 0.0f;
        for (int k = cur.kfirst; k < cur.kfirst + cur.nout; ++k) {
            const float w0o = a.out_w[2 * k], w1o = a.out_w[2 * k + 1];
            const float gk = a.row_out ? (rslot == k + 1 ? gfin : 0.0f) : (a.grad_ys + uoff(k + 1, BH32))[goff];
            if (w0o == 0.0f) adj += gk;
            else { adj = fmaf(w1o, gk, adj); carry = fmaf(w0o, gk, carry); }
        }
        if (row_ok && !a.adj0_only) (a.adj + uoff(n + 1, BH32))[goff] = adj;

        // ---- recompute the stage values of the step (own element) ----
        const float y = cur.y, ik = cur.ik, ik0 = cur.ik0;
        auto gate = [&](float hv) { return geo ? fast_tanh(hv) : 1.0f; };
        // g = tanh(sigmoid(theta) nan_to_num(raw)), raw = q or q * (input state); keeps (1 - g^2), clipped raw, finiteness
        auto gfun = [&](float q, float yy, float& om, float& rc, bool& fin) {
            const float raw = mul_y ? q * yy : q;
            fin = snsde_finite(raw);
            rc = snsde_nan_to_num(raw);
            if (g_raw) { om = 1.0f; fin = true; return raw; }      // (VAR: g = the raw value itself: dg / d raw = 1, no theta)
            const float g = fast_tanh(sig_theta * rc);
            om = 1.0f - g * g;
            return g;
        };
        float om0, om1, om2, om3, rc0, rc1, rc2, rc3;
        bool fi0, fi1, fi2, fi3;
        uint32_t zb[3];
        float zc[3];
#pragma unroll
        for (int st = 0; st < 3; ++st) {
            zb[st] = __builtin_bit_cast(uint32_t, cur.z[st]);
            zc[st] = smooth ? cur.z[st] : __builtin_bit_cast(float, zb[st] & ~((1u << SRK_BITS) - 1u));      // (smooth: z is saved as it is)
        }
        const float f0 = f_lin ? zc[0] : fast_tanh(zc[0] * gate(y));
        const float g0 = gfun(cur.q[0], y, om0, rc0, fi0);
        const float h01 = y + f0 * h;
        const float h11 = y + 0.25f * f0 * h + SRK_B1_10 * g0 * rdt;
        const float f1 = f_lin ? zc[1] : fast_tanh(zc[1] * gate(h01));
        const float g1 = gfun(cur.q[1], h11, om1, rc1, fi1);
        const float rh = 1.0f / h, rrdt = 1.0f / rdt;      // (the forward's two divisions per step; every `/ h`, `/ rdt` below multiplies)
        const float ik0h = ik0 * rh;
        const float h02 = y + 0.25f * f0 * h + 0.25f * f1 * h + (g0 + 0.5f * g1) * ik0h;
        const float f2 = f_lin ? zc[2] : fast_tanh(zc[2] * gate(h02));
        const float h12 = y + f0 * h + SRK_B1_20 * g0 * rdt;
        const float g2 = gfun(cur.q[2], h12, om2, rc2, fi2);
        const float h13 = y + 0.25f * f2 * h + (SRK_B1_30 * g0 + SRK_B1_31 * g1 + SRK_B1_32 * g2) * rdt;
        const float g3 = gfun(cur.q[3], h13, om3, rc3, fi3);
        (void)g3;

        const float av = adj;
        const float ikk = 0.5f * (ik * ik - h);
        const float ikkk = (ik * ik * ik - 3.0f * h * ik) * (1.0f / 6.0f);
        const float a1 = ik, a2 = ikk * rrdt, a3 = ik0h, a4 = ikkk * rh;
        const float wg0 = srk_w0(a1, a2, a3, a4);
        const float wg1 = srk_w1(a1, a2, a3, a4);
        const float wg2 = srk_w2(a1, a2, a3, a4);
        float yb = carry + av;
        float fb0 = av * (h * (1.0f / 6.0f)), fb1 = fb0, fb2 = av * (h * (2.0f / 3.0f));
        float gb0 = wg0 * av, gb1 = wg1 * av, gb2 = wg2 * av;
        const float gb3 = a4 * av;

        // cotangent of a diffusion evaluation -> input of its net chain (+ the direct term of raw = q * y); theta's share
        auto net_in = [&](float gb, float om, float rc, bool fin, float q, float hin, float& direct) {
            const float rb = g_raw ? gb : (fin ? gb * om * sig_theta : 0.0f);
            if (!g_raw) th_acc = fmaf(gb * om * rowf, rc, th_acc);
            direct = mul_y ? rb * q : 0.0f;
            float qb = mul_y ? rb * hin : rb;
            if constexpr (NN == 2) { if (!net_lin) qb = q > 0.0f ? qb : 0.0f; }      // (VAR: the net may end in its linear layer)
            return qb;
        };
        // cotangent of a drift evaluation F = tanh(z * gate(hin)) -> input of its chain (+ the gate's direct term)
        auto drift_in = [&](float fb, float F, float z, float hin, float& direct) {
            const float ty = gate(hin);
            const float dzt = f_lin ? fb : fb * (1.0f - F * F);
            direct = geo ? dzt * z * (1.0f - ty * ty) : 0.0f;
            return dzt * ty;
        };
        float dd, nd, dres = 0.0f, nres = 0.0f;
        // VAR, smooth activations: d act / dx at the saved pre-activations (drift layers of the three passes, the net's hidden layer of
        // the four evaluations) instead of the relu bits
        float df[VAR ? 3 : 1][NHID + 1], nf[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (VAR) {
            if (smooth) {
#pragma unroll
                for (int st = 0; st < 3; ++st)
#pragma unroll
                    for (int k = 0; k <= NHID; ++k) df[st][k] = swish_grad(curp.dpre[st][k], act_scale);
                if constexpr (NN == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) nf[e] = swish_grad(curp.npre[e], act_scale);
                }
            }
        }

        // ---- G3 = g(t0 + h/4, H1_3): the tail evaluation (second set of net slots of pass 3n + 2) ----
        float qb = net_in(gb3, om3, rc3, fi3, cur.q[3], h13, nd);
        chains(false, 0, 0.0f, zb[2], true, 3 * n + 2, NB0 + NN, qb, ((zb[2] >> (NHID + 2)) & 1u) != 0, dres, nres, nullptr, nf[3]);
        float hb = nres + nd;
        yb += hb; fb2 = fmaf(0.25f * h, hb, fb2);
        gb0 = fmaf(SRK_B1_30 * rdt, hb, gb0); gb1 = fmaf(SRK_B1_31 * rdt, hb, gb1); gb2 = fmaf(SRK_B1_32 * rdt, hb, gb2);
        // ---- G2 = g(t0 + h, H1_2) beside the drift at (t0 + h/2, H0_2) ----
        qb = net_in(gb2, om2, rc2, fi2, cur.q[2], h12, nd);
        float dz = drift_in(fb2, f2, zc[2], h02, dd);
        chains(true, 3 * n + 2, dz, zb[2], true, 3 * n + 2, NB0, qb, ((zb[2] >> (NHID + 1)) & 1u) != 0, dres, nres, df[VAR ? 2 : 0], nf[2]);
        hb = nres + nd;
        float d = dres + dd;
        yb += hb; fb0 = fmaf(h, hb, fb0); gb0 = fmaf(SRK_B1_20 * rdt, hb, gb0);
        yb += d;
        fb0 = fmaf(0.25f * h, d, fb0); fb1 = fmaf(0.25f * h, d, fb1);
        gb0 = fmaf(ik0h, d, gb0); gb1 = fmaf(0.5f * ik0h, d, gb1);
        // ---- G1 = g(t0 + h/4, H1_1) beside the drift at (t0 + h, H0_1) ----
        qb = net_in(gb1, om1, rc1, fi1, cur.q[1], h11, nd);
        dz = drift_in(fb1, f1, zc[1], h01, dd);
        chains(true, 3 * n + 1, dz, zb[1], true, 3 * n + 1, NB0, qb, ((zb[1] >> (NHID + 1)) & 1u) != 0, dres, nres, df[VAR ? 1 : 0], nf[1]);
        hb = nres + nd;
        d = dres + dd;
        yb += hb; fb0 = fmaf(0.25f * h, hb, fb0); gb0 = fmaf(SRK_B1_10 * rdt, hb, gb0);
        yb += d; fb0 = fmaf(h, d, fb0);
        // ---- G0 and F0, both at (t0, y) ----
        qb = net_in(gb0, om0, rc0, fi0, cur.q[0], y, nd);
        dz = drift_in(fb0, f0, zc[0], y, dd);
        chains(true, 3 * n, dz, zb[0], true, 3 * n, NB0, qb, ((zb[0] >> (NHID + 1)) & 1u) != 0, dres, nres, df[0], nf[0]);
        adj = yb + (nres + nd) + (dres + dd);
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
int launch_m4n_rev(const RevArgs& a, hipStream_t stream) {
    if constexpr (!CF::FITS) return SNSDE_ERR_UNSUPPORTED;
    else {
        const size_t lds_bytes = (size_t)CF::LDS_FLOATS * sizeof(float);
        static SnsdeLdsAttr lds_attr;   // per instantiation and device
        if (const int rc = snsde_lds_attr(reinterpret_cast<const void*>(snsde_m4n_srk_reverse_kernel<CF>), lds_bytes, lds_attr)) return rc;
        hipLaunchKernelGGL(snsde_m4n_srk_reverse_kernel<CF>, dim3((a.B + 3) / 4), dim3(CF::NT), lds_bytes, stream, a);
        return hipGetLastError() == hipSuccess ? SNSDE_OK : SNSDE_ERR_LAUNCH;
    }
}

inline bool m4n_rev_instantiated(int H, int NHID, int NN) {
    if (!(H == 16 || H == 32 || H == 64 || H == 128) || NHID < 0 || NHID > 3 || NN < 1 || NN > 2) return false;
    return m4nr_nlds(H, NHID, NN) >= 0;
}

template <int H>
int dispatch_m4n_rev(const RevPlan& p, const RevArgs& a, hipStream_t st) {
    if (a.act_fn != 0 || a.f_out != 0 || a.g_out != 0) {      // field variants: the two-layer nets of NeuralSDEFunc-shaped fields (fields.py)
#define SNSDE_NRV(NHID_) if (p.NHID == NHID_ && p.NN == 2) return launch_m4n_rev<CfgNR<H, NHID_, 2, true>>(a, st);
        SNSDE_NRV(0) SNSDE_NRV(1) SNSDE_NRV(2)
#undef SNSDE_NRV
        return SNSDE_ERR_UNSUPPORTED;
    }
#define SNSDE_NR(NHID_, NN_) if (p.NHID == NHID_ && p.NN == NN_) return launch_m4n_rev<CfgNR<H, NHID_, NN_>>(a, st);
    SNSDE_NR(0, 1) SNSDE_NR(0, 2) SNSDE_NR(1, 1) SNSDE_NR(1, 2) SNSDE_NR(2, 1) SNSDE_NR(2, 2) SNSDE_NR(3, 1) SNSDE_NR(3, 2)
#undef SNSDE_NR
    return SNSDE_ERR_UNSUPPORTED;
}

int dispatch_m4n_rev_h16(const RevPlan& p, const RevArgs& a, hipStream_t st);
int dispatch_m4n_rev_h32(const RevPlan& p, const RevArgs& a, hipStream_t st);
int dispatch_m4n_rev_h64(const RevPlan& p, const RevArgs& a, hipStream_t st);
int dispatch_m4n_rev_h128(const RevPlan& p, const RevArgs& a, hipStream_t st);

}  // namespace snsde_mfma
