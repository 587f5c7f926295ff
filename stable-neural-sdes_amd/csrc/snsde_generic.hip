// Generic fused Neural-SDE solver kernel for gfx950: every (input_option, noise_option) of the
// reference's Diffusion_model, any H / HH / C / NL.  One workgroup owns GR batch rows for ALL
// solver steps (rows are independent SDEs: no inter-workgroup traffic); the state y, the layer
// activations and X(t) live in LDS; weights are pre-transposed (W^T, K padded to 4) in the
// workspace so that lanes read consecutive output features (coalesced, L2-resident).
//
// Reference behaviour being fused, per solver step:
//   A10 X(t)           controldiffeq/interpolate.py:263-276
//   A7  f(t, y)        models_sde/neuralsde.py:295-302 (+186-231)
//   A8  g(t, y)        models_sde/neuralsde.py:304-307 (+233-293)
//   A5  bm(t0, t1)     supplied dW, or Philox4x32-10 keyed by (seed; global row, step, col/4)
//   A4/A6 Euler / Milstein update, A3 output interpolation (torchsde 0.2.5, restated)
// The MFMA fast path for the headline configurations lives in snsde_mfma.hip.
#include <stddef.h>

#include "snsde_internal.h"

namespace {

constexpr int GR = 8;    // batch rows per workgroup
constexpr int GT = 256;  // threads per workgroup (4 waves)

struct PackJob {
    SnsdeLayer layer[SNSDE_MAX_HIDDEN + 6];
    int32_t n;
};

__global__ void snsde_pack_kernel(const float* __restrict__ params, float* __restrict__ ws, PackJob job) {
    const SnsdeLayer L = snsde_kernarg_element<SnsdeLayer>(16 + offsetof(PackJob, layer), blockIdx.y);     // kernarg: params, ws, job
    const int total = L.Kpad * L.N;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int k = i / L.N, n = i - k * L.N;
        float v = 0.0f;
        if (k < L.K) {
            const int sk = (k < L.K - L.tshift) ? k + L.tshift : k - (L.K - L.tshift);
            v = params[L.src_w + n * L.K + sk];
        }
        ws[L.w + i] = v;
    }
}

// Time-only part of the diffusion for noise_option 12,13,16,17 (neuralsde.py:266-277): identical for
// every batch row, so it is evaluated once per solver step: gt[n][j] = noise_t(tau_n)[j]
// (relu applied for 16/17).
__global__ void snsde_time_table_kernel(const float* __restrict__ params, const float* __restrict__ step_tab,
                                        float* __restrict__ gt, SnsdeLayer nt0, SnsdeLayer nt1, int H, int no,
                                        int row_stride, int sin_col, int off_sigma = -1, int off_sigma_diag = -1) {
    extern __shared__ float hbuf[];
    const int n = blockIdx.x;   // one block per table row (a solver step, or an SRK stage time)
    const float* st = step_tab + (size_t)n * row_stride;
    snsde_time_table_row(params, st[0], st[sin_col], st[sin_col + 1], gt + (size_t)n * H, nt0, nt1, H, no, hbuf, off_sigma,
                         off_sigma_diag);
}

struct GenericArgs {
    SnsdeDims d;
    SnsdeNet net;
    const float* params;
    const float* ws;
    const float* coeffs;
    const float* step_tab;
    const int32_t* out_step;
    const float* out_w;
    const float* y0;
    const float* dW;
    float* ys;
    float* traj;
    float* dW_out;
    const int32_t* row_out;   // (B) per-row output slot (ys / grad_ys are then (B, H)) or null
    int64_t row_offset;
    uint64_t seed;
    const uint64_t* seed_dev;   // device-resident key (overrides seed) or null
    int32_t eval_mode;
    float* eval_f;
    float* eval_g;
    int32_t ldy, ldw, ldx;  // LDS row strides (floats, multiples of 4)
};

// out[r][n] = act(bias[n] + sum_k in[r][k] * wt[k][n]) for the GR rows of the tile.
template <int RPT>
__device__ __forceinline__ void dense_rows(const float* __restrict__ wt, float b, int K4, int N, const float* in,
                                           int ldin, float* out, int ldout, bool relu, int n, int r0) {
    constexpr int G = GR / RPT;
    float acc[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) acc[i] = b;
    const float* wp = wt + n;
    for (int k4 = 0; k4 < K4; ++k4) {
        const float w0 = wp[0], w1 = wp[N], w2 = wp[2 * N], w3 = wp[3 * N];
        wp += 4 * N;
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const float4 a = *reinterpret_cast<const float4*>(in + (r0 + i * G) * ldin + 4 * k4);
            acc[i] = fmaf(a.x, w0, acc[i]);
            acc[i] = fmaf(a.y, w1, acc[i]);
            acc[i] = fmaf(a.z, w2, acc[i]);
            acc[i] = fmaf(a.w, w3, acc[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) out[(r0 + i * G) * ldout + n] = relu ? fmaxf(acc[i], 0.0f) : acc[i];
}

__device__ void dense(const float* __restrict__ params, const float* __restrict__ ws, const SnsdeLayer& L,
                      const float* in, int ldin, float* out, int ldout, bool relu) {
    const int N = L.N, K4 = L.Kpad >> 2;
    const float* wt = ws + L.w;
    const int np = (N + 63) & ~63;
    if (np <= 64) {            // 4 row groups of 64 lanes
        const int n = threadIdx.x & 63, rg = threadIdx.x >> 6;
        if (n < N) dense_rows<GR / 4>(wt, params[L.src_b + n], K4, N, in, ldin, out, ldout, relu, n, rg);
    } else if (np <= 128) {    // 2 row groups
        const int n = threadIdx.x & 127, rg = threadIdx.x >> 7;
        if (n < N) dense_rows<GR / 2>(wt, params[L.src_b + n], K4, N, in, ldin, out, ldout, relu, n, rg);
    } else {
        for (int n = threadIdx.x; n < N; n += GT)
            dense_rows<GR>(wt, params[L.src_b + n], K4, N, in, ldin, out, ldout, relu, n, 0);
    }
}

// ---- wide-workgroup variants (SRK forward, adjoints): GW threads, one output column per thread, the GR rows split over
// GW / NP thread groups; compiled once (noinline) - these kernels call them from dozens of sites, and inlined copies drove
// the register allocator into kilobytes of scratch per lane.  Per-output summation order is the same as in dense().
constexpr int GW = 512;

template <int NP>
__device__ __forceinline__ void dense_group(const float* __restrict__ wt, const float* __restrict__ bias, int K4, int N,
                                            const float* in, int ldin, float* out, int ldout, bool relu) {
    constexpr int G0 = GW / NP, RG = G0 > GR ? GR : G0;
    const int n = threadIdx.x % NP, rg = threadIdx.x / NP;
    if (rg < RG && n < N) dense_rows<GR / RG>(wt, bias[n], K4, N, in, ldin, out, ldout, relu, n, rg);
}

__device__ __noinline__ void dense_w(const float* __restrict__ params, const float* __restrict__ ws, const SnsdeLayer& L,
                                     const float* in, int ldin, float* out, int ldout, bool relu, const float* zero_bias = nullptr) {
    const int N = L.N, K4 = L.Kpad >> 2;
    const float* wt = ws + L.w;
    const float* bias = zero_bias ? zero_bias : params + L.src_b;     // zero_bias: N zeros (tangent passes carry no bias)
    if (N <= 64) dense_group<64>(wt, bias, K4, N, in, ldin, out, ldout, relu);
    else if (N <= 128) dense_group<128>(wt, bias, K4, N, in, ldin, out, ldout, relu);
    else if (N <= 256) dense_group<256>(wt, bias, K4, N, in, ldin, out, ldout, relu);
    else if (N <= 512) dense_group<512>(wt, bias, K4, N, in, ldin, out, ldout, relu);
    else {
        for (int n = threadIdx.x; n < N; n += GW)
            dense_rows<GR>(wt, bias[n], K4, N, in, ldin, out, ldout, relu, n, 0);
    }
}

__global__ void __launch_bounds__(GT) snsde_generic_kernel(GenericArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const SnsdeDims& d = a.d;
    const SnsdeNet& net = a.net;
    const int H = d.H, C = d.C, B = d.B, io = d.io, no = d.no;
    const int ldy = a.ldy, ldw = a.ldw, ldx = a.ldx;
    float* ybuf = lds;
    float* bufA = ybuf + GR * ldy;
    float* bufB = bufA + GR * ldw;
    float* bufC = bufB + GR * ldw;
    float* xbuf = bufC + GR * ldw;
    const int lds_floats = GR * (ldy + 3 * ldw + ldx);
    const int tid = threadIdx.x;
    const int row0 = blockIdx.x * GR;

    for (int i = tid; i < lds_floats; i += GT) lds[i] = 0.0f;
    __syncthreads();
    const float* ysrc = a.eval_mode ? a.y0 : a.y0;
    for (int i = tid; i < GR * H; i += GT) {
        const int r = i / H, j = i - r * H, row = row0 + r;
        if (row < B) {
            const float v = ysrc[(size_t)row * H + j];
            ybuf[r * ldy + j] = v;
            if (!a.eval_mode) {
                a.ys[(size_t)row * H + j] = v;
                if (a.traj) a.traj[(size_t)row * H + j] = v;
            }
        }
    }
    const float sig_theta = snsde_sigmoid(a.params[net.off_theta]);
    const float exp_sigma = (net.off_sigma >= 0) ? expf(a.params[net.off_sigma]) : 0.0f;
    const bool uses_x = (io == 0 || io == 2 || io == 4 || io == 6);
    const bool uses_emb = (io == 2 || io == 4 || io == 6);
    const bool noise_net = (no == 14 || no == 15 || no == 18 || no == 19);
    const bool noise_tab = (no == 12 || no == 13 || no == 16 || no == 17);
    const float* gt = a.ws + (net.gt_tab >= 0 ? net.gt_tab : 0);
    const int Q = (H + 3) >> 2;
    const size_t BH = (size_t)B * H;
    int kout = 0;

    for (int n = 0; n < d.N; ++n) {
        const float* st = a.step_tab + (size_t)n * SNSDE_STEP_STRIDE;
        const float t0 = st[0], h = st[1], sn = st[2], cs = st[3], frac = st[4], sqh = st[6];
        const int idx = __float_as_int(st[5]);
        if (tid < GR) {
            ybuf[tid * ldy + H] = sn;
            ybuf[tid * ldy + H + 1] = cs;
        }
        if (uses_x) {
            for (int i = tid; i < GR * C; i += GT) {
                const int r = i / C, c = i - r * C, row = row0 + r;
                float v = 0.0f;
                if (row < B) {
                    const float* cp = a.coeffs + ((size_t)row * (d.L - 1) + idx) * (4 * C) + c;
                    v = snsde_spline_eval(cp[0], cp[C], cp[2 * C], cp[3 * C], frac);
                }
                xbuf[r * ldx + c] = v;
            }
        }
        __syncthreads();
        // ---- drift f (neuralsde.py:295-302) ----
        float* cur;
        float* oth;
        if (io == 0) {
            dense(a.params, a.ws, net.init, xbuf, ldx, bufA, ldw, true);
            cur = bufA; oth = bufB;
        } else if (!uses_emb) {
            dense(a.params, a.ws, net.in, ybuf, ldy, bufA, ldw, true);
            cur = bufA; oth = bufB;
        } else {
            dense(a.params, a.ws, net.in, ybuf, ldy, bufA, ldw, false);        // yy  -> cat[:, 0:H)
            dense(a.params, a.ws, net.init, xbuf, ldx, bufA + H, ldw, false);  // Xt  -> cat[:, H:2H)
            __syncthreads();
            dense(a.params, a.ws, net.emb, bufA, ldw, bufB, ldw, true);
            cur = bufB; oth = bufA;
        }
        __syncthreads();
        for (int l = 0; l < net.n_hid; ++l) {
            dense(a.params, a.ws, net.hid[l], cur, ldw, oth, ldw, true);
            float* t = cur; cur = oth; oth = t;
            __syncthreads();
        }
        dense(a.params, a.ws, net.out, cur, ldw, oth, ldw, false);
        float* zbuf = oth;
        float* nbuf = bufC;
        if (noise_net) {  // neuralsde.py:270-273, 278-281
            dense(a.params, a.ws, net.ny0, ybuf, ldy, cur, ldw, no >= 18);
            __syncthreads();
            if (no >= 18) dense(a.params, a.ws, net.ny1, cur, ldw, bufC, ldw, true);
            else nbuf = cur;
        }
        __syncthreads();
        // ---- diffusion g, Brownian increment, state update ----
        int kout_next = kout;
        for (int i = tid; i < GR * Q; i += GT) {
            const int r = i / Q, q = i - r * Q, row = row0 + r;
            if (row >= B) continue;
            float zn[4] = {0.f, 0.f, 0.f, 0.f};
            if (!a.eval_mode && a.dW == nullptr) {
                // (row, 4-step block, column) -> 4 normals; this kernel recomputes the call each step and keeps z[n&3]
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float z4[4];
                    snsde_philox_normal4(a.seed_dev ? *a.seed_dev : a.seed, (uint32_t)(a.row_offset + row), (uint32_t)(n >> 2), (uint32_t)(4 * q + e), z4);
                    zn[e] = (n & 3) == 0 ? z4[0] : (n & 3) == 1 ? z4[1] : (n & 3) == 2 ? z4[2] : z4[3];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 4 * q + e;
                if (j >= H) break;
                const float y = ybuf[r * ldy + j];
                float z = zbuf[r * ldw + j];
                if (io == 5 || io == 6) z *= tanhf(y);
                const float f = tanhf(z);
                float raw = 0.0f, draw = 0.0f;  // raw diffusion and d raw / d y (Milstein)
                switch (no) {
                    case 0: break;
                    case 1: raw = exp_sigma; break;
                    case 2: raw = exp_sigma * t0; break;
                    case 3: raw = exp_sigma * y; draw = exp_sigma; break;
                    case 4: raw = expf(a.params[net.off_sigma_diag + j]); break;
                    case 5: raw = expf(a.params[net.off_sigma_diag + j]) * t0; break;
                    case 6: draw = expf(a.params[net.off_sigma_diag + j]); raw = draw * y; break;
                    case 7: raw = sqrtf(y); break;
                    case 8: raw = y * y * y; draw = 3.0f * y * y; break;
                    case 9: raw = snsde_sigmoid(y); draw = raw * (1.0f - raw); break;
                    case 10: raw = fmaxf(y, 0.0f); draw = y > 0.0f ? 1.0f : 0.0f; break;
                    case 11: raw = t0 * y; draw = t0; break;
                    case 12: case 16: raw = gt[(size_t)n * H + j]; break;
                    case 13: case 17: draw = gt[(size_t)n * H + j]; raw = draw * y; break;
                    case 14: case 18: raw = nbuf[r * ldw + j]; break;
                    case 15: case 19: raw = nbuf[r * ldw + j] * y; break;
                }
                const float g = tanhf(sig_theta * snsde_nan_to_num(raw));
                if (a.eval_mode) {
                    a.eval_f[(size_t)row * H + j] = f;
                    a.eval_g[(size_t)row * H + j] = g;
                    continue;
                }
                const float dw = a.dW ? a.dW[(size_t)n * BH + (size_t)row * H + j] : zn[e] * sqh;
                float ynew = fmaf(g, dw, fmaf(f, h, y));
                if (d.method == SNSDE_MILSTEIN) {
                    const float fin = snsde_finite(raw) ? 1.0f : 0.0f;  // finite raw
                    const float dg = (1.0f - g * g) * sig_theta * draw * fin;
                    ynew = fmaf(0.5f * (g * dg), fmaf(dw, dw, -h), ynew);
                }
                ybuf[r * ldy + j] = ynew;
                if (a.traj) a.traj[(size_t)(n + 1) * BH + (size_t)row * H + j] = ynew;
                if (a.dW_out) a.dW_out[(size_t)n * BH + (size_t)row * H + j] = dw;
                int k = kout;
                while (k < d.T - 1 && a.out_step[k] == n) {
                    const float w0 = a.out_w[2 * k], w1 = a.out_w[2 * k + 1];
                    const float o = (w0 == 0.0f) ? ynew : snsde_interp_out(w0, w1, y, ynew);
                    if (!a.row_out) a.ys[(size_t)(k + 1) * BH + (size_t)row * H + j] = o;
                    else if (a.row_out[row] == k + 1) a.ys[(size_t)row * H + j] = o;
                    ++k;
                }
            }
        }
        if (a.eval_mode) return;
        while (kout_next < d.T - 1 && a.out_step[kout_next] == n) ++kout_next;
        kout = kout_next;
        __syncthreads();
    }
}

// =====================================================================================================
// SRK (SRID2, strong order 1.5 for diagonal noise; torchsde `method='srk'`, the torch_ists default
// nsde_model.py:67): per step three drift evaluations f(t0,y), f(t0+h,H0_1), f(t0+h/2,H0_2) and four diffusion
// evaluations g(t0,y), g(t0+h/4,H1_1), g(t0+h,H1_2), g(t0+h/4,H1_3), combined with the increments
// I_k, I_k0 (space-time Levy integral), I_kk=(I_k^2-h)/2, I_kkk=(I_k^3-3hI_k)/6.  Tableau: oracle/sde_oracle.py.
// Same tile structure as the generic Euler kernel; every option pair (input_option, noise_option).
// =====================================================================================================
struct SrkArgs {
    GenericArgs g;
    const float* srk_tab;   // (N, 4, SNSDE_SRK_STRIDE): stage times c = 0, 1/4, 1/2, 1
    const float* dU;
    float* dU_out;
    int32_t ldf;
};

__global__ void __launch_bounds__(GW) snsde_generic_srk_kernel(SrkArgs sa) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const GenericArgs& a = sa.g;
    const SnsdeDims& d = a.d;
    const SnsdeNet& net = a.net;
    const int H = d.H, C = d.C, B = d.B, io = d.io, no = d.no;
    const int ldy = a.ldy, ldw = a.ldw, ldx = a.ldx, ldf = sa.ldf;
    float* Y = lds;                       // state y_n
    float* S0 = Y + GR * ldy;             // drift stage state (+ time features)
    float* S1 = S0 + GR * ldy;            // diffusion stage state (+ time features)
    float* bufA = S1 + GR * ldy;
    float* bufB = bufA + GR * ldw;
    float* bufC = bufB + GR * ldw;
    float* xbuf = bufC + GR * ldw;
    float* V = xbuf + GR * ldx;           // 8 value planes [GR][ldf]: f0 f1 f2 g0 g1 g2 dW dU
    const int plane = GR * ldf;
    float* F0 = V; float* F1 = V + plane; float* F2 = V + 2 * plane;
    float* G0 = V + 3 * plane; float* G1 = V + 4 * plane; float* G2 = V + 5 * plane;
    float* DW = V + 6 * plane; float* DU = V + 7 * plane;
    const int lds_floats = GR * (3 * ldy + 3 * ldw + ldx + 8 * ldf);
    const int tid = threadIdx.x, row0 = blockIdx.x * GR;
    for (int i = tid; i < lds_floats; i += GW) lds[i] = 0.0f;
    __syncthreads();
    for (int i = tid; i < GR * H; i += GW) {
        const int r = i / H, j = i - r * H, row = row0 + r;
        if (row < B) {
            const float v = a.y0[(size_t)row * H + j];
            Y[r * ldy + j] = v;
            a.ys[(size_t)row * H + j] = v;
            if (a.traj) a.traj[(size_t)row * H + j] = v;
        }
    }
    const float sig_theta = snsde_sigmoid(a.params[net.off_theta]);
    const float exp_sigma = (net.off_sigma >= 0) ? expf(a.params[net.off_sigma]) : 0.0f;
    const bool uses_x = (io == 0 || io == 2 || io == 4 || io == 6);
    const bool uses_emb = (io == 2 || io == 4 || io == 6);
    const bool noise_net = (no == 14 || no == 15 || no == 18 || no == 19);
    const float* gt = a.ws + (net.gt_tab >= 0 ? net.gt_tab : 0);
    const size_t BH = (size_t)B * H;
    int kout = 0;
    __syncthreads();

    auto for_elems = [&](auto&& fn) {
        for (int i = tid; i < GR * H; i += GW) {
            const int r = i / H, j = i - r * H;
            fn(r, j, row0 + r);
        }
        __syncthreads();
    };
    // f(tp, state in sbuf) -> fout plane
    auto drift = [&](float* sbuf, const float* tp, float* fout) {
        const float frac = tp[3];
        const int idx = __float_as_int(tp[4]);
        if (tid < GR) { sbuf[tid * ldy + H] = tp[1]; sbuf[tid * ldy + H + 1] = tp[2]; }
        if (uses_x) {
            for (int i = tid; i < GR * C; i += GW) {
                const int r = i / C, c = i - r * C, row = row0 + r;
                float v = 0.0f;
                if (row < B) {
                    const float* cp = a.coeffs + ((size_t)row * (d.L - 1) + idx) * (4 * C) + c;
                    v = snsde_spline_eval(cp[0], cp[C], cp[2 * C], cp[3 * C], frac);
                }
                xbuf[r * ldx + c] = v;
            }
        }
        __syncthreads();
        float* cur;
        float* oth;
        if (io == 0) {
            dense_w(a.params, a.ws, net.init, xbuf, ldx, bufA, ldw, true);
            cur = bufA; oth = bufB;
        } else if (!uses_emb) {
            dense_w(a.params, a.ws, net.in, sbuf, ldy, bufA, ldw, true);
            cur = bufA; oth = bufB;
        } else {
            dense_w(a.params, a.ws, net.in, sbuf, ldy, bufA, ldw, false);
            dense_w(a.params, a.ws, net.init, xbuf, ldx, bufA + H, ldw, false);
            __syncthreads();
            dense_w(a.params, a.ws, net.emb, bufA, ldw, bufB, ldw, true);
            cur = bufB; oth = bufA;
        }
        __syncthreads();
        for (int l = 0; l < net.n_hid; ++l) {
            dense_w(a.params, a.ws, net.hid[l], cur, ldw, oth, ldw, true);
            float* t = cur; cur = oth; oth = t;
            __syncthreads();
        }
        dense_w(a.params, a.ws, net.out, cur, ldw, oth, ldw, false);
        __syncthreads();
        for_elems([&](int r, int j, int) {
            float z = oth[r * ldw + j];
            if (io == 5 || io == 6) z *= tanhf(sbuf[r * ldy + j]);
            fout[r * ldf + j] = tanhf(z);
        });
    };
    // diffusion-net output buffer for the state in sbuf at stage time tp (noise_option 14/15/18/19)
    auto diffusion_net = [&](float* sbuf, const float* tp) -> const float* {
        if (!noise_net) return bufC;
        if (tid < GR) { sbuf[tid * ldy + H] = tp[1]; sbuf[tid * ldy + H + 1] = tp[2]; }
        __syncthreads();
        dense_w(a.params, a.ws, net.ny0, sbuf, ldy, bufA, ldw, no >= 18);
        __syncthreads();
        if (no < 18) return bufA;
        dense_w(a.params, a.ws, net.ny1, bufA, ldw, bufC, ldw, true);
        __syncthreads();
        return bufC;
    };
    auto g_elem = [&](float y, float t, int n, int slot, int r, int j, const float* nb) {
        float raw = 0.0f;
        switch (no) {
            case 0: break;
            case 1: raw = exp_sigma; break;
            case 2: raw = exp_sigma * t; break;
            case 3: raw = exp_sigma * y; break;
            case 4: raw = expf(a.params[net.off_sigma_diag + j]); break;
            case 5: raw = expf(a.params[net.off_sigma_diag + j]) * t; break;
            case 6: raw = expf(a.params[net.off_sigma_diag + j]) * y; break;
            case 7: raw = sqrtf(y); break;
            case 8: raw = y * y * y; break;
            case 9: raw = snsde_sigmoid(y); break;
            case 10: raw = fmaxf(y, 0.0f); break;
            case 11: raw = t * y; break;
            case 12: case 16: raw = gt[((size_t)n * 4 + slot) * H + j]; break;
            case 13: case 17: raw = gt[((size_t)n * 4 + slot) * H + j] * y; break;
            case 14: case 18: raw = nb[r * ldw + j]; break;
            case 15: case 19: raw = nb[r * ldw + j] * y; break;
        }
        return tanhf(sig_theta * snsde_nan_to_num(raw));
    };
    auto diffusion = [&](float* sbuf, const float* tp, int n, int slot, float* gout) {
        const float* nb = diffusion_net(sbuf, tp);
        const float t = tp[0];
        for_elems([&](int r, int j, int) { gout[r * ldf + j] = g_elem(sbuf[r * ldy + j], t, n, slot, r, j, nb); });
    };

    for (int n = 0; n < d.N; ++n) {
        const float* st = a.step_tab + (size_t)n * SNSDE_STEP_STRIDE;
        const float h = st[1], rdt = st[6];
        const float* tp0 = sa.srk_tab + (size_t)n * 4 * SNSDE_SRK_STRIDE;   // t0
        const float* tpq = tp0 + SNSDE_SRK_STRIDE;                          // t0 + h/4
        const float* tph = tp0 + 2 * SNSDE_SRK_STRIDE;                      // t0 + h/2
        const float* tp1 = tp0 + 3 * SNSDE_SRK_STRIDE;                      // t0 + h
        // increments of the step
        for_elems([&](int r, int j, int row) {
            float dw = 0.0f, du = 0.0f;
            if (row < B) {
                const size_t off = (size_t)n * BH + (size_t)row * H + j;
                if (a.dW) { dw = a.dW[off]; du = sa.dU[off]; }
                else {
                    float z4[4], x4[4];
                    snsde_philox_normal4(a.seed_dev ? *a.seed_dev : a.seed, (uint32_t)(a.row_offset + row), (uint32_t)(n >> 2), (uint32_t)j, z4, 0u);
                    snsde_philox_normal4(a.seed_dev ? *a.seed_dev : a.seed, (uint32_t)(a.row_offset + row), (uint32_t)(n >> 2), (uint32_t)j, x4, 1u);
                    const int k = n & 3;
                    dw = (k == 0 ? z4[0] : k == 1 ? z4[1] : k == 2 ? z4[2] : z4[3]) * rdt;
                    const float xi = k == 0 ? x4[0] : k == 1 ? x4[1] : k == 2 ? x4[2] : x4[3];
                    du = h * fmaf(sqrtf(h / 12.0f), xi, 0.5f * dw);
                }
                if (a.dW_out) a.dW_out[off] = dw;
                if (sa.dU_out) sa.dU_out[off] = du;
            }
            DW[r * ldf + j] = dw;
            DU[r * ldf + j] = du;
            S0[r * ldy + j] = Y[r * ldy + j];
            S1[r * ldy + j] = Y[r * ldy + j];
        });
        // stage 0
        drift(S0, tp0, F0);
        diffusion(S1, tp0, n, 0, G0);
        // stage 1: H0 = y + f0 h ; H1 = y + f0 h/4 - g0 sqrt(h)/2
        for_elems([&](int r, int j, int) {
            const float y = Y[r * ldy + j], f0 = F0[r * ldf + j], g0 = G0[r * ldf + j];
            S0[r * ldy + j] = y + f0 * h;
            S1[r * ldy + j] = y + 0.25f * f0 * h + SRK_B1_10 * g0 * rdt;
        });
        drift(S0, tp1, F1);
        diffusion(S1, tpq, n, 1, G1);
        // stage 2: H0 = y + (f0 + f1) h/4 + (g0 + g1/2) I_k0/h ; H1 = y + f0 h + g0 sqrt(h)
        for_elems([&](int r, int j, int) {
            const float y = Y[r * ldy + j], f0 = F0[r * ldf + j], f1 = F1[r * ldf + j];
            const float g0 = G0[r * ldf + j], g1 = G1[r * ldf + j], du = DU[r * ldf + j];
            S0[r * ldy + j] = y + 0.25f * f0 * h + 0.25f * f1 * h + g0 * du / h + 0.5f * g1 * du / h;
            S1[r * ldy + j] = y + f0 * h + SRK_B1_20 * g0 * rdt;
        });
        drift(S0, tph, F2);
        diffusion(S1, tp1, n, 3, G2);
        // stage 3: H1 = y + f2 h/4 + (2 g0 - g1 + g2/2) sqrt(h)  (its drift has weight 0)
        for_elems([&](int r, int j, int) {
            const float y = Y[r * ldy + j];
            S1[r * ldy + j] = y + 0.25f * F2[r * ldf + j] * h +
                              (SRK_B1_30 * G0[r * ldf + j] + SRK_B1_31 * G1[r * ldf + j] + SRK_B1_32 * G2[r * ldf + j]) * rdt;
        });
        const float* nb3 = diffusion_net(S1, tpq);
        // combination
        int kend = kout;
        while (kend < d.T - 1 && a.out_step[kend] == n) ++kend;
        for_elems([&](int r, int j, int row) {
            const float y = Y[r * ldy + j];
            const float ik = DW[r * ldf + j], ik0 = DU[r * ldf + j];
            const float ikk = 0.5f * (ik * ik - h);
            const float ikkk = (ik * ik * ik - 3.0f * h * ik) / 6.0f;
            const float g3 = g_elem(S1[r * ldy + j], tpq[0], n, 1, r, j, nb3);
            const float a1 = ik, a2 = ikk / rdt, a3 = ik0 / h, a4 = ikkk / h;
            const float w0 = srk_w0(a1, a2, a3, a4);
            const float w1 = srk_w1(a1, a2, a3, a4);
            const float w2 = srk_w2(a1, a2, a3, a4);
            const float w3 = a4;
            float ynew = y + (F0[r * ldf + j] + F1[r * ldf + j]) * (h / 6.0f) + F2[r * ldf + j] * (2.0f * h / 3.0f);
            ynew += w0 * G0[r * ldf + j] + w1 * G1[r * ldf + j] + w2 * G2[r * ldf + j] + w3 * g3;
            Y[r * ldy + j] = ynew;
            if (row < B) {
                if (a.traj) a.traj[(size_t)(n + 1) * BH + (size_t)row * H + j] = ynew;
                for (int k = kout; k < kend; ++k) {
                    const float c0 = a.out_w[2 * k], c1 = a.out_w[2 * k + 1];
                    const float o = (c0 == 0.0f) ? ynew : snsde_interp_out(c0, c1, y, ynew);
                    if (!a.row_out) a.ys[(size_t)(k + 1) * BH + (size_t)row * H + j] = o;
                    else if (a.row_out[row] == k + 1) a.ys[(size_t)row * H + j] = o;
                }
            }
        });
        kout = kend;
    }
}

// =====================================================================================================
// Generic adjoint (backward) kernel: discretise-then-optimise adjoint of the Euler / Milstein step for every
// input_option and every noise_option (Milstein: all but 7), any H / HH / C / NL within the LDS budget; the diffusion nets
// (14 / 15 / 18 / 19, dense dg/dy) add the net's forward, its transposed chain and - under Milstein - its tangent pass.
// Per step (last to first) a workgroup re-evaluates the drift chain on its GR rows from the saved state y_n
// (activations stay in LDS: one buffer per layer), applies the elementwise derivatives of f, g and the Milstein
// term to the adjoint, and walks the chain backwards with the ORIGINAL (out, in) weight layout
// (delta_in[k] = sum_n W[n][k] delta_out[n]: lanes over k read W rows coalesced).  Output: every adjoint a_n.
// =====================================================================================================
struct AdjArgs {
    GenericArgs g;          // params, packed forward weights (ws), coeffs, step table, dims
    const float* traj;      // (N+1, B, H)
    const float* dW_used;   // (N, B, H)
    const float* grad_ys;   // (T, B, H)
    float* adj;             // (N+1, B, H)
    int32_t nbuf;           // activation buffers: n_hid + 1
};

// out[r][k] (+)= sum_n W[n][k] * in[r][n]  for k in [k0, k0 + Kn): W row-major (N, ldk)
template <int RPT>
__device__ __forceinline__ void dense_T_rows(const float* __restrict__ wp, int ldk, int N, const float* in, int ldin, float* out,
                                             int ldout, int k, int r0) {
    constexpr int G = GR / RPT;
    float acc[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) acc[i] = 0.0f;
#pragma unroll 8
    for (int n = 0; n < N; ++n) {
        const float w = wp[(size_t)n * ldk];
#pragma unroll
        for (int i = 0; i < RPT; ++i) acc[i] = fmaf(in[(r0 + i * G) * ldin + n], w, acc[i]);
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) out[(r0 + i * G) * ldout + k] = acc[i];
}

template <int KP>
__device__ __forceinline__ void dense_T_group(const float* __restrict__ W, int ldk, int k0, int Kn, int N, const float* in,
                                              int ldin, float* out, int ldout) {
    constexpr int G0 = GW / KP, RG = G0 > GR ? GR : G0;
    const int k = threadIdx.x % KP, rg = threadIdx.x / KP;
    if (rg < RG && k < Kn) dense_T_rows<GR / RG>(W + k0 + k, ldk, N, in, ldin, out, ldout, k, rg);
}

__device__ __noinline__ void dense_T(const float* __restrict__ W, int ldk, int k0, int Kn, int N, const float* in, int ldin,
                                     float* out, int ldout) {
    if (Kn <= 64) dense_T_group<64>(W, ldk, k0, Kn, N, in, ldin, out, ldout);
    else if (Kn <= 128) dense_T_group<128>(W, ldk, k0, Kn, N, in, ldin, out, ldout);
    else if (Kn <= 256) dense_T_group<256>(W, ldk, k0, Kn, N, in, ldin, out, ldout);
    else {
        for (int k = threadIdx.x; k < Kn; k += GW) dense_T_rows<GR>(W + k0 + k, ldk, N, in, ldin, out, ldout, k, 0);
    }
}

// =====================================================================================================
// Milstein with a diffusion net on [tau, y] (noise_option 14/15/18/19).  torchsde's diagonal-noise Milstein takes
// g dg/dy (dW^2 - h) as a VJP of g with cotangent g (dW^2 - h): y1 = y + f h + g dW + 1/2 J_g(y)^T (g (dW^2 - h)).
// With a dense J_g that is one transposed pass through the net per step: dL/draw -> [q > 0] -> W2^T -> [h1 > 0] -> W1_y^T
// (+ the direct factor of the `raw = net(y) y` options).  Same tile structure as the SRK kernel (GW threads, dense_w / dense_T).
// =====================================================================================================
__global__ void __launch_bounds__(GW) snsde_generic_milnet_kernel(GenericArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const SnsdeDims& d = a.d;
    const SnsdeNet& net = a.net;
    const int H = d.H, C = d.C, B = d.B, io = d.io, no = d.no;
    const int ldy = a.ldy, ldw = a.ldw, ldx = a.ldx;
    float* ybuf = lds;
    float* bufA = ybuf + GR * ldy;
    float* bufB = bufA + GR * ldw;
    float* bufC = bufB + GR * ldw;      // diffusion net output / its cotangent
    float* bufD = bufC + GR * ldw;      // first net layer (no 18/19) / transposed-chain scratch
    float* bufE = bufD + GR * ldw;      // transposed-chain scratch
    float* xbuf = bufE + GR * ldw;
    const int lds_floats = GR * (ldy + 5 * ldw + ldx);
    const int tid = threadIdx.x, row0 = blockIdx.x * GR;
    for (int i = tid; i < lds_floats; i += GW) lds[i] = 0.0f;
    __syncthreads();
    for (int i = tid; i < GR * H; i += GW) {
        const int r = i / H, j = i - r * H, row = row0 + r;
        if (row < B) {
            const float v = a.y0[(size_t)row * H + j];
            ybuf[r * ldy + j] = v;
            a.ys[(size_t)row * H + j] = v;
            if (a.traj) a.traj[(size_t)row * H + j] = v;
        }
    }
    const float sig_theta = snsde_sigmoid(a.params[net.off_theta]);
    const bool uses_x = (io == 0 || io == 2 || io == 4 || io == 6);
    const bool uses_emb = (io == 2 || io == 4 || io == 6);
    const bool net2 = (no >= 18), net_y = (no == 15 || no == 19);
    const size_t BH = (size_t)B * H;
    int kout = 0;
    auto for_elems = [&](auto&& fn) {
        for (int i = tid; i < GR * H; i += GW) {
            const int r = i / H, j = i - r * H;
            fn(r, j, row0 + r);
        }
        __syncthreads();
    };
    for (int n = 0; n < d.N; ++n) {
        const float* st = a.step_tab + (size_t)n * SNSDE_STEP_STRIDE;
        const float h = st[1], frac = st[4], sqh = st[6];
        const int idx = __float_as_int(st[5]);
        if (tid < GR) { ybuf[tid * ldy + H] = st[2]; ybuf[tid * ldy + H + 1] = st[3]; }
        if (uses_x) {
            for (int i = tid; i < GR * C; i += GW) {
                const int r = i / C, c = i - r * C, row = row0 + r;
                float v = 0.0f;
                if (row < B) {
                    const float* cp = a.coeffs + ((size_t)row * (d.L - 1) + idx) * (4 * C) + c;
                    v = snsde_spline_eval(cp[0], cp[C], cp[2 * C], cp[3 * C], frac);
                }
                xbuf[r * ldx + c] = v;
            }
        }
        __syncthreads();
        // ---- drift (neuralsde.py:295-302) ----
        float* cur;
        float* oth;
        if (io == 0) { dense_w(a.params, a.ws, net.init, xbuf, ldx, bufA, ldw, true); cur = bufA; oth = bufB; }
        else if (!uses_emb) { dense_w(a.params, a.ws, net.in, ybuf, ldy, bufA, ldw, true); cur = bufA; oth = bufB; }
        else {
            dense_w(a.params, a.ws, net.in, ybuf, ldy, bufA, ldw, false);
            dense_w(a.params, a.ws, net.init, xbuf, ldx, bufA + H, ldw, false);
            __syncthreads();
            dense_w(a.params, a.ws, net.emb, bufA, ldw, bufB, ldw, true);
            cur = bufB; oth = bufA;
        }
        __syncthreads();
        for (int l = 0; l < net.n_hid; ++l) {
            dense_w(a.params, a.ws, net.hid[l], cur, ldw, oth, ldw, true);
            float* t = cur; cur = oth; oth = t;
            __syncthreads();
        }
        dense_w(a.params, a.ws, net.out, cur, ldw, oth, ldw, false);
        float* zbuf = oth;                   // pre-tanh drift, then the step without the transposed-chain term
        // ---- diffusion net (neuralsde.py:270-273, 278-281): first layer -> bufD (no 18/19) or straight to bufC ----
        dense_w(a.params, a.ws, net.ny0, ybuf, ldy, net2 ? bufD : bufC, ldw, net2);
        __syncthreads();
        if (net2) {
            dense_w(a.params, a.ws, net.ny1, bufD, ldw, bufC, ldw, true);
            __syncthreads();
        }
        // ---- f, g, increment; everything of the step that is elementwise; cotangent of the net output ----
        for_elems([&](int r, int j, int row) {
            const float y = ybuf[r * ldy + j];
            float z = zbuf[r * ldw + j];
            if (io == 5 || io == 6) z *= tanhf(y);
            const float f = tanhf(z);
            const float nb = bufC[r * ldw + j];
            const float raw = net_y ? nb * y : nb;
            const float g = tanhf(sig_theta * snsde_nan_to_num(raw));
            float dw = 0.0f;
            if (row < B) {
                if (a.dW) dw = a.dW[(size_t)n * BH + (size_t)row * H + j];
                else {
                    float z4[4];
                    snsde_philox_normal4(a.seed_dev ? *a.seed_dev : a.seed, (uint32_t)(a.row_offset + row), (uint32_t)(n >> 2), (uint32_t)j, z4);
                    dw = ((n & 3) == 0 ? z4[0] : (n & 3) == 1 ? z4[1] : (n & 3) == 2 ? z4[2] : z4[3]) * sqh;
                }
                if (a.dW_out) a.dW_out[(size_t)n * BH + (size_t)row * H + j] = dw;
            }
            const float dgr = snsde_finite(raw) ? (1.0f - g * g) * sig_theta : 0.0f;       // dg / d raw
            const float craw = g * fmaf(dw, dw, -h) * dgr;                                    // cotangent of raw
            float base = fmaf(g, dw, fmaf(f, h, y));
            if (net_y) base = fmaf(0.5f * craw, nb, base);                                    // direct factor: d(net y)/dy = net
            zbuf[r * ldw + j] = base;
            bufC[r * ldw + j] = (net2 && !(nb > 0.0f)) ? 0.0f : (net_y ? craw * y : craw);
        });
        // ---- J_net^T ----
        const float* dfirst = bufC;
        if (net2) {
            dense_T(a.params + net.ny1.src_w, net.ny1.K, 0, net.ny1.K, net.ny1.N, bufC, ldw, bufE, ldw);
            __syncthreads();
            for (int i = tid; i < GR * net.ny0.N; i += GW) {
                const int r = i / net.ny0.N, j = i - r * net.ny0.N;
                if (!(bufD[r * ldw + j] > 0.0f)) bufE[r * ldw + j] = 0.0f;
            }
            __syncthreads();
            dfirst = bufE;
        }
        float* dy = net2 ? bufC : bufE;
        dense_T(a.params + net.ny0.src_w, net.ny0.K, net.ny0.tshift, H, net.ny0.N, dfirst, ldw, dy, ldw);
        __syncthreads();
        int kend = kout;
        while (kend < d.T - 1 && a.out_step[kend] == n) ++kend;
        for_elems([&](int r, int j, int row) {
            const float y = ybuf[r * ldy + j];
            const float ynew = fmaf(0.5f, dy[r * ldw + j], zbuf[r * ldw + j]);
            ybuf[r * ldy + j] = ynew;
            if (row < B) {
                if (a.traj) a.traj[(size_t)(n + 1) * BH + (size_t)row * H + j] = ynew;
                for (int k = kout; k < kend; ++k) {
                    const float c0 = a.out_w[2 * k], c1 = a.out_w[2 * k + 1];
                    const float o = (c0 == 0.0f) ? ynew : snsde_interp_out(c0, c1, y, ynew);
                    if (!a.row_out) a.ys[(size_t)(k + 1) * BH + (size_t)row * H + j] = o;
                    else if (a.row_out[row] == k + 1) a.ys[(size_t)row * H + j] = o;
                }
            }
        });
        kout = kend;
    }
}

__global__ void __launch_bounds__(GW) snsde_generic_adjoint_kernel(AdjArgs aa) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const GenericArgs& a = aa.g;
    const SnsdeDims& d = a.d;
    const SnsdeNet& net = a.net;
    const int H = d.H, C = d.C, B = d.B, io = d.io, no = d.no;
    const int ldy = a.ldy, ldw = a.ldw, ldx = a.ldx;
    const int nact = net.n_hid + 1;                 // z0, hidden outputs (post-relu)
    float* ybuf = lds;                              // y_n | sin, cos
    float* xbuf = ybuf + GR * ldy;
    float* cat = xbuf + GR * ldx;                   // [yy | Xt] (emb) / scratch
    float* act = cat + GR * ldw;                    // act[l][GR][ldw], l < nact
    float* zo = act + (size_t)nact * GR * ldw;      // zout, later delta ping
    float* dl = zo + GR * ldw;                      // delta pong
    float* abuf = dl + GR * ldw;                    // adjoint a (GR x ldy)
    float* ayb = abuf + GR * ldy;                   // a_y accumulator
    const bool noise_net = (no == 14 || no == 15 || no == 18 || no == 19);
    const bool net2 = (no == 18 || no == 19), net_y = (no == 15 || no == 19);
    float* gnb = ayb + GR * ldy;                    // diffusion net (no 18/19): output layer, then its delta
    const bool milnet = noise_net && d.method == SNSDE_MILSTEIN;
    float* gp = gnb + (noise_net ? GR * ldw : 0);   // Milstein with a diffusion net: tangent J_net a of the adjoint
    float* zb = gp + GR * ldw;                      //                                a row of zeros (bias of the tangent passes)
    const int lds_floats = GR * (3 * ldy + ldx + (3 + nact + (noise_net ? 1 : 0) + (milnet ? 1 : 0)) * ldw) + (milnet ? ldw : 0);
    const int tid = threadIdx.x, row0 = blockIdx.x * GR;
    for (int i = tid; i < lds_floats; i += GW) lds[i] = 0.0f;
    __syncthreads();
    const float sig_theta = snsde_sigmoid(a.params[net.off_theta]);
    const float exp_sigma = (net.off_sigma >= 0) ? expf(a.params[net.off_sigma]) : 0.0f;
    const bool uses_x = (io == 0 || io == 2 || io == 4 || io == 6);
    const bool uses_emb = (io == 2 || io == 4 || io == 6);
    const bool geo = (io == 5 || io == 6);
    const float* gt = a.ws + (net.gt_tab >= 0 ? net.gt_tab : 0);
    const size_t BH = (size_t)B * H;
    const float mil = (d.method == SNSDE_MILSTEIN) ? 0.5f : 0.0f;
    // diffusion net on [tau, y] (neuralsde.py:270-273, 278-281): its first layer lands in `cat` (free once z0 exists), the
    // second in gnb; the transposed chain borrows `dl` (free until the drift's transposed chain starts)
    float* nraw = net2 ? gnb : cat;

    auto for_elems = [&](auto&& fn) {
        for (int i = tid; i < GR * H; i += GW) {
            const int r = i / H, j = i - r * H;
            fn(r, j, row0 + r);
        }
        __syncthreads();
    };

    for (int n = d.N - 1; n >= 0; --n) {
        const float* st = a.step_tab + (size_t)n * SNSDE_STEP_STRIDE;
        const float t0 = st[0], h = st[1], frac = st[4];
        const int idx = __float_as_int(st[5]);
        const int nout = __float_as_int(st[8]), kfirst = __float_as_int(st[9]);
        // adjoint of y_{n+1}: add the output gradients emitted after step n; keep the y_n share in ayb
        for_elems([&](int r, int j, int row) {
            float av = abuf[r * ldy + j], carry = 0.0f;
            if (row < B) {
                for (int k = kfirst; k < kfirst + nout; ++k) {
                    const float w0 = a.out_w[2 * k], w1 = a.out_w[2 * k + 1];
                    const float gk = a.row_out ? (a.row_out[row] == k + 1 ? aa.grad_ys[(size_t)row * H + j] : 0.0f)
                                               : aa.grad_ys[(size_t)(k + 1) * BH + (size_t)row * H + j];
                    if (w0 == 0.0f) av += gk; else { av = fmaf(w1, gk, av); carry = fmaf(w0, gk, carry); }
                }
                aa.adj[(size_t)(n + 1) * BH + (size_t)row * H + j] = av;
                ybuf[r * ldy + j] = aa.traj[(size_t)n * BH + (size_t)row * H + j];
            }
            abuf[r * ldy + j] = av;
            ayb[r * ldy + j] = carry;
        });
        if (tid < GR) { ybuf[tid * ldy + H] = st[2]; ybuf[tid * ldy + H + 1] = st[3]; }
        if (uses_x) {
            for (int i = tid; i < GR * C; i += GW) {
                const int r = i / C, c = i - r * C, row = row0 + r;
                float v = 0.0f;
                if (row < B) {
                    const float* cp = a.coeffs + ((size_t)row * (d.L - 1) + idx) * (4 * C) + c;
                    v = snsde_spline_eval(cp[0], cp[C], cp[2 * C], cp[3 * C], frac);
                }
                xbuf[r * ldx + c] = v;
            }
        }
        __syncthreads();
        // ---- forward re-evaluation, activations kept ----
        float* z0 = act;
        if (io == 0) dense_w(a.params, a.ws, net.init, xbuf, ldx, z0, ldw, true);
        else if (!uses_emb) dense_w(a.params, a.ws, net.in, ybuf, ldy, z0, ldw, true);
        else {
            dense_w(a.params, a.ws, net.in, ybuf, ldy, cat, ldw, false);
            dense_w(a.params, a.ws, net.init, xbuf, ldx, cat + H, ldw, false);
            __syncthreads();
            dense_w(a.params, a.ws, net.emb, cat, ldw, z0, ldw, true);
        }
        __syncthreads();
        for (int l = 0; l < net.n_hid; ++l) {
            dense_w(a.params, a.ws, net.hid[l], act + (size_t)l * GR * ldw, ldw, act + (size_t)(l + 1) * GR * ldw, ldw, true);
            __syncthreads();
        }
        dense_w(a.params, a.ws, net.out, act + (size_t)net.n_hid * GR * ldw, ldw, zo, ldw, false);
        __syncthreads();
        if (noise_net) {
            dense_w(a.params, a.ws, net.ny0, ybuf, ldy, cat, ldw, net2);
            __syncthreads();
            if (net2) {
                dense_w(a.params, a.ws, net.ny1, cat, ldw, gnb, ldw, true);
                __syncthreads();
            }
            if (milnet) {      // tangent of the net along the adjoint: J_net a (abuf's time columns are zero; no bias)
                dense_w(a.params, a.ws, net.ny0, abuf, ldy, net2 ? dl : gp, ldw, false, zb);
                __syncthreads();
                if (net2) {
                    for (int i = tid; i < GR * net.ny0.N; i += GW) {
                        const int r = i / net.ny0.N, j = i - r * net.ny0.N;
                        if (!(cat[r * ldw + j] > 0.0f)) dl[r * ldw + j] = 0.0f;
                    }
                    __syncthreads();
                    dense_w(a.params, a.ws, net.ny1, dl, ldw, gp, ldw, false, zb);
                    __syncthreads();
                }
            }
        }
        // ---- elementwise derivatives: delta_zout -> zo (in place), direct y terms -> ayb ----
        for_elems([&](int r, int j, int row) {
            const float y = ybuf[r * ldy + j], z = zo[r * ldw + j], av = abuf[r * ldy + j];
            const float dw = (row < B) ? aa.dW_used[(size_t)n * BH + (size_t)row * H + j] : 0.0f;
            float ty = 1.0f, zt = z;
            if (geo) { ty = tanhf(y); zt = z * ty; }
            const float f = tanhf(zt);
            const float dzt = av * h * (1.0f - f * f);
            float accy = ayb[r * ldy + j] + av;
            float dz = dzt;
            if (geo) { dz = dzt * ty; accy = fmaf(dzt * z, 1.0f - ty * ty, accy); }
            // diffusion: raw(y), raw', raw''
            float raw = 0.0f, r1 = 0.0f, r2 = 0.0f;
            switch (no) {
                case 0: break;
                case 1: raw = exp_sigma; break;
                case 2: raw = exp_sigma * t0; break;
                case 3: raw = exp_sigma * y; r1 = exp_sigma; break;
                case 4: raw = expf(a.params[net.off_sigma_diag + j]); break;
                case 5: raw = expf(a.params[net.off_sigma_diag + j]) * t0; break;
                case 6: r1 = expf(a.params[net.off_sigma_diag + j]); raw = r1 * y; break;
                case 7: raw = sqrtf(y); r1 = 0.5f / raw; r2 = -0.25f / (raw * y); break;
                case 8: raw = y * y * y; r1 = 3.0f * y * y; r2 = 6.0f * y; break;
                case 9: raw = snsde_sigmoid(y); r1 = raw * (1.0f - raw); r2 = r1 * (1.0f - 2.0f * raw); break;
                case 10: raw = fmaxf(y, 0.0f); r1 = y > 0.0f ? 1.0f : 0.0f; break;
                case 11: raw = t0 * y; r1 = t0; break;
                case 12: case 16: raw = gt[(size_t)n * H + j]; break;
                case 13: case 17: r1 = gt[(size_t)n * H + j]; raw = r1 * y; break;
                case 14: case 18: raw = nraw[r * ldw + j]; break;
                case 15: case 19: r1 = nraw[r * ldw + j]; raw = r1 * y; break;      // r1: the direct y factor only
                default: break;
            }
            const bool fin = snsde_finite(raw);
            const float g = tanhf(sig_theta * snsde_nan_to_num(raw));
            float dnet = 0.0f;
            if (fin) {
                const float sech = 1.0f - g * g;
                const float g1 = sech * sig_theta * r1;
                accy = fmaf(av * dw, g1, accy);
                if (mil != 0.0f && !noise_net) {
                    const float g2 = sech * sig_theta * r2 - 2.0f * g * g1 * sig_theta * r1;
                    accy = fmaf(av * mil * (g1 * g1 + g * g2), dw * dw - h, accy);
                }
                dnet = av * dw * sech * sig_theta * (net_y ? y : 1.0f);     // cotangent of the net's output
                if (milnet) {
                    // a . d/dy [1/2 J_raw^T u],  u = g dg/draw (dW^2 - h):  with p = J_net a,  w = (J_raw a) (dW^2 - h) d(g dg/draw)/draw
                    //   raw = net:     J_net^T w                                   raw = net y:  p u + w net + J_net^T (a u + w y)
                    const float nbv = nraw[r * ldw + j];
                    const float pj = (net2 && !(nbv > 0.0f)) ? 0.0f : gp[r * ldw + j];
                    const float v = fmaf(dw, dw, -h), s1 = sech * sig_theta;
                    const float u = g * s1 * v, sp = sig_theta * s1 * (1.0f - 3.0f * g * g);
                    if (net_y) {
                        const float w = fmaf(y, pj, nbv * av) * v * sp;
                        accy = fmaf(mil, fmaf(pj, u, w * nbv), accy);
                        dnet = fmaf(mil, fmaf(av, u, w * y), dnet);
                    } else {
                        dnet = fmaf(mil, pj * v * sp, dnet);
                    }
                }
            }
            if (noise_net) nraw[r * ldw + j] = (net2 && !(nraw[r * ldw + j] > 0.0f)) ? 0.0f : dnet;
            zo[r * ldw + j] = dz;
            ayb[r * ldy + j] = accy;
        });
        if (noise_net) {     // J_net^T: (second layer, relu mask of the hidden layer,) y columns of the first layer
            const float* dfirst = cat;
            if (net2) {
                dense_T(a.params + net.ny1.src_w, net.ny1.K, 0, net.ny1.K, net.ny1.N, gnb, ldw, dl, ldw);
                __syncthreads();
                for (int i = tid; i < GR * net.ny0.N; i += GW) {
                    const int r = i / net.ny0.N, j = i - r * net.ny0.N;
                    if (!(cat[r * ldw + j] > 0.0f)) dl[r * ldw + j] = 0.0f;
                }
                __syncthreads();
                dfirst = dl;
            }
            float* dy = net2 ? gnb : dl;
            dense_T(a.params + net.ny0.src_w, net.ny0.K, net.ny0.tshift, H, net.ny0.N, dfirst, ldw, dy, ldw);
            __syncthreads();
            for_elems([&](int r, int j, int) { ayb[r * ldy + j] += dy[r * ldw + j]; });
        }
        // ---- transposed chain ----
        float* cur = zo;
        float* oth = dl;
        dense_T(a.params + net.out.src_w, net.out.K, 0, net.out.K, net.out.N, cur, ldw, oth, ldw);
        __syncthreads();
        for (int l = net.n_hid; l >= 0; --l) {
            // oth holds dL/d(post-relu z_l): mask with z_l > 0, then move one layer down
            const float* zl = act + (size_t)l * GR * ldw;
            const int width = (l == 0 && io != 0) ? net.in.N : (l == 0 ? H : net.hid[l - 1].N);
            for (int i = tid; i < GR * width; i += GW) {
                const int r = i / width, j = i - r * width;
                if (!(zl[r * ldw + j] > 0.0f)) oth[r * ldw + j] = 0.0f;
            }
            __syncthreads();
            float* t = cur; cur = oth; oth = t;     // cur = dL/d(pre-activation z_l)
            if (l > 0) {
                dense_T(a.params + net.hid[l - 1].src_w, net.hid[l - 1].K, 0, net.hid[l - 1].K, net.hid[l - 1].N, cur, ldw, oth, ldw);
                __syncthreads();
            }
        }
        // first stage: cur = dL/d(pre-activation of z0)
        if (io != 0) {
            const float* din = cur;
            if (uses_emb) {   // through emb to the yy half of the concatenation
                dense_T(a.params + net.emb.src_w, net.emb.K, 0, H, net.emb.N, cur, ldw, oth, ldw);
                __syncthreads();
                din = oth;
            }
            float* dst = (din == oth) ? cur : oth;
            dense_T(a.params + net.in.src_w, net.in.K, net.in.tshift, H, net.in.N, din, ldw, dst, ldw);
            __syncthreads();
            for_elems([&](int r, int j, int) { abuf[r * ldy + j] = ayb[r * ldy + j] + dst[r * ldw + j]; });
        } else {
            for_elems([&](int r, int j, int) { abuf[r * ldy + j] = ayb[r * ldy + j]; });
        }
    }
    for (int i = tid; i < GR * H; i += GW) {      // ys[0] = y0
        const int r = i / H, j = i - r * H, row = row0 + r;
        if (row < B) aa.adj[(size_t)row * H + j] = abuf[r * ldy + j] +
            ((!a.row_out || a.row_out[row] == 0) ? aa.grad_ys[(size_t)row * H + j] : 0.0f);
    }
}


// =====================================================================================================
// SRK adjoint: discretise-then-optimise backward of the SRID2 step above (every noise_option, any dims in the LDS budget).
// Per step (last to first) the workgroup re-evaluates the three drift stages from the saved state y_n and the saved
// increments (I_k, I_k0), then walks the stage graph backwards:
//     Fbar_s, Gbar_s  <-  a (alpha_s h, w_s)  +  the H0/H1 combinations of later stages
//     H1bar_s = Gbar_s * dg/dy(t1_s, H1_s)            (elementwise)
//     H0bar_s = J_f(t0_s, H0_s)^T Fbar_s              (transposed dense chain, as in the Euler adjoint)
// The activations of the LAST forward drift evaluation (stage 2) are reused by its VJP; stages 1 and 0 are
// re-evaluated right before theirs: 5 drift forwards + 3 VJPs per step.
// =====================================================================================================
struct SrkAdjArgs {
    GenericArgs g;
    const float* srk_tab;
    const float* traj;      // (N+1, B, H)
    const float* dW_used;   // (N, B, H)  I_k
    const float* dU_used;   // (N, B, H)  I_k0
    const float* grad_ys;
    float* adj;             // (N+1, B, H)
    int32_t ldf;
};

__global__ void __launch_bounds__(GW) snsde_generic_srk_adjoint_kernel(SrkAdjArgs aa) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const GenericArgs& a = aa.g;
    const SnsdeDims& d = a.d;
    const SnsdeNet& net = a.net;
    const int H = d.H, C = d.C, B = d.B, io = d.io, no = d.no;
    const int ldy = a.ldy, ldw = a.ldw, ldx = a.ldx, ldf = aa.ldf;
    const int nact = net.n_hid + 1;
    float* Y = lds;                                 // y_n
    float* S0 = Y + GR * ldy;                       // drift stage state | sin, cos
    float* xbuf = S0 + GR * ldy;
    float* cat = xbuf + GR * ldx;
    float* act = cat + GR * ldw;                    // act[l][GR][ldw]
    float* zo = act + (size_t)nact * GR * ldw;
    float* dl = zo + GR * ldw;
    float* V = dl + GR * ldw;                       // 16 planes [GR][ldf]
    const int plane = GR * ldf;
    float* F0 = V; float* F1 = V + plane; float* F2 = V + 2 * plane;
    float* G0 = V + 3 * plane; float* G1 = V + 4 * plane; float* G2 = V + 5 * plane;
    float* FB0 = V + 6 * plane; float* FB1 = V + 7 * plane; float* FB2 = V + 8 * plane;
    float* GB0 = V + 9 * plane; float* GB1 = V + 10 * plane; float* GB2 = V + 11 * plane;
    float* YB = V + 12 * plane; float* AV = V + 13 * plane; float* DW = V + 14 * plane; float* DU = V + 15 * plane;
    const bool noise_net = (no == 14 || no == 15 || no == 18 || no == 19);
    const bool net2 = (no == 18 || no == 19), net_y = (no == 15 || no == 19);
    float* S1 = V + 16 * plane;                     // diffusion nets: diffusion stage state | sin, cos
    float* gnb = S1 + GR * ldy;                     //                 output layer of the two-layer net / its delta
    const int lds_floats = GR * (2 * ldy + ldx + (3 + nact) * ldw + 16 * ldf + (noise_net ? ldy + ldw : 0));
    const int tid = threadIdx.x, row0 = blockIdx.x * GR;
    for (int i = tid; i < lds_floats; i += GW) lds[i] = 0.0f;
    __syncthreads();
    const float sig_theta = snsde_sigmoid(a.params[net.off_theta]);
    const float exp_sigma = (net.off_sigma >= 0) ? expf(a.params[net.off_sigma]) : 0.0f;
    const bool uses_x = (io == 0 || io == 2 || io == 4 || io == 6);
    const bool uses_emb = (io == 2 || io == 4 || io == 6);
    const bool geo = (io == 5 || io == 6);
    const float* gt = a.ws + (net.gt_tab >= 0 ? net.gt_tab : 0);
    const size_t BH = (size_t)B * H;

    auto for_elems = [&](auto&& fn) {
        for (int i = tid; i < GR * H; i += GW) {
            const int r = i / H, j = i - r * H;
            fn(r, j, row0 + r);
        }
        __syncthreads();
    };
    // drift forward at (stage time tp, state in S0), activations kept in act[] / zo
    auto fwd_chain = [&](const float* tp) {
        const float frac = tp[3];
        const int idx = __float_as_int(tp[4]);
        if (tid < GR) { S0[tid * ldy + H] = tp[1]; S0[tid * ldy + H + 1] = tp[2]; }
        if (uses_x) {
            for (int i = tid; i < GR * C; i += GW) {
                const int r = i / C, c = i - r * C, row = row0 + r;
                float v = 0.0f;
                if (row < B) {
                    const float* cp = a.coeffs + ((size_t)row * (d.L - 1) + idx) * (4 * C) + c;
                    v = snsde_spline_eval(cp[0], cp[C], cp[2 * C], cp[3 * C], frac);
                }
                xbuf[r * ldx + c] = v;
            }
        }
        __syncthreads();
        float* z0 = act;
        if (io == 0) dense_w(a.params, a.ws, net.init, xbuf, ldx, z0, ldw, true);
        else if (!uses_emb) dense_w(a.params, a.ws, net.in, S0, ldy, z0, ldw, true);
        else {
            dense_w(a.params, a.ws, net.in, S0, ldy, cat, ldw, false);
            dense_w(a.params, a.ws, net.init, xbuf, ldx, cat + H, ldw, false);
            __syncthreads();
            dense_w(a.params, a.ws, net.emb, cat, ldw, z0, ldw, true);
        }
        __syncthreads();
        for (int l = 0; l < net.n_hid; ++l) {
            dense_w(a.params, a.ws, net.hid[l], act + (size_t)l * GR * ldw, ldw, act + (size_t)(l + 1) * GR * ldw, ldw, true);
            __syncthreads();
        }
        dense_w(a.params, a.ws, net.out, act + (size_t)net.n_hid * GR * ldw, ldw, zo, ldw, false);
        __syncthreads();
    };
    auto f_value = [&](float* fout) {
        for_elems([&](int r, int j, int) {
            float z = zo[r * ldw + j];
            if (geo) z *= tanhf(S0[r * ldy + j]);
            fout[r * ldf + j] = tanhf(z);
        });
    };
    // g(t, y) and dg/dy for the elementwise diffusions
    auto g_and_prime = [&](float y, float t, int n, int slot, int j, float& g1, float nbv = 0.0f, float* gs = nullptr) {
        float raw = 0.0f, r1 = 0.0f;
        switch (no) {
            case 14: case 18: raw = nbv; break;                  // nbv: the diffusion net's output for this element
            case 15: case 19: r1 = nbv; raw = nbv * y; break;    // (g1: the direct y factor only)
            case 0: break;
            case 1: raw = exp_sigma; break;
            case 2: raw = exp_sigma * t; break;
            case 3: raw = exp_sigma * y; r1 = exp_sigma; break;
            case 4: raw = expf(a.params[net.off_sigma_diag + j]); break;
            case 5: raw = expf(a.params[net.off_sigma_diag + j]) * t; break;
            case 6: r1 = expf(a.params[net.off_sigma_diag + j]); raw = r1 * y; break;
            case 7: raw = sqrtf(y); r1 = 0.5f / raw; break;
            case 8: raw = y * y * y; r1 = 3.0f * y * y; break;
            case 9: raw = snsde_sigmoid(y); r1 = raw * (1.0f - raw); break;
            case 10: raw = fmaxf(y, 0.0f); r1 = y > 0.0f ? 1.0f : 0.0f; break;
            case 11: raw = t * y; r1 = t; break;
            case 12: case 16: raw = gt[((size_t)n * 4 + slot) * H + j]; break;
            case 13: case 17: r1 = gt[((size_t)n * 4 + slot) * H + j]; raw = r1 * y; break;
            default: break;
        }
        const float g = tanhf(sig_theta * snsde_nan_to_num(raw));
        const float dgr = snsde_finite(raw) ? (1.0f - g * g) * sig_theta : 0.0f;     // dg / d raw
        g1 = dgr * r1;
        if (gs) *gs = dgr;
        return g;
    };
    // diffusion net on [tau, state] at the stage time of tp, state in S1 (neuralsde.py:270-273, 278-281): raw output buffer.
    // First layer in `cat`, second in gnb; both and `dl` are free between a drift forward and its VJP.
    auto gnet_fwd = [&](const float* tp) -> float* {
        if (tid < GR) { S1[tid * ldy + H] = tp[1]; S1[tid * ldy + H + 1] = tp[2]; }
        __syncthreads();
        dense_w(a.params, a.ws, net.ny0, S1, ldy, cat, ldw, net2);
        __syncthreads();
        if (!net2) return cat;
        dense_w(a.params, a.ws, net.ny1, cat, ldw, gnb, ldw, true);
        __syncthreads();
        return gnb;
    };
    // J_net^T applied to the output cotangent the caller left in the raw-output buffer: d/d state (ldw stride)
    auto gnet_vjp = [&]() -> const float* {
        const float* dfirst = cat;
        if (net2) {
            dense_T(a.params + net.ny1.src_w, net.ny1.K, 0, net.ny1.K, net.ny1.N, gnb, ldw, dl, ldw);
            __syncthreads();
            for (int i = tid; i < GR * net.ny0.N; i += GW) {
                const int r = i / net.ny0.N, j = i - r * net.ny0.N;
                if (!(cat[r * ldw + j] > 0.0f)) dl[r * ldw + j] = 0.0f;
            }
            __syncthreads();
            dfirst = dl;
        }
        float* dy = net2 ? gnb : dl;
        dense_T(a.params + net.ny0.src_w, net.ny0.K, net.ny0.tshift, H, net.ny0.N, dfirst, ldw, dy, ldw);
        __syncthreads();
        return dy;
    };
    // g at (stage time tp, state(r, j)) -> plane G
    auto gnet_value = [&](auto&& state, const float* tp, float* G) {
        for_elems([&](int r, int j, int) { S1[r * ldy + j] = state(r, j); });
        const float* nb = gnet_fwd(tp);
        for_elems([&](int r, int j, int) {
            float gp;
            G[r * ldf + j] = g_and_prime(S1[r * ldy + j], tp[0], 0, 0, j, gp, nb[r * ldw + j]);
        });
    };
    // hb = J_g(tp, state)^T cot, handed to `consume` in two additive parts (direct y factor, then the net's chain)
    auto gnet_stage_vjp = [&](auto&& state, const float* tp, auto&& cot, auto&& consume) {
        for_elems([&](int r, int j, int) { S1[r * ldy + j] = state(r, j); });
        float* nb = gnet_fwd(tp);
        for_elems([&](int r, int j, int) {
            const float y = S1[r * ldy + j], nbv = nb[r * ldw + j];
            float gp, gs;
            g_and_prime(y, tp[0], 0, 0, j, gp, nbv, &gs);
            const float v = cot(r, j);
            if (net_y) consume(r, j, v * gp);
            nb[r * ldw + j] = (net2 && !(nbv > 0.0f)) ? 0.0f : v * gs * (net_y ? y : 1.0f);
        });
        const float* dy = gnet_vjp();
        for_elems([&](int r, int j, int) { consume(r, j, dy[r * ldw + j]); });
    };
    // J_f(stage)^T FB for the stage whose activations are in act[] / zo and whose state is in S0:
    // result added to YB; returns the buffer holding dL/dH0 (ldw stride) for the caller's combinations
    auto vjp = [&](const float* FB) -> const float* {
        for_elems([&](int r, int j, int) {
            const float z = zo[r * ldw + j], v = FB[r * ldf + j];
            float ty = 1.0f, zt = z;
            if (geo) { ty = tanhf(S0[r * ldy + j]); zt = z * ty; }
            const float f = tanhf(zt);
            const float dzt = v * (1.0f - f * f);
            zo[r * ldw + j] = geo ? dzt * ty : dzt;
            cat[r * ldw + j] = geo ? dzt * z * (1.0f - ty * ty) : 0.0f;     // direct y term of the tanh(y) gate
        });
        float* cur = zo;
        float* oth = dl;
        dense_T(a.params + net.out.src_w, net.out.K, 0, net.out.K, net.out.N, cur, ldw, oth, ldw);
        __syncthreads();
        for (int l = net.n_hid; l >= 0; --l) {
            const float* zl = act + (size_t)l * GR * ldw;
            const int width = (l == 0 && io != 0) ? net.in.N : (l == 0 ? H : net.hid[l - 1].N);
            for (int i = tid; i < GR * width; i += GW) {
                const int r = i / width, j = i - r * width;
                if (!(zl[r * ldw + j] > 0.0f)) oth[r * ldw + j] = 0.0f;
            }
            __syncthreads();
            float* t = cur; cur = oth; oth = t;
            if (l > 0) {
                dense_T(a.params + net.hid[l - 1].src_w, net.hid[l - 1].K, 0, net.hid[l - 1].K, net.hid[l - 1].N, cur, ldw, oth, ldw);
                __syncthreads();
            }
        }
        float* dst = oth;
        if (io != 0) {
            const float* din = cur;
            if (uses_emb) {
                dense_T(a.params + net.emb.src_w, net.emb.K, 0, H, net.emb.N, cur, ldw, oth, ldw);
                __syncthreads();
                din = oth;
            }
            dst = (din == oth) ? cur : oth;
            dense_T(a.params + net.in.src_w, net.in.K, net.in.tshift, H, net.in.N, din, ldw, dst, ldw);
            __syncthreads();
            for_elems([&](int r, int j, int) { dst[r * ldw + j] += cat[r * ldw + j]; });
        } else {
            for_elems([&](int r, int j, int) { dst[r * ldw + j] = cat[r * ldw + j]; });
        }
        return dst;
    };

    for (int n = d.N - 1; n >= 0; --n) {
        const float* st = a.step_tab + (size_t)n * SNSDE_STEP_STRIDE;
        const float h = st[1], rdt = st[6];
        const int nout = __float_as_int(st[8]), kfirst = __float_as_int(st[9]);
        const float* tp0 = aa.srk_tab + (size_t)n * 4 * SNSDE_SRK_STRIDE;
        const float* tpq = tp0 + SNSDE_SRK_STRIDE;
        const float* tph = tp0 + 2 * SNSDE_SRK_STRIDE;
        const float* tp1 = tp0 + 3 * SNSDE_SRK_STRIDE;
        // adjoint of y_{n+1} (+ output gradients), saved state and increments of the step
        for_elems([&](int r, int j, int row) {
            float av = AV[r * ldf + j], carry = 0.0f, y = 0.0f, dw = 0.0f, du = 0.0f;
            if (row < B) {
                for (int k = kfirst; k < kfirst + nout; ++k) {
                    const float w0 = a.out_w[2 * k], w1 = a.out_w[2 * k + 1];
                    const float gk = a.row_out ? (a.row_out[row] == k + 1 ? aa.grad_ys[(size_t)row * H + j] : 0.0f)
                                               : aa.grad_ys[(size_t)(k + 1) * BH + (size_t)row * H + j];
                    if (w0 == 0.0f) av += gk; else { av = fmaf(w1, gk, av); carry = fmaf(w0, gk, carry); }
                }
                const size_t off = (size_t)n * BH + (size_t)row * H + j;
                aa.adj[off + BH] = av;
                y = aa.traj[off]; dw = aa.dW_used[off]; du = aa.dU_used[off];
            }
            AV[r * ldf + j] = av;
            YB[r * ldf + j] = carry + av;
            Y[r * ldy + j] = y; S0[r * ldy + j] = y;
            DW[r * ldf + j] = dw; DU[r * ldf + j] = du;
        });
        if (noise_net) {
            // ---- forward stages (every diffusion evaluation is a net pass on its own stage state) ----
            fwd_chain(tp0);
            f_value(F0);
            gnet_value([&](int r, int j) { return Y[r * ldy + j]; }, tp0, G0);
            for_elems([&](int r, int j, int) { S0[r * ldy + j] = Y[r * ldy + j] + F0[r * ldf + j] * h; });     // H0_1
            fwd_chain(tp1);
            f_value(F1);
            auto h11 = [&](int r, int j) { return Y[r * ldy + j] + 0.25f * F0[r * ldf + j] * h + SRK_B1_10 * G0[r * ldf + j] * rdt; };
            auto h12 = [&](int r, int j) { return Y[r * ldy + j] + F0[r * ldf + j] * h + SRK_B1_20 * G0[r * ldf + j] * rdt; };
            gnet_value(h11, tpq, G1);
            for_elems([&](int r, int j, int) {
                const float du = DU[r * ldf + j];
                S0[r * ldy + j] = Y[r * ldy + j] + 0.25f * F0[r * ldf + j] * h + 0.25f * F1[r * ldf + j] * h +
                                  G0[r * ldf + j] * du / h + 0.5f * G1[r * ldf + j] * du / h;                 // H0_2
            });
            fwd_chain(tph);        // activations of this evaluation feed the first drift VJP below
            f_value(F2);
            gnet_value(h12, tp1, G2);
            // ---- backward: cotangents of the combination ----
            for_elems([&](int r, int j, int) {
                const float av = AV[r * ldf + j], ik = DW[r * ldf + j], ik0 = DU[r * ldf + j];
                const float ikk = 0.5f * (ik * ik - h);
                const float ikkk = (ik * ik * ik - 3.0f * h * ik) / 6.0f;
                const float a1 = ik, a2 = ikk / rdt, a3 = ik0 / h, a4 = ikkk / h;
                FB0[r * ldf + j] = av * (h / 6.0f); FB1[r * ldf + j] = av * (h / 6.0f); FB2[r * ldf + j] = av * (2.0f * h / 3.0f);
                GB0[r * ldf + j] = (srk_w0(a1, a2, a3, a4)) * av;
                GB1[r * ldf + j] = (srk_w1(a1, a2, a3, a4)) * av;
                GB2[r * ldf + j] = (srk_w2(a1, a2, a3, a4)) * av;
            });
            // stage 3: g(t0 + h/4, H1_3), H1_3 = y + f2 h/4 + (2 g0 - g1 + g2/2) sqrt(h), weight I_kkk / h
            gnet_stage_vjp([&](int r, int j) {
                               return Y[r * ldy + j] + 0.25f * F2[r * ldf + j] * h +
                                      (SRK_B1_30 * G0[r * ldf + j] + SRK_B1_31 * G1[r * ldf + j] + SRK_B1_32 * G2[r * ldf + j]) * rdt; },
                           tpq,
                           [&](int r, int j) {
                               const float ik = DW[r * ldf + j];
                               return AV[r * ldf + j] * ((ik * ik * ik - 3.0f * h * ik) / 6.0f) / h; },
                           [&](int r, int j, float hb) {
                               YB[r * ldf + j] += hb; FB2[r * ldf + j] = fmaf(0.25f * h, hb, FB2[r * ldf + j]);
                               GB0[r * ldf + j] = fmaf(SRK_B1_30 * rdt, hb, GB0[r * ldf + j]);
                               GB1[r * ldf + j] = fmaf(SRK_B1_31 * rdt, hb, GB1[r * ldf + j]);
                               GB2[r * ldf + j] = fmaf(SRK_B1_32 * rdt, hb, GB2[r * ldf + j]); });
            // stage 2, diffusion half: H1_2 = y + f0 h + g0 sqrt(h)
            gnet_stage_vjp(h12, tp1, [&](int r, int j) { return GB2[r * ldf + j]; },
                           [&](int r, int j, float hb) {
                               YB[r * ldf + j] += hb; FB0[r * ldf + j] = fmaf(h, hb, FB0[r * ldf + j]);
                               GB0[r * ldf + j] = fmaf(SRK_B1_20 * rdt, hb, GB0[r * ldf + j]); });
            // stage 2, drift half: H0_2 = y + (f0 + f1) h/4 + (g0 + g1/2) I_k0/h
            {
                const float* dH = vjp(FB2);
                for_elems([&](int r, int j, int) {
                    const float dv = dH[r * ldw + j], du = DU[r * ldf + j];
                    YB[r * ldf + j] += dv;
                    FB0[r * ldf + j] = fmaf(0.25f * h, dv, FB0[r * ldf + j]);
                    FB1[r * ldf + j] = fmaf(0.25f * h, dv, FB1[r * ldf + j]);
                    GB0[r * ldf + j] = fmaf(du / h, dv, GB0[r * ldf + j]);
                    GB1[r * ldf + j] = fmaf(0.5f * du / h, dv, GB1[r * ldf + j]);
                });
            }
            // stage 1: H1_1 = y + f0 h/4 - g0 sqrt(h)/2 (diffusion at t0 + h/4), H0_1 = y + f0 h (drift at t0 + h)
            gnet_stage_vjp(h11, tpq, [&](int r, int j) { return GB1[r * ldf + j]; },
                           [&](int r, int j, float hb) {
                               YB[r * ldf + j] += hb; FB0[r * ldf + j] = fmaf(0.25f * h, hb, FB0[r * ldf + j]);
                               GB0[r * ldf + j] = fmaf(SRK_B1_10 * rdt, hb, GB0[r * ldf + j]); });
            for_elems([&](int r, int j, int) { S0[r * ldy + j] = Y[r * ldy + j] + F0[r * ldf + j] * h; });
            fwd_chain(tp1);
            {
                const float* dH = vjp(FB1);
                for_elems([&](int r, int j, int) {
                    const float dv = dH[r * ldw + j];
                    YB[r * ldf + j] += dv;
                    FB0[r * ldf + j] = fmaf(h, dv, FB0[r * ldf + j]);
                });
            }
            // stage 0: both evaluated at (t0, y)
            gnet_stage_vjp([&](int r, int j) { return Y[r * ldy + j]; }, tp0, [&](int r, int j) { return GB0[r * ldf + j]; },
                           [&](int r, int j, float hb) { YB[r * ldf + j] += hb; });
            for_elems([&](int r, int j, int) { S0[r * ldy + j] = Y[r * ldy + j]; });
            fwd_chain(tp0);
            {
                const float* dH = vjp(FB0);
                for_elems([&](int r, int j, int) { AV[r * ldf + j] = YB[r * ldf + j] + dH[r * ldw + j]; });
            }
            continue;
        }
        // ---- forward stages ----
        fwd_chain(tp0);
        f_value(F0);
        for_elems([&](int r, int j, int) {
            const float y = Y[r * ldy + j];
            float g1;
            const float g0 = g_and_prime(y, tp0[0], n, 0, j, g1);
            G0[r * ldf + j] = g0;
            S0[r * ldy + j] = y + F0[r * ldf + j] * h;                                   // H0_1
        });
        fwd_chain(tp1);
        f_value(F1);
        for_elems([&](int r, int j, int) {
            const float y = Y[r * ldy + j], f0 = F0[r * ldf + j], f1 = F1[r * ldf + j], g0 = G0[r * ldf + j];
            const float du = DU[r * ldf + j];
            float g1p;
            const float g1 = g_and_prime(y + 0.25f * f0 * h + SRK_B1_10 * g0 * rdt, tpq[0], n, 1, j, g1p);   // at H1_1
            G1[r * ldf + j] = g1;
            S0[r * ldy + j] = y + 0.25f * f0 * h + 0.25f * f1 * h + g0 * du / h + 0.5f * g1 * du / h;  // H0_2
        });
        fwd_chain(tph);        // activations of this evaluation feed the first VJP below
        f_value(F2);
        // ---- backward: stage 3 and the diffusion half of stage 2 (elementwise) ----
        for_elems([&](int r, int j, int) {
            const float y = Y[r * ldy + j], av = AV[r * ldf + j];
            const float f0 = F0[r * ldf + j], f2 = F2[r * ldf + j], g0 = G0[r * ldf + j], g1 = G1[r * ldf + j];
            const float ik = DW[r * ldf + j], ik0 = DU[r * ldf + j];
            float gp;
            const float h12 = y + f0 * h + SRK_B1_20 * g0 * rdt;
            const float g2 = g_and_prime(h12, tp1[0], n, 3, j, gp);
            const float g2p = gp;
            const float ikk = 0.5f * (ik * ik - h);
            const float ikkk = (ik * ik * ik - 3.0f * h * ik) / 6.0f;
            const float a1 = ik, a2 = ikk / rdt, a3 = ik0 / h, a4 = ikkk / h;
            const float w0 = srk_w0(a1, a2, a3, a4);
            const float w1 = srk_w1(a1, a2, a3, a4);
            const float w2 = srk_w2(a1, a2, a3, a4);
            const float w3 = a4;
            float fb0 = av * (h / 6.0f), fb1 = fb0, fb2 = av * (2.0f * h / 3.0f);
            float gb0 = w0 * av, gb1 = w1 * av, gb2 = w2 * av, yb = YB[r * ldf + j];
            // stage 3: H1_3 = y + f2 h/4 + (2 g0 - g1 + g2/2) sqrt(h), evaluated at t0 + h/4
            const float h13 = y + 0.25f * f2 * h + (SRK_B1_30 * g0 + SRK_B1_31 * g1 + SRK_B1_32 * g2) * rdt;
            g_and_prime(h13, tpq[0], n, 1, j, gp);
            float hb = w3 * av * gp;
            yb += hb; fb2 = fmaf(0.25f * h, hb, fb2);
            gb0 = fmaf(SRK_B1_30 * rdt, hb, gb0); gb1 = fmaf(SRK_B1_31 * rdt, hb, gb1); gb2 = fmaf(SRK_B1_32 * rdt, hb, gb2);
            // stage 2, diffusion half: H1_2 = y + f0 h + g0 sqrt(h)
            hb = gb2 * g2p;
            yb += hb; fb0 = fmaf(h, hb, fb0); gb0 = fmaf(SRK_B1_20 * rdt, hb, gb0);
            FB0[r * ldf + j] = fb0; FB1[r * ldf + j] = fb1; FB2[r * ldf + j] = fb2;
            GB0[r * ldf + j] = gb0; GB1[r * ldf + j] = gb1; GB2[r * ldf + j] = gb2;
            G2[r * ldf + j] = g2;
            YB[r * ldf + j] = yb;
        });
        // stage 2, drift half: H0_2 = y + (f0 + f1) h/4 + (g0 + g1/2) I_k0/h
        {
            const float* dH = vjp(FB2);
            for_elems([&](int r, int j, int) {
                const float dv = dH[r * ldw + j], du = DU[r * ldf + j];
                YB[r * ldf + j] += dv;
                FB0[r * ldf + j] = fmaf(0.25f * h, dv, FB0[r * ldf + j]);
                FB1[r * ldf + j] = fmaf(0.25f * h, dv, FB1[r * ldf + j]);
                GB0[r * ldf + j] = fmaf(du / h, dv, GB0[r * ldf + j]);
                GB1[r * ldf + j] = fmaf(0.5f * du / h, dv, GB1[r * ldf + j]);
            });
        }
        // stage 1: H1_1 = y + f0 h/4 - g0 sqrt(h)/2 (diffusion at t0 + h/4), H0_1 = y + f0 h (drift at t0 + h)
        for_elems([&](int r, int j, int) {
            const float y = Y[r * ldy + j], f0 = F0[r * ldf + j], g0 = G0[r * ldf + j];
            float gp;
            g_and_prime(y + 0.25f * f0 * h + SRK_B1_10 * g0 * rdt, tpq[0], n, 1, j, gp);
            const float hb = GB1[r * ldf + j] * gp;
            YB[r * ldf + j] += hb;
            FB0[r * ldf + j] = fmaf(0.25f * h, hb, FB0[r * ldf + j]);
            GB0[r * ldf + j] = fmaf(SRK_B1_10 * rdt, hb, GB0[r * ldf + j]);
            S0[r * ldy + j] = y + f0 * h;
        });
        fwd_chain(tp1);
        {
            const float* dH = vjp(FB1);
            for_elems([&](int r, int j, int) {
                const float dv = dH[r * ldw + j];
                YB[r * ldf + j] += dv;
                FB0[r * ldf + j] = fmaf(h, dv, FB0[r * ldf + j]);
            });
        }
        // stage 0: both evaluated at (t0, y)
        for_elems([&](int r, int j, int) {
            const float y = Y[r * ldy + j];
            float gp;
            g_and_prime(y, tp0[0], n, 0, j, gp);
            YB[r * ldf + j] = fmaf(GB0[r * ldf + j], gp, YB[r * ldf + j]);
            S0[r * ldy + j] = y;
        });
        fwd_chain(tp0);
        {
            const float* dH = vjp(FB0);
            for_elems([&](int r, int j, int) { AV[r * ldf + j] = YB[r * ldf + j] + dH[r * ldw + j]; });
        }
    }
    for (int i = tid; i < GR * H; i += GW) {      // ys[0] = y0
        const int r = i / H, j = i - r * H, row = row0 + r;
        if (row < B) aa.adj[(size_t)row * H + j] = AV[r * ldf + j] +
            ((!a.row_out || a.row_out[row] == 0) ? aa.grad_ys[(size_t)row * H + j] : 0.0f);
    }
}

__global__ void snsde_spline_kernel(const float* __restrict__ coeffs, int B, int L, int C, int index, float frac,
                                    int derivative, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C;
    const float* cp = coeffs + ((size_t)b * (L - 1) + index) * (4 * C) + c;
    out[i] = derivative ? snsde_spline_deriv(cp[C], cp[2 * C], cp[3 * C], frac)
                        : snsde_spline_eval(cp[0], cp[C], cp[2 * C], cp[3 * C], frac);
}

inline int round4(int x) { return (x + 3) & ~3; }

}  // namespace

int snsde_time_table_srk_launch(const float* params, const float* srk_tab, float* gt, const SnsdeNet& net, int H, int no,
                                int n_rows, hipStream_t stream) {
    // time-only diffusion at the stage times of every step: gt[(n*4 + slot)][H], rows of the SRK stage table
    hipLaunchKernelGGL(snsde_time_table_kernel, dim3(n_rows), dim3(128), H * sizeof(float), stream, params, srk_tab,
                       gt, net.nt0, net.nt1, H, no, SNSDE_SRK_STRIDE, 1, net.off_sigma, net.off_sigma_diag);
    return hipGetLastError() == hipSuccess ? SNSDE_OK : SNSDE_ERR_LAUNCH;
}

int snsde_generic_workspace_floats(const snsde_solve* s, const SnsdeNet& net, size_t* floats) {
    size_t f = (size_t)net.packed_floats;
    if (net.gt_tab >= 0) f = (size_t)net.gt_tab + (size_t)s->n_steps * s->model.hidden_channels * (s->method == SNSDE_SRK ? 4 : 1);
    *floats = f;
    return SNSDE_OK;
}

int snsde_generic_launch(const snsde_solve* s, const SnsdeNet& net, hipStream_t stream, int eval_mode,
                         const float* eval_y, float* eval_f, float* eval_g, const float* step_row_dev) {
    const snsde_model& m = s->model;
    float* ws = static_cast<float*>(s->workspace);
    // 1. pack weights (W^T, K padded, time features rotated last)
    PackJob job;
    job.n = 0;
    auto add = [&](const SnsdeLayer& L) { if (L.present && L.w >= 0) job.layer[job.n++] = L; };
    add(net.init); add(net.in); add(net.emb);
    for (int i = 0; i < net.n_hid; ++i) add(net.hid[i]);
    add(net.out); add(net.ny0); add(net.ny1);
    const bool prepare = !(s->flags & SNSDE_FLAG_REUSE_PREPARED) || eval_mode;
    if (job.n > 0 && prepare) {
        hipLaunchKernelGGL(snsde_pack_kernel, dim3(32, job.n), dim3(256), 0, stream, s->params, ws, job);
    }
    // 2. time-only diffusion table
    const int no = m.noise_option;
    const float* step_tab = eval_mode ? step_row_dev : s->step_tab;
    const int n_steps = eval_mode ? 1 : s->n_steps;
    if (net.gt_tab >= 0 && prepare) {
        hipLaunchKernelGGL(snsde_time_table_kernel, dim3(n_steps), dim3(128), m.hidden_channels * sizeof(float), stream,
                           s->params, step_tab, ws + net.gt_tab, net.nt0, net.nt1, m.hidden_channels, no,
                           SNSDE_STEP_STRIDE, 2);
    }
    // 3. the fused solve
    GenericArgs a;
    a.d = SnsdeDims{s->batch, m.hidden_channels, m.hidden_hidden_channels, m.input_channels, s->knots,
                    m.num_hidden_layers, m.input_option, m.noise_option, n_steps, s->n_out, s->method};
    a.net = net;
    a.params = s->params;
    a.ws = ws;
    a.coeffs = s->coeffs;
    a.step_tab = step_tab;
    a.out_step = s->out_step;
    a.out_w = s->out_w;
    a.y0 = eval_mode ? eval_y : s->y0;
    a.dW = s->dW;
    a.ys = s->ys;
    a.traj = s->traj;
    a.dW_out = s->dW_out;
    a.row_out = eval_mode ? nullptr : s->row_out;
    a.row_offset = s->row_offset;
    a.seed = s->seed; a.seed_dev = s->seed_dev;
    a.eval_mode = eval_mode;
    a.eval_f = eval_f;
    a.eval_g = eval_g;
    const int H = m.hidden_channels, HH = m.hidden_hidden_channels;
    a.ldy = round4(H + 2);
    int wmax = 2 * H > HH ? 2 * H : HH;
    a.ldw = round4(wmax) + 4;
    a.ldx = round4(m.input_channels);
    if (!eval_mode && s->method == SNSDE_MILSTEIN && (no == 14 || no == 15 || no == 18 || no == 19)) {
        // Milstein through a diffusion net: its own kernel (one transposed pass through the net per step)
        const size_t bytes = (size_t)GR * (a.ldy + 5 * a.ldw + a.ldx) * sizeof(float);
        if (bytes > 160 * 1024) return SNSDE_ERR_LDS;
        if (bytes > 64 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(snsde_generic_milnet_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
            return SNSDE_ERR_LDS;
        hipLaunchKernelGGL(snsde_generic_milnet_kernel, dim3((s->batch + GR - 1) / GR), dim3(GW), bytes, stream, a);
        return hipGetLastError() == hipSuccess ? SNSDE_OK : SNSDE_ERR_LAUNCH;
    }
    const size_t lds_bytes = (size_t)GR * (a.ldy + 3 * a.ldw + a.ldx) * sizeof(float);
    if (lds_bytes > 160 * 1024) return SNSDE_ERR_LDS;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(snsde_generic_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return SNSDE_ERR_LDS;
    }
    const int grid = (s->batch + GR - 1) / GR;
    hipLaunchKernelGGL(snsde_generic_kernel, dim3(grid), dim3(GT), lds_bytes, stream, a);
    return hipGetLastError() == hipSuccess ? SNSDE_OK : SNSDE_ERR_LAUNCH;
}

bool snsde_generic_backward_supported(const snsde_solve* s) {
    const int no = s->model.noise_option;
    const bool noise_net = (no == 14 || no == 15 || no == 18 || no == 19);   // dense diffusion Jacobian: one more buffer
    const int H = s->model.hidden_channels, HH = s->model.hidden_hidden_channels;
    const int wmax = 2 * H > HH ? 2 * H : HH;
    const size_t ldy = round4(H + 2), ldw = round4(wmax) + 4, ldx = round4(s->model.input_channels);
    if (s->method == SNSDE_SRK) {     // SRK adjoint: bounded by its LDS planes
        const size_t fl = (size_t)GR * (2 * ldy + ldx + (3 + s->model.num_hidden_layers) * ldw + 16 * round4(H) +
                                        (noise_net ? ldy + ldw : 0));
        return fl * sizeof(float) <= 160 * 1024;
    }
    if (s->method != SNSDE_EULER && s->method != SNSDE_MILSTEIN) return false;
    if (s->method == SNSDE_MILSTEIN && no == 7) return false;
    const bool milnet = noise_net && s->method == SNSDE_MILSTEIN;
    return ((size_t)GR * (3 * ldy + ldx + (3 + s->model.num_hidden_layers + (noise_net ? 1 : 0) + (milnet ? 1 : 0)) * ldw) +
            (milnet ? ldw : 0)) * sizeof(float) <= 160 * 1024;
}

// The adjoint kernels need the generic packed weights and the time-only diffusion table: prepared here in the
// BACKWARD workspace, so the forward may have run on any kernel family (e.g. the MFMA SRK variant).
int snsde_generic_backward_launch(const snsde_backward* b, const SnsdeNet& net, hipStream_t stream) {
    const snsde_solve* s = &b->fwd;
    const snsde_model& m = s->model;
    size_t need = 0;
    snsde_generic_workspace_floats(s, net, &need);
    if (!b->workspace || b->workspace_bytes < need * sizeof(float)) return SNSDE_ERR_WORKSPACE;
    float* bws = static_cast<float*>(b->workspace);
    {
        PackJob job;
        job.n = 0;
        auto add = [&](const SnsdeLayer& L) { if (L.present && L.w >= 0) job.layer[job.n++] = L; };
        add(net.init); add(net.in); add(net.emb);
        for (int i = 0; i < net.n_hid; ++i) add(net.hid[i]);
        add(net.out); add(net.ny0); add(net.ny1);
        if (job.n > 0) hipLaunchKernelGGL(snsde_pack_kernel, dim3(32, job.n), dim3(256), 0, stream, s->params, bws, job);
        if (net.gt_tab >= 0) {
            if (s->method == SNSDE_SRK) {
                if (!s->srk_tab) return SNSDE_ERR_NULL;
                hipLaunchKernelGGL(snsde_time_table_kernel, dim3(s->n_steps * 4), dim3(128), m.hidden_channels * sizeof(float),
                                   stream, s->params, s->srk_tab, bws + net.gt_tab, net.nt0, net.nt1, m.hidden_channels,
                                   m.noise_option, SNSDE_SRK_STRIDE, 1);
            } else {
                hipLaunchKernelGGL(snsde_time_table_kernel, dim3(s->n_steps), dim3(128), m.hidden_channels * sizeof(float),
                                   stream, s->params, s->step_tab, bws + net.gt_tab, net.nt0, net.nt1, m.hidden_channels,
                                   m.noise_option, SNSDE_STEP_STRIDE, 2);
            }
        }
    }
    AdjArgs aa;
    GenericArgs& a = aa.g;
    a.d = SnsdeDims{s->batch, m.hidden_channels, m.hidden_hidden_channels, m.input_channels, s->knots,
                    m.num_hidden_layers, m.input_option, m.noise_option, s->n_steps, s->n_out, s->method};
    a.net = net;
    a.params = s->params; a.ws = bws; a.coeffs = s->coeffs;
    a.step_tab = s->step_tab; a.out_step = s->out_step; a.out_w = s->out_w; a.y0 = s->y0; a.dW = nullptr;
    a.ys = nullptr; a.traj = nullptr; a.dW_out = nullptr; a.row_out = s->row_out; a.row_offset = 0; a.seed = 0; a.seed_dev = nullptr; a.eval_mode = 0;
    a.eval_f = nullptr; a.eval_g = nullptr;
    const int H = m.hidden_channels, HH = m.hidden_hidden_channels;
    a.ldy = round4(H + 2);
    const int wmax = 2 * H > HH ? 2 * H : HH;
    a.ldw = round4(wmax) + 4;
    a.ldx = round4(m.input_channels);
    if (s->method == SNSDE_SRK) {
        if (!s->dU_out || !s->srk_tab) return SNSDE_ERR_NULL;
        SrkAdjArgs sa;
        sa.g = a;
        sa.srk_tab = s->srk_tab; sa.traj = s->traj; sa.dW_used = s->dW_out; sa.dU_used = s->dU_out;
        sa.grad_ys = b->grad_ys; sa.adj = b->adj; sa.ldf = round4(H);
        const bool nn = (m.noise_option == 14 || m.noise_option == 15 || m.noise_option == 18 || m.noise_option == 19);
        const size_t bytes = (size_t)GR * (2 * a.ldy + a.ldx + (3 + net.n_hid + 1) * a.ldw + 16 * sa.ldf + (nn ? a.ldy + a.ldw : 0)) * sizeof(float);
        if (bytes > 160 * 1024) return SNSDE_ERR_LDS;
        if (bytes > 64 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(snsde_generic_srk_adjoint_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
            return SNSDE_ERR_LDS;
        hipLaunchKernelGGL(snsde_generic_srk_adjoint_kernel, dim3((s->batch + GR - 1) / GR), dim3(GW), bytes, stream, sa);
        return hipGetLastError() == hipSuccess ? SNSDE_OK : SNSDE_ERR_LAUNCH;
    }
    aa.traj = s->traj; aa.dW_used = s->dW_out; aa.grad_ys = b->grad_ys; aa.adj = b->adj; aa.nbuf = net.n_hid + 1;
    const bool nn = (m.noise_option == 14 || m.noise_option == 15 || m.noise_option == 18 || m.noise_option == 19);
    const bool mn = nn && s->method == SNSDE_MILSTEIN;
    const size_t lds_bytes = ((size_t)GR * (3 * a.ldy + a.ldx + (3 + net.n_hid + 1 + (nn ? 1 : 0) + (mn ? 1 : 0)) * a.ldw) +
                              (mn ? a.ldw : 0)) * sizeof(float);
    if (lds_bytes > 160 * 1024) return SNSDE_ERR_LDS;
    if (lds_bytes > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(snsde_generic_adjoint_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
        return SNSDE_ERR_LDS;
    hipLaunchKernelGGL(snsde_generic_adjoint_kernel, dim3((s->batch + GR - 1) / GR), dim3(GW), lds_bytes, stream, aa);
    return hipGetLastError() == hipSuccess ? SNSDE_OK : SNSDE_ERR_LAUNCH;
}

int snsde_srk_launch(const snsde_solve* s, const SnsdeNet& net, hipStream_t stream) {
    const snsde_model& m = s->model;
    float* ws = static_cast<float*>(s->workspace);
    if (!(s->flags & SNSDE_FLAG_REUSE_PREPARED)) {
        PackJob job;
        job.n = 0;
        auto add = [&](const SnsdeLayer& L) { if (L.present && L.w >= 0) job.layer[job.n++] = L; };
        add(net.init); add(net.in); add(net.emb);
        for (int i = 0; i < net.n_hid; ++i) add(net.hid[i]);
        add(net.out); add(net.ny0); add(net.ny1);
        if (job.n > 0) hipLaunchKernelGGL(snsde_pack_kernel, dim3(32, job.n), dim3(256), 0, stream, s->params, ws, job);
        if (net.gt_tab >= 0)   // time-only diffusion at the 4 stage times of every step: gt[(n*4 + slot)][H]
            hipLaunchKernelGGL(snsde_time_table_kernel, dim3(s->n_steps * 4), dim3(128), m.hidden_channels * sizeof(float),
                               stream, s->params, s->srk_tab, ws + net.gt_tab, net.nt0, net.nt1, m.hidden_channels,
                               m.noise_option, SNSDE_SRK_STRIDE, 1);
    }
    SrkArgs sa;
    GenericArgs& a = sa.g;
    a.d = SnsdeDims{s->batch, m.hidden_channels, m.hidden_hidden_channels, m.input_channels, s->knots,
                    m.num_hidden_layers, m.input_option, m.noise_option, s->n_steps, s->n_out, s->method};
    a.net = net;
    a.params = s->params; a.ws = ws; a.coeffs = s->coeffs; a.step_tab = s->step_tab; a.out_step = s->out_step;
    a.out_w = s->out_w; a.y0 = s->y0; a.dW = s->dW; a.ys = s->ys; a.traj = s->traj; a.dW_out = s->dW_out;
    a.row_out = s->row_out; a.row_offset = s->row_offset; a.seed = s->seed; a.seed_dev = s->seed_dev; a.eval_mode = 0; a.eval_f = nullptr; a.eval_g = nullptr;
    const int H = m.hidden_channels, HH = m.hidden_hidden_channels;
    a.ldy = round4(H + 2);
    const int wmax = 2 * H > HH ? 2 * H : HH;
    a.ldw = round4(wmax) + 4;
    a.ldx = round4(m.input_channels);
    sa.srk_tab = s->srk_tab; sa.dU = s->dU; sa.dU_out = s->dU_out; sa.ldf = round4(H);
    const size_t lds_bytes = (size_t)GR * (3 * a.ldy + 3 * a.ldw + a.ldx + 8 * sa.ldf) * sizeof(float);
    if (lds_bytes > 160 * 1024) return SNSDE_ERR_LDS;
    if (lds_bytes > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(snsde_generic_srk_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
        return SNSDE_ERR_LDS;
    hipLaunchKernelGGL(snsde_generic_srk_kernel, dim3((s->batch + GR - 1) / GR), dim3(GW), lds_bytes, stream, sa);
    return hipGetLastError() == hipSuccess ? SNSDE_OK : SNSDE_ERR_LAUNCH;
}

int snsde_spline_launch(const float* coeffs, int32_t B, int32_t L, int32_t C, int32_t index, float frac,
                        int32_t derivative, float* out, hipStream_t stream) {
    const int total = B * C;
    hipLaunchKernelGGL(snsde_spline_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, coeffs, B, L, C, index,
                       frac, derivative, out);
    return hipGetLastError() == hipSuccess ? SNSDE_OK : SNSDE_ERR_LAUNCH;
}
