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
#include "snsde_internal.h"

namespace {

constexpr int GR = 8;    // batch rows per workgroup
constexpr int GT = 256;  // threads per workgroup (4 waves)

struct PackJob {
    SnsdeLayer layer[SNSDE_MAX_HIDDEN + 6];
    int32_t n;
};

__global__ void snsde_pack_kernel(const float* __restrict__ params, float* __restrict__ ws, PackJob job) {
    const SnsdeLayer L = job.layer[blockIdx.y];
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
                                        float* __restrict__ gt, SnsdeLayer nt0, SnsdeLayer nt1, int H, int no) {
    extern __shared__ float hbuf[];
    const int n = blockIdx.x;
    const float sn = step_tab[n * SNSDE_STEP_STRIDE + 2], cs = step_tab[n * SNSDE_STEP_STRIDE + 3];
    const bool two = (no == 16 || no == 17);
    for (int j = threadIdx.x; j < H; j += blockDim.x) {
        const float v = fmaf(cs, params[nt0.src_w + 2 * j + 1], sn * params[nt0.src_w + 2 * j]) + params[nt0.src_b + j];
        if (two) hbuf[j] = fmaxf(v, 0.0f);
        else gt[n * H + j] = v;
    }
    if (!two) return;
    __syncthreads();
    for (int j = threadIdx.x; j < H; j += blockDim.x) {
        float acc = 0.0f;
        const float* w = params + nt1.src_w + (size_t)j * H;
        for (int k = 0; k < H; ++k) acc = fmaf(hbuf[k], w[k], acc);
        gt[n * H + j] = fmaxf(acc + params[nt1.src_b + j], 0.0f);
    }
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
    int64_t row_offset;
    uint64_t seed;
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
                    snsde_philox_normal4(a.seed, (uint32_t)(a.row_offset + row), (uint32_t)(n >> 2), (uint32_t)(4 * q + e), z4);
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
                    const float fin = (raw - raw == 0.0f) ? 1.0f : 0.0f;  // finite raw
                    const float dg = (1.0f - g * g) * sig_theta * draw * fin;
                    ynew = fmaf(0.5f * (g * dg), fmaf(dw, dw, -h), ynew);
                }
                ybuf[r * ldy + j] = ynew;
                if (a.traj) a.traj[(size_t)(n + 1) * BH + (size_t)row * H + j] = ynew;
                if (a.dW_out) a.dW_out[(size_t)n * BH + (size_t)row * H + j] = dw;
                int k = kout;
                while (k < d.T - 1 && a.out_step[k] == n) {
                    const float w0 = a.out_w[2 * k], w1 = a.out_w[2 * k + 1];
                    a.ys[(size_t)(k + 1) * BH + (size_t)row * H + j] = (w0 == 0.0f) ? ynew : w0 * y + w1 * ynew;
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

int snsde_time_table_launch(const float* params, const float* step_tab, float* gt, const SnsdeLayer& nt0,
                            const SnsdeLayer& nt1, int H, int no, int n_steps, hipStream_t stream) {
    hipLaunchKernelGGL(snsde_time_table_kernel, dim3(n_steps), dim3(128), H * sizeof(float), stream, params, step_tab,
                       gt, nt0, nt1, H, no);
    return hipGetLastError() == hipSuccess ? SNSDE_OK : SNSDE_ERR_LAUNCH;
}

int snsde_generic_workspace_floats(const snsde_solve* s, const SnsdeNet& net, size_t* floats) {
    size_t f = (size_t)net.packed_floats;
    if (net.gt_tab >= 0) f = (size_t)net.gt_tab + (size_t)s->n_steps * s->model.hidden_channels;
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
                           s->params, step_tab, ws + net.gt_tab, net.nt0, net.nt1, m.hidden_channels, no);
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
    a.row_offset = s->row_offset;
    a.seed = s->seed;
    a.eval_mode = eval_mode;
    a.eval_f = eval_f;
    a.eval_g = eval_g;
    const int H = m.hidden_channels, HH = m.hidden_hidden_channels;
    a.ldy = round4(H + 2);
    int wmax = 2 * H > HH ? 2 * H : HH;
    a.ldw = round4(wmax) + 4;
    a.ldx = round4(m.input_channels);
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

int snsde_spline_launch(const float* coeffs, int32_t B, int32_t L, int32_t C, int32_t index, float frac,
                        int32_t derivative, float* out, hipStream_t stream) {
    const int total = B * C;
    hipLaunchKernelGGL(snsde_spline_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, coeffs, B, L, C, index,
                       frac, derivative, out);
    return hipGetLastError() == hipSuccess ? SNSDE_OK : SNSDE_ERR_LAUNCH;
}
