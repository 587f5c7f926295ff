// Spline coefficient construction on the GPU: the reference builds these offline with
// Python loops over batch x channel x time (controldiffeq/interpolate.py:9-155, minutes on real datasets).
// One thread owns one scalar series (batch row b, channel c) and runs the sequential parts (Thomas solve over the
// OBSERVED knots, re-expansion onto every original sub-interval) with the reference's operation order and no FMA
// contraction; per-series temporaries live in a [L][series] workspace so neighbouring threads stay coalesced.
#include "snsde_internal.h"

namespace {

struct SplineArgs {
    const float* times;   // (L)
    const float* X;       // (B, L, C), NaN = missing
    float* out;           // (B, L-1, 4C) = cat[a, b, two_c, three_d]
    int32_t* oidx;        // workspace [L][S]
    float* nd;            // workspace [L][S]
    float* nb;            // workspace [L][S]
    float* kd;            // workspace [L][S]
    int32_t B, L, C;
};

__global__ void snsde_natural_spline_kernel(SplineArgs a) {
#pragma clang fp contract(off)
    const int S = a.B * a.C, L = a.L, C = a.C;
    const int sidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (sidx >= S) return;
    const int b = sidx / C, c = sidx - b * C;
    const float* x = a.X + (size_t)b * L * C + c;          // x[j * C]
    float* out = a.out + (size_t)b * (L - 1) * 4 * C + c;  // out[j * 4C + k * C]
    // ---- observed knots; the ends are imputed with the first / last observation (interpolate.py:100-114)
    int first = -1, last = -1;
    for (int j = 0; j < L; ++j) {
        const float v = x[(size_t)j * C];
        if (v == v) { if (first < 0) first = j; last = j; }
    }
    if (first < 0) {   // every entry missing: zero coefficients (interpolate.py:84-92)
        for (int j = 0; j < L - 1; ++j)
            for (int k = 0; k < 4; ++k) out[(size_t)j * 4 * C + k * C] = 0.0f;
        return;
    }
    const float xfirst = x[(size_t)first * C], xlast = x[(size_t)last * C];
    int m = 0;
    for (int j = 0; j < L; ++j) {
        const float v = x[(size_t)j * C];
        if (v == v || j == 0 || j == L - 1) { a.oidx[(size_t)m * S + sidx] = j; ++m; }
    }
    auto tc = [&](int i) { return a.times[a.oidx[(size_t)i * S + sidx]]; };
    auto xc = [&](int i) {
        const int j = a.oidx[(size_t)i * S + sidx];
        const float v = x[(size_t)j * C];
        return (v == v) ? v : (j == 0 ? xfirst : xlast);
    };
    // ---- knot derivatives of the natural spline through the m observed knots (interpolate.py:9-55, misc.py:12-66)
    if (m > 2) {
        float rec_prev = 0.0f, sc_prev = 0.0f, ndp = 0.0f, nbp = 0.0f;
        for (int i = 0; i < m; ++i) {
            float rec = 0.0f, sc = 0.0f;
            if (i < m - 1) {
                rec = 1.0f / (tc(i + 1) - tc(i));
                sc = (3.0f * (xc(i + 1) - xc(i))) * (rec * rec);
            }
            float diag, rhs;
            if (i == 0) { diag = rec; rhs = sc; }
            else if (i == m - 1) { diag = 0.0f + rec_prev; rhs = 0.0f + sc_prev; }
            else { diag = rec + rec_prev; rhs = sc + sc_prev; }
            diag = diag * 2.0f;
            float ndv, nbv;
            if (i == 0) { ndv = diag; nbv = rhs; }
            else {
                const float w = rec_prev / ndp;
                ndv = diag - w * rec_prev;
                nbv = rhs - w * nbp;
            }
            a.nd[(size_t)i * S + sidx] = ndv;
            a.nb[(size_t)i * S + sidx] = nbv;
            ndp = ndv; nbp = nbv; rec_prev = rec; sc_prev = sc;
        }
        float kn = nbp / ndp;
        a.kd[(size_t)(m - 1) * S + sidx] = kn;
        for (int i = m - 2; i >= 0; --i) {
            const float rec = 1.0f / (tc(i + 1) - tc(i));
            kn = (a.nb[(size_t)i * S + sidx] - rec * kn) / a.nd[(size_t)i * S + sidx];
            a.kd[(size_t)i * S + sidx] = kn;
        }
    }
    // ---- coefficients on every original interval (interpolate.py:116-150)
    int p = -1;
    float A = 0.f, Bc = 0.f, C2 = 0.f, D3 = 0.f, tprev = 0.f;
    for (int j = 0; j < L - 1; ++j) {
        const float tj = a.times[j];
        if (p + 1 < m - 1 && tj >= tc(p + 1)) {
            ++p;
            tprev = tc(p);
            const float x0 = xc(p), x1 = xc(p + 1);
            if (m == 2) {
                A = x0; Bc = (x1 - x0) / (tc(1) - tc(0)); C2 = 0.0f; D3 = 0.0f;
            } else {
                const float rec = 1.0f / (tc(p + 1) - tprev);
                const float six = 2.0f * (3.0f * (x1 - x0));
                const float k0 = a.kd[(size_t)p * S + sidx], k1 = a.kd[(size_t)(p + 1) * S + sidx];
                A = x0; Bc = k0;
                C2 = (six * rec - 4.0f * k0 - 2.0f * k1) * rec;
                D3 = (-six * rec + 3.0f * (k0 + k1)) * (rec * rec);
            }
        }
        const float off = tprev - tj;
        const float a_inner = (0.5f * C2 - D3 * off / 3.0f) * off;
        out[(size_t)j * 4 * C] = A + (a_inner - Bc) * off;
        out[(size_t)j * 4 * C + C] = Bc + (D3 * off - C2) * off;
        out[(size_t)j * 4 * C + 2 * C] = C2 - 2.0f * D3 * off;
        out[(size_t)j * 4 * C + 3 * C] = D3;
    }
}

// torchcde hermite_cubic_coefficients_with_backward_differences: missing values filled linearly in
// time between observed neighbours (ends: nearest observation; all-missing: 0), then per interval
// a = x_k, b = previous secant slope (own slope on the first interval), two_c = 4 (m_k - b)/h, three_d = -3 (m_k - b)/h^2.
__global__ void snsde_hermite_kernel(const float* __restrict__ times, const float* __restrict__ X, float* __restrict__ outp,
                                     int B, int L, int C) {
#pragma clang fp contract(off)
    const int S = B * C;
    const int sidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (sidx >= S) return;
    const int b = sidx / C, c = sidx - b * C;
    const float* x = X + (size_t)b * L * C + c;
    float* out = outp + (size_t)b * (L - 1) * 4 * C + c;
    int first = -1;
    for (int j = 0; j < L && first < 0; ++j) { const float v = x[(size_t)j * C]; if (v == v) first = j; }
    // filled value at position j given the previous observed (pj, pv) and the next observed found by scanning
    int pj = -1; float pv = 0.0f;
    int nj = first; float nv = first >= 0 ? x[(size_t)first * C] : 0.0f;
    auto filled = [&](int j) {
        if (first < 0) return 0.0f;
        const float v = x[(size_t)j * C];
        if (v == v) { pj = j; pv = v; return v; }
        if (nj >= 0 && nj < j) { nj = -1; }
        if (nj < 0 || nj <= j) {      // find the next observation after j
            nj = -1;
            for (int q = j + 1; q < L; ++q) { const float w = x[(size_t)q * C]; if (w == w) { nj = q; nv = w; break; } }
            if (nj < 0) nj = L;       // none
        }
        if (pj < 0) return nv;        // leading gap
        if (nj >= L) return pv;       // trailing gap
        const float wgt = (times[j] - times[pj]) / (times[nj] - times[pj]);
        return pv + wgt * (nv - pv);
    };
    float x0 = filled(0), mprev = 0.0f;
    for (int j = 0; j < L - 1; ++j) {
        const float x1 = filled(j + 1);
        const float h = times[j + 1] - times[j];
        const float mk = (x1 - x0) / h;
        const float bb = (j == 0) ? mk : mprev;
        out[(size_t)j * 4 * C] = x0;
        out[(size_t)j * 4 * C + C] = bb;
        out[(size_t)j * 4 * C + 2 * C] = 4.0f * (mk - bb) / h;
        out[(size_t)j * 4 * C + 3 * C] = -3.0f * (mk - bb) / (h * h);
        mprev = mk; x0 = x1;
    }
}

}  // namespace

extern "C" {

size_t snsde_spline_workspace_bytes(int32_t batch, int32_t knots, int32_t channels) {
    if (batch <= 0 || knots < 2 || channels <= 0) return 0;
    return (size_t)4 * knots * batch * channels * sizeof(float) + 256;
}

int snsde_natural_cubic_coeffs(const float* times, const float* X, int32_t batch, int32_t knots, int32_t channels,
                               float* coeffs, void* workspace, size_t workspace_bytes, void* hip_stream) {
    if (!times || !X || !coeffs || !workspace) return SNSDE_ERR_NULL;
    if (batch <= 0 || knots < 2 || channels <= 0) return SNSDE_ERR_DIMS;
    if (workspace_bytes < snsde_spline_workspace_bytes(batch, knots, channels)) return SNSDE_ERR_WORKSPACE;
    const size_t n = (size_t)knots * batch * channels;
    SplineArgs a;
    a.times = times; a.X = X; a.out = coeffs;
    a.oidx = static_cast<int32_t*>(workspace);
    a.nd = reinterpret_cast<float*>(a.oidx + n);
    a.nb = a.nd + n;
    a.kd = a.nb + n;
    a.B = batch; a.L = knots; a.C = channels;
    const int S = batch * channels;
    hipLaunchKernelGGL(snsde_natural_spline_kernel, dim3((S + 127) / 128), dim3(128), 0, static_cast<hipStream_t>(hip_stream), a);
    return hipGetLastError() == hipSuccess ? SNSDE_OK : SNSDE_ERR_LAUNCH;
}

int snsde_hermite_coeffs(const float* times, const float* X, int32_t batch, int32_t knots, int32_t channels,
                         float* coeffs, void* hip_stream) {
    if (!times || !X || !coeffs) return SNSDE_ERR_NULL;
    if (batch <= 0 || knots < 2 || channels <= 0) return SNSDE_ERR_DIMS;
    const int S = batch * channels;
    hipLaunchKernelGGL(snsde_hermite_kernel, dim3((S + 127) / 128), dim3(128), 0, static_cast<hipStream_t>(hip_stream),
                       times, X, coeffs, batch, knots, channels);
    return hipGetLastError() == hipSuccess ? SNSDE_OK : SNSDE_ERR_LAUNCH;
}

}  // extern "C"

// ---- initial state from the control path (stand-alone form; the MFMA forward folds it into its prepare launch) ----
namespace {
__global__ void snsde_z0_kernel(SnsdeZ0Job z) { snsde_z0_rows(z, blockIdx.x, gridDim.x); }
}

int snsde_z0_launch(const snsde_solve* s, hipStream_t stream) {
    if (!s->z0_weight || !s->z0_bias || !s->y0 || !s->step_tab) return SNSDE_ERR_NULL;
    SnsdeZ0Job z{s->z0_weight, s->z0_bias, s->coeffs, s->step_tab, const_cast<float*>(s->y0), s->batch,
                 s->model.hidden_channels, s->model.input_channels, s->knots};
    const int total = s->batch * s->model.hidden_channels;
    int grid = (total + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(snsde_z0_kernel, dim3(grid), dim3(256), 0, stream, z);
    return hipGetLastError() == hipSuccess ? SNSDE_OK : SNSDE_ERR_LAUNCH;
}

