// Composed parameter blocks of the tutorial-style fields (include/snsde.h: snsde_affine_compose / _backward): adjacent affine maps
// multiplied out on the device, and the block's gradient carried back to the field's own tensors - one launch each.
// The matrices are at most 256 x 514 (a field's layers at the instantiated hidden sizes): a workgroup per output row, threads over the
// output columns, plain fp32 FMA loops over K (the operands sit in L2 after the first row; the whole call is a few microseconds - the
// point is the launch count of the torch formulation).
// Reference: tutorial/simple OU process - Neural LSDE / LNSDE / GSDE / SDE .ipynb cell 7 (the fields these blocks come from).
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "snsde_internal.h"

namespace {

struct AffineJobs {
    snsde_affine_job job[SNSDE_MAX_AFFINE_JOBS];
    int32_t row0[SNSDE_MAX_AFFINE_JOBS + 1];      // first workgroup of each job
    int32_t n;
};

__device__ __forceinline__ int find_job(const AffineJobs& J, int blk) {
    int j = 0;
    while (j + 1 < J.n && blk >= J.row0[j + 1]) ++j;
    return j;
}

// forward: block (job, output row r)
__global__ void __launch_bounds__(256) snsde_affine_compose_kernel(AffineJobs J, float* __restrict__ dst) {
    const int j = find_job(J, blockIdx.x);
    const snsde_affine_job& q = J.job[j];
    const int r = blockIdx.x - J.row0[j];
    const int cout = q.Cin + (q.zero_col >= 0 ? 1 : 0);
    float* w = dst + q.dst_w + (size_t)r * cout;
    for (int c = threadIdx.x; c < cout; c += blockDim.x) {
        const int ci = q.zero_col < 0 ? c : (c < q.zero_col ? c : c - 1);
        float v = 0.0f;
        if (c != q.zero_col) {
            if (!q.w_outer) v = q.w_inner[(size_t)r * q.Cin + ci];
            else {
                const float* wo = q.w_outer + (size_t)r * q.K;
                for (int k = 0; k < q.K; ++k) v = fmaf(wo[k], q.w_inner[(size_t)k * q.Cin + ci], v);
            }
        }
        w[c] = v;
    }
    if (threadIdx.x == 0 && q.dst_b >= 0) {
        float v = 0.0f;
        if (!q.w_outer) v = q.b_inner ? q.b_inner[r] : 0.0f;
        else {
            if (q.b_inner) { const float* wo = q.w_outer + (size_t)r * q.K; for (int k = 0; k < q.K; ++k) v = fmaf(wo[k], q.b_inner[k], v); }
            if (q.b_outer) v += q.b_outer[r];
        }
        dst[q.dst_b + r] = v;
    }
}

// backward: per job two families of rows - block b < R: row b of dL/dW_outer (+ dL/db_outer[b]) or, for a copy job, row b of
// dL/dW_inner (+ dL/db_inner[b]); block b >= R (compositions only): row b - R of dL/dW_inner (+ dL/db_inner[b - R])
__global__ void __launch_bounds__(256) snsde_affine_backward_kernel(AffineJobs J, const float* __restrict__ gdst, float* __restrict__ gsrc) {
    const int j = find_job(J, blockIdx.x);
    const snsde_affine_job& q = J.job[j];
    const int b = blockIdx.x - J.row0[j];
    const int cout = q.Cin + (q.zero_col >= 0 ? 1 : 0);
    auto gcol = [&](int ci) { return q.zero_col < 0 ? ci : (ci < q.zero_col ? ci : ci + 1); };      // source column -> block column
    const float* G = gdst + q.dst_w;                       // (R, cout)
    const float* gb = q.dst_b >= 0 ? gdst + q.dst_b : nullptr;
    if (!q.w_outer) {                                      // copy
        if (q.g_w_inner >= 0)
            for (int c = threadIdx.x; c < q.Cin; c += blockDim.x) gsrc[q.g_w_inner + (size_t)b * q.Cin + c] = G[(size_t)b * cout + gcol(c)];
        if (threadIdx.x == 0 && q.g_b_inner >= 0) gsrc[q.g_b_inner + b] = gb ? gb[b] : 0.0f;
        return;
    }
    if (b < q.R) {                                         // dW_outer[b, k] = sum_c G[b, c] W_inner[k, c] + gb[b] b_inner[k];  db_outer[b] = gb[b]
        const float gbb = gb ? gb[b] : 0.0f;
        if (q.g_w_outer >= 0)
            for (int k = threadIdx.x; k < q.K; k += blockDim.x) {
                const float* wi = q.w_inner + (size_t)k * q.Cin;
                float v = 0.0f;
                for (int c = 0; c < q.Cin; ++c) v = fmaf(G[(size_t)b * cout + gcol(c)], wi[c], v);
                if (q.b_inner) v = fmaf(gbb, q.b_inner[k], v);
                gsrc[q.g_w_outer + (size_t)b * q.K + k] = v;
            }
        if (threadIdx.x == 0 && q.g_b_outer >= 0) gsrc[q.g_b_outer + b] = gbb;
        return;
    }
    const int k = b - q.R;                                 // dW_inner[k, c] = sum_r W_outer[r, k] G[r, c];  db_inner[k] = sum_r W_outer[r, k] gb[r]
    if (q.g_w_inner >= 0)
        for (int c = threadIdx.x; c < q.Cin; c += blockDim.x) {
            float v = 0.0f;
            for (int r = 0; r < q.R; ++r) v = fmaf(q.w_outer[(size_t)r * q.K + k], G[(size_t)r * cout + gcol(c)], v);
            gsrc[q.g_w_inner + (size_t)k * q.Cin + c] = v;
        }
    if (threadIdx.x == 0 && q.g_b_inner >= 0) {
        float v = 0.0f;
        if (gb) for (int r = 0; r < q.R; ++r) v = fmaf(q.w_outer[(size_t)r * q.K + k], gb[r], v);
        gsrc[q.g_b_inner + k] = v;
    }
}

int fill(const snsde_affine_job* jobs, int32_t n, bool backward, AffineJobs* J) {
    if (!jobs) return SNSDE_ERR_NULL;
    if (n <= 0 || n > SNSDE_MAX_AFFINE_JOBS) return SNSDE_ERR_DIMS;
    int row = 0;
    for (int i = 0; i < n; ++i) {
        const snsde_affine_job& q = jobs[i];
        if (!q.w_inner) return SNSDE_ERR_NULL;
        if (q.R <= 0 || q.Cin <= 0 || (q.w_outer && q.K <= 0) || q.zero_col > q.Cin || q.dst_w < 0) return SNSDE_ERR_DIMS;
        J->job[i] = q;
        J->row0[i] = row;
        row += q.R + ((backward && q.w_outer) ? q.K : 0);
    }
    J->row0[n] = row;
    J->n = n;
    return SNSDE_OK;
}

}  // namespace

extern "C" {

int snsde_affine_compose(const snsde_affine_job* jobs, int32_t n_jobs, float* dst, void* hip_stream) {
    if (!dst) return SNSDE_ERR_NULL;
    AffineJobs J;
    if (const int rc = fill(jobs, n_jobs, false, &J)) return rc;
    hipLaunchKernelGGL(snsde_affine_compose_kernel, dim3(J.row0[n_jobs]), dim3(256), 0, static_cast<hipStream_t>(hip_stream), J, dst);
    return hipGetLastError() == hipSuccess ? SNSDE_OK : SNSDE_ERR_LAUNCH;
}

int snsde_affine_compose_backward(const snsde_affine_job* jobs, int32_t n_jobs, const float* grad_dst, float* grad_src, void* hip_stream) {
    if (!grad_dst || !grad_src) return SNSDE_ERR_NULL;
    AffineJobs J;
    if (const int rc = fill(jobs, n_jobs, true, &J)) return rc;
    hipLaunchKernelGGL(snsde_affine_backward_kernel, dim3(J.row0[n_jobs]), dim3(256), 0, static_cast<hipStream_t>(hip_stream), J, grad_dst, grad_src);
    return hipGetLastError() == hipSuccess ? SNSDE_OK : SNSDE_ERR_LAUNCH;
}

}  // extern "C"
