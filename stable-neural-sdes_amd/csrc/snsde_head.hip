// Readout head of the wrappers in one launch (inference): out = W2 relu(bn(W1 act(x) + b1)) + b2
//   classification  NeuralSDE.linear        Linear, BatchNorm1d (running statistics), ReLU, Dropout (identity), Linear
//                                           (benchmark_classification/models_sde/neuralsde.py:59-61, 119)
//   forecasting     NeuralSDE_forecasting   Linear, ReLU, Linear on the last output_time states (benchmark_forecasting/...:186)
//   torch_ists      NeuralSDE               Tanh, Linear, ReLU, Linear (nsde_model.py)
// A workgroup takes HR rows: the input rows sit in LDS, W1 is streamed through LDS in 32-column slabs (coalesced reads,
// transposed on the way in so the compute loop reads it conflict-free with lanes over output features), the hidden rows stay
// in LDS for the second layer.  fp32 FMA chains, sequential in k.
#include "snsde_internal.h"

namespace {

constexpr int HR = 4;      // rows per workgroup (measured 8 / 4 / 2: 13.7 / 9.9 / 9.5 us at 1024 x 128 x 128)
constexpr int HT = 256;    // threads
constexpr int HK = 32;     // slab width
// HPRE (template): slab elements a thread carries while the previous slab is consumed = hidden / 8, instantiated 8 .. 64
// (measured: 128-column slabs - a 128-wide layer through LDS in one piece - are slower, 24 vs 19 us at 1024 x 128 x 128)

__host__ __device__ __forceinline__ int round_up(int x, int m) { return (x + m - 1) / m * m; }

template <int HPRE>
__global__ void __launch_bounds__(HT) snsde_head_kernel(snsde_head h) {
    constexpr int CC = HPRE > 32 ? 2 : 1;      // output features per thread
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int K = h.in_features, N = h.hidden, O = h.out_features;
    const int Kp = round_up(K, HK), ldx = Kp + 4, ldh = N + 1, ldt = N + 1;
    float* xs = lds;                 // [HR][ldx]   input rows, zero-padded to whole slabs
    float* hs = xs + HR * ldx;       // [HR][ldh]   hidden rows
    float* wt = hs + HR * ldh;       // [HK][ldt]   the current slab of W1, transposed
    const int tid = threadIdx.x, row0 = blockIdx.x * HR;
    for (int i = tid; i < HR * Kp; i += HT) {
        const int r = i / Kp, k = i - r * Kp, row = row0 + r;
        float v = (row < h.rows && k < K) ? h.x[(size_t)row * K + k] : 0.0f;
        if (h.input_tanh) v = tanhf(v);
        xs[r * ldx + k] = v;
    }
    // slab element q of this thread: flat index i = tid + q HT over (feature nn = i / HK, column kk = i % HK): a wave reads
    // two 128-byte row segments of W1 per load
    const int cnt = (N * HK + HT - 1) / HT;
    float pre[HPRE];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < HPRE; ++q) {
            const int i = tid + q * HT, nn = i / HK, kk = i - nn * HK;
            pre[q] = (q < cnt && nn < N && k0 + kk < K) ? h.w1[(size_t)nn * K + k0 + kk] : 0.0f;
        }
    };
    auto publish = [&]() {
#pragma unroll
        for (int q = 0; q < HPRE; ++q) {
            const int i = tid + q * HT, nn = i / HK, kk = i - nn * HK;
            if (q < cnt && nn < N) wt[kk * ldt + nn] = pre[q];
        }
    };
    float acc[CC][HR];       // output features tid (and tid + HT)
#pragma unroll
    for (int c = 0; c < CC; ++c)
#pragma unroll
        for (int r = 0; r < HR; ++r) acc[c][r] = 0.0f;
    fetch(0);
    for (int k0 = 0; k0 < Kp; k0 += HK) {
        __syncthreads();             // xs written / previous slab consumed
        publish();
        __syncthreads();
        if (k0 + HK < Kp) fetch(k0 + HK);      // in flight while this slab is consumed
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            const int n = tid + c * HT;
            if (n >= N) break;
#pragma unroll
            for (int k4 = 0; k4 < HK / 4; ++k4) {
                const float w0 = wt[(4 * k4) * ldt + n], w1 = wt[(4 * k4 + 1) * ldt + n];
                const float w2 = wt[(4 * k4 + 2) * ldt + n], w3 = wt[(4 * k4 + 3) * ldt + n];
#pragma unroll
                for (int r = 0; r < HR; ++r) {
                    const float4 xv = *reinterpret_cast<const float4*>(xs + r * ldx + k0 + 4 * k4);
                    acc[c][r] = fmaf(xv.x, w0, acc[c][r]);
                    acc[c][r] = fmaf(xv.y, w1, acc[c][r]);
                    acc[c][r] = fmaf(xv.z, w2, acc[c][r]);
                    acc[c][r] = fmaf(xv.w, w3, acc[c][r]);
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CC; ++c) {
        const int n = tid + c * HT;
        if (n >= N) break;
        float sc = 1.0f, sh = 0.0f;
        if (h.bn_mean) {       // BatchNorm1d with running statistics: (v - mean) / sqrt(var + eps) * weight + bias
            sc = 1.0f / sqrtf(h.bn_var[n] + h.bn_eps);
            if (h.bn_weight) sc *= h.bn_weight[n];
            sh = (h.bn_bias ? h.bn_bias[n] : 0.0f) - h.bn_mean[n] * sc;
        }
        const float b = h.b1 ? h.b1[n] : 0.0f;
#pragma unroll
        for (int r = 0; r < HR; ++r) hs[r * ldh + n] = fmaxf(fmaf(acc[c][r] + b, sc, sh), 0.0f);
    }
    __syncthreads();
    // second layer: 16 lanes per output feature (coalesced pieces of its W2 row), all rows of the tile per pass
    const int kl = tid & 15, og = tid >> 4;
    for (int o0 = 0; o0 < O; o0 += HT / 16) {
        const int o = o0 + og;
        float part[HR];
#pragma unroll
        for (int r = 0; r < HR; ++r) part[r] = 0.0f;
        if (o < O) {
            const float* wp = h.w2 + (size_t)o * N;
            for (int k = kl; k < N; k += 16) {
                const float w = wp[k];
#pragma unroll
                for (int r = 0; r < HR; ++r) part[r] = fmaf(hs[r * ldh + k], w, part[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < HR; ++r) {
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) part[r] += __shfl_xor(part[r], off, 64);
        }
        if (o < O && kl == 0) {
            const float b = h.b2 ? h.b2[o] : 0.0f;
#pragma unroll
            for (int r = 0; r < HR; ++r)
                if (row0 + r < h.rows) h.out[(size_t)(row0 + r) * O + o] = part[r] + b;
        }
    }
}

}  // namespace

extern "C" int snsde_readout_head(const snsde_head* h, void* hip_stream) {
    if (!h) return SNSDE_ERR_NULL;
    if (h->struct_size != sizeof(snsde_head)) return SNSDE_ERR_ABI;
    if (!h->x || !h->w1 || !h->w2 || !h->out) return SNSDE_ERR_NULL;
    if (h->rows <= 0 || h->in_features <= 0 || h->hidden <= 0 || h->out_features <= 0) return SNSDE_ERR_DIMS;
    if ((h->bn_mean != nullptr) != (h->bn_var != nullptr)) return SNSDE_ERR_NULL;
    if (h->hidden > 2 * HT) return SNSDE_ERR_DIMS;       // at most two output features per thread
    const size_t bytes = ((size_t)HR * (round_up(h->in_features, HK) + 4) + (size_t)(HR + HK) * (h->hidden + 1)) * sizeof(float);
    if (bytes > 160 * 1024) return SNSDE_ERR_LDS;
    const int pre = (h->hidden * HK + HT - 1) / HT;
    auto launch = [&](auto kernel) {
        if (bytes > 64 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
            return SNSDE_ERR_LDS;
        hipLaunchKernelGGL(kernel, dim3((h->rows + HR - 1) / HR), dim3(HT), bytes, static_cast<hipStream_t>(hip_stream), *h);
        return SNSDE_OK;
    };
    const int rc = pre <= 8 ? launch(snsde_head_kernel<8>) : (pre <= 16 ? launch(snsde_head_kernel<16>)
                 : (pre <= 32 ? launch(snsde_head_kernel<32>) : launch(snsde_head_kernel<64>)));
    if (rc) return rc;
    return hipGetLastError() == hipSuccess ? SNSDE_OK : SNSDE_ERR_LAUNCH;
}
