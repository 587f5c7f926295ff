// Microbenchmark: what a per-step global store costs a wave that is otherwise issuing f32 MFMAs (the training-mode forward
// saves five (N, B, H) planes per step).  512 threads = 8 waves per workgroup, one workgroup per CU, every wave runs
// ITERS x [32 MFMAs + K stores]; variants: dword per lane in the kernels' layout (4 rows x 16 features = four 64-byte segments per
// wave), the same bytes as dwordx4 from 16 lanes, fully coalesced dword, and loads instead of stores.
// build: hipcc --offload-arch=gfx950 -O3 -o store_cost store_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K, int MODE>
__global__ void __launch_bounds__(512, 2) k(float* __restrict__ buf, float* __restrict__ out, int iters, int B, int H) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a0 = lane * 0.001f, b0 = 0.5f + lane;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0;
    const int r = lane & 3, s = (lane >> 2) & 3, q = lane >> 4;
    const int row = blockIdx.x * 4 + r, fcol = wave * 16 + 4 * q + s;
    const size_t BH = (size_t)B * H;
    float acc = 0.f;
    f32x4 pre = {0, 0, 0, 0};
    for (int n = 0; n < iters; ++n) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, b0, c0, 0, 0, 4);
            c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(b0, a0, c1, 0, 0, 4);
        }
        const float v = c0[0] + c1[1];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            float* p = buf + ((size_t)n * K + j) * BH;
            if (MODE == 0) p[(size_t)row * H + fcol] = v;                                   // kernel layout: dword per lane
            else if (MODE == 1) { if (lane < 16) *reinterpret_cast<f32x4*>(p + (size_t)(blockIdx.x * 4 + (lane >> 2)) * H + wave * 16 + 4 * (lane & 3)) = f32x4{v, v, v, v}; }
            else if (MODE == 2) p[(size_t)blockIdx.x * 512 + threadIdx.x] = v;              // fully coalesced dword
            else if (MODE == 3) acc += p[(size_t)row * H + fcol];                           // loads in the kernel layout
            else if (MODE == 4) { if (j == 0) *reinterpret_cast<f32x4*>(buf + (size_t)n * K * BH + ((size_t)row * H + fcol) * 4) = f32x4{v, v, v, v}; }   // ONE dwordx4 per lane: 4 planes interleaved
            else if (MODE == 5) { if (j == 0) { const f32x4 t = *reinterpret_cast<const f32x4*>(buf + (size_t)n * K * BH + ((size_t)row * H + fcol) * 4); acc += t[0] + t[3]; } }
            else { const f32x4 t = *reinterpret_cast<const f32x4*>(buf + (size_t)((n + 1) % iters) * K * BH + ((size_t)row * H + fcol) * 4); if (j == 0) { acc += pre[0] + pre[3]; pre = t; } }   // x4 load one step ahead
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[0] + acc;
}

template <int K, int MODE> double run(float* buf, float* out, int iters, int B, int H) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<K, MODE><<<B / 4, 512>>>(buf, out, iters, B, H);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) k<K, MODE><<<B / 4, 512>>>(buf, out, iters, B, H);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1e3 / iters;     // us per iteration
}

int main() {
    const int B = 1024, H = 128, iters = 100;
    float *buf, *out;
    hipMalloc(&buf, (size_t)iters * 5 * B * H * 4); hipMalloc(&out, (size_t)B / 4 * 512 * 4);
    hipMemset(buf, 0, (size_t)iters * 5 * B * H * 4);
    const char* names[7] = {"dword per lane, kernel layout (4 x 64 B per wave)", "dwordx4 from lanes 0-15 (same bytes)", "dword per lane, coalesced 256 B", "dword LOAD per lane, kernel layout", "ONE dwordx4 store per lane (K planes interleaved, K = 4 / 5 only meaningful)", "ONE dwordx4 load per lane, used at once", "ONE dwordx4 load per lane, one iteration ahead"};
    double base = run<0, 0>(buf, out, iters, B, H);
    printf("32 MFMAs per iteration, no memory op: %.3f us per iteration\n", base);
#define ROW(M) printf("%-52s K=1 %+.3f  K=3 %+.3f  K=5 %+.3f us per iteration over the base\n", names[M], run<1, M>(buf, out, iters, B, H) - base, \
                      run<3, M>(buf, out, iters, B, H) - base, run<5, M>(buf, out, iters, B, H) - base);
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6)
    return 0;
}
