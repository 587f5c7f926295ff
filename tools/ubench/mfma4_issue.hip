// Microbenchmark: issue behaviour of v_mfma_f32_4x4x1_16b_f32 on gfx950 (one or two waves per SIMD), alone and with
// VALU / DPP / transcendental / LDS fillers between the MFMAs.  Prints cycles per MFMA (s_memtime) for each variant.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma4_issue mfma4_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 0, 0, 4)

template <int VARIANT>
__global__ void __launch_bounds__(512) k(float* out, unsigned long long* cyc, int iters) {
    __shared__ float lds[4096];
    const int lane = threadIdx.x & 63;
    float a0 = lane * 0.001f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
    float b0 = 0.5f + lane, b1 = b0 * 0.5f;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float v0 = lane, v1 = 1.0f, v2 = 2.0f, v3 = 3.0f;
    lds[threadIdx.x] = lane;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (VARIANT == 0) {          // 1 chain
                MFMA(c0, a0, b0); MFMA(c0, a1, b1); MFMA(c0, a2, b0); MFMA(c0, a3, b1);
            } else if constexpr (VARIANT == 1) {   // 2 chains
                MFMA(c0, a0, b0); MFMA(c1, a1, b1); MFMA(c0, a2, b0); MFMA(c1, a3, b1);
            } else if constexpr (VARIANT == 2) {   // 4 chains
                MFMA(c0, a0, b0); MFMA(c1, a1, b1); MFMA(c2, a2, b0); MFMA(c3, a3, b1);
            } else if constexpr (VARIANT == 3) {   // 2 chains + 1 fma per mfma
                MFMA(c0, a0, b0); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v0) : "v"(v1));
                MFMA(c1, a1, b1); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v2) : "v"(v1));
                MFMA(c0, a2, b0); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v3) : "v"(v1));
                MFMA(c1, a3, b1); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v0) : "v"(v1));
            } else if constexpr (VARIANT == 4) {   // 2 chains + 2 fma per mfma
                MFMA(c0, a0, b0); asm volatile("v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %1, %1, %2, %2" : "+v"(v0), "+v"(v2) : "v"(v1));
                MFMA(c1, a1, b1); asm volatile("v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %1, %1, %2, %2" : "+v"(v3), "+v"(v0) : "v"(v1));
                MFMA(c0, a2, b0); asm volatile("v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %1, %1, %2, %2" : "+v"(v2), "+v"(v3) : "v"(v1));
                MFMA(c1, a3, b1); asm volatile("v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %1, %1, %2, %2" : "+v"(v0), "+v"(v2) : "v"(v1));
            } else if constexpr (VARIANT == 5) {   // 2 chains + 1 transcendental per mfma
                MFMA(c0, a0, b0); asm volatile("v_exp_f32 %0, %0" : "+v"(v0));
                MFMA(c1, a1, b1); asm volatile("v_rcp_f32 %0, %0" : "+v"(v2));
                MFMA(c0, a2, b0); asm volatile("v_exp_f32 %0, %0" : "+v"(v3));
                MFMA(c1, a3, b1); asm volatile("v_rcp_f32 %0, %0" : "+v"(v0));
            } else if constexpr (VARIANT == 6) {   // 2 chains + 1 dpp per mfma (independent regs)
                MFMA(c0, a0, b0); asm volatile("v_add_f32_dpp %0, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(v0) : "v"(v1));
                MFMA(c1, a1, b1); asm volatile("v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(v2) : "v"(v1));
                MFMA(c0, a2, b0); asm volatile("v_add_f32_dpp %0, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(v3) : "v"(v1));
                MFMA(c1, a3, b1); asm volatile("v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(v0) : "v"(v1));
            } else if constexpr (VARIANT == 7) {   // 2 chains + v_mad_u64_u32 per mfma
                unsigned long long q;
                MFMA(c0, a0, b0); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(q) : "v"(v0), "v"(v1) : "vcc"); v2 += (float)(unsigned)q;
                MFMA(c1, a1, b1);
                MFMA(c0, a2, b0); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(q) : "v"(v3), "v"(v1) : "vcc"); v2 += (float)(unsigned)q;
                MFMA(c1, a3, b1);
            } else if constexpr (VARIANT == 8) {   // VALU only: 8 dependent fma per "4 mfma" slot => latency
                asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1" : "+v"(v0) : "v"(v1));
            } else if constexpr (VARIANT == 9) {   // VALU only: 4 independent fma
                asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(v0), "+v"(v2), "+v"(v3), "+v"(a3) : "v"(v1));
            } else if constexpr (VARIANT == 10) {  // 2 chains + ds_read_b128 every 4 mfma
                f32x4 t;
                MFMA(c0, a0, b0); MFMA(c1, a1, b1);
                asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((lane & 15) * 16));
                MFMA(c0, a2, b0); MFMA(c1, a3, b1);
                asm volatile("s_waitcnt lgkmcnt(0)"); v2 += t[0];
            } else if constexpr (VARIANT == 11) {  // dependent transcendental chain latency (4 per slot)
                asm volatile("v_exp_f32 %0, %0\n v_rcp_f32 %0, %0\n v_exp_f32 %0, %0\n v_rcp_f32 %0, %0" : "+v"(v0));
            } else if constexpr (VARIANT == 12) {  // dependent DPP chain (4 per slot)
                asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                             "s_nop 1\n v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(v0));
            } else if constexpr (VARIANT == 13) {  // 1 chain + 1 fma per mfma
                MFMA(c0, a0, b0); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v0) : "v"(v1));
                MFMA(c0, a1, b1); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v2) : "v"(v1));
                MFMA(c0, a2, b0); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v3) : "v"(v1));
                MFMA(c0, a3, b1); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v0) : "v"(v1));
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    f32x4 c = c0 + c1 + c2 + c3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = c[0] + c[1] + c[2] + c[3] + v0 + v2 + v3 + a3;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int V> void run(const char* name, int threads, float* out, unsigned long long* cyc) {
    const int iters = 2000, grid = 256;
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid * threads / 64);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto x : h) s += x; s /= h.size();
    const double slots = (double)iters * 8 * 4;   // "MFMA slots" per wave
    printf("%-44s waves/SIMD=%d  memtime ticks/slot %.2f   wall ns/slot %.3f\n", name, threads / 256, s / slots, ms * 1e6 / slots);
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    for (int threads : {256, 512}) {
        run<0>("mfma 1 chain", threads, out, cyc);
        run<1>("mfma 2 chains", threads, out, cyc);
        run<2>("mfma 4 chains", threads, out, cyc);
        run<13>("mfma 1 chain + 1 fma", threads, out, cyc);
        run<3>("mfma 2 chains + 1 fma", threads, out, cyc);
        run<4>("mfma 2 chains + 2 fma", threads, out, cyc);
        run<5>("mfma 2 chains + 1 exp/rcp", threads, out, cyc);
        run<6>("mfma 2 chains + 1 dpp", threads, out, cyc);
        run<7>("mfma 2 chains + 0.5 mad_u64_u32+cvt+add", threads, out, cyc);
        run<10>("mfma 2 chains + ds_read_b128 wait /4", threads, out, cyc);
        run<8>("valu: dependent fma (per op)", threads, out, cyc);
        run<9>("valu: independent fma (per op)", threads, out, cyc);
        run<11>("valu: dependent exp/rcp (per op)", threads, out, cyc);
        run<12>("valu: dependent dpp + nop1 (per op)", threads, out, cyc);
    }
    return 0;
}
