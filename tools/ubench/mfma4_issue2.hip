// Microbenchmark 2: what else shares the issue port with v_mfma_f32_4x4x1_16b_f32 on gfx950 (SALU, s_nop, packed f32,
// blocked vs interleaved placement), the same for v_mfma_f32_16x16x4_f32, and the LDS write -> barrier -> read round trip.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma4_issue2 mfma4_issue2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 0, 0, 4)
#define MFMA16(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0)
#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(v1))

template <int VARIANT>
__global__ void __launch_bounds__(512) k(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int lane = threadIdx.x & 63;
    float a0 = lane * 0.001f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
    float b0 = 0.5f + lane, b1 = b0 * 0.5f;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float v0 = lane, v1 = 1.0f, v2 = 2.0f, v3 = 3.0f;
    f32x2 p0 = {v0, v1}, p1 = {v2, v3};
    int sacc = 0;
    lds[threadIdx.x] = lane;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (VARIANT == 0) {          // 2 chains + s_nop 0 per mfma
                MFMA(c0, a0, b0); asm volatile("s_nop 0"); MFMA(c1, a1, b1); asm volatile("s_nop 0");
                MFMA(c0, a2, b0); asm volatile("s_nop 0"); MFMA(c1, a3, b1); asm volatile("s_nop 0");
            } else if constexpr (VARIANT == 1) {   // 2 chains + s_add per mfma
                MFMA(c0, a0, b0); asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc)); MFMA(c1, a1, b1); asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
                MFMA(c0, a2, b0); asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc)); MFMA(c1, a3, b1); asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
            } else if constexpr (VARIANT == 2) {   // 2 chains + pk_fma per mfma
                MFMA(c0, a0, b0); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p0) : "v"(p1));
                MFMA(c1, a1, b1); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p0) : "v"(p1));
                MFMA(c0, a2, b0); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p0) : "v"(p1));
                MFMA(c1, a3, b1); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p0) : "v"(p1));
            } else if constexpr (VARIANT == 3) {   // pk_fma alone (4 per slot group)
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p0) : "v"(p1));
            } else if constexpr (VARIANT == 4) {   // blocked: 4 mfma then 4 fma
                MFMA(c0, a0, b0); MFMA(c1, a1, b1); MFMA(c0, a2, b0); MFMA(c1, a3, b1);
                FMA(v0); FMA(v2); FMA(v3); FMA(v0);
            } else if constexpr (VARIANT == 5) {   // 16x16x4: 2 chains, 4 mfma
                MFMA16(c0, a0, b0); MFMA16(c1, a1, b1); MFMA16(c0, a2, b0); MFMA16(c1, a3, b1);
            } else if constexpr (VARIANT == 6) {   // 16x16x4 + 2 fma per mfma
                MFMA16(c0, a0, b0); FMA(v0); FMA(v2); MFMA16(c1, a1, b1); FMA(v3); FMA(v0);
                MFMA16(c0, a2, b0); FMA(v2); FMA(v3); MFMA16(c1, a3, b1); FMA(v0); FMA(v2);
            } else if constexpr (VARIANT == 7) {   // 16x16x4 + 4 fma per mfma
                MFMA16(c0, a0, b0); FMA(v0); FMA(v2); FMA(v3); FMA(v0); MFMA16(c1, a1, b1); FMA(v2); FMA(v3); FMA(v0); FMA(v2);
                MFMA16(c0, a2, b0); FMA(v3); FMA(v0); FMA(v2); FMA(v3); MFMA16(c1, a3, b1); FMA(v0); FMA(v2); FMA(v3); FMA(v0);
            } else if constexpr (VARIANT == 8) {   // 16x16x4 + 6 fma per mfma
                MFMA16(c0, a0, b0); FMA(v0); FMA(v2); FMA(v3); FMA(v0); FMA(v2); FMA(v3); MFMA16(c1, a1, b1); FMA(v0); FMA(v2); FMA(v3); FMA(v0); FMA(v2); FMA(v3);
                MFMA16(c0, a2, b0); FMA(v0); FMA(v2); FMA(v3); FMA(v0); FMA(v2); FMA(v3); MFMA16(c1, a3, b1); FMA(v0); FMA(v2); FMA(v3); FMA(v0); FMA(v2); FMA(v3);
            } else if constexpr (VARIANT == 9) {   // LDS write -> barrier -> 4 reads -> wait   (one round trip per "4 slots")
                float w = v0;
                asm volatile("ds_write_b32 %0, %1" :: "v"(threadIdx.x * 4), "v"(w) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");
                f32x4 r0, r1, r2, r3;
                asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:64\n ds_read_b128 %2, %4 offset:128\n ds_read_b128 %3, %4 offset:192\n s_waitcnt lgkmcnt(0)"
                             : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"((lane & 15) * 16) : "memory");
                v0 = r0[0] + r1[1] + r2[2] + r3[3];
                asm volatile("s_barrier" ::: "memory");
            } else if constexpr (VARIANT == 10) {  // 2 chains + ds_read_b128 (no wait) per 2 mfma
                f32x4 t;
                MFMA(c0, a0, b0); MFMA(c1, a1, b1);
                asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((lane & 15) * 16));
                MFMA(c0, a2, b0); MFMA(c1, a3, b1);
                asm volatile("ds_read_b128 %0, %1 offset:256" : "=v"(t) : "v"((lane & 15) * 16));
            } else if constexpr (VARIANT == 11) {  // 2 chains + dependent VALU on the mfma RESULT every 4 (epilogue-like latency)
                MFMA(c0, a0, b0); MFMA(c1, a1, b1); MFMA(c0, a2, b0); MFMA(c1, a3, b1);
                v0 += c0[0] + c1[0];
            } else if constexpr (VARIANT == 12) {  // v_max / v_bfi / cndmask mix alone: 4 independent simple ops
                asm volatile("v_max_f32 %0, %0, %2\n v_bfi_b32 %1, %2, %1, %0\n v_max_f32 %0, %0, %2\n v_bfi_b32 %1, %2, %1, %0" : "+v"(v0), "+v"(v2) : "v"(v1));
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    f32x4 c = c0 + c1 + c2 + c3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = c[0] + c[1] + c[2] + c[3] + v0 + v2 + v3 + a3 + p0[0] + p0[1] + sacc;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int V> void run(const char* name, int threads, float* out, unsigned long long* cyc) {
    const int iters = 1000, grid = 256;
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(threads), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(threads), 0, 0, out, cyc, iters);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid * threads / 64);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto x : h) s += x; s /= h.size();
    const double slots = (double)iters * 8 * 4;
    printf("%-52s waves/SIMD=%d  ticks/slot %7.2f   wall ns/slot %7.3f\n", name, threads / 256, s / slots, ms * 1e6 / slots);
}

int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    for (int threads : {256, 512}) {
        run<0>("mfma4 2ch + s_nop 0", threads, out, cyc);
        run<1>("mfma4 2ch + s_add", threads, out, cyc);
        run<2>("mfma4 2ch + pk_fma", threads, out, cyc);
        run<3>("pk_fma alone (per op)", threads, out, cyc);
        run<12>("max/bfi alone (per op)", threads, out, cyc);
        run<4>("blocked 4 mfma4 + 4 fma (per pair)", threads, out, cyc);
        run<10>("mfma4 2ch + 0.5 ds_read_b128 nowait", threads, out, cyc);
        run<11>("mfma4 2ch + result use every 4", threads, out, cyc);
        run<5>("mfma16x16x4 2ch", threads, out, cyc);
        run<6>("mfma16x16x4 + 2 fma", threads, out, cyc);
        run<7>("mfma16x16x4 + 4 fma", threads, out, cyc);
        run<8>("mfma16x16x4 + 6 fma", threads, out, cyc);
        run<9>("lds write->barrier->4 reads->barrier (x4 per trip)", threads, out, cyc);
    }
    return 0;
}
