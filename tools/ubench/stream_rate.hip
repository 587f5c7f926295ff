// Per-CU streaming rate of L2-resident data (the H = 256 weight stream: 768 KB per workgroup-step, 16 waves x 48 blocks of 1 KB):
//   v<D>: global_load_dwordx4 -> VGPRs, D blocks in flight per wave (asm, counted vmcnt)
//   lds : global_load_lds_dwordx4 -> per-wave LDS ring of 8 slots, read back with ds_read_b128 (the snsde_m4s_kernel path)
// usage: stream_rate <workgroups> <mode: 2|4|6|lds>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BLOCKS = 48, STEPS = 200;
__device__ __forceinline__ uint64_t uni(const float* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;      // (through uint32_t: the builtin returns int, which would sign-extend)
}
template <int DEPTH>
__global__ void __launch_bounds__(1024, 1) k_vgpr(const float* __restrict__ w, float* out) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t base = uni(w + (size_t)wave * BLOCKS * 256);
    const uint32_t voff = lane * 16;
    f32x4 r[DEPTH];
    f32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < DEPTH; ++i)
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(r[i]) : "v"(voff + i * 1024), "s"(base) : "memory");
    for (int s = 0; s < STEPS; ++s) {
#pragma unroll
        for (int g = 0; g < BLOCKS; ++g) {
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r[g % DEPTH]) : "n"(DEPTH - 1));
            acc += r[g % DEPTH];
            asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "+v"(r[g % DEPTH]) : "v"(voff + ((g + DEPTH) % BLOCKS) * 1024), "s"(base) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * 1024 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
__global__ void __launch_bounds__(1024, 1) k_lds(const float* __restrict__ w, float* out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t base = uni(w + (size_t)wave * BLOCKS * 256);
    const uint32_t ringb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)(lds) + wave * 8192;
    const uint32_t ra = ringb + lane * 16, voff = lane * 16;
    f32x4 acc = {0, 0, 0, 0};
#define REFILL(blk) { const uint32_t m0_ = __builtin_amdgcn_readfirstlane(ringb + ((blk) % 8) * 1024); \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0_), "v"(voff + (blk) * 1024), "s"(base) : "memory"); }
#pragma unroll
    for (int i = 0; i < 8; ++i) REFILL(i)
    for (int s = 0; s < STEPS; ++s) {
#pragma unroll
        for (int g = 0; g < BLOCKS; g += 2) {
            f32x4 a0, a1;
            asm volatile("s_waitcnt vmcnt(6)\n\tds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(a0), "=&v"(a1) : "v"(ra), "n"((g % 8) * 1024), "n"((g % 8) * 1024 + 1024) : "memory");
            acc += a0 + a1;
            REFILL((g + 8) % BLOCKS) REFILL((g + 9) % BLOCKS)
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * 1024 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 32;
    const char* mode = argc > 2 ? argv[2] : "4";
    const size_t nfl = (size_t)16 * BLOCKS * 256;
    std::vector<float> h(nfl, 1.0f);
    float *w, *o;
    if (hipMalloc(&w, nfl * 4) != hipSuccess || hipMalloc(&o, (size_t)wgs * 1024 * 4) != hipSuccess) return 1;
    hipMemcpy(w, h.data(), nfl * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&] {
        if (!strcmp(mode, "lds")) hipLaunchKernelGGL(k_lds, dim3(wgs), dim3(1024), 131072, 0, w, o);
        else if (!strcmp(mode, "2")) hipLaunchKernelGGL(k_vgpr<2>, dim3(wgs), dim3(1024), 0, 0, w, o);
        else if (!strcmp(mode, "6")) hipLaunchKernelGGL(k_vgpr<6>, dim3(wgs), dim3(1024), 0, 0, w, o);
        else hipLaunchKernelGGL(k_vgpr<4>, dim3(wgs), dim3(1024), 0, 0, w, o);
    };
    launch(); if (hipDeviceSynchronize() != hipSuccess) { printf("fault in mode %s\n", mode); return 2; }
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)16 * BLOCKS * 1024 * STEPS;
    printf("mode %-3s %4d workgroups: %7.3f ms  %6.1f B/clk/CU at 2.4 GHz  (%.2f us per 768 KB step)\n", mode, wgs, ms,
           bytes / (ms * 1e-3) / 2.4e9, ms * 1e3 / STEPS);
    fflush(stdout);
    return 0;
}
