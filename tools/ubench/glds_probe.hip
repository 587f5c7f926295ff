// Probe of global_load_lds_dwordx4 on gfx950: does the instruction's immediate offset move the LDS destination as well as
// the global source?  (saddr form, M0 = LDS base.)  Prints where the 1 KiB landed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifndef HI_BYTES
#define HI_BYTES 2048
#endif
__global__ void k(const float* g, float* out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 40000; i += blockDim.x) lds[i] = -1.0f;
    __syncthreads();
    const uint32_t voff = lane * 16;
    const uint32_t m = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)(lds) + HI_BYTES;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\ts_waitcnt vmcnt(0)"
                 :: "s"(m), "v"(voff), "s"(g) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 40000; i += blockDim.x) out[i] = lds[i];
}
int main() {
    std::vector<float> h(8192);
    for (int i = 0; i < 8192; ++i) h[i] = (float)i;
    float *g, *o;
    hipMalloc(&g, 8192 * 4); hipMalloc(&o, 40000 * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
    hipMemcpy(g, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 160000, 0, g, o);
    std::vector<float> r(40000);
    hipMemcpy(r.data(), o, 40000 * 4, hipMemcpyDeviceToHost);
    int first = -1, last = -1;
    for (int i = 0; i < 40000; ++i) if (r[i] >= 0) { if (first < 0) first = i; last = i; }
    printf("landed floats [%d, %d]: first value %g (global float index), M0 pointed at float %d, imm offset 1024 B = 256 floats\n",
           first, last, first >= 0 ? r[first] : -1.0f, HI_BYTES / 4);
    return 0;
}
