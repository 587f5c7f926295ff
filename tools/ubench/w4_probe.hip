// Probe of the wave-owns-rows building blocks (csrc/snsde_w4_kernel.h): layout of v_mfma_f32_4x4x1_16b_f32 under CBSZ = 4 / ABID,
// the quad transpose, and the issue cost of a 64-MFMA layer + epilogue from one wave per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o w4_probe w4_probe.hip && ./w4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <utility>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int ABID> __device__ __forceinline__ f32x4 mfma_bk(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, ABID, 0); }
template <int... KB>
__device__ __forceinline__ void gemm64(const float (&xt)[4], const float* w, f32x4& c, f32x4& d, std::integer_sequence<int, KB...>) {
    ((c = mfma_bk<KB>(xt[0], w[4 * KB], c), d = mfma_bk<KB>(xt[1], w[4 * KB + 1], d), c = mfma_bk<KB>(xt[2], w[4 * KB + 2], c),
      d = mfma_bk<KB>(xt[3], w[4 * KB + 3], d)), ...);
}
// eight v_cndmask_b32_dpp: D = VCC ? src1 : dpp(src0).  (Written with selects around __builtin_amdgcn_update_dpp hipcc moves the DPP
// moves INTO the select's EXEC-masked region, where their source lanes are inactive and read as zero.)
__device__ __forceinline__ void quad_transpose(const float (&v)[4], float (&t)[4], bool, bool) {
    float a0, a1, a2, a3;
    asm volatile(
        "s_mov_b64 vcc, %12\n\t"                                       // even lanes keep their own register
        "s_nop 1\n\t"
        "v_cndmask_b32_dpp %0, %9, %8, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"     // a0 = even ? v0 : v1[lane ^ 1]
        "v_cndmask_b32_dpp %2, %11, %10, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"   // a2 = even ? v2 : v3[lane ^ 1]
        "s_mov_b64 vcc, %13\n\t"
        "v_cndmask_b32_dpp %1, %8, %9, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"     // a1 = odd ? v1 : v0[lane ^ 1]
        "v_cndmask_b32_dpp %3, %10, %11, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"   // a3 = odd ? v3 : v2[lane ^ 1]
        "s_mov_b64 vcc, %14\n\t"                                       // lanes 0, 1 of a quad
        "v_cndmask_b32_dpp %4, %2, %0, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"     // t0 = lo ? a0 : a2[lane ^ 2]
        "s_nop 0\n\t"
        "v_cndmask_b32_dpp %5, %3, %1, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"     // t1 = lo ? a1 : a3[lane ^ 2]
        "s_mov_b64 vcc, %15\n\t"
        "v_cndmask_b32_dpp %6, %0, %2, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"     // t2 = hi ? a2 : a0[lane ^ 2]
        "v_cndmask_b32_dpp %7, %1, %3, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"     // t3 = hi ? a3 : a1[lane ^ 2]
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3])
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "s"(0x5555555555555555ull), "s"(0xaaaaaaaaaaaaaaaaull), "s"(0x3333333333333333ull),
          "s"(0xccccccccccccccccull)
        : "vcc");
}
// 1. raw layout dump: D of ONE mfma with A = lane id, B = 100 * lane id, C = 0, cbsz 4 abid 1
__global__ void dump(float* out) {
    const int lane = threadIdx.x;
    f32x4 c = {0, 0, 0, 0};
    c = mfma_bk<1>((float)lane, 100.0f * lane, c);
    for (int i = 0; i < 4; ++i) out[i * 64 + lane] = c[i];
    float v[4] = {(float)(lane), 100.0f + lane, 200.0f + lane, 300.0f + lane}, t[4];
    quad_transpose(v, t, lane & 1, lane & 2);
    for (int i = 0; i < 4; ++i) out[256 + i * 64 + lane] = t[i];
}
// 2. layer: out[4][64] = X[4][64] . W[64][64]^T
__global__ void layer(const float* X, const float* W, float* out) {
    const int lane = threadIdx.x;
    float y[4], yt[4], w[64];
    for (int i = 0; i < 4; ++i) y[i] = X[i * 64 + lane];
    quad_transpose(y, yt, lane & 1, lane & 2);
    for (int k = 0; k < 64; ++k) w[k] = W[lane * 64 + k];
    f32x4 c = {0, 0, 0, 0}, d = c;
    gemm64(yt, w, c, d, std::make_integer_sequence<int, 16>{});
    for (int i = 0; i < 4; ++i) out[i * 64 + lane] = c[i] + d[i];
}
// 3. timing: NL dependent layers per step, relu + transpose hand-off, STEPS steps, one wave per workgroup
template <int NL, int CHAINS>
__global__ void __launch_bounds__(64) chain(const float* W, float* out, int steps, long long* cyc) {
    const int lane = threadIdx.x;
    float w[NL][64], y[4], yt[4];
    for (int l = 0; l < NL; ++l) for (int k = 0; k < 64; ++k) w[l][k] = W[(l * 64 + lane) * 64 + k] * 0.05f;
    for (int i = 0; i < 4; ++i) y[i] = 0.01f * (lane + i);
    quad_transpose(y, yt, lane & 1, lane & 2);
    const long long t0 = __builtin_readcyclecounter();
    for (int n = 0; n < steps; ++n) {
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            f32x4 c = {0.1f, 0.1f, 0.1f, 0.1f}, d = {0, 0, 0, 0};
            gemm64(yt, w[l], c, d, std::make_integer_sequence<int, 16>{});
            float v[4];
            for (int i = 0; i < 4; ++i) v[i] = fmaxf(c[i] + d[i], 0.0f);
            quad_transpose(v, yt, lane & 1, lane & 2);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    for (int i = 0; i < 4; ++i) out[(blockIdx.x * 4 + i) * 64 + lane] = yt[i];
    if (blockIdx.x == 0 && lane == 0) *cyc = t1 - t0;
}
template <int NL>
__global__ void __launch_bounds__(64) chain_nomfma(const float* W, float* out, int steps, long long* cyc) {
    const int lane = threadIdx.x;
    float y[4], yt[4];
    for (int i = 0; i < 4; ++i) y[i] = 0.01f * (lane + i);
    quad_transpose(y, yt, lane & 1, lane & 2);
    const long long t0 = __builtin_readcyclecounter();
    for (int n = 0; n < steps; ++n) {
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            float v[4];
            for (int i = 0; i < 4; ++i) v[i] = fmaxf(yt[i] + yt[(i + 1) & 3], 0.0f);
            quad_transpose(v, yt, lane & 1, lane & 2);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    for (int i = 0; i < 4; ++i) out[(blockIdx.x * 4 + i) * 64 + lane] = yt[i];
    if (blockIdx.x == 0 && lane == 0) *cyc = t1 - t0;
}
int main() {
    float *d_out, *d_X, *d_W; long long* d_c;
    hipMalloc(&d_out, 4 << 20); hipMalloc(&d_X, 4096); hipMalloc(&d_W, 3 * 64 * 64 * 4); hipMalloc(&d_c, 8);
    std::vector<float> h(512), X(256), W(3 * 4096), o(256);
    dump<<<1, 64>>>(d_out);
    hipMemcpy(h.data(), d_out, 2048, hipMemcpyDeviceToHost);
    printf("D of one MFMA (A = lane, B = 100 lane, cbsz 4 abid 1): value = A_src * B_src\n");
    for (int i = 0; i < 4; ++i) { printf(" reg %d:", i); for (int l = 0; l < 12; ++l) printf(" %6.0f", h[i * 64 + l]); printf(" ... lane 63: %6.0f\n", h[i * 64 + 63]); }
    printf("quad transpose of v[i] = 100 i + lane:\n");
    for (int i = 0; i < 4; ++i) { printf(" t[%d]:", i); for (int l = 0; l < 8; ++l) printf(" %4.0f", h[256 + i * 64 + l]); printf("\n"); }
    srand(1);
    for (auto& v : X) v = rand() / (float)RAND_MAX - 0.5f;
    for (auto& v : W) v = rand() / (float)RAND_MAX - 0.5f;
    hipMemcpy(d_X, X.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(d_W, W.data(), 3 * 16384, hipMemcpyHostToDevice);
    layer<<<1, 64>>>(d_X, d_W, d_out);
    hipMemcpy(o.data(), d_out, 1024, hipMemcpyDeviceToHost);
    double err = 0;
    for (int i = 0; i < 4; ++i) for (int f = 0; f < 64; ++f) { double r = 0; for (int k = 0; k < 64; ++k) r += (double)X[i * 64 + k] * W[f * 64 + k]; err = fmax(err, fabs(r - o[i * 64 + f])); }
    printf("layer max |err| = %g\n", err);
    const int steps = 2000;
    for (int wgs : {1, 256, 1024, 2048}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        long long cyc = 0; float ms;
        chain<3, 2><<<wgs, 64>>>(d_W, d_out, steps, d_c); hipDeviceSynchronize();
        hipEventRecord(e0); chain<3, 2><<<wgs, 64>>>(d_W, d_out, steps, d_c); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&cyc, d_c, 8, hipMemcpyDeviceToHost);
        printf("3 layers x 64 MFMA + relu + transpose, %4d waves (one per workgroup): %.1f ns per step, %.0f clocks(100MHz?) raw %lld; per layer %.1f ns\n", wgs, ms * 1e6 / steps, (double)cyc / steps, cyc, ms * 1e6 / steps / 3);
    }
    for (int wgs : {1024, 2048}) {
        long long cyc = 0;
        chain_nomfma<3><<<wgs, 64>>>(d_W, d_out, steps, d_c); hipDeviceSynchronize();
        hipMemcpy(&cyc, d_c, 8, hipMemcpyDeviceToHost);
        printf("epilogue only (add, relu, transpose) x 3, %d waves: %.0f cycles per step = %.0f per layer\n", wgs, (double)cyc / steps, (double)cyc / steps / 3);
    }
    return 0;
}
