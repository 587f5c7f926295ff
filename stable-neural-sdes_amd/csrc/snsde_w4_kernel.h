// "Wave-owns-rows" kernels for H = 64 models with a diffusion net (noise_option 14 / 15 / 18 / 19; BASELINE config 4:
// neuralsde_3_18, B = 2048, H = 64) - round 5.
//
// The 4-row tiles of snsde_mfma_kernel / snsde_m4n_kernel split a layer's OUTPUT FEATURES over the waves of a workgroup: at H = 64
// a wave issues 16 - 20 MFMAs per layer and then pays the fixed per-layer hand-off (k-slot reduce-scatter by DPP, LDS store,
// s_barrier, LDS operand reads) - 2.16 VALU instructions per MFMA on the issue port the f32 MFMAs share, three to eleven barriers
// per step, MFMA-busy 28 % (profiles/r04_pmc_net_kernels.txt).  Here ONE WAVE owns all 64 features of its 4 rows:
//
//   * v_mfma_f32_4x4x1_16b_f32 as a rank-1 update of a 4 x 64 tile: the 16 blocks are the 16 feature quads, the A operand is the
//     activation column k of the 4 rows - held by ONE lane quad and broadcast to all blocks by CBSZ = 4 / ABID = k / 4 -, the B
//     operand is row k of W^T: lane l holds W[l][k].  D: lane l, register i = out[row i][feature l].  A layer is K MFMAs of one wave;
//     a lane keeps ITS output feature's weight row in registers (64 - 66 per layer, read straight from the nn.Linear layout of
//     `params`: no pack kernel, no workspace);
//   * the hand-off to the next layer is a 4 x 4 transpose inside every lane quad (D layout: lane 4b + j, register i  ->  A layout:
//     lane 4b + i, register j): eight v_cndmask_b32_dpp, NO LDS, NO barrier.  Bias enters through the accumulator;
//   * drift chain and diffusion-net chain of a step are independent until the update: a workgroup is TWO waves on the same 4 rows -
//     wave 0 the drift MLP, wave 1 the net, Philox and g - which exchange (f | g, dW) through 6 KB of LDS behind ONE s_barrier per
//     step (double-buffered by step parity) and then both form y' = y + f h + g dW bit-identically.  2048 rows = 1024 waves = every
//     SIMD of the chip, one wave each.
//
// Training-mode saves follow snsde_mfma_kernel's / snsde_m4n_kernel's layouts (act_save slots, relu signs folded into the saved z,
// SRK: stage planes), so the 4-row-tile adjoints and the weight-gradient pass read them unchanged.  This file also holds the
// adjoints on the same wave groups: snsde_w4_euler_reverse_kernel and snsde_w4_srk_reverse_kernel (drift wave + net wave + two
// gradient waves per tile: every weight gradient accumulated in registers inside the adjoint, no delta planes) and the two
// reduction kernels of their per-tile blocks.
#pragma once
#include <utility>

#include "snsde_mfma_kernels.h"

namespace snsde_w4 {

using snsde_mfma::f32x4;
using snsde_mfma::fast_tanh;
using snsde_mfma::snsde_pack_signs;
using snsde_mfma::uoff;

struct W4Args {
    const float* params;
    const float* step_tab;
    const float* out_w;
    const float* y0;
    const float* dW;
    float* ys;
    float* traj;
    float* dW_out;
    float* act_save;
    const float* srk_tab;     // SRK: (N, 4, SNSDE_SRK_STRIDE) stage times t0 + {0, 1/4, 1/2, 1} h: [1] sin, [2] cos
    const float* dU;          // SRK: supplied I_k0 (with dW) or null
    float* dU_out;
    float* stage_save;        // SRK training: (3N + 1, 3, B, H)  H0 | H1 | H1_3
    const int32_t* row_out;
    int64_t row_offset;
    uint64_t seed;
    const uint64_t* seed_dev;
    int32_t B, N, T, no, geo, nsave;
    int32_t off_theta;
    int32_t w_in, b_in, k_in, t_in;       // linear_in: weight offset, bias offset, K (64 or 66), leading time columns (0 or 2)
    int32_t w_hid[3], b_hid[3];
    int32_t w_out, b_out;
    int32_t w_n0, b_n0, w_n1, b_n1;       // noise_y[.0] (64, 66) and noise_y.2 (64, 64)
};

// D = A(4 rows, column k: block ABID of the A register, broadcast) x B(W^T row k) + C
template <int ABID>
__device__ __forceinline__ f32x4 mfma_bk(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, ABID, 0); }

// out[4][lane] (+)= sum_{k < 64} xt[k % 4](block k / 4) * w[k]   - two accumulator chains
template <int... KB>
__device__ __forceinline__ void gemm64(const float (&xt)[4], const float* w, f32x4& c, f32x4& d, std::integer_sequence<int, KB...>) {
    ((c = mfma_bk<KB>(xt[0], w[4 * KB], c), d = mfma_bk<KB>(xt[1], w[4 * KB + 1], d), c = mfma_bk<KB>(xt[2], w[4 * KB + 2], c),
      d = mfma_bk<KB>(xt[3], w[4 * KB + 3], d)), ...);
}

// ... with the weight column parked in the wave's private LDS slice ([k / 4][lane][4] floats: every lane reads back the 16 bytes it
// wrote - conflict-free, no barrier; LDS as an extension of the register file, as snsde_m4n_kernel.h does)
template <int... KB>
__device__ __forceinline__ void gemm64_lds(const float (&xt)[4], const float* wl, f32x4& c, f32x4& d, std::integer_sequence<int, KB...>) {
    f32x4 w[16];
    ((w[KB] = *reinterpret_cast<const f32x4*>(wl + KB * 256)), ...);
    ((c = mfma_bk<KB>(xt[0], w[KB][0], c), d = mfma_bk<KB>(xt[1], w[KB][1], d), c = mfma_bk<KB>(xt[2], w[KB][2], c),
      d = mfma_bk<KB>(xt[3], w[KB][3], d)), ...);
}

// 4 x 4 transpose inside every lane quad: in: lane 4b + j, v[i] = X[i][4b + j]  ->  out: lane 4b + i, t[j] = X[i][4b + j].
// Two butterfly stages of four v_cndmask_b32_dpp each (D = VCC ? src1 : dpp(src0)); the four lane masks arrive in SGPR pairs.
// Inline asm on purpose: written as selects around __builtin_amdgcn_update_dpp, hipcc sinks the DPP moves INTO the select's
// EXEC-masked region, where their source lanes are inactive and read as zero (tools/ubench/w4_probe.hip shows both forms).
__device__ __forceinline__ void quad_transpose(const float (&v)[4], float (&t)[4]) {
    float a0, a1, a2, a3;
    asm volatile(
        "s_mov_b64 vcc, %12\n\t"                                                                   // even lanes
        "s_nop 1\n\t"                                                                              // (VALU write -> DPP read of v[])
        "v_cndmask_b32_dpp %0, %9, %8, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"     // a0 = even ? v0 : v1[lane ^ 1]
        "v_cndmask_b32_dpp %2, %11, %10, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"   // a2 = even ? v2 : v3[lane ^ 1]
        "s_mov_b64 vcc, %13\n\t"                                                                   // odd lanes
        "v_cndmask_b32_dpp %1, %8, %9, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"     // a1 = odd ? v1 : v0[lane ^ 1]
        "v_cndmask_b32_dpp %3, %10, %11, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"   // a3 = odd ? v3 : v2[lane ^ 1]
        "s_mov_b64 vcc, %14\n\t"                                                                   // lanes 0, 1 of every quad
        "v_cndmask_b32_dpp %4, %2, %0, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"     // t0 = lo ? a0 : a2[lane ^ 2]
        "s_nop 0\n\t"
        "v_cndmask_b32_dpp %5, %3, %1, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"     // t1 = lo ? a1 : a3[lane ^ 2]
        "s_mov_b64 vcc, %15\n\t"                                                                   // lanes 2, 3
        "v_cndmask_b32_dpp %6, %0, %2, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"     // t2 = hi ? a2 : a0[lane ^ 2]
        "v_cndmask_b32_dpp %7, %1, %3, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"     // t3 = hi ? a3 : a1[lane ^ 2]
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3])
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "s"(0x5555555555555555ull), "s"(0xaaaaaaaaaaaaaaaaull), "s"(0x3333333333333333ull),
          "s"(0xccccccccccccccccull)
        : "vcc");
}

// LDS hand-off barrier: the exchange buffers are the only memory the waves share, so only the LDS queue is drained.  __syncthreads()
// also waits for vmcnt(0) - every global store of the step (outputs, trajectory, training saves) would be waited for at the next
// step's barrier: +25 us per 71-step solve with an output per step.
__device__ __forceinline__ void pair_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// fast_tanh (snsde_mfma_kernels.h: same polynomial / exponential branches, same operation order => the same bits) on a PAIR of values:
// the multiplies and fmas as v_pk_mul_f32 / v_pk_fma_f32 (two lanes' worth per issue slot); exp2 / rcp stay per element.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fast_tanh2(f32x2 x) {
    const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
    const f32x2 x2 = x * x;
    f32x2 p = x2 * -0.0088632355299021967f + 0.021869488536155203f;
    p = x2 * p + -0.053968253968253971f;
    p = x2 * p + 0.13333333333333333f;
    p = x2 * p + -0.33333333333333333f;
    p = (x * x2) * p + x;
    const f32x2 e = ax * -2.8853900817779268f;
    const f32x2 t = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
    const f32x2 den = t + 1.0f;
    const f32x2 q = (1.0f - t) * f32x2{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
    return f32x2{ax[0] < 0.25f ? p[0] : copysignf(q[0], x[0]), ax[1] < 0.25f ? p[1] : copysignf(q[1], x[1])};
}
__device__ __forceinline__ void fast_tanh4(float (&x)[4]) {
    const f32x2 a = fast_tanh2(f32x2{x[0], x[1]}), b = fast_tanh2(f32x2{x[2], x[3]});
    x[0] = a[0]; x[1] = a[1]; x[2] = b[0]; x[3] = b[1];
}

template <int NHID_, int NN_, bool TIME_, bool SAVE_>
struct CfgW {
    static constexpr int NHID = NHID_, NN = NN_;
    static constexpr bool TIME = TIME_, SAVE = SAVE_;       // SAVE: training mode (act_save planes, relu signs in the saved z)
    static constexpr int H = 64, KIN = TIME ? 66 : 64;
    static constexpr int ZSLOT = NHID + 1;
};

// -DW4_TRACE (tools/w4_trace.py): s_memtime stamps at the phase boundaries of a step, summed per wave of workgroup 0 and left in the
// first floats of dW_out
#ifdef W4_TRACE
#define W4_T(i) { const long long t_ = __builtin_readcyclecounter(); tr_acc[i] += (float)(t_ - tr_last); tr_last = t_; }
#else
#define W4_T(i)
#endif

template <class CF>
__global__ void __launch_bounds__(256, 2) snsde_w4_euler_kernel(W4Args a) {
#ifdef W4_TRACE
    float tr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tr_last = __builtin_readcyclecounter();
#endif
    constexpr int H = 64, NHID = CF::NHID, NN = CF::NN;
    constexpr bool TIME = CF::TIME;
    using Seq = std::make_integer_sequence<int, 16>;
    // A workgroup = TWO wave pairs (two 4-row tiles): four waves land on the four SIMDs of a CU, one each, whatever the grid size
    // (two-wave workgroups were placed two-deep on some SIMDs at 512 workgroups: 2048 rows 148 us against 97 us at 1024 rows); the
    // pairs only share the s_barrier.
    __shared__ float xchg_all[2][2][3][4][H];       // [pair][step parity][f | g | dW][row][feature]
    __shared__ float zstash_all[2][2][4][4][H];     // [pair] Philox normals [block parity][row][step of the block][feature] (net wave only)

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = wv & 1, pair = wv >> 1;
    float (*xchg)[3][4][H] = xchg_all[pair];
    float (*zstash)[4][4][H] = zstash_all[pair];
    // Ragged batches: the last tile (and the idle second pair of an odd tile count) is moved BACK onto the last four rows instead of
    // masking rows - those rows are then solved twice, bit-identically (same global row => same inputs and Philox counters), and
    // stored twice with the same values; no per-row validity tests, branches or SGPR masks anywhere in the loop (B >= 4: host side).
    const int B = a.B;
    const int row_t = (blockIdx.x * 2 + pair) * 4;
    const int row0 = row_t + 4 <= B ? row_t : B - 4;
    const uint32_t BH = (uint32_t)B * H;
    const float* P = a.params;

    float y[4], yt[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = a.y0[(size_t)(row0 + i) * H + lane];
    quad_transpose(y, yt);

    const int n_steps = a.N;
    // step-table row: [0] t0, [1] h, [2] sin, [3] cos, [6] sqrt h, [8] outputs after the step, [9] index of the first
    struct Row { float h, sn, cs, sqh; int nout, kfirst; };
    // step table / output weights through the CONSTANT address space: wave-uniform reads of memory the kernel never writes become
    // scalar loads (s_load, lgkmcnt).  As plain global pointers hipcc cannot prove them invariant against the kernel's own stores and
    // uses vector loads - whose s_waitcnt vmcnt(0) then also waits for every output / trajectory store in flight (+ 770 cycles per
    // step with an output per step).
    typedef const float __attribute__((address_space(4)))* CP;
    const CP step_tab_c = (CP)(uintptr_t)a.step_tab, out_w_c = (CP)(uintptr_t)a.out_w;
    auto load_row = [&](int n) {
        CP st = step_tab_c + (size_t)n * SNSDE_STEP_STRIDE;
        Row q;
        q.h = st[1]; q.sn = st[2]; q.cs = st[3]; q.sqh = st[6];
        q.nout = __float_as_int(st[8]); q.kfirst = __float_as_int(st[9]);
        return q;
    };
    const uint32_t lo = (uint32_t)(row0 * H + lane);
    auto store4 = [&](float* p, const float (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) p[lo + (uint32_t)(i * H)] = v[i];
    };
    // The hidden activations' training saves as ONE 16-byte store per lane instead of four 4-byte ones, same (B, H) row-major layout:
    // in the transposed (A-operand) layout, which the next layer needs anyway, lane 4q + j holds row j, features 4q .. 4q + 3 - 16
    // contiguous bytes of row j.  A store instruction costs the wave ~37 issue cycles whatever its width; planes that exist in the D
    // layout only (z, q, the states) keep their four stores: the eight-DPP transpose costs more than it saves (measured: + 12 us).
    const uint32_t lot = (uint32_t)((row0 + (lane & 3)) * H + (lane >> 2) * 4);
    auto store4x = [&](float* p, const float (&t)[4]) { *reinterpret_cast<float4*>(p + lot) = float4{t[0], t[1], t[2], t[3]}; };
    auto save_x = [&](int n, int slot, const float (&t)[4]) { store4x(a.act_save + uoff(n, (uint32_t)a.nsave * BH, (uint32_t)slot, BH), t); };
    auto save = [&](int n, int slot, const float (&v)[4]) { store4(a.act_save + uoff(n, (uint32_t)a.nsave * BH, (uint32_t)slot, BH), v); };
    // relu epilogue of a hidden layer: v = relu(c + d) in the D layout (the bias came in through c), transposed into the next
    // layer's A layout; sign bit `bit` of every row's word
    auto relu_hand_off = [&](const f32x4& c, const f32x4& d, float (&v)[4], float (&vt)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(c[i] + d[i], 0.0f);
        quad_transpose(v, vt);
    };

    // Philox normals: one call = the four steps of a block for one (row, feature).  Step n refills row n & 3 for the NEXT block (one
    // call per step, by the net wave), through a wave-private LDS stash: ds traffic takes no VALU issue slots, a register stash costs
    // ~28 selects per step.  A burst of four calls every fourth step instead would stall the drift wave at that step's barrier.
    // (Measured and dropped: rows 0, 1 refilled by the drift wave - its 245 registers then spill; 2048 rows 113 -> 119 us.)
    const bool phx = a.dW == nullptr;
    const uint64_t seed = a.seed_dev ? *a.seed_dev : a.seed;
    auto philox_refill = [&](int n) {
        const int k = n & 3, blk = n >> 2;
        float zz[4];
        snsde_philox_normal4(seed, (uint32_t)(a.row_offset + row0 + k), (uint32_t)(blk + 1), (uint32_t)lane, zz, 0u);
#pragma unroll
        for (int e = 0; e < 4; ++e) zstash[(blk + 1) & 1][k][e][lane] = zz[e];
    };
    if (wave == 0) {
        // ================================ drift wave ================================
        // (parking the first matrix in LDS as the SRK kernel does in training mode: 112 -> 114 us here, nothing spills)
        float wi[CF::KIN], wh[NHID > 0 ? NHID : 1][H], wo[H], bi, bh[NHID > 0 ? NHID : 1], bo;
        {
            const float* w = P + a.w_in + (size_t)lane * CF::KIN;
#pragma unroll
            for (int k = 0; k < H; ++k) wi[k] = w[(TIME ? 2 : 0) + k];
            if constexpr (TIME) { wi[64] = w[0]; wi[65] = w[1]; }
            bi = P[a.b_in + lane];
#pragma unroll
            for (int l = 0; l < NHID; ++l) {
                const float* q = P + a.w_hid[l] + (size_t)lane * H;
#pragma unroll
                for (int k = 0; k < H; ++k) wh[l][k] = q[k];
                bh[l] = P[a.b_hid[l] + lane];
            }
            const float* q = P + a.w_out + (size_t)lane * H;
#pragma unroll
            for (int k = 0; k < H; ++k) wo[k] = q[k];
            bo = P[a.b_out + lane];
        }
        int rslot[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) rslot[i] = a.row_out ? a.row_out[row0 + i] : -1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            if (!a.row_out || rslot[i] == 0) a.ys[(size_t)(row0 + i) * H + lane] = y[i];
            if (a.traj) a.traj[(size_t)(row0 + i) * H + lane] = y[i];
        }
        const bool geo = a.geo != 0;
        Row nxt = load_row(0);
        for (int n = 0; n < n_steps; ++n) {
            const Row cur = nxt;
            nxt = load_row(n + 1 < n_steps ? n + 1 : n);
            const int kf = cur.kfirst < a.T - 1 ? (cur.kfirst < 0 ? 0 : cur.kfirst) : a.T - 2;
            const float ow0 = out_w_c[2 * kf], ow1 = out_w_c[2 * kf + 1];       // (consumed at the end of the step)
            uint32_t sgn[4] = {0u, 0u, 0u, 0u};
            float v[4], vt[4];
            W4_T(0)

            {   // first layer on [y | sin t, cos t]
                f32x4 c = {bi, bi, bi, bi}, d = {0.f, 0.f, 0.f, 0.f};
                gemm64(yt, wi, c, d, Seq{});
                if constexpr (TIME) {
                    c = mfma_bk<0>(cur.sn, wi[64], c);
                    d = mfma_bk<0>(cur.cs, wi[65], d);
                }
                relu_hand_off(c, d, v, vt);
                if constexpr (CF::SAVE) {
                    save_x(n, 0, vt);
#pragma unroll
                    for (int i = 0; i < 4; ++i) sgn[i] = v[i] > 0.0f ? 1u : 0u;
                }
            }
            W4_T(1)
#pragma unroll
            for (int l = 0; l < NHID; ++l) {
                f32x4 c = {bh[l], bh[l], bh[l], bh[l]}, d = {0.f, 0.f, 0.f, 0.f};
                gemm64(vt, wh[l], c, d, Seq{});
                relu_hand_off(c, d, v, vt);
                if constexpr (CF::SAVE) {
                    save_x(n, 1 + l, vt);
#pragma unroll
                    for (int i = 0; i < 4; ++i) sgn[i] |= (v[i] > 0.0f ? 1u : 0u) << (1 + l);
                }
            }
            W4_T(2)
            float f[4], z[4];
            {
                f32x4 c = {bo, bo, bo, bo}, d = {0.f, 0.f, 0.f, 0.f};
                gemm64(vt, wo, c, d, Seq{});
#pragma unroll
                for (int i = 0; i < 4; ++i) { z[i] = c[i] + d[i]; f[i] = z[i]; }
                W4_T(3)
                if (geo) {
                    float ty[4] = {y[0], y[1], y[2], y[3]};
                    fast_tanh4(ty);
#pragma unroll
                    for (int i = 0; i < 4; ++i) f[i] = z[i] * ty[i];
                }
                fast_tanh4(f);
            }
            W4_T(4)
            float* xp = &xchg[n & 1][0][0][0];
#pragma unroll
            for (int i = 0; i < 4; ++i) xp[i * H + lane] = f[i];
            if constexpr (CF::SAVE) {
                float zs[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) zs[i] = snsde_pack_signs(z[i], sgn[i], NHID + 1);
                save(n, CF::ZSLOT, zs);
            }
            pair_barrier();
            W4_T(5)
            float yn[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float g = xp[(4 + i) * H + lane], dw = xp[(8 + i) * H + lane];
                yn[i] = fmaf(g, dw, fmaf(f[i], cur.h, y[i]));
            }
            if (a.traj) store4(a.traj + uoff(n + 1, BH), yn);
            if (cur.nout > 0) {          // (wave-uniform; the first output's weights were fetched at the top of the step)
                auto emit = [&](int k, float w0, float w1) {
                    float o[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (w0 == 0.0f) ? yn[i] : snsde_interp_out(w0, w1, y[i], yn[i]);
                    if (!a.row_out) store4(a.ys + uoff(k + 1, BH), o);
                    else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (rslot[i] == k + 1) a.ys[lo + (uint32_t)(i * H)] = o[i];
                    }
                };
                emit(kf, ow0, ow1);
                for (int k = kf + 1; k < cur.kfirst + cur.nout; ++k) emit(k, out_w_c[2 * k], out_w_c[2 * k + 1]);   // (several outputs inside one step: rare)
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = yn[i];
            quad_transpose(y, yt);
            W4_T(6)
        }
#ifdef W4_TRACE
        if (blockIdx.x == 0 && pair == 0 && lane == 0 && a.dW_out) for (int i = 0; i < 8; ++i) a.dW_out[i] = tr_acc[i];
#endif
    } else {
        // ================================ diffusion-net wave ================================
        float wn0[66], wn1[NN > 1 ? H : 1], b0, b1 = 0.0f;
        {
            const float* w = P + a.w_n0 + (size_t)lane * 66;
#pragma unroll
            for (int k = 0; k < H; ++k) wn0[k] = w[2 + k];
            wn0[64] = w[0]; wn0[65] = w[1];
            b0 = P[a.b_n0 + lane];
            if constexpr (NN > 1) {
                const float* q = P + a.w_n1 + (size_t)lane * H;
#pragma unroll
                for (int k = 0; k < H; ++k) wn1[k] = q[k];
                b1 = P[a.b_n1 + lane];
            }
        }
        const float sig_theta = snsde_sigmoid(P[a.off_theta]);
        const bool mul_y = a.no == 15 || a.no == 19;
        if (phx) {       // block 0 (the drift wave's first read of the stash is behind step 0's barrier)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float zz[4];
                snsde_philox_normal4(seed, (uint32_t)(a.row_offset + row0 + i), 0u, (uint32_t)lane, zz, 0u);
#pragma unroll
                for (int e = 0; e < 4; ++e) zstash[0][i][e][lane] = zz[e];
            }
        }
        Row nxt = load_row(0);
        for (int n = 0; n < n_steps; ++n) {
            const Row cur = nxt;
            nxt = load_row(n + 1 < n_steps ? n + 1 : n);
            float dw[4];
            W4_T(0)
            if (phx) {
                const int k = n & 3, blk = n >> 2;
#pragma unroll
                for (int i = 0; i < 4; ++i) dw[i] = zstash[blk & 1][i][k][lane] * cur.sqh;
                philox_refill(n);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) dw[i] = (a.dW + uoff(n, BH))[lo + (uint32_t)(i * H)];
            }
            W4_T(1)
            float q[4], v[4], vt[4];
            {
                f32x4 c = {b0, b0, b0, b0}, d = {0.f, 0.f, 0.f, 0.f};
                gemm64(yt, wn0, c, d, Seq{});
                c = mfma_bk<0>(cur.sn, wn0[64], c);
                d = mfma_bk<0>(cur.cs, wn0[65], d);
                if constexpr (NN == 2) {
                    relu_hand_off(c, d, v, vt);
                    if constexpr (CF::SAVE) save_x(n, CF::ZSLOT + 1, vt);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) q[i] = c[i] + d[i];
                    if constexpr (CF::SAVE) save(n, CF::ZSLOT + 1, q);
                }
            }
            W4_T(2)
            if constexpr (NN == 2) {
                f32x4 c = {b1, b1, b1, b1}, d = {0.f, 0.f, 0.f, 0.f};
                gemm64(vt, wn1, c, d, Seq{});
#pragma unroll
                for (int i = 0; i < 4; ++i) q[i] = fmaxf(c[i] + d[i], 0.0f);
                if constexpr (CF::SAVE) save(n, CF::ZSLOT + 2, q);
            }
            W4_T(3)
            float g[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float raw = q[i] * (mul_y ? y[i] : 1.0f);      // (a select, not a branch per element; q * 1 is exact)
                g[i] = sig_theta * snsde_nan_to_num(raw);
            }
            fast_tanh4(g);
            W4_T(4)
            float* xp = &xchg[n & 1][0][0][0];
#pragma unroll
            for (int i = 0; i < 4; ++i) { xp[(4 + i) * H + lane] = g[i]; xp[(8 + i) * H + lane] = dw[i]; }
#ifndef W4_TRACE
            if (a.dW_out) store4(a.dW_out + uoff(n, BH), dw);
#endif
            pair_barrier();
            W4_T(5)
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = fmaf(g[i], dw[i], fmaf(xp[i * H + lane], cur.h, y[i]));
            quad_transpose(y, yt);
            W4_T(6)
        }
#ifdef W4_TRACE
        if (blockIdx.x == 0 && pair == 0 && lane == 0 && a.dW_out) for (int i = 0; i < 8; ++i) a.dW_out[64 + i] = tr_acc[i];
#endif
    }
}

// =====================================================================================================================================
// SRK (torchsde's SRID2; torch_ists' default method, nsde_model.py:63-74) on the same wave pair.  Per solver step three drift
// evaluations F0 = f(t0, y), F1 = f(t0 + h, H0_1), F2 = f(t0 + h/2, H0_2) - the drift wave - and four net evaluations G0 = g(t0, y),
// G1 = g(t0 + h/4, H1_1), G2 = g(t0 + h, H1_2), G3 = g(t0 + h/4, H1_3) - the net wave, which also draws (I_k, I_k0), forms every H1
// and the step's result and publishes it.  Five exchanges per step, each behind one s_barrier:
//     B1: F0 ->, <- G0, I_k0     B2: F1 ->, <- G1     B3: F2 ->     (the net wave: G2 beside F2, then G3 alone)     B4: <- y'
// (H0_1 needs F0 only; H0_2 needs F0, F1, G0, G1, I_k0; H1_2 needs F0, G0 only, so G1 and G2 do not wait for the drift wave.)
// Training-mode saves = snsde_m4n_kernel's (act_save per pass 3n + s with 2 NN + NHID + 2 slots, the pass's drift signs and the hidden
// signs of the net evaluation beside it / of the fourth evaluation in the low bits of the saved z, stage_save planes H0 | H1 | H1_3):
// snsde_m4n_rev_kernel.h + the weight-gradient pass read them unchanged; the wave-group adjoint below (snsde_w4_srk_reverse_kernel)
// reads the same planes.
// =====================================================================================================================================
template <int NHID, bool SAVE> __host__ __device__ constexpr int w4srk_fwd_lds_floats() { return 2 * 12 * 256 + 2 * 4096 + ((SAVE && NHID >= 1) ? 2 * 4096 : 0); }

template <class CF>
__global__ void __launch_bounds__(256, 2) snsde_w4_srk_kernel(W4Args a) {
    constexpr int H = 64, NHID = CF::NHID, NN = CF::NN;
    constexpr bool TIME = CF::TIME, SAVE = CF::SAVE;
    constexpr int NSAVE = NHID + 2 + 2 * NN, ZSLOT = NHID + 1, NP = 3;
    constexpr int SRK_BITS = NHID + 1 + (NN == 2 ? 2 : 0);
    using Seq = std::make_integer_sequence<int, 16>;
    // exchange planes of a pair: F0 F1 F2 | G0 G1 | I_k0 | y' | hidden-sign words of G0..G3 (training)
    enum { XF0 = 0, XF1, XF2, XG0, XG1, XDU, XY, XS0, XS1, XS2, XS3, XN };
    static_assert(XN <= 12, "w4srk_fwd_lds_floats");
    // dynamic LDS: exchange planes [pair][XN][4][H] | Philox stash [pair][stream: I_k, xi][block parity][row][step of the block][feature]
    // | training mode with a hidden drift layer: the drift wave's first matrix, parked (see snsde_w4_euler_kernel: 19 - 23 spilled
    // registers, seventeen scratch reloads per step each waiting behind the step's save stores - that, not the stores themselves, was
    // the "cost of the saves": 483 -> 40x us)
    constexpr bool PARK = SAVE && NHID >= 1;
    extern __shared__ __attribute__((aligned(16))) float w4srk_fwd_lds[];

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = wv & 1, pair = wv >> 1;
    float (*xchg)[4][H] = reinterpret_cast<float (*)[4][H]>(w4srk_fwd_lds + pair * XN * 256);
    float (*zstash)[2][4][4][H] = reinterpret_cast<float (*)[2][4][4][H]>(w4srk_fwd_lds + 2 * XN * 256 + pair * 4096);
    float* wpark = w4srk_fwd_lds + 2 * XN * 256 + 2 * 4096 + pair * 4096;      // [16][64][4]
    const int B = a.B;
    const int row_t = (blockIdx.x * 2 + pair) * 4;
    const int row0 = row_t + 4 <= B ? row_t : B - 4;       // (ragged tail: see snsde_w4_euler_kernel)
    const uint32_t BH = (uint32_t)B * H;
    const float* P = a.params;
    typedef const float __attribute__((address_space(4)))* CP;
    const CP step_tab_c = (CP)(uintptr_t)a.step_tab, out_w_c = (CP)(uintptr_t)a.out_w, srk_c = (CP)(uintptr_t)a.srk_tab;
    const uint32_t lo = (uint32_t)(row0 * H + lane);
    auto store4 = [&](float* p, const float (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) p[lo + (uint32_t)(i * H)] = v[i];
    };
    // 16-byte stores of the values that exist in the transposed layout anyway (see snsde_w4_euler_kernel): hidden activations and
    // the stage states (each is the operand of the evaluation it feeds)
    const uint32_t lot = (uint32_t)((row0 + (lane & 3)) * H + (lane >> 2) * 4);
    auto store4x = [&](float* p, const float (&t)[4]) { *reinterpret_cast<float4*>(p + lot) = float4{t[0], t[1], t[2], t[3]}; };
    auto save_x = [&](int pass, int slot, const float (&t)[4]) { store4x(a.act_save + uoff(pass, (uint32_t)NSAVE * BH, (uint32_t)slot, BH), t); };
    auto save = [&](int pass, int slot, const float (&v)[4]) { store4(a.act_save + uoff(pass, (uint32_t)NSAVE * BH, (uint32_t)slot, BH), v); };
    auto put = [&](int plane, const float (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) xchg[plane][i][lane] = v[i];
    };
    auto get = [&](int plane, float (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = xchg[plane][i][lane];
    };
    auto relu_hand_off = [&](const f32x4& c, const f32x4& d, float (&v)[4], float (&vt)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(c[i] + d[i], 0.0f);
        quad_transpose(v, vt);
    };

    float y[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = a.y0[(size_t)(row0 + i) * H + lane];
    const int n_steps = a.N;

    if (wave == 0) {
        // ================================ drift wave ================================
        float wi[PARK ? 1 : H], wt0 = 0.0f, wt1 = 0.0f, wh[NHID > 0 ? NHID : 1][H], wo[H], bi, bh[NHID > 0 ? NHID : 1], bo;
        const float* wil = wpark + lane * 4;
        {
            const float* w = P + a.w_in + (size_t)lane * CF::KIN;
#pragma unroll
            for (int k = 0; k < H; ++k) {
                if constexpr (PARK) wpark[(k >> 2) * 256 + lane * 4 + (k & 3)] = w[(TIME ? 2 : 0) + k];
                else wi[k] = w[(TIME ? 2 : 0) + k];
            }
            if constexpr (TIME) { wt0 = w[0]; wt1 = w[1]; }
            bi = P[a.b_in + lane];
#pragma unroll
            for (int l = 0; l < NHID; ++l) {
                const float* q = P + a.w_hid[l] + (size_t)lane * H;
#pragma unroll
                for (int k = 0; k < H; ++k) wh[l][k] = q[k];
                bh[l] = P[a.b_hid[l] + lane];
            }
            const float* q = P + a.w_out + (size_t)lane * H;
#pragma unroll
            for (int k = 0; k < H; ++k) wo[k] = q[k];
            bo = P[a.b_out + lane];
        }
        const bool geo = a.geo != 0;
        // one drift evaluation at (sn, cs) on the state `x` (D layout): f into `f`, training: act_save slots of `pass`, z (kept) and signs
        auto drift = [&](const float (&x)[4], float sn, float cs, int pass, float (&f)[4], float (&z)[4], uint32_t (&sgn)[4]) {
            float xt[4], v[4], vt[4];
            quad_transpose(x, xt);
            if constexpr (SAVE) { if (a.stage_save) store4x(a.stage_save + uoff(pass, NP * BH), xt); }      // H0 of the pass
            {
                f32x4 c = {bi, bi, bi, bi}, d = {0.f, 0.f, 0.f, 0.f};
                if constexpr (PARK) gemm64_lds(xt, wil, c, d, Seq{});
                else gemm64(xt, wi, c, d, Seq{});
                if constexpr (TIME) { c = mfma_bk<0>(sn, wt0, c); d = mfma_bk<0>(cs, wt1, d); }
                relu_hand_off(c, d, v, vt);
                if constexpr (SAVE) {
                    save_x(pass, 0, vt);
#pragma unroll
                    for (int i = 0; i < 4; ++i) sgn[i] = v[i] > 0.0f ? 1u : 0u;
                }
            }
#pragma unroll
            for (int l = 0; l < NHID; ++l) {
                f32x4 c = {bh[l], bh[l], bh[l], bh[l]}, d = {0.f, 0.f, 0.f, 0.f};
                gemm64(vt, wh[l], c, d, Seq{});
                relu_hand_off(c, d, v, vt);
                if constexpr (SAVE) {
                    save_x(pass, 1 + l, vt);
#pragma unroll
                    for (int i = 0; i < 4; ++i) sgn[i] |= (v[i] > 0.0f ? 1u : 0u) << (1 + l);
                }
            }
            f32x4 c = {bo, bo, bo, bo}, d = {0.f, 0.f, 0.f, 0.f};
            gemm64(vt, wo, c, d, Seq{});
#pragma unroll
            for (int i = 0; i < 4; ++i) { z[i] = c[i] + d[i]; f[i] = z[i]; }
            if (geo) {
                float tx[4] = {x[0], x[1], x[2], x[3]};
                fast_tanh4(tx);
#pragma unroll
                for (int i = 0; i < 4; ++i) f[i] = z[i] * tx[i];
            }
            fast_tanh4(f);
        };
        // training: the pass's saved z carries its drift signs and the hidden sign word(s) the net wave published
        auto save_z = [&](int pass, const float (&z)[4], const uint32_t (&sgn)[4], int plane_a, int plane_b) {
            float zs[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t sg = sgn[i];
                if constexpr (NN == 2) {
                    sg |= __float_as_uint(xchg[plane_a][i][lane]) << (NHID + 1);
                    if (plane_b >= 0) sg |= __float_as_uint(xchg[plane_b][i][lane]) << (NHID + 2);
                }
                zs[i] = snsde_pack_signs(z[i], sg, SRK_BITS);
            }
            save(pass, ZSLOT, zs);
        };
        for (int n = 0; n < n_steps; ++n) {
            CP st = step_tab_c + (size_t)n * SNSDE_STEP_STRIDE;
            CP sk = srk_c + (size_t)n * 4 * SNSDE_SRK_STRIDE;
            const float h = st[1];
            const float s0 = sk[1], c0 = sk[2], s1 = sk[3 * SNSDE_SRK_STRIDE + 1], c1 = sk[3 * SNSDE_SRK_STRIDE + 2],
                        s2 = sk[2 * SNSDE_SRK_STRIDE + 1], c2 = sk[2 * SNSDE_SRK_STRIDE + 2];
            const float rh = 1.0f / h;
            float f0[4], f1[4], f2[4], z[4], g0[4], g1[4], du[4], x[4];
            uint32_t sgn[4] = {0u, 0u, 0u, 0u};
            // ---- pass 0: F0 at (t0, y) ----
            drift(y, s0, c0, 3 * n, f0, z, sgn);
            put(XF0, f0);
            pair_barrier();                                   // B1
            get(XG0, g0); get(XDU, du);
            if constexpr (SAVE) save_z(3 * n, z, sgn, XS0, -1);
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = y[i] + f0[i] * h;                     // H0_1
            // ---- pass 1: F1 at (t0 + h, H0_1) ----
            drift(x, s1, c1, 3 * n + 1, f1, z, sgn);
            put(XF1, f1);
            pair_barrier();                                   // B2
            get(XG1, g1);
            if constexpr (SAVE) save_z(3 * n + 1, z, sgn, XS1, -1);
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = y[i] + 0.25f * f0[i] * h + 0.25f * f1[i] * h + (g0[i] + 0.5f * g1[i]) * (du[i] * rh);   // H0_2
            // ---- pass 2: F2 at (t0 + h/2, H0_2) ----
            drift(x, s2, c2, 3 * n + 2, f2, z, sgn);
            put(XF2, f2);
            pair_barrier();                                   // B3
            pair_barrier();                                   // B4: the net wave's G3, the step's result
            get(XY, y);
            if constexpr (SAVE) save_z(3 * n + 2, z, sgn, XS2, XS3);
        }
        if constexpr (SAVE) {      // the final state's planes (pass 3N): what a further step's first evaluations would have written
            if (a.stage_save) {
                float yt[4];
                quad_transpose(y, yt);
                store4x(a.stage_save + uoff(3 * n_steps, NP * BH), yt);
                store4x(a.stage_save + uoff(3 * n_steps, NP * BH, 1, BH), yt);
            }
        }
    } else {
        // ================================ diffusion-net wave ================================
        float wn0[66], wn1[NN > 1 ? H : 1], b0, b1 = 0.0f;
        {
            const float* w = P + a.w_n0 + (size_t)lane * 66;
#pragma unroll
            for (int k = 0; k < H; ++k) wn0[k] = w[2 + k];
            wn0[64] = w[0]; wn0[65] = w[1];
            b0 = P[a.b_n0 + lane];
            if constexpr (NN > 1) {
                const float* q = P + a.w_n1 + (size_t)lane * H;
#pragma unroll
                for (int k = 0; k < H; ++k) wn1[k] = q[k];
                b1 = P[a.b_n1 + lane];
            }
        }
        const float sig_theta = snsde_sigmoid(P[a.off_theta]);
        const bool mul_y = a.no == 15 || a.no == 19;
        const bool phx = a.dW == nullptr;
        const uint64_t seed = a.seed_dev ? *a.seed_dev : a.seed;
        int rslot[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) rslot[i] = a.row_out ? a.row_out[row0 + i] : -1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (!a.row_out || rslot[i] == 0) a.ys[(size_t)(row0 + i) * H + lane] = y[i];
            if (a.traj) a.traj[(size_t)(row0 + i) * H + lane] = y[i];
        }
        // Philox: stream 0 = the increments' normals, stream 1 = xi of the space-time Levy area; step n refills row n & 3 of the next block
        auto refill = [&](int row_k, int blk) {
#pragma unroll
            for (int sidx = 0; sidx < 2; ++sidx) {
                float zz[4];
                snsde_philox_normal4(seed, (uint32_t)(a.row_offset + row0 + row_k), (uint32_t)blk, (uint32_t)lane, zz, (uint32_t)sidx);
#pragma unroll
                for (int e = 0; e < 4; ++e) zstash[sidx][blk & 1][row_k][e][lane] = zz[e];
            }
        };
        if (phx) {
#pragma unroll
            for (int i = 0; i < 4; ++i) refill(i, 0);
        }
        // one net evaluation at (sn, cs) on the state x: g into `g`; training: the net slots of `pass` (slot0 / slot0 + 1) and the hidden
        // sign word into exchange plane `splane`
        auto net = [&](const float (&x)[4], float sn, float cs, int pass, int slot0, int splane, float (&g)[4]) {
            float xt[4], v[4], vt[4], q[4];
            quad_transpose(x, xt);
            if constexpr (SAVE) {      // H1 of the evaluation (plane 1; the fourth evaluation of a step: plane 2 of pass 3n + 2)
                if (a.stage_save) store4x(a.stage_save + uoff(pass, NP * BH, slot0 == ZSLOT + 1 ? 1u : 2u, BH), xt);
            }
            {
                f32x4 c = {b0, b0, b0, b0}, d = {0.f, 0.f, 0.f, 0.f};
                gemm64(xt, wn0, c, d, Seq{});
                c = mfma_bk<0>(sn, wn0[64], c);
                d = mfma_bk<0>(cs, wn0[65], d);
                if constexpr (NN == 2) {
                    relu_hand_off(c, d, v, vt);
                    if constexpr (SAVE) {
                        save_x(pass, slot0, vt);
#pragma unroll
                        for (int i = 0; i < 4; ++i) xchg[splane][i][lane] = __uint_as_float(v[i] > 0.0f ? 1u : 0u);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) q[i] = c[i] + d[i];
                    if constexpr (SAVE) save(pass, slot0, q);
                }
            }
            if constexpr (NN == 2) {
                f32x4 c = {b1, b1, b1, b1}, d = {0.f, 0.f, 0.f, 0.f};
                gemm64(vt, wn1, c, d, Seq{});
#pragma unroll
                for (int i = 0; i < 4; ++i) q[i] = fmaxf(c[i] + d[i], 0.0f);
                if constexpr (SAVE) save(pass, slot0 + 1, q);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float raw = q[i] * (mul_y ? x[i] : 1.0f);
                g[i] = sig_theta * snsde_nan_to_num(raw);
            }
            fast_tanh4(g);
        };
        for (int n = 0; n < n_steps; ++n) {
            CP st = step_tab_c + (size_t)n * SNSDE_STEP_STRIDE;
            CP sk = srk_c + (size_t)n * 4 * SNSDE_SRK_STRIDE;
            const float h = st[1], sqh = st[6];
            const int nout = __float_as_int(st[8]), kfirst = __float_as_int(st[9]);
            const float s0 = sk[1], c0 = sk[2], sq = sk[SNSDE_SRK_STRIDE + 1], cq = sk[SNSDE_SRK_STRIDE + 2],
                        s1 = sk[3 * SNSDE_SRK_STRIDE + 1], c1 = sk[3 * SNSDE_SRK_STRIDE + 2];
            const int kf = kfirst < a.T - 1 ? (kfirst < 0 ? 0 : kfirst) : a.T - 2;
            const float ow0 = out_w_c[2 * kf], ow1 = out_w_c[2 * kf + 1];
            const float rh = 1.0f / h, rsqh = 1.0f / sqh;
            float ik[4], ik0[4];
            if (phx) {
                const int k = n & 3, blk = n >> 2;
                const float sh12 = sqrtf(h / 12.0f);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ik[i] = zstash[0][blk & 1][i][k][lane] * sqh;
                    ik0[i] = h * fmaf(sh12, zstash[1][blk & 1][i][k][lane], 0.5f * ik[i]);
                }
                refill(k, blk + 1);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) { ik[i] = (a.dW + uoff(n, BH))[lo + (uint32_t)(i * H)]; ik0[i] = (a.dU + uoff(n, BH))[lo + (uint32_t)(i * H)]; }
            }
            float g0[4], g1[4], g2[4], g3[4], f0[4], f1[4], f2[4], x[4];
            // ---- G0 at (t0, y) ----
            net(y, s0, c0, 3 * n, ZSLOT + 1, XS0, g0);
            put(XG0, g0); put(XDU, ik0);
            pair_barrier();                                   // B1
            get(XF0, f0);
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = y[i] + 0.25f * f0[i] * h + SRK_B1_10 * g0[i] * sqh;      // H1_1
            // ---- G1 at (t0 + h/4, H1_1) ----
            net(x, sq, cq, 3 * n + 1, ZSLOT + 1, XS1, g1);
            put(XG1, g1);
            pair_barrier();                                   // B2
            get(XF1, f1);
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = y[i] + f0[i] * h + SRK_B1_20 * g0[i] * sqh;                     // H1_2
            // ---- G2 at (t0 + h, H1_2) ----
            net(x, s1, c1, 3 * n + 2, ZSLOT + 1, XS2, g2);
            // everything of the step's result that does not need F2 / G3, while the drift wave finishes F2 (same association as the
            // other kernels' `y + (f0 + f1) h/6 + f2 2h/3;  += w0 g0 + w1 g1 + w2 g2 + a4 g3`)
            float pa[4], ps[4], a4v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float ikk = 0.5f * (ik[i] * ik[i] - h);
                const float ikkk = (ik[i] * ik[i] * ik[i] - 3.0f * h * ik[i]) * (1.0f / 6.0f);
                const float a1 = ik[i], a2 = ikk * rsqh, a3 = ik0[i] * rh, a4 = ikkk * rh;
                const float w0 = srk_w0(a1, a2, a3, a4);
                const float w1 = srk_w1(a1, a2, a3, a4);
                const float w2 = srk_w2(a1, a2, a3, a4);
                pa[i] = y[i] + (f0[i] + f1[i]) * (h * (1.0f / 6.0f));
                ps[i] = w0 * g0[i] + w1 * g1[i] + w2 * g2[i];
                a4v[i] = a4;
            }
            pair_barrier();                                   // B3
            get(XF2, f2);
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = y[i] + 0.25f * f2[i] * h + (SRK_B1_30 * g0[i] + SRK_B1_31 * g1[i] + SRK_B1_32 * g2[i]) * sqh;   // H1_3
            // ---- G3 at (t0 + h/4, H1_3), then the step ----
            net(x, sq, cq, 3 * n + 2, ZSLOT + NN + 1, XS3, g3);
            float yn[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = pa[i] + f2[i] * (h * (2.0f / 3.0f));
                v += ps[i] + a4v[i] * g3[i];
                yn[i] = v;
            }
            put(XY, yn);
            pair_barrier();                                   // B4
            // (the state planes of pass 3n + 3 are written by the next step's first evaluations; after the last step: below)
            if (a.traj) store4(a.traj + uoff(n + 1, BH), yn);
            if (a.dW_out) store4(a.dW_out + uoff(n, BH), ik);
            if (a.dU_out) store4(a.dU_out + uoff(n, BH), ik0);
            if (nout > 0) {
                auto emit = [&](int k, float w0, float w1) {
                    float o[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (w0 == 0.0f) ? yn[i] : snsde_interp_out(w0, w1, y[i], yn[i]);
                    if (!a.row_out) store4(a.ys + uoff(k + 1, BH), o);
                    else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (rslot[i] == k + 1) a.ys[lo + (uint32_t)(i * H)] = o[i];
                    }
                };
                emit(kf, ow0, ow1);
                for (int k = kf + 1; k < kfirst + nout; ++k) emit(k, out_w_c[2 * k], out_w_c[2 * k + 1]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = yn[i];
        }
    }
}

// =====================================================================================================================================
// Adjoint of the Euler solve on the same wave pair (discretise-then-optimise, the contract of snsde_mfma_reverse_kernel: every a_n or
// dL/dy0 only, delta_save slots  dz | hidden deltas.. | first-layer delta | dq | net hidden delta  for the weight-gradient GEMMs, the
// per-tile partial sums of dL/d sigmoid(theta)).  Walking a step backwards is two independent transposed chains on the step's
// cotangent a_{n+1}:  drift  a h (1 - f^2){tanh y} -> W_out^T -> [h > 0] -> W_hid^T -> [z0 > 0] -> W_in,y^T   (drift wave: it owns
// the adjoint, adds the output gradients and publishes a), net  a dW (1 - g^2) sigma(theta){y}[q > 0] -> W2^T -> [hn > 0] -> W1,y^T
// (net wave, with the direct term of raw = q y); two exchanges per step.  A lane holds COLUMN k of a weight matrix (64 coalesced loads
// from `params`): out[i][k] = sum_l delta[i][l] W[l][k] is the same rank-1 MFMA with the reduction over the forward's output features.
// The relu masks of the drift chain are bits of the saved z (snsde_pack_signs), the net's hidden mask its saved activation.
// =====================================================================================================================================
struct W4RevArgs {
    const float* params;
    const float* step_tab;
    const float* out_w;
    const float* traj;
    const float* act;
    const float* dW;          // increments used by the forward, or null: regenerated from Philox (seed, row_offset)
    const float* grad_ys;
    float* adj;
    float* delta;
    float* dth_part;          // (tiles, 4): this kernel leaves the tile's sum in entry 0 and zeros in 1 .. 3
    float* gpart;             // FUSED: (tiles, w4g_block_floats) per-tile weight / bias gradient sums
    const int32_t* row_out;
    uint64_t seed;
    int64_t row_offset;
    int32_t B, N, T, no, geo, nsave, nslots, adj0_only, off_theta;
    int32_t w_in, k_in, t_in, w_hid[3], w_out, w_n0, w_n1;
};

// FUSED = the weight gradients inside the adjoint (round 5; VERDICT r4 item 4 in the form that fits this layout).  The tile adjoint
// writes one delta plane per layer and step (five at the K4 shape) which the weight-gradient GEMMs read back together with the saved
// activations: the adjoint is bound by that traffic (169 us for 450 MB), the GEMM launch takes another 130 us.  Here a tile is FOUR
// waves: besides the drift and the net wave two GRADIENT waves that hold  G[l][k] = sum_{n, row} delta[row][l] in[row][k]  of every
// layer in accumulator registers (64 per layer and lane).  The outer product of a row is the same rank-1 MFMA once more: A = the
// layer INPUT of the row in the D layout (four consecutive k of lane quad kq, broadcast by CBSZ / ABID = kq), B = the delta of the row
// in the D layout (lane l holds delta[row][l]) - plain registers, no staging: 16 MFMAs per row and layer, 64 per layer and step, as
// many as the layer's transposed GEMM, on SIMDs the pair does not use.  The drift / net waves publish their deltas to LDS (double
// buffered by step parity), the gradient waves walk one step behind, read the inputs (saved activations, states) straight from HBM
// in the D layout, keep the bias and time-column sums, and leave one block per tile for snsde_w4_grad_reduce_kernel.  No delta_save.
__host__ __device__ constexpr int w4g_layer_floats() { return 64 * 64 + 64; }                                // G^T [k][l] | bias [l]
__host__ __device__ constexpr int w4g_block_floats(int NHID, int NN) { return (NHID + 2 + NN) * w4g_layer_floats() + 256; }
__host__ __device__ constexpr int w4g_d_time(int NHID) { return (NHID + 2) * w4g_layer_floats(); }            // tsin | tcos of linear_in
__host__ __device__ constexpr int w4g_n_off(int NHID, int e) { return w4g_d_time(NHID) + 128 + e * w4g_layer_floats(); }
__host__ __device__ constexpr int w4g_n_time(int NHID, int NN) { return w4g_n_off(NHID, NN); }

template <int V> using IC = std::integral_constant<int, V>;

template <int ABID> __device__ __forceinline__ void outer4(const float (&in)[4], const float (&dl)[4], f32x4& acc) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = mfma_bk<ABID>(in[i], dl[i], acc);
}
template <int... KQ>
__device__ __forceinline__ void outer64(const float (&in)[4], const float (&dl)[4], f32x4 (&acc)[16], std::integer_sequence<int, KQ...>) {
    (outer4<KQ>(in, dl, acc[KQ]), ...);
}

template <class CF, bool FUSED>
__global__ void __launch_bounds__(FUSED ? 512 : 256, 2) snsde_w4_euler_reverse_kernel(W4RevArgs a) {
#ifdef W4_TRACE
    float tr_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tr_last = __builtin_readcyclecounter();
#endif
    constexpr int H = 64, NHID = CF::NHID, NN = CF::NN;
    constexpr int ZSLOT = NHID + 1, NB0 = NHID + 2, ND = NHID + 2;
    constexpr int WPT = FUSED ? 4 : 2;                 // waves per tile
    using Seq = std::make_integer_sequence<int, 16>;
    enum { XA = 0, XN, XCNT };
    __shared__ float xchg_all[2][XCNT][4][H];
    __shared__ float dpl_all[FUSED ? 2 : 1][FUSED ? 2 : 1][FUSED ? ND + NN : 1][4][H];      // [pair][step parity][delta plane][row][feature]
    // the drift wave's third matrix (W_in,y^T) is parked in LDS when the drift has a hidden layer: 192 resident weight registers
    // spilled (24 - 56 bytes per lane, reloaded inside the step loop)
    constexpr bool PARK = NHID >= 1;
    __shared__ __attribute__((aligned(16))) float wpark_all[PARK ? 2 : 1][PARK ? 16 : 1][H][4];
    __shared__ float zblk_all[2][4][4][H];      // regenerated Philox normals of the block of four steps being walked [pair][row][step][feature]

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // role (0 drift, 1 net, 2 / 3 the gradient waves) and tile of a wave.  FUSED: the dispatcher deals the eight waves of a workgroup
    // round the four SIMDs (wave w and w + 4 share one), so a heavy wave (drift, drift gradients: 192 MFMAs per step) is paired with a
    // light one of the OTHER tile (net, net gradients: 128): two drift waves on one SIMD cost 274 us per K4 adjoint against ...
    const int wave = FUSED ? ((0x11332200u >> (4 * wv)) & 3) : wv % WPT;
    const int pair = FUSED ? ((0x5Au >> wv) & 1) : wv / WPT;           // waves 0..7: D_A D_B GD_A GD_B | GN_B GN_A N_B N_A
    float (*xchg)[4][H] = xchg_all[pair];
    float (*zblk)[4][H] = zblk_all[pair];
    const int B = a.B;
    const int tile = blockIdx.x * 2 + pair;
    const int row_t = tile * 4;
    const int row0 = row_t + 4 <= B ? row_t : B - 4;       // ragged tail: moved back onto the last four rows (see the forward) ...
    const bool live = row_t < B;                           // (the idle second pair of an odd tile count repeats the last tile)
    float rowf[4];                                         // ... whose repeated rows must not enter the theta sum twice
#pragma unroll
    for (int i = 0; i < 4; ++i) rowf[i] = (live && row0 + i >= row_t) ? 1.0f : 0.0f;
    const uint32_t BH = (uint32_t)B * H;
    const uint32_t SBH = (uint32_t)a.nsave * BH, DBH = (uint32_t)a.nslots * BH;
    const float* P = a.params;
    typedef const float __attribute__((address_space(4)))* CP;
    const CP step_tab_c = (CP)(uintptr_t)a.step_tab, out_w_c = (CP)(uintptr_t)a.out_w;
    const uint32_t lo = (uint32_t)(row0 * H + lane);
    auto load4 = [&](const float* p, float (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = p[lo + (uint32_t)(i * H)];
    };
    auto store4 = [&](float* p, const float (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) p[lo + (uint32_t)(i * H)] = v[i];
    };
    auto put = [&](int plane, const float (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) xchg[plane][i][lane] = v[i];
    };
    auto get = [&](int plane, float (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = xchg[plane][i][lane];
    };
    // FUSED: a delta plane of step n for the gradient waves (rows repeated by a ragged tail are zeroed: they must not enter the sums twice)
    auto publish = [&](int n, int plane, const float (&v)[4]) {
        if constexpr (FUSED) {
#pragma unroll
            for (int i = 0; i < 4; ++i) dpl_all[pair][n & 1][plane][i][lane] = v[i] * rowf[i];
        }
    };
    // column `lane` of a (64, K) nn.Linear weight, columns c0 .. c0 + 63: w[l] = W[l][c0 + lane]
    auto load_col = [&](float (&w)[H], int off, int K, int c0) {
        const float* q = P + off + c0 + lane;
#pragma unroll
        for (int l = 0; l < H; ++l) w[l] = q[(size_t)l * K];
    };
    const int N = a.N;

    if (wave == 0) {
        // ================================ drift wave: owns the adjoint ================================
        float wo[H], wh[NHID > 0 ? NHID : 1][H], wi[PARK ? 1 : H];
        load_col(wo, a.w_out, H, 0);
#pragma unroll
        for (int l = 0; l < NHID; ++l) load_col(wh[l], a.w_hid[l], H, 0);
        const float* wil = &wpark_all[PARK ? pair : 0][0][lane][0];
        if constexpr (PARK) {
            const float* q = P + a.w_in + a.t_in + lane;
#pragma unroll
            for (int l = 0; l < H; ++l) wpark_all[pair][l >> 2][lane][l & 3] = q[(size_t)l * a.k_in];
        } else load_col(wi, a.w_in, a.k_in, a.t_in);
        const bool geo = a.geo != 0;
        // (register budget: 192 weight registers; the next step's y is fetched into `y` itself once the step is done with it and the
        //  per-row output gradient is re-read where an output is emitted - with both kept in registers the wave spilled ten of them)
        int rslot[4];
        float adj[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) rslot[i] = a.row_out ? a.row_out[row0 + i] : -1;
        // gradient of the first output emitted after step n, fetched a step ahead (an exposed load at the top of every step otherwise:
        // 2000 of a step's cycles); per-row outputs: the one (B, H) plane, resident
        float gpre[4] = {0.f, 0.f, 0.f, 0.f};
        // the step's scalars (h, its outputs and the first one's weights) likewise: a scalar-cache miss per step and a dependent one
        // behind it, waited for at the step's first barrier; fetched behind that barrier for the step below
        // (two stages, so that no load waits for another one inside the window: the table row two steps ahead, the output's weights
        //  and gradient - addressed by that row - one step ahead)
        float h_c, w0_c, w1_c, h_p = 0.0f;
        int nout_c, kf_c, nout_p = 0, kf_p = 0;
        auto load_row = [&](int n, float& h, int& nout, int& kf) {
            CP st = step_tab_c + (size_t)n * SNSDE_STEP_STRIDE;
            h = st[1]; nout = __float_as_int(st[8]); kf = __float_as_int(st[9]);
        };
        auto load_out = [&](int nout, int kf) {
            const int kc = kf < a.T - 1 ? (kf < 0 ? 0 : kf) : a.T - 2;
            w0_c = out_w_c[2 * kc]; w1_c = out_w_c[2 * kc + 1];
            if (!a.row_out && nout > 0) load4(a.grad_ys + uoff(kf + 1, BH), gpre);
        };
        auto step_scalars = [&](int n) {      // behind the first barrier of step n + 1: everything the top of step n reads
            h_c = h_p; nout_c = nout_p; kf_c = kf_p;
            load_out(nout_c, kf_c);
            if (n > 0) load_row(n - 1, h_p, nout_p, kf_p);
        };
        if (a.row_out) load4(a.grad_ys, gpre);
        load_row(N - 1, h_c, nout_c, kf_c);
        load_out(nout_c, kf_c);
        if (N > 1) load_row(N - 2, h_p, nout_p, kf_p);
        float y[4], z[4], zn[4];
        load4(a.traj + uoff(N - 1, BH), y);
        load4(a.act + uoff(N - 1, SBH, ZSLOT, BH), z);
        for (int n = N - 1; n >= 0; --n) {
            if (n > 0) load4(a.act + uoff(n - 1, SBH, ZSLOT, BH), zn);      // next step's z: a full step ahead of its use
            const float h = h_c, w0f = w0_c, w1f = w1_c;
            const int nout = nout_c, kfirst = kf_c;
            float carry[4] = {0.f, 0.f, 0.f, 0.f};
            for (int k = kfirst; k < kfirst + nout; ++k) {      // outputs emitted after step n: ys[k + 1] = y_{n+1} or w0 y_n + w1 y_{n+1}
                const float w0 = k == kfirst ? w0f : out_w_c[2 * k], w1 = k == kfirst ? w1f : out_w_c[2 * k + 1];
                float gk[4];
                if (a.row_out) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) gk[i] = rslot[i] == k + 1 ? gpre[i] : 0.0f;
                } else if (k == kfirst) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) gk[i] = gpre[i];
                } else load4(a.grad_ys + uoff(k + 1, BH), gk);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (w0 == 0.0f) adj[i] += gk[i];
                    else { adj[i] = fmaf(w1, gk[i], adj[i]); carry[i] = fmaf(w0, gk[i], carry[i]); }
                }
            }
            put(XA, adj);
            W4_T(0) pair_barrier(); W4_T(1)                                     // B1: the net wave takes a_{n+1}
            if (n > 0) step_scalars(n - 1);
            if (!a.adj0_only) store4(a.adj + uoff(n + 1, BH), adj);
            // dz = a h (1 - f^2) {tanh y};  direct y term of the gated drift
            float dz[4], ay[4], zc[4], f[4], ty[4] = {1.f, 1.f, 1.f, 1.f};
            uint32_t zb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                zb[i] = __float_as_uint(z[i]);
                zc[i] = __uint_as_float(zb[i] & ~((1u << (NHID + 1)) - 1u));
                f[i] = zc[i];
            }
            if (geo) {
#pragma unroll
                for (int i = 0; i < 4; ++i) ty[i] = y[i];
                fast_tanh4(ty);
#pragma unroll
                for (int i = 0; i < 4; ++i) f[i] = zc[i] * ty[i];
            }
            fast_tanh4(f);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float dzt = adj[i] * h * (1.0f - f[i] * f[i]);
                dz[i] = dzt * ty[i];
                ay[i] = geo ? fmaf(dzt * zc[i], 1.0f - ty[i] * ty[i], adj[i]) : adj[i];
            }
            if (n > 0) load4(a.traj + uoff(n - 1, BH), y);       // (y_n is not needed below)
            if (a.delta) store4(a.delta + uoff(n, DBH), dz);
            publish(n, 0, dz);
            float v[4], vt[4];
            quad_transpose(dz, vt);
            {   // W_out^T, masked by the last hidden layer's relu sign (act slot NHID)
                f32x4 c = {0.f, 0.f, 0.f, 0.f}, d = c;
                gemm64(vt, wo, c, d, Seq{});
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = ((zb[i] >> NHID) & 1u) ? c[i] + d[i] : 0.0f;
                if (a.delta) store4(a.delta + uoff(n, DBH, 1, BH), v);
                publish(n, 1, v);
                quad_transpose(v, vt);
            }
#pragma unroll
            for (int l = NHID - 1; l >= 0; --l) {   // W_hid[l]^T, masked by the sign of act slot l
                f32x4 c = {0.f, 0.f, 0.f, 0.f}, d = c;
                gemm64(vt, wh[l], c, d, Seq{});
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = ((zb[i] >> l) & 1u) ? c[i] + d[i] : 0.0f;
                if (a.delta) store4(a.delta + uoff(n, DBH, (uint32_t)(NHID - l + 1), BH), v);
                publish(n, NHID - l + 1, v);
                quad_transpose(v, vt);
            }
            float od[4];
            {   // W_in[:, y columns]^T
                f32x4 c = {0.f, 0.f, 0.f, 0.f}, d = c;
                if constexpr (PARK) gemm64_lds(vt, wil, c, d, Seq{});
                else gemm64(vt, wi, c, d, Seq{});
#pragma unroll
                for (int i = 0; i < 4; ++i) od[i] = c[i] + d[i];
            }
            W4_T(2) pair_barrier(); W4_T(3)                                     // B2: the net chain's share of a_n
            float on[4];
            get(XN, on);
#pragma unroll
            for (int i = 0; i < 4; ++i) adj[i] = (ay[i] + od[i] + carry[i]) + on[i];
#pragma unroll
            for (int i = 0; i < 4; ++i) z[i] = zn[i];
        }
        {   // ys[0] = y0
            float g0[4];
            if (a.row_out) {
#pragma unroll
                for (int i = 0; i < 4; ++i) g0[i] = rslot[i] == 0 ? gpre[i] : 0.0f;
            } else load4(a.grad_ys, g0);
#pragma unroll
            for (int i = 0; i < 4; ++i) adj[i] += g0[i];
            store4(a.adj, adj);
        }
        if (a.dth_part && live && lane < 3) a.dth_part[(size_t)tile * 4 + 1 + lane] = 0.0f;
#ifdef W4_TRACE
        if (blockIdx.x == 0 && pair == 0 && lane < 12 && !a.adj0_only) { float v = 0; for (int i = 0; i < 12; ++i) v = lane == i ? tr_acc[i] : v; a.adj[uoff(2, BH) + 0 * 64 + lane] = v; }
#endif
    } else if (wave == 1) {
        // ================================ diffusion-net wave ================================
        float w1t[NN > 1 ? H : 1], w0t[H];
        if constexpr (NN > 1) load_col(w1t, a.w_n1, H, 0);
        load_col(w0t, a.w_n0, 66, 2);
        const float sig_theta = snsde_sigmoid(P[a.off_theta]);
        const bool mul_y = a.no == 15 || a.no == 19;
        const bool phx = a.dW == nullptr;
        float th_acc = 0.0f;
        int zblk_id = -1;
        float y[4], q[4], hm[4], dw[4], yn[4], qn[4], hmn[4], dwn[4];
        auto fetch = [&](int n, float (&yy)[4], float (&qq)[4], float (&hh)[4], float (&ww)[4]) {
            load4(a.traj + uoff(n, BH), yy);
            load4(a.act + uoff(n, SBH, ZSLOT + NN, BH), qq);
            if constexpr (NN == 2) load4(a.act + uoff(n, SBH, ZSLOT + 1, BH), hh);
            if (!phx) load4(a.dW + uoff(n, BH), ww);
            else {
                if ((n >> 2) != zblk_id) {
                    zblk_id = n >> 2;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {      // (wave-private LDS: a register array indexed by n & 3 lands in scratch)
                        float zz[4];
                        snsde_philox_normal4(a.seed, (uint32_t)(a.row_offset + row0 + i), (uint32_t)zblk_id, (uint32_t)lane, zz);
#pragma unroll
                        for (int e = 0; e < 4; ++e) zblk[i][e][lane] = zz[e];
                    }
                }
                const float sqh = (step_tab_c + (size_t)n * SNSDE_STEP_STRIDE)[6];
#pragma unroll
                for (int i = 0; i < 4; ++i) ww[i] = zblk[i][n & 3][lane] * sqh;
            }
        };
        fetch(N - 1, y, q, hm, dw);
        for (int n = N - 1; n >= 0; --n) {
            if (n > 0) fetch(n - 1, yn, qn, hmn, dwn);
            // g and its factors do not need the adjoint: before the barrier
            float g[4], raw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { raw[i] = q[i] * (mul_y ? y[i] : 1.0f); g[i] = sig_theta * snsde_nan_to_num(raw[i]); }
            fast_tanh4(g);
            W4_T(0) pair_barrier(); W4_T(1)                                     // B1
            float av[4], dq[4], dir[4], vt[4], v[4];
            get(XA, av);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float om = 1.0f - g[i] * g[i];
                const bool fin = snsde_finite(raw[i]);
                const float dr = fin ? av[i] * dw[i] * om * sig_theta : 0.0f;
                float d = mul_y ? dr * y[i] : dr;
                if constexpr (NN == 2) d = q[i] > 0.0f ? d : 0.0f;
                dq[i] = d;
                dir[i] = (mul_y && fin) ? av[i] * om * (sig_theta * q[i]) * dw[i] : 0.0f;      // d(q y)/dy = q: the direct term
                th_acc = fmaf(av[i] * dw[i] * om * rowf[i], snsde_nan_to_num(raw[i]), th_acc);
            }
            if (a.delta) store4(a.delta + uoff(n, DBH, NB0, BH), dq);
            publish(n, ND, dq);
            quad_transpose(dq, vt);
            if constexpr (NN == 2) {
                f32x4 c = {0.f, 0.f, 0.f, 0.f}, d = c;
                gemm64(vt, w1t, c, d, Seq{});
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = hm[i] > 0.0f ? c[i] + d[i] : 0.0f;
                if (a.delta) store4(a.delta + uoff(n, DBH, NB0 + 1, BH), v);
                publish(n, ND + 1, v);
                quad_transpose(v, vt);
            }
            float on[4];
            {
                f32x4 c = {0.f, 0.f, 0.f, 0.f}, d = c;
                gemm64(vt, w0t, c, d, Seq{});
#pragma unroll
                for (int i = 0; i < 4; ++i) on[i] = (c[i] + d[i]) + dir[i];
            }
            put(XN, on);
            W4_T(2) pair_barrier(); W4_T(3)                                     // B2
#pragma unroll
            for (int i = 0; i < 4; ++i) { y[i] = yn[i]; q[i] = qn[i]; hm[i] = hmn[i]; dw[i] = dwn[i]; }
        }
        if (a.dth_part) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) th_acc += __shfl_down(th_acc, off, 64);
            if (lane == 0 && live) a.dth_part[(size_t)tile * 4] = th_acc;
        }
#ifdef W4_TRACE
        if (blockIdx.x == 0 && pair == 0 && lane < 12 && !a.adj0_only) { float v = 0; for (int i = 0; i < 12; ++i) v = lane == i ? tr_acc[i] : v; a.adj[uoff(2, BH) + 1 * 64 + lane] = v; }
#endif
    } else if constexpr (FUSED) {
        // ================================ gradient waves: wave 2 the drift layers, wave 3 the net's ================================
        const bool dside = wave == 2;
        constexpr int NLG = ND > NN ? ND : NN;               // layers a gradient wave accumulates (drift: ND, net: NN)
        f32x4 acc[NLG][16];
        float bacc[NLG], tsn = 0.0f, tcs = 0.0f;
#pragma unroll
        for (int g = 0; g < NLG; ++g) {
            bacc[g] = 0.0f;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[g][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // layer inputs of step n in the D layout: drift wave layer g reads act slot NHID - g (g <= NHID), the first layer the state y_n;
        // net: layer 0 of a two-layer net reads its hidden activation (slot ZSLOT + 1), the layer on [tau, y] the state
        float in_cur[NLG][4];
        const int nl = dside ? ND : NN, p0 = dside ? 0 : ND;
        auto fetch_in = [&](int n, auto gc) {
            constexpr int g = decltype(gc)::value;
            if (dside) {
                if constexpr (g <= NHID) load4(a.act + uoff(n, SBH, (uint32_t)(NHID - g), BH), in_cur[g]);
                else if constexpr (g < ND) load4(a.traj + uoff(n, BH), in_cur[g]);
            } else if constexpr (g < NN) {
                if constexpr (NN == 2 && g == 0) load4(a.act + uoff(n, SBH, ZSLOT + 1, BH), in_cur[0]);
                else load4(a.traj + uoff(n, BH), in_cur[NN - 1]);
            }
        };
        // layer g of step m from in_cur; its input of step m - 1 is fetched in place as soon as the layer's MFMAs are issued
        auto layer = [&](auto gc, int m, float sn, float cs) {
            constexpr int g = decltype(gc)::value;
            if (g < nl) {
                float dl[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) dl[i] = dpl_all[pair][m & 1][p0 + g][i][lane];
                outer64(in_cur[g], dl, acc[g], Seq{});
                if (m > 0) fetch_in(m - 1, gc);
                const float sum = (dl[0] + dl[1]) + (dl[2] + dl[3]);
                bacc[g] += sum;
                if (g == nl - 1) { tsn = fmaf(sum, sn, tsn); tcs = fmaf(sum, cs, tcs); }      // the layer on [tau, state]
            }
        };
        auto accumulate = [&](int m) {
            CP st = step_tab_c + (size_t)m * SNSDE_STEP_STRIDE;
            const float sn = st[2], cs = st[3];
            layer(IC<0>{}, m, sn, cs);
            if constexpr (NLG > 1) layer(IC<1>{}, m, sn, cs);
            if constexpr (NLG > 2) layer(IC<2>{}, m, sn, cs);
            if constexpr (NLG > 3) layer(IC<3>{}, m, sn, cs);
        };
        fetch_in(N - 1, IC<0>{});
        if constexpr (NLG > 1) fetch_in(N - 1, IC<1>{});
        if constexpr (NLG > 2) fetch_in(N - 1, IC<2>{});
        if constexpr (NLG > 3) fetch_in(N - 1, IC<3>{});
        // one step behind the chains: the sums of step n + 1 (its planes complete since B2 of that step, the other parity) while the
        // drift / net waves walk step n
        for (int n = N - 1; n >= -1; --n) {                     // (one call site: the accumulators stay in registers)
            if (n >= 0) { W4_T(0) pair_barrier(); W4_T(1) }                     // B1
            if (n < N - 1) accumulate(n + 1);
            if (n >= 0) { W4_T(2) pair_barrier(); W4_T(3) }                     // B2
        }
#ifdef W4_TRACE
        if (blockIdx.x == 0 && pair == 0 && lane < 12 && !a.adj0_only) { float v = 0; for (int i = 0; i < 12; ++i) v = lane == i ? tr_acc[i] : v; a.adj[uoff(2, BH) + wave * 64 + lane] = v; }
#endif
        if (live && a.gpart) {
            float* blk = a.gpart + (size_t)tile * w4g_block_floats(NHID, NN);
#pragma unroll
            for (int g = 0; g < NLG; ++g) {
                if (g < nl) {
                    float* gp = blk + (dside ? g * w4g_layer_floats() : w4g_n_off(NHID, g));
#pragma unroll
                    for (int q = 0; q < 16; ++q)
#pragma unroll
                        for (int i = 0; i < 4; ++i) gp[(4 * q + i) * 64 + lane] = acc[g][q][i];       // G^T [k][l]
                    gp[4096 + lane] = bacc[g];
                }
            }
            float* tp = blk + (dside ? w4g_d_time(NHID) : w4g_n_time(NHID, NN));
            tp[lane] = tsn; tp[64 + lane] = tcs;
        }
    }
}

// =====================================================================================================================================
// Adjoint of the SRID2 solve on the wave groups, weight gradients included (the contract of snsde_m4n_srk_reverse_kernel + the
// weight-gradient pass in one launch; no delta planes).  A step walked backwards is four phases
//     G3 | G2 || drift 2 | G1 || drift 1 | G0 || drift 0
// (snsde_m4n_rev_kernel.h: stage dependencies and the glue of the cotangents Fbar_s, Gbar_e).  The drift wave keeps ybar and Fbar_s and
// runs the three drift chains, the net wave keeps Gbar_e and runs the four net chains; after every phase they exchange the chain
// results (hb = the net chain's cotangent of its input state, d = the drift chain's).  Neither wave re-derives the stage states: the
// forward saved them (stage_save planes H0_s | H1_e | H1_3), so F_s = tanh(z_s {tanh H0_s}) and G_e = g(q_e {H1_e}) need the saved
// pre-activations only (K4's model - no gate, raw = q - reads no state plane at all).  The relu masks are bits of the saved z: the
// drift wave publishes the net's four hidden-mask bits with the step's cotangent.  Every per-phase input is re-fetched IN PLACE for
// the step below right after its last use (a full step ahead, no second register set).  Two gradient waves per tile accumulate
// G[l][k] of every layer over the 3 drift passes / 4 net evaluations of every step (see the Euler kernel above): they work on step
// n + 1 - one unit (pass / evaluation) per phase - while the chains of step n run, the delta planes double buffered by step parity.
// =====================================================================================================================================
struct W4SrkRevArgs {
    const float* params;
    const float* step_tab;
    const float* srk_tab;
    const float* out_w;
    const float* act;         // (3N, nsave, B, H)
    const float* stage;       // (3N + 1, 3, B, H)
    const float* dW;          // I_k used by the forward
    const float* dU;          // I_k0
    const float* grad_ys;
    float* adj;
    float* dth_part;
    float* gpart;
    const int32_t* row_out;
    int32_t B, N, T, no, geo, nsave, adj0_only, off_theta;
    int32_t w_in, k_in, t_in, w_hid[3], w_out, w_n0, w_n1;
};

template <int NHID, int NN, bool MULY = false> __host__ __device__ constexpr int w4srk_rev_lds_floats() {
    return 2 * (6 * 256 + 2 * (3 * (NHID + 2) + 4 * NN) * 256 + (NHID >= 1 ? 16 * 256 : 0) + ((MULY && NN == 2) ? 16 * 256 : 0));
}

// (GEO: the drift is gated by tanh of its input state, input_option 5; MULY: raw = q * state, noise_option 15 / 19 - compile-time:
//  the state planes they need are twelve / sixteen resident registers the other models do not have room for)
template <int NHID_, int NN_, bool GEO_, bool MULY_> struct CfgSR { static constexpr int NHID = NHID_, NN = NN_; static constexpr bool GEO = GEO_, MULY = MULY_; };

template <class CF>
__global__ void __launch_bounds__(512, 2) snsde_w4_srk_reverse_kernel(W4SrkRevArgs a) {
#ifdef W4_TRACE
    float tr_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tr_last = __builtin_readcyclecounter();
#endif
    constexpr int H = 64, NHID = CF::NHID, NN = CF::NN;
    constexpr bool geo = CF::GEO, mul_y = CF::MULY;
    constexpr int ZSLOT = NHID + 1, ND = NHID + 2, NP = 3;
    constexpr int SRK_BITS = NHID + 1 + (NN == 2 ? 2 : 0);
    constexpr bool PARK = NHID >= 1;
    using Seq = std::make_integer_sequence<int, 16>;
    // dynamic LDS, per tile: exchange planes | delta planes [step parity][3 ND + 4 NN] | the drift wave's parked W_in,y^T
    enum { XA = 0, XM, XD, XH, XCNT = XH + 2 };            // (XH: two planes, alternating by phase)
    constexpr int NDP = 3 * ND + 4 * NN;
    // raw = q * state with a two-layer net: the net wave keeps the four state planes besides its two matrices and spilled 25
    // registers (scratch reloads in the step loop); its W1,y^T is parked like the drift wave's W_in,y^T
    constexpr bool NPARK = mul_y && NN == 2;
    constexpr int TILE_FLOATS = w4srk_rev_lds_floats<NHID, NN, mul_y>() / 2;
    static_assert(TILE_FLOATS >= XCNT * 256 + 2 * NDP * 256 + (PARK ? 16 * 256 : 0) + (NPARK ? 16 * 256 : 0), "LDS layout");
    extern __shared__ __attribute__((aligned(16))) float w4srk_lds[];

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = (0x11332200u >> (4 * wv)) & 3;      // role / tile of a wave: see snsde_w4_euler_reverse_kernel
    const int pair = (0x5Au >> wv) & 1;
    float* xchg = w4srk_lds + pair * TILE_FLOATS;        // [XCNT][4][64]
    float* dpl = xchg + XCNT * 256;                      // [2][NDP][4][64]
    float* wpark = dpl + 2 * NDP * 256;                  // [16][64][4]
    float* npark = wpark + (PARK ? 16 * 256 : 0);        // [16][64][4]
    const int B = a.B;
    const int tile = blockIdx.x * 2 + pair;
    const int row_t = tile * 4;
    const int row0 = row_t + 4 <= B ? row_t : B - 4;
    const bool live = row_t < B;
    float rowf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rowf[i] = (live && row0 + i >= row_t) ? 1.0f : 0.0f;
    const uint32_t BH = (uint32_t)B * H;
    const uint32_t SBH = (uint32_t)a.nsave * BH, PBH = (uint32_t)NP * BH;
    const float* P = a.params;
    typedef const float __attribute__((address_space(4)))* CP;
    const CP step_tab_c = (CP)(uintptr_t)a.step_tab, out_w_c = (CP)(uintptr_t)a.out_w, srk_c = (CP)(uintptr_t)a.srk_tab;
    const uint32_t lo = (uint32_t)(row0 * H + lane);
    auto load4 = [&](const float* p, float (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = p[lo + (uint32_t)(i * H)];
    };
    auto store4 = [&](float* p, const float (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) p[lo + (uint32_t)(i * H)] = v[i];
    };
    // the gradient waves read their layer inputs as ONE 16-byte load per lane (snsde_w4_euler_kernel: lane 4q + j takes row j, features
    // 4q .. 4q + 3); the values arrive in the transposed layout and are turned (eight DPP moves) where the unit uses them - not
    // behind the load, which is issued a unit ahead.  (The chain waves keep their four 4-byte loads: the transposes on their critical
    // path cost more than the load slots they free - adjoint 519 -> 544 us with them.)
    const uint32_t lot = (uint32_t)((row0 + (lane & 3)) * H + (lane >> 2) * 4);
    auto load4x = [&](const float* p, float (&t)[4]) {
        const float4 q4 = *reinterpret_cast<const float4*>(p + lot);
        t[0] = q4.x; t[1] = q4.y; t[2] = q4.z; t[3] = q4.w;
    };
    auto store4t = [&](float* p, const float (&v)[4]) {
        float t[4];
        quad_transpose(v, t);
        *reinterpret_cast<float4*>(p + lot) = float4{t[0], t[1], t[2], t[3]};
    };
    auto put = [&](int plane, const float (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) xchg[plane * 256 + i * 64 + lane] = v[i];
    };
    auto get = [&](int plane, float (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = xchg[plane * 256 + i * 64 + lane];
    };
    auto publish = [&](int n, int plane, const float (&v)[4]) {      // delta plane of step n for the gradient waves (ragged tail: zeroed)
#pragma unroll
        for (int i = 0; i < 4; ++i) dpl[((n & 1) * NDP + plane) * 256 + i * 64 + lane] = v[i] * rowf[i];
    };
    auto load_col = [&](float (&w)[H], int off, int K, int c0) {
        const float* q = P + off + c0 + lane;
#pragma unroll
        for (int l = 0; l < H; ++l) w[l] = q[(size_t)l * K];
    };
    const int N = a.N;

    if (wave == 0) {
        // ================================ drift wave: ybar, Fbar_s, the three drift chains ================================
        float wo[H], wh[NHID > 0 ? NHID : 1][H], wi[PARK ? 1 : H];
        load_col(wo, a.w_out, H, 0);
#pragma unroll
        for (int l = 0; l < NHID; ++l) load_col(wh[l], a.w_hid[l], H, 0);
        const float* wil = wpark + lane * 4;
        if constexpr (PARK) {
            const float* q = P + a.w_in + a.t_in + lane;
#pragma unroll
            for (int l = 0; l < H; ++l) wpark[(l >> 2) * 256 + lane * 4 + (l & 3)] = q[(size_t)l * a.k_in];
        } else load_col(wi, a.w_in, a.k_in, a.t_in);
        int rslot[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) rslot[i] = a.row_out ? a.row_out[row0 + i] : -1;
        float adj[4] = {0.f, 0.f, 0.f, 0.f};
        // gradient of the first output emitted after step n, fetched a step ahead (an exposed load at the top of every step otherwise:
        // 2000 of a step's cycles); per-row outputs: the one (B, H) plane, resident
        float gpre[4] = {0.f, 0.f, 0.f, 0.f};
        // the step's scalars (h, its outputs and the first one's weights) likewise: a scalar-cache miss per step and a dependent one
        // behind it, waited for at the step's first barrier; fetched behind that barrier for the step below
        // (two stages, so that no load waits for another one inside the window: the table row two steps ahead, the output's weights
        //  and gradient - addressed by that row - one step ahead)
        float h_c, w0_c, w1_c, h_p = 0.0f;
        int nout_c, kf_c, nout_p = 0, kf_p = 0;
        auto load_row = [&](int n, float& h, int& nout, int& kf) {
            CP st = step_tab_c + (size_t)n * SNSDE_STEP_STRIDE;
            h = st[1]; nout = __float_as_int(st[8]); kf = __float_as_int(st[9]);
        };
        auto load_out = [&](int nout, int kf) {
            const int kc = kf < a.T - 1 ? (kf < 0 ? 0 : kf) : a.T - 2;
            w0_c = out_w_c[2 * kc]; w1_c = out_w_c[2 * kc + 1];
            if (!a.row_out && nout > 0) load4(a.grad_ys + uoff(kf + 1, BH), gpre);
        };
        auto step_scalars = [&](int n) {      // behind the first barrier of step n + 1: everything the top of step n reads
            h_c = h_p; nout_c = nout_p; kf_c = kf_p;
            load_out(nout_c, kf_c);
            if (n > 0) load_row(n - 1, h_p, nout_p, kf_p);
        };
        if (a.row_out) load4(a.grad_ys, gpre);
        load_row(N - 1, h_c, nout_c, kf_c);
        load_out(nout_c, kf_c);
        if (N > 1) load_row(N - 2, h_p, nout_p, kf_p);
        float z[3][4], h0[geo ? 3 : 1][4];
        auto fetch = [&](int n, auto sc) {
            constexpr int s2 = decltype(sc)::value;
            load4(a.act + uoff(3 * n + s2, SBH, ZSLOT, BH), z[s2]);
            if constexpr (geo) load4(a.stage + uoff(3 * n + s2, PBH), h0[s2]);
        };
#pragma unroll
        for (int i = 0; i < 4; ++i) h0[0][i] = 0.0f;
        fetch(N - 1, IC<0>{}); fetch(N - 1, IC<1>{}); fetch(N - 1, IC<2>{});
        for (int n = N - 1; n >= 0; --n) {
            const float h = h_c, w0f = w0_c, w1f = w1_c;
            const int nout = nout_c, kfirst = kf_c;
            float carry[4] = {0.f, 0.f, 0.f, 0.f};
            for (int k = kfirst; k < kfirst + nout; ++k) {
                const float w0 = k == kfirst ? w0f : out_w_c[2 * k], w1 = k == kfirst ? w1f : out_w_c[2 * k + 1];
                float gk[4];
                if (a.row_out) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) gk[i] = rslot[i] == k + 1 ? gpre[i] : 0.0f;
                } else if (k == kfirst) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) gk[i] = gpre[i];
                } else load4(a.grad_ys + uoff(k + 1, BH), gk);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (w0 == 0.0f) adj[i] += gk[i];
                    else { adj[i] = fmaf(w1, gk[i], adj[i]); carry[i] = fmaf(w0, gk[i], carry[i]); }
                }
            }
            {   // the step's cotangent and the net's hidden-mask bits (bit e: evaluation e) for the net wave
                float msk[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    uint32_t m = 0;
                    if constexpr (NN == 2) {
                        const uint32_t b0 = __float_as_uint(z[0][i]), b1 = __float_as_uint(z[1][i]), b2 = __float_as_uint(z[2][i]);
                        m = ((b0 >> (NHID + 1)) & 1u) | (((b1 >> (NHID + 1)) & 1u) << 1) | (((b2 >> (NHID + 1)) & 3u) << 2);
                    }
                    msk[i] = __uint_as_float(m);
                }
                put(XA, adj); put(XM, msk);
            }
            W4_T(0) pair_barrier(); W4_T(1)                                     // B0: the net wave takes a_{n+1} and its masks
            if (n > 0) step_scalars(n - 1);
            if (!a.adj0_only) store4t(a.adj + uoff(n + 1, BH), adj);
            float yb[4], fb0[4], fb1[4], fb2[4], hb[4], dr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                yb[i] = carry[i] + adj[i];
                fb0[i] = adj[i] * (h * (1.0f / 6.0f)); fb1[i] = fb0[i]; fb2[i] = adj[i] * (h * (2.0f / 3.0f));
            }
            // drift pass s2 walked backwards: Fbar -> dz -> W_out^T -> [mask] -> W_hid^T -> [mask] -> W_in,y^T (+ the gate's direct term);
            // the pass's deltas go to the gradient waves, its saved z / state are re-fetched for step n - 1
            auto drift_pass = [&](auto sc, const float (&fb)[4], float (&out)[4]) {
                constexpr int s2 = decltype(sc)::value;
                uint32_t zb[4];
                float dz[4], dd[4], v[4], vt[4];
                {
                    float zc[4], ty[4], F[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        zb[i] = __float_as_uint(z[s2][i]);
                        zc[i] = __uint_as_float(zb[i] & ~((1u << SRK_BITS) - 1u));
                        ty[i] = 1.0f; F[i] = zc[i];
                    }
                    if constexpr (geo) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) ty[i] = h0[s2][i];
                        fast_tanh4(ty);
#pragma unroll
                        for (int i = 0; i < 4; ++i) F[i] = zc[i] * ty[i];
                    }
                    fast_tanh4(F);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float dzt = fb[i] * (1.0f - F[i] * F[i]);
                        dd[i] = geo ? dzt * zc[i] * (1.0f - ty[i] * ty[i]) : 0.0f;
                        dz[i] = dzt * ty[i];
                    }
                }
                if (n > 0) fetch(n - 1, sc);
                publish(n, s2 * ND, dz);
                quad_transpose(dz, vt);
                {
                    f32x4 c = {0.f, 0.f, 0.f, 0.f}, d = c;
                    gemm64(vt, wo, c, d, Seq{});
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = ((zb[i] >> NHID) & 1u) ? c[i] + d[i] : 0.0f;
                    publish(n, s2 * ND + 1, v);
                    quad_transpose(v, vt);
                }
#pragma unroll
                for (int l = NHID - 1; l >= 0; --l) {
                    f32x4 c = {0.f, 0.f, 0.f, 0.f}, d = c;
                    gemm64(vt, wh[l], c, d, Seq{});
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = ((zb[i] >> l) & 1u) ? c[i] + d[i] : 0.0f;
                    publish(n, s2 * ND + NHID - l + 1, v);
                    quad_transpose(v, vt);
                }
                f32x4 c = {0.f, 0.f, 0.f, 0.f}, d = c;
                if constexpr (PARK) gemm64_lds(vt, wil, c, d, Seq{});
                else gemm64(vt, wi, c, d, Seq{});
#pragma unroll
                for (int i = 0; i < 4; ++i) out[i] = (c[i] + d[i]) + dd[i];
            };
            W4_T(2) pair_barrier(); W4_T(3)                                     // B1: G3's chain is done
            get(XH + 1, hb);
#pragma unroll
            for (int i = 0; i < 4; ++i) { yb[i] += hb[i]; fb2[i] = fmaf(0.25f * h, hb[i], fb2[i]); }
            // ---- drift pass 2 beside G2 ----
            drift_pass(IC<2>{}, fb2, dr);
            put(XD, dr);
            W4_T(4) pair_barrier(); W4_T(5)                                     // B2
            get(XH, hb);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                yb[i] += hb[i]; fb0[i] = fmaf(h, hb[i], fb0[i]);
                yb[i] += dr[i]; fb0[i] = fmaf(0.25f * h, dr[i], fb0[i]); fb1[i] = fmaf(0.25f * h, dr[i], fb1[i]);
            }
            // ---- drift pass 1 beside G1 ----
            drift_pass(IC<1>{}, fb1, dr);
            W4_T(6) pair_barrier(); W4_T(7)                                     // B3
            get(XH + 1, hb);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                yb[i] += hb[i]; fb0[i] = fmaf(0.25f * h, hb[i], fb0[i]);
                yb[i] += dr[i]; fb0[i] = fmaf(h, dr[i], fb0[i]);
            }
            // ---- drift pass 0 beside G0 ----
            drift_pass(IC<0>{}, fb0, dr);
            W4_T(8) pair_barrier(); W4_T(9)                                     // B4
            get(XH, hb);
#pragma unroll
            for (int i = 0; i < 4; ++i) adj[i] = yb[i] + hb[i] + dr[i];
        }
        {
            float g0[4];
            if (a.row_out) {
#pragma unroll
                for (int i = 0; i < 4; ++i) g0[i] = rslot[i] == 0 ? gpre[i] : 0.0f;
            } else load4(a.grad_ys, g0);
#pragma unroll
            for (int i = 0; i < 4; ++i) adj[i] += g0[i];
            store4(a.adj, adj);
        }
        if (a.dth_part && live && lane < 3) a.dth_part[(size_t)tile * 4 + 1 + lane] = 0.0f;
#ifdef W4_TRACE
        if (blockIdx.x == 0 && pair == 0 && lane < 12 && !a.adj0_only) { float v = 0; for (int i = 0; i < 12; ++i) v = lane == i ? tr_acc[i] : v; a.adj[uoff(2, BH) + 0 * 64 + lane] = v; }
#endif
    } else if (wave == 1) {
        // ================================ net wave: Gbar_e, the four net chains ================================
        float w1t[NN > 1 ? H : 1], w0t[NPARK ? 1 : H];
        if constexpr (NN > 1) load_col(w1t, a.w_n1, H, 0);
        const float* w0l = npark + lane * 4;
        if constexpr (NPARK) {
            const float* qw = P + a.w_n0 + 2 + lane;
#pragma unroll
            for (int l = 0; l < H; ++l) npark[(l >> 2) * 256 + lane * 4 + (l & 3)] = qw[(size_t)l * 66];
        } else load_col(w0t, a.w_n0, 66, 2);
        const float sig_theta = snsde_sigmoid(P[a.off_theta]);
        float th_acc = 0.0f;
        float q[4][4], h1[mul_y ? 4 : 1][4], ik[4], ik0[4];
        auto fetch = [&](int n, auto ec) {
            constexpr int e = decltype(ec)::value;
            const int p = 3 * n + (e < 3 ? e : 2);
            load4(a.act + uoff(p, SBH, (uint32_t)(e < 3 ? ZSLOT + NN : ZSLOT + 2 * NN), BH), q[e]);
            if constexpr (mul_y) load4(a.stage + uoff(p, PBH, e < 3 ? 1u : 2u, BH), h1[e]);
        };
#pragma unroll
        for (int i = 0; i < 4; ++i) h1[0][i] = 1.0f;
        fetch(N - 1, IC<0>{}); fetch(N - 1, IC<1>{}); fetch(N - 1, IC<2>{}); fetch(N - 1, IC<3>{});
        load4(a.dW + uoff(N - 1, BH), ik);
        load4(a.dU + uoff(N - 1, BH), ik0);
        float h_c = (step_tab_c + (size_t)(N - 1) * SNSDE_STEP_STRIDE)[1], rdt_c = (step_tab_c + (size_t)(N - 1) * SNSDE_STEP_STRIDE)[6];
        for (int n = N - 1; n >= 0; --n) {
            const float h = h_c, rdt = rdt_c;
            const float rh = 1.0f / h, rrdt = 1.0f / rdt;
            float wg[4][4], ik0h[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float ikk = 0.5f * (ik[i] * ik[i] - h);
                const float ikkk = (ik[i] * ik[i] * ik[i] - 3.0f * h * ik[i]) * (1.0f / 6.0f);
                ik0h[i] = ik0[i] * rh;
                const float a1 = ik[i], a2 = ikk * rrdt, a3 = ik0h[i], a4 = ikkk * rh;
                wg[0][i] = srk_w0(a1, a2, a3, a4);
                wg[1][i] = srk_w1(a1, a2, a3, a4);
                wg[2][i] = srk_w2(a1, a2, a3, a4);
                wg[3][i] = a4;
            }
            if (n > 0) { load4(a.dW + uoff(n - 1, BH), ik); load4(a.dU + uoff(n - 1, BH), ik0); }
            W4_T(0) pair_barrier(); W4_T(1)                                     // B0
            if (n > 0) {      // (the scalars of the step below: see the drift wave)
                CP st = step_tab_c + (size_t)(n - 1) * SNSDE_STEP_STRIDE;
                h_c = st[1]; rdt_c = st[6];
            }
            float msk[4], gb[4][4], hb[4];
            {
                float av[4];
                get(XA, av); get(XM, msk);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < 4; ++i) gb[e][i] = wg[e][i] * av[i];
            }
            // net evaluation e walked backwards: Gbar_e -> qb -> [W2^T -> mask] -> W1,y^T (+ the direct term of raw = q H1_e; theta's
            // share); its deltas go to the gradient waves, its saved q / state are re-fetched for step n - 1
            auto net_eval = [&](auto ec, float (&out)[4]) {
                constexpr int e = decltype(ec)::value;
                float qb[4], nd[4], v[4], vt[4];
                {
                    float g[4], rc[4];
                    bool fin[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float raw = q[e][i] * h1[mul_y ? e : 0][i];
                        fin[i] = snsde_finite(raw);
                        rc[i] = snsde_nan_to_num(raw);
                        g[i] = sig_theta * rc[i];
                    }
                    fast_tanh4(g);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float om = 1.0f - g[i] * g[i];
                        const float rb = fin[i] ? gb[e][i] * om * sig_theta : 0.0f;
                        th_acc = fmaf(gb[e][i] * om * rowf[i], rc[i], th_acc);
                        nd[i] = mul_y ? rb * q[e][i] : 0.0f;
                        float t = rb * h1[mul_y ? e : 0][i];
                        if constexpr (NN == 2) t = q[e][i] > 0.0f ? t : 0.0f;
                        qb[i] = t;
                    }
                }
                if (n > 0) fetch(n - 1, ec);
                publish(n, 3 * ND + e * NN, qb);
                quad_transpose(qb, vt);
                if constexpr (NN == 2) {
                    f32x4 c = {0.f, 0.f, 0.f, 0.f}, d = c;
                    gemm64(vt, w1t, c, d, Seq{});
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = ((__float_as_uint(msk[i]) >> e) & 1u) ? c[i] + d[i] : 0.0f;
                    publish(n, 3 * ND + e * NN + 1, v);
                    quad_transpose(v, vt);
                }
                f32x4 c = {0.f, 0.f, 0.f, 0.f}, d = c;
                if constexpr (NPARK) gemm64_lds(vt, w0l, c, d, Seq{});
                else gemm64(vt, w0t, c, d, Seq{});
#pragma unroll
                for (int i = 0; i < 4; ++i) out[i] = (c[i] + d[i]) + nd[i];
            };
            // ---- G3 ----
            net_eval(IC<3>{}, hb);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                gb[0][i] = fmaf(SRK_B1_30 * rdt, hb[i], gb[0][i]); gb[1][i] = fmaf(SRK_B1_31 * rdt, hb[i], gb[1][i]); gb[2][i] = fmaf(SRK_B1_32 * rdt, hb[i], gb[2][i]);
            }
            put(XH + 1, hb);
            W4_T(2) pair_barrier(); W4_T(3)                                     // B1
            // ---- G2 beside drift pass 2 ----
            net_eval(IC<2>{}, hb);
#pragma unroll
            for (int i = 0; i < 4; ++i) gb[0][i] = fmaf(SRK_B1_20 * rdt, hb[i], gb[0][i]);
            put(XH, hb);
            W4_T(4) pair_barrier(); W4_T(5)                                     // B2
            {
                float dr[4];
                get(XD, dr);
#pragma unroll
                for (int i = 0; i < 4; ++i) { gb[0][i] = fmaf(ik0h[i], dr[i], gb[0][i]); gb[1][i] = fmaf(0.5f * ik0h[i], dr[i], gb[1][i]); }
            }
            // ---- G1 beside drift pass 1 ----
            net_eval(IC<1>{}, hb);
#pragma unroll
            for (int i = 0; i < 4; ++i) gb[0][i] = fmaf(SRK_B1_10 * rdt, hb[i], gb[0][i]);
            put(XH + 1, hb);
            W4_T(6) pair_barrier(); W4_T(7)                                     // B3
            // ---- G0 beside drift pass 0 ----
            net_eval(IC<0>{}, hb);
            put(XH, hb);
            W4_T(8) pair_barrier(); W4_T(9)                                     // B4
        }
        if (a.dth_part) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) th_acc += __shfl_down(th_acc, off, 64);
            if (lane == 0 && live) a.dth_part[(size_t)tile * 4] = th_acc;
        }
#ifdef W4_TRACE
        if (blockIdx.x == 0 && pair == 0 && lane < 12 && !a.adj0_only) { float v = 0; for (int i = 0; i < 12; ++i) v = lane == i ? tr_acc[i] : v; a.adj[uoff(2, BH) + 1 * 64 + lane] = v; }
#endif
    } else {
        // ================================ gradient waves: wave 2 the drift layers, wave 3 the net's ================================
        const bool dside = wave == 2;
        constexpr int NLG = ND > NN ? ND : NN;
        f32x4 acc[NLG][16];
        float bacc[NLG], tsn = 0.0f, tcs = 0.0f;
#pragma unroll
        for (int g = 0; g < NLG; ++g) {
            bacc[g] = 0.0f;
#pragma unroll
            for (int qq = 0; qq < 16; ++qq) acc[g][qq] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const int nl = dside ? ND : NN, nunits = dside ? 3 : 4;       // layers per unit; units (drift passes / net evaluations) per step
        float in_cur[NLG][4];      // (as loaded: transposed layout)
        // input of layer g of unit u of step m: drift pass u: act slots NHID - g, the layer on [tau, state] the pass's state
        // H0_u; net evaluation u: its hidden activation (two-layer nets), the layer on [tau, state] the state H1_u
        auto fetch_in = [&](int m, int u, auto gc) {
            constexpr int g = decltype(gc)::value;
            if (dside) {
                if constexpr (g <= NHID) load4x(a.act + uoff(3 * m + u, SBH, (uint32_t)(NHID - g), BH), in_cur[g]);
                else if constexpr (g < ND) load4x(a.stage + uoff(3 * m + u, PBH), in_cur[g]);
            } else if constexpr (g < NN) {
                const int p = 3 * m + (u < 3 ? u : 2);
                if constexpr (NN == 2 && g == 0) load4x(a.act + uoff(p, SBH, (uint32_t)(u < 3 ? ZSLOT + 1 : ZSLOT + NN + 1), BH), in_cur[0]);
                else load4x(a.stage + uoff(p, PBH, u < 3 ? 1u : 2u, BH), in_cur[NN - 1]);
            }
        };
        // the walk over (step, unit), units of a step in the order the chains finish them; in_cur holds the inputs of (pm, pu): a
        // layer's input of the unit after it is fetched in place as soon as the layer's MFMAs are issued
        int pm = N - 1, pu = nunits - 1;
        auto layer = [&](auto gc, int m2, int u2, float sn, float cs, int p0) {
            constexpr int g = decltype(gc)::value;
            if (g < nl) {
                float dl[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) dl[i] = dpl[((pm & 1) * NDP + p0 + g) * 256 + i * 64 + lane];
                float ind[4];
                quad_transpose(in_cur[g], ind);
                outer64(ind, dl, acc[g], Seq{});
                if (m2 >= 0) fetch_in(m2, u2, gc);
                const float sum = (dl[0] + dl[1]) + (dl[2] + dl[3]);
                bacc[g] += sum;
                if (g == nl - 1) { tsn = fmaf(sum, sn, tsn); tcs = fmaf(sum, cs, tcs); }
            }
        };
        auto unit = [&]() {
            const int m2 = pu > 0 ? pm : pm - 1, u2 = pu > 0 ? pu - 1 : nunits - 1;
            CP sk = srk_c + (size_t)pm * 4 * SNSDE_SRK_STRIDE;
            // stage time of the unit: drift passes at t0, t0 + h, t0 + h/2 (rows 0, 3, 2), net evaluations at t0, t0 + h/4, t0 + h, t0 + h/4
            const int c = dside ? (pu == 0 ? 0 : (pu == 1 ? 3 : 2)) : (pu == 0 ? 0 : (pu == 2 ? 3 : 1));
            const float sn = sk[c * SNSDE_SRK_STRIDE + 1], cs = sk[c * SNSDE_SRK_STRIDE + 2];
            const int p0 = dside ? pu * ND : 3 * ND + pu * NN;
            layer(IC<0>{}, m2, u2, sn, cs, p0);
            if constexpr (NLG > 1) layer(IC<1>{}, m2, u2, sn, cs, p0);
            if constexpr (NLG > 2) layer(IC<2>{}, m2, u2, sn, cs, p0);
            if constexpr (NLG > 3) layer(IC<3>{}, m2, u2, sn, cs, p0);
            pm = m2; pu = u2;
        };
        fetch_in(pm, pu, IC<0>{});
        if constexpr (NLG > 1) fetch_in(pm, pu, IC<1>{});
        if constexpr (NLG > 2) fetch_in(pm, pu, IC<2>{});
        if constexpr (NLG > 3) fetch_in(pm, pu, IC<3>{});
        for (int n = N - 1; n >= 0; --n) {
            const bool work = n < N - 1;                        // the units of step n + 1 (complete since its B4)
            W4_T(0) pair_barrier(); W4_T(1)                                     // B0
            if (work && !dside) unit();
            W4_T(2) pair_barrier(); W4_T(3)                                     // B1
            if (work) unit();
            W4_T(4) pair_barrier(); W4_T(5)                                     // B2
            if (work) unit();
            W4_T(6) pair_barrier(); W4_T(7)                                     // B3
            if (work) unit();
            W4_T(8) pair_barrier(); W4_T(9)                                     // B4
        }
        for (int u = 0; u < nunits; ++u) unit();               // step 0
#ifdef W4_TRACE
        W4_T(10)
#endif
#ifdef W4_TRACE
        if (blockIdx.x == 0 && pair == 0 && lane < 12 && !a.adj0_only) { float v = 0; for (int i = 0; i < 12; ++i) v = lane == i ? tr_acc[i] : v; a.adj[uoff(2, BH) + wave * 64 + lane] = v; }
#endif
        if (live && a.gpart) {
            float* blk = a.gpart + (size_t)tile * w4g_block_floats(NHID, NN);
#pragma unroll
            for (int g = 0; g < NLG; ++g) {
                if (g < nl) {
                    float* gp = blk + (dside ? g * w4g_layer_floats() : w4g_n_off(NHID, g));
#pragma unroll
                    for (int qq = 0; qq < 16; ++qq)
#pragma unroll
                        for (int i = 0; i < 4; ++i) gp[(4 * qq + i) * 64 + lane] = acc[g][qq][i];
                    gp[4096 + lane] = bacc[g];
                }
            }
            float* tp = blk + (dside ? w4g_d_time(NHID) : w4g_n_time(NHID, NN));
            tp[lane] = tsn; tp[64 + lane] = tcs;
        }
    }
}

// Sum of the per-tile gradient blocks (stage 1: tiles split over blockIdx.y, stage 2 assembles the flat gradient).
struct W4GSeg { int32_t src, dst, ld, col, kind; };      // kind 0: weight G^T [k][l] -> W[l][col + k]; 1: vector [l]; 2: column `col` of W
struct W4GReduce {
    const float* gpart; float* part2; float* grad; const float* dth_part; const float* params;
    int32_t tiles, block, nsplit, nseg, off_theta, n_dth;
    W4GSeg seg[16];
};
__global__ void __launch_bounds__(256) snsde_w4_grad_reduce1_kernel(W4GReduce r) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= r.block) return;
    const int per = (r.tiles + r.nsplit - 1) / r.nsplit, t0 = blockIdx.y * per, t1 = min(t0 + per, r.tiles);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int t = t0;
    for (; t + 3 < t1; t += 4) {        // four loads in flight, a fixed association: the result does not depend on the launch
        s0 += r.gpart[(size_t)t * r.block + e]; s1 += r.gpart[(size_t)(t + 1) * r.block + e];
        s2 += r.gpart[(size_t)(t + 2) * r.block + e]; s3 += r.gpart[(size_t)(t + 3) * r.block + e];
    }
    for (; t < t1; ++t) s0 += r.gpart[(size_t)t * r.block + e];
    r.part2[(size_t)blockIdx.y * r.block + e] = (s0 + s1) + (s2 + s3);
}
__global__ void __launch_bounds__(256) snsde_w4_grad_reduce2_kernel(W4GReduce r) {
    if (blockIdx.x == gridDim.x - 1) {      // last block: theta from the adjoint's per-tile sums of dL/d sigmoid(theta)
        __shared__ float red[256];
        float s = 0.0f;
        for (int i = threadIdx.x; i < r.n_dth; i += 256) s += r.dth_part[i];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
        if (threadIdx.x == 0) { const float sg = snsde_sigmoid(r.params[r.off_theta]); r.grad[r.off_theta] = red[0] * sg * (1.0f - sg); }
        return;
    }
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= r.block) return;
    float s = 0.0f;
    for (int k = 0; k < r.nsplit; ++k) s += r.part2[(size_t)k * r.block + e];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i >= r.nseg) break;
        const W4GSeg g = r.seg[i];
        const int len = g.kind == 0 ? 4096 : 64, q = e - g.src;
        if (q < 0 || q >= len) continue;
        if (g.kind == 0) r.grad[g.dst + (q & 63) * g.ld + g.col + (q >> 6)] = s;
        else if (g.kind == 1) r.grad[g.dst + q] = s;
        else r.grad[g.dst + q * g.ld + g.col] = s;
        return;
    }
}

}  // namespace snsde_w4
