// H = 256, 4-row tiles, PARTIAL WEIGHT RESIDENCY (round 6; VERDICT r3 - r5 "H = 256 partial residency"): eight waves of two
// 16-feature tiles instead of snsde_m4s_kernel.h's sixteen waves of one.
//
// Why: three 256 x 256 fp32 layers (786 KB) do not fit a CU, so snsde_m4s_kernel streams ALL of them every step, L2 -> LDS ring ->
// MFMA A operand; its step (7.7 - 8.6 us) sits on the stream (5.2 - 6.2 us for 768 KB at the ~62 B/clk a CU can pull),
// and with sixteen waves at 128 registers each nothing is left to keep weights in.  Eight waves get 256 registers each:
//   * every wave owns TWO tiles (32 features); per streamed layer and tile the FIRST RK = 4 of the 16 k-blocks stay in registers
//     for the whole solve (4 x 4 x 2 tiles x 3 layers = 96 VGPRs per wave, 192 KB per CU: a quarter of the model), the other
//     NS = 12 are streamed exactly as before (global_load_lds_dwordx4 into a per-wave, per-tile ring of R = 6 one-KB slots, read
//     back with ds_read_b128, a slot re-requested R blocks ahead - into the next layer / next step - as soon as it has been read);
//   * a layer starts on its resident k-blocks (32 MFMAs per wave with nothing to wait for but the B operands), which is when the
//     ring's refills of the previous layer's tail land; the stream never idles and carries 3/4 of the bytes;
//   * the B operands (activation rows, lanes 0 - 15, broadcast by blgp:4) are read once per k-block and feed both tiles' MFMAs:
//     half the LDS operand reads per MFMA of the one-tile kernel.
// The k order and the two accumulator chains per tile are the one-tile kernel's (blocks 0 .. 15 ascending; c: fragments 0, 2,
// d: 1, 3; reduce-scatter, then the bias), so the results are BIT-IDENTICAL to snsde_m4s_kernel's - which is how the GPU tests pin
// it (tests/test_gpu_parity.py::test_h256_two_tile_kernel_is_bit_identical_to_the_streamed_one) besides the oracle parity cases.
// Reference semantics: benchmark_classification/models_sde/neuralsde.py:295-307 (f, g), SURVEY.md A3-A6 (stepping).
#pragma once
#include "snsde_m4s_kernel.h"

namespace snsde_mfma {

template <int NHID_, int KUXT_, int SAVE_>
struct CfgS2 {
    static constexpr int H = 256, NHID = NHID_, KUXT = KUXT_;
    static constexpr bool SAVE = SAVE_ != 0;
    static constexpr int NW = 8, NT = 512, KUH = 16, TPW = 2;
    static constexpr int LDY = ld_for(16 * KUH, 16);
    static constexpr int LDX = ld_for(16 * (KUXT > 0 ? KUXT : 1), 16);
    static constexpr int LDA = LDY;
    static constexpr int NLAYER = NHID + 2, NSAVE = NHID + 2, ZSLOT = NHID + 1;
    static constexpr int ROWCH = 128, RS = 8;
    static constexpr int RK = 4;                              // resident k-blocks per streamed layer and tile
    static constexpr int NS = KUH - RK;                       // streamed k-blocks per layer and tile
    static constexpr int R = 6;                               // ring slots per tile (1 KB each)
    static constexpr int NSTR = NHID + 2;                     // layers: y, hidden.., out
    static constexpr int BIAS0 = 4 * (LDY + 2 * LDX + 2 * LDA) + (ROWCH + 3) * RS;   // [NLAYER][H] biases (read per layer: no registers)
    static constexpr int ZST0 = BIAS0 + NLAYER * H;           // [NW][2][4][64] Philox normals of four steps for the wave's two tiles
    static constexpr int RING0 = ZST0 + NW * TPW * 4 * 64;    // float offset of the rings (16-byte aligned: every term is a multiple of 4)
    static constexpr int LDS_FLOATS = RING0 + NW * TPW * R * 256;
    static_assert(NS % R == 0 && R % 2 == 0, "static ring slots: a layer's streamed blocks are whole turns of the ring");
};

// LDS-DMA of k-block SRC (static) of a tile's layer slice `sb` into the LDS bytes [ring_m0 + DST, + 1 KB).  Source = sb + vo4[SRC / 4]
// (lane * 16 + (SRC / 4) * 4096) + imm, imm = (SRC % 4) * 1024; the immediate also moves the LDS destination
// (tools/ubench/glds_probe.hip), so M0 = ring_m0 + DST - imm.  The M0 sum is formed INSIDE the asm: as C++ expressions the ~100
// distinct (base + constant) scalars of the step loop are loop invariants to hipcc, which hoists them all and spills ~150 SGPRs.
template <int SRC, int DST>
__device__ __forceinline__ void s2_refill(uint32_t ring_m0, const uint32_t (&vo4)[4], uint64_t sb) {
    constexpr int IMM = (SRC % 4) * 1024;
#ifdef SNSDE_S2_NO_STREAM      // development knock-out (build.py variant): no weight stream at all - stale ring contents, timing only
    return;
#endif
    asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4"
                 :: "s"(ring_m0), "v"(vo4[SRC / 4]), "s"(sb), "n"(DST - IMM), "n"(IMM) : "memory", "scc");
}

// c/d (+)= W[k-block] . b for one tile: the one-tile kernel's chain (c: fragments 0, 2; d: 1, 3)
#define SNSDE_S2_MFMA4(cc, dd, av, bv) \
    cc = __builtin_amdgcn_mfma_f32_4x4x1f32(av[0], bv[0], cc, 0, 0, 4); dd = __builtin_amdgcn_mfma_f32_4x4x1f32(av[1], bv[1], dd, 0, 0, 4); \
    cc = __builtin_amdgcn_mfma_f32_4x4x1f32(av[2], bv[2], cc, 0, 0, 4); dd = __builtin_amdgcn_mfma_f32_4x4x1f32(av[3], bv[3], dd, 0, 0, 4);

// One streamed k-block in flight: its B operand (lanes 0 - 15) and the two tiles' A fragments out of their rings
struct S2Set { f32x4 b, a0, a1; };

// issue the three reads of streamed block UB (k-block RK + UB): the B operand needs nothing, the A fragments need the block's ring
// slots landed - VM = refills that may stay in flight (the younger ones: vector-memory operations complete in order)
template <class CF, int UB, int BOFF, int VM>
__device__ __forceinline__ void s2_issue(S2Set& t, uint32_t baddr, uint32_t ra) {
    constexpr int SLOT = UB % CF::R;
    asm volatile("s_mov_b64 exec, 0xffff\n\tds_read_b128 %0, %[b] offset:%[o0]\n\ts_mov_b64 exec, -1\n\t"
                 "s_waitcnt vmcnt(%[vm])\n\t"
                 "ds_read_b128 %1, %[a] offset:%[s0]\n\tds_read_b128 %2, %[a] offset:%[s1]"
                 : "=&v"(t.b), "=&v"(t.a0), "=&v"(t.a1)
                 : [b] "v"(baddr), [a] "v"(ra), [o0] "n"(BOFF + (CF::RK + UB) * 64), [s0] "n"(SLOT * 1024), [s1] "n"(CF::R * 1024 + SLOT * 1024),
                   [vm] "n"(VM)
                 : "memory");
}
template <int CNT> __device__ __forceinline__ void s2_wait(S2Set& t) {
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(t.b), "+v"(t.a0), "+v"(t.a1) : "n"(CNT));
}

// consume streamed block UB: wait for its reads (LATER = LDS reads issued after them), re-request its two ring slots R blocks ahead
// (same layer from sb*, or - UB >= NS - R - the first streamed blocks of the layer the stream runs into, sn*), 8 MFMAs
template <class CF, int UB, int LATER>
__device__ __forceinline__ void s2_consume(S2Set& t, uint32_t ring_m0, const uint32_t (&vo4)[4], uint64_t sb0, uint64_t sb1, uint64_t sn0,
                                           uint64_t sn1, f32x4 (&c)[2], f32x4 (&d)[2]) {
    constexpr int SLOT = UB % CF::R, UN = (UB + CF::R) % CF::NS;
    constexpr bool SAME = UB + CF::R < CF::NS;
    // (the reads went out as b, a0, a1: tile 0's MFMAs start when b and a0 have landed, tile 1's fragment may still be in flight)
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(t.b), "+v"(t.a0) : "n"(LATER + 1));
    s2_refill<CF::RK + UN, SLOT * 1024>(ring_m0, vo4, SAME ? sb0 : sn0);
    SNSDE_S2_MFMA4(c[0], d[0], t.a0, t.b)
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(t.a1) : "n"(LATER));
    s2_refill<CF::RK + UN, CF::R * 1024 + SLOT * 1024>(ring_m0, vo4, SAME ? sb1 : sn1);
    SNSDE_S2_MFMA4(c[1], d[1], t.a1, t.b)
    __builtin_amdgcn_sched_barrier(0);
}

// One layer: resident k-blocks 0 .. RK - 1 from registers, then the NS streamed ones, software-pipelined one block deep: the
// reads of block i + 1 are in flight while block i's eight MFMAs issue (with two waves per SIMD there is no third wave to cover an
// exposed ds_read latency per block).  sb0 / sb1: this layer's slices of the wave's two tiles; sn0 / sn1: those of the layer the
// stream runs into (the next one, or the next step's first).
template <class CF, int BOFF>
__device__ __forceinline__ void s2_layer(const float (&wr)[2][CF::RK * 4], uint32_t baddr, uint32_t ra, uint32_t ring_m0,
                                         const uint32_t (&vo4)[4], uint64_t sb0, uint64_t sb1, uint64_t sn0, uint64_t sn1, f32x4 (&c)[2], f32x4 (&d)[2]) {
    static_assert(CF::RK == 4 && CF::NS == 12 && CF::R == 6, "schedule below");
    S2Set t0, t1;
    {
        LeanB<4> b;
        asm volatile("s_mov_b64 exec, 0xffff\n\tds_read_b128 %0, %[a] offset:%[o0]\n\tds_read_b128 %1, %[a] offset:%[o1]\n\t"
                     "ds_read_b128 %2, %[a] offset:%[o2]\n\tds_read_b128 %3, %[a] offset:%[o3]\n\ts_mov_b64 exec, -1"
                     : "=&v"(b.v[0]), "=&v"(b.v[1]), "=&v"(b.v[2]), "=&v"(b.v[3])
                     : [a] "v"(baddr), [o0] "n"(BOFF), [o1] "n"(BOFF + 64), [o2] "n"(BOFF + 128), [o3] "n"(BOFF + 192));
        // the first streamed block's reads go out behind them: of the 2R refills in flight only its own two must have landed
        s2_issue<CF, 0, BOFF, 2 * CF::R - 2>(t0, baddr, ra);
        asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(b.v[0]), "+v"(b.v[1]));
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                c[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[j][4 * u], b.v[u][0], c[j], 0, 0, 4);
                d[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[j][4 * u + 1], b.v[u][1], d[j], 0, 0, 4);
                c[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[j][4 * u + 2], b.v[u][2], c[j], 0, 0, 4);
                d[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[j][4 * u + 3], b.v[u][3], d[j], 0, 0, 4);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(b.v[2]), "+v"(b.v[3]));
#pragma unroll
        for (int u = 2; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                c[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[j][4 * u], b.v[u][0], c[j], 0, 0, 4);
                d[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[j][4 * u + 1], b.v[u][1], d[j], 0, 0, 4);
                c[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[j][4 * u + 2], b.v[u][2], c[j], 0, 0, 4);
                d[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[j][4 * u + 3], b.v[u][3], d[j], 0, 0, 4);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // streamed blocks: block i + 1's reads wait for its slots with the 2R - 4 younger refills left in flight (blocks i + 2 .. i + R - 1
    // of both tiles; block i's own slots are re-requested only after this)
#define SNSDE_S2_STEP(I, TA, TB) \
    s2_issue<CF, I + 1, BOFF, 2 * CF::R - 4>(TB, baddr, ra); \
    s2_consume<CF, I, 3>(TA, ring_m0, vo4, sb0, sb1, sn0, sn1, c, d);
    SNSDE_S2_STEP(0, t0, t1) SNSDE_S2_STEP(1, t1, t0) SNSDE_S2_STEP(2, t0, t1) SNSDE_S2_STEP(3, t1, t0)
    SNSDE_S2_STEP(4, t0, t1) SNSDE_S2_STEP(5, t1, t0) SNSDE_S2_STEP(6, t0, t1) SNSDE_S2_STEP(7, t1, t0)
    SNSDE_S2_STEP(8, t0, t1) SNSDE_S2_STEP(9, t1, t0) SNSDE_S2_STEP(10, t0, t1)
#undef SNSDE_S2_STEP
    s2_consume<CF, 11, 0>(t1, ring_m0, vo4, sb0, sb1, sn0, sn1, c, d);
}

template <class CF>
__global__ void __launch_bounds__(CF::NT, 1) snsde_m4s2_kernel(MfmaArgs a) {
    constexpr int H = CF::H, NT = CF::NT, NHID = CF::NHID, KUH = CF::KUH, KUXT = CF::KUXT, RK = CF::RK;
    constexpr int LDY = CF::LDY, LDX = CF::LDX, LDA = CF::LDA, RS = CF::RS;
    constexpr bool SAVE = CF::SAVE;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ybuf = lds;                       // [4][LDY]  y
    float* xbuf = ybuf + 4 * LDY;            // [2][4][LDX]  X(t) (xc) | sin t, cos t | 0..   (step parity)
    float* bufA = xbuf + 8 * LDX;            // [4][LDA]
    float* bufB = bufA + 4 * LDA;            // [4][LDA]
    float* rowtab = bufB + 4 * LDA;          // [ROWCH + 3][RS]
    float* lbias = lds + CF::BIAS0;          // [NLAYER][H]
    float* ring = lds + CF::RING0;           // [NW][2][R][256]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 3, s = (lane >> 2) & 3, q = lane >> 4;
    const int row0 = blockIdx.x * 4;
    const int B = a.B, C = a.C, N = a.N;
    const int row = row0 + r;
    const bool row_ok = row < B;
    const int rowc = row_ok ? row : B - 1;
    const size_t BH = (size_t)B * H;
    int fo[2];
    uint32_t fo4[2], goff4[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        fo[j] = (2 * wave + j) * 16 + 4 * q + s;
        fo4[j] = (uint32_t)(fo[j] * sizeof(float));
        goff4[j] = (uint32_t)(((size_t)rowc * H + fo[j]) * sizeof(float));
    }
    const int xc = a.lean_xc;
    const bool time_on = a.lean_time != 0, geo = a.lean_geo != 0;
    const int f_out = a.f_out;
    const bool g_raw = a.g_out == SNSDE_DIFFUSION_RAW;

    // ---- resident: the xt block, the first RK k-blocks of every streamed layer, the biases; streamed: SGPR bases per layer and tile
    int li = 0;
    float wxt[2][(KUXT > 0 ? KUXT : 1) * 4];
    if constexpr (KUXT > 0) {
        lean_load_w<KUXT>(wxt[0], a.ws + a.w_off[li], 2 * wave, lane);
        lean_load_w<KUXT>(wxt[1], a.ws + a.w_off[li], 2 * wave + 1, lane);
        ++li;
    }
    uint64_t sb[CF::NSTR][2];
    float wr[CF::NSTR][2][RK * 4];
#pragma unroll
    for (int l = 0; l < CF::NSTR; ++l) {
        const float* base = a.ws + a.w_off[li++];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float* sl = base + (size_t)(2 * wave + j) * KUH * 256;
            sb[l][j] = lean_uniform(sl);
#pragma unroll
            for (int u = 0; u < RK; ++u) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(sl + (size_t)u * 256 + lane * 4);
                wr[l][j][4 * u] = v[0]; wr[l][j][4 * u + 1] = v[1]; wr[l][j][4 * u + 2] = v[2]; wr[l][j][4 * u + 3] = v[3];
            }
        }
    }
    float* const zst = lds + CF::ZST0 + wave * (2 * 4 * 64) + lane;      // this lane's slots: [tile][k] at (tile * 4 + k) * 64
    const uint32_t ring_m0 = __builtin_amdgcn_readfirstlane(lean_lds_addr(ring) + (uint32_t)wave * (2 * CF::R * 1024));
    const uint32_t ra = ring_m0 + (uint32_t)lane * 16u;
    const uint32_t vo4[4] = {(uint32_t)lane * 16u, (uint32_t)lane * 16u + 4096u, (uint32_t)lane * 16u + 8192u, (uint32_t)lane * 16u + 12288u};

    for (int i = tid; i < 4 * (LDY + 2 * LDX + 2 * LDA); i += NT) lds[i] = 0.0f;
    for (int i = tid; i < CF::NLAYER * H; i += NT) lbias[i] = a.ws[a.bias_off + i];      // added after the k-slot reduction
    auto fill_rows = [&](int base) {
        for (int i = tid; i < (CF::ROWCH + 3) * RS; i += NT) {
            const int j = i % RS;
            int rr = base + i / RS + (j == 0 ? 0 : (j == 7 ? 2 : 1));
            rr = rr < N - 1 ? rr : N - 1;
            const int src = j == 0 ? 1 : j == 1 ? 6 : j == 4 ? (a.raw_time ? 0 : 2) : j == 5 ? (a.raw_time ? 10 : 3) : j == 6 ? 4 : j == 7 ? 5 : 10;
            rowtab[i] = a.step_tab[(size_t)rr * SNSDE_STEP_STRIDE + src];
        }
    };
    fill_rows(0);
    __syncthreads();

    const float sig_theta = snsde_sigmoid(a.params[a.off_theta]);
    const int no = a.no;
    const bool tab = a.gt_off >= 0;
    const float* gt = a.gt_ext ? a.gt_ext : a.ws + (tab ? a.gt_off : 0);
    const bool mul_y = (no == 13 || no == 17 || no == 3 || no == 6 || no == 11);
    const bool yfun = (no >= 7 && no <= 10);
    const bool mil = a.method == SNSDE_MILSTEIN;
    const bool phx = a.dW == nullptr;
    const uint32_t grow = (uint32_t)(a.row_offset + row);
    const uint64_t seed = a.seed_dev ? *a.seed_dev : a.seed;
    const int rslot = a.row_out ? a.row_out[rowc] : -1;

    float yv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        yv[j] = a.y0[(size_t)rowc * H + fo[j]];
        ybuf[r * LDY + fo[j]] = yv[j];
        if (row_ok) {
            a.ys[(size_t)row * H + fo[j]] = yv[j];
            if constexpr (SAVE) { if (a.traj) a.traj[(size_t)row * H + fo[j]] = yv[j]; }
        }
    }

    // ---- the [X(t) | sin t, cos t] entries of the tile: one entry per lane of the first 4 * xw lanes ------------------------
    const int xw = xc + (time_on ? 2 : 0);
    float ca, cb, cc, cd;
    const size_t cstride = (size_t)(a.L - 1) * 4 * C;
    const bool xok = KUXT > 0 && tid < 4 * xw;
    const int xrr = xok ? tid / xw : 0, xcol = xok ? tid - xrr * xw : 0;
    const int xdst = xok ? xrr * LDX + xcol : -1;
    const int xkind = xcol < xc ? 0 : (xcol == xc ? 1 : 2);
    const uint32_t cvo = (uint32_t)(((row0 + xrr < B ? xrr : B - 1 - row0) * cstride + (xcol < xc ? xcol : 0)) * sizeof(float));
    const bool has_x = KUXT > 0 && xc > 0;
    const float* ctile = a.coeffs + (size_t)row0 * cstride;
    const uint32_t cstep = (uint32_t)(C * sizeof(float));
    const uint32_t cidx = (uint32_t)(4 * C * sizeof(float));
    auto load_coeffs = [&](int idx) {
        if (__builtin_expect(has_x, 1)) {
            const uint32_t io = (uint32_t)idx * cidx;
            lean_gload4(ca, cb, cc, cd, cvo + io, cvo + io + cstep, cvo + io + 2 * cstep, cvo + io + 3 * cstep, ctile);
        }
    };
    auto vm_wait_all = [&](float& d0, float& d1, float& g0, float& g1) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(d0), "+v"(d1), "+v"(g0), "+v"(g1), "+v"(ca), "+v"(cb), "+v"(cc), "+v"(cd));
    };
    // the step's prefetches (coefficients, increments, table entries) are older than the 2R ring blocks that may still be in flight
    auto vm_wait = [&](float& d0, float& d1, float& g0, float& g1) {
        asm volatile("s_waitcnt vmcnt(%8)" : "+v"(d0), "+v"(d1), "+v"(g0), "+v"(g1), "+v"(ca), "+v"(cb), "+v"(cc), "+v"(cd) : "n"(2 * CF::R));
    };
    auto store_xt = [&](float* xb, float frac, float sn, float cs) {
        if constexpr (KUXT > 0) {
            float v = 0.0f;
            if (__builtin_expect(has_x, 1)) {
                const float x3 = cd * frac;
                float q3 = x3 * 0.333333343f;
                q3 = fmaf(fmaf(-3.0f, q3, x3), 0.333333343f, q3);
                v = ca + (cb + (0.5f * cc + q3) * frac) * frac;
            }
            v = xkind == 0 ? v : (xkind == 1 ? sn : cs);
            if (xdst >= 0) xb[xdst] = v;
        }
    };

    // Brownian increments of step i for the two owned elements: one Philox block per element and four steps, kept in registers
    // (same counters as every other kernel: bit-identical increments)
    // (the normals of four steps live in the wave's LDS stash: the 256-register budget goes to resident weights)
    auto next_dw = [&](int i, float sqh, float& out0, float& out1) {
        if (__builtin_expect(phx, 1)) {
            const int k = i & 3;
            if (k == 0) {
                float z0[4], z1[4];
                snsde_philox_normal4(seed, grow, (uint32_t)(i >> 2), (uint32_t)fo[0], z0);
                snsde_philox_normal4(seed, grow, (uint32_t)(i >> 2), (uint32_t)fo[1], z1);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) { zst[kk * 64] = z0[kk]; zst[(4 + kk) * 64] = z1[kk]; }
            }
            out0 = zst[k * 64] * sqh;
            out1 = zst[(4 + k) * 64] * sqh;
            return;
        }
        lean_gload(out0, goff4[0], a.dW + (size_t)i * BH);
        lean_gload(out1, goff4[1], a.dW + (size_t)i * BH);
    };

    auto gpart = [&](float y, float gtv, float dwv, float hh) -> float {
        float g = 0.0f, draw = 0.0f;
        if (__builtin_expect(yfun, 0)) {
            float p1, p2;
            const float raw = snsde_phi(no, y, p1, p2);
            g = fast_tanh(sig_theta * snsde_nan_to_num(raw));
            draw = snsde_finite(raw) ? p1 : 0.0f;
        } else {
            const float raw = mul_y ? gtv * y : gtv;
            if (__builtin_expect(g_raw, 0)) {
                float yp = fmaf(raw, dwv, y);
                if (mil && mul_y) yp = fmaf(0.5f * raw * gtv, fmaf(dwv, dwv, -hh), yp);
                return yp;
            }
            g = LEAN_TANH_G(sig_theta * raw);
            draw = (mul_y && snsde_finite(raw)) ? gtv : 0.0f;
        }
        float yp = fmaf(g, dwv, y);
        if (__builtin_expect(mil, 0)) yp = fmaf(0.5f * (g * ((1.0f - g * g) * sig_theta * draw)), fmaf(dwv, dwv, -hh), yp);
        return yp;
    };

    // ---- inputs of step 0; pieces of X(t_1) ---------------------------------------------------------------------------
    float dw_cur[2] = {0.f, 0.f}, gt_cur[2] = {0.f, 0.f};
    f32x2 qa;           // (h_n, sqrt h_{n+1})
    f32x4 qb;           // (sin, cos, frac of step n+1, idx of step n+2)
    ca = cb = cc = cd = 0.0f;
    {
        const float* g0 = a.step_tab;
        load_coeffs(__float_as_int(g0[5]));
        float du0 = 0.0f, du1 = 0.0f;
        vm_wait_all(du0, du1, gt_cur[0], gt_cur[1]);
        store_xt(xbuf, g0[4], a.raw_time ? g0[0] : g0[2], a.raw_time ? 0.0f : g0[3]);
        next_dw(0, g0[6], dw_cur[0], dw_cur[1]);
        if (tab) { lean_gload(gt_cur[0], fo4[0], gt); lean_gload(gt_cur[1], fo4[1], gt); }
        load_coeffs(__float_as_int(a.step_tab[(size_t)(N > 1 ? 1 : 0) * SNSDE_STEP_STRIDE + 5]));
        vm_wait_all(dw_cur[0], dw_cur[1], gt_cur[0], gt_cur[1]);
        qa = *reinterpret_cast<const f32x2*>(rowtab);
        qb = *reinterpret_cast<const f32x4*>(rowtab + 4);
    }
    __syncthreads();
    // the rings' first turns: streamed blocks 0 .. R - 1 (k-blocks RK ..) of layer 0, in consumption order (tile 0's pair, tile 1's pair)
    static_assert(CF::R == 6 && RK == 4, "initial fill below");
    // (in CONSUMPTION order - per block: tile 0's slot, tile 1's slot - which is what the vmcnt of the first layer's first block counts on)
#define SNSDE_S2_FILL(UB) \
    s2_refill<RK + UB, UB * 1024>(ring_m0, vo4, sb[0][0]); s2_refill<RK + UB, CF::R * 1024 + UB * 1024>(ring_m0, vo4, sb[0][1]);
    SNSDE_S2_FILL(0) SNSDE_S2_FILL(1) SNSDE_S2_FILL(2) SNSDE_S2_FILL(3) SNSDE_S2_FILL(4) SNSDE_S2_FILL(5)
#undef SNSDE_S2_FILL

    const uint32_t yrow = lean_lds_addr(ybuf + r * LDY + 4 * s);
    const uint32_t xrow = lean_lds_addr(xbuf + r * LDX + 4 * s);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    LeanB<(KUXT > 0 ? KUXT : 1)> bx{};
    if constexpr (KUXT > 0) lean_read_b_carried(xrow, bx);
    int n = 0;
    float yold[2] = {yv[0], yv[1]};
    for (int ko = 0; ko < a.T - 1; ++ko) {
    const int n_end = a.out_step[ko];
    for (; n <= n_end; ++n) {
        const int rbase = (n / CF::ROWCH) * CF::ROWCH;
        if (n > 0 && n == rbase) {
            fill_rows(rbase);
            __syncthreads();
        }
        const bool more = n + 1 < N;
        [[maybe_unused]] uint32_t sgn[2] = {0u, 0u};
        asm volatile("" : "+v"(qa), "+v"(qb));
        const float h = qa[0];
        // ---- first layer: the [X(t_n) | tau_n] part from registers, then the y part (resident blocks, streamed blocks) ------------
        f32x4 c[2] = {zero4, zero4}, d[2] = {zero4, zero4};
        if constexpr (KUXT > 0) { lean_gemm<15, KUXT>(wxt[0], bx, c[0], d[0]); lean_gemm<15, KUXT>(wxt[1], bx, c[1], d[1]); }
        float ypart[2];
        ypart[0] = gpart(yv[0], gt_cur[0], dw_cur[0], h);
        ypart[1] = gpart(yv[1], gt_cur[1], dw_cur[1], h);
        store_xt(xbuf + ((n + 1) & 1) * (4 * LDX), qb[2], qb[0], qb[1]);
        load_coeffs(__float_as_int(qb[3]));
        __builtin_amdgcn_sched_barrier(0);
        s2_layer<CF, 0>(wr[0], yrow, ra, ring_m0, vo4, sb[0][0], sb[0][1], sb[1][0], sb[1][1], c, d);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float pre = m4_reduce_scatter(c[j] + d[j]) + lbias[fo[j]];
            const float o = fmaxf(pre, 0.0f);
            bufA[r * LDA + fo[j]] = o;
            if constexpr (SAVE) {
                if (a.act_save && row_ok) lean_gstore(o, goff4[j], a.act_save + ((size_t)n * CF::NSAVE) * BH);
                sgn[j] = o > 0.0f ? 1u : 0u;      // relu signs of this lane's elements (snsde_pack_signs)
            }
        }
        __syncthreads();
        // ---- hidden layers; the next step's increments and diffusion-table entries are fetched in the first window ------------
        float dw_nxt[2] = {0.f, 0.f}, gt_nxt[2] = {0.f, 0.f};
        auto prep = [&]() {
            const int n1 = more ? n + 1 : n;
            next_dw(n1, qa[1], dw_nxt[0], dw_nxt[1]);
            if (__builtin_expect(tab, 1)) { lean_gload(gt_nxt[0], fo4[0], gt + (size_t)n1 * H); lean_gload(gt_nxt[1], fo4[1], gt + (size_t)n1 * H); }
        };
        constexpr int OFFA = (8 * LDX + 4 * LDY) * 4, OFFB = OFFA + 4 * LDA * 4;     // bufA / bufB rows from the y rows, bytes
#pragma unroll
        for (int l = 0; l < NHID; ++l) {
            const bool toB = (l % 2 == 0);
            if (l == 0) prep();
            __builtin_amdgcn_sched_barrier(0);
            c[0] = zero4; c[1] = zero4; d[0] = zero4; d[1] = zero4;
            if (toB) s2_layer<CF, OFFA>(wr[1 + l], yrow, ra, ring_m0, vo4, sb[1 + l][0], sb[1 + l][1], sb[2 + l][0], sb[2 + l][1], c, d);
            else s2_layer<CF, OFFB>(wr[1 + l], yrow, ra, ring_m0, vo4, sb[1 + l][0], sb[1 + l][1], sb[2 + l][0], sb[2 + l][1], c, d);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float pre = m4_reduce_scatter(c[j] + d[j]) + lbias[(1 + l) * H + fo[j]];
                const float o = fmaxf(pre, 0.0f);
                (toB ? bufB : bufA)[r * LDA + fo[j]] = o;
                if constexpr (SAVE) {
                    if (a.act_save && row_ok) lean_gstore(o, goff4[j], a.act_save + ((size_t)n * CF::NSAVE + 1 + l) * BH);
                    sgn[j] |= (o > 0.0f ? 1u : 0u) << (1 + l);
                }
            }
            __syncthreads();
        }
        // ---- output layer (its last chunks refill the rings with the NEXT step's first streamed blocks), f, update ----------------
        if (NHID == 0) prep();
        __builtin_amdgcn_sched_barrier(0);
        c[0] = zero4; c[1] = zero4; d[0] = zero4; d[1] = zero4;
        if (NHID % 2 == 0) s2_layer<CF, OFFA>(wr[NHID + 1], yrow, ra, ring_m0, vo4, sb[NHID + 1][0], sb[NHID + 1][1], sb[0][0], sb[0][1], c, d);
        else s2_layer<CF, OFFB>(wr[NHID + 1], yrow, ra, ring_m0, vo4, sb[NHID + 1][0], sb[NHID + 1][1], sb[0][0], sb[0][1], c, d);
        vm_wait(dw_nxt[0], dw_nxt[1], gt_nxt[0], gt_nxt[1]);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float z = m4_reduce_scatter(c[j] + d[j]) + lbias[(NHID + 1) * H + fo[j]];
            if constexpr (SAVE) {      // the saved pre-tanh drift carries the step's relu signs in its low NHID + 1 bits (the adjoint's masks)
                if (a.act_save && row_ok)
                    lean_gstore(snsde_pack_signs(z, sgn[j], NHID + 1), goff4[j], a.act_save + ((size_t)n * CF::NSAVE + CF::ZSLOT) * BH);
            }
            if (__builtin_expect(geo, 0)) z *= fast_tanh(yv[j]);
            float f;
            if (__builtin_expect(f_out != SNSDE_DRIFT_TANH, 0)) f = f_out == SNSDE_DRIFT_TIMES_Y ? z * yv[j] : z;
            else f = LEAN_TANH_F(z);
            const float ynew = fmaf(f, h, ypart[j]);
            yold[j] = yv[j];
            yv[j] = ynew;
            ybuf[r * LDY + fo[j]] = ynew;
            if constexpr (SAVE) {
                if (row_ok) {
                    if (a.traj) lean_gstore(ynew, goff4[j], a.traj + (size_t)(n + 1) * BH);
                    if (a.dW_out) lean_gstore(dw_cur[j], goff4[j], a.dW_out + (size_t)n * BH);
                }
            }
            dw_cur[j] = dw_nxt[j]; gt_cur[j] = gt_nxt[j];
        }
        asm volatile("ds_read_b64 %0, %2\n\tds_read_b128 %1, %2 offset:16"
                     : "+v"(qa), "+v"(qb) : "v"(lean_lds_addr(rowtab + (n + 1 - rbase) * RS)));
        if constexpr (KUXT > 0) lean_read_b_carried(xrow + ((n + 1) & 1) * (4 * LDX * 4), bx);
        __syncthreads();
    }
    if (row_ok) {
        const float w0 = a.out_w[2 * ko], w1 = a.out_w[2 * ko + 1];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float o = (w0 == 0.0f) ? yv[j] : snsde_interp_out(w0, w1, yold[j], yv[j]);
            const size_t go = (size_t)rowc * H + fo[j];
            if (!a.row_out) a.ys[(size_t)(ko + 1) * BH + go] = o;
            else if (rslot == ko + 1) a.ys[go] = o;
        }
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the rings' last refills (never consumed) land before the wave ends
}

template <class CF>
int launch_stream2(const MfmaArgs& a, hipStream_t stream) {
    const size_t lds_bytes = (size_t)CF::LDS_FLOATS * sizeof(float);
    static SnsdeLdsAttr lds_attr;   // per instantiation and device
    if (const int rc = snsde_lds_attr(reinterpret_cast<const void*>(snsde_m4s2_kernel<CF>), lds_bytes, lds_attr)) return rc;
    const int grid = (a.B + 3) / 4;
    hipLaunchKernelGGL(snsde_m4s2_kernel<CF>, dim3(grid), dim3(CF::NT), lds_bytes, stream, a);
    return hipGetLastError() == hipSuccess ? SNSDE_OK : SNSDE_ERR_LAUNCH;
}

}  // namespace snsde_mfma
