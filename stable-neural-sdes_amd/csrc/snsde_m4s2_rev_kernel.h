// Adjoint of the H = 256 Euler / Milstein solve on 4-row tiles, TWO tiles per wave with a quarter of every transposed matrix resident
// (round 6; the forward's snsde_m4s2_kernel.h applied to the FL / RING branch of snsde_mfma_reverse_kernel): eight waves at 256
// registers, per transposed GEMM and tile the first RK = 4 k-blocks in registers, the other 12 through the per-wave, per-tile LDS
// rings (s2_layer), one B-operand read feeding both tiles.  The reference's own fields only (relu, tanh drift, tanh(sigmoid(theta) raw)
// diffusion, elementwise noise options, y-dependent drifts): the tutorial variants, the diffusion nets, the accumulator column and
// input_option 0 stay on the general kernel.
// Same recursion, same chains (c: fragments 0, 2; d: 1, 3; reduce-scatter), the theta partial sums kept per TILE (slot 2 wave + j of
// the 16 per workgroup the parameter pass reduces) => bit-identical adjoints, deltas and partial sums; SNSDE_FLAG_STREAM_ALL keeps
// the sixteen-wave kernel (tests/test_gpu_parity.py::test_h256_two_tile_adjoint_is_bit_identical_to_the_streamed_one).
// What it computes: include/snsde.h (snsde_backward, mode 1); reference: loss.backward() through the unrolled solver,
// benchmark_classification/common_sde.py:158-160.
#pragma once
#include "snsde_m4s2_kernel.h"

namespace snsde_mfma {

template <int NHID_, int GEO_>
struct CfgS2R {
    static constexpr int H = 256, NHID = NHID_;
    static constexpr bool GEO = GEO_ != 0;
    static constexpr int NW = 8, NT = 512, KUH = 16, TPW = 2, M = 4;
    static constexpr int LDA = ld_for(16 * KUH, 16);
    static constexpr int ND = NHID + 2, NG = ND, NBUF = NHID + 2, NSAVE = NHID + 2, ZSLOT = NHID + 1;
    static constexpr int ROWCH = 128;
    static constexpr int RK = 4, NS = KUH - RK, R = 6;
    static constexpr int RING0 = NBUF * M * LDA + (ROWCH + 1) * SNSDE_STEP_STRIDE;      // multiple of 4 floats
    static constexpr int LDS_FLOATS = RING0 + NW * TPW * R * 256;
    static_assert(RING0 % 4 == 0, "ring alignment");
};

template <class CF>
__global__ void __launch_bounds__(CF::NT, 1) snsde_m4s2_reverse_kernel(RevArgs a) {
    constexpr int H = CF::H, M = CF::M, NT = CF::NT, NHID = CF::NHID, NG = CF::NG, ND = CF::ND, LDA = CF::LDA, KUH = CF::KUH, RK = CF::RK;
    constexpr int NSAVE = CF::NSAVE, NS = CF::NBUF;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* rowtab = lds + NS * M * LDA;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 3, s = (lane >> 2) & 3, fsub = 4 * (lane >> 4);
    const int row0 = blockIdx.x * M, B = a.B;
    const int row = row0 + r, rowc = row < B ? row : B - 1;
    const bool row_ok = row < B;
    const size_t BH = (size_t)B * H;
    int fcol[2];
    uint32_t goff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { fcol[j] = (2 * wave + j) * 16 + fsub + s; goff[j] = (uint32_t)(rowc * H + fcol[j]); }

    // ---- resident: the first RK k-blocks of every transposed matrix, both tiles; streamed: SGPR bases ------------------------------
    uint64_t sb[NG][2];
    float wr[NG][2][RK * 4];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const float* base = a.ws + a.w_off[g];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float* sl = base + (size_t)(2 * wave + j) * KUH * 256;
            sb[g][j] = lean_uniform(sl);
#pragma unroll
            for (int u = 0; u < RK; ++u) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(sl + (size_t)u * 256 + lane * 4);
                wr[g][j][4 * u] = v[0]; wr[g][j][4 * u + 1] = v[1]; wr[g][j][4 * u + 2] = v[2]; wr[g][j][4 * u + 3] = v[3];
            }
        }
    }
    const uint32_t ring_m0 = __builtin_amdgcn_readfirstlane(lean_lds_addr(lds + CF::RING0) + (uint32_t)wave * (2 * CF::R * 1024));
    const uint32_t ra = ring_m0 + (uint32_t)lane * 16u;
    const uint32_t vo4[4] = {(uint32_t)lane * 16u, (uint32_t)lane * 16u + 4096u, (uint32_t)lane * 16u + 8192u, (uint32_t)lane * 16u + 12288u};
    static_assert(CF::R == 6 && RK == 4, "initial fill below");
    // (in CONSUMPTION order - per block: tile 0's slot, tile 1's slot - which is what the vmcnt of the first layer's first block counts on)
#define SNSDE_S2_FILL(UB) \
    s2_refill<RK + UB, UB * 1024>(ring_m0, vo4, sb[0][0]); s2_refill<RK + UB, CF::R * 1024 + UB * 1024>(ring_m0, vo4, sb[0][1]);
    SNSDE_S2_FILL(0) SNSDE_S2_FILL(1) SNSDE_S2_FILL(2) SNSDE_S2_FILL(3) SNSDE_S2_FILL(4) SNSDE_S2_FILL(5)
#undef SNSDE_S2_FILL
    for (int i = tid; i < NS * M * LDA; i += NT) lds[i] = 0.0f;

    const float sig_theta = snsde_sigmoid(a.params[a.off_theta]);
    const bool mul_y = (a.no == 13 || a.no == 17 || a.no == 15 || a.no == 19 || a.no == 3 || a.no == 6 || a.no == 11);
    const bool yfun = (a.no >= 7 && a.no <= 10);
    const float mil = (a.method == SNSDE_MILSTEIN) ? 0.5f : 0.0f;

    auto fill_rows = [&](int base) {
        for (int i = tid; i < (CF::ROWCH + 1) * SNSDE_STEP_STRIDE; i += NT) {
            const int rr = base + i / SNSDE_STEP_STRIDE;
            rowtab[i] = a.step_tab[(size_t)(rr < a.N ? rr : a.N - 1) * SNSDE_STEP_STRIDE + i % SNSDE_STEP_STRIDE];
        }
    };

    float adj[2] = {0.f, 0.f}, gfin[2];
    const int rslot = a.row_out ? a.row_out[rowc] : -1;
#pragma unroll
    for (int j = 0; j < 2; ++j) gfin[j] = a.row_out ? a.grad_ys[goff[j]] : 0.0f;
    int rbase = -1;
    const bool dsum = a.ds_part != nullptr && a.gt != nullptr;
    const float rowf = row_ok ? 1.0f : 0.0f;
    float th_acc[2] = {0.f, 0.f};          // per TILE: the parameter pass sums 16 partials per workgroup, one per tile

    struct StepIn { float y[2], z[2], dw[2], gq[2]; };
    float zblk[2][4];
    int zblk_id = -1;
    const uint32_t grow = (uint32_t)(a.row_offset + row);
    const uint32_t BH32 = (uint32_t)BH, SBH = (uint32_t)NSAVE * BH32, NSBH = (uint32_t)NS * BH32;
    auto prefetch = [&](int n, StepIn& p) {
        if (!a.dW) {
            if ((n >> 2) != zblk_id) {
                zblk_id = n >> 2;
                snsde_philox_normal4(a.seed, grow, (uint32_t)zblk_id, (uint32_t)fcol[0], zblk[0]);
                snsde_philox_normal4(a.seed, grow, (uint32_t)zblk_id, (uint32_t)fcol[1], zblk[1]);
            }
            const float sqh = a.step_tab[uoff(n, SNSDE_STEP_STRIDE) + 6];
            const bool odd = (n & 1) != 0, hi = (n & 2) != 0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float lo2 = odd ? zblk[j][1] : zblk[j][0], hi2 = odd ? zblk[j][3] : zblk[j][2];
                p.dw[j] = (hi ? hi2 : lo2) * sqh;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            p.y[j] = (a.traj + uoff(n, BH32))[goff[j]];
            p.z[j] = (a.act + uoff(n, SBH, CF::ZSLOT, BH32))[goff[j]];
            if (a.dW) p.dw[j] = (a.dW + uoff(n, BH32))[goff[j]];
            p.gq[j] = a.gt ? (a.gt + uoff(n, H))[fcol[j]] : 0.0f;
        }
    };
    StepIn cur, nxt;
    prefetch(a.N - 1, cur);
    const uint32_t baddr = lean_lds_addr(lds + r * LDA + 4 * s);

    for (int n = a.N - 1; n >= 0; --n) {
#pragma unroll
        for (int j = 0; j < 2; ++j) { nxt.y[j] = cur.y[j]; nxt.z[j] = cur.z[j]; nxt.dw[j] = cur.dw[j]; nxt.gq[j] = cur.gq[j]; }
        if (n > 0) prefetch(n - 1, nxt);
        const int nb = (n / CF::ROWCH) * CF::ROWCH;
        if (nb != rbase) {
            __syncthreads();
            rbase = nb;
            fill_rows(rbase);
            __syncthreads();
        }
        const float* st = rowtab + (n - rbase) * SNSDE_STEP_STRIDE;
        const float h = st[1];
        const int nout = __float_as_int(st[8]), kfirst = __float_as_int(st[9]);

        float carry[2] = {0.f, 0.f};
        for (int k = kfirst; k < kfirst + nout; ++k) {
            const float w0 = a.out_w[2 * k], w1 = a.out_w[2 * k + 1];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float gk = a.row_out ? (rslot == k + 1 ? gfin[j] : 0.0f) : (a.grad_ys + uoff(k + 1, BH32))[goff[j]];
                if (w0 == 0.0f) adj[j] += gk;
                else { adj[j] = fmaf(w1, gk, adj[j]); carry[j] = fmaf(w0, gk, carry[j]); }
            }
        }
        if (row_ok && !a.adj0_only) {
#pragma unroll
            for (int j = 0; j < 2; ++j) (a.adj + uoff(n + 1, BH32))[goff[j]] = adj[j];
        }
        // ---- elementwise: d(f h + g dW)/d(zout, y) applied to the adjoint (the general kernel's reference-field branch) ----------------
        float ay[2], dz[2], dsv[2];
        uint32_t zb[2];
        const uint32_t zclear = ~((1u << (NHID + 1)) - 1u);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            dsv[j] = 0.0f;
            zb[j] = __builtin_bit_cast(uint32_t, cur.z[j]);
            const float y = cur.y[j], z = __builtin_bit_cast(float, zb[j] & zclear), dw = cur.dw[j], gq = cur.gq[j];
            const float av = adj[j];
            float ty = 1.0f, zt = z;
            if constexpr (CF::GEO) { ty = fast_tanh(y); zt = z * ty; }
            const float f = fast_tanh(zt);
            const float dzt = av * h * (1.0f - f * f);
            float acc_y = av;
            if constexpr (CF::GEO) { dz[j] = dzt * ty; acc_y = fmaf(dzt * z, 1.0f - ty * ty, acc_y); }
            else dz[j] = dzt;
            float q1 = 0.0f, q2 = 0.0f;
            const float raw = yfun ? snsde_phi(a.no, y, q1, q2) : (mul_y ? gq * y : gq);
            const float rcv = snsde_nan_to_num(raw);
            const float g = fast_tanh(sig_theta * rcv);
            const bool finite = snsde_finite(raw);
            const float om = 1.0f - g * g;
            if (yfun) {
                if (finite) {
                    const float g1 = om * sig_theta * q1;
                    const float g2 = om * sig_theta * q2 - 2.0f * g * g1 * sig_theta * q1;
                    const float qq = mil * fmaf(dw, dw, -h);
                    acc_y = fmaf(av, fmaf(qq, fmaf(g1, g1, g * g2), dw * g1), acc_y);
                    th_acc[j] = fmaf(av * rowf * om, fmaf(qq * q1, fmaf(sig_theta * rcv, fmaf(-3.0f * g, g, 1.0f), g), dw * rcv), th_acc[j]);
                } else {
                    th_acc[j] = fmaf(av * rowf * om * dw, rcv, th_acc[j]);
                }
            } else if (mul_y && finite) {
                const float c = sig_theta * gq;
                const float dm = mil * fmaf(dw, dw, -h) * c * fmaf(-3.0f * g, g, 1.0f);
                acc_y = fmaf(av * om * c, dw + dm, acc_y);
            }
            ay[j] = acc_y;
            if (dsum) {
                const float du = av * dw * om * rowf;
                th_acc[j] = fmaf(du, rcv, th_acc[j]);
                float d = finite ? du * sig_theta * (mul_y ? y : 1.0f) : 0.0f;
                if (mul_y && finite && mil != 0.0f) {
                    const float c = sig_theta * gq;
                    const float ex = av * rowf * om * (mil * fmaf(dw, dw, -h)) * fmaf(fmaf(-3.0f * g, g, 1.0f) * c, y, g);
                    th_acc[j] = fmaf(ex, gq, th_acc[j]);
                    d = fmaf(ex, sig_theta, d);
                }
                dsv[j] = d;
            }
        }
        if (dsum) {     // sum over the tile's four rows (lane & 3), one writer lane per feature
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v = dsv[j];
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
                if (r == 0) (a.ds_part + ((size_t)blockIdx.x * a.N + n) * H + fcol[j])[0] = v;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            lds[r * LDA + fcol[j]] = dz[j];
            if (a.delta && row_ok) (a.delta + uoff(n, NSBH))[goff[j]] = dz[j];
        }
        __syncthreads();
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        constexpr int BUFB = M * LDA * 4;      // bytes per LDS buffer
        // transposed GEMM G: its input is LDS buffer G; the weight stream runs into GEMM G + 1 (the NEXT step's GEMM 0 after the last)
#define SNSDE_S2R_GEMM(G)                                                                                                              \
        if constexpr (G < NG) {                                                                                                        \
            f32x4 c[2] = {zero4, zero4}, d[2] = {zero4, zero4};                                                                        \
            constexpr int GN = (G + 1) % NG;                                                                                           \
            s2_layer<CfgS2<1, 1, 0>, G * BUFB>(wr[G], baddr, ra, ring_m0, vo4, sb[G][0], sb[G][1], sb[GN][0], sb[GN][1], c, d);        \
            constexpr bool MID = G < ND - 1;                                                                                           \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                            \
                const float o = m4_reduce_scatter(c[j] + d[j]);                                                                        \
                if constexpr (MID) {                                                                                                   \
                    const float dv = ((zb[j] >> (NHID - G)) & 1u) ? o : 0.0f;      /* relu sign of act slot NHID - G */                 \
                    lds[(G + 1) * M * LDA + r * LDA + fcol[j]] = dv;                                                                   \
                    if (a.delta && row_ok) (a.delta + uoff(n, NSBH, G + 1, BH32))[goff[j]] = dv;                                       \
                } else {                                                                                                               \
                    adj[j] = ay[j] + o + carry[j];      /* end of the drift chain */                                                   \
                }                                                                                                                      \
            }                                                                                                                          \
            if constexpr (MID) __syncthreads();                                                                                        \
        }
        SNSDE_S2R_GEMM(0) SNSDE_S2R_GEMM(1) SNSDE_S2R_GEMM(2) SNSDE_S2R_GEMM(3)
#undef SNSDE_S2R_GEMM
#pragma unroll
        for (int j = 0; j < 2; ++j) { cur.y[j] = nxt.y[j]; cur.z[j] = nxt.z[j]; cur.dw[j] = nxt.dw[j]; cur.gq[j] = nxt.gq[j]; }
    }
    if (row_ok) {     // ys[0] = y0
#pragma unroll
        for (int j = 0; j < 2; ++j) a.adj[goff[j]] = adj[j] + (a.row_out ? (rslot == 0 ? gfin[j] : 0.0f) : a.grad_ys[goff[j]]);
    }
    if ((dsum || yfun) && a.dth_part) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float t = th_acc[j];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
            if (lane == 0) a.dth_part[blockIdx.x * 16 + 2 * wave + j] = t;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the rings' last requests land before the wave ends
}

template <class CF>
int launch_rev2(const RevArgs& a, hipStream_t stream) {
    const size_t lds_bytes = (size_t)CF::LDS_FLOATS * sizeof(float);
    static SnsdeLdsAttr lds_attr;   // per instantiation and device
    if (const int rc = snsde_lds_attr(reinterpret_cast<const void*>(snsde_m4s2_reverse_kernel<CF>), lds_bytes, lds_attr)) return rc;
    const int grid = (a.B + CF::M - 1) / CF::M;
    hipLaunchKernelGGL(snsde_m4s2_reverse_kernel<CF>, dim3(grid), dim3(CF::NT), lds_bytes, stream, a);
    return hipGetLastError() == hipSuccess ? SNSDE_OK : SNSDE_ERR_LAUNCH;
}

}  // namespace snsde_mfma
