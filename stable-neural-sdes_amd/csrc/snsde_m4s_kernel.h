// Lean 4-row-tile forward kernel for H = 256 (the K5 shape): the weights no longer fit in the register file, so the
// y / hidden / output layers are STREAMED every step, L2 -> LDS, through a per-wave ring, and consumed from there.
//
// Budget behind the design (MI355X: 512 KB of VGPRs + 160 KB of LDS per CU; one 4-row workgroup of 16 waves per CU at
// B = 1024): three 256 x 256 layers are 786 KB of fp32 weights, so at most ~1/6 of them could stay on chip next to the
// working registers; the previous H = 256 instantiation of the general kernel streamed them into VGPRs four k-blocks
// at a time (4 KB in flight per wave -> ~9.4 us per step of pure load latency, plus 58-120 spilled VGPRs).  Here
//   * every wave owns 16 features; per layer its weight slice is 16 k-blocks of 1 KB (lane-linear float4 fragments, the
//     packed workspace layout of the lean kernel);
//   * the slice is copied by global_load_lds_dwordx4 (LDS-DMA: no VGPRs, asynchronous) into the wave's ring of R = 8
//     1 KB slots and read back as the MFMA A operand with ds_read_b128; a slot is refilled with the block R positions
//     ahead (wrapping into the next step: the stream never stops) as soon as its A fragments have been read, so 4-8 KB
//     per wave (64-128 KB per CU) are always in flight;
//   * the ring is private to its wave: no barrier covers it, and the kernel's own s_waitcnt vmcnt(N) are counted by hand
//     (every load / LDS-DMA of the step loop is issued from inline asm; vector-memory operations complete in order, so
//     operations the count does not know about only make a wait stricter);
//   * the [X(t) | sin t, cos t] block (KUXT <= 3 k-blocks) and the bias fragments stay in registers; Philox normals are
//     kept in three registers between refills instead of an LDS stash (the LDS belongs to the ring).
// Everything else (diffusion at the step top, reduce-scatter, one prefetch wait per step,
// table rows read in place) is the lean kernel's, snsde_m4_kernel.h.  Same MFMA chains (k order, two accumulators): the
// results are bit-identical to the general M4 kernel's up to the tanh form (see snsde_m4_kernel.h).
// Reference semantics: benchmark_classification/models_sde/neuralsde.py:295-307 (f, g), SURVEY.md A3-A6 (stepping).
#pragma once
#include "snsde_m4_kernel.h"

namespace snsde_mfma {

template <int NHID_, int KUXT_, int SAVE_, int ACT_ = 0>
struct CfgS {
    static constexpr bool SWISH = ACT_ != 0;
    static constexpr int H = 256, NHID = NHID_, KUXT = KUXT_;
    static constexpr bool SAVE = SAVE_ != 0;
    static constexpr int NW = 16, NT = 1024, KUH = 16;
    static constexpr int LDY = ld_for(16 * KUH, 16);
    static constexpr int LDX = ld_for(16 * (KUXT > 0 ? KUXT : 1), 16);
    static constexpr int LDA = LDY;
    static constexpr int NLAYER = NHID + 2, NSAVE = NHID + 2, ZSLOT = NHID + 1;
    static constexpr int ROWCH = 128, RS = 8;
    static constexpr int XI = 1;                              // 4 rows x (<= 48 columns) spread over 1024 lanes
    static constexpr int R = 8, CH = 2;                       // ring slots per wave; k-blocks consumed (and refilled) together
    static constexpr int NSTR = NHID + 2;                     // streamed layers: y, hidden.., out
    static constexpr int TB = KUH * NSTR;                     // streamed k-blocks per step and wave
    static constexpr int RING0 = 4 * (LDY + 2 * LDX + 2 * LDA) + (ROWCH + 3) * RS;   // float offset of the rings
    static constexpr int LDS_FLOATS = RING0 + NW * R * 256;
    static_assert(TB % R == 0 && KUH % R == 0 && R % CH == 0, "static ring slots");
};

template <class CF>
__global__ void __launch_bounds__(CF::NT, 1) snsde_m4s_kernel(MfmaArgs a) {
    constexpr int H = CF::H, NT = CF::NT, NHID = CF::NHID, KUH = CF::KUH, KUXT = CF::KUXT;
    constexpr int LDY = CF::LDY, LDX = CF::LDX, LDA = CF::LDA, RS = CF::RS;
    constexpr bool SAVE = CF::SAVE;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ybuf = lds;                       // [4][LDY]  y
    float* xbuf = ybuf + 4 * LDY;            // [2][4][LDX]  X(t) (xc) | sin t, cos t | 0..   (step parity)
    float* bufA = xbuf + 8 * LDX;            // [4][LDA]
    float* bufB = bufA + 4 * LDA;            // [4][LDA]
    float* rowtab = bufB + 4 * LDA;          // [ROWCH + 3][RS]
    float* ring = lds + CF::RING0;           // [NW][R][256]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 3, s = (lane >> 2) & 3, q = lane >> 4;
    const int fo = wave * 16 + 4 * q + s;
    const int row0 = blockIdx.x * 4;
    const int B = a.B, C = a.C, N = a.N;
    const int row = row0 + r;
    const bool row_ok = row < B;
    const int rowc = row_ok ? row : B - 1;
    const size_t BH = (size_t)B * H;
    const size_t goff = (size_t)rowc * H + fo;
    const uint32_t fo4 = (uint32_t)(fo * sizeof(float)), goff4 = (uint32_t)(goff * sizeof(float));
    const int xc = a.lean_xc;
    const bool time_on = a.lean_time != 0, geo = a.lean_geo != 0;
    const float act_scale = a.act == SNSDE_ACT_LIPSWISH ? 0.909f : 1.0f;
    const int f_out = a.f_out;
    const bool g_raw = a.g_out == SNSDE_DIFFUSION_RAW;

    // ---- resident: the xt block and the bias fragments; streamed: one SGPR base per layer ---------------------------------
    int li = 0;
    float wxt[(KUXT > 0 ? KUXT : 1) * 4];
    if constexpr (KUXT > 0) lean_load_w<KUXT>(wxt, a.ws + a.w_off[li++], wave, lane);
    uint64_t sb[CF::NSTR];
#pragma unroll
    for (int l = 0; l < CF::NSTR; ++l) sb[l] = lean_uniform(a.ws + a.w_off[li++] + (size_t)wave * KUH * 256);
    float bias_own[CF::NLAYER];        // added after the k-slot reduction (one register per layer)
#pragma unroll
    for (int l = 0; l < CF::NLAYER; ++l) bias_own[l] = a.ws[a.bias_off + l * H + fo];
    const uint32_t ringb = lean_lds_addr(ring) + (uint32_t)wave * (CF::R * 1024);
    const uint32_t m0v = __builtin_amdgcn_readfirstlane(ringb + 4096u);
    const uint32_t ra = ringb + (uint32_t)lane * 16u;
    const uint32_t vo_lo = (uint32_t)lane * 16u + 4096u, vo_hi = vo_lo + 8192u;

    for (int i = tid; i < 4 * (LDY + 2 * LDX + 2 * LDA); i += NT) lds[i] = 0.0f;
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

    float yv = a.y0[goff];
    ybuf[r * LDY + fo] = yv;
    if (row_ok) {
        a.ys[(size_t)row * H + fo] = yv;
        if constexpr (SAVE) { if (a.traj) a.traj[(size_t)row * H + fo] = yv; }
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
    // the step's prefetches (coefficients, increment, table entry) have landed: they are older than the R LDS-DMA blocks
    // that may still be in flight
    auto vm_wait_all = [&](float& dwn, float& gtn) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(dwn), "+v"(gtn), "+v"(ca), "+v"(cb), "+v"(cc), "+v"(cd));
    };
    auto vm_wait = [&](float& dwn, float& gtn) {
        asm volatile("s_waitcnt vmcnt(%6)" : "+v"(dwn), "+v"(gtn), "+v"(ca), "+v"(cb), "+v"(cc), "+v"(cd) : "n"(CF::R));
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

    // Brownian increment of step i: one Philox block gives the element's normals of four consecutive steps, kept in
    // registers (same counters as every other kernel: bit-identical increments)
    float zr[4] = {0.f, 0.f, 0.f, 0.f};
    auto next_dw = [&](int i, float sqh) -> float {
        if (__builtin_expect(phx, 1)) {
            const int k = i & 3;
            if (k == 0) snsde_philox_normal4(seed, grow, (uint32_t)(i >> 2), (uint32_t)fo, zr);
            const float z = k == 0 ? zr[0] : (k == 1 ? zr[1] : (k == 2 ? zr[2] : zr[3]));
            return z * sqh;
        }
        float v;
        lean_gload(v, goff4, a.dW + (size_t)i * BH);
        return v;
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
    float dw_cur, gt_cur = 0.0f;
    f32x2 qa;           // (h_n, sqrt h_{n+1})
    f32x4 qb;           // (sin, cos, frac of step n+1, idx of step n+2)
    ca = cb = cc = cd = 0.0f;
    {
        const float* g0 = a.step_tab;
        load_coeffs(__float_as_int(g0[5]));
        float dummy = 0.0f;
        vm_wait_all(dummy, gt_cur);
        store_xt(xbuf, g0[4], a.raw_time ? g0[0] : g0[2], a.raw_time ? 0.0f : g0[3]);
        dw_cur = next_dw(0, g0[6]);
        if (tab) lean_gload(gt_cur, fo4, gt);
        load_coeffs(__float_as_int(a.step_tab[(size_t)(N > 1 ? 1 : 0) * SNSDE_STEP_STRIDE + 5]));
        vm_wait_all(dw_cur, gt_cur);
        qa = *reinterpret_cast<const f32x2*>(rowtab);
        qb = *reinterpret_cast<const f32x4*>(rowtab + 4);
    }
    __syncthreads();
    // the ring's first R blocks (layer 0, k-blocks 0 .. 7)
    stream_refill<0>(m0v, vo_lo, vo_hi, sb[0]); stream_refill<1>(m0v, vo_lo, vo_hi, sb[0]);
    stream_refill<2>(m0v, vo_lo, vo_hi, sb[0]); stream_refill<3>(m0v, vo_lo, vo_hi, sb[0]);
    stream_refill<4>(m0v, vo_lo, vo_hi, sb[0]); stream_refill<5>(m0v, vo_lo, vo_hi, sb[0]);
    stream_refill<6>(m0v, vo_lo, vo_hi, sb[0]); stream_refill<7>(m0v, vo_lo, vo_hi, sb[0]);

    const uint32_t yrow = lean_lds_addr(ybuf + r * LDY + 4 * s);
    const uint32_t xrow = lean_lds_addr(xbuf + r * LDX + 4 * s);
    float* const aown = bufA + r * LDA + fo;
    float* const bown = bufB + r * LDA + fo;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    LeanB<(KUXT > 0 ? KUXT : 1)> bx{};
    if constexpr (KUXT > 0) lean_read_b_carried(xrow, bx);
    int n = 0;
    float yold = yv;
    for (int ko = 0; ko < a.T - 1; ++ko) {
    const int n_end = a.out_step[ko];
    for (; n <= n_end; ++n) {
        const int rbase = (n / CF::ROWCH) * CF::ROWCH;
        if (n > 0 && n == rbase) {
            fill_rows(rbase);
            __syncthreads();
        }
        const bool more = n + 1 < N;
        [[maybe_unused]] uint32_t sgn = 0;
        asm volatile("" : "+v"(qa), "+v"(qb));
        const float h = qa[0];
        // ---- first layer: the [X(t_n) | tau_n] part from registers, then the streamed y part ------------------------------
        f32x4 c = zero4, d = zero4;
        if constexpr (KUXT > 0) lean_gemm<15, KUXT>(wxt, bx, c, d);
        const float ypart = gpart(yv, gt_cur, dw_cur, h);
        store_xt(xbuf + ((n + 1) & 1) * (4 * LDX), qb[2], qb[0], qb[1]);
        load_coeffs(__float_as_int(qb[3]));
        __builtin_amdgcn_sched_barrier(0);
        stream_layer<0>(yrow, ra, m0v, vo_lo, vo_hi, sb[0], sb[1], c, d);
        {
            const float pre = m4_reduce_scatter(c + d) + bias_own[0];
            const float o = CF::SWISH ? lean_swish(pre, act_scale) : fmaxf(pre, 0.0f);
            *aown = o;
            if constexpr (SAVE) { if (a.act_save && row_ok) lean_gstore(o, goff4, a.act_save + ((size_t)n * CF::NSAVE) * BH); }
            // relu signs of this lane's element (snsde_pack_signs).  NHID == 1: both rectified outputs are still in this lane's LDS slots
            // when z is stored and are read back there (the 128-register budget of this kernel has no room to carry them: (1, 3) spilled)
            if constexpr (SAVE && !CF::SWISH && NHID != 1) sgn = o > 0.0f ? 1u : 0u;
        }
        __syncthreads();
        // ---- hidden layers; the next step's increment and diffusion-table entry are fetched in the first window -------------
        float dw_nxt = 0.0f, gt_nxt = 0.0f;
        auto prep = [&]() {
            const int n1 = more ? n + 1 : n;
            dw_nxt = next_dw(n1, qa[1]);
            if (__builtin_expect(tab, 1)) lean_gload(gt_nxt, fo4, gt + (size_t)n1 * H);
        };
        constexpr int OFFA = (8 * LDX + 4 * LDY) * 4, OFFB = OFFA + 4 * LDA * 4;     // bufA / bufB rows from the y rows, bytes
#pragma unroll
        for (int l = 0; l < NHID; ++l) {
            const bool toB = (l % 2 == 0);
            if (l == 0) prep();
            __builtin_amdgcn_sched_barrier(0);
            c = zero4; d = zero4;
            if (toB) stream_layer<OFFA>(yrow, ra, m0v, vo_lo, vo_hi, sb[1 + l], sb[2 + l], c, d);
            else stream_layer<OFFB>(yrow, ra, m0v, vo_lo, vo_hi, sb[1 + l], sb[2 + l], c, d);
            const float pre = m4_reduce_scatter(c + d) + bias_own[1 + l];
            const float o = CF::SWISH ? lean_swish(pre, act_scale) : fmaxf(pre, 0.0f);
            *(toB ? bown : aown) = o;
            if constexpr (SAVE) { if (a.act_save && row_ok) lean_gstore(o, goff4, a.act_save + ((size_t)n * CF::NSAVE + 1 + l) * BH); }
            if constexpr (SAVE && !CF::SWISH && NHID != 1) sgn |= (o > 0.0f ? 1u : 0u) << (1 + l);
            __syncthreads();
        }
        // ---- output layer (its last two chunks refill the ring with the NEXT step's first blocks), f, update ------------------
        if (NHID == 0) prep();
        __builtin_amdgcn_sched_barrier(0);
        c = zero4; d = zero4;
        if (NHID % 2 == 0) stream_layer<OFFA>(yrow, ra, m0v, vo_lo, vo_hi, sb[NHID + 1], sb[0], c, d);
        else stream_layer<OFFB>(yrow, ra, m0v, vo_lo, vo_hi, sb[NHID + 1], sb[0], c, d);
        vm_wait(dw_nxt, gt_nxt);
        float z = m4_reduce_scatter(c + d) + bias_own[NHID + 1];
        if constexpr (SAVE) {      // the saved pre-tanh drift carries the step's relu signs in its low NHID + 1 bits (the adjoint's masks)
            if (a.act_save && row_ok) {
                if constexpr (!CF::SWISH && NHID == 1) sgn = (*aown > 0.0f ? 1u : 0u) | (*bown > 0.0f ? 2u : 0u);
                lean_gstore(CF::SWISH ? z : snsde_pack_signs(z, sgn, NHID + 1), goff4, a.act_save + ((size_t)n * CF::NSAVE + CF::ZSLOT) * BH);
            }
        }
        if (__builtin_expect(geo, 0)) z *= fast_tanh(yv);
        float f;
        if (__builtin_expect(f_out != SNSDE_DRIFT_TANH, 0)) f = f_out == SNSDE_DRIFT_TIMES_Y ? z * yv : z;
        else f = LEAN_TANH_F(z);
        const float ynew = fmaf(f, h, ypart);
        yold = yv;
        yv = ynew;
        ybuf[r * LDY + fo] = ynew;
        if constexpr (SAVE) {
            if (row_ok) {
                if (a.traj) lean_gstore(ynew, goff4, a.traj + (size_t)(n + 1) * BH);
                if (a.dW_out) lean_gstore(dw_cur, goff4, a.dW_out + (size_t)n * BH);
            }
        }
        dw_cur = dw_nxt; gt_cur = gt_nxt;
        asm volatile("ds_read_b64 %0, %2\n\tds_read_b128 %1, %2 offset:16"
                     : "+v"(qa), "+v"(qb) : "v"(lean_lds_addr(rowtab + (n + 1 - rbase) * RS)));
        if constexpr (KUXT > 0) lean_read_b_carried(xrow + ((n + 1) & 1) * (4 * LDX * 4), bx);
        __syncthreads();
    }
    if (row_ok) {
        const float w0 = a.out_w[2 * ko], w1 = a.out_w[2 * ko + 1];
        const float o = (w0 == 0.0f) ? yv : snsde_interp_out(w0, w1, yold, yv);
        if (!a.row_out) a.ys[(size_t)(ko + 1) * BH + goff] = o;
        else if (rslot == ko + 1) a.ys[goff] = o;
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the ring's last refills (never consumed) land before the wave ends
}

template <class CF>
int launch_stream(const MfmaArgs& a, hipStream_t stream) {
    const size_t lds_bytes = (size_t)CF::LDS_FLOATS * sizeof(float);
    static SnsdeLdsAttr lds_attr;   // per instantiation and device
    if (const int rc = snsde_lds_attr(reinterpret_cast<const void*>(snsde_m4s_kernel<CF>), lds_bytes, lds_attr)) return rc;
    const int grid = (a.B + 3) / 4;
    hipLaunchKernelGGL(snsde_m4s_kernel<CF>, dim3(grid), dim3(CF::NT), lds_bytes, stream, a);
    return hipGetLastError() == hipSuccess ? SNSDE_OK : SNSDE_ERR_LAUNCH;
}

}  // namespace snsde_mfma
