// MFMA fast path of the fused Neural-SDE solver for gfx950 (CDNA4).
//
// One persistent workgroup owns a tile of M batch rows for ALL solver steps.  Per step the MLP drift
// (neuralsde.py:295-302) is a chain of small GEMMs  out^T (features x rows) = W (features x K) . act^T,
// executed on the f32 matrix cores (exact f32: an MFMA is bit-for-bit an fmaf chain):
//
//   * every wave of the workgroup owns a slice of OUTPUT FEATURES of every layer and keeps that slice
//     of every weight matrix RESIDENT IN REGISTERS (VGPR+AGPR, one wave per SIMD, 512 registers/lane)
//     for the whole solve: zero weight traffic per step;
//   * weights are the MFMA A operand, activations the B operand ("transposed" product), so a batch row
//     stays in one lane column: the D fragment of a layer IS one 16-byte LDS store per lane, and the
//     next layer's B fragment is plain ds_read_b128 of the row-major LDS activation buffer (the k-slot
//     to feature assignment is arbitrary as long as A and B agree: k(u, s, e) = 16u + 4s + e);
//   * the state y, dW, f, g and the Euler/Milstein update live in registers in the D layout of the last
//     layer; the time-only diffusion MLP of noise_option 16/17 is a per-step table (hoisted);
//   * activations cross waves through padded LDS buffers (row stride = 8 or 16 mod 64 floats:
//     conflict-free ds_read_b128), one s_barrier per layer.
//
// Two tile flavours share the code:
//   M16: v_mfma_f32_16x16x4_f32, 16 rows per workgroup  (lane = 16*s + row).        Large batches.
//   M4 : v_mfma_f32_4x4x1_16b_f32, 4 rows per workgroup: the 16 independent 4x4 blocks are used as
//        4 k-slots x 4 feature quads (lane = 16*q + 4*s + row); the k-slot partial sums are combined by a
//        DPP reduce-scatter (lane s keeps feature 4q+s: one element per lane from there on), and the B operand
//        (a function of lane & 15 only) is read by lanes 0-15 and broadcast by the MFMA (blgp:4).
//        Fills all 256 CUs at batch 1024.

#include <stddef.h>

#include "snsde_m4_kernel.h"
#include "snsde_m4n_mil_rev_kernel.h"

using namespace snsde_mfma;

namespace {

// packed[dst + ((w*TPW + t)*KU + u)*256 + lane*4 + e] = W[feature][k(u,s,e)]
// `direct`: the folded layers' in-range entries (and folded bias) are written by the fold blocks of the same launch
// (snsde_prepare_kernel), this pass only writes their zero padding.
__device__ __forceinline__ int packed_index(int flavor, int KU, int feat, int k);

// fsrc: where the folded products live (fold_tmp offsets): the workspace being packed into, or - backward pack after a folded forward -
// the FORWARD's workspace, whose prepare launch already formed emb[:, 0:H] . linear_in (no second fold launch in the backward)
__device__ __forceinline__ void pack_layer(const float* __restrict__ params, float* __restrict__ ws, const float* __restrict__ fsrc,
                                           const MfmaPackJob& job, const MfmaLayerPack& L, int bx, int nbx, bool direct) {
    const int per_wave = job.TPW * L.KU * 256;
    const int total = job.NW * per_wave;
    const bool skip = direct && L.fold && !L.transpose;
    const int Kown = L.t_on ? L.K - L.tshift : L.K;     // columns of the layer's own block
    for (int i = bx * blockDim.x + threadIdx.x; i < total; i += nbx * blockDim.x) {
        const int e = i & 3, lane = (i >> 2) & 63, blk = i >> 8;
        const int u = blk % L.KU, wt = blk / L.KU;  // wt = w*TPW + t
        int feat, s;
        if (job.flavor == 0) { feat = lane & 15; s = lane >> 4; }
        else { feat = 4 * (lane >> 4) + (lane & 3); s = (lane >> 2) & 3; }
        feat += 16 * wt;
        const int k = 16 * u + 4 * s + e;
        float v = 0.0f;
        if (L.transpose) {
            if (feat < L.N && k < L.K)
                v = L.fold ? fsrc[L.fold_tmp + k * L.src_ld + L.col_off + feat] : params[L.src_w + k * L.src_ld + L.col_off + feat];
        } else if (feat < L.N && k < Kown) {
            const int sk = L.t_on ? k + L.tshift : ((k < L.K - L.tshift) ? k + L.tshift : k - (L.K - L.tshift));
            if (skip) continue;
            v = L.fold ? fsrc[L.fold_tmp + feat * L.K + sk] : params[L.src_w + feat * L.K + sk];
        } else if (k >= L.hole0 && k < L.hole1) {
            continue;                                   // columns another piece owns (the redirected time columns)
        }
        ws[L.dst + i] = v;
    }
    if (L.t_on && !skip && !L.transpose) {              // leading time columns -> the [X(t) | sin t, cos t] block
        for (int i = bx * blockDim.x + threadIdx.x; i < L.N * L.tshift; i += nbx * blockDim.x) {
            const int feat = i / L.tshift, j = i - feat * L.tshift;
            ws[L.t_dst + packed_index(job.flavor, L.t_KU, feat, L.t_col0 + j)] =
                L.fold ? fsrc[L.fold_tmp + feat * L.K + j] : params[L.src_w + feat * L.K + j];
        }
    }
    // bias table [row][H]
    if (bx == 0 && L.bias_row >= 0 && !skip) {
        for (int j = threadIdx.x; j < job.H; j += blockDim.x) {
            float b = 0.0f;
            if (j < L.N) {
                if (L.fold) {
                    b = ws[job.fold_bias_tmp + j];
                } else {
                    b = params[L.src_b + j];
                }
            }
            ws[job.bias_off + L.bias_row * job.H + j] = b;
        }
    }
}

// kernarg offsets of the by-value job structs (their `layer` arrays are indexed by blockIdx.y: snsde_kernarg_element)
constexpr size_t PACK_JOB_OFF = 24;      // snsde_mfma_pack_kernel(params, ws, fold_src, job)
constexpr size_t PREP_JOB_OFF = (16 + sizeof(FoldJob) + alignof(MfmaPackJob) - 1) / alignof(MfmaPackJob) * alignof(MfmaPackJob);
static_assert(alignof(FoldJob) == 8 && alignof(MfmaPackJob) == 4, "kernarg layout of snsde_prepare_kernel");

__global__ void snsde_mfma_pack_kernel(const float* __restrict__ params, float* __restrict__ ws, const float* __restrict__ fold_src,
                                       MfmaPackJob job) {
    const MfmaLayerPack L = snsde_kernarg_element<MfmaLayerPack>(PACK_JOB_OFF + offsetof(MfmaPackJob, layer), blockIdx.y);
    pack_layer(params, ws, fold_src ? fold_src : ws, job, L, blockIdx.x, gridDim.x, false);
}

// position of weight (feature, k) inside a layer's packed fragment block (inverse of the pack loop's index map)
__device__ __forceinline__ int packed_index(int flavor, int KU, int feat, int k) {
    const int wt = feat >> 4, fl = feat & 15, u = k >> 4, s = (k >> 2) & 3, e = k & 3;
    const int lane = flavor == 0 ? 16 * s + fl : 16 * (fl >> 2) + 4 * s + (fl & 3);
    return ((wt * KU + u) * 64 + lane) * 4 + e;
}

// One "prepare" launch: blockIdx.y in {0,1} = the two folded products F = E[:, col:col+H] . W (one block per
// output row f, lanes over the K columns, coalesced reads of W rows), blockIdx.y == 2 = the time-only diffusion
// table (one block per solver step).
__device__ __forceinline__ void fold_block(const float* __restrict__ params, float* __restrict__ ws, const FoldJob& job,
                                           const MfmaPackJob* pk, const MfmaLayerPack* pkL, float* erow) {      // pkL = &pk->layer[blockIdx.y]
    const int pc = blockIdx.y, H = job.H;
    if (pc == 2) {
        const int n = blockIdx.x;
        if (!job.tab_on || n >= job.n_steps) return;
        const float* st = job.step_tab + (size_t)n * SNSDE_STEP_STRIDE;
        snsde_time_table_row(params, st[0], st[2], st[3], ws + job.tab_off + (size_t)n * H, job.nt0, job.nt1, H, job.no, erow,
                             job.off_sigma, job.off_sigma_diag);
        return;
    }
    const int f = blockIdx.x;
    if (!job.fold_on || f >= H || pc >= job.n_pieces) return;
    const float* e = params + job.emb_w + (size_t)f * 2 * H;
    for (int j = threadIdx.x; j < 2 * H; j += blockDim.x) erow[j] = e[j];
    __syncthreads();
    const float* W = params + job.src_w[pc];
    const int K = job.K[pc];
    const float* ec = erow + job.col[pc];
    auto emit = [&](int k, float val) {
        ws[job.tmp[pc] + f * K + k] = val;
        if (pk) {     // forward prepare: straight into the packed MFMA fragment layout (source column k -> packed column kp)
            const MfmaLayerPack& L = *pkL;
            if (L.t_on && k < L.tshift) {
                ws[L.t_dst + packed_index(pk->flavor, L.t_KU, f, L.t_col0 + k)] = val;    // time column -> the xt block
            } else {
                const int kp = (k >= L.tshift) ? k - L.tshift : k + (L.K - L.tshift);
                ws[L.dst + packed_index(pk->flavor, L.KU, f, kp)] = val;
            }
        }
    };
    // A thread owns columns k and k + blockDim (K > 256: H = 256 with its two time columns) and walks them TOGETHER: the walk is a chain
    // of H / 16 dependent round trips, and a second pass for two leftover columns doubled it (prepare launch 24 us at K5, 7 us at K2)
    for (int k = threadIdx.x; k < K; k += 2 * blockDim.x) {
        const int k2 = k + blockDim.x;
        const bool two = k2 < K;
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;   // four independent chains hide the load latency
        float b0 = 0.0f, b1 = 0.0f, b2 = 0.0f, b3 = 0.0f;
        int j = 0;
        for (; j + 15 < H; j += 16) {      // 16 (32) loads in flight per round trip (same accumulation order as the 4-wide loop;
                                           // 32 / 64 in flight measured 3x slower: the kernel's 256-thread blocks lose their registers to it)
            float wv[16], xv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) wv[i] = W[(size_t)(j + i) * K + k];
            if (two) {
#pragma unroll
                for (int i = 0; i < 16; ++i) xv[i] = W[(size_t)(j + i) * K + k2];
            }
#pragma unroll
            for (int i = 0; i < 16; i += 4) {
                a0 = fmaf(ec[j + i], wv[i], a0);
                a1 = fmaf(ec[j + i + 1], wv[i + 1], a1);
                a2 = fmaf(ec[j + i + 2], wv[i + 2], a2);
                a3 = fmaf(ec[j + i + 3], wv[i + 3], a3);
            }
            if (two) {
#pragma unroll
                for (int i = 0; i < 16; i += 4) {
                    b0 = fmaf(ec[j + i], xv[i], b0);
                    b1 = fmaf(ec[j + i + 1], xv[i + 1], b1);
                    b2 = fmaf(ec[j + i + 2], xv[i + 2], b2);
                    b3 = fmaf(ec[j + i + 3], xv[i + 3], b3);
                }
            }
        }
        for (; j + 3 < H; j += 4) {
            a0 = fmaf(ec[j], W[(size_t)j * K + k], a0);
            a1 = fmaf(ec[j + 1], W[(size_t)(j + 1) * K + k], a1);
            a2 = fmaf(ec[j + 2], W[(size_t)(j + 2) * K + k], a2);
            a3 = fmaf(ec[j + 3], W[(size_t)(j + 3) * K + k], a3);
            if (two) {
                b0 = fmaf(ec[j], W[(size_t)j * K + k2], b0);
                b1 = fmaf(ec[j + 1], W[(size_t)(j + 1) * K + k2], b1);
                b2 = fmaf(ec[j + 2], W[(size_t)(j + 2) * K + k2], b2);
                b3 = fmaf(ec[j + 3], W[(size_t)(j + 3) * K + k2], b3);
            }
        }
        for (; j < H; ++j) {
            a0 = fmaf(ec[j], W[(size_t)j * K + k], a0);
            if (two) b0 = fmaf(ec[j], W[(size_t)j * K + k2], b0);
        }
        emit(k, (a0 + a1) + (a2 + a3));
        if (two) emit(k2, (b0 + b1) + (b2 + b3));
    }
    if (pc == 0 && threadIdx.x < 64) {   // folded bias: b_emb + E1 b_in + E2 b_init (one wave, shuffle reduction)
        float acc = 0.0f;
        for (int j = threadIdx.x; j < H; j += 64)
            acc += erow[j] * params[job.b_in + j] + erow[H + j] * params[job.b_init + j];
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        if (threadIdx.x == 0) {
            const float b = acc + params[job.b_emb + f];
            ws[job.bias_tmp + f] = b;
            if (pk) {
                for (int i = 0; i < 2; ++i)
                    if (pk->layer[i].bias_row >= 0) ws[pk->bias_off + pk->layer[i].bias_row * pk->H + f] = b;
            }
        }
    }
}

__global__ void snsde_fold_kernel(const float* __restrict__ params, float* __restrict__ ws, FoldJob job) {
    extern __shared__ float erow[];
    fold_block(params, ws, job, nullptr, nullptr, erow);
}

// Forward prepare in ONE launch: blockIdx.y 0/1 = folded products (written to the temp the backward reads AND to their
// packed fragments), 2 = time-only diffusion table, 3 + l = packing of layer l (16 blocks each).
__global__ void snsde_prepare_kernel(const float* __restrict__ params, float* __restrict__ ws, FoldJob fj, MfmaPackJob job) {
    extern __shared__ float erow[];
    if (blockIdx.y < 3) {
        const MfmaLayerPack L = snsde_kernarg_element<MfmaLayerPack>(PREP_JOB_OFF + offsetof(MfmaPackJob, layer), blockIdx.y < 2 ? blockIdx.y : 0);
        fold_block(params, ws, fj, fj.fold_on ? &job : nullptr, &L, erow);
        return;
    }
    if ((int)blockIdx.y == 3 + job.n_layers) { snsde_z0_rows(fj.z0, blockIdx.x, gridDim.x); return; }   // y0 = W0 X(ts[0]) + b0
    if (blockIdx.x < 16) {
        const MfmaLayerPack L = snsde_kernarg_element<MfmaLayerPack>(PREP_JOB_OFF + offsetof(MfmaPackJob, layer), blockIdx.y - 3);
        pack_layer(params, ws, ws, job, L, blockIdx.x, 16, fj.fold_on != 0);
    }
}

// SRK variant: one step-table row per drift pass (stage times t0, t0 + h, t0 + h/2 = slots 0, 3, 2 of the stage table);
// outputs are emitted after the last pass of a step only.
__global__ void snsde_srk_expand_kernel(const float* __restrict__ step_tab, const float* __restrict__ srk_tab,
                                        float* __restrict__ out, int n_steps, int raw_time) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * n_steps) return;
    const int n = i / 3, stg = i - 3 * n;
    const int slot = stg == 0 ? 0 : (stg == 1 ? 3 : 2);
    const float* st = step_tab + (size_t)n * SNSDE_STEP_STRIDE;
    const float* tp = srk_tab + ((size_t)n * 4 + slot) * SNSDE_SRK_STRIDE;
    float* o = out + (size_t)i * SNSDE_STEP_STRIDE;
    o[0] = tp[0]; o[1] = st[1]; o[4] = tp[3]; o[5] = tp[4]; o[6] = st[6]; o[7] = st[7];
    o[2] = raw_time ? tp[0] : tp[1]; o[3] = raw_time ? 0.0f : tp[2];      // time features: [sin t, cos t] or [t, 0] (SNSDE_TIME_RAW)
    o[8] = stg == 2 ? st[8] : __int_as_float(0); o[9] = st[9];
    // [10], [11]: sin / cos of the DIFFUSION stage time evaluated beside this pass (snsde_m4n_kernel.h): t0, t0 + h/4, t0 + h
    const float* tn = srk_tab + ((size_t)n * 4 + (stg == 0 ? 0 : (stg == 1 ? 1 : 3))) * SNSDE_SRK_STRIDE;
    o[10] = raw_time ? tn[0] : tn[1]; o[11] = raw_time ? 0.0f : tn[2];
}

// tutorial-style fields (variant switches of snsde_model / a caller-supplied noise table): the lean 4-row-tile kernels only
static bool variant_of(const snsde_solve* s) {
    const snsde_model& m = s->model;
    return m.activation != 0 || m.drift_output != 0 || m.diffusion_output != 0 || m.time_feature != 0 || s->noise_table != nullptr;
}

// Which configurations the fast path is instantiated for.
// Wave-owns-rows forward (snsde_w4.hip): forced by hint 2; under `auto` up to 6144 rows
static bool w4_takes(const snsde_solve* s, const SnsdeNet& net, int flavor_hint) {
    // measured at the K4 shape (tools/time_w4.py, profiles/r05_time_w4.txt): 2048 rows 109 us against 156 (4-row tiles) / 216 (16-row
    // tiles), 6144 rows 275 against 316 (16-row tiles); from 8192 rows the 16-row tiles win (a third wave per SIMD does not fit)
    // SRK through a net has no 16-row flavour: the wave pair at every batch size (2048 rows 352 us against 456, 4096 rows 540 / 900)
    return (flavor_hint == 2 || (flavor_hint == -1 && (s->batch <= 6144 || s->method == SNSDE_SRK))) && snsde_w4_supported(s, net);
}

MfmaPlan make_plan(const snsde_solve* s, const SnsdeNet& net, int flavor_hint) {
    MfmaPlan p{};
    const int hint_in = flavor_hint;
    if (flavor_hint == 2) flavor_hint = 1;      // (the wave-owns-rows forward, snsde_w4.hip: every plan-side decision as for 4-row tiles)
    const snsde_model& m = s->model;
    const int H = m.hidden_channels, io = m.input_option, no = m.noise_option;
    p.ok = false;
    // kernel = 'w4' is a strict request: where the wave-pair kernels do not take the configuration there is no plan, so that the
    // path query, the launch, snsde_backward_supported and snsde_save_layout all report the same thing
    if (hint_in == 2 && !snsde_w4_supported(s, net)) return p;
    if (m.hidden_hidden_channels != H) return p;
    // the MFMA kernels form their per-step save offsets from 32-bit uniform factors (uoff): slots x B x H must fit.  Refused HERE
    // (not only in the launchers) so that `auto` falls through to the generic kernels and snsde_backward_supported reports what the
    // launchers accept
    if ((uint64_t)16 * (uint64_t)(s->batch > 0 ? s->batch : 0) * (uint64_t)H >= (1ull << 32)) return p;
    const bool srk = s->method == SNSDE_SRK;
    // SRK: 4-row tiles; 16-row tiles for the elementwise diffusions at H = 64 / 128, C <= 32 (large batches)
    const bool srk_m16_ok = srk && (H == 64 || H == 128) && !(no == 14 || no == 15 || no == 18 || no == 19) && m.input_channels <= 32 &&
                            m.activation == 0 && m.drift_output == 0 && m.diffusion_output == 0 && m.time_feature == 0 && !s->noise_table;
    if (srk && flavor_hint == 0 && !srk_m16_ok) return p;
    if (!(H == 256 || H == 128 || H == 64 || H == 32 || H == 16)) return p;
    if (!(io >= 0 && io <= 6)) return p;
    const bool noise_net = (no == 14 || no == 15 || no == 18 || no == 19);
    // table noise: raw = (row of a per-step table) x {1, y}: the time-only noise MLPs and the closed forms in t, sigma
    const bool tab_noise = (no >= 1 && no <= 6) || no == 11 || no == 12 || no == 13 || no == 16 || no == 17;
    const bool y_noise = (no >= 7 && no <= 10);      // raw = phi(y): sqrt y, y^3, sigmoid y, relu y
    if (!(no == 0 || tab_noise || y_noise || noise_net)) return p;
    // SRK / Milstein through a diffusion net: the 4-row-tile kernels of snsde_m4n_kernel.h (embedded or latent-only drifts)
    const bool variant_m = m.activation != 0 || m.drift_output != 0 || m.diffusion_output != 0 || m.time_feature != 0;
    const int kuxn = (io == 2 || io == 4 || io == 6) ? (m.input_channels > 32 ? 5 : 2) : 0;
    // Euler with a net takes them too where the general kernel has no (spill-free) instantiation: a wide control path behind
    // the embedding (C > 32, e.g. the sepsis channels), and H = 128 on 4-row tiles (its five resident matrices spill there)
    bool m4n = noise_net && (s->method != SNSDE_EULER || variant_m);      // (field variants with a net: these kernels, every method)
    if (noise_net && !m4n && io != 0 && flavor_hint != 0 && m.num_hidden_layers <= 4 &&
        m4n_instantiated(H, kuxn, m.num_hidden_layers - 1, no >= 18 ? 2 : 1, SNSDE_EULER)) {
        const long r4 = ((s->batch + 3) / 4 + 255) / 256, r16 = ((s->batch + 15) / 16 + 255) / 256;
        const bool m4_tiles = flavor_hint == 1 || 10 * r4 <= 22 * r16;
        if (kuxn == 5 || (H == 128 && m4_tiles)) m4n = true;
    }
    if (m4n && (flavor_hint == 0 || io == 0 || m.num_hidden_layers > 4 ||
                !m4n_instantiated(H, kuxn, m.num_hidden_layers - 1, no >= 18 ? 2 : 1, s->method)))
        return p;
    if (noise_net && !m4n && (io == 0 || io == 2 || io == 4 || io == 6) && m.input_channels > 32) return p;   // nets + wide control: generic
    if (srk && m.input_channels > 32 && (io == 0 || m.num_hidden_layers > 3)) return p;   // wide control under SRK: embedded drifts, NL <= 3
    const bool emb = (io == 2 || io == 4 || io == 6);
    const int nhid = m.num_hidden_layers - 1;
    if (nhid > 3) return p;
    p.NN = noise_net ? (no >= 18 ? 2 : 1) : 0;
    if ((emb || io == 0) && m.input_channels > 80) return p;
    p.H = H; p.IO = io; p.NHID = nhid;
    p.KUX = (emb || io == 0) ? (m.input_channels > 32 ? 5 : 2) : 1;    // 16-wide k-blocks of the control channels (C <= 32 / <= 80)
    p.TPW = 1;
    p.NW = H / (16 * p.TPW);
    // flavour: a solve costs (rounds of resident workgroups) x (steps) x (step latency); an M16 step takes ~2.2x an
    // M4 step (measured at H=128: 4.9 vs 2.2 us) but carries 4x the rows, so M16 wins as soon as M4 needs more than
    // two rounds per M16 round.  Narrow models (H <= 64: 4 or fewer waves per workgroup) co-reside on a CU.
    {
        const int slots = 256 * (p.NW <= 4 ? 8 / (p.NW < 2 ? 2 : p.NW) : 1);
        const int slots16 = srk ? 256 : slots;      // (the SRK variant's 16-row workgroups do not co-reside: H = 64 at 6144 rows ran two rounds)
        const long r4 = ((s->batch + 3) / 4 + slots - 1) / slots, r16 = ((s->batch + 15) / 16 + slots16 - 1) / slots16;
        // (round 4, profiles/r04_sweep_flavour.txt: the ratio of an M16 round to an M4 round is 3.5 at H = 256 - streamed weights - and
        //  1.8 at H = 64, where two co-resident 4-row workgroups slow each other down; 2.2 elsewhere)
        const long ratio10 = H == 256 ? 35 : ((H == 64 && !srk) ? 18 : 22);      // (measured under Euler; the SRK variant keeps 2.2)
        p.FL = flavor_hint >= 0 ? flavor_hint : (10 * r4 > ratio10 * r16 ? 0 : 1);
    }
    p.SRK = srk ? 1 : 0;
    if ((srk && !srk_m16_ok) || m4n) p.FL = 1;
    // the wave-pair kernels (snsde_w4.hip) take the forward AND the Euler adjoint: their per-tile partial sums are per 4 rows
    if (hint_in < 0 && w4_takes(s, net, -1)) p.FL = 1;
    p.M4N = m4n ? 1 : 0; p.KUXN = kuxn;
    p.FOLD = (emb && (nhid > 1 || p.KUX > 2 || srk || noise_net || !(s->flags & SNSDE_FLAG_EXACT_ORDER))) ? 1 : 0;   // exact order: NL <= 2, C <= 32 only
    // lean M4 kernel (snsde_m4_kernel.h): 4-row tiles, Euler / Milstein, elementwise diffusions, 32 <= H <= 128; the time
    // features share the control path's k-block
    const bool usex = emb || io == 0, timef = io >= 3;
    const int xcn = usex ? m.input_channels : 0;
    int kuxt = (xcn + (timef ? 2 : 0) + 15) / 16;
    if (kuxt == 4 || kuxt == 5) kuxt = 6;        // instantiated: 0, 1, 2, 3, 6 blocks
    p.KUXT = kuxt;
    p.LEAN = (p.FL == 1 && !srk && !m4n && p.NN == 0 && (!emb || p.FOLD) && kuxt <= 6 && lean_fits(H, nhid, kuxt, io != 0)) ? 1 : 0;
    const bool variant = m.activation != 0 || m.drift_output != 0 || m.diffusion_output != 0 || m.time_feature != 0 ||
                         s->noise_table != nullptr;
    // tutorial-style fields: lean kernel instantiations with an activation switch (Euler / Milstein), or the general kernel's
    // SRK variant (embedded drifts, C <= 32, H <= 128, a supplied table at the four stage times or no diffusion)
    if (variant && m4n && (s->noise_table || io == 0)) return p;
    const bool variant_net = variant && m4n;      // NeuralSDEFunc-shaped fields: snsde_m4n_kernel.h carries the switches
    const bool variant_srk = variant && srk && !m4n && emb && H <= 128 && p.KUX == 2 && (no == 0 || ((no == 12 || no == 13) && s->noise_table));
    if (variant && !variant_srk && !variant_net &&
        !(p.LEAN && (s->noise_table == nullptr || no == 12 || no == 13) && (no == 0 || tab_noise) && kuxt <= 3 &&
          io != 0 && io != 5 && io != 6))
        return p;
    // workspace layout: bias rows | time-only diffusion table | SRK pass table | packed fragments | fold temps.  The first
    // three do not depend on the tile flavour / kernel variant, so the backward finds the table whatever forward ran.
    int n = 0, rows = 0, woff = 0;
    auto add = [&](const SnsdeLayer& L, int KU, int fold_col, bool bias) -> MfmaLayerPack& {
        MfmaLayerPack& q = p.layer[n++];
        q = MfmaLayerPack{};
        q.src_w = L.src_w; q.src_b = L.src_b; q.K = L.K; q.tshift = L.tshift; q.N = L.N; q.KU = KU; q.dst = woff;
        q.fold = fold_col >= 0 ? 1 : 0;
        q.fold_w = net.emb.src_w; q.fold_col = fold_col >= 0 ? fold_col : 0; q.fold_ld = 2 * H;
        q.bias_row = bias ? rows++ : -1;
        q.fold_tmp = -1;
        woff += p.NW * p.TPW * KU * 256;
        return q;
    };
    const int KUH = H / 16;
    const int KUYv = KUH + (io >= 3 ? 1 : 0);
    if (p.LEAN) {
        // kernel load order: xt block ([X(t) | sin t, cos t], if any), y block, hidden.., out
        if (kuxt > 0) {
            if (usex) {
                MfmaLayerPack& q = add(net.init, kuxt, p.FOLD ? H : -1, io == 0);
                if (timef) { q.hole0 = xcn; q.hole1 = xcn + 2; }
            } else {             // time features only: an all-padding block whose columns 0, 1 the `in` piece fills
                MfmaLayerPack& q = p.layer[n++];
                q = MfmaLayerPack{};
                q.KU = kuxt; q.dst = woff; q.bias_row = -1; q.fold_tmp = -1; q.hole0 = 0; q.hole1 = 2;
                woff += p.NW * p.TPW * kuxt * 256;
            }
        }
        if (io != 0) {
            const int xt_dst = p.layer[0].dst;
            MfmaLayerPack& q = add(net.in, KUH, p.FOLD ? 0 : -1, true);
            if (timef) { q.t_on = 1; q.t_dst = xt_dst; q.t_KU = kuxt; q.t_col0 = xcn; }
        }
        if (p.FOLD) {
            p.fold_b_in = net.in.src_b; p.fold_b_init = net.init.src_b; p.fold_b_emb = net.emb.src_b;
            p.fold_emb_w = net.emb.src_w;
        }
    } else if (p.FOLD) {
        // kernel load order: wx (init piece), wy (in piece); ONE bias row (written by the `in` piece)
        add(net.init, p.KUX, H, false);
        add(net.in, m4n ? KUH + 1 : KUYv, 0, true);     // (the net kernels always carry the time block: zero columns without it)
        p.fold_b_in = net.in.src_b; p.fold_b_init = net.init.src_b; p.fold_b_emb = net.emb.src_b;
        p.fold_emb_w = net.emb.src_w;
    } else if (io == 0) {
        add(net.init, p.KUX, -1, true);                  // z0 = initial_network(X(t))
    } else {
        if (emb) add(net.init, p.KUX, -1, true);
        add(net.in, m4n ? KUH + 1 : KUYv, -1, true);
        if (emb) add(net.emb, 2 * KUH, -1, true);
    }
    for (int l = 0; l < nhid; ++l) add(net.hid[l], KUH, -1, true);
    add(net.out, KUH, -1, true);
    if (p.NN >= 1) add(net.ny0, KUH + 1, -1, true);
    if (p.NN >= 2) add(net.ny1, KUH, -1, true);
    if (m4n && s->method == SNSDE_MILSTEIN) {      // the VJP through the net: ny1^T, ny0[:, y columns]^T
        auto add_t = [&](const SnsdeLayer& L, int col_off) {
            MfmaLayerPack& q = p.layer[n++];
            q = MfmaLayerPack{};
            q.src_w = L.src_w; q.src_b = L.src_b; q.K = H; q.N = H; q.KU = KUH; q.dst = woff;
            q.transpose = 1; q.src_ld = L.K; q.col_off = col_off; q.bias_row = -1; q.fold_tmp = -1;
            woff += p.NW * KUH * 256;
        };
        if (p.NN >= 2) add_t(net.ny1, 0);
        add_t(net.ny0, net.ny0.tshift);
    }
    p.n_bias_rows = rows;
    p.n_layers = n;
    int off = 0;
    p.bias_off = off;
    off += rows * H;
    off = (off + 3) & ~3;
    p.gt_off = tab_noise ? off : -1;
    if (p.gt_off >= 0) off += s->n_steps * H * (srk ? 4 : 1);     // SRK: the four stage times of every step
    p.srk_tab_off = -1;
    if (srk) { p.srk_tab_off = off; off += 3 * s->n_steps * SNSDE_STEP_STRIDE; }
    off = (off + 3) & ~3;
    for (int i = 0; i < n; ++i) {
        p.layer[i].dst += off;
        if (p.layer[i].t_on) p.layer[i].t_dst += off;
    }
    off += woff;
    if (p.FOLD) {   // temps for the folded products
        for (int i = 0; i < 2; ++i) { p.layer[i].fold_tmp = off; off += H * p.layer[i].K; }
        p.fold_bias_tmp = off; off += H;
        off = (off + 3) & ~3;
    }
    p.total_floats = off;
    // path-integral accumulator column (snsde.h: kl_column1): the lean 4-row-tile kernel (Euler / Milstein, H <= 128) and the SRK
    // variant of the general kernel on 4-row tiles, for the LatentSDE mapping only: linear drift output, raw additive table
    if (s->kl_column1 != 0) {
        const bool shape_ok = s->kl_column1 >= 2 && s->kl_column1 <= H && m.drift_output == SNSDE_DRIFT_LINEAR &&
                              m.diffusion_output == SNSDE_DIFFUSION_RAW && s->noise_table != nullptr && no == 12 && p.NN == 0 && io != 0;
        const bool kernel_ok = (p.LEAN && H <= 128 && !srk) || (srk && p.FL == 1 && !m4n);
        if (!shape_ok || !kernel_ok) return p;
    }
    p.ok = true;
    return p;
}

RevPlan make_rev_plan(const snsde_solve* s, const SnsdeNet& net, const MfmaPlan& fp) {
    RevPlan p{};
    p.ok = false;
    if (!fp.ok || (s->method != SNSDE_EULER && s->method != SNSDE_MILSTEIN && s->method != SNSDE_SRK)) return p;
    // diffusion nets: Euler on the general adjoint kernel, SRK on snsde_m4n_rev_kernel.h (Milstein: the generic adjoint)
    const bool m4n_rev = fp.M4N && !variant_of(s) &&
                         ((s->method == SNSDE_SRK && m4n_rev_instantiated(fp.H, fp.NHID, fp.NN)) ||
                          (s->method == SNSDE_MILSTEIN && m4n_mil_rev_instantiated(fp.H, fp.NHID, fp.NN)));
    // ... field variants under SRK / Milstein on the VAR instantiations of the same two kernels (two-layer nets: NeuralSDEFunc; round 5)
    const bool variant_srk_net_rev = fp.M4N && variant_of(s) && (s->method == SNSDE_SRK || s->method == SNSDE_MILSTEIN) && fp.H <= 128 &&
                                     fp.IO != 0 && fp.NN == 2 && fp.NHID <= 2 && !s->noise_table &&
                                     s->model.diffusion_output != SNSDE_DIFFUSION_TANH && s->model.drift_output != SNSDE_DRIFT_TIMES_Y &&
                                     (s->method == SNSDE_SRK ? m4n_rev_instantiated(fp.H, fp.NHID, fp.NN)
                                                             : m4n_mil_rev_instantiated(fp.H, fp.NHID, fp.NN));
    if (fp.NN != 0 && s->method != SNSDE_EULER && !m4n_rev && !variant_srk_net_rev) return p;
    // field variants with a net (NeuralSDEFunc-shaped, fields.py): Euler on the general adjoint kernel's 4-row tiles
    const bool variant_net_rev = fp.M4N && variant_of(s) && s->method == SNSDE_EULER && fp.FL == 1 && fp.H <= 128 && fp.IO != 0 &&
                                 !s->noise_table && s->model.diffusion_output != SNSDE_DIFFUSION_TANH &&
                                 s->model.drift_output != SNSDE_DRIFT_TIMES_Y;
    if (fp.M4N && variant_of(s) && !variant_net_rev && !variant_srk_net_rev) return p;
    p.M4N = (m4n_rev || variant_srk_net_rev) ? (s->method == SNSDE_SRK ? 1 : 2) : 0;
    // tutorial-style fields under SRK (the general kernel's SRK variant ran the forward): the SRK adjoint kernel carries the
    // switches; a raw diffusion from a supplied table (rows = the four stage times of every step) or none
    const bool variant_srk_rev = fp.SRK && s->method == SNSDE_SRK && variant_of(s) && !fp.M4N && fp.FL == 1 && fp.H <= 128 &&
                                 fp.IO != 5 && fp.IO != 6 && fp.IO != 0 &&
                                 ((s->model.diffusion_output == SNSDE_DIFFUSION_RAW && s->noise_table != nullptr) ||
                                  s->model.noise_option == 0);
    // tutorial-style fields: the register-resident lean forward (its training-mode instantiations), Euler / Milstein
    if (variant_of(s) && !variant_net_rev && !variant_srk_net_rev && !variant_srk_rev && !(fp.LEAN && fp.FL == 1 && fp.H <= 128 && !fp.SRK && s->method != SNSDE_SRK &&
                           (s->model.activation == SNSDE_ACT_RELU || lean_act_save_fits(fp.H, fp.NHID, fp.KUXT)) &&
                           (s->model.diffusion_output == SNSDE_DIFFUSION_RAW || s->model.noise_option == 0) &&
                           (s->noise_table != nullptr || s->model.noise_option == 0)))
        return p;      // (a raw diffusion from a supplied table, or none: theta and noise_t take no part)
    p.SRK = fp.SRK;
    const int H = fp.H, io = fp.IO;
    if (p.SRK && io == 0) return p;      // (the SRK adjoint kernel has no y-free variant: the generic adjoint takes it)
    p.H = H; p.NHID = fp.NHID; p.GEO = (io == 5 || io == 6) ? 1 : 0; p.FL = fp.SRK ? 1 : fp.FL; p.NW = fp.NW; p.NN = fp.NN;      // (the SRK adjoint: 4-row tiles, whatever tiles the forward ran on)
    p.emb = (io == 2 || io == 4 || io == 6) ? 1 : 0;
    p.IO0 = io == 0 ? 1 : 0;
    int off = 0, n = 0;
    auto add_t = [&](const SnsdeLayer& L, int col_off, int fold_tmp) {
        MfmaLayerPack& q = p.layer[n++];
        q = MfmaLayerPack{};
        q.src_w = L.src_w; q.src_b = L.src_b; q.K = H; q.N = H; q.KU = H / 16; q.dst = off; q.tshift = 0;
        q.transpose = 1; q.src_ld = L.K; q.col_off = col_off; q.bias_row = -1;
        q.fold = fold_tmp >= 0 ? 1 : 0; q.fold_tmp = fold_tmp;
        off += p.NW * (H / 16) * 256;
    };
    // fold temp first (so its offset is known to the pack job)
    int fold_tmp = -1;
    const int packed = (p.NHID + 2 + (p.M4N == 2 ? 2 * p.NN : p.NN)) * p.NW * (H / 16) * 256;
    if (p.emb) fold_tmp = packed;
    add_t(net.out, 0, -1);
    for (int l = p.NHID - 1; l >= 0; --l) add_t(net.hid[l], 0, -1);
    if (!p.IO0) add_t(net.in, net.in.tshift, fold_tmp);      // the y-free drift has no first_y^T
    if (p.M4N == 2) {
        // Milstein through the net (snsde_m4n_mil_rev_kernel.h): forward-layout W1_y, [W2], then [W2^T], W1_y^T
        auto add_f = [&](const SnsdeLayer& L, int tshift) {
            MfmaLayerPack& q = p.layer[n++];
            q = MfmaLayerPack{};
            q.src_w = L.src_w; q.src_b = L.src_b; q.K = L.K; q.N = H; q.KU = H / 16; q.dst = off; q.tshift = tshift;
            q.bias_row = -1; q.fold_tmp = -1;        // (KU = H / 16 blocks: the y columns only, the rotated time columns fall off)
            off += p.NW * (H / 16) * 256;
        };
        add_f(net.ny0, net.ny0.tshift);
        if (p.NN == 2) { add_f(net.ny1, 0); add_t(net.ny1, 0, -1); }
        add_t(net.ny0, net.ny0.tshift, -1);
    } else {
        if (p.NN == 2) add_t(net.ny1, 0, -1);                   // diffusion net, output layer first
        if (p.NN >= 1) add_t(net.ny0, net.ny0.tshift, -1);      // its y columns
    }
    p.n_layers = n;
    p.fold_tmp = fold_tmp;
    p.total_floats = packed + (p.emb ? H * net.in.K + H : 0) + 16;
    p.nwg = (s->batch + (p.FL ? 4 : 16) - 1) / (p.FL ? 4 : 16);
    p.ds_off = p.dth_off = 0;
    const int no_ = s->model.noise_option;
    if (fp.gt_off >= 0 || p.NN > 0 || (no_ >= 7 && no_ <= 10)) {     // the adjoint kernel also leaves the diffusion-side parameter sums
        size_t o = ((size_t)p.total_floats + 3) & ~(size_t)3;
        if (fp.gt_off >= 0) { p.ds_off = o; o += (size_t)p.nwg * s->n_steps * H * (p.SRK ? 4 : 1); }   // time-only noise MLP: d/d s_n
        p.dth_off = o; o += (size_t)p.nwg * p.NW;                                    // d/d sigmoid(theta)
        if (o + 16 > 0x7fffffff) return p;
        p.total_floats = (int)(o + 16);
    }
    p.ok = true;
    return p;
}

}  // namespace


// time table kernel lives in snsde_generic.hip

bool snsde_mfma_supported(const snsde_solve* s, const SnsdeNet& net) { return make_plan(s, net, -1).ok; }

// adjoint on the wave groups: Euler or SRK (SRID2) with a diffusion net, 4-row-tile plan, every a_n or dL/dy0 only
static bool w4_rev_takes(const snsde_solve* s, const SnsdeNet& net, const MfmaPlan& fp, const RevPlan& p, int hint) {
    // (p.NW >= 4: the per-tile theta partials - four floats per 4-row tile - live in the plan's dth block of nwg x NW floats)
    const bool method_ok = s->method == SNSDE_SRK ? (p.SRK && p.M4N == 1 && p.NW >= 4) : (!p.M4N && !p.SRK);
    return hint != 0 && hint != 1 && p.ok && p.FL == 1 && method_ok && fp.NN > 0 && s->kl_column1 == 0 &&
           (hint == 2 || s->batch <= 6144) && snsde_w4_rev_supported(s, net);
}

// which MFMA kernel family a forward launch of this descriptor takes (SNSDE_PATH_*; 0: none)
int snsde_mfma_path(const snsde_solve* s, const SnsdeNet& net, int flavor_hint) {
    const MfmaPlan p = make_plan(s, net, flavor_hint);
    if (!p.ok) return SNSDE_PATH_NONE;
    if (flavor_hint != 0 && flavor_hint != 1 && w4_takes(s, net, flavor_hint)) return SNSDE_PATH_MFMA_W4;
    if (p.SRK) return SNSDE_PATH_MFMA_SRK;
    if (p.LEAN) return p.H == 256 ? SNSDE_PATH_LEAN_STREAMED : SNSDE_PATH_LEAN;
    return p.FL ? SNSDE_PATH_MFMA_M4 : SNSDE_PATH_MFMA_M16;
}

size_t snsde_mfma_workspace_floats(const snsde_solve* s, const SnsdeNet& net) {
    // the packed layout depends on the kernel variant (the lean 4-row kernel merges the time features into the control
    // block): size the workspace for whichever variant a later launch of this descriptor may select
    size_t need = 0;
    for (int hint = 0; hint <= 1; ++hint) {
        MfmaPlan p = make_plan(s, net, hint);
        if (p.ok && (size_t)p.total_floats > need) need = (size_t)p.total_floats;
    }
    return need;
}

int snsde_mfma_launch(const snsde_solve* s, const SnsdeNet& net, hipStream_t stream, int flavor_hint) {
    MfmaPlan p = make_plan(s, net, flavor_hint);
    if (!p.ok) return SNSDE_ERR_UNSUPPORTED;
    // the kernels form their per-step save offsets from 32-bit uniform factors (uoff): slots x B x H must fit
    if ((uint64_t)16 * (uint64_t)s->batch * (uint64_t)s->model.hidden_channels >= (1ull << 32)) return SNSDE_ERR_UNSUPPORTED;
    float* ws = static_cast<float*>(s->workspace);
    const bool w4 = w4_takes(s, net, flavor_hint);
    if (flavor_hint == 2 && !w4) return SNSDE_ERR_UNSUPPORTED;
    if (w4 && !s->act_save) {
        // inference on the wave-owns-rows kernel: it reads the nn.Linear layout of `params` itself - no packing, no tables
        if (s->z0_weight) { const int rc = snsde_z0_launch(s, stream); if (rc) return rc; }
        return snsde_w4_launch(s, net, stream);
    }
    if (!(s->flags & SNSDE_FLAG_REUSE_PREPARED)) {
        MfmaPackJob job{};
        for (int i = 0; i < p.n_layers; ++i) job.layer[i] = p.layer[i];
        job.n_layers = p.n_layers; job.flavor = p.FL; job.TPW = p.TPW; job.NW = p.NW; job.bias_off = p.bias_off; job.H = p.H;
        job.fold_b_in = p.fold_b_in; job.fold_b_init = p.fold_b_init; job.fold_b_emb = p.fold_b_emb;
        job.fold_emb_w = p.fold_emb_w;
        job.fold_bias_tmp = p.fold_bias_tmp;
        {   // folded products + time-only diffusion table + fragment packing in ONE launch
            FoldJob fj{};
            fj.fold_on = p.FOLD; fj.H = p.H; fj.n_pieces = 2;
            if (p.FOLD) {
                fj.emb_w = p.fold_emb_w;
                for (int i = 0; i < 2; ++i) {
                    fj.src_w[i] = p.layer[i].src_w; fj.K[i] = p.layer[i].K; fj.col[i] = p.layer[i].fold_col;
                    fj.tmp[i] = p.layer[i].fold_tmp;
                }
                fj.b_in = p.fold_b_in; fj.b_init = p.fold_b_init; fj.b_emb = p.fold_b_emb; fj.bias_tmp = p.fold_bias_tmp;
            }
            fj.tab_on = p.gt_off >= 0 && !p.SRK && !s->noise_table; fj.tab_off = p.gt_off; fj.n_steps = s->n_steps; fj.no = s->model.noise_option;
            fj.nt0 = net.nt0; fj.nt1 = net.nt1; fj.step_tab = s->step_tab;
            fj.off_sigma = net.off_sigma; fj.off_sigma_diag = net.off_sigma_diag;
            int gx = fj.tab_on && s->n_steps > p.H ? s->n_steps : p.H;
            if (gx < 16) gx = 16;
            fj.z0 = SnsdeZ0Job{};
            if (s->z0_weight) {      // one initial-state element per thread: the slice wants B H / 256 blocks (the others exit early)
                fj.z0 = SnsdeZ0Job{s->z0_weight, s->z0_bias, s->coeffs, s->step_tab, const_cast<float*>(s->y0), s->batch,
                                   s->model.hidden_channels, s->model.input_channels, s->knots};
                int gz = (s->batch * s->model.hidden_channels + 255) / 256;
                if (gz > 2048) gz = 2048;
                if (gz > gx) gx = gz;
            }
            hipLaunchKernelGGL(snsde_prepare_kernel, dim3(gx, 3 + p.n_layers + (s->z0_weight ? 1 : 0)), dim3(256),
                               2 * p.H * sizeof(float), stream, s->params, ws, fj, job);
        }
        if (p.SRK) {
            if (!s->srk_tab) return SNSDE_ERR_NULL;
            hipLaunchKernelGGL(snsde_srk_expand_kernel, dim3((3 * s->n_steps + 127) / 128), dim3(128), 0, stream, s->step_tab,
                               s->srk_tab, ws + p.srk_tab_off, s->n_steps, s->model.time_feature == SNSDE_TIME_RAW ? 1 : 0);
            if (p.gt_off >= 0 && !s->noise_table) {
                const int rc = snsde_time_table_srk_launch(s->params, s->srk_tab, ws + p.gt_off, net, p.H,
                                                           s->model.noise_option, s->n_steps * 4, stream);
                if (rc) return rc;
            }
        }
    } else if (s->z0_weight) {
        const int rc = snsde_z0_launch(s, stream);
        if (rc) return rc;
    }
    if (w4) return snsde_w4_launch(s, net, stream);      // (training: the prepare launch above keeps the workspace as the adjoint expects it)
    MfmaArgs a{};
    a.params = s->params; a.ws = ws; a.coeffs = s->coeffs; a.step_tab = s->step_tab; a.out_step = s->out_step;
    a.out_w = s->out_w; a.y0 = s->y0; a.dW = s->dW; a.ys = s->ys; a.traj = s->traj; a.dW_out = s->dW_out;
    a.act_save = s->act_save; a.row_out = s->row_out;
    a.stage_save = p.SRK ? s->stage_save : nullptr;
    if (p.SRK) {
        if (s->dW && !s->dU) return SNSDE_ERR_NULL;
        a.step_tab = ws + p.srk_tab_off; a.dU = s->dU; a.dU_out = s->dU_out;
        a.act = s->model.activation; a.f_out = s->model.drift_output; a.g_out = s->model.diffusion_output;
        a.raw_time = s->model.time_feature; a.gt_ext = s->noise_table;       // (N, 4, H): the table at the four stage times
    }
    a.row_offset = s->row_offset; a.seed = s->seed; a.seed_dev = s->seed_dev;
    a.acc_col = s->kl_column1 - 1; a.acc_a = s->kl_prior_a; a.acc_b = s->kl_prior_b;
    a.B = s->batch; a.L = s->knots; a.C = s->model.input_channels; a.N = s->n_steps; a.T = s->n_out;
    a.method = s->method; a.no = s->model.noise_option;
    a.off_theta = net.off_theta; a.gt_off = p.gt_off; a.bias_off = p.bias_off;
    for (int i = 0; i < p.n_layers; ++i) a.w_off[i] = p.layer[i].dst;
    if (p.M4N) {
        a.lean_geo = (p.IO == 5 || p.IO == 6) ? 1 : 0;
        a.act = s->model.activation; a.f_out = s->model.drift_output; a.g_out = s->model.diffusion_output;
        a.raw_time = s->model.time_feature;
        if (p.H == 128) return dispatch_m4n_h128(p, a, stream);
        if (p.H == 64) return dispatch_m4n_h64(p, a, stream);
        if (p.H == 32) return dispatch_m4n_h32(p, a, stream);
        if (p.H == 16) return dispatch_m4n_h16(p, a, stream);
        return SNSDE_ERR_UNSUPPORTED;
    }
    if (p.LEAN) {
        const int io = p.IO;
        a.lean_xc = (io == 0 || io == 2 || io == 4 || io == 6) ? s->model.input_channels : 0;
        a.lean_time = io >= 3 ? 1 : 0;
        a.lean_geo = (io == 5 || io == 6) ? 1 : 0;
        a.act = s->model.activation; a.f_out = s->model.drift_output; a.g_out = s->model.diffusion_output;
        a.raw_time = s->model.time_feature; a.gt_ext = s->noise_table;
        if (p.H == 256) return dispatch_lean_h256(p, a, stream, (s->flags & SNSDE_FLAG_STREAM_ALL) != 0);
        if (p.H == 128 && (s->flags & SNSDE_FLAG_TWO_TILE)) {
            const int rc = dispatch_lean_h128_two_tile(p, a, stream);
            if (rc != SNSDE_ERR_UNSUPPORTED) return rc;
        }
        if (p.H == 128) return dispatch_lean_h128(p, a, stream);
        if (p.H == 64) return dispatch_lean_h64(p, a, stream);
        if (p.H == 32) return dispatch_lean_h32(p, a, stream);
        return SNSDE_ERR_UNSUPPORTED;
    }
    if (p.H == 256) return p.FL ? dispatch_fwd_m4_h256(p, a, stream) : dispatch_fwd_m16_h256(p, a, stream);
    if (p.H == 128) return p.FL ? dispatch_fwd_m4_h128(p, a, stream) : dispatch_fwd_m16_h128(p, a, stream);
    if (p.H == 64) return p.FL ? dispatch_fwd_m4_h64(p, a, stream) : dispatch_fwd_m16_h64(p, a, stream);
    if (p.H == 32) return p.FL ? dispatch_fwd_m4_h32(p, a, stream) : dispatch_fwd_m16_h32(p, a, stream);
    if (p.H == 16) return p.FL ? dispatch_fwd_m4_h16(p, a, stream) : dispatch_fwd_m16_h16(p, a, stream);
    return SNSDE_ERR_UNSUPPORTED;
}


// ---- backward host side ---------------------------------------------------------------------------------

const float* snsde_mfma_srk_pass_table(const snsde_solve* s, const SnsdeNet& net) {
    MfmaPlan p = make_plan(s, net, -1);
    return (p.ok && p.SRK && s->workspace) ? static_cast<const float*>(s->workspace) + p.srk_tab_off : nullptr;
}

const float* snsde_mfma_gt_table(const snsde_solve* s, const SnsdeNet& net) {
    if (s->noise_table) return s->noise_table;
    MfmaPlan p = make_plan(s, net, -1);
    return (p.ok && p.gt_off >= 0 && s->workspace) ? static_cast<const float*>(s->workspace) + p.gt_off : nullptr;
}

bool snsde_mfma_backward_supported(const snsde_solve* s, const SnsdeNet& net) {
    return make_rev_plan(s, net, make_plan(s, net, variant_of(s) ? 1 : -1)).ok;
}

static int flavor_hint_of(const snsde_solve* s) {
    if (variant_of(s)) return 1;      // tutorial-style fields: 4-row tiles only (snsde_solve_forward launches them that way)
    return s->kernel == SNSDE_KERNEL_MFMA_M16 ? 0 : (s->kernel == SNSDE_KERNEL_MFMA_M4 ? 1 : (s->kernel == SNSDE_KERNEL_MFMA_W4 ? 2 : -1));
}

bool snsde_mfma_backward_partials(const snsde_solve* s, const SnsdeNet& net, int* nwg, int* waves, size_t* ds_off,
                                  size_t* dth_off) {
    RevPlan p = make_rev_plan(s, net, make_plan(s, net, flavor_hint_of(s)));
    if (!p.ok || p.dth_off == 0) return false;
    *nwg = p.nwg; *waves = p.NW; *ds_off = p.ds_off; *dth_off = p.dth_off;
    return true;
}

static size_t w4_gpart_off(const RevPlan& p) { return ((size_t)p.total_floats + 63) & ~(size_t)63; }

size_t snsde_mfma_backward_workspace_floats(const snsde_solve* s, const SnsdeNet& net) {
    const int hint = flavor_hint_of(s);
    const MfmaPlan fp = make_plan(s, net, hint);
    RevPlan p = make_rev_plan(s, net, fp);
    if (!p.ok) return 0;
    if (w4_rev_takes(s, net, fp, p, hint)) return w4_gpart_off(p) + snsde_w4_grad_floats(s);
    return (size_t)p.total_floats;
}

bool snsde_mfma_w4_fused(const snsde_backward* b, const SnsdeNet& net, size_t* gpart_off, size_t* dth_off) {
    return snsde_mfma_w4_fused_solve(&b->fwd, net, gpart_off, dth_off);
}

bool snsde_mfma_w4_fused_solve(const snsde_solve* s, const SnsdeNet& net, size_t* gpart_off, size_t* dth_off) {
    const int hint = flavor_hint_of(s);
    const MfmaPlan fp = make_plan(s, net, hint);
    const RevPlan p = make_rev_plan(s, net, fp);
    if (!w4_rev_takes(s, net, fp, p, hint)) return false;
    if (gpart_off) *gpart_off = w4_gpart_off(p);
    if (dth_off) *dth_off = p.dth_off;
    return true;
}

int snsde_mfma_backward_launch(const snsde_backward* b, const SnsdeNet& net, hipStream_t stream) {
    const snsde_solve* s = &b->fwd;
    const int hint = flavor_hint_of(s);
    MfmaPlan fp = make_plan(s, net, hint);
    RevPlan p = make_rev_plan(s, net, fp);
    if (!p.ok) return SNSDE_ERR_UNSUPPORTED;
    if ((uint64_t)16 * (uint64_t)s->batch * (uint64_t)s->model.hidden_channels >= (1ull << 32)) return SNSDE_ERR_UNSUPPORTED;   // (uoff)
    float* ws = static_cast<float*>(b->workspace);
    if (w4_rev_takes(s, net, fp, p, hint)) {
        // the wave-pair adjoint reads the nn.Linear layout of `params` itself (columns of the weights): no fold, no pack launch;
        // its gradient waves leave the weight-gradient sums per tile behind the plan's own workspace (snsde_mfma_w4_fused)
        if (!(s->dW_out ? s->dW_out : s->dW) && s->seed_dev) return SNSDE_ERR_NULL;
        return snsde_w4_rev_launch(b, net, p.dth_off ? ws + p.dth_off : nullptr, ws + w4_gpart_off(p), stream);
    }
    // first_y = emb[:, 0:H] . linear_in (all columns; the pack step picks the y columns).  A forward that ran with the folded first
    // layer left exactly this product in ITS workspace (piece `in` of snsde_prepare_kernel, same (H, K_in) layout): the pack
    // reads it there and the backward needs no fold launch of its own (6.4 us + a launch gap per K2 training step)
    const float* fold_src = nullptr;
    int fwd_fold_tmp = -1;
    if (p.emb && fp.FOLD && s->workspace) {
        for (int i = 0; i < fp.n_layers; ++i)
            if (fp.layer[i].fold && fp.layer[i].src_w == net.in.src_w && !fp.layer[i].transpose) fwd_fold_tmp = fp.layer[i].fold_tmp;
        if (fwd_fold_tmp >= 0) fold_src = static_cast<const float*>(s->workspace);
    }
    if (p.emb && !fold_src) {
        FoldJob fj{};
        fj.emb_w = net.emb.src_w; fj.H = p.H; fj.n_pieces = 1; fj.fold_on = 1;
        fj.src_w[0] = net.in.src_w; fj.K[0] = net.in.K; fj.col[0] = 0; fj.tmp[0] = p.fold_tmp;
        fj.b_in = net.in.src_b; fj.b_init = net.init.src_b; fj.b_emb = net.emb.src_b;
        fj.bias_tmp = p.fold_tmp + p.H * net.in.K;
        hipLaunchKernelGGL(snsde_fold_kernel, dim3(p.H, 1), dim3(256), 2 * p.H * sizeof(float), stream, s->params, ws, fj);
    }
    MfmaPackJob job{};
    for (int i = 0; i < p.n_layers; ++i) {
        job.layer[i] = p.layer[i];
        if (fold_src && job.layer[i].fold) job.layer[i].fold_tmp = fwd_fold_tmp;
    }
    job.n_layers = p.n_layers; job.flavor = p.FL; job.TPW = 1; job.NW = p.NW; job.bias_off = 0; job.H = p.H;
    hipLaunchKernelGGL(snsde_mfma_pack_kernel, dim3(16, p.n_layers), dim3(256), 0, stream, s->params, ws, fold_src, job);
    RevArgs a{};
    a.params = s->params; a.ws = ws;
    a.gt = s->noise_table ? s->noise_table : (fp.gt_off >= 0 ? static_cast<const float*>(s->workspace) + fp.gt_off : nullptr);
    a.act_fn = s->model.activation; a.f_out = s->model.drift_output; a.g_out = s->model.diffusion_output;
    a.nsave = s->model.num_hidden_layers + 1 + fp.NN + (s->model.activation != SNSDE_ACT_RELU ? s->model.num_hidden_layers + (fp.NN == 2 ? 1 : 0) : 0);
    if (s->method == SNSDE_SRK && fp.NN > 0)      // (snsde_save_layout: the fourth evaluation's net slots, smooth: + its hidden pre-activation)
        a.nsave += fp.NN + ((fp.NN == 2 && s->model.activation != SNSDE_ACT_RELU) ? 1 : 0);
    a.step_tab = s->step_tab; a.out_w = s->out_w; a.traj = s->traj; a.act = s->act_save;
    // increments: the ones the forward wrote out, else the supplied ones, else (Philox, host key) regenerated by the kernel
    a.dW = s->dW_out ? s->dW_out : s->dW;
    a.seed = s->seed; a.row_offset = s->row_offset;
    if (!a.dW && (p.SRK || p.M4N || s->seed_dev)) return SNSDE_ERR_NULL;      // (regeneration: the Euler / Milstein kernel, host key)
    a.grad_ys = b->grad_ys; a.adj = b->adj; a.delta = b->delta_save; a.row_out = s->row_out;
    a.adj0_only = (b->flags & SNSDE_BWD_ADJ0_ONLY) ? 1 : 0;
    a.acc_col = s->kl_column1 - 1; a.acc_a = s->kl_prior_a; a.acc_b = s->kl_prior_b;
    if (a.acc_col >= 0 && (p.FL != 1 || p.M4N || p.H > 128)) return SNSDE_ERR_UNSUPPORTED;      // (the 4-row-tile Euler / SRK adjoints carry it)
    if (a.adj0_only && p.M4N == 2) return SNSDE_ERR_OPTION;      // (its weight-gradient jobs read every a_n)
    if (p.SRK) {
        if (!s->dU_out) return SNSDE_ERR_NULL;
        a.dU = s->dU_out;
    }
    a.ds_part = p.ds_off ? ws + p.ds_off : nullptr;
    a.dth_part = p.dth_off ? ws + p.dth_off : nullptr;
    a.B = s->batch; a.N = s->n_steps; a.T = s->n_out; a.no = s->model.noise_option; a.off_theta = net.off_theta; a.method = s->method;
    for (int i = 0; i < p.n_layers; ++i) a.w_off[i] = p.layer[i].dst;
    if (p.M4N == 2) {
        a.geo = p.GEO;
        if (p.H == 128) return dispatch_m4n_mil_rev_h128(p, a, stream);
        if (p.H == 64) return dispatch_m4n_mil_rev_h64(p, a, stream);
        if (p.H == 32) return dispatch_m4n_mil_rev_h32(p, a, stream);
        if (p.H == 16) return dispatch_m4n_mil_rev_h16(p, a, stream);
        return SNSDE_ERR_UNSUPPORTED;
    }
    if (p.M4N) {
        a.geo = p.GEO;
        if (p.H == 128) return dispatch_m4n_rev_h128(p, a, stream);
        if (p.H == 64) return dispatch_m4n_rev_h64(p, a, stream);
        if (p.H == 32) return dispatch_m4n_rev_h32(p, a, stream);
        if (p.H == 16) return dispatch_m4n_rev_h16(p, a, stream);
        return SNSDE_ERR_UNSUPPORTED;
    }
    if (p.H == 256 && !(s->flags & SNSDE_FLAG_STREAM_ALL)) {      // two tiles per wave, a quarter of the transposed weights resident (round 6)
        const int rc = dispatch_rev_h256_two_tile(p, a, stream);
        if (rc != SNSDE_ERR_UNSUPPORTED) return rc;
    }
    if (p.H == 256) return dispatch_rev_h256(p, a, stream);
    if (p.H == 128) return dispatch_rev_h128(p, a, stream);
    if (p.H == 64) return dispatch_rev_h64(p, a, stream);
    if (p.H == 32) return dispatch_rev_h32(p, a, stream);
    if (p.H == 16) return dispatch_rev_h16(p, a, stream);
    return SNSDE_ERR_UNSUPPORTED;
}
