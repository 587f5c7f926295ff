// Host side of the wave-owns-rows kernels (snsde_w4_kernel.h: forward Euler / SRK, their adjoints with the weight gradients inside, the
// reduction of the per-tile gradient blocks): which solves they take, argument set-up, dispatch.
#include "snsde_w4_kernel.h"

namespace snsde_w4 {

static bool shape_ok(const snsde_solve* s, const SnsdeNet& net) {
    const snsde_model& m = s->model;
    const int io = m.input_option, no = m.noise_option;
    if (m.hidden_channels != 64 || m.hidden_hidden_channels != 64) return false;
    if (!(io == 1 || io == 3 || io == 5)) return false;                      // latent-only drifts (no control path in f)
    if (!(no == 14 || no == 15 || no == 18 || no == 19)) return false;       // diffusion nets
    if (s->method != SNSDE_EULER && s->method != SNSDE_SRK) return false;
    if (m.activation != 0 || m.drift_output != 0 || m.diffusion_output != 0 || m.time_feature != 0 || s->noise_table) return false;
    if (m.num_hidden_layers < 1 || m.num_hidden_layers > 2) return false;      // (two hidden layers: 258 weight registers, spills)
    if (s->kl_column1 != 0 || s->batch < 4) return false;      // (a tile is four rows; ragged tails overlap the previous tile)
    if ((uint64_t)16 * (uint64_t)s->batch * 64u >= (1ull << 32)) return false;   // 32-bit save offsets (uoff)
    if (net.ny0.K != 66 || net.in.K != (io >= 3 ? 66 : 64)) return false;
    return true;
}

template <int NHID, int NN, bool TIME>
static int launch(const W4Args& a, hipStream_t st) {
    const dim3 grid((a.B + 7) / 8), block(256);
    if (a.srk_tab) {
        if (a.act_save) {      // (training mode parks the drift wave's first matrix: more than 64 KB of dynamic LDS)
            const size_t lds_bytes = (size_t)w4srk_fwd_lds_floats<NHID, true>() * sizeof(float);
            static SnsdeLdsAttr lds_attr;   // per instantiation and device
            void (*kern)(W4Args) = snsde_w4_srk_kernel<CfgW<NHID, NN, TIME, true>>;
            if (const int rc = snsde_lds_attr(reinterpret_cast<const void*>(kern), lds_bytes, lds_attr)) return rc;
            hipLaunchKernelGGL(kern, grid, block, lds_bytes, st, a);
        } else {
            constexpr size_t lds_bytes = (size_t)w4srk_fwd_lds_floats<NHID, false>() * sizeof(float);
            hipLaunchKernelGGL((snsde_w4_srk_kernel<CfgW<NHID, NN, TIME, false>>), grid, block, lds_bytes, st, a);
        }
    } else if (a.act_save) hipLaunchKernelGGL((snsde_w4_euler_kernel<CfgW<NHID, NN, TIME, true>>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((snsde_w4_euler_kernel<CfgW<NHID, NN, TIME, false>>), grid, block, 0, st, a);
    return hipGetLastError() == hipSuccess ? SNSDE_OK : SNSDE_ERR_LAUNCH;
}

// adjoint of the SRID2 solve with the weight gradients inside (snsde_w4_srk_reverse_kernel); gpart is mandatory: there is no
// delta-plane variant of this kernel
template <int NHID, int NN, bool GEO, bool MULY>
static int launch_srk_rev(const W4SrkRevArgs& a, hipStream_t st) {
    using CF = CfgSR<NHID, NN, GEO, MULY>;
    const size_t lds_bytes = (size_t)w4srk_rev_lds_floats<NHID, NN, MULY>() * sizeof(float);
    static SnsdeLdsAttr lds_attr;   // per instantiation and device
    if (const int rc = snsde_lds_attr(reinterpret_cast<const void*>(snsde_w4_srk_reverse_kernel<CF>), lds_bytes, lds_attr)) return rc;
    hipLaunchKernelGGL((snsde_w4_srk_reverse_kernel<CF>), dim3((a.B + 7) / 8), dim3(512), lds_bytes, st, a);
    return hipGetLastError() == hipSuccess ? SNSDE_OK : SNSDE_ERR_LAUNCH;
}

static int srk_rev_launch(const snsde_backward* b, const SnsdeNet& net, float* dth_part, float* gpart, hipStream_t stream) {
    const snsde_solve* s = &b->fwd;
    const snsde_model& m = s->model;
    const float* ik = s->dW_out ? s->dW_out : s->dW;
    const float* ik0 = s->dU_out ? s->dU_out : s->dU;
    if (!gpart) return SNSDE_ERR_UNSUPPORTED;
    if (!s->stage_save || !s->srk_tab || !ik || !ik0) return SNSDE_ERR_NULL;
    W4SrkRevArgs a{};
    a.params = s->params; a.step_tab = s->step_tab; a.srk_tab = s->srk_tab; a.out_w = s->out_w; a.act = s->act_save; a.stage = s->stage_save;
    a.dW = ik; a.dU = ik0; a.grad_ys = b->grad_ys; a.adj = b->adj; a.dth_part = dth_part; a.gpart = gpart; a.row_out = s->row_out;
    a.B = s->batch; a.N = s->n_steps; a.T = s->n_out; a.no = m.noise_option; a.geo = m.input_option == 5 ? 1 : 0;
    const int nn = (m.noise_option >= 18) ? 2 : 1;
    a.nsave = snsde_act_slots(&m) + nn;      // (snsde_save_layout: the fourth evaluation's net slots)
    a.adj0_only = (b->flags & SNSDE_BWD_ADJ0_ONLY) ? 1 : 0; a.off_theta = net.off_theta;
    a.w_in = net.in.src_w; a.k_in = net.in.K; a.t_in = net.in.tshift;
    const int nhid = m.num_hidden_layers - 1;
    for (int l = 0; l < nhid; ++l) a.w_hid[l] = net.hid[l].src_w;
    a.w_out = net.out.src_w; a.w_n0 = net.ny0.src_w; a.w_n1 = net.ny1.src_w;
    const bool muly = m.noise_option == 15 || m.noise_option == 19;
#define W4SR_CASE(NH, N2) if (nhid == NH && nn == N2) { \
        if (a.geo) return muly ? launch_srk_rev<NH, N2, true, true>(a, stream) : launch_srk_rev<NH, N2, true, false>(a, stream); \
        return muly ? launch_srk_rev<NH, N2, false, true>(a, stream) : launch_srk_rev<NH, N2, false, false>(a, stream); }
    W4SR_CASE(0, 1) W4SR_CASE(0, 2) W4SR_CASE(1, 1) W4SR_CASE(1, 2)
#undef W4SR_CASE
    return SNSDE_ERR_UNSUPPORTED;
}

}  // namespace snsde_w4

bool snsde_w4_supported(const snsde_solve* s, const SnsdeNet& net) { return snsde_w4::shape_ok(s, net); }

int snsde_w4_launch(const snsde_solve* s, const SnsdeNet& net, hipStream_t stream) {
    using namespace snsde_w4;
    if (!shape_ok(s, net)) return SNSDE_ERR_UNSUPPORTED;
    if (s->method == SNSDE_SRK && (!s->srk_tab || (s->dW && !s->dU))) return SNSDE_ERR_NULL;
    const snsde_model& m = s->model;
    W4Args a{};
    a.params = s->params; a.step_tab = s->step_tab; a.out_w = s->out_w; a.y0 = s->y0; a.dW = s->dW; a.ys = s->ys; a.traj = s->traj;
    a.dW_out = s->dW_out; a.act_save = s->act_save; a.row_out = s->row_out; a.row_offset = s->row_offset; a.seed = s->seed;
    a.seed_dev = s->seed_dev;
    a.B = s->batch; a.N = s->n_steps; a.T = s->n_out; a.no = m.noise_option; a.geo = m.input_option == 5 ? 1 : 0;
    a.nsave = snsde_act_slots(&m);
    if (s->method == SNSDE_SRK) { a.srk_tab = s->srk_tab; a.dU = s->dU; a.dU_out = s->dU_out; a.stage_save = s->stage_save; }
    a.off_theta = net.off_theta;
    a.w_in = net.in.src_w; a.b_in = net.in.src_b; a.k_in = net.in.K; a.t_in = net.in.tshift;
    const int nhid = m.num_hidden_layers - 1;
    for (int l = 0; l < nhid; ++l) { a.w_hid[l] = net.hid[l].src_w; a.b_hid[l] = net.hid[l].src_b; }
    a.w_out = net.out.src_w; a.b_out = net.out.src_b;
    a.w_n0 = net.ny0.src_w; a.b_n0 = net.ny0.src_b; a.w_n1 = net.ny1.src_w; a.b_n1 = net.ny1.src_b;
    const int nn = (m.noise_option >= 18) ? 2 : 1;
    const bool timef = m.input_option >= 3;
#define W4_CASE(NH, N2, TM) if (nhid == NH && nn == N2 && timef == TM) return launch<NH, N2, TM>(a, stream);
    W4_CASE(0, 1, false) W4_CASE(0, 1, true) W4_CASE(0, 2, false) W4_CASE(0, 2, true)
    W4_CASE(1, 1, false) W4_CASE(1, 1, true) W4_CASE(1, 2, false) W4_CASE(1, 2, true)
#undef W4_CASE
    return SNSDE_ERR_UNSUPPORTED;
}

// ---- adjoint of the Euler solve (snsde_w4_euler_reverse_kernel) --------------------------------------------------------------
bool snsde_w4_rev_supported(const snsde_solve* s, const SnsdeNet& net) {
    // Euler: snsde_w4_euler_reverse_kernel (regenerates host-keyed Philox increments); SRK: snsde_w4_srk_reverse_kernel (reads the
    // increments and the stage states the forward wrote)
    // (a device-resident Philox key - graph replays - cannot be regenerated from: the forward then leaves its increments in dW_out,
    //  snsde_mfma_backward_launch refuses the launch without them)
    return (s->method == SNSDE_EULER || s->method == SNSDE_SRK) && snsde_w4::shape_ok(s, net);
}

size_t snsde_w4_grad_floats(const snsde_solve* s) {      // per-tile gradient blocks + the stage-1 sums of the reduction
    const int nhid = s->model.num_hidden_layers - 1, nn = s->model.noise_option >= 18 ? 2 : 1;
    const size_t block = (size_t)snsde_w4::w4g_block_floats(nhid, nn), tiles = (size_t)(s->batch + 3) / 4;
    return (tiles + 16) * block;
}

int snsde_w4_rev_launch(const snsde_backward* b, const SnsdeNet& net, float* dth_part, float* gpart, hipStream_t stream) {
    using namespace snsde_w4;
    const snsde_solve* s = &b->fwd;
    if (!snsde_w4_rev_supported(s, net)) return SNSDE_ERR_UNSUPPORTED;
    if (!s->act_save || !b->grad_ys || !b->adj) return SNSDE_ERR_NULL;
    const snsde_model& m = s->model;
    if (s->method == SNSDE_SRK) return snsde_w4::srk_rev_launch(b, net, dth_part, gpart, stream);
    if (!s->traj) return SNSDE_ERR_NULL;
    W4RevArgs a{};
    a.params = s->params; a.step_tab = s->step_tab; a.out_w = s->out_w; a.traj = s->traj; a.act = s->act_save;
    a.dW = s->dW_out ? s->dW_out : s->dW;
    a.grad_ys = b->grad_ys; a.adj = b->adj; a.dth_part = dth_part; a.row_out = s->row_out;
    a.gpart = gpart;
    a.delta = gpart ? nullptr : b->delta_save;      // fused weight gradients: no delta planes are written
    a.seed = s->seed; a.row_offset = s->row_offset;
    a.B = s->batch; a.N = s->n_steps; a.T = s->n_out; a.no = m.noise_option; a.geo = m.input_option == 5 ? 1 : 0;
    a.nsave = snsde_act_slots(&m); a.nslots = a.nsave;
    a.adj0_only = (b->flags & SNSDE_BWD_ADJ0_ONLY) ? 1 : 0; a.off_theta = net.off_theta;
    a.w_in = net.in.src_w; a.k_in = net.in.K; a.t_in = net.in.tshift;
    const int nhid = m.num_hidden_layers - 1;
    for (int l = 0; l < nhid; ++l) a.w_hid[l] = net.hid[l].src_w;
    a.w_out = net.out.src_w; a.w_n0 = net.ny0.src_w; a.w_n1 = net.ny1.src_w;
    const int nn = (m.noise_option >= 18) ? 2 : 1;
    const dim3 grid((a.B + 7) / 8);
#define W4R_CASE(NH, N2) if (nhid == NH && nn == N2) { \
        if (gpart) hipLaunchKernelGGL((snsde_w4_euler_reverse_kernel<CfgW<NH, N2, false, false>, true>), grid, dim3(512), 0, stream, a); \
        else hipLaunchKernelGGL((snsde_w4_euler_reverse_kernel<CfgW<NH, N2, false, false>, false>), grid, dim3(256), 0, stream, a); \
        return hipGetLastError() == hipSuccess ? SNSDE_OK : SNSDE_ERR_LAUNCH; }
    W4R_CASE(0, 1) W4R_CASE(0, 2) W4R_CASE(1, 1) W4R_CASE(1, 2)
#undef W4R_CASE
    return SNSDE_ERR_UNSUPPORTED;
}

// dL/d params from the per-tile blocks of the fused adjoint: zero fill, two reduction launches (deterministic association)
int snsde_w4_grad_reduce_launch(const snsde_backward* b, const SnsdeNet& net, float* grad_params, int32_t n_params, float* gpart,
                                const float* dth_part, hipStream_t stream) {
    using namespace snsde_w4;
    const snsde_solve* s = &b->fwd;
    const snsde_model& m = s->model;
    const int nhid = m.num_hidden_layers - 1, nn = m.noise_option >= 18 ? 2 : 1, nd = nhid + 2;
    W4GReduce r{};
    r.tiles = (s->batch + 3) / 4; r.block = w4g_block_floats(nhid, nn); r.nsplit = r.tiles < 16 ? r.tiles : 16;
    r.gpart = gpart; r.part2 = gpart + (size_t)r.tiles * r.block; r.grad = grad_params; r.dth_part = dth_part; r.params = s->params;
    r.off_theta = net.off_theta; r.n_dth = r.tiles * 4;
    int n = 0;
    auto weight = [&](int src, const SnsdeLayer& L, int col) { r.seg[n++] = W4GSeg{src, L.src_w, L.K, col, 0}; r.seg[n++] = W4GSeg{src + 4096, L.src_b, 0, 0, 1}; };
    auto tcols = [&](int src, const SnsdeLayer& L) { r.seg[n++] = W4GSeg{src, L.src_w, L.K, 0, 2}; r.seg[n++] = W4GSeg{src + 64, L.src_w, L.K, 1, 2}; };
    weight(0, net.out, 0);
    for (int g = 1; g <= nhid; ++g) weight(g * w4g_layer_floats(), net.hid[nhid - g], 0);
    weight((nd - 1) * w4g_layer_floats(), net.in, net.in.tshift);
    if (net.in.tshift == 2) tcols(w4g_d_time(nhid), net.in);
    if (nn == 2) { weight(w4g_n_off(nhid, 0), net.ny1, 0); weight(w4g_n_off(nhid, 1), net.ny0, 2); }
    else weight(w4g_n_off(nhid, 0), net.ny0, 2);
    tcols(w4g_n_time(nhid, nn), net.ny0);
    r.nseg = n;
    if (hipMemsetAsync(grad_params, 0, (size_t)n_params * sizeof(float), stream) != hipSuccess) return SNSDE_ERR_LAUNCH;
    const int gx = (r.block + 255) / 256;
    hipLaunchKernelGGL(snsde_w4_grad_reduce1_kernel, dim3(gx, r.nsplit), dim3(256), 0, stream, r);
    hipLaunchKernelGGL(snsde_w4_grad_reduce2_kernel, dim3(gx + 1), dim3(256), 0, stream, r);
    return hipGetLastError() == hipSuccess ? SNSDE_OK : SNSDE_ERR_LAUNCH;
}
