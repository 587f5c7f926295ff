// C-ABI entry points of libsnsde.so (include/snsde.h): parameter layout, torchsde-style fixed-step
// time grid (host), argument validation and kernel dispatch.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "snsde_internal.h"

namespace {

struct ParamEntry {
    char name[40];
    int64_t offset;
    int32_t rows, cols;  // bias / vectors: rows = n, cols = 0 ; theta: (1,1)
};

constexpr int MAX_PARAMS = 2 * (SNSDE_MAX_HIDDEN + 8) + 4;

int validate_model(const snsde_model* m) {
    if (!m) return SNSDE_ERR_NULL;
    if (m->input_channels <= 0 || m->hidden_channels <= 0 || m->hidden_hidden_channels <= 0 ||
        m->num_hidden_layers <= 0)
        return SNSDE_ERR_DIMS;
    if (m->num_hidden_layers - 1 > SNSDE_MAX_HIDDEN) return SNSDE_ERR_UNSUPPORTED;
    if (m->input_option < 0 || m->input_option > 6 || m->noise_option < 0 || m->noise_option > 19)
        return SNSDE_ERR_OPTION;
    if (m->activation < 0 || m->activation > SNSDE_ACT_SILU || m->drift_output < 0 || m->drift_output > SNSDE_DRIFT_TIMES_Y ||
        m->diffusion_output < 0 || m->diffusion_output > SNSDE_DIFFUSION_RAW_NET || m->time_feature < 0 ||
        m->time_feature > SNSDE_TIME_RAW)
        return SNSDE_ERR_OPTION;
    if (m->diffusion_output == SNSDE_DIFFUSION_RAW_NET && m->noise_option != 18 && m->noise_option != 19) return SNSDE_ERR_OPTION;
    const int io = m->input_option;
    // emb = Linear(2H, H) consumes cat[yy (HH), Xt (H)] and io 0 feeds Xt (H) to the HH-wide MLP:
    // both need HH == H (neuralsde.py:150-158, 206-210)
    if ((io == 0 || io == 2 || io == 4 || io == 6) && m->hidden_hidden_channels != m->hidden_channels)
        return SNSDE_ERR_DIMS;
    return SNSDE_OK;
}

// state_dict order of the reference Diffusion_model (neuralsde.py:142-179): direct parameters
// (theta, sigma, sigma_diag) first, then the sub-modules in definition order.
int build_params(const snsde_model* m, ParamEntry* e, int* count, int64_t* total) {
    int rc = validate_model(m);
    if (rc) return rc;
    const int C = m->input_channels, H = m->hidden_channels, HH = m->hidden_hidden_channels;
    const int io = m->input_option, no = m->noise_option;
    int n = 0;
    int64_t off = 0;
    auto vec = [&](const char* name, int len) {
        snprintf(e[n].name, sizeof(e[n].name), "%s", name);
        e[n].offset = off; e[n].rows = len; e[n].cols = 0; off += len; ++n;
    };
    auto lin = [&](const char* name, int rows, int cols) {
        snprintf(e[n].name, sizeof(e[n].name), "%s.weight", name);
        e[n].offset = off; e[n].rows = rows; e[n].cols = cols; off += (int64_t)rows * cols; ++n;
        snprintf(e[n].name, sizeof(e[n].name), "%s.bias", name);
        e[n].offset = off; e[n].rows = rows; e[n].cols = 0; off += rows; ++n;
    };
    snprintf(e[n].name, sizeof(e[n].name), "theta");
    e[n].offset = off; e[n].rows = 1; e[n].cols = 1; off += 1; ++n;
    if (no >= 1 && no <= 3) vec("sigma", 1);
    if (no >= 4 && no <= 6) vec("sigma_diag", H);
    lin("initial_network", H, C);
    lin("linear_in", HH, (io >= 3) ? H + 2 : H);
    if (io == 2 || io == 4 || io == 6) lin("emb", H, 2 * H);
    for (int i = 0; i < m->num_hidden_layers - 1; ++i) {
        char nm[32];
        snprintf(nm, sizeof(nm), "linears.%d", i);
        lin(nm, HH, HH);
    }
    lin("linear_out", H, HH);
    if (no == 12 || no == 13) lin("noise_t", H, 2);
    if (no == 14 || no == 15) lin("noise_y", H, H + 2);
    if (no == 16 || no == 17) { lin("noise_t.0", H, 2); lin("noise_t.2", H, H); }
    if (no == 18 || no == 19) { lin("noise_y.0", H, H + 2); lin("noise_y.2", H, H); }
    *count = n;
    *total = off;
    return SNSDE_OK;
}

int find(const ParamEntry* e, int n, const char* name) {
    for (int i = 0; i < n; ++i)
        if (strcmp(e[i].name, name) == 0) return i;
    return -1;
}

}  // namespace

int snsde_build_net(const snsde_model& m, int32_t n_steps, SnsdeNet* net) {
    ParamEntry e[MAX_PARAMS];
    int n = 0;
    int64_t total = 0;
    int rc = build_params(&m, e, &n, &total);
    if (rc) return rc;
    if (total > 0x7fffffffLL) return SNSDE_ERR_DIMS;
    memset(net, 0, sizeof(*net));
    const int io = m.input_option, no = m.noise_option;
    int32_t woff = 0;
    auto fill = [&](SnsdeLayer& L, const char* name, int tshift, bool packed) {
        char w[48], b[48];
        snprintf(w, sizeof(w), "%s.weight", name);
        snprintf(b, sizeof(b), "%s.bias", name);
        const int iw = find(e, n, w), ib = find(e, n, b);
        if (iw < 0) { L.present = 0; L.w = -1; return; }
        L.present = 1;
        L.src_w = (int32_t)e[iw].offset;
        L.src_b = (int32_t)e[ib].offset;
        L.N = e[iw].rows;
        L.K = e[iw].cols;
        L.Kpad = (L.K + 3) & ~3;
        L.tshift = tshift;
        if (packed) { L.w = woff; woff += L.Kpad * L.N; } else { L.w = -1; }
    };
    const bool uses_x = (io == 0 || io == 2 || io == 4 || io == 6);
    fill(net->init, "initial_network", 0, uses_x);
    if (!uses_x) net->init.w = -1;
    fill(net->in, "linear_in", io >= 3 ? 2 : 0, io != 0);
    if (io == 0) net->in.w = -1;
    fill(net->emb, "emb", 0, true);
    net->n_hid = m.num_hidden_layers - 1;
    for (int i = 0; i < net->n_hid; ++i) {
        char nm[32];
        snprintf(nm, sizeof(nm), "linears.%d", i);
        fill(net->hid[i], nm, 0, true);
    }
    fill(net->out, "linear_out", 0, true);
    if (no == 14 || no == 15) fill(net->ny0, "noise_y", 2, true);
    if (no == 18 || no == 19) { fill(net->ny0, "noise_y.0", 2, true); fill(net->ny1, "noise_y.2", 0, true); }
    if (no == 12 || no == 13) fill(net->nt0, "noise_t", 0, false);
    if (no == 16 || no == 17) { fill(net->nt0, "noise_t.0", 0, false); fill(net->nt1, "noise_t.2", 0, false); }
    net->off_theta = (int32_t)e[find(e, n, "theta")].offset;
    int i = find(e, n, "sigma");
    net->off_sigma = i >= 0 ? (int32_t)e[i].offset : -1;
    i = find(e, n, "sigma_diag");
    net->off_sigma_diag = i >= 0 ? (int32_t)e[i].offset : -1;
    net->packed_floats = (woff + 3) & ~3;
    net->gt_tab = (no == 12 || no == 13 || no == 16 || no == 17) ? net->packed_floats : -1;
    (void)n_steps;
    return SNSDE_OK;
}

extern "C" {

int snsde_version(void) { return SNSDE_VERSION; }

int snsde_abi_check(int version, size_t sizeof_model, size_t sizeof_solve, size_t sizeof_backward, size_t sizeof_head) {
    // 0 = "this binding does not declare that struct" (a forward-only binding has no snsde_backward / snsde_head)
    auto same = [](size_t got, size_t want) { return got == 0 || got == want; };
    return (version == SNSDE_VERSION && same(sizeof_model, sizeof(snsde_model)) && same(sizeof_solve, sizeof(snsde_solve)) &&
            same(sizeof_backward, sizeof(snsde_backward)) && same(sizeof_head, sizeof(snsde_head))) ? SNSDE_OK : SNSDE_ERR_ABI;
}

const char* snsde_strerror(int code) {
    switch (code) {
        case SNSDE_OK: return "ok";
        case SNSDE_ERR_NULL: return "required pointer is NULL";
        case SNSDE_ERR_DIMS: return "bad or inconsistent dimensions";
        case SNSDE_ERR_OPTION: return "input_option must be 0..6 and noise_option 0..19";
        case SNSDE_ERR_UNSUPPORTED: return "configuration not supported by this build";
        case SNSDE_ERR_WORKSPACE: return "workspace too small";
        case SNSDE_ERR_LDS: return "configuration exceeds the LDS budget";
        case SNSDE_ERR_TS: return "ts must be strictly increasing, dt > 0 and representable progress in float32";
        case SNSDE_ERR_LAUNCH: return "HIP kernel launch failed";
        case SNSDE_ERR_INDEX: return "index out of range";
        case SNSDE_ERR_ABI: return "struct_size / version mismatch: the binding was built against another include/snsde.h";
        default: return "unknown error";
    }
}

int snsde_param_count(const snsde_model* m) {
    ParamEntry e[MAX_PARAMS];
    int n = 0;
    int64_t total = 0;
    int rc = build_params(m, e, &n, &total);
    return rc ? rc : n;
}

int64_t snsde_param_numel(const snsde_model* m) {
    ParamEntry e[MAX_PARAMS];
    int n = 0;
    int64_t total = 0;
    int rc = build_params(m, e, &n, &total);
    return rc ? rc : total;
}

int snsde_param_info(const snsde_model* m, int index, char* name, int name_cap, int64_t* offset, int32_t* rows,
                     int32_t* cols) {
    ParamEntry e[MAX_PARAMS];
    int n = 0;
    int64_t total = 0;
    int rc = build_params(m, e, &n, &total);
    if (rc) return rc;
    if (index < 0 || index >= n) return SNSDE_ERR_INDEX;
    if (name && name_cap > 0) snprintf(name, name_cap, "%s", e[index].name);
    if (offset) *offset = e[index].offset;
    if (rows) *rows = e[index].rows;
    if (cols) *cols = e[index].cols;
    return SNSDE_OK;
}

// torchsde 0.2.5 BaseSDESolver.integrate time bookkeeping in float32.
static int walk_grid(const float* ts, int32_t T, double dt, const float* times, int32_t L, int32_t cap,
                     float* step_tab, int32_t* out_step, float* out_w, int32_t* n_steps) {
    if (!ts) return SNSDE_ERR_NULL;
    if (T < 2) return SNSDE_ERR_TS;
    if (!(dt > 0)) return SNSDE_ERR_TS;
    for (int k = 1; k < T; ++k)
        if (!(ts[k] > ts[k - 1])) return SNSDE_ERR_TS;
    const float step = (float)dt;
    const float t_end = ts[T - 1];
    float curr = ts[0], prev = ts[0];
    int32_t n = 0;
    for (int k = 1; k < T; ++k) {
        const float out_t = ts[k];
        while (curr < out_t) {
            volatile float nx = curr + step;  // force fp32 rounding
            float nxt = nx;
            if (t_end < nxt) nxt = t_end;
            if (!(nxt > curr)) return SNSDE_ERR_TS;
            if (step_tab) {
                if (n >= cap) return SNSDE_ERR_DIMS;
                float* r = step_tab + (size_t)n * SNSDE_STEP_STRIDE;
                volatile float h = nxt - curr;
                r[0] = curr;
                r[1] = h;
                r[2] = sinf(curr);
                r[3] = cosf(curr);
                int idx = 0;
                if (times) {
                    int cnt = 0;
                    for (int j = 0; j < L; ++j) cnt += (curr > times[j]) ? 1 : 0;
                    idx = cnt - 1;
                    if (idx < 0) idx = 0;
                    if (idx > L - 2) idx = L - 2;
                    volatile float fr = curr - times[idx];
                    r[4] = fr;
                } else {
                    r[4] = 0.0f;
                }
                memcpy(&r[5], &idx, sizeof(float));
                r[6] = sqrtf(h);
                r[7] = nxt;
                r[8] = r[9] = r[10] = r[11] = 0.0f;
            }
            prev = curr;
            curr = nxt;
            ++n;
        }
        if (out_step) {
            out_step[k - 1] = n - 1;
            if (step_tab) {   // per-step output bookkeeping: count and first output index
                float* r = step_tab + (size_t)(n - 1) * SNSDE_STEP_STRIDE;
                int32_t cnt, first;
                memcpy(&cnt, &r[8], 4);
                memcpy(&first, &r[9], 4);
                if (cnt == 0) first = k - 1;
                ++cnt;
                memcpy(&r[8], &cnt, 4);
                memcpy(&r[9], &first, 4);
            }
            volatile float denom = curr - prev;
            volatile float a = curr - out_t, b = out_t - prev;
            out_w[2 * (k - 1)] = a / denom;
            out_w[2 * (k - 1) + 1] = b / denom;
        }
    }
    *n_steps = n;
    return SNSDE_OK;
}

int snsde_grid_count(const float* ts, int32_t n_out, double dt, int32_t* n_steps) {
    if (!n_steps) return SNSDE_ERR_NULL;
    return walk_grid(ts, n_out, dt, nullptr, 0, 0, nullptr, nullptr, nullptr, n_steps);
}

int snsde_grid_build(const float* ts, int32_t n_out, double dt, const float* times, int32_t knots, int32_t n_steps,
                     float* step_tab, int32_t* out_step, float* out_w) {
    if (!step_tab || !out_step || !out_w || !times) return SNSDE_ERR_NULL;
    if (knots < 2) return SNSDE_ERR_DIMS;
    int32_t n = 0;
    int rc = walk_grid(ts, n_out, dt, times, knots, n_steps, step_tab, out_step, out_w, &n);
    if (rc) return rc;
    return n == n_steps ? SNSDE_OK : SNSDE_ERR_DIMS;
}

int snsde_grid_srk_build(const float* step_tab, int32_t n_steps, const float* times, int32_t knots, float* srk_tab) {
    if (!step_tab || !times || !srk_tab) return SNSDE_ERR_NULL;
    if (n_steps <= 0 || knots < 2) return SNSDE_ERR_DIMS;
    const float cs[4] = {0.0f, 0.25f, 0.5f, 1.0f};
    for (int n = 0; n < n_steps; ++n) {
        const float t0 = step_tab[(size_t)n * SNSDE_STEP_STRIDE], h = step_tab[(size_t)n * SNSDE_STEP_STRIDE + 1];
        for (int c = 0; c < 4; ++c) {
            volatile float ch = cs[c] * h;
            volatile float t = t0 + ch;
            float* r = srk_tab + ((size_t)n * 4 + c) * SNSDE_SRK_STRIDE;
            int cnt = 0;
            for (int j = 0; j < knots; ++j) cnt += (t > times[j]) ? 1 : 0;
            int idx = cnt - 1;
            if (idx < 0) idx = 0;
            if (idx > knots - 2) idx = knots - 2;
            volatile float fr = t - times[idx];
            r[0] = t; r[1] = sinf(t); r[2] = cosf(t); r[3] = fr;
            memcpy(&r[4], &idx, sizeof(float));
            r[5] = r[6] = r[7] = 0.0f;
        }
    }
    return SNSDE_OK;
}

// field variants beyond the reference's Diffusion_model (tutorial fields): served by the lean 4-row-tile MFMA kernel only
static bool is_variant(const snsde_model& m) {
    return m.activation != 0 || m.drift_output != 0 || m.diffusion_output != 0 || m.time_feature != 0;
}

static int validate_solve(const snsde_solve* s, bool eval) {
    if (!s) return SNSDE_ERR_NULL;
    if (s->struct_size != sizeof(snsde_solve)) return SNSDE_ERR_ABI;      // stale binding: refuse before reading any field
    if (!s) return SNSDE_ERR_NULL;
    int rc = validate_model(&s->model);
    if (rc) return rc;
    if (s->batch <= 0 || s->knots < 2) return SNSDE_ERR_DIMS;
    if (!s->params || !s->coeffs || !s->workspace) return SNSDE_ERR_NULL;
    if (!eval) {
        if (s->n_steps <= 0 || s->n_out < 2) return SNSDE_ERR_DIMS;
        if (!s->step_tab || !s->out_step || !s->out_w || !s->y0 || !s->ys) return SNSDE_ERR_NULL;
        if (s->method != SNSDE_EULER && s->method != SNSDE_MILSTEIN && s->method != SNSDE_SRK) return SNSDE_ERR_OPTION;
        if (s->method == SNSDE_SRK && !s->srk_tab) return SNSDE_ERR_NULL;
        if (s->method == SNSDE_SRK && s->dW && !s->dU) return SNSDE_ERR_NULL;   // supplied dW needs its Levy integral
        const int no = s->model.noise_option;
        // Milstein: g dg/dy in closed form where g_i depends on y through y_i only, a transposed pass through the
        // diffusion net for 14/15/18/19 (generic kernels); sqrt(y) has no finite derivative at the clipped values
        if (s->method == SNSDE_MILSTEIN && no == 7) return SNSDE_ERR_UNSUPPORTED;
        if (s->noise_table && no != 12 && no != 13) return SNSDE_ERR_OPTION;   // a supplied table is the time-only factor
        if ((s->z0_weight != nullptr) != (s->z0_bias != nullptr)) return SNSDE_ERR_NULL;
        if (s->kl_column1 < 0 || s->kl_column1 > s->model.hidden_channels || s->reserved2 != 0) return SNSDE_ERR_DIMS;
    }
    return SNSDE_OK;
}

size_t snsde_workspace_bytes(const snsde_solve* s) {
    if (!s || s->struct_size != sizeof(snsde_solve)) return 0;
    SnsdeNet net;
    if (snsde_build_net(s->model, s->n_steps, &net)) return 0;
    size_t f = 0;
    snsde_solve tmp = *s;
    if (tmp.n_steps < 1) tmp.n_steps = 1;
    snsde_generic_workspace_floats(&tmp, net, &f);
    const size_t fm = snsde_mfma_workspace_floats(&tmp, net);
    if (fm > f) f = fm;
    return (f + 64) * sizeof(float);
}

int snsde_solve_forward(const snsde_solve* s, void* hip_stream) {
    int rc = validate_solve(s, false);
    if (rc) return rc;
    if (s->workspace_bytes < snsde_workspace_bytes(s)) return SNSDE_ERR_WORKSPACE;
    SnsdeNet net;
    rc = snsde_build_net(s->model, s->n_steps, &net);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    if (is_variant(s->model) || s->noise_table) {      // tutorial-style fields: the 4-row-tile MFMA kernels or nothing
        if (s->kernel == SNSDE_KERNEL_GENERIC || s->kernel == SNSDE_KERNEL_MFMA_M16) return SNSDE_ERR_UNSUPPORTED;
        return snsde_mfma_launch(s, net, st, 1);
    }
    if (s->method == SNSDE_SRK) {   // SRK: MFMA variant (M4 tiles) where instantiated, else the generic (all-options) family
        if (s->kernel == SNSDE_KERNEL_MFMA_M16) return snsde_mfma_launch(s, net, st, 0);     // (H = 64 / 128, elementwise diffusions)
        if (s->kernel == SNSDE_KERNEL_MFMA_M4) return snsde_mfma_launch(s, net, st, 1);
        if (s->kernel == SNSDE_KERNEL_MFMA_W4) return snsde_mfma_launch(s, net, st, 2);
        if (s->kernel == SNSDE_KERNEL_MFMA) return snsde_mfma_launch(s, net, st, -1);
        if (s->kernel == SNSDE_KERNEL_AUTO && snsde_mfma_supported(s, net)) return snsde_mfma_launch(s, net, st, -1);
        if (s->kernel != SNSDE_KERNEL_AUTO && s->kernel != SNSDE_KERNEL_GENERIC) return SNSDE_ERR_OPTION;
        if (s->z0_weight && (rc = snsde_z0_launch(s, st)) != SNSDE_OK) return rc;
        return snsde_srk_launch(s, net, st);
    }
    switch (s->kernel) {
        case SNSDE_KERNEL_AUTO:
            if (snsde_mfma_supported(s, net)) return snsde_mfma_launch(s, net, st, -1);
            break;
        case SNSDE_KERNEL_GENERIC: break;
        case SNSDE_KERNEL_MFMA: return snsde_mfma_launch(s, net, st, -1);
        case SNSDE_KERNEL_MFMA_M16: return snsde_mfma_launch(s, net, st, 0);
        case SNSDE_KERNEL_MFMA_M4: return snsde_mfma_launch(s, net, st, 1);
        case SNSDE_KERNEL_MFMA_W4: return snsde_mfma_launch(s, net, st, 2);
        default: return SNSDE_ERR_OPTION;
    }
    if (s->z0_weight && (rc = snsde_z0_launch(s, st)) != SNSDE_OK) return rc;
    return snsde_generic_launch(s, net, static_cast<hipStream_t>(hip_stream), 0, nullptr, nullptr, nullptr, nullptr);
}

// Host-only query: the kernel family snsde_solve_forward would launch for this descriptor (needs model, batch, knots,
// n_steps, method, kernel, flags, input dims; no device pointers).  SNSDE_PATH_NONE = no kernel covers the request
// (snsde_solve_forward returns SNSDE_ERR_UNSUPPORTED and the host layer takes its tensor loop).
int snsde_forward_path(const snsde_solve* s) {
    if (!s || s->struct_size != sizeof(snsde_solve) || validate_model(&s->model) || s->batch <= 0 || s->knots < 2 || s->n_steps <= 0) return SNSDE_PATH_NONE;
    const int no = s->model.noise_option;
    if (s->method == SNSDE_MILSTEIN && no == 7) return SNSDE_PATH_NONE;
    SnsdeNet net;
    if (snsde_build_net(s->model, s->n_steps, &net)) return SNSDE_PATH_NONE;
    const bool variant = is_variant(s->model) || s->noise_table;
    if (variant) {
        if (s->kernel == SNSDE_KERNEL_GENERIC || s->kernel == SNSDE_KERNEL_MFMA_M16) return SNSDE_PATH_NONE;
        return snsde_mfma_path(s, net, 1);
    }
    const int hint = s->kernel == SNSDE_KERNEL_MFMA_M16 ? 0 : (s->kernel == SNSDE_KERNEL_MFMA_M4 ? 1 : (s->kernel == SNSDE_KERNEL_MFMA_W4 ? 2 : -1));
    if (s->method == SNSDE_SRK) {
        if (s->kernel == SNSDE_KERNEL_MFMA_M16) return snsde_mfma_path(s, net, 0);
        if (s->kernel == SNSDE_KERNEL_MFMA_M4) return snsde_mfma_path(s, net, 1);
        if (s->kernel == SNSDE_KERNEL_MFMA_W4) return snsde_mfma_path(s, net, 2);
        if (s->kernel == SNSDE_KERNEL_MFMA) return snsde_mfma_path(s, net, -1);
        if (s->kernel == SNSDE_KERNEL_AUTO && snsde_mfma_supported(s, net)) return snsde_mfma_path(s, net, -1);
        return SNSDE_PATH_GENERIC_SRK;
    }
    if (s->kernel == SNSDE_KERNEL_GENERIC) return SNSDE_PATH_GENERIC;
    const int path = snsde_mfma_path(s, net, hint);
    if (path != SNSDE_PATH_NONE || s->kernel != SNSDE_KERNEL_AUTO) return path;
    return SNSDE_PATH_GENERIC;
}

int snsde_eval_fg(const snsde_solve* s, const float* step_row, const float* y, float* f_out, float* g_out,
                  void* hip_stream) {
    int rc = validate_solve(s, true);
    if (rc) return rc;
    if (!step_row || !y || !f_out || !g_out) return SNSDE_ERR_NULL;
    if (is_variant(s->model) || s->noise_table) return SNSDE_ERR_UNSUPPORTED;
    snsde_solve tmp = *s;
    tmp.n_steps = 1;
    tmp.n_out = 2;
    tmp.dW = nullptr;
    tmp.traj = nullptr;
    tmp.dW_out = nullptr;
    if (s->workspace_bytes < snsde_workspace_bytes(&tmp)) return SNSDE_ERR_WORKSPACE;
    SnsdeNet net;
    rc = snsde_build_net(s->model, 1, &net);
    if (rc) return rc;
    return snsde_generic_launch(&tmp, net, static_cast<hipStream_t>(hip_stream), 1, y, f_out, g_out, step_row);
}

int snsde_act_slots(const snsde_model* m) {
    int rc = validate_model(m);
    if (rc) return rc;
    const int no = m->noise_option;   // + the diffusion net's activations (hidden for 18/19, output)
    // smooth activations (tutorial fields): the pre-activations of the NL activated layers as well (their derivative)
    // (+ the hidden pre-activation of a two-layer diffusion net, the last slot)
    return m->num_hidden_layers + 1 + ((no == 18 || no == 19) ? 2 : ((no == 14 || no == 15) ? 1 : 0)) +
           (m->activation != SNSDE_ACT_RELU ? m->num_hidden_layers + ((no == 18 || no == 19) ? 1 : 0) : 0);
}

int snsde_save_layout(const snsde_solve* s, int32_t* act_slots, int32_t* stage_planes, int32_t* delta_slots) {
    if (!s) return SNSDE_ERR_NULL;
    if (s->struct_size != sizeof(snsde_solve)) return SNSDE_ERR_ABI;
    int slots = snsde_act_slots(&s->model);
    if (slots < 0) return slots;
    const int no = s->model.noise_option;
    const int nn = (no == 18 || no == 19) ? 2 : ((no == 14 || no == 15) ? 1 : 0);
    int planes = 1;
    if (s->method == SNSDE_SRK && nn > 0) {
        slots += nn; planes = 3;
        if (nn == 2 && s->model.activation != SNSDE_ACT_RELU) slots += 1;      // (+ the fourth evaluation's hidden pre-activation)
    }
    if (act_slots) *act_slots = slots;
    if (stage_planes) *stage_planes = planes;
    // Milstein through a diffusion net: the adjoint also leaves the tangent pass's factors (second-order parameter terms)
    if (delta_slots) {
        *delta_slots = slots + ((s->method == SNSDE_MILSTEIN && nn > 0) ? (nn == 2 ? 3 : 1) : 0);
        // 0: the adjoint of this solve accumulates the weight gradients itself (wave-pair adjoint, snsde_w4_kernel.h): no delta planes
        SnsdeNet net;
        if (nn > 0 && s->batch > 0 && s->n_steps > 0 && snsde_build_net(s->model, s->n_steps, &net) == SNSDE_OK &&
            snsde_backward_supported(s) == 1 && snsde_mfma_w4_fused_solve(s, net, nullptr, nullptr))
            *delta_slots = 0;
    }
    return SNSDE_OK;
}

int snsde_backward_supported(const snsde_solve* s) {
    if (!s || s->struct_size != sizeof(snsde_solve) || validate_model(&s->model)) return 0;
    if (s->method == SNSDE_MILSTEIN && s->model.noise_option == 7) return 0;     // no forward kernel either (validate_solve)
    SnsdeNet net;
    if (snsde_build_net(s->model, s->n_steps, &net)) return 0;
    if (is_variant(s->model) || s->noise_table)                // tutorial-style fields: the 4-row-tile MFMA adjoint or nothing
        return (s->kernel != SNSDE_KERNEL_GENERIC && s->kernel != SNSDE_KERNEL_MFMA_M16 && snsde_mfma_backward_supported(s, net)) ? 1 : 0;
    // 1: MFMA adjoint kernel (forward on the MFMA path with act_save); 2: generic adjoint kernel (forward on the
    // generic kernel, traj + dW_out only); 0: no fused backward for this configuration
    if (snsde_mfma_backward_supported(s, net) && s->kernel != SNSDE_KERNEL_GENERIC) return 1;
    return snsde_generic_backward_supported(s) ? 2 : 0;
}

size_t snsde_backward_workspace_bytes(const snsde_backward* b) {
    if (!b || b->struct_size != sizeof(snsde_backward) || b->fwd.struct_size != sizeof(snsde_solve)) return 0;
    SnsdeNet net;
    if (snsde_build_net(b->fwd.model, b->fwd.n_steps, &net)) return 0;
    size_t f = snsde_mfma_backward_workspace_floats(&b->fwd, net), g = 0;
    snsde_generic_workspace_floats(&b->fwd, net, &g);      // the generic adjoints pack their own weights / tables
    return ((f > g ? f : g) + 64) * sizeof(float);
}

int snsde_solve_backward(const snsde_backward* b, void* hip_stream) {
    if (!b) return SNSDE_ERR_NULL;
    if (b->struct_size != sizeof(snsde_backward)) return SNSDE_ERR_ABI;
    int rc = validate_solve(&b->fwd, false);
    if (rc) return rc;
    if (!b->grad_ys || !b->adj || !b->workspace || !b->fwd.traj) return SNSDE_ERR_NULL;
    // increments: dW_out, or the supplied dW, or - MFMA Euler / Milstein adjoint, Philox with a host key - regenerated in-kernel
    if (!b->fwd.dW_out && !b->fwd.dW && b->fwd.seed_dev) return SNSDE_ERR_NULL;
    SnsdeNet net;
    rc = snsde_build_net(b->fwd.model, b->fwd.n_steps, &net);
    if (rc) return rc;
    const int mode = snsde_backward_supported(&b->fwd);
    if (mode == 0) return SNSDE_ERR_UNSUPPORTED;
    if (mode == 2) {
        if (!b->fwd.dW_out) return SNSDE_ERR_NULL;
        if (b->flags & SNSDE_BWD_ADJ0_ONLY) return SNSDE_ERR_OPTION;     // (its parameter pass reads every a_n)
        if (b->delta_save) return SNSDE_ERR_UNSUPPORTED;    // the generic adjoint writes adjoints only
        if (b->workspace_bytes < snsde_backward_workspace_bytes(b)) return SNSDE_ERR_WORKSPACE;
        return snsde_generic_backward_launch(b, net, static_cast<hipStream_t>(hip_stream));
    }
    if (!b->fwd.act_save) return SNSDE_ERR_NULL;
    if (b->workspace_bytes < snsde_backward_workspace_bytes(b)) return SNSDE_ERR_WORKSPACE;
    return snsde_mfma_backward_launch(b, net, static_cast<hipStream_t>(hip_stream));
}

size_t snsde_param_gradients_workspace_bytes(const snsde_backward* b) {
    if (!b || b->struct_size != sizeof(snsde_backward) || b->fwd.struct_size != sizeof(snsde_solve)) return 0;
    SnsdeNet net;
    if (snsde_build_net(b->fwd.model, b->fwd.n_steps, &net)) return 0;
    if (snsde_backward_supported(&b->fwd) != 1) return 0;
    return snsde_wgrad_workspace_floats(b, net) * sizeof(float);
}

int snsde_param_gradients(const snsde_backward* b, float* grad_params, void* workspace, size_t workspace_bytes,
                          void* hip_stream) {
    if (!b || !grad_params || !workspace) return SNSDE_ERR_NULL;
    if (b->struct_size != sizeof(snsde_backward)) return SNSDE_ERR_ABI;
    int rc = validate_solve(&b->fwd, false);
    if (rc) return rc;
    if (!b->adj || !b->fwd.traj || !b->fwd.act_save || !b->fwd.workspace)
        return SNSDE_ERR_NULL;
    SnsdeNet net;
    rc = snsde_build_net(b->fwd.model, b->fwd.n_steps, &net);
    if (rc) return rc;
    if (snsde_backward_supported(&b->fwd) != 1) return SNSDE_ERR_UNSUPPORTED;
    if (!b->delta_save && !snsde_mfma_w4_fused(b, net, nullptr, nullptr)) return SNSDE_ERR_NULL;      // (delta_slots == 0: no planes)
    // the adjoint's workspace is an INPUT of this pass (its per-workgroup diffusion-side sums; on the wave-group path the per-tile
    // weight-gradient blocks themselves): the descriptor must still carry it, at the size the adjoint was given
    if (!b->workspace) return SNSDE_ERR_NULL;
    if (b->workspace_bytes < snsde_backward_workspace_bytes(b)) return SNSDE_ERR_WORKSPACE;
    if (workspace_bytes < snsde_param_gradients_workspace_bytes(b)) return SNSDE_ERR_WORKSPACE;
    return snsde_wgrad_launch(b, net, grad_params, (int32_t)snsde_param_numel(&b->fwd.model), static_cast<float*>(workspace),
                              static_cast<hipStream_t>(hip_stream));
}

int snsde_backward_with_gradients(const snsde_backward* b, float* grad_params, void* pg_workspace, size_t pg_workspace_bytes,
                                  void* hip_stream) {
    if (!b || !grad_params || !pg_workspace) return SNSDE_ERR_NULL;
    if (b->struct_size != sizeof(snsde_backward)) return SNSDE_ERR_ABI;
    int rc = validate_solve(&b->fwd, false);
    if (rc) return rc;
    if (!b->grad_ys || !b->adj || !b->workspace || !b->fwd.traj || !b->fwd.act_save || !b->fwd.workspace)
        return SNSDE_ERR_NULL;
    if (!b->fwd.dW_out && !b->fwd.dW && b->fwd.seed_dev) return SNSDE_ERR_NULL;
    SnsdeNet net;
    rc = snsde_build_net(b->fwd.model, b->fwd.n_steps, &net);
    if (rc) return rc;
    if (snsde_backward_supported(&b->fwd) != 1) return SNSDE_ERR_UNSUPPORTED;
    if (!b->delta_save && !snsde_mfma_w4_fused(b, net, nullptr, nullptr)) return SNSDE_ERR_NULL;      // (delta_slots == 0: no planes)
    if (b->workspace_bytes < snsde_backward_workspace_bytes(b)) return SNSDE_ERR_WORKSPACE;
    if (pg_workspace_bytes < snsde_param_gradients_workspace_bytes(b)) return SNSDE_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    rc = snsde_mfma_backward_launch(b, net, st);
    if (rc) return rc;
    return snsde_wgrad_launch(b, net, grad_params, (int32_t)snsde_param_numel(&b->fwd.model), static_cast<float*>(pg_workspace), st);
}

int snsde_spline_evaluate(const float* coeffs, int32_t batch, int32_t knots, int32_t channels, int32_t index,
                          float frac, int32_t derivative, float* out, void* hip_stream) {
    if (!coeffs || !out) return SNSDE_ERR_NULL;
    if (batch <= 0 || knots < 2 || channels <= 0) return SNSDE_ERR_DIMS;
    if (index < 0 || index > knots - 2) return SNSDE_ERR_INDEX;
    return snsde_spline_launch(coeffs, batch, knots, channels, index, frac, derivative, out,
                               static_cast<hipStream_t>(hip_stream));
}

}  // extern "C"
