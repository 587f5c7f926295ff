/*
 * snsde.h - C ABI of the MI355X-native Neural-SDE integration engine (libsnsde.so).
 *
 * The reference (yongkyung-oh/Stable-Neural-SDEs) is pure Python: it has no native interface.
 * The entry points below are what a binding for the hot path
 *
 *     torchsde.sdeint(sde=Diffusion_model, y0, ts, dt, method=...)
 *         benchmark_classification/models_sde/neuralsde.py:71-82   (call site, "A1" in SURVEY.md 8a)
 *         benchmark_forecasting/models_sde/neuralsde.py:71-82,145-156
 *         torch-ists/torch_ists/diff_module/NSDE/nsde_model.py:63-74
 *     torchcde.CubicSpline(coeffs, times).evaluate(t)
 *         benchmark_classification/models_sde/neuralsde.py:181-184, 296   (set_X / f)
 *         benchmark_classification/controldiffeq/interpolate.py:263-276    (vendored twin)
 *
 * would bind.  Plain C types, device pointers and sizes only; no torch types.  Every buffer is
 * caller-owned.  Launch functions only ENQUEUE work on the given HIP stream: they never allocate,
 * never synchronise and never throw, so a solve can be captured into a hipGraph.  All functions
 * return SNSDE_OK (0) or a negative error code (snsde_strerror()).  The library is stateless and
 * re-entrant; concurrent calls on different streams / devices are legal.
 *
 * All tensors are float32, row-major, contiguous.
 */
#ifndef SNSDE_H
#define SNSDE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version.  2 (round 4): every descriptor struct starts with `struct_size` = sizeof(the struct) as the CALLER compiled it;
 * the library compares it with its own sizeof and refuses a mismatch with SNSDE_ERR_ABI, so a binding built against an older
 * header (the structs grew fields in rounds 2 - 3 while the version stayed 1) fails loudly instead of being read past its end.
 * snsde_abi_check() lets a binding verify all of its struct sizes once, at load time.                                        */
#define SNSDE_VERSION 2

/* exported entry points: the library is built with -fvisibility=hidden, only these names are in its dynamic symbol table */
#define SNSDE_API __attribute__((visibility("default")))

enum {
    SNSDE_OK = 0,
    SNSDE_ERR_NULL = -1,         /* required pointer is NULL                                  */
    SNSDE_ERR_DIMS = -2,         /* non-positive / inconsistent dimension                      */
    SNSDE_ERR_OPTION = -3,       /* input_option not in 0..6 or noise_option not in 0..19      */
    SNSDE_ERR_UNSUPPORTED = -4,  /* valid request this build has no kernel for                 */
    SNSDE_ERR_WORKSPACE = -5,    /* workspace too small (see snsde_workspace_bytes)             */
    SNSDE_ERR_LDS = -6,          /* configuration exceeds the 160 KiB LDS budget of a CU        */
    SNSDE_ERR_TS = -7,           /* ts not strictly increasing / dt <= 0 / no fp32 progress     */
    SNSDE_ERR_LAUNCH = -8,       /* hipLaunchKernel reported an error                           */
    SNSDE_ERR_INDEX = -9,        /* index out of range                                          */
    SNSDE_ERR_ABI = -10          /* struct_size / version of the caller's binding differs from the library's */
};

enum { SNSDE_EULER = 0, SNSDE_MILSTEIN = 1, SNSDE_SRK = 2 /* SRID2, strong order 1.5, diagonal noise */ };

/* kernel selection (for tests / benchmarking); 0 lets the library choose */
enum {
    SNSDE_KERNEL_AUTO = 0,      /* MFMA fast path when the configuration is covered, else generic  */
    SNSDE_KERNEL_GENERIC = 1,   /* VALU kernel: every (input_option, noise_option), any dims        */
    SNSDE_KERNEL_MFMA = 2,      /* MFMA fast path, tile flavour chosen from the batch size          */
    SNSDE_KERNEL_MFMA_M16 = 3,  /* 16-row tiles (v_mfma_f32_16x16x4_f32)                            */
    SNSDE_KERNEL_MFMA_M4 = 4,   /* 4-row tiles  (v_mfma_f32_4x4x1_16b_f32)                          */
    SNSDE_KERNEL_MFMA_W4 = 5    /* 4-row tiles owned by ONE wave (H = 64 with a diffusion net, Euler): drift wave + net wave */
};

/* flags: REUSE_PREPARED skips the weight packing / time-table kernels; legal when `params`,
 * `step_tab` and `workspace` are unchanged since the previous call that ran them (e.g. every
 * batch of an evaluation epoch, or graph replays between optimizer steps). */
enum {
    SNSDE_FLAG_REUSE_PREPARED = 1,
    /* keep the reference's operation order yy = linear_in(..); z = emb(cat[yy, Xt]) on the MFMA path.  Default
     * (flag clear): emb o linear_in and emb o initial_network are pre-multiplied once per solve (there is no
     * non-linearity between them, neuralsde.py:200-210), which removes one layer and one barrier per step;
     * the result differs from the unfused order by float32 round-off only. */
    SNSDE_FLAG_EXACT_ORDER = 2,
    /* H = 256 on 4-row tiles: keep the fully streamed kernel (sixteen waves, every weight through the LDS ring each step) instead of
     * the two-tiles-per-wave kernel that holds a quarter of every layer in registers.  Same results bit for bit (same MFMA chains);
     * exists for A/B measurements and the bit-identity test. */
    SNSDE_FLAG_STREAM_ALL = 4,
    /* H = 128 on 4-row tiles (Euler / Milstein, elementwise diffusions, relu fields): four waves of two tiles, one wave per SIMD with
     * the whole register file (snsde_m4t_kernel.h), instead of the eight-wave lean kernel.  Same results bit for bit.  An A/B switch:
     * measured slower at the K2 shape, so it is opt-in. */
    SNSDE_FLAG_TWO_TILE = 8
};

/* Variants of the vector field beyond the benchmark Diffusion_model: the tutorial's Neural LSDE / LNSDE / GSDE fields
 * (tutorial notebooks, cell 7: MLPs with LipSwish activations, raw time feature, un-squashed drift and diffusion) map
 * onto the same fused step with these four switches (all 0 = the reference Diffusion_model, neuralsde.py:186-307).   */
enum { SNSDE_ACT_RELU = 0, SNSDE_ACT_LIPSWISH = 1 /* 0.909 silu(x) */, SNSDE_ACT_SILU = 2 };
enum { SNSDE_DRIFT_TANH = 0 /* f = tanh(z) (z * tanh(y) first for input_option 5/6) */, SNSDE_DRIFT_LINEAR = 1 /* f = z */,
       SNSDE_DRIFT_TIMES_Y = 2 /* f = z * y */ };
enum { SNSDE_DIFFUSION_TANH = 0 /* g = tanh(sigmoid(theta) nan_to_num(raw)) */, SNSDE_DIFFUSION_RAW = 1 /* g = raw */,
       SNSDE_DIFFUSION_RAW_NET = 2 /* noise_option 18 / 19 only: g = raw AND the net's last layer is not rectified (the tutorial's */
                                   /* NeuralSDEFunc: g = g_net(noise_in([t, y])), an MLP that ends in a Linear)                  */ };
enum { SNSDE_TIME_SINCOS = 0 /* linear_in sees [sin t, cos t, y] */, SNSDE_TIME_RAW = 1 /* [t, 0, y] */ };

/* Shape of a Diffusion_model: neuralsde.py:123-179 constructor arguments (+ the variant switches above). */
typedef struct snsde_model {
    int32_t input_channels;          /* C  */
    int32_t hidden_channels;         /* H  */
    int32_t hidden_hidden_channels;  /* HH (the reference requires HH == H whenever emb is used) */
    int32_t num_hidden_layers;       /* NL >= 1; there are NL-1 `linears` */
    int32_t input_option;            /* 0..6  */
    int32_t noise_option;            /* 0..19 */
    int32_t activation;              /* SNSDE_ACT_*        (0 for the reference's models) */
    int32_t drift_output;            /* SNSDE_DRIFT_*      (0 ...)                        */
    int32_t diffusion_output;        /* SNSDE_DIFFUSION_*  (0 ...)                        */
    int32_t time_feature;            /* SNSDE_TIME_*       (0 ...)                        */
} snsde_model;

/* ---- parameter block ------------------------------------------------------------------------
 * The engine reads ONE flat float32 buffer holding every parameter of the Diffusion_model in
 * state_dict order with the reference's names and nn.Linear layout (weight (out,in) row-major):
 *   theta (1,1) | sigma (1) | sigma_diag (H) | initial_network.{weight (H,C),bias} |
 *   linear_in.{weight (HH, H or H+2),bias} | emb.{weight (H,2H),bias} | linears.i.{weight,bias} |
 *   linear_out.{weight (H,HH),bias} | noise_t[.0/.2].* | noise_y[.0/.2].*
 * (neuralsde.py:142-179).  These functions describe that layout so a host can fill it. */
SNSDE_API int     snsde_param_count(const snsde_model* m);                 /* number of tensors, or <0 */
SNSDE_API int64_t snsde_param_numel(const snsde_model* m);                 /* total floats, or <0      */
SNSDE_API int     snsde_param_info(const snsde_model* m, int index, char* name, int name_cap,
                         int64_t* offset, int32_t* rows, int32_t* cols);

/* ---- fixed-step time grid (host, CPU) --------------------------------------------------------
 * torchsde 0.2.5 BaseSDESolver.integrate bookkeeping, emulated in float32:
 *   curr = ts[0]; for out_t in ts[1:]: while curr < out_t: next = min(curr + dt, ts[-1]); step
 * plus, per step, the spline interval of t0 on the knot grid `times`
 * (interpolate.py:263-268: idx = clamp(#{j: times[j] < t} - 1, 0, L-2), frac = t - times[idx]).
 *
 * step_tab row (SNSDE_STEP_STRIDE floats): [0] t0, [1] h = t1-t0, [2] sin(t0), [3] cos(t0), [4] frac,
 * [5] idx (int32 bit pattern), [6] sqrt(h), [7] t1, [8] number of outputs emitted after this step
 * (int32 bits), [9] index k of the first of them (int32 bits), [10..11] zero.
 * out_step[k] = index of the solver step after which output k+1 is emitted; out_w[2k], out_w[2k+1] =
 * linear_interp weights (t1-t)/(t1-t0), (t-t0)/(t1-t0).                                          */
#define SNSDE_STEP_STRIDE 12
SNSDE_API int snsde_grid_count(const float* ts, int32_t n_out, double dt, int32_t* n_steps);
SNSDE_API int snsde_grid_build(const float* ts, int32_t n_out, double dt, const float* times, int32_t knots,
                     int32_t n_steps, float* step_tab, int32_t* out_step, float* out_w);
/* Stage times of the SRK scheme: per solver step the four evaluation times t0 + c*h, c = 0, 1/4, 1/2, 1
 * (float32 arithmetic as torchsde's `t0 + c * dt`), each as SNSDE_SRK_STRIDE floats:
 * t, sin t, cos t, frac, idx (int32 bits), 0, 0, 0.   srk_tab is (n_steps, 4, SNSDE_SRK_STRIDE). */
#define SNSDE_SRK_STRIDE 8
SNSDE_API int snsde_grid_srk_build(const float* step_tab, int32_t n_steps, const float* times, int32_t knots, float* srk_tab);

/* ---- the solve --------------------------------------------------------------------------------
 * Replaces torchsde.sdeint(sde=Diffusion_model, y0, ts, dt, method) (neuralsde.py:78-82):
 * every solver step fuses X(t) (A10), f (A7), g (A8), the Brownian increment (A5) and the
 * Euler / Milstein update (A4/A6) for a tile of batch rows; rows are independent.            */
typedef struct snsde_solve {
    uint32_t struct_size;  /* = sizeof(snsde_solve) of the caller's header (SNSDE_ERR_ABI otherwise)              */
    snsde_model model;
    int32_t  batch;        /* B: rows on this device                                             */
    int32_t  knots;        /* L: len(times); coeffs has L-1 intervals                            */
    int32_t  n_steps;      /* N (snsde_grid_count)                                               */
    int32_t  n_out;        /* T = len(ts)                                                        */
    int32_t  method;       /* SNSDE_EULER | SNSDE_MILSTEIN | SNSDE_SRK                           */
    int32_t  kernel;       /* SNSDE_KERNEL_*                                                     */
    int32_t  flags;        /* SNSDE_FLAG_*                                                       */
    int32_t  reserved;     /* must be 0                                                          */
    int64_t  row_offset;   /* global index of local row 0 (batch shards keep the global Philox   */
                           /* stream: counter = (row_offset + row, step, col/4, 0))              */
    uint64_t seed;         /* Philox key                                                         */
    const float*   params;    /* device, snsde_param_numel floats                                */
    const float*   coeffs;    /* device (B, L-1, 4C) = cat[a, b, two_c, three_d]                  */
    const float*   step_tab;  /* device (N, SNSDE_STEP_STRIDE)                                   */
    const int32_t* out_step;  /* device (T-1)                                                    */
    const float*   out_w;     /* device (T-1, 2)                                                 */
    const float*   y0;        /* device (B, H)                                                   */
    const float*   dW;        /* device (N, B, H) supplied increments bm(t0,t1), or NULL: Philox */
    float*         ys;        /* device (T, B, H) out; ys[0] = y0                                */
    float*         traj;      /* optional device (N+1, B, H): every solver state                 */
    float*         dW_out;    /* optional device (N, B, H): the increments actually used         */
    const float*   srk_tab;   /* device (N, 4, SNSDE_SRK_STRIDE), method SNSDE_SRK only              */
    const float*   dU;        /* device (N, B, H) supplied space-time Levy integrals I_k0 (SRK with   */
                              /* supplied dW), or NULL: h*(dW/2 + sqrt(h/12) xi), xi from Philox      */
    float*         dU_out;    /* optional device (N, B, H): the I_k0 actually used                    */
    float*         act_save;  /* optional device (N, snsde_act_slots, B, H): per-step activations the MFMA adjoint */
                              /* needs (training mode; see snsde_solve_backward).  Models with the relu activation: */
                              /* the last drift slot (the pre-tanh drift output z; one per pass under SRK) carries, */
                              /* in its low num_hidden_layers mantissa bits, the signs of the step's / pass's        */
                              /* rectified layer outputs (bit k = [slot k > 0] of the same element; through a two-  */
                              /* layer diffusion net under SRK / Milstein also the net's hidden signs, one or two    */
                              /* bits above) - the adjoints' relu masks, so that they do not re-read the activation  */
                              /* planes; z itself is exact to 1e-6 .. 1e-5 relative there                            */
    float*         stage_save;/* SRK training on the MFMA path: optional device (3N + 1, planes, B, H), the input state of */
                              /* every drift pass (act_save / delta_save are then indexed by pass, 3N of them); planes = 1, */
                              /* or 3 with a diffusion net (snsde_save_layout)                                              */
    const uint64_t* seed_dev; /* optional device pointer to the Philox key: read when the kernel starts and used  */
                              /* instead of `seed`, so a captured hipGraph draws fresh increments on every replay */
                              /* (the owner updates the value in-stream between replays).                         */
    const float*   noise_table;/* optional device (N, H): the time-only factor of the diffusion per solver step, supplied by  */
                              /* the caller instead of being evaluated from noise_t (noise_option 12: raw = table[n],     */
                              /* 13: raw = table[n] * y).  Used for fields whose time-only diffusion net is not noise_t.   */
    const int32_t* row_out;   /* optional device (B): per-row output selection (the gather of NeuralSDE.forward,  */
                              /* neuralsde.py:115-116).  When set, ys is (B, H) with ys[b] = the solution at      */
                              /* ts[row_out[b]] (0 <= row_out[b] < n_out), and the backward's grad_ys is (B, H).  */
    const float*   z0_weight; /* optional device (H, C) + z0_bias (H): the wrapper's initial_network.  When set, the   */
    const float*   z0_bias;   /* solve starts from y0 = z0_weight . X(ts[0]) + z0_bias (NeuralSDE._prepare_initial_state, */
                              /* neuralsde.py:63-69) and `y0` is an OUTPUT: the caller passes an uninitialised (B, H)      */
                              /* buffer, the library fills it in the same launch that packs the weights.                   */
    void*          workspace; /* device scratch, >= snsde_workspace_bytes()                      */
    size_t         workspace_bytes;
    /* Path-integral accumulator column (torch-ists LatentSDE.f_aug / g_aug, diff_module/NSDE/latent_sde.py:60-89): when
     * kl_column1 = 1 + c > 0, state column c is NOT driven by the drift net but integrates, with the scheme's own drift
     * weights and no noise,  u(t, y) = 1/2 sum_{j < c} ((f_j(t, y) - (kl_prior_a y_j + kl_prior_b)) / g_j)^2  - the KL rate
     * between the posterior drift f and the linear prior drift theta (mu - y) (a = -theta, b = theta mu) under the shared
     * diagonal diffusion g (the solve's additive noise_table, |g| floored at 1e-7 as the reference's _stable_division).
     * Columns above c must be padding (zero weights).  Field variants on the 4-row-tile kernels only (SNSDE_DRIFT_LINEAR,
     * SNSDE_DIFFUSION_RAW with a noise_table, noise_option 12): Euler / Milstein on the lean kernel, SRK on the SRK variant;
     * the MFMA adjoints carry the accumulator's cotangent back into the drift (snsde_backward_supported == 1).            */
    int32_t        kl_column1;
    float          kl_prior_a;
    float          kl_prior_b;
    int32_t        reserved2;  /* must be 0                                                          */
} snsde_solve;

SNSDE_API size_t snsde_workspace_bytes(const snsde_solve* s);
SNSDE_API int    snsde_solve_forward(const snsde_solve* s, void* hip_stream);

/* ---- backward of the solve (discretise-then-optimise adjoint of the fixed-step scheme) ------
 * Replaces autograd THROUGH the unrolled solver loop (`loss.backward()` in
 * benchmark_classification/common_sde.py:160, ~25 autograd nodes per solver step): given dL/d ys it runs the adjoint
 * recursion of the scheme (Euler: a_n = a_{n+1} + h (df/dy)^T a_{n+1} + (dg/dy)^T (a_{n+1} * dW_n); Milstein and SRID2: the
 * reverse of their step formulas) backwards over the saved trajectory and writes EVERY a_n (adj[0] = dL/dy0).
 * snsde_backward_supported:
 *   1 = MFMA adjoint kernels: forward on the MFMA path with traj (+ dW_out unless the increments are supplied or Philox ones
 *       can be regenerated; + dU_out, stage_save for SRK) and act_save; Euler, Milstein and SRK, for the elementwise diffusions
 *       AND through the diffusion nets (noise_option 14/15/18/19; Milstein through a two-layer net at H = 128 and the nets at
 *       H = 256 excepted: mode 2); fills delta_save, and snsde_param_gradients then forms every parameter gradient ON THE DEVICE
 *       (split-R MFMA weight-gradient GEMMs over the saved activations / deltas, diffusion-side reductions, the folded
 *       first layer's algebra) in the flat layout of `params`;
 *   2 = generic adjoint kernels (forward on any kernel; traj + dW_out (+ dU_out for SRK) only; Euler, Milstein and SRK, any
 *       dims within the LDS budget, every noise_option (Milstein: all but 7); delta_save must be NULL): adjoints only, the
 *       host layer takes the parameter gradients from one batched evaluation of the step function;
 *   0 = none. */
typedef struct snsde_backward {
    uint32_t     struct_size;/* = sizeof(snsde_backward)                                                            */
    snsde_solve  fwd;        /* the forward descriptor (traj, dW_out, act_save filled by the forward)      */
    const float* grad_ys;    /* device (T, B, H): dL/d ys                                                 */
    float*       adj;        /* device (N+1, B, H) out: adjoint of every solver state                     */
    float*       delta_save; /* optional device (N, snsde_act_slots, B, H) out: per step, slot 0 = dL/d zout and */
                             /* slot g = dL/d(pre-activation of hidden layer slots-1-g): the left factors of    */
                             /* the weight-gradient GEMMs  dW_layer = sum delta^T . layer_input                  */
    void*        workspace;  /* device scratch >= snsde_backward_workspace_bytes (separate from fwd's)    */
    size_t       workspace_bytes;
    float*       grad_noise_table; /* snsde_param_gradients with fwd.noise_table set: optional device (N, H) out,      */
                             /* dL/d noise_table (the caller back-propagates it through whatever produced the table) */
    int32_t      flags;      /* SNSDE_BWD_ADJ0_ONLY: `adj` is (B, H) and receives dL/dy0 only - the MFMA adjoint kernels   */
                             /* (mode 1) need no a_n in memory, except Milstein through a diffusion net, whose second-    */
                             /* order weight-gradient jobs read them (there and in mode 2 the flag is refused: ERR_OPTION) */
    int32_t      reserved;
} snsde_backward;
enum { SNSDE_BWD_ADJ0_ONLY = 1 };

SNSDE_API int    snsde_act_slots(const snsde_model* m);             /* activation tensors saved per step, or <0: layer outputs (first, hidden.., */
                                                          /* pre-tanh drift) [+ diffusion-net slots]; models with a smooth activation  */
                                                          /* (SNSDE_ACT_LIPSWISH / SILU) also save every pre-activation (NL more slots,  */
                                                          /* + the hidden pre-activation of a two-layer diffusion net, the last slot)   */
SNSDE_API int    snsde_save_layout(const snsde_solve* s, int32_t* act_slots, int32_t* stage_planes, int32_t* delta_slots);
                                                          /* training-mode buffers of THIS solve (model + method): act_save /     */
                                                          /* delta_save are (passes, act_slots, B, H), stage_save (passes + 1,     */
                                                          /* stage_planes, B, H); passes = N (3N for SRK).  SRK through a          */
                                                          /* diffusion net (noise_option 14/15/18/19) saves a second set of net    */
                                                          /* slots (the step's fourth diffusion evaluation) and three stage planes */
                                                          /* (drift input H0 | diffusion input H1 | H1 of the fourth evaluation);  */
                                                          /* delta_save is (passes, delta_slots, B, H): act_slots, plus the tangent */
                                                          /* factors of Milstein through a diffusion net; delta_slots == 0: the      */
                                                          /* adjoint of this solve accumulates the weight gradients itself (Euler /  */
                                                          /* SRK at H = 64 with a diffusion net: per-tile sums in the backward       */
                                                          /* workspace), delta_save is not written and may be NULL                   */
SNSDE_API int    snsde_backward_supported(const snsde_solve* s);    /* 1 / 2 / 0, see above                          */
SNSDE_API size_t snsde_backward_workspace_bytes(const snsde_backward* b);
/* INVARIANT between forward and backward (mode 1): `fwd.workspace` is untouched AND `fwd.params` holds the values the forward ran
 * with.  The adjoint re-packs its transposed weights from the CURRENT params, but takes the folded first-layer product
 * emb . linear_in (input_option 2 / 4 / 6) from the forward's workspace: an in-place parameter update between the two calls, or a
 * second forward through the same workspace, mixes old and new weights without an error.  (torchsde.sdeint keeps both: the
 * autograd node owns the workspace and runs before the optimizer step.)                                                        */
SNSDE_API int    snsde_solve_backward(const snsde_backward* b, void* hip_stream);

/* Parameter gradients of the fused solve (mode 1 = MFMA path only): after snsde_solve_forward (traj, dW_out, act_save
 * kept; fwd.workspace untouched since) and snsde_solve_backward (adj, delta_save; b->workspace still the buffer that
 * call used — it holds the adjoint kernel's per-workgroup diffusion-side sums), writes dL/d params into
 * grad_params (device, snsde_param_numel floats, same flat layout as `params`, overwritten) — what autograd
 * accumulates through the unrolled loop (benchmark_classification/common_sde.py:158-160): split-R MFMA GEMMs
 * sum_r delta^T . input with per-workgroup partials and one deterministic reduction, the elementwise diffusion
 * reductions (theta, the time-only noise MLP; Euler and Milstein) and the first-layer/emb algebra.
 * `b` is the descriptor snsde_solve_backward ran with, workspace included: the adjoint's workspace is an INPUT here (its
 * per-workgroup diffusion-side sums; with delta_slots == 0 the per-tile weight-gradient blocks) - SNSDE_ERR_NULL without it,
 * SNSDE_ERR_WORKSPACE when workspace_bytes < snsde_backward_workspace_bytes(b). */
SNSDE_API size_t snsde_param_gradients_workspace_bytes(const snsde_backward* b);
SNSDE_API int    snsde_param_gradients(const snsde_backward* b, float* grad_params, void* workspace, size_t workspace_bytes,
                             void* hip_stream);

/* snsde_solve_backward followed by snsde_param_gradients as ONE call (mode 1 solves only; same arguments, same results bit for
 * bit): one transition from the host language and one validation instead of two, the launches back to back on the stream.
 * Enqueue-only and capturable like the two calls it replaces; returns SNSDE_ERR_UNSUPPORTED where snsde_backward_supported()
 * != 1 (the caller then uses the separate calls).                                                                            */
SNSDE_API int snsde_backward_with_gradients(const snsde_backward* b, float* grad_params, void* pg_workspace, size_t pg_workspace_bytes,
                                            void* hip_stream);

/* ---- cubic spline evaluation (A10) -----------------------------------------------------------
 * out[b, c] = a + (b + (0.5*two_c + three_d*frac/3)*frac)*frac   on interval `index`
 * (derivative != 0: b + (two_c + three_d*frac)*frac), operation order as
 * controldiffeq/interpolate.py:270-283 (bit-exact with the CPU reference).                      */
SNSDE_API int snsde_spline_evaluate(const float* coeffs, int32_t batch, int32_t knots, int32_t channels,
                          int32_t index, float frac, int32_t derivative, float* out, void* hip_stream);

/* ---- spline coefficient construction (A11 / A12; offline preprocessing in the reference) ------------
 * natural: controldiffeq.natural_cubic_spline_coeffs (interpolate.py:161-228) incl. its missing-value handling;
 * hermite: torchcde.hermite_cubic_coefficients_with_backward_differences (datasets/common.py:82-84).
 * times (L) and X (B, L, C) device float32, NaN = missing; coeffs (B, L-1, 4C) = cat[a, b, two_c, three_d] out. */
SNSDE_API size_t snsde_spline_workspace_bytes(int32_t batch, int32_t knots, int32_t channels);
SNSDE_API int snsde_natural_cubic_coeffs(const float* times, const float* X, int32_t batch, int32_t knots, int32_t channels,
                               float* coeffs, void* workspace, size_t workspace_bytes, void* hip_stream);
SNSDE_API int snsde_hermite_coeffs(const float* times, const float* X, int32_t batch, int32_t knots, int32_t channels,
                         float* coeffs, void* hip_stream);

/* vector-field probe: one evaluation of f(t,y) and g(t,y) (neuralsde.py:295-307) through the same
 * device code the solver uses.  `step_row` = one step_tab row (device, SNSDE_STEP_STRIDE floats)
 * describing t; y, f_out, g_out are device (B, H).  Uses s->model/batch/knots/params/coeffs/
 * workspace only.                                                                             */
SNSDE_API int snsde_eval_fg(const snsde_solve* s, const float* step_row, const float* y, float* f_out,
                  float* g_out, void* hip_stream);

/* Kernel family a forward solve of this descriptor runs on (host-side query, no device access): the coverage table in
 * profiles/ is generated from it.                                                                                  */
enum { SNSDE_PATH_NONE = 0,          /* no kernel: snsde_solve_forward returns SNSDE_ERR_UNSUPPORTED                 */
       SNSDE_PATH_GENERIC = 1,       /* generic kernel family (every option, any dims; one wave per row group)      */
       SNSDE_PATH_MFMA_M16 = 2,      /* MFMA, 16-row tiles                                                          */
       SNSDE_PATH_MFMA_M4 = 3,       /* MFMA, 4-row tiles, general kernel                                           */
       SNSDE_PATH_LEAN = 4,          /* MFMA, 4-row tiles, lean kernel (register-resident weights, H = 32/64/128)    */
       SNSDE_PATH_LEAN_STREAMED = 5, /* MFMA, 4-row tiles, lean kernel with L2 -> LDS streamed weights (H = 256)     */
       SNSDE_PATH_GENERIC_SRK = 6,   /* SRK on the generic family                                                   */
       SNSDE_PATH_MFMA_SRK = 7,      /* SRK on the MFMA 4-row tiles                                                 */
       SNSDE_PATH_MFMA_W4 = 8 };     /* MFMA, 4 rows per wave pair (csrc/snsde_w4_kernel.h: H = 64, diffusion nets, Euler / SRK) */
SNSDE_API int snsde_forward_path(const snsde_solve* s);

/* Readout head of the wrappers in one launch (inference; replaces the 4-5 tensor ops of `self.linear(z)`,
 * benchmark_classification/models_sde/neuralsde.py:59-61,119; benchmark_forecasting/models_sde/neuralsde.py:186; torch_ists
 * nsde_model.py):  out = W2 relu(bn(W1 act(x) + b1)) + b2  with act = tanh when input_tanh else identity and bn the
 * BatchNorm1d inference transform (v - mean) / sqrt(var + eps) * weight + bias when bn_mean is set.  All pointers device,
 * fp32, row-major contiguous; nn.Linear layout (out_features, in_features).                                              */
typedef struct snsde_head {
    uint32_t struct_size;    /* = sizeof(snsde_head)                                  */
    int32_t rows, in_features, hidden, out_features;
    int32_t input_tanh;
    float   bn_eps;
    const float* x;          /* (rows, in_features)                                   */
    const float* w1;         /* (hidden, in_features)                                 */
    const float* b1;         /* (hidden) or NULL                                      */
    const float* bn_mean;    /* (hidden) running mean, or NULL: no normalisation      */
    const float* bn_var;     /* (hidden) running variance (with bn_mean)              */
    const float* bn_weight;  /* (hidden) or NULL                                      */
    const float* bn_bias;    /* (hidden) or NULL                                      */
    const float* w2;         /* (out_features, hidden)                                */
    const float* b2;         /* (out_features) or NULL                                */
    float*       out;        /* (rows, out_features)                                  */
} snsde_head;
SNSDE_API int snsde_readout_head(const snsde_head* h, void* hip_stream);

/* ---- composed parameter blocks (tutorial-style fields) ---------------------------------------------------------
 * The tutorial notebooks' fields (cell 7 of the tutorial notebooks: linear_in / emb / f_net / linear_out, NeuralSDEFunc's f_net / g_net on
 * [t, y]) map onto the Diffusion_model parameter block by multiplying adjacent affine maps out: W' = W_outer W_inner,
 * b' = W_outer b_inner + b_outer.  These two calls build the block from the field's own tensors and carry the block's gradient back
 * to them, ONE launch each (in torch the same is ~25 + ~35 small launches per training step, which is what bounded that step).
 * A job writes one (weight, bias) pair of the block:
 *   w_outer == NULL : copy         W' = W_inner (R x Cin), b' = b_inner
 *   else            : composition  W' = W_outer (R x K) . W_inner (K x Cin),  b' = W_outer b_inner + b_outer
 * zero_col >= 0 inserts a zero column at that index of W' (the field's [t, y] columns -> the block's [t, 0, y]).
 * dst_w / dst_b: float offsets into `dst` (snsde_param_info); the caller zero-fills what no job writes.
 * The backward reads grad_dst at the same offsets and writes dL/d(each source tensor) at g_* float offsets of `grad_src`
 * (-1: not wanted).  All pointers device, fp32, row-major contiguous; at most SNSDE_MAX_AFFINE_JOBS jobs per call. */
#define SNSDE_MAX_AFFINE_JOBS 12
typedef struct snsde_affine_job {
    const float* w_outer;    /* (R, K) or NULL                                  */
    const float* b_outer;    /* (R) or NULL (only with w_outer)                 */
    const float* w_inner;    /* (K, Cin); copy jobs: (R, Cin)                   */
    const float* b_inner;    /* (K) (copy jobs: (R)) or NULL: b' = b_outer / 0  */
    int32_t R, K, Cin;       /* copy jobs: K ignored                            */
    int32_t zero_col;        /* -1: none                                        */
    int64_t dst_w, dst_b;    /* into dst / grad_dst                             */
    int64_t g_w_outer, g_b_outer, g_w_inner, g_b_inner;   /* into grad_src (backward only; -1: skip) */
} snsde_affine_job;
SNSDE_API int snsde_affine_compose(const snsde_affine_job* jobs, int32_t n_jobs, float* dst, void* hip_stream);
SNSDE_API int snsde_affine_compose_backward(const snsde_affine_job* jobs, int32_t n_jobs, const float* grad_dst, float* grad_src,
                                            void* hip_stream);

SNSDE_API int         snsde_version(void);
/* SNSDE_OK when `version` == SNSDE_VERSION and the four sizes equal the library's sizeof(snsde_model / snsde_solve /
 * snsde_backward / snsde_head); SNSDE_ERR_ABI otherwise.  A size of 0 means "this binding does not declare that struct"
 * (a forward-only binding has no snsde_backward / snsde_head) and is not compared.  A binding calls it once after loading
 * the library.                                                                                                          */
SNSDE_API int         snsde_abi_check(int version, size_t sizeof_model, size_t sizeof_solve, size_t sizeof_backward, size_t sizeof_head);
SNSDE_API const char* snsde_strerror(int code);

#ifdef __cplusplus
}
#endif
#endif /* SNSDE_H */
