"""``sdeint`` with the call contract of torchsde 0.2.5 at the reference's four call sites
(benchmark_classification/models_sde/neuralsde.py:71-82, benchmark_forecasting/...:71-82,145-156,
torch-ists/.../nsde_model.py:63-74):

    sdeint(sde=func, y0=z0, ts=ts, dt=dt, method='euler', options={'dt': dt})  ->  (T, B, H)

Dispatch
  * ``sde`` honours the Diffusion_model contract (engine.recognise) and ``y0`` is a CUDA tensor:
    ONE fused HIP solve (libsnsde.so).  No fallback: if the library is missing this raises.
  * anything else (arbitrary ``sde.f/g`` such as the tutorial's vector fields, or CPU tensors = the
    reference's CPU plumbing configuration): the same fixed-step scheme written with tensor ops,
    calling ``sde.f`` / ``sde.g`` once per step.

Fixed-step semantics (restated from torchsde 0.2.5; its source is not in the
reference tree): time accumulates in float32 by repeated ``curr_t + dt`` clamped to ``ts[-1]``;
outputs are linearly interpolated between the two solver states bracketing each ``ts[k]``;
Euler ``y + f*h + g*dW``; Milstein adds ``0.5 * g * dg/dy * (dW^2 - h)`` (Ito, diagonal noise).
"""
import os
import warnings

import numpy as np
import torch

from . import engine
from .controldiffeq import _HostTimes

METHODS = ('euler', 'milstein', 'srk')


class BrownianIncrements:
    """Minimal Brownian-motion object: ``bm(ta, tb)`` ~ N(0, (tb - ta) I) of shape (B, H).

    Increments over disjoint intervals are independent draws from one generator; unlike
    torchsde.BrownianInterval it does not support re-querying overlapping intervals (the fixed-step
    solvers never do)."""

    def __init__(self, t0=0.0, t1=1.0, size=None, dtype=torch.float32, device=None, entropy=None, **kwargs):
        self.shape = tuple(size)
        self.dtype, self.device = dtype, device
        self.generator = torch.Generator(device=device if device is not None else 'cpu')
        self.generator.manual_seed(int(entropy) if entropy is not None else int(torch.empty((), dtype=torch.int64).random_().item()))

    levy_area_approximation = 'space-time'

    def __call__(self, ta, tb=None, return_U=False, **kwargs):
        h = (torch.as_tensor(tb, dtype=self.dtype) - torch.as_tensor(ta, dtype=self.dtype))
        z = torch.randn(self.shape, dtype=self.dtype, device=self.device, generator=self.generator)
        h = h.to(z.device)
        W = z * h.sqrt()
        if not return_U:
            return W
        xi = torch.randn(self.shape, dtype=self.dtype, device=self.device, generator=self.generator)
        return W, h * (0.5 * W + (h / 12).sqrt() * xi)     # U = int_ta^tb (W_s - W_ta) ds


BrownianInterval = BrownianIncrements      # torchsde.BrownianInterval(t0, t1, size, dtype, device, entropy, ...) call sites


class BaseSDE(torch.nn.Module):
    """torchsde.BaseSDE: an nn.Module that records its noise / SDE type (torch-ists .../NSDE/latent_sde.py:31 subclasses
    ``torchsde.SDEIto``)."""

    def __init__(self, noise_type, sde_type):
        super().__init__()
        if noise_type not in ('diagonal', 'scalar', 'additive', 'general'):
            raise ValueError(f"Expected noise type in ('diagonal', 'scalar', 'additive', 'general'), but found {noise_type}")
        if sde_type not in ('ito', 'stratonovich'):
            raise ValueError(f"Expected sde type in ('ito', 'stratonovich'), but found {sde_type}")
        self.noise_type = noise_type
        self.sde_type = sde_type


class SDEIto(BaseSDE):
    def __init__(self, noise_type):
        super().__init__(noise_type=noise_type, sde_type='ito')


class SDEStratonovich(BaseSDE):
    def __init__(self, noise_type):
        super().__init__(noise_type=noise_type, sde_type='stratonovich')


def _as_ts(ts, y0):
    if not torch.is_tensor(ts):
        if not isinstance(ts, (tuple, list)) or not all(isinstance(t, (float, int)) for t in ts):
            raise ValueError("Evaluation times `ts` must be a 1-D Tensor or list/tuple of floats.")
        ts = torch.tensor(ts, dtype=y0.dtype, device=y0.device)
    if ts.dim() != 1 or ts.numel() < 2:
        raise ValueError("Evaluation times `ts` must be a 1-D Tensor with at least two entries.")
    return ts


def _fresh_seed():
    # drawn from torch's CPU generator: reproducible under torch.manual_seed, no device sync
    return int(torch.empty((), dtype=torch.int64).random_().item())


_CAPTURE_SEEDS = {}
_UNFUSED_WARNED = set()


def _dev_key(device):
    device = torch.device(device)
    return device.index if device.index is not None else torch.cuda.current_device()


def _capture_seed(device):
    """Device-resident Philox key for solves recorded into a hipGraph: the increment is part of the graph, so every
    replay integrates against fresh Brownian increments (a host-drawn seed would be frozen into the recording)."""
    state = _CAPTURE_SEEDS.get(_dev_key(device))
    if state is None:
        raise RuntimeError("sdeint without options['seed'] inside a CUDA graph capture: call "
                           "stable_neural_sdes_amd.torchsde.prepare_graph_capture(device) before capturing")
    state.add_(1)
    return state


def prepare_graph_capture(device):
    """Allocate the device-resident seed used by solves that are recorded into a CUDA/HIP graph (call once, outside
    the capture).  The key is drawn from torch's CPU generator, so torch.manual_seed makes replays reproducible."""
    key = _dev_key(device)
    if key not in _CAPTURE_SEEDS:
        _CAPTURE_SEEDS[key] = torch.tensor([_fresh_seed() & 0x3FFFFFFFFFFFFFFF], dtype=torch.int64,
                                           device=torch.device('cuda', key))
    return _CAPTURE_SEEDS[key]


def sdeint(sde, y0, ts, bm=None, method=None, dt=1e-3, adaptive=False, rtol=1e-5, atol=1e-4, dt_min=1e-5,
           options=None, names=None, logqp=False, extra=False, extra_solver_state=None, **unused_kwargs):
    if unused_kwargs:
        warnings.warn(f"Unexpected arguments {unused_kwargs}")
    if adaptive:
        raise NotImplementedError("adaptive stepping is not implemented; the reference only uses fixed dt")
    if logqp or extra or extra_solver_state is not None:
        raise NotImplementedError("logqp / extra solver state are not implemented")
    if not torch.is_tensor(y0) or y0.dim() != 2:
        raise ValueError("`y0` must be a 2-dimensional tensor of shape (batch, channels).")
    ts = _as_ts(ts, y0)
    options = dict(options or {})
    if method is None:
        method = 'srk'   # torchsde's default for Ito / diagonal noise
    if method not in METHODS:
        raise ValueError(f"Expected method in {METHODS}, but found {method}.")
    if not (float(dt) > 0):
        raise ValueError("`dt` must be positive.")
    backend = options.get('backend', 'auto')
    if backend not in ('auto', 'hip', 'torch'):
        raise ValueError("options['backend'] must be 'auto', 'hip' or 'torch'")

    # names={'drift': 'f', 'diffusion': 'g'} is the default mapping: still the fused path
    default_names = names is None or (names.get('drift', 'f') == 'f' and names.get('diffusion', 'g') == 'g'
                                      and not (set(names) - {'drift', 'diffusion'}))
    rec = engine.recognise(sde) if default_names else None
    want_hip = backend == 'hip' or (backend == 'auto' and rec is not None and y0.is_cuda)
    if 'z0_linear' in options and not (want_hip and rec is not None and y0.is_cuda):
        y0 = _materialise_z0(sde, y0, ts, options)      # only the fused solve evaluates the initial state itself
    if want_hip:
        if rec is None:
            raise ValueError("options['backend']='hip' needs an sde honouring the Diffusion_model contract")
        if not y0.is_cuda:
            raise ValueError("the HIP engine needs CUDA (ROCm) tensors")
        return _sdeint_hip(sde, rec, y0, ts, bm, method, float(dt), options)
    if default_names and rec is None and backend == 'auto' and y0.is_cuda:
        ys = _sdeint_composed(sde, y0, ts, bm, method, float(dt), options)     # tutorial-style fields (fields.py)
        if ys is not None:
            return ys
    if not default_names and rec is None and backend == 'auto' and y0.is_cuda:
        return _sdeint_latent(sde, y0, ts, bm, method, float(dt), options, names)    # LatentSDE-shaped modules (falls back itself)
    return _sdeint_torch(sde, y0, ts, bm, method, float(dt), options, names)


def sdeint_adjoint(sde, y0, ts, bm=None, method=None, adjoint_method=None, adjoint_adaptive=False, adjoint_rtol=1e-5,
                   adjoint_atol=1e-4, adjoint_options=None, adjoint_params=None, names=None, **kwargs):
    """torchsde.sdeint_adjoint's call contract (in-tree user: torch-ists .../NSDE/latent_sde.py:134-141).  Gradients come
    from the adjoint of the DISCRETE scheme: for a Diffusion_model on CUDA the fused HIP adjoint kernels
    (snsde_solve_backward + snsde_param_gradients: no autograd graph over the steps, memory O(N B H) of saved states
    rather than ~25 autograd nodes per step), otherwise autograd through the tensor-op loop.  torchsde integrates the
    continuous adjoint SDE backwards instead; the two agree to the discretisation error of the forward scheme.  The
    adjoint_* solver options have no counterpart here: they are accepted for signature compatibility and a non-default value
    is reported once per process."""
    given = [n for n, v, d in (('adjoint_method', adjoint_method, None), ('adjoint_adaptive', adjoint_adaptive, False),
                               ('adjoint_rtol', adjoint_rtol, 1e-5), ('adjoint_atol', adjoint_atol, 1e-4),
                               ('adjoint_options', adjoint_options, None), ('adjoint_params', adjoint_params, None)) if v != d]
    if given and 'sdeint_adjoint' not in _UNFUSED_WARNED:
        _UNFUSED_WARNED.add('sdeint_adjoint')
        warnings.warn(f"sdeint_adjoint: {', '.join(given)} ignored - gradients are the adjoint of the DISCRETE scheme "
                      "(fused HIP adjoint kernels for a Diffusion_model on CUDA, autograd through the step loop otherwise), "
                      "not torchsde's continuous stochastic adjoint; both agree to the forward scheme's discretisation error.")
    return sdeint(sde, y0, ts, bm=bm, method=method, names=names, **kwargs)


def _materialise_z0(sde, y0, ts, options):
    """options['z0_linear'] = the wrapper's `initial_network`: y0 is a placeholder and the solve starts from
    initial_network(X(ts[0])) (NeuralSDE._prepare_initial_state, neuralsde.py:63-69).  Evaluated here with tensor ops for
    every path but the no-grad fused solve, which computes it inside its prepare launch."""
    lin = options.pop('z0_linear')
    return lin(sde.X.evaluate(ts[0])).to(y0.dtype)


def _z0_fusable(lin, sde, y0):
    w, b = lin.weight, lin.bias
    return (b is not None and w.is_cuda and w.dtype == torch.float32 and b.dtype == torch.float32 and w.is_contiguous()
            and tuple(w.shape) == (y0.shape[1], sde.input_channels) and not torch.is_grad_enabled())


class _DrawnIncrements:
    """bm(ta, tb[, return_U]) over increments that were already drawn from the caller's Brownian object, in call order: a
    fallback from the fused path to the tensor-op loop must not query a stateful `bm` a second time."""

    def __init__(self, dW, dU=None):
        self.dW, self.dU, self.n = dW, dU, 0

    def __call__(self, ta, tb=None, return_U=False, **kwargs):
        i, self.n = self.n, self.n + 1
        return (self.dW[i], self.dU[i]) if return_U else self.dW[i]


def _sdeint_hip(sde, rec, y0, ts, bm, method, dt, options):
    model, layout, numel = rec
    z0_lin = None
    if 'z0_linear' in options:
        options = dict(options)
        if (_z0_fusable(options['z0_linear'], sde, y0) and bm is None and options.get('kernel', 'auto') == 'auto'
                and not options.get('save_traj', False)):
            z0_lin = options.pop('z0_linear')
        else:
            y0 = _materialise_z0(sde, y0, ts, options)
    pidx = engine.param_index(sde, layout)
    needs_grad = torch.is_grad_enabled() and (y0.requires_grad or any(p.requires_grad for p in pidx.params))
    dev = y0.device
    coeffs = sde.coeffs
    if coeffs.dim() != 3 or coeffs.shape[0] != y0.shape[0]:
        raise ValueError("sde.coeffs must have shape (batch, len(times) - 1, 4 * input_channels)")
    coeffs = coeffs.detach().to(device=dev, dtype=torch.float32).contiguous()
    y0c = y0.detach().to(torch.float32).contiguous()
    times_host = _HostTimes.get(sde.times)
    ts_host = _HostTimes.get(ts)
    grid = engine.step_grid(ts_host, dt, times_host, dev)
    dW = dU = None
    if bm is not None:
        t0 = torch.from_numpy(grid.t0)
        t1 = torch.from_numpy(grid.t1)
        if method == 'srk':   # torchsde: I_k, I_k0 = bm(t0, t1, return_U=True)
            pairs = [bm(t0[n], t1[n], return_U=True) for n in range(grid.N)]
            dW = torch.stack([p[0].to(device=dev, dtype=torch.float32) for p in pairs]).contiguous()
            dU = torch.stack([p[1].to(device=dev, dtype=torch.float32) for p in pairs]).contiguous()
        else:
            dW = torch.stack([bm(t0[n], t1[n]).to(device=dev, dtype=torch.float32) for n in range(grid.N)]).contiguous()
    seed = options.get('seed')
    if seed is None:
        seed = _capture_seed(dev) if torch.cuda.is_current_stream_capturing() else _fresh_seed()
    elif not torch.is_tensor(seed):
        seed = int(seed)
    if 'row_offset' not in options:
        # one process per GPU (DDP): ranks that seed identically must not integrate against identical Brownian paths.
        # Philox counters use the global row, so shard r of equal-sized shards starts at row r * local_batch.
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            options = dict(options, row_offset=dist.get_rank() * int(y0.shape[0]))
    row_out = options.get('row_out')
    if row_out is not None:     # per-row output selection fused into the solve: the result is (B, H)
        row_out = row_out.to(device=dev, dtype=torch.int32).contiguous()
        options = dict(options, row_out=row_out)
    if options.get('kernel', 'auto') == 'auto' and not options.get('save_traj', False) and not options.get('recompute'):
        pad = engine.padding_plan(model, y0c.shape[0], coeffs.shape[1] + 1, grid.N, method)
        if pad is not None:       # a hidden size without MFMA instantiation: solve the zero-padded model (exact)
            if z0_lin is not None:
                y0 = _materialise_z0(sde, y0, ts, {'z0_linear': z0_lin})
                y0c, z0_lin = y0.detach().to(torch.float32).contiguous(), None
            out = _sdeint_padded(sde, rec, pad, coeffs, grid, y0, dW, dU, method, seed, options, row_out, needs_grad)
            if out is not None:
                return out
    if needs_grad:
        mode = engine.backward_mode(model, y0c.shape[0], coeffs.shape[1] + 1, grid, method, options.get('kernel', 'auto'),
                                    bool(options.get('exact_order', False)))
        if mode == 0 and not options.get('strict', False):
            # no fused adjoint for this configuration (Milstein with sqrt(y); shapes beyond the generic adjoint's LDS budget):
            # differentiate through the unfused tensor-op loop on the same device rather than fail the
            # reference's training loop; options={'strict': True} raises instead
            key = (sde.input_option, sde.noise_option, method)
            if key not in _UNFUSED_WARNED:
                _UNFUSED_WARNED.add(key)
                warnings.warn(f"sdeint: no fused backward for input_option={key[0]}, noise_option={key[1]}, method={method!r}; "
                              "differentiating through the unfused tensor-op loop (slow).")
            return _sdeint_torch(sde, y0, ts, bm if dW is None else _DrawnIncrements(dW, dU), method, dt, options, None)
        return _FusedSolve.apply(sde, rec, coeffs, grid, times_host, (dW, dU), method, seed, options, y0, *pidx.params)
    flat = engine.flatten_params(sde, layout, numel, dev)
    call = engine.SolveCall(model, flat, coeffs, grid, y0c, dW=dW, method=method, seed=seed,
                            row_offset=int(options.get('row_offset', 0)), kernel=options.get('kernel', 'auto'),
                            save_traj=bool(options.get('save_traj', False)),
                            exact_order=bool(options.get('exact_order', False)), dU=dU, row_out=row_out,
                            z0_linear=None if z0_lin is None else (z0_lin.weight.detach(), z0_lin.bias.detach().contiguous()))
    try:
        ys = call.launch()
    except engine._lib.SnsdeError as exc:
        # a valid request no kernel covers (Milstein with noise_option 7, sqrt(y)): same behaviour as the gradient path, the
        # unfused tensor-op loop, unless strict
        # (-6, SNSDE_ERR_LDS: a hidden size whose per-tile buffers exceed the LDS budget of the only kernel family that covers
        # the request — the generic Milstein kernel for the diffusion nets above H ~ 460 — is the same situation)
        if exc.code not in (-4, -6) or options.get('strict', False):
            raise
        if z0_lin is not None:
            y0 = _materialise_z0(sde, y0, ts, {'z0_linear': z0_lin})
        return _sdeint_torch(sde, y0, ts, bm if dW is None else _DrawnIncrements(dW, dU), method, dt, options, None)
    if options.get('save_traj', False):
        sde.last_trajectory = call.traj
    return ys.to(y0.dtype)


def _sdeint_composed(sde, y0, ts, bm, method, dt, options):
    """Fused solve of a tutorial-style field (tutorial/*.ipynb cell 7; fields.compose): no-grad calls only, the lean
    4-row-tile kernel with the variant switches.  None = not such a field / not covered: the caller takes the generic
    stepper."""
    from . import fields
    needs_grad = torch.is_grad_enabled() and (y0.requires_grad or any(p.requires_grad for p in sde.parameters()))
    field = fields.compose(sde)
    coeffs = getattr(sde, 'coeffs', None)
    if field is None or not torch.is_tensor(coeffs) or coeffs.dim() != 3 or coeffs.shape[0] != y0.shape[0]:
        return None
    dev = y0.device
    capturing = torch.cuda.is_current_stream_capturing()
    if capturing and (bm is not None or field.verified.get(str(dev)) is not True):
        return None      # graph capture: solves (training ones too) of a mapping that was verified before the capture (a warm-up solve)
    coeffs = coeffs.detach().to(device=dev, dtype=torch.float32).contiguous()
    times_host = _HostTimes.get(sde.times)
    if not fields.verify(field, coeffs, times_host, dev):
        return None
    grid = engine.step_grid(_HostTimes.get(ts), dt, times_host, dev)
    dW = dU = None
    if bm is not None and not field.parts.get('ode', False):      # (the ODE field has no diffusion: nothing to draw)
        t0, t1 = torch.from_numpy(grid.t0), torch.from_numpy(grid.t1)
        if method == 'srk':
            pairs = [bm(t0[n], t1[n], return_U=True) for n in range(grid.N)]
            dW = torch.stack([p[0].to(device=dev, dtype=torch.float32) for p in pairs]).contiguous()
            dU = torch.stack([p[1].to(device=dev, dtype=torch.float32) for p in pairs]).contiguous()
        else:
            dW = torch.stack([bm(t0[n], t1[n]).to(device=dev, dtype=torch.float32) for n in range(grid.N)]).contiguous()
    # the times the time-only diffusion factor is tabulated at: the step times; SRK: the four stage times of every step
    tab_times = grid.d_t0 if method != 'srk' else engine.srk_stage_times(grid)
    seed = options.get('seed')
    if seed is None:     # recorded solves read a device-resident key that the recording itself advances: fresh noise per replay
        seed = _capture_seed(dev) if capturing else _fresh_seed()
    elif not torch.is_tensor(seed):
        seed = int(seed)
    row_offset = options.get('row_offset')
    if row_offset is None:
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        row_offset = dist.get_rank() * int(y0.shape[0]) if multi else 0
    row_out = options.get('row_out')
    if row_out is not None:
        row_out = row_out.to(device=dev, dtype=torch.int32).contiguous()
    if needs_grad:
        # training: the composition and the module's own g stay in the autograd graph; the solve between them is the fused
        # forward + adjoint + weight-gradient pass (flat-block and table gradients flow back through those graphs)
        if engine.backward_mode(field.model, int(y0.shape[0]), coeffs.shape[1] + 1, grid, method, table=field.tabulated) != 1:
            return None
        flat = field.flat(dev, grad=True)
        tab = field.noise_table(tab_times, dev, grad=True) if field.tabulated else None
        return _ComposedSolve.apply(field.model, coeffs, grid, dW, method, seed, int(row_offset), row_out, y0, flat, tab, dU)
    # options={'trust_versions': True}: the cached composed block / table are keyed on the parameters' addresses and version
    # counters alone (no content fingerprint = no device->host read per solve); in-place edits through `.data` are then the
    # caller's to avoid
    field.trust_versions = bool(options.get('trust_versions', False))
    flat, tab = field.inference_inputs(tab_times, dev)
    call = engine.SolveCall(field.model, flat, coeffs, grid, y0.detach().to(torch.float32).contiguous(), dW=dW, dU=dU,
                            method=method, seed=seed, row_offset=int(row_offset), row_out=row_out, noise_table=tab)
    try:
        return call.launch().to(y0.dtype)
    except engine._lib.SnsdeError as exc:
        if exc.code not in (-4, -6):
            raise
        return None


def _draw_increments(bm, grid, y0, method, options):
    """Every increment of a solve up front, (N, B, H) I_k (and I_k0 for SRK): from the caller's Brownian object, in step
    order, or from a torch generator (options['seed'])."""
    dev = y0.device
    t0s, t1s = torch.from_numpy(grid.t0).to(dev), torch.from_numpy(grid.t1).to(dev)
    hs = (t1s - t0s).to(y0.dtype)
    if bm is None:
        gen = torch.Generator(device=dev)
        seed = options.get('seed')
        gen.manual_seed(int(seed) if seed is not None and not torch.is_tensor(seed) else _fresh_seed())
        hcol = hs.reshape(-1, *([1] * y0.dim()))
        dW = torch.randn((grid.N,) + tuple(y0.shape), dtype=y0.dtype, device=dev, generator=gen) * hcol.sqrt()
        dU = None
        if method == 'srk':      # I_k0 = h (I_k / 2 + sqrt(h / 12) xi): the space-time Levy integral
            xi = torch.randn((grid.N,) + tuple(y0.shape), dtype=y0.dtype, device=dev, generator=gen)
            dU = hcol * (0.5 * dW + (hcol / 12).sqrt() * xi)
        return dW, dU
    if method == 'srk':
        pairs = [bm(t0s[n], t1s[n], return_U=True) for n in range(grid.N)]
        return (torch.stack([p[0].to(device=dev, dtype=y0.dtype) for p in pairs]),
                torch.stack([p[1].to(device=dev, dtype=y0.dtype) for p in pairs]))
    return torch.stack([bm(t0s[n], t1s[n]).to(device=dev, dtype=y0.dtype) for n in range(grid.N)]), None


def _sdeint_latent(sde, y0, ts, bm, method, dt, options, names):
    """torch-ists' LatentSDE through names={'drift': 'f_aug', 'diffusion': 'g_aug'} (latent_sde.py:60-89, 134-141): the state
    is [latent | KL accumulator] and the accumulator never feeds back, so the solve splits into
      * the LATENT dynamics - posterior-drift MLP of [sin t, cos t, y], constant shared diffusion - on the fused kernels
        (fields.compose_latent: forward, adjoint and weight gradients as for the tutorial fields), returning every state, and
      * the accumulator = the scheme's own update of the last channel, evaluated for ALL steps at once: one batched call of the
        module's f_aug / g_aug (the same `_srk_step` / Euler update the tensor-op loop runs) on the (N B) states of the solve.
    Autograd joins the two: the batched step's graph gives d/d parameters and the cotangents of every state, which enter the
    fused adjoint as output gradients.  N sequential launches of ~25-100 kernels become one solve + one batched step.
    Falls back to the tensor-op loop (same increments) for anything it does not recognise."""
    from . import fields
    field = None
    if y0.dim() == 2 and y0.shape[1] >= 2 and not torch.cuda.is_current_stream_capturing() and 'row_out' not in options:
        field = fields.compose_latent(sde, names, int(y0.shape[1]))
    if field is None:
        return _sdeint_torch(sde, y0, ts, bm, method, dt, options, names)
    dev, B = y0.device, int(y0.shape[0])
    Hl, P = field.parts['latent'], field.model.hidden_channels
    ts_host = _HostTimes.get(ts)
    times_host = np.array([ts_host[0], ts_host[-1]], dtype=np.float32)
    grid = engine.step_grid(ts_host, dt, times_host, dev)
    # Increments: the caller's `bm` or, with options['seed'], the generator stream the tensor-op loop draws (same results on both
    # paths) are drawn up front; an unseeded call of the in-solve accumulator path lets the kernels draw Philox increments
    # (no (N, B, H) tensors of normals at all) and only draws here if it has to fall back
    acc = field.parts.get('acc')
    # (a learnable prior / diffusion - any tensor h() or g() can reach that requires grad - keeps the split solve below, whose
    #  quadrature goes through the module's own f_aug and carries that gradient; the in-solve constants are plain floats)
    in_solve = (acc is not None and P > Hl and os.environ.get('SNSDE_LATENT_SPLIT') != '1' and
                not (torch.is_grad_enabled() and any(t.requires_grad for t in field.parts.get('prior_leaves', ()))))
    philox = in_solve and bm is None and options.get('seed') is None
    drawn_box = []

    def increments():
        if not drawn_box:
            dW_, dU_ = _draw_increments(bm, grid, y0, method, options)
            drawn_box.append((dW_, dU_, _DrawnIncrements(dW_, dU_)))
        return drawn_box[0]
    dW, dU = (None, None) if philox else increments()[:2]

    def fallback():
        return _sdeint_torch(sde, y0, ts, increments()[2], method, dt, options, names)
    view = field.sde
    cache = field.__dict__.setdefault('_dummy_control', {})
    key = (B, str(dev))
    if key not in cache:        # the field has no control path: one zero channel on the knots [ts[0], ts[-1]]
        cache.clear()
        cache[key] = torch.zeros(B, 1, 4, device=dev, dtype=torch.float32)
    coeffs = cache[key]
    view.coeffs, view.times = coeffs, torch.from_numpy(times_host).to(dev)
    try:
        if not fields.verify(field, coeffs, times_host, dev):
            return fallback()
    except RuntimeError:
        return fallback()
    needs_grad = torch.is_grad_enabled() and (y0.requires_grad or any(p.requires_grad for p in sde.parameters()))
    widen = lambda t: None if t is None else torch.nn.functional.pad(t[..., :Hl].to(torch.float32), (0, P - Hl)).contiguous()
    # (1) the accumulator INSIDE the solve (snsde.h: kl_column1): column Hl of the padded state integrates the KL rate with the
    #     scheme's own drift weights, the adjoint kernels carry its cotangent back into the drift net - one solve over the
    #     caller's grid, no quadrature launches.  Needs the module's prior drift in the form fields.compose_latent recognised,
    #     a diffusion without gradient (the reference's sigma is a buffer) and a spare padded column.
    if in_solve:
        kl = (Hl, acc[0], acc[1])
        key = _fresh_seed() if philox else 0
        tt = grid.d_t0 if method != 'srk' else engine.srk_stage_times(grid)
        y0a = torch.nn.functional.pad(y0[:, :Hl + 1], (0, P - Hl - 1))
        try:
            if needs_grad:
                tab = field.noise_table(tt, dev, grad=True)
                if not tab.requires_grad and engine.backward_mode(field.model, B, 2, grid, method, table=True, kl_column=Hl) == 1:
                    Y = _ComposedSolve.apply(field.model, coeffs, grid, widen(dW), method, key, 0, None, y0a, field.flat(dev, grad=True),
                                             tab.detach(), widen(dU), kl)
                    return Y[:, :, :Hl + 1].to(y0.dtype)
            else:
                flat, tab = field.inference_inputs(tt, dev)
                call = engine.SolveCall(field.model, flat, coeffs, grid, y0a.detach().to(torch.float32).contiguous(), dW=widen(dW),
                                        dU=widen(dU), method=method, seed=key, noise_table=tab, kl_column=kl)
                return call.launch()[:, :, :Hl + 1].to(y0.dtype)
        except engine._lib.SnsdeError as exc:
            if exc.code not in (-4, -6):      # (no kernel for this shape: the split solve below)
                raise
    # (2) the split solve: latent dynamics fused, the accumulator as one batched quadrature over every state
    dW, dU = increments()[:2]
    full = engine.every_step_grid(grid)
    tab_times = full.d_t0 if method != 'srk' else engine.srk_stage_times(full)
    y0p = torch.nn.functional.pad(y0[:, :Hl], (0, P - Hl))
    if needs_grad:
        if engine.backward_mode(field.model, B, 2, full, method, table=True) != 1:
            return fallback()
        flat = field.flat(dev, grad=True)
        # grad=True = no cache key, i.e. no host read-back.  The reference's sigma is a buffer (no gradient); a module whose diffusion
        # is learnable keeps the table in the autograd graph, and _ComposedSolve returns dL/d table like it does for the tutorial fields
        tab = field.noise_table(tab_times, dev, grad=True)
        if not tab.requires_grad:
            tab = tab.detach()
        try:
            Y = _ComposedSolve.apply(field.model, coeffs, full, widen(dW), method, 0, 0, None, y0p, flat, tab, widen(dU))
        except engine._lib.SnsdeError as exc:
            if exc.code not in (-4, -6):      # no kernel / LDS budget: the tensor-op loop on the increments already drawn
                raise
            return fallback()
    else:
        flat, tab = field.inference_inputs(tab_times, dev)
        call = engine.SolveCall(field.model, flat, coeffs, full, y0p.detach().to(torch.float32).contiguous(), dW=widen(dW),
                                dU=widen(dU), method=method, seed=0, noise_table=tab)
        try:
            Y = call.launch()
        except engine._lib.SnsdeError as exc:
            if exc.code not in (-4, -6):
                raise
            return fallback()
    N = grid.N
    lat = Y[:, :, :Hl].to(y0.dtype)                               # (N + 1, B, Hl)
    f_aug, g_aug = _call(sde, names, 'drift', 'f'), _call(sde, names, 'diffusion', 'g')
    t_rows = full.d_t0.to(y0.dtype).reshape(N, 1, 1).expand(N, B, 1).reshape(N * B, 1)
    h_rows = torch.from_numpy(grid.t1 - grid.t0).to(device=dev, dtype=y0.dtype).reshape(N, 1, 1).expand(N, B, 1).reshape(N * B, 1)
    ya = torch.cat([lat[:-1], torch.zeros(N, B, 1, device=dev, dtype=y0.dtype)], dim=-1).reshape(N * B, Hl + 1)
    try:
        if method == 'srk':
            inc = _srk_step(f_aug, g_aug, t_rows, h_rows, ya, dW.reshape(N * B, -1), dU.reshape(N * B, -1))[:, -1]
        else:            # Euler; Milstein's correction g dg/dy vanishes for the constant diffusion (checked by the probe's g)
            inc = f_aug(t_rows, ya)[:, -1] * h_rows[:, 0] + g_aug(t_rows, ya)[:, -1] * dW.reshape(N * B, -1)[:, -1]
    except (RuntimeError, ValueError, TypeError, IndexError):
        return fallback()
    acc = torch.cat([y0[:, Hl].unsqueeze(0), y0[:, Hl].unsqueeze(0) + torch.cumsum(inc.reshape(N, B), dim=0)], dim=0)
    aug = torch.cat([lat, acc.unsqueeze(-1)], dim=-1)             # (N + 1, B, Hl + 1): every state of the augmented solve
    idx = torch.from_numpy(grid.out_step.astype(np.int64)).to(dev)
    w = torch.from_numpy(grid.out_w).to(device=dev, dtype=y0.dtype)
    if bool((grid.out_w[:, 0] == 0).all()):
        outs = aug.index_select(0, idx + 1)
    else:          # outputs inside a step: torchsde's linear interpolation between the step's end states
        outs = w[:, 0].reshape(-1, 1, 1) * aug.index_select(0, idx) + w[:, 1].reshape(-1, 1, 1) * aug.index_select(0, idx + 1)
    return torch.cat([y0.unsqueeze(0).to(aug.dtype), outs], dim=0)


def _sdeint_padded(sde, rec, pad, coeffs, grid, y0, dW, dU, method, seed, options, row_out, needs_grad):
    """Solve the zero-padded model (engine.padding_plan) on the MFMA kernels and drop the padded state components."""
    model, layout, _ = rec
    model_p, layout_p, _, P = pad
    H, dev = model.hidden_channels, y0.device
    if needs_grad and engine.backward_mode(model_p, int(y0.shape[0]), coeffs.shape[1] + 1, grid, method) != 1:
        return None
    flat = engine.padded_flat(sde, layout, layout_p, H, P, dev, needs_grad)
    widen = lambda t: None if t is None else torch.nn.functional.pad(t, (0, P - H)).contiguous()
    row_offset = int(options.get('row_offset', 0))
    if needs_grad:
        ys = _ComposedSolve.apply(model_p, coeffs, grid, widen(dW), method, seed, row_offset, row_out,
                                  torch.nn.functional.pad(y0, (0, P - H)), flat, None, widen(dU))
    else:
        call = engine.SolveCall(model_p, flat, coeffs, grid, widen(y0.detach().to(torch.float32)), dW=widen(dW), method=method,
                                seed=seed, row_offset=row_offset, dU=widen(dU), row_out=row_out)
        ys = call.launch().to(y0.dtype)
    return ys[..., :H]


class _ComposedSolve(torch.autograd.Function):
    """Differentiable fused solve of a composed (tutorial-style) field: inputs are the composed parameter block and the
    time-only diffusion table, both produced by ordinary torch ops from the module's parameters, so autograd carries the
    gradients this node returns (dL/dy0, dL/d block, dL/d table) on to the module."""

    @staticmethod
    def forward(ctx, model, coeffs, grid, dW, method, seed, row_offset, row_out, y0, flat, tab, dU=None, kl_column=None):
        y0c = y0.detach().to(torch.float32).contiguous()
        call = engine.SolveCall(model, flat.detach().contiguous(), coeffs, grid, y0c, dW=dW, method=method, seed=seed,
                                row_offset=row_offset, row_out=row_out, dU=dU,
                                noise_table=None if tab is None else tab.detach().contiguous(),
                                save_traj=True, save_dW=True, save_act=True, kl_column=kl_column)
        ys = call.launch()
        ctx.call, ctx.y0_dtype, ctx.has_tab = call, y0.dtype, tab is not None
        return ys.to(y0.dtype) if y0.dtype != ys.dtype else ys.detach()

    @staticmethod
    def backward(ctx, grad_ys):
        call = ctx.call
        out = engine.backward_with_gradients(call, grad_ys.to(torch.float32).contiguous(), adj0_only=engine.adj0_suffices(call),
                                             want_table_grad=ctx.has_tab)
        adj, gflat, gtab = out if ctx.has_tab else (out + (None,))
        return (None,) * 8 + (adj[0].to(ctx.y0_dtype), gflat, gtab, None, None)


class _FusedSolve(torch.autograd.Function):
    """Differentiable fused solve.  forward = the HIP solve in training mode (keeps every state, the increments used
    and, on the MFMA path, the per-pass activations); backward = the HIP adjoint recursion (snsde_solve_backward) for
    dL/dy0 and every adjoint a_n, then the parameter gradients: mode 1 (MFMA path) the native split-R weight-gradient
    pass (snsde_param_gradients), mode 2 (generic adjoint kernels) ONE batched evaluation of the step function over
    all (step, row) pairs whose autograd yields them.  This replaces autograd through the ~25 x N nodes of the
    unrolled loop (benchmark_classification/common_sde.py:158-160)."""

    @staticmethod
    def forward(ctx, sde, rec, coeffs, grid, times_host, increments, method, seed, options, y0, *params):
        model, layout, numel = rec
        dW, dU = increments
        flat = engine.flatten_params(sde, layout, numel, y0.device)
        y0c = y0.detach().to(torch.float32).contiguous()

        def make(kernel, save_act):
            # the increments are kept only where the backward cannot get them otherwise: the MFMA Euler / Milstein adjoint reads
            # supplied ones in place and REGENERATES Philox ones (host key) - one (N, B, H) store and load less per step
            nets = model.noise_option in (14, 15, 18, 19)
            keep_dw = not (save_act and method in ('euler', 'milstein') and not (nets and method == 'milstein')
                           and not torch.is_tensor(seed) and options.get('param_pass', 'hip') in ('hip', 'split')
                           and os.environ.get('SNSDE_KEEP_INCREMENTS') != '1')
            return engine.SolveCall(model, flat, coeffs, grid, y0c, dW=dW, method=method, seed=seed,
                                    row_offset=int(options.get('row_offset', 0)), kernel=kernel, save_traj=True,
                                    save_dW=keep_dw, save_act=save_act, exact_order=bool(options.get('exact_order', False)),
                                    row_out=options.get('row_out'), dU=dU)
        mode = engine.backward_mode(model, y0c.shape[0], coeffs.shape[1] + 1, grid, method, options.get('kernel', 'auto'),
                                    bool(options.get('exact_order', False)))
        if mode == 0:
            raise NotImplementedError(
                "the fused backward covers 'euler', 'srk' and 'milstein' for every noise_option (Milstein: all but 7, "
                "sqrt(y), whose derivative is not finite at the clipped values), within the LDS budget of the generic "
                "adjoint kernels; pass options={'backend': 'torch'} to differentiate this configuration through the "
                "tensor-op loop")
        # recompute mode (options={'recompute': steps per chunk} or SNSDE_RECOMPUTE_STEPS): keep states and increments only,
        # re-run the forward kernel chunk by chunk inside backward (engine.backward_recompute)
        ctx.recompute = 0
        if mode == 1 and method != 'srk':
            ctx.recompute = max(int(options.get('recompute', os.environ.get('SNSDE_RECOMPUTE_STEPS', 0)) or 0), 0)
            if ctx.recompute >= grid.N:      # one chunk = the whole solve: the saved-activation mode with a second forward on top
                ctx.recompute = 0            # (and the parent's states / increments kept beside the chunk's: MORE memory, K5 N = 49)
        # mode 2: the generic adjoint prepares its own weights, so the forward takes whatever kernel is fastest
        call = make(options.get('kernel', 'auto'), mode == 1 and not ctx.recompute)
        ctx.mode, ctx.method = mode, method
        ctx.param_pass = options.get('param_pass', 'hip')
        ctx.layout = (layout, numel)
        ys = call.launch()
        ctx.call, ctx.sde, ctx.grid, ctx.times_host = call, sde, grid, times_host
        ctx.y0_dtype = y0.dtype
        # a NEW tensor object for the output: returning call.ys itself would give it this node as grad_fn, and the node
        # holds the call: a reference cycle that keeps every saved tensor of the solve alive until the cyclic collector runs
        return ys.to(y0.dtype) if y0.dtype != ys.dtype else ys.detach()

    @staticmethod
    def backward(ctx, grad_ys):
        call, sde, grid = ctx.call, ctx.sde, ctx.grid
        if ctx.mode == 1 and ctx.recompute:
            g0, flat = engine.backward_recompute(call, grad_ys.to(torch.float32).contiguous(), ctx.recompute)
            grads = engine.param_index(sde, ctx.layout[0]).grads_from_flat(flat)
            return (None,) * 9 + (g0.to(ctx.y0_dtype),) + tuple(grads)
        if ctx.mode == 1:     # MFMA adjoint kernel + native weight-gradient pass on the saved activations / deltas
            if ctx.param_pass == 'torch':     # library-GEMM cross-check of the native pass
                adj, delta = engine.solve_backward(call, grad_ys.to(torch.float32).contiguous(), save_delta=True, adj0_only=False)
                grads = _parameter_gradients_gemm(sde, call, grid, adj, delta, method=ctx.method)
            else:
                if ctx.param_pass == 'split':     # the two C calls one after the other (what the fused call must reproduce bit for bit)
                    adj, delta = engine.solve_backward(call, grad_ys.to(torch.float32).contiguous(), save_delta=True,
                                                       adj0_only=engine.adj0_suffices(call))
                    flat = engine.param_gradients(call, adj, delta)
                else:
                    adj, flat = engine.backward_with_gradients(call, grad_ys.to(torch.float32).contiguous(),
                                                               adj0_only=engine.adj0_suffices(call))
                grads = engine.param_index(sde, ctx.layout[0]).grads_from_flat(flat)
        else:                 # generic adjoint kernels (any dims; Euler / Milstein / SRK) + batched autograd parameter pass
            adj = engine.solve_backward(call, grad_ys.to(torch.float32).contiguous())
            grads = _parameter_gradients(sde, call, grid, adj, method=ctx.method)
        return (None,) * 9 + (adj[0].to(ctx.y0_dtype),) + tuple(grads)


@torch.no_grad()
def _parameter_gradients_gemm(sde, call, grid, adj, delta, method='euler'):
    """Parameter gradients from the tensors the two kernels left in HBM, as plain library GEMMs:
        d layer.weight = sum_{step,row} delta_layer^T . layer_input,   d layer.bias = sum delta_layer
    (delta from the adjoint kernel, layer inputs from the forward's act_save / trajectory), plus the elementwise
    diffusion-side reductions for theta and the time-only noise MLP.  Same result as `_parameter_gradients`
    (kept as the autograd cross-check) at a fraction of the memory traffic."""
    P = dict(sde.named_parameters())
    io, no = sde.input_option, sde.noise_option
    if no not in (0, 12, 13, 16, 17):
        raise NotImplementedError("the library-GEMM cross-check pass covers noise_option 0/12/13/16/17 only; use param_pass='hip'")
    N, B, H = call.dW_out.shape
    dev = adj.device
    NB = N * B
    slots = call.act_save.shape[1]
    nhid = slots - 2
    act = call.act_save          # (N, slots, B, H): z0, hidden.., zout
    grads = {k: None for k in P}
    t0 = torch.from_numpy(grid.step_tab[:, 0].copy()).to(dev)
    hh = torch.from_numpy(grid.step_tab[:, 1].copy()).to(dev)
    Y = call.traj[:-1]
    # ---- drift side -----------------------------------------------------------------------------------
    def wgrad(d3, x3):
        # sum_{n,b} d[n,b,:]^T x[n,b,:] as a batched GEMM over the steps (K = B per batch entry) + a small sum:
        # ~5x faster than one (H x N*B)(N*B x K) GEMM, whose reduction dimension is 1e5 long
        return torch.bmm(d3.transpose(1, 2), x3).sum(0)

    d_out = delta[:, 0]
    grads['linear_out.weight'] = wgrad(d_out, act[:, nhid])
    grads['linear_out.bias'] = d_out.sum((0, 1))
    for l in range(nhid):
        d_l = delta[:, nhid - l]
        grads[f'linears.{l}.weight'] = wgrad(d_l, act[:, l])
        grads[f'linears.{l}.bias'] = d_l.sum((0, 1))
    d0 = delta[:, nhid + 1]                           # (N, B, H) w.r.t. the pre-activation of z0
    if io in (3, 4, 5, 6):
        tau = torch.stack([t0.sin(), t0.cos()], dim=-1).unsqueeze(1).expand(N, B, 2)
        yin = torch.cat([tau, Y], dim=-1)
    else:
        yin = Y
    if io in (2, 4, 6):
        idx = torch.from_numpy(grid.step_tab[:, 5].copy().view('int32').astype('int64')).to(dev)
        frac = torch.from_numpy(grid.step_tab[:, 4].copy()).to(dev).view(N, 1, 1)
        coeffs = call.keep[1]
        Cn = coeffs.shape[-1] // 4
        rows = coeffs[:, idx, :].permute(1, 0, 2)
        a_, b_, c2, d3 = (rows[..., k * Cn:(k + 1) * Cn] for k in range(4))
        Xraw = a_ + (b_ + (0.5 * c2 + d3 * frac / 3) * frac) * frac                     # (N, B, C)
        yy = torch.baddbmm(P['linear_in.bias'], yin, P['linear_in.weight'].t().expand(N, -1, -1))
        Xt = torch.baddbmm(P['initial_network.bias'], Xraw, P['initial_network.weight'].t().expand(N, -1, -1))
        grads['emb.weight'] = torch.cat([wgrad(d0, yy), wgrad(d0, Xt)], dim=1)
        grads['emb.bias'] = d0.sum((0, 1))
        dcat = torch.matmul(d0, P['emb.weight'])
        d_in, d_x = dcat[..., :H], dcat[..., H:]
        grads['initial_network.weight'] = wgrad(d_x, Xraw)
        grads['initial_network.bias'] = d_x.sum((0, 1))
    else:
        d_in = d0
    grads['linear_in.weight'] = wgrad(d_in, yin)
    grads['linear_in.bias'] = d_in.sum((0, 1))
    # ---- diffusion side: g = tanh(sigmoid(theta) * nan_to_num(raw)), raw = s_n (no 12,16) or s_n * y (13,17) ----
    if no in (12, 13, 16, 17):
        sig = P['theta'].sigmoid()
        with torch.enable_grad():
            tn = torch.cat([t0.sin().unsqueeze(-1), t0.cos().unsqueeze(-1)], dim=-1)      # (N, 2)
            net = sde.noise_t
            s_n = net(tn)
            if no >= 16:
                s_n = s_n.relu()
        sd = s_n.detach().unsqueeze(1)                                                  # (N, 1, H)
        raw = sd * Y if no in (13, 17) else sd.expand(N, B, H)
        finite = torch.isfinite(raw)
        rc = torch.nan_to_num(raw)
        g = (sig * rc).tanh()
        du = adj[1:] * call.dW_out * (1 - g * g)
        dtheta = (du * rc).sum()
        ds = du * sig * finite
        ds = (ds * Y).sum(1) if no in (13, 17) else ds.sum(1)                           # (N, H)
        if method == 'milstein' and no in (13, 17):
            # + a . d/dc [1/2 (dW^2 - h) g g'],  g' = (1 - g^2) c,  c = sigmoid(theta) s_n   (zero for y-independent g)
            c = sig * sd
            q = call.dW_out * call.dW_out - hh.view(N, 1, 1)
            ex = adj[1:] * (1 - g * g) * (0.5 * q) * ((1 - 3 * g * g) * c * Y + g) * finite
            dtheta = dtheta + (ex * sd).sum()
            ds = ds + (ex * sig).sum(1)
        grads['theta'] = (dtheta * sig * (1 - sig)).reshape(1, 1)
        net_params = [p for p in net.parameters()]
        gs = torch.autograd.grad(s_n, net_params, grad_outputs=ds)
        for (name, _), gval in zip(net.named_parameters(), gs):
            grads['noise_t.' + name] = gval
    else:   # no == 0: theta receives no gradient (g == 0)
        grads['theta'] = torch.zeros_like(P['theta'])
    return [grads[k] if grads[k] is not None else torch.zeros_like(P[k]) for k in P]


def _parameter_gradients(sde, call, grid, adj, max_rows=1 << 19, method='euler'):
    """sum over steps n and rows of  a_{n+1} . d(f(t_n, y_n) h_n + g(t_n, y_n) dW_n)/d theta  with y_n, a_{n+1}, dW_n
    constants: one batched forward of the vector field over (step, row) pairs + autograd (library GEMMs)."""
    from . import modules
    params = list(sde.parameters())
    P = dict(sde.named_parameters())
    io, no = sde.input_option, sde.noise_option
    N, B, H = call.dW_out.shape
    dev = adj.device
    t0 = torch.from_numpy(grid.step_tab[:, 0].copy()).to(dev)
    hh = torch.from_numpy(grid.step_tab[:, 1].copy()).to(dev)
    idx = torch.from_numpy(grid.step_tab[:, 5].copy().view('int32').astype('int64')).to(dev)
    frac = torch.from_numpy(grid.step_tab[:, 4].copy()).to(dev)
    coeffs = call.keep[1]
    Cn = coeffs.shape[-1] // 4
    uses_x = io in (0, 2, 4, 6)
    total = [torch.zeros_like(p) for p in params]
    steps_per_chunk = max(1, max_rows // B)
    with torch.enable_grad():
        for lo in range(0, N, steps_per_chunk):
            hi = min(N, lo + steps_per_chunk)
            n = hi - lo
            Y = call.traj[lo:hi].reshape(n * B, H)
            A = adj[lo + 1:hi + 1].reshape(n * B, H)
            DW = call.dW_out[lo:hi].reshape(n * B, H)
            col = t0[lo:hi].repeat_interleave(B).unsqueeze(-1)
            hcol = hh[lo:hi].repeat_interleave(B).unsqueeze(-1)
            tau = torch.cat([col.sin(), col.cos()], dim=-1)
            Xraw = None
            if uses_x:
                rows = coeffs[:, idx[lo:hi], :].permute(1, 0, 2)                      # (n, B, 4C)
                fr = frac[lo:hi].view(n, 1, 1)
                a_, b_, c2, d3 = (rows[..., k * Cn:(k + 1) * Cn] for k in range(4))
                Xraw = (a_ + (b_ + (0.5 * c2 + d3 * fr / 3) * fr) * fr).reshape(n * B, Cn)
            if method == 'srk':
                surrogate = (A * _srk_rows(P, io, no, grid, lo, hi, B, Y, DW, call.dU_out[lo:hi].reshape(n * B, H), coeffs,
                                           hcol)).sum()
                gs = torch.autograd.grad(surrogate, params, allow_unused=True)
                for acc, gpart in zip(total, gs):
                    if gpart is not None:
                        acc.add_(gpart)
                continue
            f = modules.drift_rows(P, io, tau, Y, Xraw)
            if method == 'milstein' and no in (14, 15, 18, 19):
                # diffusion net: torchsde's Milstein term is the VJP of g with cotangent g (dW^2 - h) (dense dg/dy)
                Yg = Y.detach().requires_grad_(True)
                g = modules.diffusion_rows(P, no, col, tau, Yg)
                gdg, = torch.autograd.grad(g, Yg, grad_outputs=g * (DW * DW - hcol), create_graph=True)
                surrogate = (A * (f * hcol + g * DW + 0.5 * gdg)).sum()
            elif method == 'milstein':   # + 1/2 g dg/dy (dW^2 - h); g is elementwise in y for the other options
                Yg = Y.detach().requires_grad_(True)
                g = modules.diffusion_rows(P, no, col, tau, Yg)
                if g.requires_grad and g.grad_fn is not None:
                    dg, = torch.autograd.grad(g.sum(), Yg, create_graph=True, allow_unused=True)
                else:
                    dg = None
                dg = torch.zeros_like(Yg) if dg is None else dg       # diffusion independent of y: no Milstein term
                surrogate = (A * (f * hcol + g * DW + 0.5 * g * dg * (DW * DW - hcol))).sum()
            else:
                g = modules.diffusion_rows(P, no, col, tau, Y)
                surrogate = (A * (f * hcol + g * DW)).sum()
            gs = torch.autograd.grad(surrogate, params, allow_unused=True)
            for acc, gpart in zip(total, gs):
                if gpart is not None:
                    acc.add_(gpart)
    return total


def _srk_rows(P, io, no, grid, lo, hi, B, Y, I_k, I_k0, coeffs, hcol):
    """One SRID2 step of every (step, row) pair of steps lo..hi-1 as a differentiable function of the parameters
    (states, increments constant): the stage times / spline intervals come from the solver's stage table."""
    from . import modules
    tab = engine.srk_table(grid)[lo:hi]   # (n, 4, stride): t, sin t, cos t, frac, interval index
    n = hi - lo
    Cn = coeffs.shape[-1] // 4
    uses_x = io in (0, 2, 4, 6)

    def at(slot):
        t = tab[:, slot, 0].to(Y.dtype).repeat_interleave(B).unsqueeze(-1)
        tau = torch.stack([tab[:, slot, 1], tab[:, slot, 2]], dim=-1).to(Y.dtype).repeat_interleave(B, dim=0)
        Xraw = None
        if uses_x:
            idx = tab[:, slot, 4].contiguous().view(torch.int32).to(torch.int64)
            fr = tab[:, slot, 3].to(Y.dtype).view(n, 1, 1)
            rows = coeffs[:, idx, :].permute(1, 0, 2)
            a_, b_, c2, d3 = (rows[..., k * Cn:(k + 1) * Cn] for k in range(4))
            Xraw = (a_ + (b_ + (0.5 * c2 + d3 * fr / 3) * fr) * fr).reshape(n * B, Cn)
        return t, tau, Xraw

    slots_f = (0, 3, 2, 0)      # C0 = 0, 1, 1/2, 0  -> stage-table slots (0, 1/4, 1/2, 1)
    slots_g = (0, 1, 3, 1)      # C1 = 0, 1/4, 1, 1/4

    def f(stage, y):
        t, tau, Xraw = at(slots_f[stage])
        return modules.drift_rows(P, io, tau, y, Xraw)

    def g(stage, y):
        t, tau, _ = at(slots_g[stage])
        return modules.diffusion_rows(P, no, t, tau, y)

    T = _SRK
    h = hcol
    rdt = h.sqrt()
    I_kk = (I_k * I_k - h) / 2
    I_kkk = (I_k ** 3 - 3 * h * I_k) / 6
    fs, gs = [], []
    y1 = Y
    for s in range(4):
        H0, H1 = Y, Y
        for j in range(s):
            H0 = H0 + T['A0'][s][j] * fs[j] * h + T['B0'][s][j] * gs[j] * I_k0 / h
            H1 = H1 + T['A1'][s][j] * fs[j] * h + T['B1'][s][j] * gs[j] * rdt
        fs.append(f(s, H0) if T['alpha'][s] != 0.0 or any(T['A0'][k][s] != 0.0 or T['A1'][k][s] != 0.0 for k in range(s + 1, 4))
                  else torch.zeros_like(Y))
        gs.append(g(s, H1))
        gw = T['beta1'][s] * I_k + T['beta2'][s] * I_kk / rdt + T['beta3'][s] * I_k0 / h + T['beta4'][s] * I_kkk / h
        y1 = y1 + T['alpha'][s] * fs[s] * h + gw * gs[s]
    return y1


def _call(sde, names, key, default):
    return getattr(sde, (names or {}).get(key, default))


# Roessler's SRI2W1 (SIAM J. Numer. Anal. 48(3), 2010) = torchsde 0.2.5 `tableaus/srid2.py`, the scheme behind method='srk' for
# diagonal noise; the same numbers as csrc/snsde_internal.h (SRK_B1_*, srk_w*) and, independently transcribed, oracle/sde_oracle.py
_SRK = dict(C0=(0.0, 1.0, 0.5, 0.0), C1=(0.0, 0.25, 1.0, 0.25),
            A0=((), (1.0,), (0.25, 0.25), (0.0, 0.0, 0.0)), A1=((), (0.25,), (1.0, 0.0), (0.0, 0.0, 0.25)),
            B0=((), (0.0,), (1.0, 0.5), (0.0, 0.0, 0.0)), B1=((), (-0.5,), (1.0, 0.0), (2.0, -1.0, 0.5)),
            alpha=(1 / 6, 1 / 6, 2 / 3, 0.0), beta1=(-1.0, 4 / 3, 2 / 3, 0.0), beta2=(1.0, -4 / 3, 1 / 3, 0.0),
            beta3=(2.0, -4 / 3, -2 / 3, 0.0), beta4=(-2.0, 5 / 3, -2 / 3, 1.0))


def _srk_step(f, g, t0, h, y, I_k, I_k0):
    """One SRID2 step.  Terms whose tableau coefficient is zero are not formed (x + 0 * z = x exactly for finite z: same values,
    a third of the tensor ops), nor is the drift of the last stage, which no later term reads (alpha[3] = 0)."""
    T = _SRK
    rdt = h.sqrt()
    I_kk = (I_k * I_k - h) / 2
    I_kkk = (I_k ** 3 - 3 * h * I_k) / 6
    fs, gs = [], []
    y1 = y
    for s in range(4):
        H0, H1 = y, y
        for j in range(s):
            if T['A0'][s][j] != 0.0:
                H0 = H0 + T['A0'][s][j] * fs[j] * h
            if T['B0'][s][j] != 0.0:
                H0 = H0 + T['B0'][s][j] * gs[j] * I_k0 / h
            if T['A1'][s][j] != 0.0:
                H1 = H1 + T['A1'][s][j] * fs[j] * h
            if T['B1'][s][j] != 0.0:
                H1 = H1 + T['B1'][s][j] * gs[j] * rdt
        need_f = T['alpha'][s] != 0.0 or any(T['A0'][r][s] != 0.0 or T['A1'][r][s] != 0.0 for r in range(s + 1, 4))
        fs.append(f(t0 + T['C0'][s] * h, H0) if need_f else None)
        gs.append(g(t0 + T['C1'][s] * h, H1))
        gw = None
        for coef, term in ((T['beta1'][s], lambda c: c * I_k), (T['beta2'][s], lambda c: c * I_kk / rdt),
                           (T['beta3'][s], lambda c: c * I_k0 / h), (T['beta4'][s], lambda c: c * I_kkk / h)):
            if coef != 0.0:        # (same expression forms as the unskipped sum: bit-identical values)
                gw = term(coef) if gw is None else gw + term(coef)
        if T['alpha'][s] != 0.0:
            y1 = y1 + T['alpha'][s] * fs[s] * h
        if gw is not None:
            y1 = y1 + gw * gs[s]
    return y1


def _sdeint_torch(sde, y0, ts, bm, method, dt, options, names):
    """Unfused scheme on tensor ops (arbitrary sde, CPU plumbing, autograd)."""
    f = _call(sde, names, 'drift', 'f')
    g = _call(sde, names, 'diffusion', 'g')
    noise_type = getattr(sde, 'noise_type', 'diagonal')
    if noise_type == 'scalar':
        # torchsde's scalar noise: g is (B, H, 1), one Brownian motion per row, g_prod = g[..., 0] * I with I (B, 1)
        # (the tutorial's Neural ODE notebook solves an ODE this way, g = 0).  Euler only: the higher-order schemes need
        # the dense Jacobian-vector product of g there.
        if method != 'euler':
            raise NotImplementedError("scalar noise: only method='euler' is implemented")
        g_scalar = g
        g = lambda t, y: g_scalar(t, y).squeeze(-1)
    elif noise_type != 'diagonal':
        raise NotImplementedError("only diagonal (and, under Euler, scalar) noise is implemented")
    if getattr(sde, 'sde_type', 'ito') != 'ito':
        raise NotImplementedError("only Ito SDEs are implemented")
    ts_host = _HostTimes.get(ts)
    grid = engine.StepGrid(ts_host, dt, np.array([0.0, 1.0], dtype=np.float32), None)
    t0s = torch.from_numpy(grid.t0).to(y0.device)
    t1s = torch.from_numpy(grid.t1).to(y0.device)
    w = torch.from_numpy(grid.out_w).to(device=y0.device, dtype=y0.dtype)
    hs = (t1s - t0s).to(y0.dtype)
    # every increment of the solve up front: (N, B, H) I_k (and I_k0 for SRK)
    if bm is None:
        gen = torch.Generator(device=y0.device)
        seed = options.get('seed')
        gen.manual_seed(int(seed) if seed is not None and not torch.is_tensor(seed) else _fresh_seed())
        hcol = hs.reshape(-1, *([1] * y0.dim()))
        wshape = (grid.N, y0.shape[0], 1) if noise_type == 'scalar' else (grid.N,) + tuple(y0.shape)
        dW_all = torch.randn(wshape, dtype=y0.dtype, device=y0.device, generator=gen) * hcol.sqrt()
        dU_all = None
        if method == 'srk':      # I_k0 = h (I_k / 2 + sqrt(h / 12) xi): the space-time Levy integral
            xi = torch.randn((grid.N,) + tuple(y0.shape), dtype=y0.dtype, device=y0.device, generator=gen)
            dU_all = hcol * (0.5 * dW_all + (hcol / 12).sqrt() * xi)
    elif method == 'srk':
        pairs = [bm(t0s[n], t1s[n], return_U=True) for n in range(grid.N)]
        dW_all = torch.stack([p[0].to(device=y0.device, dtype=y0.dtype) for p in pairs])
        dU_all = torch.stack([p[1].to(device=y0.device, dtype=y0.dtype) for p in pairs])
    else:
        dW_all = torch.stack([bm(t0s[n], t1s[n]).to(device=y0.device, dtype=y0.dtype) for n in range(grid.N)])
        dU_all = None
    needs_grad = torch.is_grad_enabled() and (y0.requires_grad or any(
        p.requires_grad for p in getattr(sde, 'parameters', lambda: [])()))
    if (y0.is_cuda and not needs_grad and method in ('euler', 'srk') and options.get('graph', True)
            and not torch.cuda.is_current_stream_capturing()):
        ys = _graphed_steps(f, g, y0, grid, t0s, hs, w, dW_all, dU_all, method)
        if ys is not None:
            return _select_rows(ys, options)
    y = y0
    ys = [y0]
    k = 0
    for n in range(grid.N):
        t0, h = t0s[n], hs[n]
        prev = y
        I = dW_all[n]
        if method == 'srk':
            y = _srk_step(f, g, t0, h, y, I, dU_all[n])
        elif method == 'euler':
            y = y + f(t0, y) * h + g(t0, y) * I
        else:
            v = I * I - h
            # g * dg/dy * v for diagonal noise; when differentiating, the cotangent keeps its dependence on y so that
            # autograd through this loop is the exact gradient of the discrete scheme (what the fused adjoint computes)
            diff = torch.is_grad_enabled() and (y.requires_grad or any(
                p.requires_grad for p in getattr(sde, 'parameters', lambda: [])()))
            with torch.enable_grad():
                yy = y if y.requires_grad else y.detach().requires_grad_(True)
                gv = g(t0, yy)
                if gv.requires_grad:
                    gdg, = torch.autograd.grad(gv, yy, grad_outputs=(gv if diff else gv.detach()) * v, allow_unused=True,
                                               create_graph=diff)
                else:            # a diffusion that depends on neither the state nor a parameter (a constant buffer): no correction
                    gdg = None
            gv = gv if diff else gv.detach()
            gdg = torch.zeros_like(y) if gdg is None else gdg
            y = y + f(t0, y) * h + gv * I + 0.5 * gdg
        while k < grid.T - 1 and grid.out_step[k] == n:
            ys.append(y if grid.out_w[k, 0] == 0 else w[k, 0] * prev + w[k, 1] * y)
            k += 1
    return _select_rows(torch.stack(ys, dim=0), options)


def _select_rows(ys, options):
    row_out = (options or {}).get('row_out')
    if row_out is not None:     # same contract as the fused path: each row's own output state, (B, H)
        idx = row_out.to(device=ys.device, dtype=torch.int64).reshape(1, -1, 1).expand(1, ys.shape[1], ys.shape[2])
        ys = ys.gather(0, idx).squeeze(0)
    return ys


def _graphed_steps(f, g, y0, grid, t0s, hs, w, dW_all, dU_all, method):
    """Generic-`sde` stepper for CUDA tensors without gradients: ONE solver step of the user's f / g —
    time, step size and increments picked from device arrays by an in-graph step counter, state updated in place — is
    recorded into a hipGraph inside this call and replayed N times, so a step costs one graph launch instead of the
    ~25-100 kernel launches of the tensor-op loop.  The graph lives only for this call (every tensor the user's module
    touches is alive and fixed meanwhile).  Returns None — the caller then runs the eager loop — when the step cannot be
    recorded (a device->host sync or data-dependent control flow inside f / g)."""
    dev = y0.device
    y_buf = y0.detach().clone()
    prev_buf = y_buf.clone()
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)

    def step():
        t = t0s.index_select(0, cnt).squeeze(0)
        h = hs.index_select(0, cnt).squeeze(0)
        I = dW_all.index_select(0, cnt).squeeze(0)
        prev_buf.copy_(y_buf)
        if method == 'euler':
            y_new = y_buf + f(t, y_buf) * h + g(t, y_buf) * I
        else:
            y_new = _srk_step(f, g, t, h, y_buf, I, dU_all.index_select(0, cnt).squeeze(0))
        y_buf.copy_(y_new)
        cnt.add_(1)

    prev_mode = torch.cuda.get_sync_debug_mode()
    try:
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            torch.cuda.set_sync_debug_mode('error')        # a sync inside f / g means the step cannot be recorded
            step()                                          # torch's capture recipe: one eager run on a side stream
            torch.cuda.set_sync_debug_mode(prev_mode)
        torch.cuda.current_stream(dev).wait_stream(side)
        y_buf.copy_(y0); cnt.zero_()
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph):
            step()
    except Exception as exc:
        torch.cuda.set_sync_debug_mode(prev_mode)
        torch.cuda.synchronize(dev)
        if os.environ.get('SNSDE_DEBUG_GRAPH') == '1':
            warnings.warn(f'generic-sde step not recorded: {type(exc).__name__}: {exc}')
        return None
    ys = torch.empty((grid.T,) + tuple(y0.shape), dtype=y0.dtype, device=dev)
    ys[0].copy_(y0)
    k = 0
    with torch.no_grad():
        for n in range(grid.N):
            graph.replay()
            while k < grid.T - 1 and grid.out_step[k] == n:
                if grid.out_w[k, 0] == 0:
                    ys[k + 1].copy_(y_buf)
                else:
                    ys[k + 1].copy_(w[k, 0] * prev_buf + w[k, 1] * y_buf)
                k += 1
    return ys
