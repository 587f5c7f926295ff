"""Host plumbing around the C ABI: model recognition, flat parameter block, time grid, launch.

PyTorch is used only for device memory and streams; all arithmetic of the hot path runs in
libsnsde.so (HIP).  Nothing here falls back to CPU math.
"""
import os
import ctypes as C
from collections import OrderedDict

import numpy as np
import torch

from . import _lib

_GRID_CACHE = OrderedDict()
_GRID_CACHE_MAX = 16


def model_struct(input_channels, hidden_channels, hidden_hidden_channels, num_hidden_layers, input_option,
                 noise_option, activation=0, drift_output=0, diffusion_output=0, time_feature=0):
    """snsde_model; the four trailing switches (include/snsde.h SNSDE_ACT_* ...) are 0 for the reference's models."""
    return _lib.Model(int(input_channels), int(hidden_channels), int(hidden_hidden_channels),
                      int(num_hidden_layers), int(input_option), int(noise_option), int(activation),
                      int(drift_output), int(diffusion_output), int(time_feature))


def recognise(sde):
    """Fast-path contract: an object exposing the reference Diffusion_model's attributes
    (models_sde/neuralsde.py:123-184) and exactly its parameter names/shapes.  Returns
    (Model struct, layout, numel) or None."""
    need = ('input_option', 'noise_option', 'hidden_channels', 'input_channels', 'coeffs', 'times')
    if not all(hasattr(sde, a) for a in need) or not hasattr(sde, 'named_parameters'):
        return None
    cached = getattr(sde, '_snsde_rec', None)     # (key, result): parameter names/shapes are fixed after construction
    key = (sde.input_option, sde.noise_option, sde.hidden_channels, sde.input_channels, len(sde._parameters),
           len(sde._modules))
    if cached is not None and cached[0] == key:
        return cached[1]
    result = _recognise(sde)
    try:
        object.__setattr__(sde, '_snsde_rec', (key, result))
    except Exception:
        pass
    return result


class ParamIndex:
    """The module's parameters resolved ONCE: in named_parameters() order (what autograd.Function.apply receives and backward
    returns) and in the C ABI's layout order, with the per-parameter (offset, numel, shape) of the flat block.  A training step
    walked named_parameters() six times (~25 us each: half of the host time of sdeint() on the K2 step, which is host-bound);
    the cached lists are validated by identity against the owning submodules' _parameters dicts (~2 us), so a re-assigned
    parameter (module.weight = nn.Parameter(...)) or a replaced layer rebuilds them."""

    def __init__(self, sde, layout):
        named = list(sde.named_parameters())
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        by_name = dict(named)
        self.layout_params = [by_name[name] for name, _, _ in layout]
        offs = {name: (off, shape) for name, off, shape in layout}
        self.sizes = [by_name[name].numel() for name, _, _ in layout]              # split sizes of the flat block (layout order)
        order = {name: i for i, (name, _, _) in enumerate(layout)}
        self.to_layout = [order[n] for n in self.names]                             # named order -> index in the layout
        self.shapes = [tuple(offs[n][1]) for n in self.names]
        self.owners = []
        for n in self.names:
            mod, parts = sde, n.split('.')
            for q in parts[:-1]:
                mod = mod._modules[q]
            self.owners.append((mod._parameters, parts[-1]))
        self.all_f32 = all(p.dtype == torch.float32 for p in self.params)
        self.layout_obj, self.layout_key = layout, tuple(layout)      # (name, offset, shape) triples: compared by VALUE when the object differs
        # the module tree the lists were read from: every (parent's _modules dict, name, child) edge and, per module, the sizes of
        # its _parameters / _modules dicts - a replaced submodule (whose old _parameters dict still holds the old tensors), an
        # added or a removed parameter / submodule all show here without walking named_modules() again (25 us per walk)
        mods = list(sde.modules())
        self.edges = [(m._modules, k, c) for m in mods for k, c in m._modules.items()]
        self.sizes_seen = [(m._parameters, len(m._parameters), m._modules, len(m._modules)) for m in mods]
        self.n_params = sum(len(m._parameters) for m in mods)

    def valid(self, sde, layout):
        if self.layout_obj is not layout and self.layout_key != tuple(layout):
            return False
        for d, k, child in self.edges:
            if d.get(k) is not child:
                return False
        for pd, npar, md, nmod in self.sizes_seen:
            if len(pd) != npar or len(md) != nmod:
                return False
        for (d, k), p in zip(self.owners, self.params):
            if d.get(k) is not p:
                return False
        return True

    # the index rides in the module's __dict__ but must not travel with copies of it: deepcopy / pickle / torch.save
    # carry nothing, the copy is indexed afresh
    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (type(None), ())

    def grads_from_flat(self, flat):
        """Per-parameter views of a flat gradient in the C ABI's layout, in named_parameters() order."""
        pieces = flat.split(self.sizes)
        out = []
        for i, shape, p in zip(self.to_layout, self.shapes, self.params):
            g = pieces[i].view(shape)
            out.append(g if p.dtype == torch.float32 else g.to(p.dtype))
        return out


def param_index(sde, layout):
    idx = sde.__dict__.get('_snsde_pidx') if hasattr(sde, '__dict__') else None
    if idx is None or not idx.valid(sde, layout):
        idx = ParamIndex(sde, layout)
        try:
            sde.__dict__['_snsde_pidx'] = idx
        except Exception:
            pass
    return idx


def _recognise(sde):
    if getattr(sde, 'sde_type', 'ito') != 'ito' or getattr(sde, 'noise_type', 'diagonal') != 'diagonal':
        return None
    params = dict(sde.named_parameters())
    lin = params.get('linear_in.weight')
    out = params.get('linear_out.weight')
    if lin is None or out is None:
        return None
    n_hidden = sum(1 for k in params if k.startswith('linears.') and k.endswith('.weight'))
    m = model_struct(sde.input_channels, sde.hidden_channels, lin.shape[0], n_hidden + 1, sde.input_option,
                     sde.noise_option)
    try:
        layout, numel = _lib.param_layout(m)
    except _lib.SnsdeError:
        return None
    if set(params) != {n for n, _, _ in layout}:
        return None
    for name, _, shape in layout:
        if tuple(params[name].shape) != tuple(shape):
            return None
    return m, layout, numel


_MFMA_SIZES = (16, 32, 64, 128, 256)
_PAD_CACHE = {}


def forward_path(model, batch, knots, n_steps, method='euler', kernel='auto', table=False):
    """Name of the kernel family a forward solve of this shape takes (_lib.PATHS; host-side query)."""
    s = _lib.Solve()
    s.model = model
    s.batch, s.knots, s.n_steps, s.n_out = int(batch), int(knots), int(n_steps), 2
    s.method = {'euler': _lib.EULER, 'milstein': _lib.MILSTEIN, 'srk': _lib.SRK}[method]
    s.kernel = _lib.KERNELS[kernel]
    s.noise_table = C.c_void_p(16) if table else None
    return _lib.PATHS[_lib.lib().snsde_forward_path(C.byref(s))]


def padding_plan(model, batch, knots, n_steps, method):
    """Hidden sizes the MFMA kernels are not instantiated for (H not in 16 / 32 / 64 / 128 / 256, or a hidden width HH != H)
    would land on the generic VALU kernels (~13x slower).  Zero-padding the model to the next instantiated size P is exact:
    padded hidden units have zero weights and biases (relu(0) = 0), padded state components have zero output weights
    (tanh(0) = 0 drift) and are read by nothing (zero first-layer / diffusion-net columns), so the real components evolve
    exactly as in the unpadded model and the padded ones are dropped from the outputs.
    Returns (padded model struct, its layout, numel, P) when that moves the solve onto an MFMA kernel, else None."""
    H, HH = model.hidden_channels, model.hidden_hidden_channels
    if H in _MFMA_SIZES and HH == H:
        return None
    if model.activation or model.drift_output or model.diffusion_output or model.time_feature:
        return None      # field variants: the padded block below is the reference Diffusion_model's, not theirs
    key = (model.input_channels, H, HH, model.num_hidden_layers, model.input_option, model.noise_option, int(batch), int(knots),
           int(n_steps), method)
    if key not in _PAD_CACHE:
        plan = None
        P = next((p for p in _MFMA_SIZES if p >= max(H, HH)), None)
        if P is not None and model.activation == 0 and forward_path(model, batch, knots, n_steps, method) in ('generic', 'generic-srk'):
            mp = model_struct(model.input_channels, P, P, model.num_hidden_layers, model.input_option, model.noise_option)
            try:
                layout_p, numel_p = _lib.param_layout(mp)
                if forward_path(mp, batch, knots, n_steps, method) not in ('none', 'generic', 'generic-srk'):
                    plan = (mp, layout_p, numel_p, P)
            except _lib.SnsdeError:
                plan = None
        _PAD_CACHE[key] = plan
    return _PAD_CACHE[key]


def padded_flat(sde, layout, layout_p, H, P, device, grad):
    """The padded model's parameter block from the module's parameters (torch ops: differentiable when grad).  Every tensor
    sits in the top-left corner of its padded shape, except emb.weight whose two H-wide column blocks (yy | X_t) go to
    columns [0, H) and [P, P + H)."""
    params = dict(sde.named_parameters())
    pieces = []
    with torch.set_grad_enabled(grad):
        for (name, _, shape), (name_p, _, shape_p) in zip(layout, layout_p):
            assert name == name_p
            w = params[name].to(device=device, dtype=torch.float32)
            if not grad:
                w = w.detach()
            if tuple(shape) == tuple(shape_p):
                z = w
            else:
                z = w.new_zeros(tuple(shape_p))
                if name == 'emb.weight':
                    z[:shape[0], :H] = w[:, :H]
                    z[:shape[0], P:P + H] = w[:, H:]
                elif w.dim() == 2:
                    z[:shape[0], :shape[1]] = w
                else:
                    z[:shape[0]] = w
            pieces.append(z.reshape(-1))
        return torch.cat(pieces)


def flatten_params(sde, layout, numel, device):
    """One float32 device buffer in the C ABI's layout (state_dict order).

    The first call concatenates the parameters and (when they are float32 tensors on `device`) re-points every
    `param.data` at its slice of that buffer, so later calls — including after in-place optimizer steps, which now
    write straight into the buffer — return it without launching anything.  Anything that re-allocates a parameter
    (`.to()`, `.double()`, a new nn.Parameter) is detected by the address check and triggers a fresh flatten."""
    plist = param_index(sde, layout).layout_params
    arena = getattr(sde, '_snsde_flat', None)
    if arena is not None and arena.device == device:
        base = arena.data_ptr()
        if all(p.data_ptr() == base + 4 * off and p.dtype == torch.float32 for p, (_, off, _) in zip(plist, layout)):
            return arena
    flat = torch.cat([p.detach().reshape(-1).to(device=device, dtype=torch.float32) for p in plist])
    assert flat.numel() == numel
    if all(p.dtype == torch.float32 and p.device == device for p in plist) and os.environ.get('SNSDE_NO_PARAM_ARENA') != '1':
        with torch.no_grad():
            for p, (_, off, shape) in zip(plist, layout):
                p.data = flat[off:off + p.numel()].view(shape)
        try:
            object.__setattr__(sde, '_snsde_flat', flat)
        except Exception:
            pass
    return flat


class StepGrid:
    """Host + device copy of the fixed-step grid (snsde_grid_build)."""

    def __init__(self, ts_host, dt, times_host, device):
        L = _lib.lib()
        ts32 = np.ascontiguousarray(ts_host, dtype=np.float32)
        times32 = np.ascontiguousarray(times_host, dtype=np.float32)
        if ts32.ndim != 1 or ts32.shape[0] < 2:
            raise ValueError('`ts` must be a 1-D sequence of at least two times.')
        n = C.c_int32()
        rc = L.snsde_grid_count(ts32.ctypes.data, ts32.shape[0], float(dt), C.byref(n))
        if rc == -7:
            raise ValueError('Evaluation times `ts` must be strictly increasing and `dt` positive '
                             '(and large enough to advance float32 time).')
        _lib.check(rc, 'snsde_grid_count')
        self.N = n.value
        self.T = ts32.shape[0]
        self.step_tab = np.zeros((self.N, _lib.SNSDE_STEP_STRIDE), dtype=np.float32)
        self.out_step = np.zeros(self.T - 1, dtype=np.int32)
        self.out_w = np.zeros((self.T - 1, 2), dtype=np.float32)
        _lib.check(L.snsde_grid_build(ts32.ctypes.data, self.T, float(dt), times32.ctypes.data, times32.shape[0],
                                      self.N, self.step_tab.ctypes.data, self.out_step.ctypes.data,
                                      self.out_w.ctypes.data), 'snsde_grid_build')
        self._times32 = times32
        self._d_srk = None
        self.t0 = self.step_tab[:, 0].copy()
        self.t1 = self.step_tab[:, 7].copy()
        self.device = device
        if device is not None and device.type == 'cuda':
            self.d_step_tab = torch.from_numpy(self.step_tab).to(device)
            self.d_t0 = torch.from_numpy(self.t0).to(device)
            self.d_out_step = torch.from_numpy(self.out_step).to(device)
            self.d_out_w = torch.from_numpy(self.out_w).to(device)


def sub_grid(grid, n0, n1):
    """The solver steps n0 .. n1-1 of `grid` as a StepGrid of their own: the SAME step rows (times, step sizes, spline
    intervals), the parent's outputs that fall inside, and one more output at the end of the last step.  Returns
    (sub grid, ks) with ks = the parent's output numbers (k + 1) of the sub grid's outputs 1 .. len(ks); its last output
    (the state after step n1 - 1) is the chunk's hand-over to the steps behind it."""
    g = StepGrid.__new__(StepGrid)
    K = n1 - n0
    tab = grid.step_tab[n0:n1].copy()
    ks = [k for k in range(grid.T - 1) if n0 <= grid.out_step[k] < n1]
    out_step = np.array([grid.out_step[k] - n0 for k in ks] + [K - 1], dtype=np.int32)
    out_w = np.array([grid.out_w[k] for k in ks] + [(0.0, 1.0)], dtype=np.float32).reshape(-1, 2)
    nout = np.zeros(K, dtype=np.int32)
    first = np.zeros(K, dtype=np.int32)
    for j, st in enumerate(out_step):
        if nout[st] == 0:
            first[st] = j
        nout[st] += 1
    tab[:, 8] = nout.view(np.float32)
    tab[:, 9] = first.view(np.float32)
    g.N, g.T = K, len(out_step) + 1
    g.step_tab, g.out_step, g.out_w = tab, out_step, out_w
    g._times32, g._d_srk = grid._times32, None
    g.t0, g.t1 = tab[:, 0].copy(), tab[:, 7].copy()
    g.device = grid.device
    g.d_step_tab = torch.from_numpy(tab).to(grid.device)
    g.d_t0 = torch.from_numpy(g.t0).to(grid.device)
    g.d_out_step = torch.from_numpy(out_step).to(grid.device)
    g.d_out_w = torch.from_numpy(out_w).to(grid.device)
    return g, [k + 1 for k in ks]


def every_step_grid(grid):
    """`grid`'s solver steps with an output after EVERY step (T = N + 1, no interpolation): the solve then returns the whole
    trajectory as its result, so a caller can attach cotangents to any state (torchsde._sdeint_latent: the KL path integral of
    a latent SDE is a quadrature over all of them).  Memoised on the grid."""
    g = grid.__dict__.get('_every_step')
    if g is not None:
        return g
    g = StepGrid.__new__(StepGrid)
    N = grid.N
    tab = grid.step_tab.copy()
    tab[:, 8] = np.ones(N, dtype=np.int32).view(np.float32)
    tab[:, 9] = np.arange(N, dtype=np.int32).view(np.float32)
    g.N, g.T = N, N + 1
    g.step_tab = tab
    g.out_step = np.arange(N, dtype=np.int32)
    g.out_w = np.tile(np.array([[0.0, 1.0]], dtype=np.float32), (N, 1))
    g._times32, g._d_srk = grid._times32, None
    g.t0, g.t1 = tab[:, 0].copy(), tab[:, 7].copy()
    g.device = grid.device
    if grid.device is not None and grid.device.type == 'cuda':
        g.d_step_tab = torch.from_numpy(tab).to(grid.device)
        g.d_t0 = torch.from_numpy(g.t0).to(grid.device)
        g.d_out_step = torch.from_numpy(g.out_step).to(grid.device)
        g.d_out_w = torch.from_numpy(g.out_w).to(grid.device)
    grid.__dict__['_every_step'] = g
    return g


def chunk_plan(grid, chunk):
    """[(n0, n1, sub grid, parent output numbers)] for the recompute-mode backward (memoised on the grid)."""
    cache = grid.__dict__.setdefault('_chunks', {})
    if chunk not in cache:
        cache[chunk] = [(n0, min(n0 + chunk, grid.N)) + sub_grid(grid, n0, min(n0 + chunk, grid.N))
                        for n0 in range(0, grid.N, chunk)]
    return cache[chunk]


def backward_recompute(call, grad_ys, chunk, stream=None):
    """Recompute-mode backward of a solve that kept only its states and increments (SolveCall with save_traj / save_dW,
    no save_act): the steps are revisited in chunks of `chunk`, last chunk first; each chunk re-runs the forward kernel
    from its saved first state with the saved increments (bit-identical states, now with the per-step activations), then
    the adjoint kernel and the weight-gradient pass on those.  Peak extra memory = one chunk's activations and deltas
    (2 x chunk x NSAVE x B x H floats) instead of N x NSAVE x B x H for the whole solve; the chunk buffers are written
    and read back within microseconds, i.e. out of the 256 MB last-level cache rather than HBM.
    Returns (dL/dy0 (B, H), flat parameter gradient)."""
    grid, N = call.grid, call.grid.N
    model, flat, coeffs = call.model, call.keep[0], call.keep[1]
    method = {_lib.EULER: 'euler', _lib.MILSTEIN: 'milstein'}[call.desc.method]
    B, H = call.traj.shape[1:]
    dev = call.traj.device
    per_row = call.keep[6] is not None          # grad_ys is (B, H): row b's gradient belongs to output row_out[b]
    row_out = call.keep[6]
    carry = torch.zeros((B, H), device=dev, dtype=torch.float32)
    total = None
    for n0, n1, sub, ks in reversed(chunk_plan(grid, chunk)):
        g = torch.zeros((sub.T, B, H), device=dev, dtype=torch.float32)
        for j, k in enumerate(ks):                # the parent's outputs inside the chunk
            if per_row:
                g[j + 1] = torch.where((row_out == k).unsqueeze(1), grad_ys, g[j + 1])
            else:
                g[j + 1] = grad_ys[k]
        g[sub.T - 1] += carry                     # everything behind the chunk acts on its last state
        # the chunk reads the parent's increments in place (no second copy: the adjoint takes `dW` where there is no `dW_out`), and
        # its adjoint + weight-gradient pass are one C call
        c = SolveCall(model, flat, coeffs, sub, call.traj[n0], dW=call.dW_out[n0:n1], method=method, kernel='auto',
                      save_traj=True, save_dW=False, save_act=True, exact_order=bool(call.base_flags & _lib.FLAG_EXACT_ORDER))
        c.launch(stream)
        adj, part = backward_with_gradients(c, g, stream=stream, adj0_only=adj0_suffices(c))
        total = part if total is None else total.add_(part)
        carry = adj[0].clone()
        # drop this chunk's activations / deltas / workspaces BEFORE the next chunk allocates its own: the caching allocator then
        # hands the same blocks out again, and the peak is ONE chunk's buffers (two chunks were alive at once before: 50 steps per
        # chunk at K2 peaked above the saved-activation mode)
        c.keep_bwd = c.keep_pg = None
        del c, adj, part, g
    if per_row:
        g0 = torch.where((row_out == 0).unsqueeze(1), grad_ys, torch.zeros_like(grad_ys))
    else:
        g0 = grad_ys[0]
    return carry + g0, total


def srk_table(grid):
    """Device (N, 4, SNSDE_SRK_STRIDE) stage-time table of a StepGrid (built on first use)."""
    if grid._d_srk is None:
        tab = np.zeros((grid.N, 4, _lib.SNSDE_SRK_STRIDE), dtype=np.float32)
        _lib.check(_lib.lib().snsde_grid_srk_build(grid.step_tab.ctypes.data, grid.N, grid._times32.ctypes.data,
                                                   grid._times32.shape[0], tab.ctypes.data), 'snsde_grid_srk_build')
        grid.srk_host = tab
        grid._d_srk = torch.from_numpy(tab).to(grid.device)
    return grid._d_srk


def srk_stage_times(grid):
    """Device (4 N,) tensor of the SRID2 stage times t0 + {0, 1/4, 1/2, 1} h of every step (column 0 of the stage table)."""
    if getattr(grid, '_d_srk_times', None) is None:
        grid._d_srk_times = srk_table(grid)[:, :, 0].reshape(-1).contiguous()
    return grid._d_srk_times


def step_grid(ts_host, dt, times_host, device):
    key = (np.asarray(ts_host, dtype=np.float32).tobytes(), float(dt),
           np.asarray(times_host, dtype=np.float32).tobytes(), str(device))
    g = _GRID_CACHE.get(key)
    if g is None:
        g = StepGrid(ts_host, dt, times_host, device)
        _GRID_CACHE[key] = g
        if len(_GRID_CACHE) > _GRID_CACHE_MAX:
            _GRID_CACHE.popitem(last=False)
    else:
        _GRID_CACHE.move_to_end(key)
    return g


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _check_f32(name, t, shape=None):
    if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
        raise ValueError(f'{name} must be a contiguous float32 CUDA tensor')
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f'{name} has shape {tuple(t.shape)}, expected {tuple(shape)}')


class SolveCall:
    """A fully prepared solve: owns references to every device buffer named in the descriptor so the
    launch itself is one C call that only enqueues kernels (hipGraph-capturable).
    stream_all / two_tile: the A/B switches of include/snsde.h (SNSDE_FLAG_STREAM_ALL: H = 256 on the fully streamed sixteen-wave
    kernels instead of the two-tile ones; SNSDE_FLAG_TWO_TILE: H = 128 on the four-wave two-tile kernel) - same results bit for bit."""

    def __init__(self, model, flat_params, coeffs, grid, y0, dW=None, method='euler', seed=0, row_offset=0,
                 kernel='auto', save_traj=False, save_dW=False, exact_order=False, save_act=False, dU=None, row_out=None,
                 noise_table=None, z0_linear=None, kl_column=None, stream_all=False, two_tile=False):
        B, H = y0.shape
        C_ = model.input_channels
        L = coeffs.shape[1] + 1
        dev = y0.device
        _check_f32('y0', y0, (B, model.hidden_channels))
        _check_f32('coeffs', coeffs, (B, L - 1, 4 * C_))
        _check_f32('params', flat_params)
        if dW is not None:
            _check_f32('dW', dW, (grid.N, B, H))
        self.model, self.grid = model, grid
        if dU is not None:
            _check_f32('dU', dU, (grid.N, B, H))
        if row_out is not None:
            if row_out.dtype != torch.int32 or not row_out.is_cuda or not row_out.is_contiguous() or tuple(row_out.shape) != (B,):
                raise ValueError('row_out must be a contiguous int32 CUDA tensor of shape (batch,)')
        if noise_table is not None:      # the time-only diffusion factor per step (SRK: at the step's four stage times)
            _check_f32('noise_table', noise_table, (4 * grid.N if method == 'srk' else grid.N, H))
        if z0_linear is not None:     # (weight (H, C), bias (H)): y0 is then an output, filled by the solve's prepare launch
            _check_f32('z0 weight', z0_linear[0], (H, C_))
            _check_f32('z0 bias', z0_linear[1], (H,))
        self.keep = (flat_params, coeffs, y0, dW, grid, dU, row_out, noise_table, z0_linear)
        # per-row output selection: one state per row instead of one plane per output time
        self.ys = torch.empty((B, H) if row_out is not None else (grid.T, B, H), device=dev, dtype=torch.float32)
        self.traj = torch.empty((grid.N + 1, B, H), device=dev, dtype=torch.float32) if save_traj else None
        self.dW_out = torch.empty((grid.N, B, H), device=dev, dtype=torch.float32) if save_dW else None
        self.act_save = self.stage_save = None
        s = _lib.Solve()
        s.model = model
        s.batch, s.knots, s.n_steps, s.n_out = B, L, grid.N, grid.T
        s.method = {'euler': _lib.EULER, 'milstein': _lib.MILSTEIN, 'srk': _lib.SRK}[method]
        # (kernel selector, flags and the kind of Philox key BEFORE the layout query below: which adjoint a solve gets - and with it
        #  whether it needs delta planes at all - depends on them)
        s.kernel = _lib.KERNELS[kernel]
        self.base_flags = (_lib.FLAG_EXACT_ORDER if exact_order else 0) | (_lib.FLAG_STREAM_ALL if stream_all else 0) | (_lib.FLAG_TWO_TILE if two_tile else 0)
        s.flags = self.base_flags
        if torch.is_tensor(seed):     # device-resident key: re-read by every launch / graph replay
            if seed.dtype != torch.int64 or not seed.is_cuda or seed.numel() != 1:
                raise ValueError('a tensor seed must be a one-element int64 CUDA tensor')
            self.keep = self.keep + (seed,)
            s.seed_dev = _ptr(seed)
        else:
            s.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        # every configuration field the plan looks at goes in BEFORE the layout / workspace queries (which adjoint a solve gets
        # depends on the supplied increments, the noise table, the accumulator column and the per-row output selection)
        s.dW, s.row_out, s.noise_table = _ptr(dW), _ptr(row_out), _ptr(noise_table)
        s.row_offset = int(row_offset)
        if kl_column is not None:     # (column, a, b): path-integral accumulator column with the linear prior drift a y + b (snsde.h)
            s.kl_column1, s.kl_prior_a, s.kl_prior_b = int(kl_column[0]) + 1, float(kl_column[1]), float(kl_column[2])
        # host-side queries of the library (save layout, workspace sizes) depend on the configuration only: memoised
        self.cfg_key = (model.input_channels, model.hidden_channels, model.hidden_hidden_channels, model.num_hidden_layers,
                        model.input_option, model.noise_option, model.activation, model.drift_output, model.diffusion_output,
                        model.time_feature, B, L, grid.N, grid.T, method, kernel, bool(exact_order), noise_table is not None,
                        dW is not None, row_out is not None, None if kl_column is None else int(kl_column[0]), torch.is_tensor(seed))
        if save_act:
            lay = _SIZE_CACHE.get(('layout',) + self.cfg_key)
            if lay is None:
                slots, planes, dslots = C.c_int32(), C.c_int32(), C.c_int32()
                _lib.check(_lib.lib().snsde_save_layout(C.byref(s), C.byref(slots), C.byref(planes), C.byref(dslots)), 'snsde_save_layout')
                lay = _SIZE_CACHE[('layout',) + self.cfg_key] = (slots.value, planes.value, dslots.value)
            nslots, nplanes, self.delta_slots = lay
            passes = 3 * grid.N if method == 'srk' else grid.N      # SRK: three drift passes per step
            self.act_save = torch.empty((passes, nslots, B, H), device=dev, dtype=torch.float32)
            if method == 'srk':
                shape = (passes + 1, B, H) if nplanes == 1 else (passes + 1, nplanes, B, H)
                self.stage_save = torch.empty(shape, device=dev, dtype=torch.float32)
        if method == 'srk':
            s.srk_tab = _ptr(srk_table(grid))
            s.dU = _ptr(dU)
            self.dU_out = torch.empty((grid.N, B, H), device=dev, dtype=torch.float32) if save_dW else None
            s.dU_out = _ptr(self.dU_out)
        s.params, s.coeffs = _ptr(flat_params), _ptr(coeffs)
        s.step_tab, s.out_step, s.out_w = _ptr(grid.d_step_tab), _ptr(grid.d_out_step), _ptr(grid.d_out_w)
        s.y0, s.ys = _ptr(y0), _ptr(self.ys)
        s.traj, s.dW_out = _ptr(self.traj), _ptr(self.dW_out)
        s.act_save = _ptr(self.act_save)
        s.stage_save = _ptr(self.stage_save)
        if z0_linear is not None:
            s.z0_weight, s.z0_bias = _ptr(z0_linear[0]), _ptr(z0_linear[1])
        nbytes = _SIZE_CACHE.get(('fwd',) + self.cfg_key)
        if nbytes is None:
            nbytes = _SIZE_CACHE[('fwd',) + self.cfg_key] = int(_lib.lib().snsde_workspace_bytes(C.byref(s)))
        self.workspace = torch.empty(max(nbytes, 256), device=dev, dtype=torch.uint8)
        s.workspace = _ptr(self.workspace)
        s.workspace_bytes = self.workspace.numel()
        self.desc = s

    def _prepared_key(self):
        """What the prepared workspace (packed weights, folded products, time tables) was built from: address and version counter
        of the parameter block (+ the supplied noise table).  In-place writes to those tensors bump the counter; writes through an
        alias with its own counter (`tensor.data`, the per-parameter views of the arena) do not - callers that update parameters
        that way build a new SolveCall per solve (torchsde.sdeint does) or pass auto_reuse=False."""
        flat, table = self.keep[0], self.keep[7]
        return (flat.data_ptr(), flat._version, None if table is None else (table.data_ptr(), table._version))

    def launch(self, stream=None, reuse_prepared=False, auto_reuse=False):
        """Enqueue the solve.  reuse_prepared=True skips weight packing / time tables (legal while the parameter block and grid are
        unchanged since the previous launch of this call).  auto_reuse=True (opt-in: evaluation epochs over frozen parameters) lets
        the call set the flag itself from the second launch on while the parameter block's version counter has not moved - the
        counter does NOT see writes through an alias (`p.data`, the per-parameter views of the arena an optimizer steps on), so a
        call that lives across optimizer steps must not use it.  The prepare launch is ~8 us of a 200 us K2 solve."""
        stream = torch.cuda.current_stream(self.ys.device) if stream is None else stream
        capturing = torch.cuda.is_current_stream_capturing()
        key = self._prepared_key()
        if auto_reuse and not reuse_prepared and not capturing and getattr(self, '_prep_key', None) == key:
            reuse_prepared = True
        self.desc.flags = self.base_flags | (_lib.FLAG_REUSE_PREPARED if reuse_prepared else 0)
        _lib.check(_lib.lib().snsde_solve_forward(C.byref(self.desc), C.c_void_p(stream.cuda_stream)),
                   'snsde_solve_forward')
        if not capturing:             # a recorded prepare launch has not run: it prepared nothing an eager launch could reuse
            self._prep_key = key
        return self.ys


def backward_supported(call):
    """0 = no fused backward; 1 = MFMA adjoint kernel (needs save_act); 2 = generic adjoint kernel (forward must run
    on the generic kernel; needs traj + dW_out only)."""
    return int(_lib.lib().snsde_backward_supported(C.byref(call.desc)))


class _BoundedCache(dict):
    """Memo of host-side library queries per configuration; one entry per distinct (batch, knots, steps, outputs, ...) a process
    solves - dropped wholesale past `cap` entries (ragged last batches of many lengths must not grow it without bound)."""
    cap = 4096

    def __setitem__(self, k, v):
        if len(self) >= self.cap:
            self.clear()
        super().__setitem__(k, v)


_MODE_CACHE = _BoundedCache()
_SIZE_CACHE = _BoundedCache()      # host-side size queries of the library per configuration (SolveCall.cfg_key)


def backward_mode(model, batch, knots, grid, method, kernel='auto', exact_order=False, table=False, kl_column=None):
    """backward_supported for a solve that has not been allocated yet (memoised per configuration); table: the solve
    supplies a noise_table."""
    key = (table, model.input_channels, model.hidden_channels, model.hidden_hidden_channels, model.num_hidden_layers,
           model.input_option, model.noise_option, model.activation, model.drift_output, model.diffusion_output,
           model.time_feature, batch, knots, grid.N, grid.T, method, kernel, exact_order, kl_column)
    hit = _MODE_CACHE.get(key)
    if hit is None:
        s = _lib.Solve()
        s.model = model
        s.batch, s.knots, s.n_steps, s.n_out = batch, knots, grid.N, grid.T
        s.method = {'euler': _lib.EULER, 'milstein': _lib.MILSTEIN, 'srk': _lib.SRK}[method]
        s.kernel = _lib.KERNELS[kernel]
        s.flags = _lib.FLAG_EXACT_ORDER if exact_order else 0
        s.noise_table = C.c_void_p(16) if table else None      # (only its presence matters to the query)
        if kl_column is not None:
            s.kl_column1 = int(kl_column) + 1
        hit = int(_lib.lib().snsde_backward_supported(C.byref(s)))
        _MODE_CACHE[key] = hit
    return hit


def adj0_suffices(call):
    """The MFMA adjoint + native weight-gradient pass of this solve never read the intermediate adjoints from memory (all but
    Milstein through a diffusion net, whose second-order weight-gradient jobs do)."""
    return not (call.desc.method == _lib.MILSTEIN and call.model.noise_option in (14, 15, 18, 19))


def solve_backward(call, grad_ys, stream=None, save_delta=False, adj0_only=False):
    """Adjoint recursion over a finished training-mode solve (SolveCall with save_traj/save_dW/save_act):
    returns adj (N+1, B, H), adj[n] = dL/dy_n; adj[0] is the gradient w.r.t. y0.  adj0_only (MFMA adjoint kernels, mode 1,
    not Milstein through a diffusion net): adj is (1, B, H) - the intermediate adjoints stay on chip."""
    if call.traj is None:
        raise ValueError('backward needs a solve run with save_traj (and save_act on the MFMA path)')
    _check_f32('grad_ys', grad_ys, tuple(call.ys.shape))
    b = _lib.Backward()
    b.fwd = call.desc
    b.fwd.flags = call.base_flags
    adj = torch.empty_like(call.traj[:1]) if adj0_only else torch.empty_like(call.traj)
    b.flags = _lib.BWD_ADJ0_ONLY if adj0_only else 0
    delta = None
    if save_delta and call.act_save is not None and getattr(call, 'delta_slots', 1) != 0:      # (passes, delta slots, B, H): snsde_save_layout
        shp = call.act_save.shape                                                                # (0 slots: the adjoint sums the weight gradients itself)
        delta = torch.empty((shp[0], getattr(call, 'delta_slots', shp[1]), shp[2], shp[3]), device=call.act_save.device, dtype=torch.float32)
    b.grad_ys, b.adj, b.delta_save = _ptr(grad_ys), _ptr(adj), _ptr(delta)
    nbytes = _lib.lib().snsde_backward_workspace_bytes(C.byref(b))
    ws = torch.empty(max(nbytes, 256), device=adj.device, dtype=torch.uint8)
    b.workspace, b.workspace_bytes = _ptr(ws), ws.numel()
    stream = torch.cuda.current_stream(adj.device) if stream is None else stream
    _lib.check(_lib.lib().snsde_solve_backward(C.byref(b), C.c_void_p(stream.cuda_stream)), 'snsde_solve_backward')
    call.keep_bwd = (ws, grad_ys)
    call.bwd_desc = b
    return (adj, delta) if save_delta else adj


def param_gradients(call, adj, delta, stream=None, want_table_grad=False):
    """Flat parameter gradient (the C ABI's layout) of a finished MFMA-path solve + adjoint: snsde_param_gradients
    (split-R MFMA weight-gradient GEMMs, diffusion reductions, first-layer algebra), all on the device.
    want_table_grad (solves with a supplied noise_table): returns (grad, dL/d noise_table (N, H); SRK: (4 N, H))."""
    b = _lib.Backward()
    b.fwd = call.desc
    b.fwd.flags = call.base_flags
    b.adj, b.delta_save = _ptr(adj), _ptr(delta)
    tab_grad = None
    if want_table_grad:
        rows = call.grid.N * (4 if call.desc.method == _lib.SRK else 1)      # SRK: one row per stage time
        tab_grad = torch.zeros((rows, call.model.hidden_channels), device=adj.device, dtype=torch.float32)
        b.grad_noise_table = _ptr(tab_grad)
    bws = call.keep_bwd[0]          # the adjoint's workspace: holds its per-workgroup diffusion-side sums
    b.workspace, b.workspace_bytes = _ptr(bws), bws.numel()
    L = _lib.lib()
    nbytes = L.snsde_param_gradients_workspace_bytes(C.byref(b))
    if nbytes == 0:
        raise NotImplementedError('snsde_param_gradients covers the MFMA-path configurations only')
    ws = torch.empty(nbytes, device=adj.device, dtype=torch.uint8)
    grad = torch.empty(call.keep[0].numel(), device=adj.device, dtype=torch.float32)
    stream = torch.cuda.current_stream(adj.device) if stream is None else stream
    _lib.check(L.snsde_param_gradients(C.byref(b), _ptr(grad), _ptr(ws), ws.numel(), C.c_void_p(stream.cuda_stream)),
               'snsde_param_gradients')
    call.keep_pg = (ws, adj, delta, tab_grad)
    return (grad, tab_grad) if want_table_grad else grad


def backward_with_gradients(call, grad_ys, stream=None, adj0_only=True, want_table_grad=False):
    """solve_backward + param_gradients as ONE C call (snsde_backward_with_gradients; mode 1 solves): same results, one
    host -> library transition.  Returns (adj, flat gradient[, dL/d noise_table])."""
    if call.traj is None or call.act_save is None:
        raise ValueError('backward needs a solve run with save_traj and save_act')
    _check_f32('grad_ys', grad_ys, tuple(call.ys.shape))
    dev = call.traj.device
    b = _lib.Backward()
    b.fwd = call.desc
    b.fwd.flags = call.base_flags
    adj = torch.empty_like(call.traj[:1]) if adj0_only else torch.empty_like(call.traj)
    b.flags = _lib.BWD_ADJ0_ONLY if adj0_only else 0
    shp = call.act_save.shape
    delta = None          # (0 delta slots: the adjoint of this solve sums the weight gradients itself, include/snsde.h)
    if getattr(call, 'delta_slots', shp[1]) != 0:
        delta = torch.empty((shp[0], getattr(call, 'delta_slots', shp[1]), shp[2], shp[3]), device=dev, dtype=torch.float32)
    b.grad_ys, b.adj, b.delta_save = _ptr(grad_ys), _ptr(adj), _ptr(delta)
    L = _lib.lib()
    tab_grad = None
    if want_table_grad:
        rows = call.grid.N * (4 if call.desc.method == _lib.SRK else 1)      # SRK: one row per stage time
        tab_grad = torch.zeros((rows, call.model.hidden_channels), device=dev, dtype=torch.float32)
        b.grad_noise_table = _ptr(tab_grad)
    key = ('bwd', adj0_only, want_table_grad) + call.cfg_key
    sizes = _SIZE_CACHE.get(key)
    if sizes is None:
        b.workspace, b.workspace_bytes = C.c_void_p(16), 1 << 40      # (only its presence matters to the size queries)
        sizes = _SIZE_CACHE[key] = (max(int(L.snsde_backward_workspace_bytes(C.byref(b))), 256),
                                    int(L.snsde_param_gradients_workspace_bytes(C.byref(b))))
    if sizes[1] == 0:
        raise NotImplementedError('snsde_backward_with_gradients covers the MFMA-path configurations only')
    ws = torch.empty(sizes[0], device=dev, dtype=torch.uint8)
    b.workspace, b.workspace_bytes = _ptr(ws), ws.numel()
    pws = torch.empty(sizes[1], device=dev, dtype=torch.uint8)
    grad = torch.empty(call.keep[0].numel(), device=dev, dtype=torch.float32)
    stream = torch.cuda.current_stream(dev) if stream is None else stream
    _lib.check(L.snsde_backward_with_gradients(C.byref(b), _ptr(grad), _ptr(pws), pws.numel(), C.c_void_p(stream.cuda_stream)),
               'snsde_backward_with_gradients')
    call.keep_bwd = (ws, grad_ys)
    call.keep_pg = (pws, adj, delta, tab_grad)
    call.bwd_desc = b
    return (adj, grad, tab_grad) if want_table_grad else (adj, grad)


def eval_fg(model, flat_params, coeffs, times_host, t, y, kernel='auto'):
    """f(t, y), g(t, y) through the solver's device code (snsde_eval_fg)."""
    B, H = y.shape
    L = coeffs.shape[1] + 1
    dev = y.device
    t = float(np.float32(t))
    # one-row step table for time t (h is irrelevant here); nextafter keeps ts increasing
    ts = np.array([t, np.nextafter(np.float32(t), np.float32(np.inf)) + np.float32(1.0)], dtype=np.float32)
    g = StepGrid(ts, 2.0 + abs(t), times_host, dev)
    s = _lib.Solve()
    s.model = model
    s.batch, s.knots, s.n_steps, s.n_out = B, L, 1, 2
    s.kernel = _lib.KERNELS[kernel]
    s.params, s.coeffs = _ptr(flat_params), _ptr(coeffs)
    nbytes = _lib.lib().snsde_workspace_bytes(C.byref(s))
    ws = torch.empty(max(nbytes, 256), device=dev, dtype=torch.uint8)
    s.workspace, s.workspace_bytes = _ptr(ws), ws.numel()
    f = torch.empty_like(y)
    gg = torch.empty_like(y)
    stream = torch.cuda.current_stream(dev)
    _lib.check(_lib.lib().snsde_eval_fg(C.byref(s), _ptr(g.d_step_tab), _ptr(y), _ptr(f), _ptr(gg),
                                        C.c_void_p(stream.cuda_stream)), 'snsde_eval_fg')
    return f, gg


def head_layers(seq):
    """(input_tanh, linear1, batchnorm or None, linear2) when `seq` is one of the wrappers' readout heads in a state the
    fused head reproduces - [Tanh,] Linear, [BatchNorm1d on running statistics,] ReLU, [Dropout in eval mode,] Linear - else
    None (benchmark_classification/models_sde/neuralsde.py:59-61; benchmark_forecasting/...:153-155; torch_ists nsde_model)."""
    mods = list(seq) if isinstance(seq, torch.nn.Sequential) else None
    if not mods:
        return None
    tanh = isinstance(mods[0], torch.nn.Tanh)
    if tanh:
        mods = mods[1:]
    if len(mods) < 3 or not isinstance(mods[0], torch.nn.Linear) or not isinstance(mods[-1], torch.nn.Linear):
        return None
    lin1, lin2, mid = mods[0], mods[-1], mods[1:-1]
    bn = None
    if isinstance(mid[0], torch.nn.BatchNorm1d):
        bn, mid = mid[0], mid[1:]
        if bn.training or bn.running_mean is None:
            return None
    if not mid or not isinstance(mid[0], torch.nn.ReLU):
        return None
    for m in mid[1:]:
        if not (isinstance(m, torch.nn.Dropout) and (not m.training or m.p == 0.0)):
            return None
    if lin2.in_features != lin1.out_features or (bn is not None and bn.num_features != lin1.out_features):
        return None
    if lin1.out_features > 128 or lin1.in_features > 256:
        # wider heads: the library GEMM chain is faster (measured 27 vs 42 us at 1280 x 256 x 256 x 14; the kernel itself
        # takes hidden sizes up to 512)
        return None
    return tanh, lin1, bn, lin2


def readout_head(x, layers, stream=None):
    """out = linear2(relu(bn(linear1(act(x))))) over the last dimension of x in ONE launch (snsde_readout_head); inference
    only - the caller checks torch.is_grad_enabled().  `layers` = (input_tanh, linear1, batchnorm or None, linear2)."""
    tanh, lin1, bn, lin2 = layers
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    _check_f32('head input', x2, (x2.shape[0], lin1.in_features))
    out = torch.empty((x2.shape[0], lin2.out_features), device=x.device, dtype=torch.float32)
    h = _lib.Head()
    h.rows, h.in_features, h.hidden, h.out_features = x2.shape[0], lin1.in_features, lin1.out_features, lin2.out_features
    h.input_tanh = 1 if tanh else 0
    keep = [x2, lin1.weight.detach().contiguous(), lin2.weight.detach().contiguous()]
    h.x, h.w1, h.w2, h.out = _ptr(x2), _ptr(keep[1]), _ptr(keep[2]), _ptr(out)
    h.b1 = _ptr(None if lin1.bias is None else lin1.bias.detach())
    h.b2 = _ptr(None if lin2.bias is None else lin2.bias.detach())
    if bn is not None:
        h.bn_eps = float(bn.eps)
        h.bn_mean, h.bn_var = _ptr(bn.running_mean), _ptr(bn.running_var)
        h.bn_weight = _ptr(None if bn.weight is None else bn.weight.detach())
        h.bn_bias = _ptr(None if bn.bias is None else bn.bias.detach())
    stream = torch.cuda.current_stream(x.device) if stream is None else stream
    _lib.check(_lib.lib().snsde_readout_head(C.byref(h), C.c_void_p(stream.cuda_stream)), 'snsde_readout_head')
    return out.reshape(*lead, lin2.out_features)


def spline_coeffs(times, X, kind='natural'):
    """Packed spline coefficients (B, L-1, 4C) on the GPU; times (L,), X (B, L, C) CUDA float32 with NaN = missing."""
    _check_f32('X', X)
    _check_f32('times', times)
    B, L, Cn = X.shape
    out = torch.empty((B, L - 1, 4 * Cn), device=X.device, dtype=torch.float32)
    stream = C.c_void_p(torch.cuda.current_stream(X.device).cuda_stream)
    lib = _lib.lib()
    if kind == 'natural':
        ws = torch.empty(lib.snsde_spline_workspace_bytes(B, L, Cn), device=X.device, dtype=torch.uint8)
        _lib.check(lib.snsde_natural_cubic_coeffs(_ptr(times), _ptr(X), B, L, Cn, _ptr(out), _ptr(ws), ws.numel(), stream),
                   'snsde_natural_cubic_coeffs')
        torch.cuda.current_stream(X.device).synchronize()    # workspace is released on return
    else:
        _lib.check(lib.snsde_hermite_coeffs(_ptr(times), _ptr(X), B, L, Cn, _ptr(out), stream), 'snsde_hermite_coeffs')
    return out


def spline_evaluate(coeffs, index, frac, derivative=False):
    B, Lm1, C4 = coeffs.shape
    _check_f32('coeffs', coeffs)
    out = torch.empty((B, C4 // 4), device=coeffs.device, dtype=torch.float32)
    stream = torch.cuda.current_stream(coeffs.device)
    _lib.check(_lib.lib().snsde_spline_evaluate(_ptr(coeffs), B, Lm1 + 1, C4 // 4, int(index), float(frac),
                                                int(bool(derivative)), _ptr(out), C.c_void_p(stream.cuda_stream)),
               'snsde_spline_evaluate')
    return out
