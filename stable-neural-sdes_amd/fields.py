"""Fused path for the tutorial-style vector fields (reference tutorial/*.ipynb, cell 7: NeuralLSDEFunc / NeuralLNSDEFunc /
NeuralGSDEFunc).  They are not Diffusion_models (different parameter names, `nn.Sequential(Linear, act, ...)` MLPs with
LipSwish, raw time, no tanh squashing) but their drift is the same chain of affine maps and pointwise activations, and
their diffusion is a function of time alone times 1 or y:

    f(t, y) = linear_out(f_net(emb([linear_in([t, y]) | y,  linear_X(X(t))])))   [* y for the GSDE field]
    g(t, y) = g_net(noise_in(feature(t)))                                       [* y for the LNSDE / GSDE fields]

`compose()` maps such a module onto the C ABI's Diffusion_model-shaped parameter block by composing adjacent affine maps
(emb' = f_net[0] o emb, linear_out' = linear_out o f_net[-1], linear_in' = [w_t, 0, W_y] or the identity) and sets the
variant switches of `snsde_model` (include/snsde.h: SNSDE_ACT_* / SNSDE_DRIFT_* / SNSDE_DIFFUSION_RAW / SNSDE_TIME_RAW);
the time-only diffusion factor is evaluated ONCE per solve for every step time through the module's own g (one batched
call over N times instead of N calls) and handed to the kernel as `snsde_solve.noise_table`.  Training keeps both of them
in the autograd graph around the fused forward + adjoint (torchsde._ComposedSolve).  The first solve of a module
checks the mapping against the module's own f / g through the kernel (a one-step probe); a module that fails the
structural match or the probe is simply not recognised and takes the generic graph-captured stepper
(torchsde._graphed_steps).
"""
import numpy as np
import torch

from . import _lib, engine

ACT_RELU, ACT_LIPSWISH, ACT_SILU = 0, 1, 2
DRIFT_TANH, DRIFT_LINEAR, DRIFT_TIMES_Y = 0, 1, 2
DIFFUSION_TANH, DIFFUSION_RAW, DIFFUSION_RAW_NET = 0, 1, 2
TIME_SINCOS, TIME_RAW = 0, 1

_PROBE = torch.tensor([-3.0, -1.0, -0.25, 0.0, 0.5, 1.0, 2.5])


def _classify_activation(mod):
    """SNSDE_ACT_* of a pointwise activation module, by its values on a probe vector (so a user-defined LipSwish class
    is matched by what it computes, not by its name); None when it is none of the three the kernel implements."""
    try:
        with torch.no_grad():
            v = mod(_PROBE.clone())
    except Exception:
        return None
    if not torch.is_tensor(v) or v.shape != _PROBE.shape:
        return None
    silu = torch.nn.functional.silu(_PROBE)
    for kind, ref in ((ACT_RELU, _PROBE.clamp_min(0)), (ACT_SILU, silu), (ACT_LIPSWISH, 0.909 * silu)):
        if torch.allclose(v, ref, rtol=1e-6, atol=1e-7):
            return kind
    return None


def _mlp_chain(net):
    """[Linear, ..., Linear] and the activation kind of an MLP written as nn.Sequential(Linear, act, ..., Linear)
    (directly, or held in an attribute `_model` as the tutorial's MLP class does); None when `net` is anything else."""
    seq = getattr(net, '_model', net)
    if not isinstance(seq, torch.nn.Sequential):
        return None
    mods = list(seq)
    if len(mods) < 3 or len(mods) % 2 == 0:
        return None
    linears, kinds = [], set()
    for i, m in enumerate(mods):
        if i % 2 == 0:
            if not isinstance(m, torch.nn.Linear) or m.bias is None:
                return None
            linears.append(m)
        else:
            kinds.add(_classify_activation(m))
    if len(kinds) != 1 or None in kinds:
        return None
    return linears, kinds.pop()


def _is_linear(m, n_in, n_out):
    return isinstance(m, torch.nn.Linear) and m.bias is not None and m.in_features == n_in and m.out_features == n_out


def _capturing(dev):
    return torch.device(dev).type == 'cuda' and torch.cuda.is_current_stream_capturing()


class _ComposeAffine(torch.autograd.Function):
    """The composed parameter block from the field's own Linear layers, natively: snsde_affine_compose (one launch) and, backward,
    snsde_affine_compose_backward (one launch) - in place of ~25 + ~35 small torch launches per training step (matmuls, cats, zeros
    and their autograd nodes), which is what bounded the tutorial fields' eager training step.
    spec: list of (name in the block, outer index or None, inner index, zero_col) over `tensors` = (w0, b0, w1, b1, ...) per layer."""

    @staticmethod
    def forward(ctx, spec, layout, numel, *tensors):
        dev = tensors[0].device
        where = {name: (off, shape) for name, off, shape in layout}
        src = [t.detach().contiguous() for t in tensors]
        sizes = [t.numel() for t in src]
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + n)
        eyes = [name for name, outer, inner, zero_col in spec if inner is None]      # identity layers: written after the launch
        spec = [e for e in spec if e[2] is not None]
        jobs = (_lib.AffineJob * len(spec))()
        for q, (name, outer, inner, zero_col) in zip(jobs, spec):
            wi, bi = src[2 * inner], src[2 * inner + 1]
            q.w_inner, q.b_inner = wi.data_ptr(), bi.data_ptr()
            q.g_w_inner, q.g_b_inner = offs[2 * inner], offs[2 * inner + 1]
            q.zero_col = zero_col
            if outer is None:
                q.w_outer = q.b_outer = None
                q.R, q.K, q.Cin = wi.shape[0], 0, wi.shape[1]
                q.g_w_outer = q.g_b_outer = -1
            else:
                wo, bo = src[2 * outer], src[2 * outer + 1]
                q.w_outer, q.b_outer = wo.data_ptr(), bo.data_ptr()
                q.R, q.K, q.Cin = wo.shape[0], wo.shape[1], wi.shape[1]
                assert wi.shape[0] == q.K
                q.g_w_outer, q.g_b_outer = offs[2 * outer], offs[2 * outer + 1]
            q.dst_w, q.dst_b = where[name + '.weight'][0], where[name + '.bias'][0]
            assert tuple(where[name + '.weight'][1]) == (q.R, q.Cin + (1 if zero_col >= 0 else 0)), (name, where[name + '.weight'][1])
        flat = torch.zeros(numel, device=dev, dtype=torch.float32)      # (theta, unused noise_t entries: zero)
        stream = torch.cuda.current_stream(dev)
        _lib.check(_lib.lib().snsde_affine_compose(jobs, len(spec), flat.data_ptr(), stream.cuda_stream), 'snsde_affine_compose')
        for name in eyes:      # (a constant of the block: no gradient flows to it)
            off, shape = where[name + '.weight']
            flat[off:off + shape[0] * shape[1]].view(shape).diagonal().fill_(1.0)
        ctx.jobs, ctx.src, ctx.offs, ctx.shapes = jobs, src, offs, [tuple(t.shape) for t in tensors]
        return flat

    @staticmethod
    def backward(ctx, gflat):
        gflat = gflat.contiguous()
        gsrc = torch.zeros(ctx.offs[-1], device=gflat.device, dtype=torch.float32)
        stream = torch.cuda.current_stream(gflat.device)
        _lib.check(_lib.lib().snsde_affine_compose_backward(ctx.jobs, len(ctx.jobs), gflat.data_ptr(), gsrc.data_ptr(), stream.cuda_stream),
                   'snsde_affine_compose_backward')
        grads = tuple(gsrc[ctx.offs[i]:ctx.offs[i + 1]].view(shape) for i, shape in enumerate(ctx.shapes))
        return (None, None, None) + grads


def _native_block(layers, spec, layout, numel, dev):
    """layers: the field's nn.Linear modules in the order `spec` indexes them; None when the native route does not apply (CPU,
    non-float32 parameters, more jobs than the C ABI takes)."""
    dev = torch.device(dev)
    if dev.type != 'cuda' or len(spec) > _lib.MAX_AFFINE_JOBS:
        return None
    tensors = []
    for lin in layers:
        w = lin.weight
        if w.dtype != torch.float32 or not w.is_cuda or (dev.index is not None and w.device.index != dev.index) or lin.bias is None:
            return None
        tensors += [lin.weight, lin.bias]
    return _ComposeAffine.apply(spec, layout, numel, *tensors)


class ComposedField:
    """A tutorial-style field mapped onto the fused step: `model` (snsde_model with the variant switches), `flat(dev)`
    (the composed parameter block, rebuilt from the module's current weights on every solve) and `noise_table(grid)`."""

    def __init__(self, sde, model, layout, numel, parts, additive):
        self.sde, self.model, self.layout, self.numel = sde, model, layout, numel
        self.parts, self.additive = parts, additive
        # tabulated: the diffusion is (a function of time alone) x {1, y}, handed to the kernel as a per-step table; False: the
        # NeuralSDEFunc shape, whose diffusion is an MLP of [t, y] evaluated by the net kernels from the composed block
        self.tabulated = 'noise_first' not in parts and not parts.get('ode', False)      # (the ODE field: no diffusion at all)
        self.verified = {}        # device -> bool (one-step probe through the kernel)
        self.trust_versions = False
        self._flat_cache = None
        self._tab_cache = None
        self._proj = None

    def _param_key(self, dev):
        """Identity of the module's current parameter VALUES: addresses and version counters (re-assigned parameters,
        optimizer steps, copy_ / load_state_dict) plus a fingerprint of the contents - a fixed random projection of all
        parameters, two small launches and one host read - because in-place updates through `param.data` (p.data.add_(..),
        clamp_ ...) leave the version counter untouched."""
        params = list(self.sde.parameters())
        if self.trust_versions:      # caller's promise (options={'trust_versions': True}): parameters change through autograd /
            return (str(dev),) + tuple((q.data_ptr(), q._version) for q in params)      # optimizers only - no host read-back
        vec = torch.cat([q.detach().reshape(-1).to(device=dev, dtype=torch.float32) for q in params])
        if self._proj is None or self._proj.numel() != vec.numel() or self._proj.device != vec.device:
            gen = torch.Generator().manual_seed(0x5DE)
            self._proj = torch.randn(vec.numel(), generator=gen).to(vec.device)
        return (str(dev), float(torch.dot(vec, self._proj))) + tuple((q.data_ptr(), q._version) for q in params)

    def flat(self, dev, grad=False):
        """The composed parameter block (float32, the C ABI's layout).  grad=True keeps the autograd graph from the module's
        parameters to the block (training: the fused backward's flat gradient flows back through the composition)."""
        if grad:
            with torch.enable_grad():
                return self._flat(dev, True)
        if _capturing(dev):       # graph capture: no host read - the composition is recorded
            with torch.no_grad():
                return self._flat(dev, False)
        # inference: the block only changes when a parameter does, so repeated solves reuse it (key: see _param_key)
        key = self._param_key(dev)
        hit = self._flat_cache
        if hit is None or hit[0] != key:
            with torch.no_grad():
                hit = (key, self._flat(dev, False))
            self._flat_cache = hit
        return hit[1]

    def inference_inputs(self, t0s, dev):
        """(parameter block, diffusion table) of a no-grad solve, cached on ONE reading of the parameters' identity."""
        if _capturing(dev):
            return self.flat(dev), self.noise_table(t0s, dev)
        key = self._param_key(dev)
        hit = self._flat_cache
        if hit is None or hit[0] != key:
            with torch.no_grad():
                hit = (key, self._flat(dev, False))
            self._flat_cache = hit
        return hit[1], (self.noise_table(t0s, dev, param_key=key) if self.tabulated else None)

    def _flat_net(self, dev, grad):
        """NeuralSDEFunc: f = f_net(linear_in([t, y])), g = g_net(noise_in([t, y])) as the block of (input_option 3,
        noise_option 18): linear_in' = f_net[0] o linear_in on [t, 0, y], hidden `linears`, linear_out' = f_net[-1];
        noise_y.0' = g_net[0] o noise_in, noise_y.2 = g_net[1]."""
        p = self.parts
        H, C_ = self.model.hidden_channels, self.model.input_channels
        if grad:      # training: the block and its backward natively, one launch each (snsde_affine_compose)
            layers = [p['mlp'][0], p['linear_in']] + list(p['mlp'][1:])
            spec = [('linear_in', 0, 1, 1)] + [(f'linears.{i}', None, 2 + i, -1) for i in range(len(p['mlp']) - 2)]
            spec.append(('linear_out', None, len(layers) - 1, -1))
            if 'noise_first' in p:
                n0 = len(layers)
                layers += [p['noise_first'], p['noise_in'], p['noise_last']]
                spec += [('noise_y.0', n0, n0 + 1, 1), ('noise_y.2', None, n0 + 2, -1)]
            out = _native_block(layers, spec, self.layout, self.numel, dev)
            if out is not None:
                return out
        f64 = dict(device=dev, dtype=torch.float32 if grad else torch.float64)
        W = lambda lin: (lin.weight if grad else lin.weight.detach()).to(**f64)
        b = lambda lin: (lin.bias if grad else lin.bias.detach()).to(**f64)

        def first(outer, inner):      # outer o inner, inner on [t, y] -> columns [t, 0, y]
            w = W(outer) @ W(inner)
            return torch.cat([w[:, :1], torch.zeros(w.shape[0], 1, **f64), w[:, 1:]], dim=1), W(outer) @ b(inner) + b(outer)
        vals = {'theta': torch.zeros(1, 1, **f64),
                'initial_network.weight': torch.zeros(H, C_, **f64), 'initial_network.bias': torch.zeros(H, **f64)}
        vals['linear_in.weight'], vals['linear_in.bias'] = first(p['mlp'][0], p['linear_in'])
        for i, lin in enumerate(p['mlp'][1:-1]):
            vals[f'linears.{i}.weight'], vals[f'linears.{i}.bias'] = W(lin), b(lin)
        vals['linear_out.weight'], vals['linear_out.bias'] = W(p['mlp'][-1]), b(p['mlp'][-1])
        if 'noise_first' in p:
            vals['noise_y.0.weight'], vals['noise_y.0.bias'] = first(p['noise_first'], p['noise_in'])
            vals['noise_y.2.weight'], vals['noise_y.2.bias'] = W(p['noise_last']), b(p['noise_last'])
        pieces = []
        for name, off, shape in self.layout:
            v = vals[name]
            assert tuple(v.shape) == tuple(shape), (name, tuple(v.shape), shape)
            pieces.append(v.reshape(-1).to(torch.float32))
        out = torch.cat(pieces)
        assert out.numel() == self.numel
        return out

    def _flat_latent(self, dev, grad):
        """LatentSDE-shaped posterior drift f = linear_out(relu(.. relu(linear_in([sin t, cos t, y])))) on the latent channels,
        zero-padded to the instantiated width P, as the block of (input_option 4, noise_option 12) with an identity embedding:
        emb = [I | 0] makes z = linear_in([tau, y]) (the folded first layer multiplies it out exactly), the control path is a
        dummy channel with zero weights."""
        p = self.parts
        P, Hl = self.model.hidden_channels, p['latent']
        f64 = dict(device=dev, dtype=torch.float32)
        W = lambda lin: (lin.weight if grad else lin.weight.detach()).to(**f64)
        b = lambda lin: (lin.bias if grad else lin.bias.detach()).to(**f64)
        pad2 = lambda w, rows, cols: torch.nn.functional.pad(w, (0, cols - w.shape[1], 0, rows - w.shape[0]))
        pad1 = lambda v, n: torch.nn.functional.pad(v, (0, n - v.shape[0]))
        w_in = W(p['linear_in'])                           # (HH, 2 + Hl): [sin t, cos t, y]
        vals = {'theta': torch.zeros(1, 1, **f64),
                'initial_network.weight': torch.zeros(P, self.model.input_channels, **f64),
                'initial_network.bias': torch.zeros(P, **f64),
                'linear_in.weight': pad2(w_in, P, P + 2), 'linear_in.bias': pad1(b(p['linear_in']), P),
                'emb.weight': torch.cat([torch.eye(P, **f64), torch.zeros(P, P, **f64)], dim=1), 'emb.bias': torch.zeros(P, **f64),
                'linear_out.weight': pad2(W(p['linear_out']), P, P), 'linear_out.bias': pad1(b(p['linear_out']), P),
                'noise_t.weight': torch.zeros(P, 2, **f64), 'noise_t.bias': torch.zeros(P, **f64)}
        for i, lin in enumerate(p['linears']):
            vals[f'linears.{i}.weight'], vals[f'linears.{i}.bias'] = pad2(W(lin), P, P), pad1(b(lin), P)
        pieces = []
        for name, off, shape in self.layout:
            v = vals[name]
            assert tuple(v.shape) == tuple(shape), (name, tuple(v.shape), shape)
            pieces.append(v.reshape(-1))
        out = torch.cat(pieces)
        assert out.numel() == self.numel
        return out

    def _flat(self, dev, grad):
        if 'latent' in self.parts:
            return self._flat_latent(dev, grad)
        if not self.tabulated:
            return self._flat_net(dev, grad)
        p = self.parts
        H = self.model.hidden_channels
        if grad:      # training: natively, one launch each way (LSDE: emb sees y itself - linear_in is the identity, a constant)
            first, last = p['mlp'][0], p['mlp'][-1]
            layers = [p['linear_X'], first, p['emb'], p['linear_out'], last] + list(p['mlp'][1:-1])
            spec = [('initial_network', None, 0, -1), ('emb', 1, 2, -1), ('linear_out', 3, 4, -1)]
            spec += [(f'linears.{i}', None, 5 + i, -1) for i in range(len(p['mlp']) - 2)]
            if p['linear_in'] is None:
                spec.append(('linear_in', None, None, -1))
            else:
                layers.append(p['linear_in'])
                spec.append(('linear_in', None, len(layers) - 1, 1))
            out = _native_block(layers, spec, self.layout, self.numel, dev)
            if out is not None:
                return out
        # float64 products for the (cached) inference block; training composes in float32: a third of the launches, and the
        # products' rounding is that of the kernels' own arithmetic
        f64 = dict(device=dev, dtype=torch.float32 if grad else torch.float64)
        W = lambda lin: (lin.weight if grad else lin.weight.detach()).to(**f64)
        b = lambda lin: (lin.bias if grad else lin.bias.detach()).to(**f64)
        first, last = p['mlp'][0], p['mlp'][-1]
        vals = {
            'initial_network.weight': W(p['linear_X']), 'initial_network.bias': b(p['linear_X']),
            'emb.weight': W(first) @ W(p['emb']), 'emb.bias': W(first) @ b(p['emb']) + b(first),
            'linear_out.weight': W(p['linear_out']) @ W(last), 'linear_out.bias': W(p['linear_out']) @ b(last) + b(p['linear_out']),
            'theta': torch.zeros(1, 1, **f64),
            'noise_t.weight': torch.zeros(H, 2, **f64), 'noise_t.bias': torch.zeros(H, **f64),
        }
        if p['linear_in'] is None:      # LSDE field: emb sees y itself
            vals['linear_in.weight'] = torch.eye(H, **f64)
            vals['linear_in.bias'] = torch.zeros(H, **f64)
        else:                           # [t, y] -> the C ABI's [time feature 0, time feature 1, y] columns
            w = W(p['linear_in'])
            vals['linear_in.weight'] = torch.cat([w[:, :1], torch.zeros(H, 1, **f64), w[:, 1:]], dim=1)
            vals['linear_in.bias'] = b(p['linear_in'])
        for i, lin in enumerate(p['mlp'][1:-1]):
            vals[f'linears.{i}.weight'] = W(lin)
            vals[f'linears.{i}.bias'] = b(lin)
        pieces = []
        for name, off, shape in self.layout:          # (the layout is contiguous in this order)
            v = vals[name]
            assert tuple(v.shape) == tuple(shape), (name, tuple(v.shape), shape)
            pieces.append(v.reshape(-1).to(torch.float32))
        out = torch.cat(pieces)
        assert out.numel() == self.numel
        return out

    def noise_table(self, t0s, dev, grad=False, param_key=None):
        """(N, H) float32: the time-only diffusion factor at every step time, through the module's own g (one batched call);
        grad=True keeps its autograd graph.  param_key: the parameter identity a caller already took for this solve
        (`inference_inputs`: one fingerprint read-back per solve, not two)."""
        N, H = t0s.shape[0], self.model.hidden_channels
        key = None
        if not grad and t0s.is_cuda and not _capturing(dev):
            # (same step times tensor of a cached grid + unchanged parameters: reuse the table)
            key = (t0s.data_ptr(), N) + (self._param_key(dev) if param_key is None else param_key)
            if self._tab_cache is not None and self._tab_cache[0] == key:
                return self._tab_cache[1]
        with torch.set_grad_enabled(grad):
            ones = torch.ones(N, H, device=dev, dtype=torch.float32)
            tab = self.sde.g(t0s.to(device=dev, dtype=torch.float32).reshape(N, 1), ones)
            if tuple(tab.shape) != (N, H):
                raise ValueError('g(t, y) over a column of times did not return (N, H)')
            tab = tab.to(torch.float32).contiguous()
        if key is not None:
            self._tab_cache = (key, tab, t0s)      # (holds t0s: its address stays valid while cached)
        return tab


def compose(sde):
    """ComposedField of a tutorial-style module, or None.  Cached on the module (its structure is fixed after
    construction; the composed weights are rebuilt per solve)."""
    slot = _slot(sde)
    if slot is not None and slot.composed is not None:
        return slot.composed[0]
    result = _compose(sde)
    if not torch.is_tensor(getattr(sde, 'coeffs', None)):
        return result          # (before set_X: the control path's channel count is not known yet - do not memoise)
    if slot is not None:
        slot.composed = (result,)
    return result


class _CacheSlot:
    """What this package memoises on a user's module (the structural mapping, its probe results, device tensors, ctypes
    structs).  It lives in the module's __dict__ but does not travel: copy.deepcopy / pickle / torch.save of the module carry an
    EMPTY slot, so a copy is recognised afresh and no CUDA tensor or ctypes object rides along."""

    def __init__(self):
        self.composed = None      # (ComposedField or None,)
        self.latent = None        # (key, ComposedField or None)

    def __deepcopy__(self, memo):
        return _CacheSlot()

    def __reduce__(self):
        return (_CacheSlot, ())


def _slot(sde):
    try:
        d = sde.__dict__
    except AttributeError:
        return None
    s = d.get('_snsde_cache')
    if not isinstance(s, _CacheSlot):
        s = d['_snsde_cache'] = _CacheSlot()
    return s


def _compose(sde):
    if not isinstance(sde, torch.nn.Module):
        return None
    if getattr(sde, 'sde_type', None) != 'ito':
        return None
    if getattr(sde, 'noise_type', None) == 'scalar':
        return _compose_ode(sde)
    if getattr(sde, 'noise_type', None) != 'diagonal':
        return None
    if hasattr(sde, 'input_option'):
        return None
    if all(hasattr(sde, a) for a in ('linear_in', 'f_net', 'noise_in', 'g_net', 'f', 'g')) and not hasattr(sde, 'emb'):
        return _compose_net(sde)
    need = ('linear_X', 'emb', 'f_net', 'linear_out', 'noise_in', 'g_net', 'f', 'g')
    if not all(hasattr(sde, a) for a in need):
        return None
    lin_X, emb, lin_out = sde.linear_X, sde.emb, sde.linear_out
    if not isinstance(lin_X, torch.nn.Linear) or lin_X.bias is None:
        return None
    C_, H = lin_X.in_features, lin_X.out_features
    chain = _mlp_chain(sde.f_net)
    if chain is None or not _is_linear(emb, 2 * H, H) or not _is_linear(lin_out, H, H):
        return None
    mlp, act = chain
    HH = mlp[0].out_features
    # the C ABI's block has emb (H, 2H) feeding HH-wide hidden layers: the reference's own constraint H == HH
    if HH != H or mlp[0].in_features != H or mlp[-1].out_features != H or mlp[-1].in_features != HH:
        return None
    if any(m.in_features != HH or m.out_features != HH for m in mlp[1:-1]):
        return None
    lin_in = getattr(sde, 'linear_in', None)
    if lin_in is not None and not _is_linear(lin_in, H + 1, H):
        return None
    io = 2 if lin_in is None else 4
    # diffusion: additive or multiplicative in y, decided by what g computes
    try:
        with torch.no_grad():
            p0 = next(sde.parameters())
            t = torch.tensor(0.37, device=p0.device, dtype=p0.dtype)
            gen = torch.Generator().manual_seed(7)
            # mixed signs: g = s(t) relu(y) / s(t) |y| must not be taken for the multiplicative form s(t) y
            y1 = (torch.rand(3, H, generator=gen) + 0.5).to(device=p0.device, dtype=p0.dtype)
            y2 = -(torch.rand(3, H, generator=gen) + 0.5).to(device=p0.device, dtype=p0.dtype)
            g1, g2 = sde.g(t, y1), sde.g(t, y2)
    except Exception:
        return None
    if tuple(g1.shape) != (3, H):
        return None
    if torch.allclose(g1, g2, rtol=1e-5, atol=1e-7):
        additive = True
    elif torch.allclose(g1 / y1, g2 / y2, rtol=1e-4, atol=1e-6):
        additive = False
    else:
        return None
    model = engine.model_struct(C_, H, HH, len(mlp) - 1, io, 12 if additive else 13, activation=act,
                                drift_output=DRIFT_LINEAR, diffusion_output=DIFFUSION_RAW,
                                time_feature=TIME_RAW if io == 4 else TIME_SINCOS)
    try:
        layout, numel = _lib.param_layout(model)
    except _lib.SnsdeError:
        return None
    parts = dict(linear_X=lin_X, emb=emb, linear_out=lin_out, linear_in=lin_in, mlp=mlp)
    field = ComposedField(sde, model, layout, numel, parts, additive)
    # f = z or z * y: decided by the probe in verify() (two candidates, the kernel's result must match one)
    return field


def _compose_ode(sde):
    """tutorial/simple OU process - Neural ODE.ipynb, NeuralODEFunc: f = f_net(linear_in([t, y])), noise_type 'scalar' with
    g = zeros (B, H, 1) - an ODE solved through sdeint.  Maps onto (input_option 3, noise_option 0) with the variant switches."""
    if hasattr(sde, 'input_option') or hasattr(sde, 'emb') or hasattr(sde, 'g_net'):
        return None
    lin_in = getattr(sde, 'linear_in', None)
    if not isinstance(lin_in, torch.nn.Linear) or lin_in.bias is None or not callable(getattr(sde, 'f', None)) \
            or not callable(getattr(sde, 'g', None)):
        return None
    H = lin_in.out_features
    fc = _mlp_chain(getattr(sde, 'f_net', None))
    if not _is_linear(lin_in, H + 1, H) or fc is None or len(fc[0]) > 4:
        return None
    mlp, act = fc
    if any(m.in_features != H or m.out_features != H for m in mlp):
        return None
    try:      # the diffusion must be identically zero, in torchsde's scalar-noise shape (B, H, 1)
        with torch.no_grad():
            p0 = lin_in.weight
            gv = sde.g(torch.tensor(0.3).to(p0), torch.ones(3, H).to(p0))
    except Exception:
        return None
    if tuple(gv.shape) != (3, H, 1) or float(gv.abs().max()) != 0.0:
        return None
    coeffs = getattr(sde, 'coeffs', None)
    C_ = int(coeffs.shape[-1]) // 4 if torch.is_tensor(coeffs) and coeffs.dim() == 3 else 1
    model = engine.model_struct(C_, H, H, len(mlp) - 1, 3, 0, activation=act, drift_output=DRIFT_LINEAR,
                                diffusion_output=DIFFUSION_RAW, time_feature=TIME_RAW)
    try:
        layout, numel = _lib.param_layout(model)
    except _lib.SnsdeError:
        return None
    return ComposedField(sde, model, layout, numel, dict(linear_in=lin_in, mlp=mlp, ode=True), additive=True)


def _compose_net(sde):
    """tutorial/simple OU process - Neural SDE.ipynb, NeuralSDEFunc: drift AND diffusion are MLPs of [t, y] (no control path
    inside the field).  Maps onto (input_option 3, noise_option 18) with the variant switches; the kernels evaluate two-layer
    nets, so g_net must be Linear, act, Linear (the notebook's num_layers = 1) and every width the hidden size."""
    lin_in, noise_in = sde.linear_in, sde.noise_in
    if not isinstance(lin_in, torch.nn.Linear) or lin_in.bias is None:
        return None
    H = lin_in.out_features
    if not _is_linear(lin_in, H + 1, H) or not _is_linear(noise_in, H + 1, H):
        return None
    fc, gc = _mlp_chain(sde.f_net), _mlp_chain(sde.g_net)
    if fc is None or gc is None or fc[1] != gc[1] or len(gc[0]) != 2 or len(fc[0]) > 4:
        return None
    mlp, act = fc
    if any(m.in_features != H or m.out_features != H for m in mlp + gc[0]):
        return None
    coeffs = getattr(sde, 'coeffs', None)
    C_ = int(coeffs.shape[-1]) // 4 if torch.is_tensor(coeffs) and coeffs.dim() == 3 else 1
    model = engine.model_struct(C_, H, H, len(mlp) - 1, 3, 18, activation=act, drift_output=DRIFT_LINEAR,
                                diffusion_output=DIFFUSION_RAW_NET, time_feature=TIME_RAW)
    try:
        layout, numel = _lib.param_layout(model)
    except _lib.SnsdeError:
        return None
    parts = dict(linear_in=lin_in, mlp=mlp, noise_in=noise_in, noise_first=gc[0][0], noise_last=gc[0][1])
    return ComposedField(sde, model, layout, numel, parts, additive=False)


class _LatentView:
    """The latent channels of a LatentSDE-shaped module as a field of their own, zero-padded to the kernels' width: what
    `verify` probes and `ComposedField` tabulates.  f / g are the MODULE'S OWN posterior drift and shared diffusion."""

    def __init__(self, sde, Hl, P):
        self.sde, self.Hl, self.P = sde, Hl, P
        self.coeffs = self.times = None

    def parameters(self):
        yield from self.sde.parameters()
        for name in ('sigma', 'theta', 'mu'):      # buffers of the prior / diffusion: part of the cached inputs' identity
            v = getattr(self.sde, name, None)
            if torch.is_tensor(v):
                yield v

    def _pad(self, v):
        return torch.nn.functional.pad(v, (0, self.P - self.Hl))

    def f(self, t, y):
        return self._pad(self.sde.f(t, y[:, :self.Hl]))

    def g(self, t, y):
        return self._pad(self.sde.g(t, y[:, :self.Hl]))


_WIDTHS = (16, 32, 64, 128, 256)


def _graph_leaves(out):
    """Leaf tensors (parameters) an autograd result depends on."""
    seen, leaves, stack = set(), [], [out.grad_fn] if out is not None and out.grad_fn is not None else []
    while stack:
        fn = stack.pop()
        if fn is None or fn in seen:
            continue
        seen.add(fn)
        v = getattr(fn, 'variable', None)
        if torch.is_tensor(v):
            leaves.append(v)
        stack.extend(nf for nf, _ in fn.next_functions)
    return leaves


def prior_watch(sde, width):
    """The tensors the prior drift h and the diffusion g of a LatentSDE-shaped module depend on: the parameters in their autograd
    graphs (a learnable prior / diffusion) plus every buffer and plain tensor attribute (the reference's theta, mu, sigma).  The
    in-solve KL accumulator keeps the prior's closed form as two Python floats (snsde_solve.kl_prior_a / _b), so these tensors'
    versions belong to the identity of the cached mapping (load_state_dict / fill_ / training a learnable prior: ADVICE r4)."""
    leaves = []
    try:
        p0 = next(sde.parameters())
        with torch.enable_grad():
            y = torch.zeros(2, width - 1, device=p0.device, dtype=p0.dtype)
            t = torch.tensor(0.3, device=p0.device, dtype=p0.dtype)
            outs = [fn(t, y) for fn in (getattr(sde, 'h', None), getattr(sde, 'g', None)) if callable(fn)]
        for o in outs:
            leaves += _graph_leaves(o)
    except Exception:
        leaves = list(sde.parameters())      # (cannot tell: every parameter counts)
    plain = [v for v in sde.__dict__.values() if torch.is_tensor(v)]
    return leaves, leaves + list(sde.buffers()) + plain


def _versions(tensors):
    return tuple((id(t), t.data_ptr(), t._version) for t in tensors)


def compose_latent(sde, names, width):
    """torch-ists' LatentSDE (diff_module/NSDE/latent_sde.py:31-89) and modules of its shape, solved through
    names={'drift': 'f_aug', 'diffusion': 'g_aug'}: the state is [latent (width - 1) | KL accumulator]; the latent channels
    follow the posterior drift f = MLP([sin t, cos t, y]) (relu) with the constant shared diffusion g, and the accumulator
    integrates 0.5 |(f - h) / g|^2 without feeding back.  Returns the ComposedField of the LATENT dynamics (the fused solve's
    part; torchsde._sdeint_latent adds the accumulator as one batched quadrature over the solve's states), or None."""
    if not isinstance(sde, torch.nn.Module) or not isinstance(names, dict):
        return None
    if names.get('drift') != 'f_aug' or names.get('diffusion') != 'g_aug' or (set(names) - {'drift', 'diffusion'}):
        return None
    # keyed on what the structural probes looked at: the layer objects and the functions themselves (a module whose layers are
    # replaced or whose f / g / f_aug / g_aug are re-bound later is recognised again instead of being solved on a stale mapping)
    fn = lambda name: (id(sde.__dict__.get(name)), id(getattr(type(sde), name, None)))
    lins = getattr(sde, 'linears', None)
    key = (width, id(getattr(sde, 'linear_in', None)), id(getattr(sde, 'linear_out', None)), id(lins),
           tuple(id(m) for m in lins) if isinstance(lins, (torch.nn.ModuleList, list, tuple)) else None,
           fn('f'), fn('g'), fn('f_aug'), fn('g_aug'), fn('h'))
    slot = _slot(sde)
    if slot is not None and slot.latent is not None and slot.latent[0] == key and slot.latent[3] == _versions(slot.latent[2]):
        return slot.latent[1]
    result = _compose_latent(sde, width)
    leaves, watch = prior_watch(sde, width)
    if result is not None:
        result.parts['prior_leaves'] = leaves      # parameters h() / g() depend on (a learnable prior or diffusion)
    if slot is not None:
        slot.latent = (key, result, watch, _versions(watch))
    return result


def _compose_latent(sde, width):
    if getattr(sde, 'sde_type', None) != 'ito' or getattr(sde, 'noise_type', None) != 'diagonal':
        return None
    if not all(callable(getattr(sde, a, None)) for a in ('f', 'g', 'f_aug', 'g_aug')):
        return None
    lin_in, lin_out, linears = getattr(sde, 'linear_in', None), getattr(sde, 'linear_out', None), getattr(sde, 'linears', None)
    if not isinstance(lin_in, torch.nn.Linear) or not isinstance(lin_out, torch.nn.Linear) or lin_in.bias is None or lin_out.bias is None:
        return None
    if not isinstance(linears, (torch.nn.ModuleList, list, tuple)):
        return None
    linears = list(linears)
    Hl, HH = width - 1, lin_in.out_features
    if Hl < 1 or lin_in.in_features != Hl + 2 or lin_out.in_features != HH or lin_out.out_features != Hl or len(linears) > 3:
        return None
    if any(not _is_linear(m, HH, HH) for m in linears):
        return None
    # what the split relies on, checked on the module's own functions: the augmented drift / diffusion restricted to the
    # latent channels are f / g, neither depends on the accumulator, g depends on neither y nor the accumulator, and the
    # accumulator carries no noise
    try:
        with torch.no_grad():
            p0 = lin_in.weight
            gen = torch.Generator().manual_seed(5)
            y = torch.randn(3, Hl, generator=gen).to(p0)
            a1, a2 = torch.zeros(3, 1).to(p0), torch.full((3, 1), 1.7).to(p0)
            t = torch.tensor(0.41).to(p0)
            fa1, fa2 = sde.f_aug(t, torch.cat([y, a1], dim=1)), sde.f_aug(t, torch.cat([y, a2], dim=1))
            ga1, ga2 = sde.g_aug(t, torch.cat([y, a1], dim=1)), sde.g_aug(t, torch.cat([-2.0 * y, a2], dim=1))
            f, g = sde.f(t, y), sde.g(t, y)
    except Exception:
        return None
    if tuple(fa1.shape) != (3, width) or tuple(ga1.shape) != (3, width) or tuple(f.shape) != (3, Hl) or tuple(g.shape) != (3, Hl):
        return None
    if not (torch.allclose(fa1, fa2) and torch.allclose(fa1[:, :Hl], f) and torch.allclose(ga1, ga2)
            and torch.allclose(ga1[:, :Hl], g) and float(ga1[:, -1].abs().max()) == 0.0):
        return None
    # The KL accumulator as a state column of the fused solve (snsde.h: kl_column1): f_aug's last channel must be
    # 1/2 sum_j ((f_j - (a y_j + b)) / g_j)^2 with a time-independent scalar linear prior drift a y + b (the module's own h) and a
    # diffusion that is constant in time - checked on the module's own functions; otherwise the accumulator stays the batched
    # quadrature of torchsde._sdeint_latent
    acc = None
    h_fn = getattr(sde, 'h', None)
    if callable(h_fn):
        try:
            with torch.no_grad():
                t2 = torch.tensor(0.83).to(p0)
                z, o = torch.zeros(3, Hl).to(p0), torch.ones(3, Hl).to(p0)
                h0, h1, h2, h0b = h_fn(t, z), h_fn(t, o), h_fn(t, y), h_fn(t2, z)
                a_, b_ = float((h1 - h0).flatten()[0]), float(h0.flatten()[0])
                gs = torch.where(g.abs() > 1e-7, g, 1e-7 * g.sign())
                u = 0.5 * (((f - (a_ * y + b_)) / gs) ** 2).sum(dim=1)
                if (tuple(h0.shape) == (3, Hl) and torch.allclose(h1 - h0, torch.full_like(h0, a_)) and torch.allclose(h0, torch.full_like(h0, b_))
                        and torch.allclose(h2, a_ * y + b_, rtol=1e-5, atol=1e-6) and torch.allclose(h0b, h0)
                        and torch.allclose(sde.g(t2, y), g) and torch.allclose(fa1[:, -1], u, rtol=1e-4, atol=1e-6)):
                    acc = (a_, b_)
        except Exception:
            acc = None
    P = next((w for w in _WIDTHS if w >= max(Hl + (1 if acc is not None else 0), HH)), None)
    if acc is not None and (P is None or P > 128):      # no spare column below the widest kernel: keep the split solve
        acc, P = None, next((w for w in _WIDTHS if w >= max(Hl, HH)), None)
    if P is None or P > 128:       # (the SRK variant and the training-mode lean kernels: H <= 128)
        return None
    model = engine.model_struct(1, P, P, len(linears) + 1, 4, 12, activation=ACT_RELU, drift_output=DRIFT_LINEAR,
                                diffusion_output=DIFFUSION_RAW, time_feature=TIME_SINCOS)
    try:
        layout, numel = _lib.param_layout(model)
    except _lib.SnsdeError:
        return None
    parts = dict(latent=Hl, linear_in=lin_in, linears=linears, linear_out=lin_out, acc=acc)
    return ComposedField(_LatentView(sde, Hl, P), model, layout, numel, parts, additive=True)


@torch.no_grad()
def verify(field, coeffs, times_host, dev):
    """One-step probe THROUGH THE KERNEL against the module's own f / g on the first rows of the batch: fixes
    drift_output (z or z * y) and confirms the whole mapping.  Memoised per device."""
    hit = field.verified.get(str(dev))
    if hit is not None:
        return hit
    sde, H = field.sde, field.model.hidden_channels
    rows = min(int(coeffs.shape[0]), 8)
    c = coeffs[:rows].contiguous()
    t_lo, t_hi = float(times_host[0]), float(times_host[-1])
    h = np.float32(max(0.05 * (t_hi - t_lo), 1e-3))
    gen = torch.Generator().manual_seed(11)
    # mixed-sign states and two distinct times: a field that uses relu(y) / |y| / clamp(y) where the kernel multiplies by y, or
    # whose time dependence differs from the composed one, must not pass on a lucky probe point
    y0 = ((torch.rand(rows, H, generator=gen) * 2.0 - 1.0) * 1.5).to(dev)
    y0 = torch.where(y0.abs() < 0.05, torch.full_like(y0, 0.7), y0)
    saved = (sde.coeffs, sde.times)
    ok = False
    drift0 = field.model.drift_output          # (kept if the probe fails here but passed on another device)

    def probe(frac, drift):
        t = np.float32(t_lo + frac * (t_hi - t_lo))
        grid = engine.StepGrid(np.array([t, t + h], dtype=np.float32), float(2 * h), times_host, dev)
        if grid.N != 1:
            return False
        hh = float(grid.t1[0] - grid.t0[0])
        tt = torch.tensor(float(grid.t0[0]), device=dev)
        f_ref, g_ref = sde.f(tt, y0).float(), sde.g(tt, y0).float()
        if g_ref.dim() == 3:           # scalar noise (the ODE field): (rows, H, 1)
            g_ref = g_ref.squeeze(-1)
        tab = field.noise_table(torch.from_numpy(grid.t0), dev) if field.tabulated else None
        flat = field.flat(dev)
        scale_f = float(f_ref.abs().max()) + 1e-6
        scale_g = float(g_ref.abs().max()) + 1e-6
        field.model.drift_output = drift
        outs = []
        for w in (0.0, 1.0):
            dW = torch.full((1, rows, H), w, device=dev)
            call = engine.SolveCall(field.model, flat, c, grid, y0, dW=dW, method='euler', noise_table=tab)
            outs.append(call.launch()[1].clone())
        f_k = (outs[0] - y0) / hh
        g_k = outs[1] - outs[0]
        return (float((f_k - f_ref).abs().max()) <= 2e-4 * scale_f + 2e-5 / hh
                and float((g_k - g_ref).abs().max()) <= 2e-4 * scale_g + 2e-5)

    try:
        sde.set_X(c, sde.times) if hasattr(sde, 'set_X') else None
        for drift in ((DRIFT_LINEAR, DRIFT_TIMES_Y) if field.tabulated else (DRIFT_LINEAR,)):
            if probe(0.31, drift) and probe(0.67, drift):
                ok = True
                break
    except (_lib.SnsdeError, ValueError, AttributeError, TypeError):
        ok = False
    finally:
        if hasattr(sde, 'set_X'):
            sde.set_X(*saved)
        if not ok:
            field.model.drift_output = drift0
    field.verified[str(dev)] = ok
    return ok
