"""Natural cubic spline control path: API of the reference's vendored ``controldiffeq`` package
(``natural_cubic_spline_coeffs`` / ``NaturalCubicSpline``;
/root/reference/benchmark_classification/controldiffeq/interpolate.py:161-283).

Coefficient construction is offline preprocessing in the reference (Python loops over
batch x channel x time, cached on disk).  Here it is one batched Thomas solve over every
(batch, channel) series at once, missing values included, written with tensor ops so it runs on
whatever device the data is on.  Evaluation (A10) on CUDA tensors goes through the HIP spline
kernel (``snsde_spline_evaluate``).
"""
import importlib.machinery
import importlib.util
import os
import sys

import torch

from . import engine

_VENDORED = None


def _vendored_package():
    """The reference's own vendored ``controldiffeq`` package (benchmark_*/controldiffeq, found on sys.path), loaded under a
    private name.  After ``stable_neural_sdes_amd.install()`` this mirror answers ``import controldiffeq``; names it does
    not implement (``cdeint`` and friends, used by the reference's CDE baselines built by the same ``make_model``) are
    served from the vendored package so those baselines keep working in the same process."""
    global _VENDORED
    if _VENDORED is None:
        _VENDORED = False
        here = os.path.dirname(os.path.abspath(__file__))
        for entry in sys.path:
            init = os.path.join(entry or '.', 'controldiffeq', '__init__.py')
            if os.path.isfile(init) and os.path.abspath(os.path.dirname(os.path.dirname(init))) != here:
                name = '_snsde_vendored_controldiffeq'
                spec = importlib.util.spec_from_file_location(name, init, submodule_search_locations=[os.path.dirname(init)])
                mod = importlib.util.module_from_spec(spec)
                sys.modules[name] = mod
                try:
                    spec.loader.exec_module(mod)
                    _VENDORED = mod
                except Exception:      # its own dependencies (torchdiffeq) are missing: nothing to delegate to
                    sys.modules.pop(name, None)
                break
    return _VENDORED or None


def __getattr__(name):
    if name.startswith('__'):
        raise AttributeError(name)
    mod = _vendored_package()
    if mod is not None and hasattr(mod, name):
        return getattr(mod, name)
    raise AttributeError(f"module 'controldiffeq' (stable_neural_sdes_amd mirror) has no attribute {name!r}: the mirror "
                         "implements natural_cubic_spline_coeffs / NaturalCubicSpline only and found no importable "
                         "vendored controldiffeq package on sys.path to delegate to")


def _series_coeffs(times, x):
    """times (L,), x (S, L) with NaN = missing  ->  a, b, two_c, three_d each (S, L-1).

    Per series: impute the first/last observation at the ends (interpolate.py:100-114), fit the natural
    spline through the observed knots only, then re-express every observed-interval cubic on each
    original sub-interval (interpolate.py:131-150).
    """
    S, L = x.shape
    dt, dev = x.dtype, x.device
    obs = ~torch.isnan(x)
    any_obs = obs.any(dim=1)
    ar = torch.arange(L, device=dev)
    first = torch.where(obs, ar, L).min(dim=1).values.clamp(max=L - 1)
    last = torch.where(obs, ar, -1).max(dim=1).values.clamp(min=0)
    rows = torch.arange(S, device=dev)
    x = x.clone()
    x[rows, 0] = torch.where(obs[:, 0], x[:, 0], x[rows, first])
    x[rows, L - 1] = torch.where(obs[:, L - 1], x[:, L - 1], x[rows, last])
    obs = obs.clone()
    obs[:, 0] = True
    obs[:, L - 1] = True
    x = torch.where(any_obs[:, None], x, torch.zeros_like(x))
    m = obs.sum(dim=1)                                   # observed knots per series (>= 2)
    # compress observed knots to the front (stable): position of knot j among the observed ones
    pos = obs.cumsum(dim=1) - 1                          # (S, L)
    order = torch.argsort((~obs).to(torch.int8), dim=1, stable=True)
    tc = times[order]                                    # (S, L) observed times first
    xc = torch.gather(torch.nan_to_num(x), 1, order)
    valid = ar[None, :] < m[:, None]                     # compressed slot holds an observed knot
    iv = ar[None, :-1] < (m[:, None] - 1)                # compressed interval i is real
    h = torch.where(iv, tc[:, 1:] - tc[:, :-1], torch.ones((), dtype=dt, device=dev))
    rec = torch.where(iv, h.reciprocal(), torch.zeros((), dtype=dt, device=dev))
    dx = torch.where(iv, xc[:, 1:] - xc[:, :-1], torch.zeros((), dtype=dt, device=dev))
    scaled = 3 * dx * rec * rec
    diag = torch.zeros(S, L, dtype=dt, device=dev)
    diag[:, :-1] += rec
    diag[:, 1:] += rec
    diag = torch.where(valid, 2 * diag, torch.ones((), dtype=dt, device=dev))
    rhs = torch.zeros(S, L, dtype=dt, device=dev)
    rhs[:, :-1] += scaled
    rhs[:, 1:] += scaled
    # Thomas algorithm (controldiffeq/misc.py:12-66), all series in lock-step
    nd = [diag[:, 0]]
    nb = [rhs[:, 0]]
    for i in range(1, L):
        w = rec[:, i - 1] / nd[i - 1]
        nd.append(diag[:, i] - w * rec[:, i - 1])
        nb.append(rhs[:, i] - w * nb[i - 1])
    kd = [None] * L
    kd[L - 1] = nb[L - 1] / nd[L - 1]
    for i in range(L - 2, -1, -1):
        kd[i] = (nb[i] - rec[:, i] * kd[i + 1]) / nd[i]
    kd = torch.stack(kd, dim=1)                          # knot derivatives (compressed slots)
    a_c = xc[:, :-1]
    b_c = kd[:, :-1]
    two_c = (6 * dx * rec - 4 * kd[:, :-1] - 2 * kd[:, 1:]) * rec
    three_d = (-6 * dx * rec + 3 * (kd[:, :-1] + kd[:, 1:])) * rec * rec
    # expand: original interval j lies in observed interval p = pos[j]
    p = pos[:, :-1]
    off = torch.gather(tc, 1, p) - times[None, :-1]      # prev observed time - time_j  (<= 0)
    ga, gb = torch.gather(a_c, 1, p), torch.gather(b_c, 1, p)
    gc, gd = torch.gather(two_c, 1, p), torch.gather(three_d, 1, p)
    a_inner = (0.5 * gc - gd * off / 3) * off
    a = ga + (a_inner - gb) * off
    b = gb + (gd * off - gc) * off
    c2 = gc - 2 * gd * off
    zero = ~any_obs[:, None]
    z = torch.zeros((), dtype=dt, device=dev)
    return (torch.where(zero, z, a), torch.where(zero, z, b), torch.where(zero, z, c2), torch.where(zero, z, gd))


def natural_cubic_spline_coeffs(t, X):
    """Same contract as the reference (interpolate.py:161-228): t (L,) increasing, X (..., L, C) with NaN for
    missing values -> four tensors (..., L-1, C): a, b, two_c, three_d."""
    if not t.is_floating_point() or not X.is_floating_point():
        raise ValueError("t and X must both be floating point/")
    if t.dim() != 1:
        raise ValueError("t must be one dimensional.")
    if bool((t[1:] <= t[:-1]).any()):
        raise ValueError("t must be monotonically increasing.")
    if X.dim() < 2:
        raise ValueError("X must have at least two dimensions, corresponding to time and channels.")
    if X.size(-2) != t.size(0):
        raise ValueError("The time dimension of X must equal the length of t.")
    if t.size(0) < 2:
        raise ValueError("Must have a time dimension of size at least 2.")
    L, Cn = X.shape[-2], X.shape[-1]
    lead = X.shape[:-2]
    if X.is_cuda and X.dtype == torch.float32 and not X.requires_grad:     # HIP construction kernel
        packed = engine.spline_coeffs(t.to(device=X.device, dtype=torch.float32).contiguous(),
                                      X.reshape(-1, L, Cn).contiguous(), 'natural')
        return tuple(packed[..., k * Cn:(k + 1) * Cn].reshape(*lead, L - 1, Cn).contiguous() for k in range(4))
    series = X.transpose(-1, -2).reshape(-1, L)
    outs = _series_coeffs(t.to(X.dtype), series)
    return tuple(o.reshape(*lead, Cn, L - 1).transpose(-1, -2).contiguous() for o in outs)


class _HostTimes:
    """Host copy of a small device vector (knot grid, output times): one device->host copy — a stream sync — per
    distinct tensor version.  The cache keeps the tensor alive, so its address cannot be recycled for different
    values while the entry exists; in-place edits bump `_version`."""
    _cache = {}

    @classmethod
    def get(cls, times):
        if not torch.is_tensor(times):
            return torch.as_tensor(times, dtype=torch.float32).numpy()
        if times.device.type == 'cpu':
            return times.detach().to(torch.float32).numpy()
        key = (times.data_ptr(), times._version, tuple(times.shape), times.stride(), times.dtype, str(times.device))
        hit = cls._cache.get(key)
        if hit is None:
            if len(cls._cache) > 64:
                cls._cache.clear()
            hit = (times.detach().to('cpu', torch.float32).numpy(), times)
            cls._cache[key] = hit
        return hit[0]

    @classmethod
    def known(cls, t):
        """True when `scalar(t)` needs no device->host copy (python number, CPU tensor, or an element of a cached vector)."""
        if not (torch.is_tensor(t) and t.is_cuda):
            return True
        if t.numel() != 1:
            return False
        try:
            base = t.untyped_storage().data_ptr()
        except Exception:
            return False
        return any(ref.untyped_storage().data_ptr() == base and key[1] == t._version and key[4] == t.dtype
                   for key, (_, ref) in cls._cache.items())

    @classmethod
    def scalar(cls, t):
        """float(t) without a stream sync when `t` is an element of a device vector whose host copy is cached (the
        usual `X.evaluate(times[0])` of the reference's models, neuralsde.py:66): the value is read from the host copy
        at the view's storage offset.  Anything else falls back to float(t)."""
        if torch.is_tensor(t) and t.is_cuda and t.numel() == 1:
            try:
                base = t.untyped_storage().data_ptr()
            except Exception:
                return float(t)
            for (ptr, version, shape, stride, dtype, dev), (host, ref) in cls._cache.items():
                if (ref.untyped_storage().data_ptr() == base and version == t._version and dtype == t.dtype
                        and len(shape) == 1 and str(t.device) == dev):
                    k, rem = divmod(t.storage_offset() - ref.storage_offset(), stride[0] if stride[0] else 1)
                    if rem == 0 and 0 <= k < shape[0]:
                        return float(host[k])
        return float(t)


class NaturalCubicSpline:
    """``NaturalCubicSpline(times, (a, b, two_c, three_d))`` with ``evaluate(t)`` / ``derivative(t)``
    (interpolate.py:231-283)."""

    def __init__(self, times, coeffs, **kwargs):
        a, b, two_c, three_d = coeffs
        self._times = times
        self._packed = torch.cat([a, b, two_c, three_d], dim=-1)
        self._channels = a.size(-1)

    @classmethod
    def from_packed(cls, times, packed):
        self = cls.__new__(cls)
        self._times = times
        self._packed = packed
        self._channels = packed.size(-1) // 4
        return self

    def _interpret_t(self, t):
        times = _HostTimes.get(self._times)
        tv = _HostTimes.scalar(t)
        t32 = times.dtype.type(tv)
        idx = int((t32 > times).sum()) - 1
        idx = min(max(idx, 0), times.shape[0] - 2)
        return t32 - times[idx], idx

    def _eval_on_device(self, t, derivative):
        """Same arithmetic with the interval found on the device (`t` is a CUDA scalar whose value the host does not know):
        no stream sync, recordable into a CUDA/HIP graph — the generic-sde stepper relies on it."""
        P, times = self._packed, self._times.to(self._packed.device)
        tt = t.to(times.dtype).reshape(())
        idx = ((tt > times).sum() - 1).clamp(0, times.shape[0] - 2)
        idx = idx.reshape(1)                      # (indexing with a 0-dim tensor would read it back on the host)
        frac = (tt - times.index_select(0, idx).squeeze(0)).to(P.dtype)
        row = P.index_select(-2, idx).squeeze(-2)
        Cn = self._channels
        a, b, c2, d3 = (row[..., k * Cn:(k + 1) * Cn] for k in range(4))
        if derivative:
            return b + (c2 + d3 * frac) * frac
        return a + (b + (0.5 * c2 + d3 * frac / 3) * frac) * frac

    def _eval(self, t, derivative):
        if torch.is_tensor(t) and t.is_cuda and self._packed.is_cuda and not _HostTimes.known(t):
            return self._eval_on_device(t, derivative)
        frac, idx = self._interpret_t(t)
        P = self._packed
        if P.is_cuda and P.dtype == torch.float32 and not P.requires_grad:
            flat = P.reshape(-1, P.shape[-2], P.shape[-1])
            flat = flat if flat.is_contiguous() else flat.contiguous()
            out = engine.spline_evaluate(flat, idx, float(frac), derivative)
            return out.reshape(*P.shape[:-2], self._channels)
        Cn = self._channels
        row = P[..., idx, :]
        a, b, c2, d3 = (row[..., k * Cn:(k + 1) * Cn] for k in range(4))
        frac = float(frac) if P.dtype != torch.float32 else frac.item()
        if derivative:
            return b + (c2 + d3 * frac) * frac
        return a + (b + (0.5 * c2 + d3 * frac / 3) * frac) * frac

    def evaluate(self, t):
        return self._eval(t, False)

    def derivative(self, t):
        return self._eval(t, True)
