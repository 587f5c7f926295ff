"""The two ``torchcde`` (0.2.5) names the reference's SDE path uses:
``CubicSpline`` (models_sde/neuralsde.py:184) and
``hermite_cubic_coefficients_with_backward_differences`` (datasets/common.py:82-84,
tests/test_neuralsde_core_alignment.py:64).  torchcde itself is not vendored in the reference; the
behaviour is restated from its published semantics."""
import torch

from .controldiffeq import NaturalCubicSpline, natural_cubic_spline_coeffs  # noqa: F401


class CubicSpline:
    """``CubicSpline(coeffs, t)`` with coeffs (..., L-1, 4C) = cat[a, b, two_c, three_d]."""

    def __init__(self, coeffs, t=None, **kwargs):
        if t is None:
            t = torch.linspace(0, coeffs.size(-2), coeffs.size(-2) + 1, dtype=coeffs.dtype, device=coeffs.device)
        if coeffs.size(-1) % 4 != 0:
            raise ValueError("The last dimension of coeffs must be 4 * channels.")
        if t.dim() != 1 or t.size(0) != coeffs.size(-2) + 1:
            raise ValueError("t must be one dimensional with one more entry than coeffs has intervals.")
        self._t = t
        self._coeffs = coeffs
        self._spline = NaturalCubicSpline.from_packed(t, coeffs)

    @property
    def grid_points(self):
        return self._t

    @property
    def interval(self):
        return torch.stack([self._t[0], self._t[-1]])

    def evaluate(self, t):
        return self._spline.evaluate(t)

    def derivative(self, t):
        return self._spline.derivative(t)


def _fill_linear(x, t):
    """NaN -> linear interpolation in time between observed neighbours; ends take the nearest observation;
    all-NaN channels become zero.  x (S, L)."""
    S, L = x.shape
    obs = ~torch.isnan(x)
    ar = torch.arange(L, device=x.device)
    big = L + 1
    prev_idx = torch.where(obs, ar, -1).cummax(dim=1).values                       # last observed <= j
    next_idx = torch.where(obs, ar, big).flip(1).cummin(dim=1).values.flip(1)       # first observed >= j
    has_prev, has_next = prev_idx >= 0, next_idx < big
    pi = prev_idx.clamp(min=0)
    ni = next_idx.clamp(max=L - 1)
    xz = torch.nan_to_num(x)
    xp, xn = torch.gather(xz, 1, pi), torch.gather(xz, 1, ni)
    tp, tn = t[pi], t[ni]
    span = (tn - tp)
    w = torch.where(span > 0, (t[None, :] - tp) / torch.where(span > 0, span, torch.ones_like(span)),
                    torch.zeros_like(span))
    mid = xp + w * (xn - xp)
    out = torch.where(has_prev & has_next, mid, torch.where(has_prev, xp, xn))
    return torch.where(has_prev | has_next, out, torch.zeros_like(out))


def hermite_cubic_coefficients_with_backward_differences(x, t=None):
    """x (..., L, C), t (L,) -> (..., L-1, 4C).  Per interval k of width h: a = x_k,
    b = previous secant slope (its own slope on the first interval), and the cubic is fixed by
    p(h) = x_{k+1}, p'(h) = m_k:  two_c = 4 (m_k - b)/h,  three_d = -3 (m_k - b)/h^2."""
    L, Cn = x.shape[-2], x.shape[-1]
    if t is None:
        t = torch.linspace(0, L - 1, L, dtype=x.dtype, device=x.device)
    t = t.to(x.dtype)
    if t.dim() != 1 or t.size(0) != L:
        raise ValueError("t must be one dimensional with the same length as the time dimension of x.")
    if L < 2:
        raise ValueError("Must have a time dimension of size at least 2.")
    if x.is_cuda and x.dtype == torch.float32 and not x.requires_grad:     # HIP construction kernel
        from . import engine
        lead = x.shape[:-2]
        out = engine.spline_coeffs(t.to(device=x.device).contiguous(), x.reshape(-1, L, Cn).contiguous(), 'hermite')
        return out.reshape(*lead, L - 1, 4 * Cn)
    if bool(torch.isnan(x).any()):
        lead = x.shape[:-2]
        series = x.transpose(-1, -2).reshape(-1, L)
        x = _fill_linear(series, t).reshape(*lead, Cn, L).transpose(-1, -2)
    h = (t[1:] - t[:-1]).unsqueeze(-1)
    m = (x[..., 1:, :] - x[..., :-1, :]) / h
    b = torch.cat([m[..., :1, :], m[..., :-1, :]], dim=-2)
    a = x[..., :-1, :]
    two_c = 4 * (m - b) / h
    three_d = -3 * (m - b) / (h * h)
    return torch.cat([a, b, two_c, three_d], dim=-1)
