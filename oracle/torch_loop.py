"""CPU ORACLE, second form (test / baseline infrastructure, NOT product code): the reference's
per-step arithmetic restated op-for-op with PyTorch CPU tensors in float32 - the same ATen/MKL
kernels the reference itself runs on CPU (``Diffusion_model.f/g`` neuralsde.py:295-307, the
vendored spline evaluate interpolate.py:263-276, and the fixed-step Euler update of torchsde).

Used for (a) ``bench.py``'s ``cpu_baseline`` ("port": the reference is Python and cannot travel to
the GPU box, and torchsde/torchcde are not installed), timed on the host cores, and (b) a
cross-check of the numpy oracle.  It deliberately keeps the reference's op granularity (one ATen
call per line of the reference) and omits only torchsde's BrownianInterval bookkeeping, so it is a
conservative (fast) stand-in for the real CPU path.
"""
import torch


def spline_evaluate(coeffs, times, t):
    C = coeffs.shape[-1] // 4
    maxlen = coeffs.shape[-2] - 1
    index = ((t > times).sum() - 1).clamp(0, maxlen)
    frac = t - times[index]
    row = coeffs[..., index, :]
    a, b, two_c, three_d = row[..., :C], row[..., C:2 * C], row[..., 2 * C:3 * C], row[..., 3 * C:]
    inner = 0.5 * two_c + three_d * frac / 3
    inner = b + inner * frac
    return a + inner * frac


def _lin(x, p, name):
    return torch.nn.functional.linear(x, p[name + '.weight'], p[name + '.bias'])


def _tau(t, y):
    tt = torch.full_like(y[:, 0], fill_value=float(t)).unsqueeze(-1)
    return tt, torch.cat((torch.sin(tt), torch.cos(tt)), dim=-1)


def drift_f(p, io, t, y, coeffs, times):
    Xt = _lin(spline_evaluate(coeffs, times, t), p, 'initial_network')
    if io in (3, 4, 5, 6):
        yy = _lin(torch.cat((_tau(t, y)[1], y), dim=-1), p, 'linear_in')
    else:
        yy = _lin(y, p, 'linear_in')
    if io == 0:
        z = Xt
    elif io in (1, 3, 5):
        z = yy
    else:
        z = _lin(torch.cat([yy, Xt], dim=-1), p, 'emb')
    z = z.relu()
    i = 0
    while f'linears.{i}.weight' in p:
        z = _lin(z, p, f'linears.{i}').relu()
        i += 1
    z = _lin(z, p, 'linear_out')
    if io in (5, 6):
        z = z * y.tanh()
    return z.tanh()


def _noise_net(p, prefix, x):
    if f'{prefix}.0.weight' in p:
        return _lin(_lin(x, p, f'{prefix}.0').relu(), p, f'{prefix}.2')
    return _lin(x, p, prefix)


def diffusion_g(p, no, t, y):
    tt, tf = _tau(t, y)
    if no == 0:
        raw = torch.zeros_like(y)
    elif no in (1, 2, 3):
        s = p['sigma'].exp().expand(y.size(0), y.size(1))
        raw = s if no == 1 else (s * tt if no == 2 else s * y)
    elif no in (4, 5, 6):
        s = p['sigma_diag'].exp().repeat(y.size(0), 1)
        raw = s if no == 4 else (s * tt if no == 5 else s * y)
    elif no == 7:
        raw = torch.sqrt(y)
    elif no == 8:
        raw = y ** 3
    elif no == 9:
        raw = y.sigmoid()
    elif no == 10:
        raw = y.relu()
    elif no == 11:
        raw = tt * y
    elif no in (12, 13):
        raw = _noise_net(p, 'noise_t', tf)
        raw = raw * y if no == 13 else raw
    elif no in (14, 15):
        raw = _noise_net(p, 'noise_y', torch.cat([tf, y], dim=-1))
        raw = raw * y if no == 15 else raw
    elif no in (16, 17):
        raw = _noise_net(p, 'noise_t', tf).relu()
        raw = raw * y if no == 17 else raw
    else:
        raw = _noise_net(p, 'noise_y', torch.cat([tf, y], dim=-1)).relu()
        raw = raw * y if no == 19 else raw
    return (p['theta'].sigmoid() * torch.nan_to_num(raw)).tanh()


@torch.no_grad()
def euler_solve(p, io, no, coeffs, times, y0, t_start, n_steps, dt, dW=None, generator=None):
    """N Euler steps of size dt from t_start; returns the final state.  dW (N,B,H) or fresh randn."""
    y = y0
    sq = float(dt) ** 0.5
    for n in range(n_steps):
        t = torch.tensor(t_start + n * dt, dtype=y0.dtype)
        I = dW[n] if dW is not None else torch.randn(y.shape, dtype=y.dtype, generator=generator) * sq
        f = drift_f(p, io, t, y, coeffs, times)
        g = diffusion_g(p, no, t, y)
        y = y + f * dt + g * I
    return y
