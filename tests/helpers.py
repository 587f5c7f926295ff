"""Shared test helpers: fixture loading and the reference's parameter naming contract."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def param_spec(io, no, NL, C, H, HH=None):
    """state_dict names/shapes of the reference Diffusion_model, in state_dict order
    (/root/reference/benchmark_classification/models_sde/neuralsde.py:123-179)."""
    HH = H if HH is None else HH
    spec = [('theta', (1, 1))]
    if no in (1, 2, 3):
        spec.append(('sigma', (1,)))
    if no in (4, 5, 6):
        spec.append(('sigma_diag', (H,)))
    spec += [('initial_network.weight', (H, C)), ('initial_network.bias', (H,))]
    k_in = H + 2 if io in (3, 4, 5, 6) else H
    spec += [('linear_in.weight', (HH, k_in)), ('linear_in.bias', (HH,))]
    if io in (2, 4, 6):
        spec += [('emb.weight', (H, 2 * H)), ('emb.bias', (H,))]
    for i in range(NL - 1):
        spec += [(f'linears.{i}.weight', (HH, HH)), (f'linears.{i}.bias', (HH,))]
    spec += [('linear_out.weight', (H, HH)), ('linear_out.bias', (H,))]
    if no in (12, 13):
        spec += [('noise_t.weight', (H, 2)), ('noise_t.bias', (H,))]
    if no in (14, 15):
        spec += [('noise_y.weight', (H, H + 2)), ('noise_y.bias', (H,))]
    if no in (16, 17):
        spec += [('noise_t.0.weight', (H, 2)), ('noise_t.0.bias', (H,)),
                 ('noise_t.2.weight', (H, H)), ('noise_t.2.bias', (H,))]
    if no in (18, 19):
        spec += [('noise_y.0.weight', (H, H + 2)), ('noise_y.0.bias', (H,)),
                 ('noise_y.2.weight', (H, H)), ('noise_y.2.bias', (H,))]
    return spec


def unflatten(vec, spec):
    out, off = {}, 0
    for name, shape in spec:
        n = int(np.prod(shape))
        out[name] = np.asarray(vec[off:off + n]).reshape(shape)
        off += n
    assert off == len(vec), (off, len(vec))
    return out


def group(npz, prefix):
    """Sub-dict of an npz whose keys start with prefix/ (prefix stripped)."""
    pre = prefix + '/'
    return {k[len(pre):]: npz[k] for k in npz.files if k.startswith(pre)}


def params_of(npz, prefix):
    return group(npz, prefix + '/param')


def random_params(rng, io, no, NL, C, H, scale=None):
    """nn.Linear-like init (U(-1/sqrt(fan_in), 1/sqrt(fan_in))) with non-trivial theta/sigma."""
    p = {}
    for name, shape in param_spec(io, no, NL, C, H):
        if name == 'theta':
            p[name] = np.array([[0.8]], dtype=np.float32)
        elif name in ('sigma', 'sigma_diag'):
            p[name] = (-0.5 + 0.3 * rng.standard_normal(shape)).astype(np.float32)
        else:
            fan_in = shape[-1] if name.endswith('weight') else None
            if fan_in is None:
                wname = name[:-4] + 'weight'
                fan_in = dict(param_spec(io, no, NL, C, H))[wname][-1]
            b = (scale or 1.0) / np.sqrt(fan_in)
            p[name] = rng.uniform(-b, b, size=shape).astype(np.float32)
    return p
