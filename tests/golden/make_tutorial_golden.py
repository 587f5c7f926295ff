#!/usr/bin/env python
"""Golden vectors of the tutorial notebooks' vector fields (build container only; needs /root/reference).

Run:  python tests/golden/make_tutorial_golden.py        (writes tests/golden/tutorial.npz)

For each of `tutorial/simple OU process - Neural {LSDE, LNSDE, LNSDE (additive), GSDE}.ipynb` the code cell that defines the
vector field (LipSwish, MLP, Neural*Func) is EXECUTED from the notebook as it lies under /root/reference - nothing of it
is stored here - over the same stand-ins as make_golden.py (`torchcde.CubicSpline` adapted onto the vendored
NaturalCubicSpline).  The fixture holds data only: the module's state_dict, inputs, f / g at a few
(t, y), and fixed-step Euler / Milstein trajectories driving the notebook's f / g on supplied increments (fp32 and fp64).
tests/tutorial_fields.TutorialField (the test-side class with the notebooks' attribute names) must load these state_dicts
and reproduce the values; the fused path is then checked against the trajectories on the GPU.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G      # noqa: E402

NOTEBOOKS = [
    # key, file, class, kind of tests.tutorial_fields.TutorialField, hidden, layers, activation, method
    ('lsde', 'simple OU process - Neural LSDE.ipynb', 'NeuralLSDEFunc', 'lsde', 32, 1, 'lipswish', 'euler'),
    ('lnsde', 'simple OU process - Neural LNSDE.ipynb', 'NeuralLNSDEFunc', 'lnsde', 32, 2, 'lipswish', 'euler'),
    ('lnsde_additive', 'simple OU process - Neural LNSDE (additive).ipynb', 'NeuralLNSDEFunc', 'lnsde_additive', 64, 1, 'relu',
     'euler'),
    ('gsde', 'simple OU process - Neural GSDE.ipynb', 'NeuralGSDEFunc', 'gsde', 32, 2, 'lipswish', 'euler'),
    ('gsde_milstein', 'simple OU process - Neural GSDE.ipynb', 'NeuralGSDEFunc', 'gsde', 64, 1, 'lipswish', 'milstein'),
    ('gsde_srk', 'simple OU process - Neural GSDE (srk solver).ipynb', 'NeuralGSDEFunc', 'gsde', 32, 1, 'lipswish', 'srk'),
    ('lnsde_srk', 'simple OU process - Neural LNSDE.ipynb', 'NeuralLNSDEFunc', 'lnsde', 64, 2, 'lipswish', 'srk'),
    ('nsde', 'simple OU process - Neural SDE.ipynb', 'NeuralSDEFunc', 'nsde', 32, 1, 'lipswish', 'euler'),
    ('nsde_srk', 'simple OU process - Neural SDE.ipynb', 'NeuralSDEFunc', 'nsde', 64, 1, 'lipswish', 'srk'),
    ('nsde_relu', 'simple OU process - Neural SDE.ipynb', 'NeuralSDEFunc', 'nsde', 16, 1, 'relu', 'milstein'),
    ('ode', 'simple OU process - Neural ODE.ipynb', 'NeuralODEFunc', 'ode', 32, 2, 'lipswish', 'euler'),
]


def field_class(nb_file, cls, torchcde):
    nb = json.load(open(os.path.join(G.REF_ROOT, 'tutorial', nb_file)))
    cells = [''.join(c['source']) for c in nb['cells'] if c['cell_type'] == 'code' and f'class {cls}(' in ''.join(c['source'])]
    assert len(cells) == 1, (nb_file, len(cells))
    src = cells[0]
    src = src[:src.index('class NDE_model')]          # LipSwish, MLP and the vector field; the wrapper needs torchsde
    ns = {'torch': torch, 'nn': torch.nn, 'torchcde': torchcde}
    exec(compile(src, nb_file, 'exec'), ns)
    return ns[cls]


class _squeezed:
    """torchsde's scalar-noise shape (the Neural ODE notebook: g = zeros (B, H, 1)) as a diagonal field for the step loop."""

    def __init__(self, m):
        self.m = m

    def f(self, t, y):
        return self.m.f(t, y)

    def g(self, t, y):
        return self.m.g(t, y).squeeze(-1)


def main():
    torch.set_num_threads(1)
    _, interp, _ = G.load_reference('benchmark_classification')
    torchcde = sys.modules['torchcde']
    gen = torch.Generator().manual_seed(2024)
    out = {}
    for key, nb_file, cls, kind, H, layers, act, method in NOTEBOOKS:
        Func = field_class(nb_file, cls, torchcde)
        B, L, C = 6, 12, 2
        times = torch.linspace(0, 1, L)
        X = G.make_path(gen, B, L, C)
        X[:, :, 0] = times                       # channel 0 = time, as generate_data builds it
        coeffs = torch.cat(interp.natural_cubic_spline_coeffs(times, X), dim=-1)
        torch.manual_seed(5)
        m = Func(C, H, H, layers, activation=act)
        with torch.no_grad():
            for _, p in m.named_parameters():
                p.add_(0.1 * torch.randn(p.shape, generator=gen))
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        m.set_X(coeffs, times)
        y0 = 0.5 * torch.randn(B, H, generator=gen)
        if kind == 'gsde':
            y0 = y0.abs() + 0.1                  # the notebook starts GSDE from softplus(.) + 1e-4
        ts, dt = times, 0.05
        N = G.count_steps(ts, dt)
        hs, curr = [], ts[0]
        for out_t in ts[1:]:
            while curr < out_t:
                nxt = min(curr + dt, ts[-1])
                hs.append(nxt - curr)
                curr = nxt
        hv = torch.stack(hs).view(N, 1, 1)
        dW = torch.randn(N, B, H, generator=gen) * hv.sqrt()
        dU = hv * (0.5 * dW + (hv / 12).sqrt() * torch.randn(N, B, H, generator=gen)) if method == 'srk' else None
        probe_t = torch.tensor([0.0, 0.37, 1.0])
        with torch.no_grad():
            fs = torch.stack([m.f(t, y0) for t in probe_t])
            gs = torch.stack([m.g(t, y0) for t in probe_t])
            ys32, n32 = G.torch_step_grid_and_solve(_squeezed(m) if kind == 'ode' else m, y0, ts, dt, dW, method, dU)
        md = Func(C, H, H, layers, activation=act).double()
        md.load_state_dict({k: v.double() for k, v in sd.items()})
        md.set_X(coeffs.double(), times.double())
        with torch.no_grad():
            ys64, n64 = G.torch_step_grid_and_solve(G._F64Times(_squeezed(md) if kind == 'ode' else md), y0.double(), ts, dt, dW.double(),
                                                    method, None if dU is None else dU.double())
        assert n32 == N and n64 == N
        k = f'T1/{key}'
        out[f'{k}/meta'] = np.array([C, H, layers])
        out[f'{k}/kind'] = np.array(kind)
        out[f'{k}/activation'] = np.array(act)
        out[f'{k}/method'] = np.array(method)
        out[f'{k}/times'] = G.npy(times)
        out[f'{k}/coeffs'] = G.npy(coeffs)
        out[f'{k}/y0'] = G.npy(y0)
        out[f'{k}/dW'] = G.npy(dW)
        if dU is not None:
            out[f'{k}/dU'] = G.npy(dU)
        out[f'{k}/dt'] = np.float64(dt)
        out[f'{k}/probe_t'] = G.npy(probe_t)
        out[f'{k}/f'] = G.npy(fs)
        out[f'{k}/g'] = G.npy(gs)
        out[f'{k}/ys32'] = G.npy(ys32)
        out[f'{k}/ys64'] = G.npy(ys64)
        for kk, v in sd.items():
            out[f'{k}/param/{kk}'] = G.npy(v)
        print(f'T1 {key}: {cls} H={H} layers={layers} {act} {method} N={N} |ys|max={ys64.abs().max():.3f} '
              f'f32-f64 max={float((ys32.double() - ys64).abs().max()):.2e}')
    G.save('tutorial.npz', out)


if __name__ == '__main__':
    main()
