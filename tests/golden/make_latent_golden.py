#!/usr/bin/env python
"""Golden vectors of torch-ists' LatentSDE (build container only; needs /root/reference).

Run:  python tests/golden/make_latent_golden.py        (writes tests/golden/latent.npz)

The reference class is imported from where it lies (torch-ists/torch_ists/diff_module/NSDE/latent_sde.py) over stand-ins for
the two absent packages (`torchsde.SDEIto` = an nn.Module carrying sde_type / noise_type, `torchcde` as in make_golden.py).
The fixture holds data only: the module's state_dict (buffers included), augmented initial states, supplied increments,
f_aug / g_aug at a few (t, y) and fixed-step Euler / SRK trajectories of the AUGMENTED system driven by the reference's own
f_aug / g_aug (fp32 and fp64).  tests/latent_field.LatentField must load the state_dicts and reproduce the values; the
fused latent solve + batched KL quadrature (torchsde._sdeint_latent) is checked against the trajectories on the GPU."""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G      # noqa: E402

CASES = [  # key, input channels, hidden (latent + 1), hidden-hidden, layers, method, outputs on the step grid?
    ('h32_euler', 3, 32, 32, 2, 'euler', True),
    ('h17_srk', 2, 17, 24, 3, 'srk', True),
    ('h64_srk', 3, 64, 64, 1, 'srk', False),
    ('h33_euler', 2, 33, 40, 2, 'euler', False),
]


class _Aug:
    def __init__(self, m, f64=False):
        self.m, self.f64 = m, f64

    def f(self, t, y):
        return self.m.f_aug(t.double() if self.f64 else t, y)

    def g(self, t, y):
        return self.m.g_aug(t.double() if self.f64 else t, y)


def main():
    torch.set_num_threads(1)
    _, _, tsde = G.load_reference('benchmark_classification')

    class SDEIto(torch.nn.Module):
        def __init__(self, noise_type):
            super().__init__()
            self.noise_type, self.sde_type = noise_type, 'ito'
    tsde.SDEIto = SDEIto
    path = os.path.join(G.REF_ROOT, 'torch-ists', 'torch_ists', 'diff_module', 'NSDE', 'latent_sde.py')
    spec = importlib.util.spec_from_file_location('ref_latent_sde', path)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    gen = torch.Generator().manual_seed(77)
    out = {}
    for key, C, H, HH, NL, method, aligned in CASES:
        torch.manual_seed(9)
        m = ref.LatentSDE(C, H, HH, NL, theta=0.8, mu=0.1, sigma=0.5)
        with torch.no_grad():
            for _, p in m.named_parameters():
                p.add_(0.1 * torch.randn(p.shape, generator=gen))
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        B = 5
        ts = torch.linspace(0, 1, 9) if aligned else torch.tensor([0.0, 0.13, 0.4, 0.77, 1.0])
        dt = 0.125 if aligned else 0.06
        y0 = torch.cat([0.5 * torch.randn(B, H - 1, generator=gen), torch.zeros(B, 1)], dim=1)
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
            fa = torch.stack([m.f_aug(t, y0) for t in probe_t])
            ga = torch.stack([m.g_aug(t, y0) for t in probe_t])
            ys32, n32 = G.torch_step_grid_and_solve(_Aug(m), y0, ts, dt, dW, method, dU)
        md = ref.LatentSDE(C, H, HH, NL, theta=0.8, mu=0.1, sigma=0.5).double()
        md.load_state_dict({k: v.double() for k, v in sd.items()})
        with torch.no_grad():
            ys64, n64 = G.torch_step_grid_and_solve(_Aug(md, True), y0.double(), ts, dt, dW.double(), method,
                                                    None if dU is None else dU.double())
        assert n32 == N and n64 == N
        k = f'L1/{key}'
        out[f'{k}/meta'] = np.array([C, H, HH, NL])
        out[f'{k}/method'] = np.array(method)
        out[f'{k}/ts'] = G.npy(ts)
        out[f'{k}/dt'] = np.float64(dt)
        out[f'{k}/y0'] = G.npy(y0)
        out[f'{k}/dW'] = G.npy(dW)
        if dU is not None:
            out[f'{k}/dU'] = G.npy(dU)
        out[f'{k}/probe_t'] = G.npy(probe_t)
        out[f'{k}/f_aug'] = G.npy(fa)
        out[f'{k}/g_aug'] = G.npy(ga)
        out[f'{k}/ys32'] = G.npy(ys32)
        out[f'{k}/ys64'] = G.npy(ys64)
        for kk, v in sd.items():
            out[f'{k}/param/{kk}'] = G.npy(v)
        print(f'L1 {key}: LatentSDE C={C} H={H} HH={HH} NL={NL} {method} N={N} |ys|max={ys64.abs().max():.3f} '
              f'KL path max={ys64[-1, :, -1].max():.3f} f32-f64 max={float((ys32.double() - ys64).abs().max()):.2e}')
    G.save('latent.npz', out)


if __name__ == '__main__':
    main()
