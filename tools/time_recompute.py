#!/usr/bin/env python3
"""Training step (sdeint forward + backward) time and peak device memory: saved-activation mode vs recompute mode
(options={'recompute': steps per chunk}) on the K2 and K5 shapes."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stable_neural_sdes_amd as S
from tests.helpers import make_problem
dev = torch.device('cuda:0')
CFG = [('K2 LNSDE euler  B=1024 H=128 N=100 ts=[0,100]', 4, 17, 2, 1024, 128, 21, 101, 'euler', False),
       ('K5 LNSDE milst. B=1024 H=256 N=49 ts=times   ', 4, 17, 2, 1024, 256, 14, 50, 'milstein', True),
       ('K3 GSDE  euler  B=4096 H=128 N=200 ts=[0,200]', 6, 17, 2, 4096, 128, 21, 201, 'euler', False)]
for name, io, no, NL, B, H, C, L, method, ts_all in CFG:
    pr = make_problem(7, io, no, NL, B, H, C, L, nan_frac=0.2)
    m = S.Diffusion_model(C, H, H, NL, input_option=io, noise_option=no)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()})
    m = m.to(dev)
    times = torch.from_numpy(pr['times']).to(dev)
    m.set_X(torch.from_numpy(pr['coeffs']).to(dev), times)
    ts = times if ts_all else times[[0, -1]]
    y0 = torch.from_numpy(pr['y0']).to(dev)
    for chunk in (0, 50, 25, 10, 5):
        def step():
            for p in m.parameters(): p.grad = None
            yy = y0.clone().requires_grad_(True)
            S.sdeint(m, yy, ts, method=method, dt=1.0, options={'seed': 1, 'recompute': chunk})[-1].square().mean().backward()
        for _ in range(3): step()
        torch.cuda.synchronize(); base = torch.cuda.memory_allocated(); torch.cuda.reset_peak_memory_stats()
        t = []
        for _ in range(10):
            torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); t.append((time.perf_counter() - t0) * 1e3)
        peak = (torch.cuda.max_memory_allocated() - base) / 2**20
        print(f'{name} | {"saved activations" if chunk == 0 else f"recompute, {chunk:3d} steps/chunk"} | fwd+bwd {np.median(t):7.3f} ms | peak extra memory {peak:8.1f} MiB', flush=True)
