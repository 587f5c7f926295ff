#!/usr/bin/env python3
"""Training step (sdeint forward + backward) by tile flavour and batch size: sweep_train_flavour.py [H io no method]."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stable_neural_sdes_amd as S
from tests.helpers import make_problem
dev = torch.device('cuda:0')
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
io = int(sys.argv[2]) if len(sys.argv) > 2 else 4
no = int(sys.argv[3]) if len(sys.argv) > 3 else 17
method = sys.argv[4] if len(sys.argv) > 4 else 'euler'
NL, C, L = 2, 21, 101
for B in (1024, 2048, 3072, 4096, 6144, 8192):
    pr = make_problem(7, io, no, NL, B, H, C, L, nan_frac=0.2)
    m = S.Diffusion_model(C, H, H, NL, input_option=io, noise_option=no)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()}); m = m.to(dev)
    times = torch.from_numpy(pr['times']).to(dev); m.set_X(torch.from_numpy(pr['coeffs']).to(dev), times)
    y0 = torch.from_numpy(pr['y0']).to(dev); ts = times[[0, -1]]
    line = f'H={H} ({io},{no}) {method} B={B:5d}'
    for kern in ('mfma4', 'mfma16', 'auto'):
        def step():
            for p in m.parameters(): p.grad = None
            yy = y0.clone().requires_grad_(True)
            S.sdeint(m, yy, ts, method=method, dt=1.0, options={'seed': 1, 'kernel': kern})[-1].square().mean().backward()
        try:
            for _ in range(3): step()
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(8): step()
            torch.cuda.synchronize(); line += f'  {kern}: {(time.perf_counter() - t) / 8 * 1e3:7.3f} ms'
        except Exception as e:
            line += f'  {kern}: n/a ({type(e).__name__})'
    print(line)
