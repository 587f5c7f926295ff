#!/usr/bin/env python3
"""Hidden sizes without an MFMA instantiation: zero-padded MFMA solve (default) vs the generic kernels."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stable_neural_sdes_amd as S
from tests.helpers import make_problem
dev = torch.device('cuda:0')
for B, H, HH, C, L, io, no in ((1024, 96, 96, 21, 101, 4, 17), (1024, 100, 100, 21, 101, 6, 17), (2048, 48, 48, 69, 72, 4, 17), (256, 200, 160, 5, 51, 3, 18)):
    pr = make_problem(7, io, no, 2, B, H, C, L, nan_frac=0.2)
    torch.manual_seed(1)
    m = S.Diffusion_model(C, H, HH, 2, input_option=io, noise_option=no).to(dev)
    times = torch.from_numpy(pr['times']).to(dev); m.set_X(torch.from_numpy(pr['coeffs']).to(dev), times)
    y0 = torch.from_numpy(pr['y0']).to(dev); ts = times[[0, -1]]
    row = f'B={B} H={H} HH={HH} C={C} N={L - 1} ({io},{no}):'
    for kern in ('auto', 'generic'):
        def fwd():
            with torch.no_grad(): S.sdeint(m, y0, ts, method='euler', dt=1.0, options={'seed': 1, 'kernel': kern})
        def fb():
            yy = y0.clone().requires_grad_(True)
            S.sdeint(m, yy, ts, method='euler', dt=1.0, options={'seed': 1, 'kernel': kern})[-1].square().mean().backward()
        res = []
        for fn in (fwd, fb):
            for _ in range(3): fn()
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(5): fn()
            torch.cuda.synchronize(); res.append((time.perf_counter() - t) / 5 * 1e3)
        row += f'  {"zero-padded MFMA" if kern == "auto" else "generic kernels"}: fwd {res[0]:.3f} ms, fwd+bwd {res[1]:.3f} ms |'
    print(row)
