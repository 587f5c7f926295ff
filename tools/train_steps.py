#!/usr/bin/env python3
"""N training steps (sdeint forward + backward) of the K2 shape in one mode, for rocprofv3 passes.
usage: train_steps.py <recompute steps per chunk, 0 = saved activations> [steps] [io no B H C L method [hermite]]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stable_neural_sdes_amd as S
from tests.helpers import make_problem
dev = torch.device('cuda:0')
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 0
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
io, no, NL, B, H, C, L = 4, 17, 2, 1024, 128, 21, 101
method = 'euler'
if len(sys.argv) > 9:
    io, no, B, H, C, L = (int(v) for v in sys.argv[3:9]); method = sys.argv[9]
pr = make_problem(7, io, no, NL, B, H, C, L, nan_frac=0.0 if len(sys.argv) > 10 else 0.2, hermite=len(sys.argv) > 10)
m = S.Diffusion_model(C, H, H, NL, input_option=io, noise_option=no)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()}); m = m.to(dev)
times = torch.from_numpy(pr['times']).to(dev); m.set_X(torch.from_numpy(pr['coeffs']).to(dev), times)
y0 = torch.from_numpy(pr['y0']).to(dev); ts = times[[0, -1]]
for _ in range(steps):
    for p in m.parameters(): p.grad = None
    yy = y0.clone().requires_grad_(True)
    S.sdeint(m, yy, ts, method=method, dt=1.0, options={'seed': 1, 'recompute': chunk})[-1].square().mean().backward()
torch.cuda.synchronize()
