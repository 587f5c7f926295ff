#!/usr/bin/env python3
"""n forward solves of the K5 shape (H = 256, 1024 rows, 49 Milstein steps, Philox) on the two-tile kernel and on the fully streamed
one, for rocprofv3 passes.  usage: k5_forward_steps.py [n] [rows]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stable_neural_sdes_amd as S
from tests.helpers import make_problem
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
H, C, L = 256, 14, 50
pr = make_problem(7, 4, 17, 2, B, H, C, L, nan_frac=0.2)
model = S.engine.model_struct(C, H, H, 2, 4, 17)
layout, numel = S._lib.param_layout(model)
flat = torch.cat([torch.from_numpy(np.asarray(pr['params'][k], np.float32).reshape(-1)) for k, _, _ in layout]).to(dev)
grid = S.engine.step_grid(pr['times'], 1.0, pr['times'], dev)
coeffs = torch.from_numpy(pr['coeffs']).to(dev); y0 = torch.from_numpy(pr['y0']).to(dev)
for all_ in (False, True):
    call = S.engine.SolveCall(model, flat, coeffs, grid, y0, method='milstein', seed=3, kernel='mfma4', stream_all=all_)
    for _ in range(n):
        call.launch()
torch.cuda.synchronize()
