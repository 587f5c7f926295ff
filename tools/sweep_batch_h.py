#!/usr/bin/env python3
"""Solve-kernel time vs batch size for both MFMA tile flavours and `auto` at other hidden sizes: sweep_batch_h.py H [io no C method]."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stable_neural_sdes_amd as S
from tests.helpers import make_problem, param_spec
dev = torch.device('cuda:0')
H = int(sys.argv[1]); IO = int(sys.argv[2]) if len(sys.argv) > 2 else 4; NO = int(sys.argv[3]) if len(sys.argv) > 3 else 17
C = int(sys.argv[4]) if len(sys.argv) > 4 else 21
METHOD = sys.argv[5] if len(sys.argv) > 5 else 'euler'
NL, L = 2, 101
pr = make_problem(1234, IO, NO, NL, 1024, H, C, L, nan_frac=0.3)
flat = torch.from_numpy(np.concatenate([pr['params'][n].reshape(-1) for n, _ in param_spec(IO, NO, NL, C, H)])).to(dev)
model = S.engine.model_struct(C, H, H, NL, IO, NO)
grid = S.engine.step_grid(np.array([0.0, 100.0], np.float32), 1.0, pr['times'], dev)
for B in (512, 1024, 2048, 3072, 4096, 6144, 8192, 16384):
    reps = max(1, B // 1024)
    coeffs = torch.from_numpy(np.tile(pr['coeffs'], (reps, 1, 1))[:B]).to(dev) if B >= 1024 else torch.from_numpy(pr['coeffs'][:B]).to(dev)
    y0 = torch.from_numpy(np.tile(pr['y0'], (reps, 1))[:B]).to(dev) if B >= 1024 else torch.from_numpy(pr['y0'][:B]).to(dev)
    line = f'H={H} ({IO},{NO}) {METHOD} B={B:6d}'
    for kern in ('mfma4', 'mfma16', 'auto'):
        try:
            call = S.engine.SolveCall(model, flat, coeffs, grid, y0, seed=1, kernel=kern, method=METHOD)
            for _ in range(3): call.launch()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
            for a, b in ev:
                a.record(); call.launch(reuse_prepared=True); b.record()
            torch.cuda.synchronize()
            ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
            line += f'  {kern}: {ms:7.3f} ms'
        except Exception as e:
            line += f'  {kern}: n/a'
    print(line)
