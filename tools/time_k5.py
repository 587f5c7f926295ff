#!/usr/bin/env python3
"""Kernel-only timings (HIP events, prepared workspace reused) of the H=256 forward solve per tile flavour.
usage: python tools/time_k5.py [H] [C]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stable_neural_sdes_amd as S
from tests.helpers import make_problem
dev = torch.device('cuda:0')
H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
C = int(sys.argv[2]) if len(sys.argv) > 2 else 14
L = 50
for B in (128, 256, 512, 1024, 2048, 4096):
    for method in ('euler', 'milstein'):
        pr = make_problem(7, 4, 17, 2, B, H, C, L, nan_frac=0.2)
        model = S.engine.model_struct(C, H, H, 2, 4, 17)
        layout, numel = S._lib.param_layout(model)
        flat = torch.cat([torch.from_numpy(np.asarray(pr['params'][n], np.float32).reshape(-1)) for n, _, _ in layout]).to(dev)
        grid = S.engine.step_grid(pr['times'], 1.0, pr['times'], dev)
        coeffs = torch.from_numpy(pr['coeffs']).to(dev); y0 = torch.from_numpy(pr['y0']).to(dev)
        row = f'B={B:5d} {method:8s}'
        for kern in ('mfma16', 'mfma4', 'auto'):
            for train in (False, True):
                try:
                    call = S.engine.SolveCall(model, flat, coeffs, grid, y0, method=method, seed=3, kernel=kern,
                                              save_traj=train, save_dW=train, save_act=train)
                    call.launch()
                    st = torch.cuda.current_stream()
                    for _ in range(3): call.launch(reuse_prepared=True)
                    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(15)]
                    for a, b in ev:
                        a.record(st); call.launch(reuse_prepared=True); b.record(st)
                    torch.cuda.synchronize()
                    t = float(np.median([a.elapsed_time(b) for a, b in ev]))
                    row += f' | {kern}{"+save" if train else ""} {t:6.3f}'
                except S._lib.SnsdeError as e:
                    row += f' | {kern}{"+save" if train else ""}  n/a '
        print(row, flush=True)
