#!/usr/bin/env python3
"""Phase timeline of the wave-group SRK adjoint (-DW4_TRACE build: python stable-neural-sdes_amd/build.py w4trace; run with
SNSDE_LIB=stable-neural-sdes_amd/libsnsde_w4trace.so): cycles per step between / at the five barriers of a step, for the four
roles of tile 0 (drift wave, net wave, drift-gradient wave, net-gradient wave)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stable_neural_sdes_amd as S
import bench
dev = torch.device('cuda:0')
for method, rows in (('srk', 2048), ('euler', 2048), ('euler', 4096)):
    sde, times, y0 = bench._module(dev, 3, 18, rows, 64, 69, 72, 77)
    model, layout, numel = S.engine.recognise(sde)
    flat = S.engine.flatten_params(sde, layout, numel, dev)
    ts = times.cpu().numpy()
    grid = S.engine.step_grid(ts, 1.0, ts, dev)
    call = S.engine.SolveCall(model, flat, sde.coeffs, grid, y0, method=method, seed=5, kernel='w4', save_traj=True, save_dW=True, save_act=True)
    call.launch()
    g = torch.randn(len(ts), rows, 64, device=dev)
    for _ in range(2):
        adj = S.engine.solve_backward(call, g, save_delta=False, adj0_only=False)
        adj = adj[0] if isinstance(adj, tuple) else adj
    torch.cuda.synchronize()
    t = adj[2].flatten()[:256].cpu().numpy().reshape(4, 64)[:, :12] / grid.N
    print(f'{method} rows {rows}: cycles per step   work->B0 wait | ->B1 wait | ->B2 wait | ->B3 wait | ->B4 wait')
    for r, name in enumerate(('drift wave', 'net wave', 'drift-grad wave', 'net-grad wave')):
        print(f'  {name:16s}: ' + ' | '.join(f'{t[r, 2 * i]:6.0f} {t[r, 2 * i + 1]:6.0f}' for i in range(5 if method == 'srk' else 2)) + f'   total {t[r, :10].sum():.0f}')
