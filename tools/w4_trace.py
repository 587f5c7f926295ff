#!/usr/bin/env python3
"""Phase timeline of the wave-owns-rows kernel (-DW4_TRACE build: python stable-neural-sdes_amd/build.py w4trace; run with
SNSDE_LIB=stable-neural-sdes_amd/libsnsde_w4trace.so): cycles per step and phase for the drift wave and the net wave of workgroup 0."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stable_neural_sdes_amd as S
import bench
dev = torch.device('cuda:0')
for rows in (1024, 2048, 4096):
    sde, times, y0 = bench._module(dev, 3, 18, rows, 64, 69, 72, 77)
    model, layout, numel = S.engine.recognise(sde)
    flat = S.engine.flatten_params(sde, layout, numel, dev)
    for outputs in ('knots', 'ends'):
        ts = (times if outputs == 'knots' else times[[0, -1]]).cpu().numpy()
        grid = S.engine.step_grid(ts, 1.0, times.cpu().numpy(), dev)
        call = S.engine.SolveCall(model, flat, sde.coeffs, grid, y0, method='euler', seed=5, kernel='w4', save_dW=True)
        call.launch(); call.launch()
        torch.cuda.synchronize()
        t = call.dW_out.flatten()[:128].cpu().numpy() / grid.N
        names_d = ['top/row', 'layer 1', 'hidden', 'out gemm', 'tanh', 'xchg+barrier', 'update+out']
        names_n = ['top/row', 'philox', 'net l1', 'net l2', 'g', 'xchg+barrier', 'update']
        print(f'rows {rows} outputs {outputs}: cycles per step')
        print('  drift wave: ' + ', '.join(f'{n} {v:.0f}' for n, v in zip(names_d, t[:7])) + f'  | total {t[:7].sum():.0f}')
        print('  net wave  : ' + ', '.join(f'{n} {v:.0f}' for n, v in zip(names_n, t[64:71])) + f'  | total {t[64:71].sum():.0f}')
