#!/usr/bin/env python3
"""Per-phase cycle timeline of the lean M4 kernel at the K2 bench shape: mean s_memtime ticks per step between consecutive
stamps, per wave of workgroup 0.  Needs the trace build (python stable-neural-sdes_amd/build.py leantrace) named by SNSDE_LIB:
    SNSDE_LIB=stable-neural-sdes_amd/libsnsde_leantrace.so python tools/lean_trace.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stable_neural_sdes_amd as S
import bench
dev = torch.device('cuda:0')
pr, params, flat, coeffs, y0 = bench.build_inputs(dev, 0)
model = S.engine.model_struct(bench.C, bench.H, bench.H, bench.NL, bench.IO, bench.NO)
grid = S.engine.step_grid(np.array([0.0, 100.0], np.float32), 1.0, pr['times'], dev)
train = len(sys.argv) > 1 and sys.argv[1] == 'train'      # training-mode instantiation (act_save / traj stores); the trace build of it spills two registers
call = S.engine.SolveCall(model, flat, coeffs, grid, y0, seed=1, kernel='mfma4', save_traj=train, save_act=train)
for _ in range(3):
    call.launch()
torch.cuda.synchronize()
t = call.ys[-1].reshape(-1)[:8 * 16].cpu().numpy().reshape(8, 16)[:, :10] / grid.N
names = ['barrier C -> top', 'reads + top filler', 'L1 mfma', 'L1 epilogue', 'barrier A', 'reads + prep', 'L2 mfma + epilogue',
         'barrier B', 'reads + L3 mfma', 'vm wait + update']
print('s_memtime ticks per step (mean over steps; the counter runs at about the shader clock - profiles/r03_ubench.txt: 9 ticks per 8-cycle MFMA slot), waves 0..7:')
for i, nme in enumerate(names):
    print(f'{nme:22s}', ' '.join(f'{v:7.0f}' for v in t[:, i]))
print(f'{"total":22s}', ' '.join(f'{v:7.0f}' for v in t.sum(1)))
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
for a, b in ev:
    a.record(); call.launch(reuse_prepared=True); b.record()
torch.cuda.synchronize()
print('kernel us (trace build%s):' % (', training mode' if train else ''), np.median([a.elapsed_time(b) for a, b in ev]) * 1e3)
