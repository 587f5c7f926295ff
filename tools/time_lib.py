#!/usr/bin/env python3
"""Kernel time of the K2 bench solve for the library named by SNSDE_LIB (development: A/B timing of kernel variants).
usage: SNSDE_LIB=... time_lib.py [kernel] [batch]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stable_neural_sdes_amd as S
import bench
kernel = sys.argv[1] if len(sys.argv) > 1 else 'auto'
if len(sys.argv) > 2:
    bench.B = int(sys.argv[2])
dev = torch.device('cuda:0')
pr, params, flat, coeffs, y0 = bench.build_inputs(dev, 0)
model = S.engine.model_struct(bench.C, bench.H, bench.H, bench.NL, bench.IO, bench.NO)
grid = S.engine.step_grid(np.array([0.0, 100.0], np.float32), 1.0, pr['times'], dev)
call = S.engine.SolveCall(model, flat, coeffs, grid, y0, seed=1, kernel=kernel)
for _ in range(5):
    call.launch()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
for a, b in ev:
    a.record(); call.launch(reuse_prepared=True); b.record()
torch.cuda.synchronize()
t = np.array([a.elapsed_time(b) for a, b in ev])
print(f"{os.path.basename(os.environ.get('SNSDE_LIB', 'libsnsde.so')):24s} kernel={kernel} B={bench.B} median {np.median(t)*1e3:7.1f} us  min {t.min()*1e3:7.1f} us")
