#!/usr/bin/env python3
"""K2 training step at the engine level (training-mode forward | adjoint + weight gradients), HIP-event medians.  SNSDE_LIB selects a
variant library (build.py variant).  usage: [SNSDE_LIB=...] python tools/time_k2_train_engine.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stable_neural_sdes_amd as S
import bench
dev = torch.device('cuda:0')
pr, params, flat, coeffs, y0 = bench.build_inputs(dev, 0)
model = S.engine.model_struct(bench.C, bench.H, bench.H, bench.NL, bench.IO, bench.NO)
grid = S.engine.step_grid(np.array([0.0, 100.0], np.float32), 1.0, pr['times'], dev)
call = S.engine.SolveCall(model, flat, coeffs, grid, y0, seed=1, save_traj=True, save_act=True)
gy = torch.randn(call.ys.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
st = torch.cuda.current_stream()
tf, tb = [], []
for it in range(40):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record(st); call.launch(); e[1].record(st)
    adj, grad = S.engine.backward_with_gradients(call, gy, adj0_only=True)
    e[2].record(st); torch.cuda.synchronize()
    if it >= 8:
        tf.append(e[0].elapsed_time(e[1])); tb.append(e[1].elapsed_time(e[2]))
print(f"{os.path.basename(os.environ.get('SNSDE_LIB', 'libsnsde.so')):22s} K2 forward(train) {np.median(tf)*1e3:7.1f} us | adjoint + weight gradients {np.median(tb)*1e3:7.1f} us | "
      f"step {(np.median(tf)+np.median(tb))*1e3:7.1f} us | grad sum {float(grad.double().sum()):.6f}")
