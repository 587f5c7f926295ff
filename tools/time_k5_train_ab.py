#!/usr/bin/env python3
"""K5 training step at the engine level (training-mode forward + adjoint + native weight gradients, HIP events, no host work in between):
the two-tile kernels (forward snsde_m4s2_kernel, adjoint snsde_m4s2_reverse_kernel) against the fully streamed sixteen-wave ones
(SNSDE_FLAG_STREAM_ALL).  usage: python tools/time_k5_train_ab.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stable_neural_sdes_amd as S
from tests.helpers import make_problem
dev = torch.device('cuda:0')
H, C, L = 256, 14, 50
print('K5: (4,17) NL=2 H=256 C=14, 49 Milstein steps, 50 outputs, Philox; ms (HIP-event medians): forward(train) | adjoint + weight gradients | step')
for B in (128, 1024):
    pr = make_problem(7, 4, 17, 2, B, H, C, L, nan_frac=0.2)
    model = S.engine.model_struct(C, H, H, 2, 4, 17)
    layout, numel = S._lib.param_layout(model)
    flat = torch.cat([torch.from_numpy(np.asarray(pr['params'][k], np.float32).reshape(-1)) for k, _, _ in layout]).to(dev)
    grid = S.engine.step_grid(pr['times'], 1.0, pr['times'], dev)
    coeffs = torch.from_numpy(pr['coeffs']).to(dev); y0 = torch.from_numpy(pr['y0']).to(dev)
    res = {}
    for all_ in (True, False):
        call = S.engine.SolveCall(model, flat, coeffs, grid, y0, method='milstein', seed=3, kernel='mfma4', stream_all=all_,
                                  save_traj=True, save_dW=False, save_act=True)
        gy = torch.randn(call.ys.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
        st = torch.cuda.current_stream()
        tf, tb = [], []
        for it in range(25):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record(st); call.launch(); e[1].record(st)
            adj, grad = S.engine.backward_with_gradients(call, gy, adj0_only=True)
            e[2].record(st); torch.cuda.synchronize()
            if it >= 5:
                tf.append(e[0].elapsed_time(e[1])); tb.append(e[1].elapsed_time(e[2]))
        res[all_] = (float(np.median(tf)), float(np.median(tb)), grad.clone())
    same = torch.equal(res[True][2], res[False][2])
    a, b = res[True], res[False]
    print(f'B={B:5d}: streamed {a[0]:.3f} | {a[1]:.3f} | {a[0] + a[1]:.3f}   two-tile {b[0]:.3f} | {b[1]:.3f} | {b[0] + b[1]:.3f}   '
          f'({(b[0] + b[1]) / (a[0] + a[1]) - 1:+.0%}), gradients bit-identical={same}', flush=True)
