#!/usr/bin/env python3
"""K2 forward solve: the eight-wave lean kernel against the four-wave two-tiles-per-wave kernel (SNSDE_FLAG_TWO_TILE), kernel-only HIP-event
medians and bit-identity.  usage: python tools/time_k2_two_tile.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stable_neural_sdes_amd as S
import bench
dev = torch.device('cuda:0')
for rows in (256, 1024, 2048):
    pr, params, flat, coeffs, y0 = bench.build_inputs(dev, 0, b=rows)
    model = S.engine.model_struct(bench.C, bench.H, bench.H, bench.NL, bench.IO, bench.NO)
    grid = S.engine.step_grid(np.array([0.0, 100.0], np.float32), 1.0, pr['times'], dev)
    for train in (False, True):
        res = {}
        for two in (False, True):
            call = S.engine.SolveCall(model, flat, coeffs, grid, y0, seed=1, kernel='mfma4', two_tile=two, save_traj=train, save_dW=train, save_act=train)
            for _ in range(5):
                call.launch()
            torch.cuda.synchronize()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
            for a, b in ev:
                a.record(); call.launch(reuse_prepared=True); b.record()
            torch.cuda.synchronize()
            t = np.array([a.elapsed_time(b) for a, b in ev])
            res[two] = (float(np.median(t)) * 1e3, float(t.min()) * 1e3, call.ys.clone(), None if not train else (call.traj.clone(), call.act_save.clone()))
        same = torch.equal(res[False][2], res[True][2]) and (not train or all(torch.equal(x, y) for x, y in zip(res[False][3], res[True][3])))
        print(f'B={rows:5d} {"train" if train else "infer"}: lean (8 waves) median {res[False][0]:7.1f} min {res[False][1]:7.1f} us | two-tile (4 waves) median '
              f'{res[True][0]:7.1f} min {res[True][1]:7.1f} us | bit-identical={same}', flush=True)
