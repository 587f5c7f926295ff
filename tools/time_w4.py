#!/usr/bin/env python3
"""K4-shaped Euler solve (3,18) B x H=64, C=69, 71 steps: wave-owns-rows kernel vs the tile kernels, kernel-only (REUSE_PREPARED) and whole call."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stable_neural_sdes_amd as S
import bench
dev = torch.device('cuda:0')
stream = torch.cuda.current_stream(dev)
method = 'euler'
argv = sys.argv[1:]
if argv and argv[0] in ('euler', 'srk', 'milstein'):
    method, argv = argv[0], argv[1:]
rows_list = [int(x) for x in argv] or [256, 512, 1024, 2048, 4096, 8192]
print('method', method)
for rows in rows_list:
    sde, times, y0 = bench._module(dev, 3, 18, rows, 64, 69, 72, 77)
    model, layout, numel = S.engine.recognise(sde)
    flat = S.engine.flatten_params(sde, layout, numel, dev)
    for outputs in ('knots', 'ends'):
        ts = (times if outputs == 'knots' else times[[0, -1]]).cpu().numpy()
        grid = S.engine.step_grid(ts, 1.0, times.cpu().numpy(), dev)
        line = f'rows {rows:5d} outputs {outputs:5s}:'
        for kernel in (('w4', 'mfma4', 'mfma16') if method == 'euler' else ('w4', 'mfma4')):
            for training in (False, True):
                call = S.engine.SolveCall(model, flat, sde.coeffs, grid, y0, method=method, seed=5, kernel=kernel, save_traj=training, save_act=training)
                call.launch(stream)
                t = bench.event_times_ms(lambda: call.launch(stream, reuse_prepared=True), stream, 20, 3)
                line += f'  {kernel}{"/train" if training else ""} {np.median(t) * 1e3:7.1f} us'
        fl = rows * 71 * 41472
        print(line, flush=True)
