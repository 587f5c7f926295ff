#!/usr/bin/env python3
"""H = 256 forward solve on 4-row tiles: the two-tiles-per-wave kernel with a quarter of every layer resident (snsde_m4s2_kernel.h)
against the fully streamed one (snsde_m4s_kernel.h, SNSDE_FLAG_STREAM_ALL) - kernel-only HIP-event medians, and bit-identity.
usage: python tools/time_k5_ab.py [C]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stable_neural_sdes_amd as S
from tests.helpers import make_problem
dev = torch.device('cuda:0')
H, L = 256, 50
C = int(sys.argv[1]) if len(sys.argv) > 1 else 14
print('K5 shape: (4,17) NL=2 H=256 C=%d, 49 steps, Philox; ms per solve (kernel only), streamed-all -> two-tile resident' % C)
for B in (128, 256, 512, 1024, 2048):
    for method in ('euler', 'milstein'):
        pr = make_problem(7, 4, 17, 2, B, H, C, L, nan_frac=0.2)
        model = S.engine.model_struct(C, H, H, 2, 4, 17)
        layout, numel = S._lib.param_layout(model)
        flat = torch.cat([torch.from_numpy(np.asarray(pr['params'][n], np.float32).reshape(-1)) for n, _, _ in layout]).to(dev)
        grid = S.engine.step_grid(pr['times'], 1.0, pr['times'], dev)
        coeffs = torch.from_numpy(pr['coeffs']).to(dev); y0 = torch.from_numpy(pr['y0']).to(dev)
        row = f'B={B:5d} {method:8s}'
        for train in (False, True):
            res = {}
            for all_ in (True, False):
                call = S.engine.SolveCall(model, flat, coeffs, grid, y0, method=method, seed=3, kernel='mfma4', stream_all=all_,
                                          save_traj=train, save_dW=train, save_act=train)
                call.launch()
                st = torch.cuda.current_stream()
                for _ in range(3): call.launch(reuse_prepared=True)
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(25)]
                for a, b in ev:
                    a.record(st); call.launch(reuse_prepared=True); b.record(st)
                torch.cuda.synchronize()
                res[all_] = (float(np.median([a.elapsed_time(b) for a, b in ev])), call.ys.clone(),
                             None if not train else (call.traj.clone(), call.act_save.clone(), call.dW_out.clone()))
            same = torch.equal(res[True][1], res[False][1]) and (not train or all(torch.equal(x, y) for x, y in zip(res[True][2], res[False][2])))
            row += f' | {"train" if train else "infer"} {res[True][0]:6.3f} -> {res[False][0]:6.3f} ({res[False][0] / res[True][0] - 1:+.0%}) bit-identical={same}'
        print(row, flush=True)
