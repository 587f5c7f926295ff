#!/usr/bin/env python3
"""Random configurations at H = 256 on 4-row tiles: the two-tile kernels (forward + adjoint + gradients) against the fully streamed
sixteen-wave ones, bit for bit.  usage: python tools/fuzz_h256.py [cases]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stable_neural_sdes_amd as S
from tests.helpers import make_problem, draw_dW, param_spec
dev = torch.device('cuda:0')
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
for ci in range(ncase):
    io = int(rng.integers(1, 7)); no = int(rng.choice([0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 11, 12, 13, 16, 17]))
    NL = int(rng.integers(1, 4)); C = int(rng.choice([1, 3, 14, 21])); B = int(rng.integers(1, 70)); L = int(rng.integers(3, 12))
    method = str(rng.choice(['euler', 'milstein']))
    pr = make_problem(9000 + ci, io, no, NL, B, 256, C, L, weight_scale=0.7)
    nt = int(rng.integers(2, 5))
    ts = np.sort(rng.choice(np.linspace(0, L - 1, 4 * (L - 1) + 1), size=nt, replace=False)).astype(np.float32)
    dt = float(rng.choice([1.0, 0.5, 0.3]))
    model = S.engine.model_struct(C, 256, 256, NL, io, no)
    flat = torch.from_numpy(np.concatenate([np.asarray(pr['params'][n], np.float32).reshape(-1) for n, _ in param_spec(io, no, NL, C, 256)])).to(dev)
    grid = S.engine.step_grid(ts, dt, pr['times'], dev)
    supplied = torch.from_numpy(draw_dW(9000 + ci, ts, dt, B, 256)).to(dev) if rng.random() < 0.5 else None
    ro = torch.from_numpy(rng.integers(0, nt, size=B).astype(np.int32)).to(dev) if rng.random() < 0.3 else None
    outs = []
    try:
        for all_ in (True, False):
            call = S.engine.SolveCall(model, flat, torch.from_numpy(pr['coeffs']).to(dev), grid, torch.from_numpy(pr['y0']).to(dev), dW=supplied,
                                      method=method, seed=11 + ci, kernel='mfma4', stream_all=all_, save_traj=True, save_dW=supplied is None, save_act=True,
                                      row_out=ro)
            ys = call.launch().clone()
            gy = torch.ones_like(ys) * 0.37 if not outs else outs[0][-1]
            adj, delta = S.engine.solve_backward(call, gy, save_delta=True)
            grad = S.engine.param_gradients(call, adj, delta)
            outs.append((ys, call.traj.clone(), call.act_save.clone(), adj.clone(), delta.clone(), grad.clone(), gy))
    except S._lib.SnsdeError as e:
        print(ci, (io, no, NL, C, B, L, method), 'skipped:', e)
        continue
    same = all(torch.equal(x, y) or (torch.isnan(x) == torch.isnan(y)).all() and torch.equal(torch.nan_to_num(x), torch.nan_to_num(y)) for x, y in zip(outs[0][:6], outs[1][:6]))
    if not same:
        bad += 1
        print('MISMATCH', ci, (io, no, NL, C, B, L, method, ts.tolist(), dt, supplied is not None, ro is not None),
              [bool(torch.equal(x, y)) for x, y in zip(outs[0][:6], outs[1][:6])])
print(f'{ncase} cases, {bad} mismatches')
