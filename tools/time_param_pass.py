#!/usr/bin/env python3
"""Time the native parameter pass (snsde_param_gradients) of one configuration: time_param_pass.py io no B H C L method
(the split heuristic's knobs SNSDE_WGRAD_BIAS / SNSDE_WGRAD_WGS are read once per process: sweep them from the shell)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stable_neural_sdes_amd as S
from tests.helpers import make_problem, param_spec
dev = torch.device('cuda:0')
io, no, B, H, C, L = (int(v) for v in sys.argv[1:7]); method = sys.argv[7]
NL = 2
pr = make_problem(7, io, no, NL, B, H, C, L, nan_frac=0.2)
flat = torch.from_numpy(np.concatenate([pr['params'][n].reshape(-1) for n, _ in param_spec(io, no, NL, C, H)])).to(dev)
model = S.engine.model_struct(C, H, H, NL, io, no)
ts = pr['times'] if method != 'euler' else np.array([pr['times'][0], pr['times'][-1]], np.float32)
grid = S.engine.step_grid(ts, 1.0, pr['times'], dev)
call = S.engine.SolveCall(model, flat, torch.from_numpy(pr['coeffs']).to(dev), grid, torch.from_numpy(pr['y0']).to(dev), method=method, seed=1,
                          save_traj=True, save_dW=True, save_act=True)
call.launch()
g = torch.randn_like(call.ys)
adj, delta = S.engine.solve_backward(call, g, save_delta=True, adj0_only=S.engine.adj0_suffices(call))
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - s) / n * 1e3
print('param pass %.3f ms  (bias=%s wgs=%s)' % (t(lambda: S.engine.param_gradients(call, adj, delta)), os.environ.get('SNSDE_WGRAD_BIAS', '4'), os.environ.get('SNSDE_WGRAD_WGS', '512')))
