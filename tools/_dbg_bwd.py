import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import stable_neural_sdes_amd as S
from tests.helpers import make_problem, draw_dW, param_spec
from oracle import sde_oracle as O
dev = torch.device('cuda:0')
io, no, NL, B, H, C, L, ts, dt = (1, 18, 2, 9, 16, 3, 8, [0, 7], 0.5)
pr = make_problem(1, io, no, NL, B, H, C, L)
dW = draw_dW(1, ts, dt, B, H)
t0, t1, *_ = O.step_grid(np.asarray(ts, np.float32), dt)
hh = (t1 - t0).astype(np.float32)[:, None, None]
dU = (hh * (0.5 * dW + np.sqrt(hh / 12) * np.random.default_rng(3).standard_normal(dW.shape).astype(np.float32))).astype(np.float32)
model = S.engine.model_struct(C, H, H, NL, io, no)
flat = torch.from_numpy(np.concatenate([np.asarray(pr['params'][n], np.float32).reshape(-1) for n, _ in param_spec(io, no, NL, C, H)])).to(dev)
grid = S.engine.step_grid(np.asarray(ts, np.float32), dt, pr['times'], dev)
call = S.engine.SolveCall(model, flat, torch.from_numpy(pr['coeffs']).to(dev), grid, torch.from_numpy(pr['y0']).to(dev),
                          dW=torch.from_numpy(dW).to(dev), dU=torch.from_numpy(dU).to(dev), method='srk', kernel='mfma4',
                          save_traj=True, save_dW=True, save_act=True)
print('mode', S.engine.backward_supported(call), 'act', call.act_save.shape, 'stage', call.stage_save.shape, flush=True)
ys = call.launch(); torch.cuda.synchronize(); print('forward ok', float(ys.abs().max()), flush=True)
g = torch.ones_like(ys)
adj, delta = S.engine.solve_backward(call, g, save_delta=True); torch.cuda.synchronize(); print('adjoint ok', float(adj.abs().max()), flush=True)
grad = S.engine.param_gradients(call, adj, delta); torch.cuda.synchronize(); print('wgrad ok', float(grad.abs().max()), flush=True)
