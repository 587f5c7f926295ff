"""SRK at large batch: 4-row vs 16-row tiles (K3-shaped GSDE (6,17), H = 128, 100 steps), beside Euler on the same tiles."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import stable_neural_sdes_amd as S
from tests.helpers import make_problem, param_spec
dev = torch.device('cuda:0')
io, no, NL, H, C, L = 6, 17, 2, 128, 21, 101
for B in (1024, 4096, 16384):
    pr = make_problem(3, io, no, NL, B, H, C, L, nan_frac=0.0)
    model = S.engine.model_struct(C, H, H, NL, io, no)
    flat = torch.from_numpy(np.concatenate([np.asarray(pr['params'][n], np.float32).reshape(-1) for n, _ in param_spec(io, no, NL, C, H)])).to(dev)
    coeffs = torch.from_numpy(pr['coeffs']).to(dev); y0 = torch.from_numpy(pr['y0']).to(dev)
    grid = S.engine.step_grid(np.array([0.0, L - 1.0], np.float32), 1.0, pr['times'], dev)
    out = []
    for method in ('euler', 'srk'):
        for kernel in ('mfma4', 'mfma16', 'auto'):
            call = S.engine.SolveCall(model, flat, coeffs, grid, y0, method=method, kernel=kernel, seed=3)
            for _ in range(3): call.launch()
            ts_ = []
            for _ in range(9):
                torch.cuda.synchronize(); t = time.perf_counter(); call.launch(); torch.cuda.synchronize(); ts_.append(time.perf_counter() - t)
            out.append(f'{method}/{kernel} {np.median(ts_) * 1e3:.3f} ms ({S.engine.forward_path(model, B, L, grid.N, method=method, kernel=kernel)})')
    print(f'B={B:6d}: ' + ' | '.join(out))
