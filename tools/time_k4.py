import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import stable_neural_sdes_amd as S
from tests.helpers import make_problem
dev = torch.device('cuda:0')
B, H, C, L = 2048, 64, 69, 72
for io, no in ((3, 18), (3, 17), (3, 14), (3, 0), (1, 18)):
    pr = make_problem(7, io, no, 2, B, H, C, L, nan_frac=0.2)
    model = S.engine.model_struct(C, H, H, 2, io, no)
    layout, numel = S._lib.param_layout(model)
    flat = torch.cat([torch.from_numpy(np.asarray(pr['params'][n], np.float32).reshape(-1)) for n, _, _ in layout]).to(dev)
    grid = S.engine.step_grid(pr['times'][[0, -1]], 1.0, pr['times'], dev)
    coeffs = torch.from_numpy(pr['coeffs']).to(dev); y0 = torch.from_numpy(pr['y0']).to(dev)
    row = f'({io},{no}) B={B} H={H} N={grid.N}:'
    for kern in ('mfma4', 'mfma16', 'auto'):
        call = S.engine.SolveCall(model, flat, coeffs, grid, y0, method='euler', seed=3, kernel=kern)
        call.launch(); st = torch.cuda.current_stream()
        for _ in range(3): call.launch(reuse_prepared=True)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(15)]
        for a, b in ev:
            a.record(st); call.launch(reuse_prepared=True); b.record(st)
        torch.cuda.synchronize()
        row += f'  {kern} {np.median([a.elapsed_time(b) for a, b in ev]) * 1e3:6.1f} us ({S.engine.forward_path(model, B, L, grid.N, kernel=kern)})'
    print(row)
