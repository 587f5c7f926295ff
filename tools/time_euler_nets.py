"""Euler through a diffusion net: sepsis-wide control path (C = 69) and H = 128, forward and training step, 'auto' vs the generic family."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import stable_neural_sdes_amd as S
from tests.helpers import make_problem
dev = torch.device('cuda:0')
for (io, no, B, H, C, L) in ((4, 18, 2048, 64, 69, 72), (1, 18, 1024, 128, 21, 50), (3, 18, 2048, 64, 69, 72), (4, 19, 1024, 128, 69, 72)):
    pr = make_problem(7, io, no, 2, B, H, C, L, nan_frac=0.2)
    m = S.Diffusion_model(C, H, H, 2, input_option=io, noise_option=no)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()}); m = m.to(dev)
    times = torch.from_numpy(pr['times']).to(dev); m.set_X(torch.from_numpy(pr['coeffs']).to(dev), times)
    y0 = torch.from_numpy(pr['y0']).to(dev)
    model = S.engine.model_struct(C, H, H, 2, io, no)
    for kernel in ('auto', 'generic'):
        opts = {'seed': 1, 'strict': True, 'kernel': kernel}
        def fw():
            with torch.no_grad():
                S.sdeint(m, y0, times[[0, -1]], method='euler', dt=1.0, options=opts)
        def fb():
            m.zero_grad(set_to_none=True)
            yy = y0.clone().requires_grad_(True)
            S.sdeint(m, yy, times[[0, -1]], method='euler', dt=1.0, options=opts)[-1].square().mean().backward()
        res = []
        for fn in (fw, fb):
            for _ in range(3): fn()
            ts_ = []
            for _ in range(15):
                torch.cuda.synchronize(); t = time.perf_counter()
                fn()
                torch.cuda.synchronize(); ts_.append(time.perf_counter() - t)
            res.append(float(np.median(ts_)) * 1e3)
        print(f'({io},{no}) euler B={B} H={H} C={C} N={L - 1} kernel={kernel} path={S.engine.forward_path(model, B, L, L - 1, kernel=kernel)}: fwd {res[0]:.3f} ms, fwd+bwd {res[1]:.2f} ms')
