import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import stable_neural_sdes_amd as S
from tests.helpers import make_problem
dev = torch.device('cuda:0')
for B in (128, 1024):
    pr = make_problem(7, 4, 17, 2, B, 256, 14, 50, nan_frac=0.2)
    m = S.Diffusion_model(14, 256, 256, 2, input_option=4, noise_option=17)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()}); m = m.to(dev)
    times = torch.from_numpy(pr['times']).to(dev); m.set_X(torch.from_numpy(pr['coeffs']).to(dev), times)
    y0 = torch.from_numpy(pr['y0']).to(dev)
    for kern in ('auto', 'generic'):
        def fwd():
            with torch.no_grad(): S.sdeint(m, y0, times, method='srk', dt=1.0, options={'seed': 1, 'kernel': kern})
        def fb():
            yy = y0.clone().requires_grad_(True)
            S.sdeint(m, yy, times, method='srk', dt=1.0, options={'seed': 1, 'kernel': kern})[-1].square().mean().backward()
        res = []
        for fn in (fwd, fb):
            try:
                for _ in range(3): fn()
                torch.cuda.synchronize(); t = time.perf_counter()
                for _ in range(5): fn()
                torch.cuda.synchronize(); res.append((time.perf_counter() - t) / 5 * 1e3)
            except Exception as e:
                res.append(float('nan'))
        print(f'SRK H=256 B={B} N=49 kernel={kern}: fwd {res[0]:.3f} ms, fwd+bwd {res[1]:.3f} ms')
