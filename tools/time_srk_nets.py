"""SRK / Euler training step with a diffusion net on [tau, y] (torch_ists `neuralsde_1_18`, benchmark `naivesde`): fused
forward + generic adjoint kernel + batched parameter pass against autograd through the tensor-op loop."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import stable_neural_sdes_amd as S
from tests.helpers import make_problem
dev = torch.device('cuda:0')
for (io, no, B, H, C, L, method) in ((1, 18, 512, 64, 5, 50, 'srk'), (1, 18, 1024, 128, 21, 50, 'srk'), (4, 18, 2048, 64, 69, 72, 'euler')):
    pr = make_problem(7, io, no, 2, B, H, C, L, nan_frac=0.2)
    m = S.Diffusion_model(C, H, H, 2, input_option=io, noise_option=no)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()}); m = m.to(dev)
    times = torch.from_numpy(pr['times']).to(dev); m.set_X(torch.from_numpy(pr['coeffs']).to(dev), times)
    y0 = torch.from_numpy(pr['y0']).to(dev)
    out = []
    for opts in ({'seed': 1, 'strict': True}, {'seed': 1, 'backend': 'torch'}):
        def fb():
            m.zero_grad(set_to_none=True)
            yy = y0.clone().requires_grad_(True)
            S.sdeint(m, yy, times, method=method, dt=1.0, options=opts)[-1].square().mean().backward()
        n = 5 if 'strict' in opts else 2
        for _ in range(2): fb()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): fb()
        torch.cuda.synchronize(); out.append((time.perf_counter() - t) / n * 1e3)
    print(f'({io},{no}) {method} B={B} H={H} C={C} N={L - 1}: fused fwd+bwd {out[0]:.2f} ms, tensor-op loop + autograd {out[1]:.1f} ms')
