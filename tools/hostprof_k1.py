import os, sys, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import stable_neural_sdes_amd as S
from tests.helpers import make_problem
dev = torch.device('cuda:0')
io, no, NL, B, H, C, L = 2, 16, 1, 256, 32, 2, 51
pr = make_problem(7, io, no, NL, B, H, C, L, nan_frac=0.2)
m = S.Diffusion_model(C, H, H, NL, input_option=io, noise_option=no)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()}); m = m.to(dev)
times = torch.from_numpy(pr['times']).to(dev); m.set_X(torch.from_numpy(pr['coeffs']).to(dev), times)
y0 = torch.from_numpy(pr['y0']).to(dev)
def step():
    yy = y0.clone().requires_grad_(True)
    S.sdeint(m, yy, times, method='euler', dt=1.0, options={'seed': 1})[-1].square().mean().backward()
for _ in range(20): step()
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize(); print('K1 fwd+bwd ms', (time.perf_counter()-t)/200*1e3)
pr_ = cProfile.Profile(); pr_.enable()
for _ in range(200): step()
torch.cuda.synchronize(); pr_.disable()
st = pstats.Stats(pr_); st.sort_stats('cumulative').print_stats(28)
