#!/usr/bin/env python3
"""cProfile of the host side of one fused training step (K2 workload): where the Python time goes."""
import cProfile, os, pstats, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stable_neural_sdes_amd as S
import bench
from tests.helpers import make_problem
dev = torch.device('cuda:0')
IO, NO, NL, B, H, C_, L, METHOD, TS_ALL = bench.IO, bench.NO, bench.NL, bench.B, bench.H, bench.C, bench.L, 'euler', False
if 'k1' in sys.argv:
    IO, NO, NL, B, H, C_, L, METHOD, TS_ALL = 2, 16, 1, 256, 32, 2, 51, 'euler', True
if 'k5' in sys.argv:
    IO, NO, NL, B, H, C_, L, METHOD, TS_ALL = 4, 17, 2, 128, 256, 14, 50, 'milstein', True
pr = make_problem(1234, IO, NO, NL, B, H, C_, L, nan_frac=0.3)
m = S.Diffusion_model(C_, H, H, NL, input_option=IO, noise_option=NO)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()})
m = m.to(dev)
m.set_X(torch.from_numpy(pr['coeffs']).to(dev), torch.from_numpy(pr['times']).to(dev))
ts = torch.from_numpy(pr['times']).to(dev) if TS_ALL else torch.tensor([0., float(L - 1)], device=dev)
y0 = torch.from_numpy(pr['y0']).to(dev)
grad = 'train' in sys.argv
def step():
    if grad:
        yy = y0.clone().requires_grad_(True)
        S.sdeint(m, yy, ts, method=METHOD, dt=1.0, options={'seed': 1})[-1].square().mean().backward()
    else:
        with torch.no_grad():
            S.sdeint(m, y0, ts, method=METHOD, dt=1.0, options={'seed': 1})
for _ in range(5): step()
torch.cuda.synchronize()
pf = cProfile.Profile(); pf.enable()
for _ in range(20): step()
pf.disable(); torch.cuda.synchronize()
st = pstats.Stats(pf); st.sort_stats('tottime').print_stats(22)
