#!/usr/bin/env python3
"""Host-side cost of one K2 training step (sdeint forward + loss.backward()): cProfile of the calling thread AND of the fused
backward node, which the autograd engine runs on its own thread (a plain cProfile sees it only as `run_backward`)."""
import cProfile, os, pstats, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stable_neural_sdes_amd as S
import bench
from stable_neural_sdes_amd import torchsde as T
from tests.helpers import make_problem
dev = torch.device('cuda:0')
pr = make_problem(1234, bench.IO, bench.NO, bench.NL, bench.B, bench.H, bench.C, bench.L, nan_frac=0.3)
m = S.Diffusion_model(bench.C, bench.H, bench.H, bench.NL, input_option=bench.IO, noise_option=bench.NO)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()}); m = m.to(dev)
m.set_X(torch.from_numpy(pr['coeffs']).to(dev), torch.from_numpy(pr['times']).to(dev))
ts = torch.tensor([0., float(bench.L - 1)], device=dev)
y0 = torch.from_numpy(pr['y0']).to(dev)
bw = cProfile.Profile()
orig = T._FusedSolve.backward
T._FusedSolve.backward = staticmethod(lambda ctx, g: bw.runcall(orig, ctx, g))
def step():
    yy = y0.clone().requires_grad_(True)
    S.sdeint(m, yy, ts, method='euler', dt=1.0, options={'seed': 1})[-1].square().mean().backward()
for _ in range(5): step()
torch.cuda.synchronize()
bw.clear()
pf = cProfile.Profile(); pf.enable()
for _ in range(50): step()
pf.disable(); torch.cuda.synchronize()
print('== calling thread (50 steps)'); pstats.Stats(pf).sort_stats('cumtime').print_stats(18)
print('== fused backward node (autograd thread, 50 steps)'); pstats.Stats(bw).sort_stats('cumtime').print_stats(18)
T._FusedSolve.backward = orig
# un-profiled host time per step: enqueue 200 steps back to back (the GPU queue never drains: host-bound if this exceeds the GPU time)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(200): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f'host enqueue time per step {(t1 - t) / 200 * 1e3:.3f} ms, wall per step {(t2 - t) / 200 * 1e3:.3f} ms')
