#!/usr/bin/env python3
"""Time forward-only, training-mode forward, adjoint kernel and the full fwd+bwd of the K2 workload."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stable_neural_sdes_amd as S
import bench
from tests.helpers import make_problem
dev = torch.device('cuda:0')
pr = make_problem(1234, bench.IO, bench.NO, bench.NL, bench.B, bench.H, bench.C, bench.L, nan_frac=0.3)
m = S.Diffusion_model(bench.C, bench.H, bench.H, bench.NL, input_option=bench.IO, noise_option=bench.NO)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()})
m = m.to(dev)
coeffs = torch.from_numpy(pr['coeffs']).to(dev); times = torch.from_numpy(pr['times']).to(dev)
m.set_X(coeffs, times)
ts = torch.tensor([0., 100.], device=dev)
y0 = torch.from_numpy(pr['y0']).to(dev)

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

def fwd_nograd():
    with torch.no_grad():
        S.sdeint(m, y0, ts, method='euler', dt=1.0, options={'seed': 1})
def fwd_bwd():
    yy = y0.clone().requires_grad_(True)
    ys = S.sdeint(m, yy, ts, method='euler', dt=1.0, options={'seed': 1})
    ys[-1].square().mean().backward()
print('sdeint forward (no_grad, incl. host plumbing): %.3f ms' % timeit(fwd_nograd))
print('sdeint forward+backward (autograd.Function):   %.3f ms' % timeit(fwd_bwd))
# components
rec = S.engine.recognise(m); model, layout, numel = rec
flat = S.engine.flatten_params(m, layout, numel, dev)
grid = S.engine.step_grid(np.array([0., 100.], np.float32), 1.0, pr['times'], dev)
call = S.engine.SolveCall(model, flat, coeffs, grid, y0, seed=1, save_traj=True, save_dW=True, save_act=True)
print('training-mode forward kernel(s):               %.3f ms' % timeit(lambda: call.launch()))
g = torch.randn_like(call.ys)
print('adjoint (fold+pack+reverse kernel):            %.3f ms' % timeit(lambda: S.engine.solve_backward(call, g)))
adj, delta = S.engine.solve_backward(call, g, save_delta=True)
from stable_neural_sdes_amd.torchsde import _parameter_gradients
print('parameter-gradient pass (batched, torch):      %.3f ms' % timeit(lambda: _parameter_gradients(m, call, grid, adj), 5))
from stable_neural_sdes_amd.torchsde import _parameter_gradients_gemm
print('parameter-gradient pass (GEMMs on saved tensors): %.3f ms' % timeit(lambda: _parameter_gradients_gemm(m, call, grid, adj, delta), 5))
print('parameter-gradient pass (native snsde_param_gradients): %.3f ms' % timeit(lambda: S.engine.param_gradients(call, adj, delta), 10))
ga = _parameter_gradients(m, call, grid, adj); gb = _parameter_gradients_gemm(m, call, grid, adj, delta)
for (nm, _), x, y in zip(m.named_parameters(), ga, gb):
    print(f'  {nm:28s} rel diff {float((x - y).abs().max() / (x.abs().max() + 1e-20)):.2e}')
