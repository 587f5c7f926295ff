#!/usr/bin/env python3
"""SRK (torch_ists default): fused forward / forward+backward against the tensor-op loop on the same GPU."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stable_neural_sdes_amd as S
from tests.helpers import make_problem
dev = torch.device('cuda:0')

def timeit(fn, n=5):
    for _ in range(2): fn()
    best = float('inf')
    for _ in range(3):      # best of three short loops: one-off allocator growth would otherwise dominate
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / n * 1e3)
    return best

for B, H, C, L in ((64, 32, 4, 50), (256, 64, 8, 100), (1024, 128, 21, 101)):
    pr = make_problem(7, 4, 17, 2, B, H, C, L, nan_frac=0.2)
    m = S.Diffusion_model(C, H, H, 2, input_option=4, noise_option=17)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()})
    m = m.to(dev)
    times = torch.from_numpy(pr['times']).to(dev)
    m.set_X(torch.from_numpy(pr['coeffs']).to(dev), times)
    y0 = torch.from_numpy(pr['y0']).to(dev)
    def run(backend, grad):
        def fn():
            yy = y0.clone().requires_grad_(grad)
            with torch.set_grad_enabled(grad):
                ys = S.sdeint(m, yy, times, method='srk', dt=1.0, options={'seed': 1, 'backend': backend})
                if grad: ys.square().mean().backward()
        return fn
    print(f'B={B} H={H} N={L - 1} srk: fused fwd {timeit(run("auto", False)):.2f} ms, fused fwd+bwd {timeit(run("auto", True)):.2f} ms | '
          f'tensor-op loop fwd {timeit(run("torch", False), 2):.1f} ms, fwd+bwd {timeit(run("torch", True), 2):.1f} ms')

# component split of the backward at the middle size
B, H, C, L = 256, 64, 8, 100
pr = make_problem(7, 4, 17, 2, B, H, C, L, nan_frac=0.2)
m = S.Diffusion_model(C, H, H, 2, input_option=4, noise_option=17)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()})
m = m.to(dev)
times = torch.from_numpy(pr['times']).to(dev)
coeffs = torch.from_numpy(pr['coeffs']).to(dev)
m.set_X(coeffs, times)
rec = S.engine.recognise(m); model, layout, numel = rec
flat = S.engine.flatten_params(m, layout, numel, dev)
grid = S.engine.step_grid(pr['times'], 1.0, pr['times'], dev)
call = S.engine.SolveCall(model, flat, coeffs, grid, torch.from_numpy(pr['y0']).to(dev), seed=1, method='srk', kernel='generic',
                          save_traj=True, save_dW=True)
print('srk forward kernel (training mode):  %.2f ms' % timeit(lambda: call.launch()))
g = torch.randn_like(call.ys)
print('srk adjoint kernel:                  %.2f ms' % timeit(lambda: S.engine.solve_backward(call, g)))
adj = S.engine.solve_backward(call, g)
from stable_neural_sdes_amd.torchsde import _parameter_gradients
print('srk parameter pass (batched autograd): %.2f ms' % timeit(lambda: _parameter_gradients(m, call, grid, adj, method='srk')))
