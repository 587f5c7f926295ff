#!/usr/bin/env python3
"""Generic-sde path (arbitrary f / g modules, e.g. the tutorial's LipSwish fields): graph-replayed stepper vs eager loop."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stable_neural_sdes_amd as S
from tests.test_gpu_parity import _TutorialField
dev = torch.device('cuda:0')
def timeit(fn, n=5):
    for _ in range(2): fn()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / n * 1e3)
    return best
for B, H, L, dt in ((16, 32, 20, 0.05), (256, 64, 100, 0.01), (1024, 128, 100, 0.01)):
    torch.manual_seed(0)
    field = _TutorialField(2, H).to(dev)
    times = torch.linspace(0, 1, L, device=dev)
    X = torch.cumsum(torch.randn(B, L, 2, device=dev) * 0.1, 1)
    field.set_X(S.torchcde.hermite_cubic_coefficients_with_backward_differences(X, times), times)
    y0 = torch.randn(B, H, device=dev)
    for method in ('euler', 'srk'):
        def run(graph):
            def fn():
                with torch.no_grad():
                    S.sdeint(field, y0, times, dt=dt, method=method, options={'seed': 1, 'graph': graph})
            return fn
        steps = int(round(1 / dt))
        print(f'B={B} H={H} steps~{steps} {method}: graph-replayed {timeit(run(True)):.2f} ms, eager loop {timeit(run(False)):.2f} ms')
