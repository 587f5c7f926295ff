"""Host-side profile of the LatentSDE wrapper's forward (split solve) under srk."""
import os, sys, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import stable_neural_sdes_amd as S
from tests.latent_field import LatentField
dev = torch.device('cuda:0')
torch.manual_seed(1)
m = LatentField(4, 32, 32, 2).to(dev)
times = torch.linspace(0, 1, 50, device=dev)
X = torch.cumsum(0.2 * torch.randn(1024, 50, 4, device=dev), dim=1)
coeffs = S.torchcde.hermite_cubic_coefficients_with_backward_differences(X, times)
method = sys.argv[1] if len(sys.argv) > 1 else 'srk'
def fwd():
    with torch.no_grad():
        m(coeffs, times, method=method, options={'seed': 3})
for _ in range(5): fwd()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(50): fwd()
torch.cuda.synchronize(); print(method, 'forward ms', (time.perf_counter() - t) / 50 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(50): fwd()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(32)
