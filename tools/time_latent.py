"""torch-ists' LatentSDE shape (tests/latent_field.LatentField) through its own forward - sdeint_adjoint with names f_aug / g_aug:
the split solve (fused latent dynamics + batched KL quadrature) vs the tensor-op loop, inference and one training step."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import stable_neural_sdes_amd as S
from tests.latent_field import LatentField
dev = torch.device('cuda:0')


def timed(fn, warm, reps):
    for _ in range(warm): fn()
    out = []
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter()
        fn()
        torch.cuda.synchronize(); out.append(time.perf_counter() - t)
    return float(np.median(out)) * 1e3


for rows, hidden, L in ((1024, 32, 50), (1024, 64, 50), (1024, 128, 50)):
    torch.manual_seed(1)
    m = LatentField(4, hidden, hidden, 2).to(dev)
    times = torch.linspace(0, 1, L, device=dev)
    X = torch.cumsum(0.2 * torch.randn(rows, L, 4, device=dev), dim=1)
    coeffs = S.torchcde.hermite_cubic_coefficients_with_backward_differences(X, times)
    for method in ('euler', 'srk'):
        res = []
        for backend in ('auto', 'torch'):
            opts = {'seed': 3, 'backend': backend}
            with torch.no_grad():
                fwd = timed(lambda: m(coeffs, times, method=method, options=opts), 2, 9 if backend == 'auto' else 2)

            def step():
                m.zero_grad(set_to_none=True)
                out, latent, kl = m(coeffs, times, method=method, options=opts)
                (out.square().mean() + 1e-3 * kl).backward()
            trn = timed(step, 2, 7 if backend == 'auto' else 2)
            res.append((fwd, trn))
        print(f'LatentSDE rows={rows} hidden={hidden} L={L} {method:6s}: forward {res[0][0]:.3f} ms (tensor-op loop {res[1][0]:.1f} ms) | '
              f'training step {res[0][1]:.3f} ms (autograd through the loop {res[1][1]:.1f} ms)')
