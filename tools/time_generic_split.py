"""Generic (VALU) kernel family: K2 forward, and the SRK training step of a diffusion-net model split into forward /
adjoint kernel / batched parameter pass (host-side timers with synchronisation around each piece)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import stable_neural_sdes_amd as S
from stable_neural_sdes_amd import torchsde as T, engine as E
from tests.helpers import make_problem
dev = torch.device('cuda:0')

def model(io, no, B, H, C, L):
    pr = make_problem(7, io, no, 2, B, H, C, L, nan_frac=0.2)
    m = S.Diffusion_model(C, H, H, 2, input_option=io, noise_option=no)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()}); m = m.to(dev)
    times = torch.from_numpy(pr['times']).to(dev); m.set_X(torch.from_numpy(pr['coeffs']).to(dev), times)
    return m, times, torch.from_numpy(pr['y0']).to(dev)

def tm(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

m, times, y0 = model(4, 17, 1024, 128, 21, 101)
ts = times[[0, -1]]
for method in ('euler', 'srk'):
    def fwd():
        with torch.no_grad(): S.sdeint(m, y0, ts, method=method, dt=1.0, options={'seed': 1, 'kernel': 'generic'})
    print(f'K2 model, generic kernel, {method}: forward {tm(fwd):.3f} ms')
def fb():
    m.zero_grad(set_to_none=True)
    yy = y0.clone().requires_grad_(True)
    S.sdeint(m, yy, ts, method='euler', dt=1.0, options={'seed': 1, 'kernel': 'generic', 'strict': True})[-1].square().mean().backward()
print(f'K2 model, generic kernels, euler: forward + backward {tm(fb):.3f} ms')

m, times, y0 = model(1, 18, 1024, 128, 21, 50)
acc = {}
def wrap(mod, name):
    orig = getattr(mod, name)
    def timed(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = orig(*a, **k)
        torch.cuda.synchronize(); acc.setdefault(name, []).append((time.perf_counter() - t) * 1e3)
        return r
    setattr(mod, name, timed)
def fwd():
    with torch.no_grad(): S.sdeint(m, y0, times, method='srk', dt=1.0, options={'seed': 1})
print(f'(1,18) srk B=1024 H=128 N=49: forward {tm(fwd):.3f} ms')
wrap(T, '_parameter_gradients'); wrap(E, 'solve_backward')
for _ in range(6):
    m.zero_grad(set_to_none=True)
    yy = y0.clone().requires_grad_(True)
    S.sdeint(m, yy, times, method='srk', dt=1.0, options={'seed': 1, 'strict': True})[-1].square().mean().backward()
torch.cuda.synchronize()
for k, v in acc.items(): print(f'   {k}: {np.median(v[2:]):.2f} ms')
