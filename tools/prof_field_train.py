"""Training steps of the K2-sized tutorial LNSDE field (1024 rows, H = 128, 100 Euler steps) for rocprofv3 --kernel-trace --stats;
prints the wall time per step and a host-side breakdown (cProfile) when run plainly."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import stable_neural_sdes_amd as S
from tests.helpers import make_problem
from tests.tutorial_fields import TutorialField
dev = torch.device('cuda:0')
kind = sys.argv[1] if len(sys.argv) > 1 else 'lnsde'
nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows, hh, cc, n = 1024, 128, 2, 100
times = np.linspace(0.0, 1.0, 11).astype(np.float32)
pr = make_problem(99, 4, 17, 2, rows, hh, cc, len(times), times=times)
torch.manual_seed(99)
field = TutorialField(kind, cc, hh, 1).to(dev)
tt = torch.from_numpy(times).to(dev)
field.set_X(torch.from_numpy(pr['coeffs']).to(dev), tt)
y0 = torch.from_numpy(pr['y0']).abs().to(dev) + 0.1
ts = tt[[0, -1]]
def step():
    field.zero_grad(set_to_none=True)
    out = S.sdeint(field, y0, ts, dt=1.0 / n, method='euler', options={'seed': 1})
    out[-1].square().mean().backward()
for _ in range(5): step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(nrep): step()
torch.cuda.synchronize(); print(kind, 'training step ms', (time.perf_counter() - t) / nrep * 1e3)
if os.environ.get('HOSTPROF'):
    import cProfile, pstats
    p = cProfile.Profile(); p.enable()
    for _ in range(nrep): step()
    torch.cuda.synchronize(); p.disable()
    pstats.Stats(p).sort_stats('cumulative').print_stats(35)
