"""Host-side profile of one training step of a tutorial-style field (BASELINE config 0 shape: LSDE field, 256 rows, H = 32, 50 Euler steps)."""
import os, sys, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import stable_neural_sdes_amd as S
from tests.helpers import make_problem
from tests.tutorial_fields import TutorialField
dev = torch.device('cuda:0')
kind = sys.argv[1] if len(sys.argv) > 1 else 'lsde'
rows, hh, cc, n = 256, 32, 2, 50
times = np.linspace(0.0, 1.0, 11).astype(np.float32)
pr = make_problem(99, 4, 17, 2, rows, hh, cc, len(times), times=times)
torch.manual_seed(99)
field = TutorialField(kind, cc, hh, 1).to(dev)
tt = torch.from_numpy(times).to(dev)
field.set_X(torch.from_numpy(pr['coeffs']).to(dev), tt)
y0 = torch.from_numpy(pr['y0']).abs().to(dev) + 0.1
opt = torch.optim.Adam(field.parameters(), lr=1e-3)
def step():
    opt.zero_grad(set_to_none=True)
    out = S.sdeint(field, y0, tt, dt=1.0 / n, method='euler')
    out[-1].square().mean().backward()
    opt.step()
for _ in range(20): step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize(); print(kind, 'training step ms', (time.perf_counter() - t) / 200 * 1e3)
pr_ = cProfile.Profile(); pr_.enable()
for _ in range(200): step()
torch.cuda.synchronize(); pr_.disable()
st = pstats.Stats(pr_); st.sort_stats('cumulative').print_stats(45)
