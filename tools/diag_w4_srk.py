"""Where do the wave-group and the 4-row-tile SRK adjoints differ (tests/test_gpu_w4.py::test_w4_srk_adjoint_equals_the_tile_adjoint)?
Per row: max |dL/dy0| difference, against the float64 loop's gradient."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stable_neural_sdes_amd as S
from tests.helpers import make_problem
DEV = 'cuda:0'
case = (5, 19, 2, 22, True)
io, no, NL, B, row_out = case
C, L, H = 5, 9, 64
pr = make_problem(9700 + B, io, no, NL, B, H, C, L)
ts = torch.from_numpy(pr['times'][[0, 2, 5, 8]]).to(DEV)
ro = torch.from_numpy(np.random.default_rng(2).integers(0, 4, size=B).astype(np.int32)).to(DEV) if row_out else None
out = {}
for kernel in ('auto', 'mfma4'):
    m = S.Diffusion_model(C, H, H, NL, input_option=io, noise_option=no)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()})
    m = m.to(DEV)
    m.set_X(torch.from_numpy(pr['coeffs']).to(DEV), torch.from_numpy(pr['times']).to(DEV))
    y0 = torch.from_numpy(pr['y0']).to(DEV).requires_grad_(True)
    opts = {'seed': 99, 'kernel': kernel, 'strict': True, 'row_out': ro}
    ys = S.sdeint(m, y0, ts, dt=0.5, method='srk', options=opts)
    w = torch.from_numpy(np.random.default_rng(3).standard_normal(tuple(ys.shape)).astype(np.float32)).to(DEV)
    (ys * w).sum().backward()
    out[kernel] = (ys.detach(), y0.grad.clone())
ya, ga = out['auto']; yb, gb = out['mfma4']
d = (ga - gb).abs()
print('rows with |d grad y0| > 1e-4:', [(int(r), float(d[r].max()), int((d[r] > 1e-4).sum())) for r in range(B) if float(d[r].max()) > 1e-4])
print('state diff max', float((ya - yb).abs().max()), 'grad scale', float(gb.abs().max()))
