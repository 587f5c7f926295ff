#!/usr/bin/env python3
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import bigcase
import stable_neural_sdes_amd as S
dev = torch.device('cuda:0')

def k3(rows, kernel):
    try:
        rep = bigcase.run_case('K3', dev, kernel=kernel, rows=rows, loop32=False)
        print('K3 rows', rows, kernel, 'finite; y0 max-rel', rep['tensors']['y0'], 'theta', rep['tensors']['theta'], flush=True)
    except AssertionError as e:
        print('K3 rows', rows, kernel, 'NON-FINITE in', e, flush=True)

# find where the K3 NaN comes from
import tests.bigcase as bc
orig_assert = None
def k3_detail(rows, kernel):
    io, no, NL, B, H, C, L, method, every, hermite, nanf = bc.CASES['K3']
    B = rows
    seed = 7000 + sum(map(ord, 'K3'))
    from tests.helpers import make_problem, draw_dW
    pr = make_problem(seed, io, no, NL, B, H, C, L, nan_frac=nanf, hermite=hermite)
    ts = np.array([pr['times'][0], pr['times'][-1]], np.float32)
    dW = draw_dW(seed, ts, 1.0, B, H)
    m = S.Diffusion_model(C, H, H, NL, input_option=io, noise_option=no)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()})
    m = m.to(dev); m.set_X(torch.from_numpy(pr['coeffs']).to(dev), torch.from_numpy(pr['times']).to(dev))
    y0 = torch.from_numpy(pr['y0']).to(dev).requires_grad_(True)
    ys = S.sdeint(m, y0, torch.from_numpy(ts).to(dev), bm=bc.ReplayBM(torch.from_numpy(dW).to(dev)), method='euler', dt=1.0,
                  options={'kernel': kernel, 'strict': True, 'save_traj': True})
    print('  forward finite', bool(torch.isfinite(ys).all()), 'max |ys|', float(ys.abs().max()))
    w = torch.randn_like(ys)
    (ys * w).sum().backward()
    bad = ~torch.isfinite(y0.grad)
    print('  y0.grad non-finite elements', int(bad.sum()), 'rows', bad.any(1).nonzero().flatten()[:20].tolist(), 'cols of first', bad[bad.any(1)][:1].nonzero()[:10].tolist() if bad.any() else None)
    for n, p in m.named_parameters():
        if p.grad is not None and not bool(torch.isfinite(p.grad).all()):
            print('  param', n, 'non-finite', int((~torch.isfinite(p.grad)).sum()))
    if bad.any():
        r = int(bad.any(1).nonzero()[0])
        print('  row', r, 'final |y| max', float(ys[-1, r].abs().max()), 'y0 grad row', y0.grad[r][:8].tolist())

for rows, kernel in ((4096, 'auto'), (4096, 'mfma4'), (4096, 'mfma16'), (4096, 'generic'), (512, 'auto'), (1024, 'auto')):
    print('K3 detail', rows, kernel, flush=True)
    try:
        k3_detail(rows, kernel)
    except Exception as e:
        print('  EXC', repr(e)[:300])
    torch.cuda.empty_cache()

# K4: per-row error structure
for kernel in ('auto', 'mfma4', 'mfma16', 'generic'):
    rep = bigcase.run_case('K4', dev, kernel=kernel, loop32=False)
    print(bigcase.format_report('K4 [' + kernel + ']', rep).split('\n')[0])
    print('   ', {n: '%.1e' % r['hip_max'] for n, r in rep['tensors'].items()})
    if kernel in ('auto',):
        # per-row y0 gradient error
        inp = rep['inputs']
        torch.save({'gy0': rep['grads']['y0'].cpu()}, '/tmp/k4_auto.pt')
    del rep; torch.cuda.empty_cache()
