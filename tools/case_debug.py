#!/usr/bin/env python3
"""Development: one MFMA_CASES entry of tests/test_gpu_parity.py, per-solver-state error of the selected kernel vs the fp64 /
fp32 oracles.  usage: [SNSDE_NO_LEAN=1] case_debug.py <case index> [kernel]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_gpu_parity as T
from tests.helpers import make_problem, draw_dW
ci = int(sys.argv[1]); kernel = sys.argv[2] if len(sys.argv) > 2 else 'mfma4'
io, no, NL, B, H, C, L, ts, dt, method = T.MFMA_CASES[ci]
times = np.linspace(0, 1, L).astype(np.float32) if ts is None else None
pr = make_problem(300 + ci, io, no, NL, B, H, C, L, times=times)
if ts is None:
    ts = pr['times']; dt = dt or max(float(np.diff(pr['times']).min()), 1e-3)
dW = draw_dW(300 + ci, ts, dt, B, H)
ys, call = T.hip_solve(pr, ts, dt, dW=dW, method=method, save_traj=True, kernel=kernel)
ref64, traj64 = T.oracle_solve(pr, ts, dt, dW, method, np.float64)
cpu32, traj32 = T.oracle_solve(pr, ts, dt, dW, method, np.float32)
tr = call.traj.cpu().numpy().astype(np.float64)
print('case', T.MFMA_CASES[ci], 'NO_LEAN' if os.environ.get('SNSDE_NO_LEAN') else 'lean')
for n in range(tr.shape[0]):
    e = np.abs(tr[n] - traj64[n]); e32 = np.abs(traj32[n].astype(np.float64) - traj64[n])
    print(f'state {n:3d}: gpu mean {e.mean():.2e} max {e.max():.2e} | cpu32 mean {e32.mean():.2e} max {e32.max():.2e} | |y| max {np.abs(traj64[n]).max():.2f}')
