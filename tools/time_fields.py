"""Tutorial-style fields through sdeint(): fused path vs the graph-replayed stepper, Euler / Milstein / SRK (1024 rows, H = 128, 100 steps)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import stable_neural_sdes_amd as S
from tests.helpers import make_problem
from tests.tutorial_fields import TutorialField
dev = torch.device('cuda:0')
rows, hh, cc, n = 1024, 128, 2, 100
times = np.linspace(0.0, 1.0, 11).astype(np.float32)
pr = make_problem(99, 4, 17, 2, rows, hh, cc, len(times), times=times)
for kind in ('gsde', 'lnsde', 'nsde', 'ode'):
    torch.manual_seed(99)
    field = TutorialField(kind, cc, hh, 1).to(dev)
    tt = torch.from_numpy(times).to(dev)
    field.set_X(torch.from_numpy(pr['coeffs']).to(dev), tt)
    y0 = torch.from_numpy(pr['y0']).abs().to(dev) + 0.1
    ts = tt[[0, -1]]
    for method in (('euler',) if kind == 'ode' else ('euler', 'milstein', 'srk')):      # (scalar noise: Euler only)
        res = []
        for backend in ('auto', 'torch'):
            with torch.no_grad():
                f = lambda: S.sdeint(field, y0, ts, dt=1.0 / n, method=method, options={'seed': 1, 'backend': backend})
                for _ in range(3): f()
                ts_ = []
                for _ in range(11 if backend == 'auto' else 3):
                    torch.cuda.synchronize(); t = time.perf_counter()
                    f()
                    torch.cuda.synchronize(); ts_.append(time.perf_counter() - t)
                res.append(float(np.median(ts_)) * 1e3)
        cf = S.fields.compose(field)
        path = S.engine.forward_path(cf.model, rows, len(times), n, method=method, table=cf.tabulated) if cf is not None else None
        print(f'{kind:6s} {method:8s} fused path {path}: sdeint {res[0]:.3f} ms | tensor-op / graph stepper {res[1]:.1f} ms')
    # one training step: loss.backward() through the fused solve vs autograd through the tensor-op loop
    for tmethod in (('euler',) if kind == 'ode' else (('euler', 'milstein', 'srk') if kind == 'nsde' else ('euler', 'srk'))):
        res = []
        for backend in ('auto', 'torch'):
            def step():
                field.zero_grad(set_to_none=True)
                out = S.sdeint(field, y0, ts, dt=1.0 / n, method=tmethod, options={'seed': 1, 'backend': backend})
                out[-1].square().mean().backward()
            for _ in range(2): step()
            ts_ = []
            for _ in range(9 if backend == 'auto' else 2):
                torch.cuda.synchronize(); t = time.perf_counter()
                step()
                torch.cuda.synchronize(); ts_.append(time.perf_counter() - t)
            res.append(float(np.median(ts_)) * 1e3)
        mode = S.engine.backward_mode(cf.model, rows, len(times), S.engine.step_grid(times[[0, -1]], 1.0 / n, times, dev), tmethod,
                                      table=cf.tabulated) if cf is not None else None
        print(f'{kind:6s} {tmethod:8s} training step (backward mode {mode}): fused {res[0]:.3f} ms | autograd through the loop {res[1]:.1f} ms')
