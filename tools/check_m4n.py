"""Development check of the diffusion-net MFMA kernels (snsde_m4n_kernel.h): SRK / Milstein trajectories vs the float64
oracle on replayed increments, the kernel family each shape takes, and timings of the verdict's target shapes."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import stable_neural_sdes_amd as S
from oracle import sde_oracle as O
from tests.helpers import make_problem, draw_dW, assert_parity, param_spec
dev = torch.device('cuda:0')


def flat_params(p, io, no, NL, C, H):
    return torch.from_numpy(np.concatenate([np.asarray(p[n], np.float32).reshape(-1) for n, _ in param_spec(io, no, NL, C, H)])).to(dev)


def solve(pr, ts, dt, dW, dU, method, kernel):
    io, no, NL, C, H = pr['io'], pr['no'], pr['NL'], pr['C'], pr['H']
    model = S.engine.model_struct(C, H, H, NL, io, no)
    grid = S.engine.step_grid(np.asarray(ts, np.float32), dt, pr['times'], dev)
    call = S.engine.SolveCall(model, flat_params(pr['params'], io, no, NL, C, H), torch.from_numpy(pr['coeffs']).to(dev), grid,
                              torch.from_numpy(pr['y0']).to(dev), dW=None if dW is None else torch.from_numpy(dW).to(dev),
                              dU=None if dU is None else torch.from_numpy(dU).to(dev), method=method, kernel=kernel, save_traj=True)
    ys = call.launch()
    torch.cuda.synchronize()
    return ys.cpu().numpy(), call


def draw_dU(seed, dW, ts, dt):
    t0, t1, *_ = O.step_grid(np.asarray(ts, np.float32), dt)
    h = (t1 - t0).astype(np.float32)[:, None, None]
    xi = np.random.default_rng(seed + 7).standard_normal(dW.shape).astype(np.float32)
    return (h * (0.5 * dW + np.sqrt(h / 12) * xi)).astype(np.float32)


CASES = [  # io, no, NL, B, H, C, L, ts, dt
    (1, 18, 2, 9, 16, 3, 8, [0, 7], 0.5), (3, 15, 3, 8, 16, 4, 8, [0, 7], 1.0), (1, 14, 1, 17, 32, 3, 9, [0, 2.5, 8], 0.5),
    (3, 18, 2, 33, 64, 5, 12, [0, 2.5, 11], 0.5), (5, 19, 2, 21, 64, 5, 9, [0, 8], 1.0), (4, 19, 2, 21, 128, 21, 10, [0, 9], 1.0),
    (2, 14, 2, 13, 32, 7, 9, [0, 3.5, 8], 0.5), (6, 15, 3, 9, 64, 40, 8, [0, 7], 1.0), (1, 18, 2, 37, 128, 5, 9, [0, 8], 1.0),
    (3, 18, 3, 11, 128, 5, 9, [0, 8], 0.5), (4, 18, 1, 11, 128, 69, 9, [0, 8], 1.0), (6, 19, 4, 7, 32, 3, 8, [0, 7], 1.0),
    (1, 18, 2, 8, 24, 3, 8, [0, 7], 0.5),
]
bad = 0
for method in ('srk', 'milstein'):
    for ci, (io, no, NL, B, H, C, L, ts, dt) in enumerate(CASES):
        pr = make_problem(900 + ci, io, no, NL, B, H, C, L)
        dW = draw_dW(900 + ci, ts, dt, B, H)
        dU = draw_dU(900 + ci, dW, ts, dt) if method == 'srk' else None
        model = S.engine.model_struct(C, H, H, NL, io, no)
        N = S.engine.step_grid(np.asarray(ts, np.float32), dt, pr['times'], dev).N
        path = S.engine.forward_path(model, B, L, N, method=method, kernel='mfma4')
        if path in ('none', 'generic', 'generic-srk'):
            print(f'{method} case {ci} ({io},{no}) NL={NL} H={H} C={C}: path {path} (not on the MFMA net kernels)')
            continue
        ys, call = solve(pr, ts, dt, dW, dU, method, 'mfma4')
        ref64, traj64 = O.solve_diffusion_model(pr['params'], io, no, pr['coeffs'], pr['times'], pr['y0'], np.asarray(ts, np.float32), dt,
                                                dW, method=method, dtype=np.float64, dU=dU)
        cpu32, _ = O.solve_diffusion_model(pr['params'], io, no, pr['coeffs'], pr['times'], pr['y0'], np.asarray(ts, np.float32), dt,
                                           dW, method=method, dtype=np.float32, dU=dU)
        try:
            rep = assert_parity(ys, ref64, cpu32, what=f'{method} case {ci}')
            assert_parity(call.traj.cpu().numpy(), traj64, what=f'{method} traj {ci}')
            print(f'{method} case {ci} ({io},{no}) NL={NL} H={H} C={C}: path {path} ok {rep}')
        except AssertionError as e:
            bad += 1
            print(f'{method} case {ci} ({io},{no}) NL={NL} H={H} C={C}: path {path} FAIL {str(e)[:300]}')
        # Philox run equals the generic kernel's (same stream specification)
        y1, _ = solve(pr, ts, dt, None, None, method, 'mfma4')
        y2, _ = solve(pr, ts, dt, None, None, method, 'generic')
        d = np.abs(y1 - y2).max() / (np.abs(y2).max() + 1e-9)
        if not d < 2e-4:
            bad += 1
            print(f'   philox run vs generic kernel: rel diff {d:.3e} FAIL')
print('failures:', bad)

# timings
for (io, no, B, H, C, L, method) in ((1, 18, 1024, 128, 21, 50, 'srk'), (3, 18, 2048, 64, 69, 72, 'srk'), (3, 18, 2048, 64, 69, 72, 'milstein'),
                                     (1, 18, 1024, 128, 21, 50, 'milstein'), (4, 19, 1024, 128, 21, 101, 'srk')):
    pr = make_problem(7, io, no, 2, B, H, C, L, nan_frac=0.2)
    model = S.engine.model_struct(C, H, H, 2, io, no)
    flat = flat_params(pr['params'], io, no, 2, C, H)
    coeffs = torch.from_numpy(pr['coeffs']).to(dev)
    y0 = torch.from_numpy(pr['y0']).to(dev)
    grid = S.engine.step_grid(pr['times'], 1.0, pr['times'], dev)
    for kernel in ('auto', 'generic'):
        call = S.engine.SolveCall(model, flat, coeffs, grid, y0, method=method, kernel=kernel, seed=3)
        for _ in range(3): call.launch()
        ts_ = []
        for _ in range(15):          # median of per-solve wall times
            torch.cuda.synchronize(); t = time.perf_counter()
            call.launch()
            torch.cuda.synchronize(); ts_.append(time.perf_counter() - t)
        print(f'({io},{no}) {method} B={B} H={H} C={C} N={grid.N} kernel={kernel} path={S.engine.forward_path(model, B, L, grid.N, method=method, kernel=kernel)}: '
              f'forward {np.median(ts_) * 1e3:.3f} ms')
