"""Development check of the SRK-through-a-diffusion-net adjoint on the MFMA path (snsde_m4n_rev_kernel.h + the native
parameter pass): gradients vs float64 autograd through the tensor-op loop, and the training-step time of the verdict's shape."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import stable_neural_sdes_amd as S
from tests.helpers import make_problem
from tests.test_gpu_parity import _check_backward
dev = torch.device('cuda:0')
CASES = [  # io, no, NL, B, H, C, L, ts, dt
    (1, 18, 2, 9, 16, 3, 8, [0, 7], 0.5), (3, 15, 3, 8, 16, 4, 8, [0, 7], 1.0), (1, 14, 1, 17, 32, 3, 9, [0, 2.5, 8], 0.5),
    (3, 18, 2, 33, 64, 5, 12, [0, 2.5, 11], 0.5), (5, 19, 2, 21, 64, 5, 9, [0, 8], 1.0), (4, 19, 2, 21, 128, 21, 10, [0, 9], 1.0),
    (2, 14, 2, 13, 32, 7, 9, [0, 3.5, 8], 0.5), (6, 15, 3, 9, 64, 40, 8, [0, 7], 1.0), (1, 18, 2, 37, 128, 5, 9, [0, 8], 1.0),
    (3, 18, 3, 11, 128, 5, 9, [0, 8], 0.5), (6, 19, 4, 7, 32, 3, 8, [0, 7], 1.0), (4, 18, 1, 11, 128, 69, 9, [0, 8], 1.0),
]
method = sys.argv[1] if len(sys.argv) > 1 else 'srk'
bad = 0
for ci, (io, no, NL, B, H, C, L, ts, dt) in enumerate(CASES):
    model = S.engine.model_struct(C, H, H, NL, io, no)
    grid = S.engine.step_grid(np.asarray(ts, np.float32), dt, np.arange(L, dtype=np.float32), dev)
    mode = S.engine.backward_mode(model, B, L, grid, method, 'auto')
    try:
        _check_backward(5000 + ci, io, no, NL, B, H, C, L, ts, dt, method, "auto", strict=True)
        print(f'{method} bwd case {ci} ({io},{no}) NL={NL} H={H} C={C}: mode {mode} ok')
    except Exception as e:
        bad += 1
        print(f'{method} bwd case {ci} ({io},{no}) NL={NL} H={H} C={C}: mode {mode} FAIL {type(e).__name__} {str(e)[:300]}')
print('failures:', bad)

for (io, no, B, H, C, L, meth) in ((1, 18, 1024, 128, 21, 50, 'srk'), (3, 18, 2048, 64, 69, 72, 'srk'), (1, 18, 512, 64, 5, 50, 'srk'),
                                  (3, 18, 2048, 64, 69, 72, 'milstein'), (1, 14, 1024, 128, 21, 50, 'milstein'), (1, 18, 512, 64, 5, 50, 'milstein')):
    if meth != method:
        continue
    pr = make_problem(7, io, no, 2, B, H, C, L, nan_frac=0.2)
    m = S.Diffusion_model(C, H, H, 2, input_option=io, noise_option=no)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()}); m = m.to(dev)
    times = torch.from_numpy(pr['times']).to(dev); m.set_X(torch.from_numpy(pr['coeffs']).to(dev), times)
    y0 = torch.from_numpy(pr['y0']).to(dev)
    for kernel in ('auto', 'generic'):
        opts = {'seed': 1, 'strict': True, 'kernel': kernel}
        def fb():
            m.zero_grad(set_to_none=True)
            yy = y0.clone().requires_grad_(True)
            S.sdeint(m, yy, times, method=meth, dt=1.0, options=opts)[-1].square().mean().backward()
        for _ in range(3): fb()
        ts_ = []
        for _ in range(15):          # median of per-step wall times (a stray allocator / host hiccup does not move it)
            torch.cuda.synchronize(); t = time.perf_counter()
            fb()
            torch.cuda.synchronize(); ts_.append(time.perf_counter() - t)
        print(f'({io},{no}) {meth} B={B} H={H} C={C} N={L - 1} kernel={kernel}: fwd+bwd {np.median(ts_) * 1e3:.2f} ms')
