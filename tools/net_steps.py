#!/usr/bin/env python3
"""Training steps (sdeint forward + backward) through the diffusion-net kernels (snsde_m4n_kernel and its adjoints) at the
shapes VERDICT r3 names: K4 shape (3,18) B=2048 H=64 C=69 N=71 and (1,18) B=1024 H=128 C=21 N=49, srk and milstein.
Run under rocprofv3 for kernel stats / PMC passes (tools/pmc_net_kernels.sh); with `time` as first argument it prints
the per-call host share instead (whole sdeint() wall time vs the HIP-event time of the solve kernel alone).
usage: net_steps.py [steps] | net_steps.py time"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stable_neural_sdes_amd as S
from tests.helpers import make_problem
dev = torch.device('cuda:0')
SHAPES = [  # tag, io, no, B, H, C, L, method
    ('K4_3_18_euler', 3, 18, 2048, 64, 69, 72, 'euler'),      # BASELINE config 4 (wave-owns-rows kernel, snsde_w4_kernel.h)
    ('K4_3_18_srk', 3, 18, 2048, 64, 69, 72, 'srk'),
    ('K4_3_18_milstein', 3, 18, 2048, 64, 69, 72, 'milstein'),
    ('naive_1_18_srk_H128', 1, 18, 1024, 128, 21, 50, 'srk'),
]
timing = len(sys.argv) > 1 and sys.argv[1] == 'time'
steps = int(sys.argv[1]) if len(sys.argv) > 1 and not timing else 4


def setup(io, no, B, H, C, L):
    pr = make_problem(7, io, no, 2, B, H, C, L, nan_frac=0.2)
    m = S.Diffusion_model(C, H, H, 2, input_option=io, noise_option=no)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()}); m = m.to(dev)
    times = torch.from_numpy(pr['times']).to(dev); m.set_X(torch.from_numpy(pr['coeffs']).to(dev), times)
    return pr, m, times, torch.from_numpy(pr['y0']).to(dev)


for tag, io, no, B, H, C, L, meth in SHAPES:
    pr, m, times, y0 = setup(io, no, B, H, C, L)
    ts = times[[0, -1]]
    opts = {'seed': 1, 'strict': True}

    def fwd():
        with torch.no_grad():
            return S.sdeint(m, y0, ts, method=meth, dt=1.0, options=opts)

    def fb():
        m.zero_grad(set_to_none=True)
        yy = y0.clone().requires_grad_(True)
        S.sdeint(m, yy, ts, method=meth, dt=1.0, options=opts)[-1].square().mean().backward()

    if not timing:
        for _ in range(steps):
            fwd()
        for _ in range(steps):
            fb()
        torch.cuda.synchronize()
        continue

    def wall(fn, n=30):
        for _ in range(5):
            fn()
        out = []
        for _ in range(n):
            torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
            out.append(time.perf_counter() - t)
        return np.median(out) * 1e3

    def pipelined(fn, n=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3

    # kernel alone: the prepared call relaunched with REUSE_PREPARED between HIP events
    model = S.engine.model_struct(C, H, H, 2, io, no)
    grid = S.engine.step_grid(ts.cpu().numpy(), 1.0, pr['times'], dev)
    from tests.helpers import param_spec
    flat = torch.from_numpy(np.concatenate([pr['params'][n].reshape(-1) for n, _ in param_spec(io, no, 2, C, H)])).to(dev)
    kern = float('nan')
    try:
        call = S.engine.SolveCall(model, flat, m.coeffs, grid, y0, seed=1, method=meth)
        for _ in range(3):
            call.launch()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
        for a, b in ev:
            a.record(); call.launch(reuse_prepared=True); b.record()
        torch.cuda.synchronize()
        kern = np.median([a.elapsed_time(b) for a, b in ev])
    except Exception as e:          # host-share line still useful without the kernel-only figure
        print('kernel-only timing unavailable:', type(e).__name__, str(e)[:200])
    print(f'{tag}: sdeint() forward wall {wall(fwd):.3f} ms, pipelined {pipelined(fwd):.3f} ms, solve kernel alone {kern:.3f} ms; '
          f'fwd+bwd wall {wall(fb):.3f} ms, pipelined {pipelined(fb):.3f} ms')
