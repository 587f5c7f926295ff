#!/usr/bin/env python3
"""Development check of the lean M4 kernel: parity of a few shapes against the fp64 oracle (supplied dW) and kernel time
of the K2 bench solve, lean vs general kernel (SNSDE_NO_LEAN=1 is read once per process, so the A/B runs as two processes).
usage: [SNSDE_LIB=...] lean_check.py [parity|time]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stable_neural_sdes_amd as S
from oracle import sde_oracle as O
from tests.helpers import assert_parity, draw_dW, make_problem, param_spec
dev = torch.device('cuda:0')


def solve(pr, ts, dt, dW, method='euler', kernel='mfma4', **kw):
    io, no, NL, C, H = pr['io'], pr['no'], pr['NL'], pr['C'], pr['H']
    model = S.engine.model_struct(C, H, H, NL, io, no)
    flat = torch.from_numpy(np.concatenate([pr['params'][n].reshape(-1) for n, _ in param_spec(io, no, NL, C, H)])).to(dev)
    grid = S.engine.step_grid(np.asarray(ts, np.float32), dt, pr['times'], dev)
    call = S.engine.SolveCall(model, flat, torch.from_numpy(pr['coeffs']).to(dev), grid, torch.from_numpy(pr['y0']).to(dev),
                              dW=None if dW is None else torch.from_numpy(dW).to(dev), method=method, kernel=kernel, **kw)
    ys = call.launch()
    torch.cuda.synchronize()
    return ys.cpu().numpy(), call


def parity():
    cases = [(4, 17, 2, 64, 128, 21, 17, 'euler'), (4, 17, 2, 7, 128, 21, 17, 'milstein'), (4, 16, 2, 33, 128, 5, 9, 'euler'),
             (6, 17, 2, 16, 128, 21, 12, 'euler'), (4, 0, 2, 16, 128, 21, 12, 'euler'), (4, 9, 2, 16, 128, 21, 12, 'euler')]
    for io, no, NL, B, H, C, L, method in cases:
        pr = make_problem(11, io, no, NL, B, H, C, L)
        ts = np.array([0., 2.5, float(L - 1)], np.float32)
        dW = draw_dW(3, ts, 0.5, B, H)
        try:
            ys, _ = solve(pr, ts, 0.5, dW, method)
        except Exception as e:
            print('case', (io, no, NL, B, H, C, L, method), 'ERR', e)
            continue
        ref, _ = O.solve_diffusion_model(pr['params'], io, no, pr['coeffs'], pr['times'], pr['y0'], ts, 0.5, dW,
                                         dtype=np.float64, method=method)
        c32, _ = O.solve_diffusion_model(pr['params'], io, no, pr['coeffs'], pr['times'], pr['y0'], ts, 0.5, dW,
                                         dtype=np.float32, method=method)
        rep = assert_parity(ys, ref, c32, what=str((io, no, method)))
        print('case', (io, no, NL, B, H, C, L, method), 'ok', {k: f'{v:.2e}' for k, v in rep.items()})


def timing():
    import bench
    kernel = 'mfma4'
    pr, params, flat, coeffs, y0 = bench.build_inputs(dev, 0)
    model = S.engine.model_struct(bench.C, bench.H, bench.H, bench.NL, bench.IO, bench.NO)
    grid = S.engine.step_grid(np.array([0.0, 100.0], np.float32), 1.0, pr['times'], dev)
    call = S.engine.SolveCall(model, flat, coeffs, grid, y0, seed=1, kernel=kernel)
    for _ in range(5):
        call.launch()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
    for a, b in ev:
        a.record(); call.launch(reuse_prepared=True); b.record()
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in ev])
    print(f"{os.path.basename(os.environ.get('SNSDE_LIB', 'libsnsde.so')):20s} NO_LEAN={os.environ.get('SNSDE_NO_LEAN', '-')} "
          f"median {np.median(t)*1e3:7.1f} us  min {t.min()*1e3:7.1f} us  sum(ys) {float(call.ys.double().sum()):.6f}")


if __name__ == '__main__':
    (parity if (len(sys.argv) > 1 and sys.argv[1] == 'parity') else timing)()
