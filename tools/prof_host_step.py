#!/usr/bin/env python3
"""Host-side (Python) time of one eager training step through sdeint at the K4 shape: cProfile of the forward call and - the autograd
engine runs it on its own thread - of _FusedSolve.backward, 200 steps each; plus the enqueue rate without waiting for the GPU."""
import cProfile, io, os, pstats, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stable_neural_sdes_amd as S
import bench
dev = torch.device('cuda:0')
method = sys.argv[1] if len(sys.argv) > 1 else 'euler'
sde, times, y0 = bench._module(dev, 3, 18, 2048, 64, 69, 72, 77)
params = list(sde.parameters())
opts = {'seed': 5, 'strict': True}
bprof = cProfile.Profile()
orig = S.torchsde._FusedSolve.backward


def patched(ctx, *g):
    bprof.enable()
    try:
        return orig(ctx, *g)
    finally:
        bprof.disable()


def step():
    for p in params:
        p.grad = None
    yy = y0.clone().requires_grad_(True)
    S.torchsde.sdeint(sde, yy, times, dt=1.0, method=method, options=opts)[-1].square().mean().backward()


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print('enqueue %.4f ms per step' % ((t1 - t0) / 200 * 1e3))
S.torchsde._FusedSolve.backward = staticmethod(patched)
fprof = cProfile.Profile()
fprof.enable()
for _ in range(200):
    step()
fprof.disable()
torch.cuda.synchronize()
for name, pr in (('forward thread', fprof), ('backward (autograd thread)', bprof)):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28)
    print('=====', name)
    print('\n'.join(l[:150] for l in s.getvalue().splitlines()[:45]))
