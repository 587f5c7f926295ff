#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05w; mkdir -p $O
cat > /tmp/hp.py <<'PY'
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import numpy as np, torch
import bench, stable_neural_sdes_amd as S
dev = torch.device('cuda:0'); stream = torch.cuda.current_stream(dev)
sde, times, y0 = bench._module(dev, 3, 18, 2048, 64, 69, 72, 77)
params = list(sde.parameters())
opts = {'seed': 5, 'strict': True}
def step():
    for p in params: p.grad = None
    yy = y0.clone().requires_grad_(True)
    S.torchsde.sdeint(sde, yy, times, dt=1.0, method='euler', options=opts)[-1].square().mean().backward()
for _ in range(20): step()
torch.cuda.synchronize()
# host time per step without waiting for the GPU (launch-only): enqueue 200 steps, time the enqueue
t0 = time.perf_counter()
for _ in range(200): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host enqueue per step %.4f ms; incl. drain %.4f ms' % ((t1 - t0) / 200 * 1e3, (t2 - t0) / 200 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45); print(s.getvalue()[:9000])
PY
timeout 300 python /tmp/hp.py > $O/hostprof.txt 2>&1 < /dev/null; head -90 $O/hostprof.txt | cut -c1-170
