#!/bin/bash
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import numpy as np, torch
import bench, stable_neural_sdes_amd as S
dev = torch.device('cuda:0'); stream = torch.cuda.current_stream(dev)
sde, times, y0 = bench._module(dev, 3, 18, 2048, 64, 69, 72, 77)
params = list(sde.parameters())
for rep in range(2):
    for kernel in ('auto', 'mfma4'):
        for ts in (times, times[[0, -1]]):
            opts = {'seed': 5, 'strict': True, 'kernel': kernel}
            def step():
                for p in params: p.grad = None
                yy = y0.clone().requires_grad_(True)
                S.torchsde.sdeint(sde, yy, ts, dt=1.0, method='euler', options=opts)[-1].square().mean().backward()
            t = bench.event_times_ms(step, stream, 30, 5)
            print(rep, kernel, 'T =', len(ts), 'fwd+bwd median %.4f p10 %.4f p90 %.4f' % (np.median(t), np.percentile(t, 10), np.percentile(t, 90)), flush=True)
PY
