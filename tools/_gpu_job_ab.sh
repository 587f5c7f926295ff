#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05u; mkdir -p $O
cat > /tmp/ab.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import numpy as np, torch
import bench, stable_neural_sdes_amd as S
dev = torch.device('cuda:0'); stream = torch.cuda.current_stream(dev)
order = sys.argv[1].split(',')
sde, times, y0 = bench._module(dev, 3, 18, 2048, 64, 69, 72, 77)
params = list(sde.parameters())
for method in order:
    opts = {'seed': 5, 'strict': True}
    def step():
        for p in params: p.grad = None
        yy = y0.clone().requires_grad_(True)
        S.torchsde.sdeint(sde, yy, times, dt=1.0, method=method, options=opts)[-1].square().mean().backward()
    for blk in range(4):
        t0 = time.perf_counter()
        t = bench.event_times_ms(step, stream, 20, 5)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 25 * 1e3
        # host-only time per step: launch without waiting
        print(method, blk, 'event median %.4f p10 %.4f p90 %.4f | wall/step %.4f' % (np.median(t), np.percentile(t, 10), np.percentile(t, 90), wall), flush=True)
PY
for ord in euler srk,euler milstein,srk,euler; do echo "== order $ord"; timeout 300 python /tmp/ab.py $ord < /dev/null 2>&1 | grep -v amdgpu.ids; done > $O/order.txt
cat $O/order.txt
