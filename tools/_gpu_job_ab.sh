#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05D; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_w4.py -x -q -m gpu > $O/w4_tests.log 2>&1 < /dev/null; echo "w4 tests rc=$?" >> $O/w4_tests.log
tail -15 $O/w4_tests.log
timeout 300 python tools/time_w4.py milstein 1024 2048 4096 < /dev/null 2>&1 | grep -v amdgpu > $O/time_w4_mil.txt; cat $O/time_w4_mil.txt
cat > /tmp/ab.py <<'PY'
import os, sys
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import numpy as np, torch
import bench, stable_neural_sdes_amd as S
dev = torch.device('cuda:0'); stream = torch.cuda.current_stream(dev)
sde, times, y0 = bench._module(dev, 3, 18, 2048, 64, 69, 72, 77)
params = list(sde.parameters())
for rep in range(2):
    for kernel in ('auto', 'mfma4'):
        opts = {'seed': 5, 'strict': True, 'kernel': kernel}
        def step():
            for p in params: p.grad = None
            yy = y0.clone().requires_grad_(True)
            S.torchsde.sdeint(sde, yy, times[[0, -1]], dt=1.0, method='milstein', options=opts)[-1].square().mean().backward()
        t = bench.event_times_ms(step, stream, 30, 5)
        print(rep, 'milstein', kernel, 'fwd+bwd median %.4f p10 %.4f p90 %.4f' % (np.median(t), np.percentile(t, 10), np.percentile(t, 90)), flush=True)
PY
timeout 300 python /tmp/ab.py < /dev/null 2>&1 | grep -v amdgpu > $O/mil_ab.txt; cat $O/mil_ab.txt
