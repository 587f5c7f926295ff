#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_w4.py -x -q -m gpu > $O/w4_tests.log 2>&1; echo "w4 tests rc=$?" >> $O/w4_tests.log
tail -15 $O/w4_tests.log
timeout 600 python -m pytest tests/test_gpu_backward_sizes.py -x -q -m gpu -k "K4" > $O/sizes.log 2>&1; tail -5 $O/sizes.log
timeout 600 python - > $O/ab.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import numpy as np, torch
import bench, stable_neural_sdes_amd as S
dev = torch.device('cuda:0'); stream = torch.cuda.current_stream(dev)
sde, times, y0 = bench._module(dev, 3, 18, 2048, 64, 69, 72, 77)
params = list(sde.parameters())
for rep in range(3):
    for method in ('srk', 'euler'):
        for kernel in ('auto', 'mfma4'):
            opts = {'seed': 5, 'strict': True, 'kernel': kernel}
            def step():
                for p in params: p.grad = None
                yy = y0.clone().requires_grad_(True)
                S.torchsde.sdeint(sde, yy, times, dt=1.0, method=method, options=opts)[-1].square().mean().backward()
            t = bench.event_times_ms(step, stream, 30, 5)
            print(rep, method, kernel, 'fwd+bwd median %.4f p10 %.4f p90 %.4f' % (np.median(t), np.percentile(t, 10), np.percentile(t, 90)), flush=True)
PY
cat $O/ab.txt
