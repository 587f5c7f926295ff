#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05I; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_w4.py -x -q -m gpu > $O/w4_tests.log 2>&1 < /dev/null; echo "w4 tests rc=$?" >> $O/w4_tests.log
tail -5 $O/w4_tests.log
cat > /tmp/ab.py <<'PY'
import os, sys
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import numpy as np, torch
import bench, stable_neural_sdes_amd as S
dev = torch.device('cuda:0'); stream = torch.cuda.current_stream(dev)
sde, times, y0 = bench._module(dev, 3, 18, 2048, 64, 69, 72, 77)
params = list(sde.parameters())
for rep in range(2):
    for method in ('srk', 'euler'):
        opts = {'seed': 5, 'strict': True}
        def step():
            for p in params: p.grad = None
            yy = y0.clone().requires_grad_(True)
            S.torchsde.sdeint(sde, yy, times, dt=1.0, method=method, options=opts)[-1].square().mean().backward()
        t = bench.event_times_ms(step, stream, 30, 5)
        print(rep, method, 'fwd+bwd median %.4f p10 %.4f p90 %.4f' % (np.median(t), np.percentile(t, 10), np.percentile(t, 90)), flush=True)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o ab -- python /tmp/ab.py > /tmp/prof.log 2>&1 < /dev/null
grep fwd /tmp/prof.log
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $GRAFT_REPO_ROOT/$O/ab_kernel_stats.csv; head -5 "$f" | cut -c1-220; else tail -5 /tmp/prof.log; fi
