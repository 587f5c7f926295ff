#!/usr/bin/env python3
"""Per-phase cycle trace of the MFMA kernel (needs libsnsde_trace.so built with -DSNSDE_TRACE)."""
import os, sys, subprocess
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# build locally first:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSNSDE_TRACE -Iinclude \
#     -Istable-neural-sdes_amd/csrc -o stable-neural-sdes_amd/libsnsde_trace.so stable-neural-sdes_amd/csrc/*.hip
os.environ['SNSDE_LIB'] = os.path.join(ROOT, 'stable-neural-sdes_amd', 'libsnsde_trace.so')
kernel = sys.argv[1] if len(sys.argv) > 1 else 'mfma4'
import stable_neural_sdes_amd as S
import bench
dev = torch.device('cuda:0')
pr, params, flat, coeffs, y0 = bench.build_inputs(dev, 0)
model = S.engine.model_struct(bench.C, bench.H, bench.H, bench.NL, bench.IO, bench.NO)
grid = S.engine.step_grid(np.array([0.0, 100.0], np.float32), 1.0, pr['times'], dev)
dW = None
if len(sys.argv) > 2 and sys.argv[2] == 'dw':
    dW = torch.randn(grid.N, bench.B, bench.H, device=dev)
call = S.engine.SolveCall(model, flat, coeffs, grid, y0, dW=dW, seed=1, kernel=kernel, save_dW=True)
for _ in range(3):
    call.launch()
torch.cuda.synchronize()
t = call.dW_out.reshape(-1)[:8 * 16].cpu().numpy().reshape(8, 16)[:, :10] / grid.N
names = ['barrier->top', 'row/coef/noise', 'L1 gemm', 'L1 store', 'L1 barrier', 'hid gemm', 'hid store+barrier',
         'out gemm', 'update', 'end barrier']
print('cycles per step (mean over steps), per wave:')
for i, nme in enumerate(names):
    print(f'{nme:20s}', ' '.join(f'{v:7.0f}' for v in t[:, i]))
print(f'{"total":20s}', ' '.join(f'{v:7.0f}' for v in t.sum(1)))
os.environ.pop('SNSDE_LIB')
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
for a, b in ev:
    a.record(); call.launch(reuse_prepared=True); b.record()
torch.cuda.synchronize()
print('kernel ms (trace build):', np.mean([a.elapsed_time(b) for a, b in ev]))
