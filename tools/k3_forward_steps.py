#!/usr/bin/env python3
"""n forward solves of K3 (GSDE (6,17), 4096 rows, H = 128, 200 Euler steps, Hermite coefficients, Philox) for rocprofv3 passes."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stable_neural_sdes_amd as S
import bench
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
pr, p0, flat, coeffs, y0 = bench.build_inputs(dev, 0, io=6, no=17, nl=2, b=4096, h=128, c=21, l=201, nan_frac=0.0, hermite=True)
model = S.engine.model_struct(21, 128, 128, 2, 6, 17)
grid = S.engine.step_grid(np.array([0.0, 200.0], np.float32), 1.0, pr['times'], dev)
call = S.engine.SolveCall(model, flat, coeffs, grid, y0, seed=2024)
for _ in range(n):
    call.launch()
torch.cuda.synchronize()
