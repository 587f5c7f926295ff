#!/usr/bin/env python3
"""Whole-model training step (NeuralSDE wrapper: z0, output-time selection, fused solve, per-row gather, readout MLP,
loss, backward, Adam) for the classification recipes of the reference (common_sde.py:107-216)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stable_neural_sdes_amd as S
from tests.helpers import make_problem
dev = torch.device('cuda:0')

def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

for name, B, H, C, L in (('neurallnsde', 1024, 128, 21, 101), ('naivesde', 2048, 64, 69, 72), ('neuralgsde', 512, 128, 21, 201)):
    pr = make_problem(5, 4, 17, 2, B, H, C, L, nan_frac=0.2)
    torch.manual_seed(0)
    model, field = S.make_sde_model(name, C, 1, H, H, 2, initial=True)
    model = model.to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    times = torch.from_numpy(pr['times']).to(dev)
    coeffs = torch.from_numpy(pr['coeffs']).to(dev)
    fi = torch.randint(2, L, (B,), device=dev)
    target = (torch.rand(B, device=dev) > 0.5).float()
    def train_step():
        pred = model(times, [coeffs], fi).squeeze(-1)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(pred, target)
        opt.zero_grad(); loss.backward(); opt.step()
    def infer():
        with torch.no_grad():
            model(times, [coeffs], fi)
    print(f'{name:12s} B={B} H={H} L={L}: inference {timeit(infer):.3f} ms, training step {timeit(train_step):.3f} ms')
