#!/usr/bin/env python3
"""Whole-model training step of a classification recipe, eager or replayed from one hipGraph, for rocprofv3 --kernel-trace:
    prof_wrapper_graph.py <neurallnsde|naivesde|neuralgsde> <eager|graph> [steps]
(the kernel timeline of a REPLAY is what explains a recorded step that is slower than the eager one)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stable_neural_sdes_amd as S
from stable_neural_sdes_amd import torchsde as T
from tests.helpers import make_problem
dev = torch.device('cuda:0')
name, mode = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
B, H, C, L = {'neurallnsde': (1024, 128, 21, 101), 'naivesde': (2048, 64, 69, 72), 'neuralgsde': (512, 128, 21, 201)}[name]
pr = make_problem(5, 4, 17, 2, B, H, C, L, nan_frac=0.2)
times = torch.from_numpy(pr['times']).to(dev); coeffs = torch.from_numpy(pr['coeffs']).to(dev)
fi = torch.randint(2, L, (B,), device=dev); target = (torch.rand(B, device=dev) > 0.5).float()
torch.manual_seed(0)
model, _ = S.make_sde_model(name, C, 1, H, H, 2, initial=True)
model = model.to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True, capturable=(mode == 'graph'))
T.prepare_graph_capture(dev)
def step():
    pred = model(times, [coeffs], fi).squeeze(-1)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(pred, target)
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
if mode == 'graph':
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for _ in range(steps): g.replay()
else:
    for _ in range(steps + 3): step()
torch.cuda.synchronize()
