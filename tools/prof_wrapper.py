#!/usr/bin/env python3
"""Eager whole-model training steps of the K2 classification recipe (single-launch Adam, as train.py), for rocprofv3
--kernel-trace --stats;  `prof_wrapper.py N eval`: N evaluation-mode inference passes instead (fused initial state + head)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stable_neural_sdes_amd as S
from tests.helpers import make_problem
dev = torch.device('cuda:0')
B, H, C, L = 1024, 128, 21, 101
pr = make_problem(5, 4, 17, 2, B, H, C, L, nan_frac=0.2)
times = torch.from_numpy(pr['times']).to(dev); coeffs = torch.from_numpy(pr['coeffs']).to(dev)
fi = torch.randint(2, L, (B,), device=dev); target = (torch.rand(B, device=dev) > 0.5).float()
torch.manual_seed(0)
model, _ = S.make_sde_model('neurallnsde', C, 1, H, H, 2, initial=True)
model = model.to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
if len(sys.argv) > 2 and sys.argv[2] == 'eval':
    model.eval()
    with torch.no_grad():
        for _ in range(int(sys.argv[1])):
            model(times, [coeffs], fi)
    torch.cuda.synchronize()
    sys.exit(0)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    pred = model(times, [coeffs], fi).squeeze(-1)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(pred, target)
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
torch.cuda.synchronize()
