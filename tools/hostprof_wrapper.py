#!/usr/bin/env python3
"""cProfile of the host side of the EAGER whole-model training step (NeuralSDE wrapper + readout + BCE + backward + fused Adam):
    hostprof_wrapper.py [neurallnsde|naivesde|neuralgsde]
Prints the wall time per step, then the functions by own time and by cumulative time."""
import cProfile, os, pstats, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stable_neural_sdes_amd as S
from tests.helpers import make_problem
dev = torch.device('cuda:0')
name = sys.argv[1] if len(sys.argv) > 1 else 'neurallnsde'
B, H, C, L = {'neurallnsde': (1024, 128, 21, 101), 'naivesde': (2048, 64, 69, 72), 'neuralgsde': (512, 128, 21, 201)}[name]
pr = make_problem(5, 4, 17, 2, B, H, C, L, nan_frac=0.2)
times = torch.from_numpy(pr['times']).to(dev); coeffs = torch.from_numpy(pr['coeffs']).to(dev)
fi = torch.randint(2, L, (B,), device=dev); target = (torch.rand(B, device=dev) > 0.5).float()
torch.manual_seed(0)
model, _ = S.make_sde_model(name, C, 1, H, H, 2, initial=True)
model = model.to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
def step():
    pred = model(times, [coeffs], fi).squeeze(-1)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(pred, target)
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
for _ in range(10): step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(50): step()
t_host = (time.perf_counter() - t) / 50 * 1e3
torch.cuda.synchronize(); t_all = (time.perf_counter() - t) / 50 * 1e3
print(f'{name}: eager step {t_all:.3f} ms wall, host enqueue {t_host:.3f} ms')
def fwd_only():
    with torch.enable_grad():
        return model(times, [coeffs], fi)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(50): fwd_only()
print(f'  model.forward (grad mode) host enqueue {(time.perf_counter() - t) / 50 * 1e3:.3f} ms')
torch.cuda.synchronize()
pf = cProfile.Profile(); pf.enable()
for _ in range(50): step()
pf.disable(); torch.cuda.synchronize()
st = pstats.Stats(pf); st.sort_stats('tottime').print_stats(28)
st.sort_stats('cumulative').print_stats(28)
