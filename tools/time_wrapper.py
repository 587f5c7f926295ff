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

from stable_neural_sdes_amd import torchsde as T

CONFIGS = (('neurallnsde', 1024, 128, 21, 101), ('naivesde', 2048, 64, 69, 72), ('neuralgsde', 512, 128, 21, 201))
EAGER = {}


def run(name, B, H, C, L, graph_mode):
    """graph_mode False: the plain eager numbers (host Philox keys: what train.py runs without graph_steps).  True: after
    torchsde.prepare_graph_capture - the keys live on the device, the forward writes its increments out for the adjoint - the
    recorded steps and, for reference, an eager step in that mode (it is what the capture warm-up runs)."""
    pr = make_problem(5, 4, 17, 2, B, H, C, L, nan_frac=0.2)
    times = torch.from_numpy(pr['times']).to(dev)
    coeffs = torch.from_numpy(pr['coeffs']).to(dev)
    torch.manual_seed(11)
    fi = torch.randint(2, L, (B,), device=dev)
    target = (torch.rand(B, device=dev) > 0.5).float()

    def build(capturable, fused=None):
        torch.manual_seed(0)
        model, _ = S.make_sde_model(name, C, 1, H, H, 2, initial=True)
        model = model.to(dev).train()
        return model, torch.optim.Adam(model.parameters(), lr=1e-3, capturable=capturable, fused=fused)

    def make_step(model, opt):
        def train_step():
            pred = model(times, [coeffs], fi).squeeze(-1)
            loss = torch.nn.functional.binary_cross_entropy_with_logits(pred, target)
            opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
        return train_step

    def record(step):
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3): step()
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        return g

    model, opt = build(False)
    train_step = make_step(model, opt)
    def infer():
        with torch.no_grad():
            model(times, [coeffs], fi)
    if not graph_mode:
        t_train, t_inf = timeit(train_step), timeit(infer)
        mf, of = build(False, fused=True)                # train.py's optimizer: single-launch Adam
        t_train_fused = timeit(make_step(mf, of))
        model.eval()
        t_eval = timeit(infer)
        z0 = torch.zeros(B, H, device=dev)
        model.func.set_X(coeffs, times)
        def solve():
            with torch.no_grad():
                S.sdeint(model.func, z0, times, dt=1.0, method='euler', options={'row_out': fi})
        EAGER[name] = (t_inf, t_train, t_train_fused, t_eval, timeit(solve))
        return
    mg, og = build(True)
    t_graph = timeit(record(make_step(mg, og)).replay)
    mfg, ofg = build(True, fused=True)
    t_graph_fused = timeit(record(make_step(mfg, ofg)).replay)
    t_train_gm = timeit(train_step)                      # an eager step in graph mode (device keys)
    model.eval()
    t_eval_graph = timeit(record(infer).replay)
    t_inf, t_train, t_train_fused, t_eval, t_solve = EAGER[name]
    print(f'{name:12s} B={B} H={H} L={L}: inference {t_inf:.3f} ms, eager training step {t_train:.3f} ms (fused Adam {t_train_fused:.3f}), '
          f'graph-replayed training step {t_graph:.3f} ms (fused Adam {t_graph_fused:.3f}; an eager step in graph mode {t_train_gm:.3f}); '
          f'eval-mode inference {t_eval:.3f} ms, graph-replayed {t_eval_graph:.3f} ms, its solve alone {t_solve:.3f} ms')


for cfg in CONFIGS:
    run(*cfg, graph_mode=False)
T.prepare_graph_capture(dev)
for cfg in CONFIGS:
    run(*cfg, graph_mode=True)
