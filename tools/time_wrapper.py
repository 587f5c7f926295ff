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
T.prepare_graph_capture(dev)

for name, B, H, C, L in (('neurallnsde', 1024, 128, 21, 101), ('naivesde', 2048, 64, 69, 72), ('neuralgsde', 512, 128, 21, 201)):
    pr = make_problem(5, 4, 17, 2, B, H, C, L, nan_frac=0.2)
    times = torch.from_numpy(pr['times']).to(dev)
    coeffs = torch.from_numpy(pr['coeffs']).to(dev)
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

    # (1) the training step recorded into one CUDA/HIP graph (device-resident Philox key => fresh noise per replay);
    #     first use of this model is on the side stream, as torch's capture recipe requires
    mg, og = build(True)
    cap_step = make_step(mg, og)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): cap_step()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cap_step()
    t_graph = timeit(g.replay)
    # (2) eager
    model, opt = build(False)
    train_step = make_step(model, opt)
    def infer():
        with torch.no_grad():
            model(times, [coeffs], fi)
    t_train, t_inf = timeit(train_step), timeit(infer)
    mf, of = build(False, fused=True)                # train.py's optimizer: single-launch Adam
    t_train_fused = timeit(make_step(mf, of))
    mfg, ofg = build(True, fused=True)
    cap2 = make_step(mfg, ofg)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): cap2()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        cap2()
    t_graph_fused = timeit(g2.replay)
    # (3) evaluation mode (BatchNorm running statistics, no dropout): eager and recorded into a graph; the bare solve beside it
    model.eval()
    t_eval = timeit(infer)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): infer()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    gi = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gi):
        infer()
    t_eval_graph = timeit(gi.replay)
    z0 = torch.zeros(B, H, device=dev)
    model.func.set_X(coeffs, times)
    def solve():
        with torch.no_grad():
            S.sdeint(model.func, z0, times, dt=1.0, method='euler', options={'row_out': fi})
    t_solve = timeit(solve)
    print(f'{name:12s} B={B} H={H} L={L}: inference {t_inf:.3f} ms, training step {t_train:.3f} ms, '
          f'graph-replayed training step {t_graph:.3f} ms; with fused Adam {t_train_fused:.3f} / {t_graph_fused:.3f} ms; eval-mode inference {t_eval:.3f} ms, graph-replayed '
          f'{t_eval_graph:.3f} ms, its solve alone {t_solve:.3f} ms')
