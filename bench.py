#!/usr/bin/env python
"""bench.py - SDE solver row-steps/s on the BASELINE.json headline configuration.

One "step" of this benchmark = ONE forward solve (torchsde.sdeint replacement) of the K2 workload
(BASELINE.json configs[1]): Neural LNSDE (input_option 4, noise_option 17), NL=2,
B=1024 rows per GPU, H=HH=128, C=21, times=arange(101), 30 % NaN observations -> natural-spline
coefficients (1024, 100, 84), dt=1 -> N=100 Euler steps, ts=[0, 100], Brownian increments from the
in-kernel Philox generator, inputs resident in HBM.  value = rows x solver-steps x K / wall time.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: the batch shards by rows, one process per GPU, no collective inside the solver
(weak scaling: every rank solves its own 1024-row shard of a 1024*N-row global batch, Philox counters
use the global row index).  Timing: barrier + synchronize on both sides, MAX over ranks.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import stable_neural_sdes_amd as S  # noqa: E402

# K2 workload -----------------------------------------------------------------------------------------
IO, NO, NL, B, H, C, L, NSTEP = 4, 17, 2, 1024, 128, 21, 101, 100
FLOP_PER_ROWSTEP = 169_728          # SURVEY.md 8d "ALGORITHMIC flops per unit", K2 (t-only diffusion hoisted)
BYTES_PER_ROWSTEP = 346             # SURVEY.md 8d algorithmic HBM bytes, Philox + final state only
PEAK_FP32_TFLOPS = 157.3            # MI355X_MICROARCH.md: fp32 vector = fp32 MFMA peak
PEAK_HBM_GBS = 8000.0
# HBM bytes per launch of the solve kernel from rocprofv3 PMC passes (FETCH_SIZE doubled per the gfx950 correction
# in MI355X_MICROARCH.md, + WRITE_SIZE; profiles/r01_pmc_traffic.txt); re-measure when the kernel's memory behaviour changes.
HBM_TRAFFIC_BYTES_PER_LAUNCH = 38137856   # K2, MFMA M4 kernel, round 1


def build_inputs(device, rank):
    from tests.helpers import make_problem, param_spec
    pr = make_problem(1234 + rank, IO, NO, NL, B, H, C, L, nan_frac=0.3)
    # weights are replicated: every rank uses rank 0's parameter draw
    p0 = make_problem(1234, IO, NO, NL, 1, H, C, L, nan_frac=0.0)['params'] if rank else pr['params']
    flat = torch.from_numpy(np.concatenate([p0[n].reshape(-1) for n, _ in param_spec(IO, NO, NL, C, H)])).to(device)
    coeffs = torch.from_numpy(pr['coeffs']).to(device)
    y0 = torch.from_numpy(pr['y0']).to(device)
    return pr, p0, flat, coeffs, y0


def usable_cores():
    """Cores this process may actually use: affinity mask and cgroup CPU quota (os.cpu_count() alone
    over-reports inside containers and oversubscribed ATen threads are pathologically slow)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(pr, params, budget_s=12.0):
    """The reference's CPU path restated op-for-op with torch CPU tensors (oracle/torch_loop.py), timed on
    this host: Euler steps of the SAME K2 workload (B=1024 rows) for ~budget_s seconds, at the ATen
    thread count that is fastest on this host (calibrated over {1, 4, 8, 16, 32, usable cores})."""
    from oracle import torch_loop as T
    p = {k: torch.from_numpy(np.asarray(v, dtype=np.float32)) for k, v in params.items()}
    coeffs, times, y0 = torch.from_numpy(pr['coeffs']), torch.from_numpy(pr['times']), torch.from_numpy(pr['y0'])
    gen = torch.Generator().manual_seed(0)
    avail = usable_cores()
    best_threads, best_rate = 1, 0.0
    for th in sorted({1, 4, 8, 16, 32, avail}):
        if th > avail:
            continue
        torch.set_num_threads(th)
        T.euler_solve(p, IO, NO, coeffs, times, y0, 0.0, 2, 1.0, generator=gen)
        t0 = time.perf_counter()
        T.euler_solve(p, IO, NO, coeffs, times, y0, 0.0, 6, 1.0, generator=gen)
        rate = 6 / (time.perf_counter() - t0)
        if rate > best_rate:
            best_threads, best_rate = th, rate
    torch.set_num_threads(best_threads)
    chunk = 10                                  # Euler steps per timing chunk (time-of-solve varies along t)
    steps, t_begin, y, tcur = 0, time.perf_counter(), y0, 0.0
    while True:
        y = T.euler_solve(p, IO, NO, coeffs, times, y, tcur, chunk, 1.0, generator=gen)
        steps += chunk
        tcur = (tcur + chunk) % NSTEP
        if tcur == 0:
            y = y0
        el = time.perf_counter() - t_begin
        if el > budget_s or steps >= 50 * NSTEP:
            break
    return {"value": B * steps / el, "unit": "row-steps/s", "cores": best_threads, "kind": "port",
            "sample": f"{steps} Euler steps of the K2 workload (B={B} rows, = {steps / NSTEP:.1f} forward solves) "
                      f"with oracle/torch_loop.py (torch {torch.__version__} CPU fp32, {best_threads} ATen threads "
                      f"= fastest of the calibration on {avail} usable cores), {el:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--kernel', default='auto')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--exact-order', action='store_true', help='keep the unfused emb(linear_in) operation order')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 and world == 1:
        raise SystemExit('for --gpus N > 1 launch with: python -m torch.distributed.run --nnodes=1 '
                         '--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...')
    assert torch.cuda.is_available(), 'bench.py needs a GPU (there is no CPU fallback for the product path)'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1 or os.environ.get('SNSDE_BENCH_FORCE_DIST') == '1':   # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)

    pr, params, flat, coeffs, y0 = build_inputs(dev, rank)
    model = S.engine.model_struct(C, H, H, NL, IO, NO)
    grid = S.engine.step_grid(np.array([0.0, float(NSTEP)], np.float32), 1.0, pr['times'], dev)
    assert grid.N == NSTEP
    call = S.engine.SolveCall(model, flat, coeffs, grid, y0, dW=None, method='euler', seed=2024,
                              row_offset=S.sharding.shard_rows(world * B, world, rank)[0], kernel=args.kernel,
                              exact_order=args.exact_order)
    stream = torch.cuda.current_stream(dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        call.launch(stream)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        call.launch(stream)          # full call: weight pack + time table + fused solve
    barrier()
    elapsed = S.sharding.max_over_ranks(time.perf_counter() - t0, device=dev)

    # dominant kernel: the fused solve alone (prepared workspace reused), HIP events on the launch stream
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record(stream)
        call.launch(stream, reuse_prepared=True)
        b.record(stream)
    torch.cuda.synchronize(dev)
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    ys = call.ys
    assert bool(torch.isfinite(ys).all()), 'non-finite solver output'

    if rank == 0:
        rowsteps = B * NSTEP
        value = world * rowsteps * args.steps / elapsed
        ach_tf = rowsteps * FLOP_PER_ROWSTEP / (kern_ms * 1e-3) / 1e12
        ach_gbs = rowsteps * BYTES_PER_ROWSTEP / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "SDE solver steps/sec (batch x steps / s), forward solve",
            "value": value, "unit": "row-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "K2: Neural LNSDE (io=4,no=17) NL=2 B=1024/GPU H=128 C=21 L=101 natural-spline "
                                   "coeffs 30% NaN, 100 Euler steps dt=1, ts=[0,100], in-kernel Philox dW",
                       "rows_per_gpu": B, "solver_steps": NSTEP, "global_rows": world * B,
                       "parallelism": f"row-shard x{world}, no collective in the solver", "kernel": args.kernel},
            "roofline": {"bound": "mfma", "achieved": ach_tf, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach_tf / PEAK_FP32_TFLOPS, "traffic": HBM_TRAFFIC_BYTES_PER_LAUNCH,
                         "kernel_ms": kern_ms, "flop_per_rowstep": FLOP_PER_ROWSTEP,
                         "hbm_frac": ach_gbs / PEAK_HBM_GBS, "hbm_achieved_GBs": ach_gbs,
                         "note": "fp32 FMA/MFMA roof binds (intensity ~490 FLOP/B); hbm_* = algorithmic 346 B/row-step"},
        }
        if not args.no_cpu_baseline and world == 1:     # reported at N = 1 only (the other ranks would idle behind it)
            out["cpu_baseline"] = cpu_baseline(pr, params)
            out["speedup_vs_cpu"] = value / world / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
