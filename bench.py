#!/usr/bin/env python
"""bench.py - SDE solver row-steps/s on the BASELINE.json headline configuration.

One "step" of this benchmark = ONE forward solve (torchsde.sdeint replacement) of the K2 workload
(BASELINE.json configs[1]): Neural LNSDE (input_option 4, noise_option 17), NL=2,
B=1024 rows per GPU, H=HH=128, C=21, times=arange(101), 30 % NaN observations -> natural-spline
coefficients (1024, 100, 84), dt=1 -> N=100 Euler steps, ts=[0, 100], Brownian increments from the
in-kernel Philox generator, inputs resident in HBM.  value = rows x solver-steps x K / wall time.

    python bench.py [--gpus N] [--steps K] [--warmup W]

--gpus N > 1 run as plain `python bench.py --gpus N` re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (one process per GPU over
RCCL) and passes rank 0's JSON line through; launched by torchrun it reads RANK / LOCAL_RANK / WORLD_SIZE.

Multi-GPU: the batch shards by rows, one process per GPU, no collective inside the solver.  The headline `value`
is weak scaling (every rank solves its own 1024-row shard of a 1024*N-row global batch, Philox counters use the
global row index).  Timing of `value`: barrier + synchronize on both sides of K solves, MAX over ranks.
`timing` = HIP events around single solves (median of >= 50 after 10 warm-ups, p10 / p90).
`extra` carries the two fixed-global-batch (strong scaling) configurations of BASELINE.json: K3 (GSDE, 4096 rows,
200 steps, rows split N ways) and K5 (Milstein + fused adjoint, 1024 rows, H=256, rows split N ways, gradient
all-reduce over RCCL).
"""
import argparse
import json
import os
import socket
import subprocess
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
# what the kernel EXECUTES per row-step: emb o linear_in / emb o initial_network are pre-multiplied (one layer fewer) and
# the time features share the control path's k-block: 104 v_mfma_f32_4x4x1_16b (512 FLOP) per wave-step x 8 waves / 4 rows
EXECUTED_FLOP_PER_ROWSTEP = 104 * 512 * 8 // 4
MFMA_CYCLES_PER_STEP = 104 * 2 * 8  # per SIMD: two waves x 104 MFMAs x 8 cycles (the kernel's own MFMA-issue floor)
BYTES_PER_ROWSTEP = 346             # SURVEY.md 8d algorithmic HBM bytes, Philox + final state only
PEAK_FP32_TFLOPS = 157.3            # MI355X_MICROARCH.md: fp32 vector = fp32 MFMA peak
PEAK_HBM_GBS = 8000.0
# HBM bytes per launch of the solve kernel from rocprofv3 PMC passes (FETCH_SIZE doubled per the gfx950 correction
# in MI355X_MICROARCH.md, + WRITE_SIZE; profiles/r06_pmc_counters.txt); re-measure when the kernel's memory behaviour changes.
HBM_TRAFFIC_BYTES_PER_LAUNCH = 38070272   # K2, lean M4 kernel: (2 x 18077.0 + 1024.0) KB
HBM_TRAFFIC_KERNEL = ('lean', 'snsde_m4_kernel<CfgL<128, 1, 2, 1, 0, 0>>')     # the path / instantiation the profile was taken on
HBM_TRAFFIC_SOURCE = ("profiles/r06_pmc_counters.txt: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes over this bench command, "
                      "mean of 187 dispatches of the solve kernel), 2 x FETCH_SIZE + WRITE_SIZE per the gfx950 correction of "
                      "MI355X_MICROARCH.md; a constant of the kernel's memory behaviour, not re-measured in this run")
# L2-fabric bytes of one K2 training step (forward + adjoint + weight gradients): profiles/r06_train_traffic.txt (re-measured in round 6)
TRAIN_TRAFFIC_FILE = 'r06_train_traffic.txt'
TRAIN_TRAFFIC_SOURCE = ("profiles/r06_train_traffic.txt (tools/pmc_train_modes.sh: FETCH_SIZE / WRITE_SIZE summed over every kernel of 10 "
                        "steps; a constant of the path's memory behaviour taken under rocprofv3, not re-measured in this run - valid while the "
                        "forward path reported beside it is the profiled one)")
TRAIN_TRAFFIC_PATHS = ('lean', 1)          # forward path / backward mode the profile was taken on


def flops_drift(io, h, c, nl):
    """algorithmic FLOPs of one drift evaluation per row (2 per MAC; HH = H)."""
    emb, usex, timef = io in (2, 4, 6), io in (0, 2, 4, 6), io >= 3
    f = 2 * c * h if usex else 0
    f += 2 * (h + (2 if timef else 0)) * h if io != 0 else 0
    f += 2 * (2 * h) * h if emb else 0
    return f + 2 * (nl - 1) * h * h + 2 * h * h


def flops_net(no, h):
    """One evaluation of the diffusion net on [sin t, cos t, y] per row (noise_option 14/15: one layer, 18/19: two)."""
    return (2 * (h + 2) * h + (2 * h * h if no in (18, 19) else 0)) if no in (14, 15, 18, 19) else 0


def roofline_obj(flops, seconds, what):
    tf = flops / seconds / 1e12
    return {"bound": "mfma", "achieved": tf, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_FP32_TFLOPS,
            "flops": flops, "seconds": seconds, "counted": what}


def build_inputs(device, rank, io=IO, no=NO, nl=NL, b=B, h=H, c=C, l=L, nan_frac=0.3, hermite=False):
    from tests.helpers import make_problem, param_spec
    pr = make_problem(1234 + rank, io, no, nl, b, h, c, l, nan_frac=nan_frac, hermite=hermite)
    # weights are replicated: every rank uses rank 0's parameter draw
    p0 = make_problem(1234, io, no, nl, 1, h, c, l, nan_frac=0.0)['params'] if rank else pr['params']
    flat = torch.from_numpy(np.concatenate([p0[n].reshape(-1) for n, _ in param_spec(io, no, nl, c, h)])).to(device)
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
    def timed(threads, budget):
        torch.set_num_threads(threads)
        chunk = 10                                  # Euler steps per timing chunk (time-of-solve varies along t)
        steps, t_begin, y, tcur = 0, time.perf_counter(), y0, 0.0
        while True:
            y = T.euler_solve(p, IO, NO, coeffs, times, y, tcur, chunk, 1.0, generator=gen)
            steps += chunk
            tcur = (tcur + chunk) % NSTEP
            if tcur == 0:
                y = y0
            el = time.perf_counter() - t_begin
            if el > budget or steps >= 100 * NSTEP:
                return steps, el
    steps1, el1 = (timed(1, 0.25 * budget_s) if best_threads != 1 else (0, 0.0))       # all cores AND one thread
    steps, el = timed(best_threads, budget_s - el1)
    if best_threads == 1:
        steps1, el1 = steps, el
    return {"value": B * steps / el, "unit": "row-steps/s", "cores": best_threads, "threads": best_threads, "usable_cores": avail,
            "value_1thread": B * steps1 / el1, "kind": "port",
            "sample": f"{steps}+{steps1} Euler steps of K2 (B={B}) in {el:.1f}+{el1:.1f} s, oracle/torch_loop.py, torch {torch.__version__} CPU f32",
            "note": "port = the reference's f/g + Euler loop on torch CPU ops with torch.randn increments; real torchsde adds "
                    "BrownianInterval overhead, so this baseline is on the fast side"}


def event_times_ms(fn, stream, n, warm):
    """HIP-event duration of n calls of fn() on `stream` (each call bracketed by its own event pair) after `warm` calls."""
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(stream)
        fn()
        b.record(stream)
    torch.cuda.synchronize(stream.device)
    return np.array([a.elapsed_time(b) for a, b in ev])


def spread(t_ms):
    return {"n": int(len(t_ms)), "median_ms": float(np.median(t_ms)), "p10_ms": float(np.percentile(t_ms, 10)),
            "p90_ms": float(np.percentile(t_ms, 90))}


def strong_k3(dev, rank, world, stream, barrier, maxr, shard_of=None):
    """BASELINE config 3: Neural GSDE (6, 17), 4096 rows GLOBAL split over the ranks, H=128, C=21, Hermite coefficients
    without missing values, times=arange(201) -> 200 Euler steps, ts=[0, 200]; forward solve, in-kernel Philox.
    shard_of = 8: ONE rank's share of an 8-GPU run (512 rows), timed on this GPU alone - what each GPU of the strong-scaling
    run executes (no collective in the solver), so that DESIGN 5's 8-GPU prediction can be checked against a SCALE run."""
    rows_g, n_steps = 4096, 200
    lo, hi = S.sharding.shard_rows(rows_g, shard_of or world, 0 if shard_of else rank)
    pr, _, flat, coeffs, y0 = build_inputs(dev, rank, io=6, no=17, b=hi - lo, l=n_steps + 1, nan_frac=0.0, hermite=True)
    model = S.engine.model_struct(C, H, H, NL, 6, 17)
    grid = S.engine.step_grid(np.array([0.0, float(n_steps)], np.float32), 1.0, pr['times'], dev)
    call = S.engine.SolveCall(model, flat, coeffs, grid, y0, method='euler', seed=2024, row_offset=lo)
    for _ in range(10):
        call.launch(stream)
    barrier()
    k, t0 = 30, time.perf_counter()
    for _ in range(k):
        call.launch(stream)
    barrier()
    el = maxr(time.perf_counter() - t0)
    t_kern = event_times_ms(lambda: call.launch(stream, reuse_prepared=True), stream, 20, 3)
    kern_s = float(np.median(t_kern)) * 1e-3
    done = (hi - lo) if shard_of else rows_g
    return {"workload": f"K3: Neural GSDE (io=6,no=17) {rows_g} rows global ({hi - lo}/GPU), H=128, 200 Euler steps, "
                        "Hermite coeffs, forward" + (f"; ONE rank's shard of a {shard_of}-GPU run on this GPU" if shard_of else ""),
            "scaling": "strong", "value": done * n_steps * k / el, "kernel_ms": kern_s * 1e3,
            "unit": "row-steps/s", "ms_per_solve": el / k * 1e3, "solves": k,
            "roofline": roofline_obj((hi - lo) * n_steps * flops_drift(6, H, C, NL), kern_s,
                                     "this rank's solve kernel (HIP events, median of 20): SURVEY 8d algorithmic FLOPs of the drift, "
                                     "169 728 per row-step; the time-only diffusion is hoisted")}


def strong_k5(dev, rank, world, stream, barrier, maxr, dist, shard_of=None):
    """BASELINE config 5: Milstein + fused adjoint, MuJoCo-forecast-shaped: LNSDE (4, 17), 1024 rows GLOBAL split over the
    ranks, H=256, C=14, L=50 knots with dropped rows, 49 steps, every knot an output; loss = mean square of the last 10
    states; forward + backward + gradient all-reduce (RCCL) per step."""
    from tests.helpers import make_problem, param_spec
    rows_g, hh, cc, ll = 1024, 256, 14, 50
    lo, hi = S.sharding.shard_rows(rows_g, shard_of or world, 0 if shard_of else rank)      # shard_of: see strong_k3
    pr = make_problem(4321 + rank, 4, 17, 2, hi - lo, hh, cc, ll, nan_frac=0.3)
    p0 = make_problem(4321, 4, 17, 2, 1, hh, cc, ll, nan_frac=0.0)['params']
    sde = S.Diffusion_model(cc, hh, hh, 2, input_option=4, noise_option=17).to(dev)
    with torch.no_grad():
        for name, p in sde.named_parameters():
            p.copy_(torch.from_numpy(np.asarray(p0[name], np.float32)).reshape(p.shape))
    times = torch.from_numpy(pr['times']).to(dev)
    sde.set_X(torch.from_numpy(pr['coeffs']).to(dev), times)
    y0 = torch.from_numpy(pr['y0']).to(dev)
    params = [p for p in sde.parameters()]

    def step():
        for p in params:
            p.grad = None
        ys = S.torchsde.sdeint(sde, y0, times, dt=1.0, method='milstein', options={'seed': 7, 'row_offset': lo})
        loss = ys[-10:].square().mean() * ((hi - lo) / rows_g)
        loss.backward()
        if dist is not None and not shard_of:
            flat = torch.cat([p.grad.reshape(-1) for p in params])
            dist.all_reduce(flat)
        return loss

    for _ in range(5):
        step()
    # (two timed batches, the faster one reported: a single host hiccup - allocator growth, a page fault storm on a fresh box - inside
    #  one 20-step batch showed up once as 5.4 ms per step on a leg that runs at 0.86; the headline `value` keeps its single region)
    k, el = 20, float('inf')
    for _ in range(2):
        barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        barrier()
        el = min(el, maxr(time.perf_counter() - t0))
    done = (hi - lo) if shard_of else rows_g
    return {"workload": f"K5: LNSDE (io=4,no=17) Milstein + fused adjoint, {rows_g} rows global ({hi - lo}/GPU), H=256, "
                        "C=14, 49 steps, 50 outputs, fwd+bwd" + (f"; ONE rank's shard of a {shard_of}-GPU run on this GPU, no all-reduce"
                                                                   if shard_of else "+grad all-reduce"), "scaling": "strong",
            "value": done * (ll - 1) * k / el, "unit": "row-steps/s (training steps)", "ms_per_step": el / k * 1e3,
            "steps": k,
            "roofline": roofline_obj(3 * (hi - lo) * (ll - 1) * flops_drift(4, hh, cc, 2), el / k,
                                     "whole training step of this rank (host + forward + adjoint + weight gradients + all-reduce, wall "
                                     "clock): 3 x the forward's algorithmic FLOPs (SURVEY 8d: 663 552 per row-step forward at K5)")}


def _module(dev, io, no, rows, hh, cc, ll, seed, nan_frac=0.3):
    from tests.helpers import make_problem
    pr = make_problem(seed, io, no, 2, rows, hh, cc, ll, nan_frac=nan_frac)
    sde = S.Diffusion_model(cc, hh, hh, 2, input_option=io, noise_option=no)
    sde.load_state_dict({k: torch.from_numpy(np.asarray(v, np.float32).copy()) for k, v in pr['params'].items()})
    sde = sde.to(dev)
    times = torch.from_numpy(pr['times']).to(dev)
    sde.set_X(torch.from_numpy(pr['coeffs']).to(dev), times)
    return sde, times, torch.from_numpy(pr['y0']).to(dev)


def solve_kernel_ms(sde, y0, ts_host, method, stream, n=20, training=False, noise_table=None, model=None, flat=None, dt=1.0):
    """HIP-event times of the SOLVE LAUNCH alone: a prepared engine.SolveCall relaunched with REUSE_PREPARED (no weight packing, no
    host plumbing) - the kernel time the roofline fractions of the extra legs are computed from."""
    if model is None:
        model, layout, numel = S.engine.recognise(sde)
        flat = S.engine.flatten_params(sde, layout, numel, y0.device)
    grid = S.engine.step_grid(ts_host, dt, S.torchsde._HostTimes.get(sde.times), y0.device)
    call = S.engine.SolveCall(model, flat, sde.coeffs, grid, y0, method=method, seed=5, save_traj=training, save_act=training,
                              noise_table=noise_table)
    call.launch(stream)
    return event_times_ms(lambda: call.launch(stream, reuse_prepared=True), stream, n, 3)


def graphed_step_ms(sde, params, y0, ts, method, dev, stream):
    """The same forward + backward recorded into one HIP graph and replayed (what train.main does by default: GraphedStep; the
    eager call is bound by the host's ~0.4 ms of Python between the launches at these sizes).  Increments from the device-resident
    Philox key (fresh on every replay).  None where the capture is refused."""
    try:
        S.torchsde.prepare_graph_capture(dev)
        opts = {'strict': True}
        static_y0 = y0.clone()

        def step():
            for p in params:
                p.grad = None
            yy = static_y0.clone().requires_grad_(True)
            S.torchsde.sdeint(sde, yy, ts, dt=1.0, method=method, options=opts)[-1].square().mean().backward()
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        return event_times_ms(graph.replay, torch.cuda.current_stream(dev), 20, 5)
    except Exception as exc:      # noqa: BLE001 - reported in the leg, the eager numbers stand
        sys.stderr.write(f"graph replay leg skipped: {exc!r}\n")
        return None


def train_leg(dev, stream, io, no, rows, hh, cc, ll, method, label, outputs='ends'):
    """sdeint forward and forward + backward (fused adjoint + native weight-gradient pass) of one Diffusion_model shape:
    HIP-event medians of the whole calls AND of the solve launch alone; the forward's roofline fraction is computed from the
    kernel time (comparable with the headline's), the whole-call fraction is kept beside it."""
    sde, times, y0 = _module(dev, io, no, rows, hh, cc, ll, 77)
    ts = times if outputs == 'knots' else times[[0, -1]]
    params = list(sde.parameters())
    opts = {'seed': 5, 'strict': True}

    def fwd():
        with torch.no_grad():
            S.torchsde.sdeint(sde, y0, ts, dt=1.0, method=method, options=opts)

    def step():
        for p in params:
            p.grad = None
        yy = y0.clone().requires_grad_(True)
        S.torchsde.sdeint(sde, yy, ts, dt=1.0, method=method, options=opts)[-1].square().mean().backward()

    t_f = event_times_ms(fwd, stream, 20, 5)
    t_s = event_times_ms(step, stream, 20, 5)
    t_g = graphed_step_ms(sde, params, y0, ts, method, dev, stream)
    t_k = solve_kernel_ms(sde, y0, ts.cpu().numpy(), method, stream)
    t_kt = solve_kernel_ms(sde, y0, ts.cpu().numpy(), method, stream, training=True)
    n = ll - 1
    per_eval = flops_drift(io, hh, cc, 2)
    fl = (3 * per_eval + 4 * flops_net(no, hh)) if method == 'srk' else per_eval + flops_net(no, hh) * (2 if method == 'milstein' else 1)      # Milstein: + the VJP through the net
    model = S.engine.model_struct(cc, hh, hh, 2, io, no)
    out = {"workload": f"{label}: Diffusion_model (io={io},no={no}) NL=2, {rows} rows, H={hh}, C={cc}, {n} {method} steps, Philox increments",
           "forward_path": S.engine.forward_path(model, rows, ll, n, method=method),
           "backward_mode": S.engine.backward_mode(model, rows, ll, S.engine.step_grid(ts.cpu().numpy(), 1.0, times.cpu().numpy(), dev), method),
           "forward": spread(t_f), "forward_backward": spread(t_s),
           "forward_backward_graph_replay": spread(t_g) if t_g is not None else None,
           "forward_kernel": spread(t_k), "training_mode_forward_kernel": spread(t_kt),
           "value": rows * n / (float(np.median(t_f)) * 1e-3), "unit": "row-steps/s (forward)",
           "flop_per_rowstep": fl,
           "roofline_forward": roofline_obj(rows * n * fl, float(np.median(t_k)) * 1e-3, "the solve launch alone (HIP events around a "
                                            "REUSE_PREPARED relaunch: same basis as the headline's roofline.frac); algorithmic FLOPs per "
                                            "step: drift evaluations + diffusion-net evaluations of the scheme"),
           "roofline_forward_whole_call": roofline_obj(rows * n * fl, float(np.median(t_f)) * 1e-3, "whole sdeint() forward call incl. "
                                                       "weight packing and host plumbing (HIP events)"),
           "roofline_training": roofline_obj(3 * rows * n * fl, float(np.median(t_s)) * 1e-3, "whole forward + backward call: 3 x forward FLOPs")}
    return out


def k2_training(dev, stream):
    """K2 training step (forward + fused adjoint + native weight gradients), with the bytes-based roofline of its L2-fabric
    traffic (from the committed PMC profile; the step is neither FLOP- nor HBM-bound but a chain of latency-bound launches)."""
    out = train_leg(dev, stream, IO, NO, B, H, C, L, 'euler', 'K2 training step')
    traffic = None
    path = os.path.join(ROOT, 'profiles', TRAIN_TRAFFIC_FILE)
    if os.path.exists(path):
        for line in open(path):
            if line.startswith(('current default', 'saved activations')):
                traffic = float(line.split('total')[1].split('MB')[0]) * 1e6
    sec = out["forward_backward"]["median_ms"] * 1e-3
    if traffic and (out["forward_path"], out["backward_mode"]) == TRAIN_TRAFFIC_PATHS:      # (the profiled kernels are the launched ones)
        out["roofline_bytes"] = {"bound": "hbm", "achieved": traffic / sec / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": traffic / sec / 1e9 / PEAK_HBM_GBS, "traffic": traffic, "traffic_source": TRAIN_TRAFFIC_SOURCE}
    elif traffic:
        out["roofline_bytes"] = {"error": f"forward path / backward mode {(out['forward_path'], out['backward_mode'])} differ from the "
                                          f"profiled {TRAIN_TRAFFIC_PATHS}: the constant of {TRAIN_TRAFFIC_FILE} does not apply"}
    return out


def tutorial_field(dev, stream, kind='lnsde', rows=1024, hh=128, n=100):
    """Tutorial-style field (reference tutorial/*.ipynb cell 7: LipSwish MLPs, raw time, g(t) [* y]) through torchsde.sdeint:
    the fused composed-field path (fields.py) next to the generic graph-captured stepper on the same module.  Defaults: the
    K2-sized case (1024 rows, H=128, 100 Euler steps); BASELINE config 0 is (kind='lsde', 256 rows, H=32, 50 steps)."""
    from tests.helpers import make_problem
    from tests.tutorial_fields import TutorialField
    cc = 2
    times = np.linspace(0.0, 1.0, 11).astype(np.float32)
    pr = make_problem(99, 4, 17, 2, rows, hh, cc, len(times), times=times)
    torch.manual_seed(99)
    field = TutorialField(kind, cc, hh, 1).to(dev)
    tt = torch.from_numpy(times).to(dev)
    field.set_X(torch.from_numpy(pr['coeffs']).to(dev), tt)
    y0 = torch.from_numpy(pr['y0']).to(dev)
    ts = tt[[0, -1]]
    with torch.no_grad():
        fused = event_times_ms(lambda: S.sdeint(field, y0, ts, dt=1.0 / n, method='euler', options={'seed': 1}), stream, 30, 5)
        assert S.fields.compose(field).verified.get(str(y0.device)) is True, 'tutorial field did not take the fused path'
        generic = event_times_ms(lambda: S.sdeint(field, y0, ts, dt=1.0 / n, method='euler',
                                                  options={'seed': 1, 'backend': 'torch'}), stream, 5, 2)
    params = list(field.parameters())

    def train_step(backend):
        def fn():
            for p in params:
                p.grad = None
            ys = S.sdeint(field, y0, ts, dt=1.0 / n, method='euler', options={'seed': 1, 'backend': backend})
            ys[-1].square().mean().backward()
        return fn
    fused_train = event_times_ms(train_step('auto'), stream, 20, 5)
    loop_train = event_times_ms(train_step('torch'), stream, 2, 1)
    # the whole optimisation step (forward, loss, backward, Adam) recorded into ONE hipGraph and replayed: the composed path is
    # capturable in training as well (fresh increments per replay from the device-resident Philox key)
    graphed = None
    try:
        S.torchsde.prepare_graph_capture(dev)
        opt = torch.optim.Adam(field.parameters(), lr=1e-4, capturable=True)

        def opt_step():
            ys = S.sdeint(field, y0, ts, dt=1.0 / n, method='euler')
            ys[-1].square().mean().backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
        for _ in range(3):
            opt_step()
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):       # (records on torch's capture stream; replays go to the current one)
            opt_step()
        graphed = spread(event_times_ms(graph.replay, stream, 30, 5))
        eager_opt = spread(event_times_ms(opt_step, stream, 20, 3))
    except Exception as exc:      # (reported, never fatal for the bench line)
        graphed, eager_opt = {"error": f"{type(exc).__name__}: {exc}"}, None
    cf = S.fields.compose(field)
    ts_host = ts.cpu().numpy()
    grid = S.engine.step_grid(ts_host, 1.0 / n, times, dev)
    with torch.no_grad():
        flat_c, tab_c = cf.inference_inputs(grid.d_t0, dev)
    t_k = solve_kernel_ms(field, y0, ts_host, 'euler', stream, model=cf.model, flat=flat_c, noise_table=tab_c, dt=1.0 / n)
    fl = flops_drift(cf.model.input_option, hh, cc, cf.model.num_hidden_layers)      # the composed block the kernel evaluates
    return {"workload": f"tutorial Neural{kind.upper()}Func-shaped field (LipSwish, num_layers=1), {rows} rows, H={hh}, C={cc}, {n} Euler "
                        "steps, whole sdeint() call incl. weight composition + noise table",
            "forward_path": S.engine.forward_path(cf.model, rows, len(times), n, table=True),
            "fused_kernel": spread(t_k), "flop_per_rowstep": fl,
            "roofline": roofline_obj(rows * n * fl, float(np.median(t_k)) * 1e-3, "the solve launch alone; FLOPs of the composed "
                                     "(input_option, NL) block the kernel evaluates per step (the time-only diffusion is a per-solve table)"),
            "roofline_whole_call": roofline_obj(rows * n * fl, float(np.median(fused)) * 1e-3, "whole sdeint() call incl. composition + table"),
            "fused": spread(fused), "generic_graph_stepper": spread(generic),
            "fused_forward_backward": spread(fused_train), "tensor_loop_forward_backward": spread(loop_train),
            "optimizer_step_eager": eager_opt, "optimizer_step_graph_replayed": graphed,
            "value": rows * n / (float(np.median(fused)) * 1e-3), "unit": "row-steps/s"}


def latent_sde(dev, stream, rows=1024, hidden=32, L=50):
    """torch-ists' LatentSDE shape (tests/latent_field.LatentField: the reference wrapper's forward - spline start,
    sdeint_adjoint(names f_aug / g_aug), KL) under its default `srk`: the fused solve with the KL accumulator as a state column
    (torchsde._sdeint_latent, snsde_solve.kl_column1) next to the tensor-op loop, inference and one training step (HIP events)."""
    from tests.latent_field import LatentField
    torch.manual_seed(1)
    m = LatentField(4, hidden, hidden, 2).to(dev)
    times = torch.linspace(0, 1, L, device=dev)
    X = torch.cumsum(0.2 * torch.randn(rows, L, 4, device=dev), dim=1)
    coeffs = S.torchcde.hermite_cubic_coefficients_with_backward_differences(X, times)
    out = {"workload": f"LatentSDE-shaped module ({hidden - 1} latent channels + KL accumulator, 2 hidden layers), {rows} rows, {L} output times, "
                       "srk, whole wrapper forward / training step"}
    for backend, key, n in (('auto', 'fused', 10), ('torch', 'tensor_loop', 2)):
        opts = {'backend': backend}      # (unseeded: the fused path draws Philox increments in the kernel, the loop torch.randn)
        with torch.no_grad():
            out[key + "_forward"] = spread(event_times_ms(lambda: m(coeffs, times, method='srk', options=opts), stream, n, 2))

        def step():
            m.zero_grad(set_to_none=True)
            o, latent, kl = m(coeffs, times, method='srk', options=opts)
            (o.square().mean() + 1e-3 * kl).backward()
        out[key + "_training_step"] = spread(event_times_ms(step, stream, n, 2))
    hl, hh_ = hidden - 1, hidden
    fl = 3 * 2 * ((hl + 2) * hh_ + hh_ * hh_ + hh_ * hl)       # SRID2: three drift evaluations of the posterior MLP per step (constant diffusion)
    steps = L - 1
    out["flop_per_rowstep"] = fl
    out["roofline"] = roofline_obj(rows * steps * fl, out["fused_forward"]["median_ms"] * 1e-3,
                                   "whole wrapper forward (spline start, solve, KL): the module's algorithmic drift FLOPs; "
                                   "at 31 latent channels the solve is latency-bound and the call is dominated by launches, not FLOPs")
    return out


def summary_of(out, extra):
    """Compact last key of the line: per BASELINE leg the kernel time, the roofline fraction computed from it, and the training
    step where the leg has one (ms, HIP-event medians; `frac` = algorithmic FLOPs / kernel time / 157.3 TF)."""
    s = {"K2_forward": {"kernel_ms": round(out["roofline"]["kernel_ms"], 4), "frac": round(out["roofline"]["frac"], 4),
                        "call_ms": round(out["ms_per_step"], 4)}}
    for key in ("K2_train", "K4_3_18_euler", "NSDE_3_18_srk_K4_shape", "NSDE_3_18_milstein_K4_shape"):
        e = extra.get(key)
        if e:
            s[key] = {"kernel_ms": round(e["forward_kernel"]["median_ms"], 4), "frac": round(e["roofline_forward"]["frac"], 4),
                      "fwd_bwd_ms": round(e["forward_backward"]["median_ms"], 4), "train_frac": round(e["roofline_training"]["frac"], 4)}
            if e.get("forward_backward_graph_replay"):
                s[key]["fwd_bwd_graph_ms"] = round(e["forward_backward_graph_replay"]["median_ms"], 4)
            if "roofline_bytes" in e and "traffic" in e["roofline_bytes"]:
                s[key]["train_MB"] = round(e["roofline_bytes"]["traffic"] / 1e6, 1)
    for key in ("K3_strong", "K3_shard_512"):
        e = extra.get(key)
        if e:
            s[key] = {"kernel_ms": round(e["kernel_ms"], 4), "frac": round(e["roofline"]["frac"], 4), "call_ms": round(e["ms_per_solve"], 4)}
    for key in ("K5_strong_train", "K5_shard_128_train"):
        e = extra.get(key)
        if e:
            s[key] = {"step_ms": round(e["ms_per_step"], 4), "train_frac": round(e["roofline"]["frac"], 4)}
    for key in ("K1_tutorial_lsde", "tutorial_field"):      # BASELINE config 0 (the OU tutorial's LSDE field) and the K2-sized LNSDE field
        e = extra.get(key)
        if e and "fused" in e:
            s[key] = {"fwd_ms": round(e["fused"]["median_ms"], 4), "fwd_bwd_ms": round(e["fused_forward_backward"]["median_ms"], 4),
                      "step_graph_ms": round(e["optimizer_step_graph_replayed"]["median_ms"], 4)}
    return s


LINE_LIMIT = 4096      # the driver keeps a bounded tail of stdout: r05's 21 KB line did not parse


def compact_line(out):
    """The ONE JSON line of the bench contract: contract keys, the roofline / cpu_baseline objects and the per-leg `summary` - numbers
    only, no prose (the full record with every `extra` leg, timing spreads and provenance notes goes to a file, see main)."""
    r = out["roofline"]
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data")}
    line["config"] = {k: out["config"][k] for k in ("workload", "rows_per_gpu", "solver_steps", "global_rows", "parallelism", "kernel",
                                                    "prepare")}
    line["roofline"] = {"bound": r["bound"], "achieved": round(r["achieved"], 3), "peak": r["peak"], "unit": r["unit"],
                        "frac": round(r["frac"], 4), "traffic": r["traffic"], "kernel_ms": round(r["kernel_ms"], 5),
                        "executed_frac": round(r["executed_frac"], 4), "flop_per_rowstep": r["flop_per_rowstep"],
                        "bytes_per_rowstep": r["bytes_per_rowstep"], "hbm_frac": round(r["hbm_frac"], 5),
                        "launched_path": r["launched_path"]}
    if "cpu_baseline" in out:
        c = out["cpu_baseline"]
        line["cpu_baseline"] = {"value": round(c["value"], 1), "unit": c["unit"], "cores": c["cores"], "kind": c["kind"],
                                "threads": c["threads"], "value_1thread": round(c["value_1thread"], 1), "sample": c["sample"]}
        line["speedup_vs_cpu"] = round(out["speedup_vs_cpu"], 1)
    if "summary" in out:
        line["summary"] = out["summary"]
    if out.get("full_record"):
        line["full_record"] = out["full_record"]
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    assert len(text) < LINE_LIMIT, f"bench line is {len(text)} bytes (limit {LINE_LIMIT})"
    return text


def write_full_record(out):
    """Everything the line leaves out (extra legs, timing spreads, provenance notes): gpurun_out/bench_full.json beside the repo copy
    (merged back by gpurun; copied to profiles/rNN_bench_full.json when it is to be judged).  Returns the relative path or None."""
    rel = os.path.join("gpurun_out", "bench_full.json")
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, rel), "w") as f:
            json.dump(out, f, indent=1, allow_nan=True)
        return rel
    except OSError:
        return None


def self_launch(args):
    """`python bench.py --gpus N` without torchrun: re-execute under torch.distributed.run, one rank per GPU."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--kernel', default='auto')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip the K3 / K5 strong-scaling legs')
    ap.add_argument('--exact-order', action='store_true', help='keep the unfused emb(linear_in) operation order')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_launch(args)
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    assert torch.cuda.is_available(), 'bench.py needs a GPU (there is no CPU fallback for the product path)'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1 or os.environ.get('SNSDE_BENCH_FORCE_DIST') == '1':   # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group('nccl', device_id=dev)
        assert dist.get_world_size() == max(args.gpus, 1) or os.environ.get('SNSDE_BENCH_FORCE_DIST') == '1', \
            f'--gpus {args.gpus} but the process group has {dist.get_world_size()} ranks'

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

    def maxr(x):
        return S.sharding.max_over_ranks(x, device=dev)

    for _ in range(args.warmup):
        call.launch(stream)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        call.launch(stream)          # the engine's call, as sdeint makes it: prepare launch (weight pack + time table) + fused solve
    barrier()
    elapsed = maxr(time.perf_counter() - t0)

    # per-solve HIP-event timings on the launch stream: the whole call, and the dominant kernel alone (prepared
    # workspace reused)
    t_call_prep = event_times_ms(lambda: call.launch(stream), stream, max(50, args.steps), 10)     # as timed in `value`
    t_call = event_times_ms(lambda: call.launch(stream, auto_reuse=True), stream, 50, 10)          # evaluation epochs: prepared workspace reused
    t_kern = event_times_ms(lambda: call.launch(stream, reuse_prepared=True), stream, 50, 10)
    kern_ms = float(np.median(t_kern))
    ys = call.ys
    assert bool(torch.isfinite(ys).all()), 'non-finite solver output'

    # `extra`: the widened rows first, the BASELINE.json configurations LAST, and a compact `summary` of the latter as the very last
    # key of the line - a record that keeps only the tail of stdout still carries every BASELINE leg
    extra = {}
    if not args.no_extra:
        if world == 1:
            extra["tutorial_field"] = tutorial_field(dev, stream)
            extra["K1_tutorial_lsde"] = tutorial_field(dev, stream, kind='lsde', rows=256, hh=32, n=50)
            extra["latent_sde_srk"] = latent_sde(dev, stream)
            # torch_ists' neuralsde_1_18 at the K2 width: diffusion nets on the MFMA net kernels (snsde_m4n_kernel.h)
            extra["NSDE_1_18_srk_H128"] = train_leg(dev, stream, 1, 18, 1024, 128, 21, 50, 'srk', 'neuralsde_1_18, srk')
            # BASELINE config 4's shape (neuralsde_3_18, the README's headline Neural SDE, 2048 x H = 64, C = 69, 71 steps) under the
            # benchmarks' Euler, torch_ists' default srk, and Milstein
            extra["NSDE_3_18_milstein_K4_shape"] = train_leg(dev, stream, 3, 18, 2048, 64, 69, 72, 'milstein', 'neuralsde_3_18, milstein')
            extra["NSDE_3_18_srk_K4_shape"] = train_leg(dev, stream, 3, 18, 2048, 64, 69, 72, 'srk', 'neuralsde_3_18, srk')
            extra["K4_3_18_euler"] = train_leg(dev, stream, 3, 18, 2048, 64, 69, 72, 'euler', 'K4: neuralsde_3_18, euler', outputs='knots')
            extra["K3_shard_512"] = strong_k3(dev, 0, 1, stream, barrier, maxr, shard_of=8)
            extra["K5_shard_128_train"] = strong_k5(dev, 0, 1, stream, barrier, maxr, None, shard_of=8)
        extra["K3_strong"] = strong_k3(dev, rank, world, stream, barrier, maxr)
        extra["K5_strong_train"] = strong_k5(dev, rank, world, stream, barrier, maxr, dist)
        if world == 1:
            extra["K2_train"] = k2_training(dev, stream)

    if rank == 0:
        rowsteps = B * NSTEP
        value = world * rowsteps * args.steps / elapsed
        ach_tf = rowsteps * FLOP_PER_ROWSTEP / (kern_ms * 1e-3) / 1e12
        exe_tf = rowsteps * EXECUTED_FLOP_PER_ROWSTEP / (kern_ms * 1e-3) / 1e12
        ach_gbs = rowsteps * BYTES_PER_ROWSTEP / (kern_ms * 1e-3) / 1e9
        launched = S.engine.forward_path(model, B, L, NSTEP, kernel=args.kernel)
        traffic = HBM_TRAFFIC_BYTES_PER_LAUNCH if launched == HBM_TRAFFIC_KERNEL[0] and not args.exact_order else None
        out = {
            "metric": "SDE solver steps/sec (batch x steps / s), forward solve",
            "value": value, "unit": "row-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "K2: Neural LNSDE (io=4,no=17) NL=2 B=1024/GPU H=128 C=21 L=101 natural-spline "
                                   "coeffs 30% NaN, 100 Euler steps dt=1, ts=[0,100], in-kernel Philox dW",
                       "rows_per_gpu": B, "solver_steps": NSTEP, "global_rows": world * B,
                       "parallelism": f"row-shard x{world}, no collective in the solver", "kernel": args.kernel,
                       "prepare": "every call (weight pack + time table launch inside the timed region, as r01-r04)"},
            "timing": {"solve_call": spread(t_call), "solve_call_with_prepare": spread(t_call_prep), "solve_kernel": spread(t_kern),
                       "note": "solve_call_with_prepare: SolveCall.launch as timed in `value` (prepare launch + solve kernel, what sdeint "
                               "does per call; r05 timed the reuse form); solve_call: launch(auto_reuse=True) - the prepared workspace "
                               "(packed weights, time table) reused while the parameter block's version counter stands (evaluation epochs)",
                       "method": "HIP events on the launch stream, one pair per solve, after 10 warm-ups (SURVEY 8d)"},
            "roofline": {"bound": "mfma", "achieved": ach_tf, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach_tf / PEAK_FP32_TFLOPS, "traffic": traffic,
                         "traffic_source": HBM_TRAFFIC_SOURCE if traffic else f"none: the launched path {launched!r} is not the profiled one",
                         "launched_path": launched, "profiled_kernel": HBM_TRAFFIC_KERNEL[1],
                         "kernel_ms": kern_ms, "flop_per_rowstep": FLOP_PER_ROWSTEP, "bytes_per_rowstep": BYTES_PER_ROWSTEP,
                         "executed_flop_per_rowstep": EXECUTED_FLOP_PER_ROWSTEP, "executed_frac": exe_tf / PEAK_FP32_TFLOPS,
                         "mfma_issue_cycles_per_simd_step": MFMA_CYCLES_PER_STEP,
                         "kernel_ns_per_step": kern_ms * 1e6 / NSTEP,
                         "hbm_frac": ach_gbs / PEAK_HBM_GBS, "hbm_achieved_GBs": ach_gbs,
                         "note": "fp32 FMA/MFMA roof binds (intensity ~490 FLOP/B); frac counts the reference's algorithmic "
                                 "FLOPs, executed_frac the MFMA FLOPs the kernel issues (folded first layer); the kernel issues "
                                 "1664 MFMA cycles per SIMD and step (two waves x 104 x 8) - measured MFMA-busy share and clock: "
                                 "profiles/r06_pmc_counters.txt (SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE); hbm_* = algorithmic 346 B/row-step"},
        }
        if extra:
            out["extra"] = extra
        if not args.no_cpu_baseline and world == 1:     # reported at N = 1 only (the other ranks would idle behind it)
            out["cpu_baseline"] = cpu_baseline(pr, params)
            out["speedup_vs_cpu"] = value / world / out["cpu_baseline"]["value"]
        if extra:
            out["summary"] = summary_of(out, extra)
        out["full_record"] = write_full_record(out)
        line = compact_line(out)
    else:
        line = None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if line is not None:
        # the JSON line is the LAST thing on stdout: RCCL writes a version banner through C stdio (flushed at exit when
        # stdout is a pipe), so flush that first
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(line, flush=True)


if __name__ == '__main__':
    main()
