"""Worker of tests/test_gpu_dropin.py::test_hip_engine_under_ddp_nccl_...: launched by torch.distributed.run, one process per
GPU, backend nccl (= RCCL on ROCm).  SURVEY.md 8e / benchmark_classification/common_sde.py:157-162 (the training step the
DDP wrapper surrounds).

Every rank
  1. builds the same NeuralSDE (LNSDE field) and wraps it in DistributedDataParallel; the engine's parameter arena
     (engine.flatten_params re-points param.data into one buffer) is created by the first forward INSIDE DDP;
  2. solves its contiguous row shard of a global batch with the global Philox row offsets and checks the shard's states
     bit-for-bit against the single-process solve of the whole batch run on the same GPU;
  3. runs loss.backward() under DDP and checks the all-reduced (averaged) gradients against the gradient of the
     single-process full-batch loss;
  4. replays the training step from a captured graph (forward + fused adjoint + parameter pass + all-reduce-free local
     step) and checks the replayed loss against the eager one.
"""
import copy
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stable_neural_sdes_amd as S  # noqa: E402
from tests.helpers import make_problem  # noqa: E402


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist.init_process_group('nccl', device_id=dev)
    assert dist.get_world_size() == world

    B_local, H, C, L = 24, 64, 5, 9
    Bg = B_local * world
    pr = make_problem(77, 4, 17, 2, Bg, H, C, L)
    times = torch.from_numpy(pr['times']).to(dev)
    coeffs_g = torch.from_numpy(pr['coeffs']).to(dev)
    fi_g = torch.from_numpy(np.random.default_rng(1).integers(0, L, Bg)).to(dev)
    target_g = torch.from_numpy((np.random.default_rng(2).random(Bg) > 0.5).astype(np.float32)).to(dev)
    lo, hi = S.sharding.shard_rows(Bg, world, rank)

    torch.manual_seed(5)                       # identical initialisation on every rank
    model, field = S.make_sde_model('neurallnsde', C, 1, H, H, 2, initial=True)
    model = model.to(dev).eval()               # eval: BatchNorm uses running statistics (shard-size independent)
    single = copy.deepcopy(model)
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local])

    def loss_of(m, sl, row_offset, scale):
        pred = m(times, [coeffs_g[sl]], fi_g[sl], options={'seed': 1234, 'row_offset': row_offset, 'backend': 'hip'})
        return torch.nn.functional.binary_cross_entropy_with_logits(pred.squeeze(-1), target_g[sl], reduction='sum') * scale

    # --- 2. sharded states == single-process states (bit for bit) ---------------------------------------------------
    with torch.no_grad():
        single.func.set_X(coeffs_g, times)
        z0 = single.initial_network(single.func.X.evaluate(times[0]))
        full = S.sdeint(single.func, z0, times, dt=1.0, method='euler', options={'seed': 1234, 'backend': 'hip'})
        model.func.set_X(coeffs_g[lo:hi], times)
        mine = S.sdeint(model.func, z0[lo:hi], times, dt=1.0, method='euler',
                        options={'seed': 1234, 'row_offset': lo, 'backend': 'hip'})
        assert torch.equal(mine, full[:, lo:hi]), 'sharded trajectories differ from the single-process solve'
        # the default row offset under an initialised process group = rank * local batch
        mine2 = S.sdeint(model.func, z0[lo:hi], times, dt=1.0, method='euler', options={'seed': 1234, 'backend': 'hip'})
        assert torch.equal(mine2, mine), 'default row_offset under torch.distributed is not rank * local_batch'

    # --- 3. DDP gradient (mean over ranks of the shard losses x world / Bg) == full-batch gradient -------------------
    loss = loss_of(ddp, slice(lo, hi), lo, world / Bg)      # DDP averages over ranks: sum_r (world / Bg) L_r / world
    loss.backward()
    assert getattr(model.func, '_snsde_flat', None) is not None, 'parameter arena was not created'
    full_loss = loss_of(single, slice(0, Bg), 0, 1.0 / Bg)
    full_loss.backward()
    ref = dict(single.named_parameters())
    for name, p in model.named_parameters():
        g, gr = p.grad, ref[name].grad
        if gr is None:
            continue
        scale = float(gr.abs().max()) + 1e-12
        err = float((g - gr).abs().max()) / scale
        assert err < 2e-4, (name, err, scale)
    # the arena survived DDP's bucket views: an optimizer step writes where the kernels read
    opt = torch.optim.SGD(ddp.parameters(), lr=1e-2)
    opt.step()
    flat = model.func._snsde_flat
    layout, numel = S.engine.recognise(model.func)[1:]
    assert S.engine.flatten_params(model.func, layout, numel, dev) is flat

    # --- 4. graph-captured training step of the local model (capture-safe seed) ---------------------------------------
    # (the eager losses above keep their autograd graphs - and AccumulateGrad nodes bound to the default stream - alive;
    # torch requires them gone before a capture on another stream)
    del loss, full_loss, ref, g, gr, p
    import gc
    gc.collect()
    S.torchsde.prepare_graph_capture(dev)
    opt2 = torch.optim.Adam(single.parameters(), lr=1e-3, capturable=True)
    stream = torch.cuda.Stream(device=dev)
    stream.wait_stream(torch.cuda.current_stream(dev))
    static_loss = None

    c_l, f_l, t_l = coeffs_g[lo:hi].contiguous(), fi_g[lo:hi].contiguous(), target_g[lo:hi].contiguous()

    def step():
        pred = single(times, [c_l], f_l, options={'backend': 'hip', 'row_offset': lo})
        l = torch.nn.functional.binary_cross_entropy_with_logits(pred.squeeze(-1), t_l)
        opt2.zero_grad(set_to_none=True)
        l.backward()
        opt2.step()
        return l

    with torch.cuda.stream(stream):
        for _ in range(3):
            step()
    torch.cuda.current_stream(dev).wait_stream(stream)
    torch.cuda.synchronize(dev)
    graph = torch.cuda.CUDAGraph()
    # (thread_local: the process group's watchdog thread makes HIP calls of its own while this thread captures)
    with torch.cuda.graph(graph, capture_error_mode='thread_local'):
        static_loss = step()
    vals = []
    for _ in range(3):
        graph.replay()
        vals.append(float(static_loss))
    assert all(np.isfinite(vals)) and len(set(vals)) > 1, vals     # fresh increments and moving weights on every replay

    dist.barrier()
    if rank == 0:
        print(f'ddp worker ok world={world}')
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
