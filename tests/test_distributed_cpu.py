"""N>1 path on CPU: two gloo processes shard a batch by rows exactly like bench.py / a multi-GPU run does
(stable_neural_sdes_amd.sharding), each solves its shard with the ORACLE driven by the Philox stream
specification at its global row_offset, and the gathered result must equal the unsharded solve bit-for-bit.
No collective is needed on the data path; the only reduction is the max-over-ranks timing."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, seed, out_q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import stable_neural_sdes_amd as S
        from oracle import sde_oracle as O
        from tests.helpers import make_problem
        pr = make_problem(77, 4, 17, 2, B, 16, 3, 9)       # every rank builds the same global problem
        lo, hi = S.sharding.shard_rows(B, world, rank)
        ts, dt = np.array([0., 3., 8.], np.float32), 1.0
        t0, t1, *_ = O.step_grid(ts, dt)
        dW = O.philox_dW(seed, lo, hi - lo, 16, t0, t1)     # global-row counters: row_offset = lo
        ys, _ = O.solve_diffusion_model(pr['params'], 4, 17, pr['coeffs'][lo:hi], pr['times'], pr['y0'][lo:hi], ts, dt,
                                        dW, dtype=np.float32)
        full = S.sharding.gather_rows(torch.from_numpy(ys), B, dim=1)
        slowest = S.sharding.max_over_ranks(1.0 + rank)
        if rank == 0:
            out_q.put((full.numpy(), slowest))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_row_sharding_two_process_gloo_matches_single_process():
    sys.path.insert(0, ROOT)
    from oracle import sde_oracle as O
    from tests.helpers import make_problem
    B, seed, world = 11, 4242, 2       # odd batch: unequal shards
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    full, slowest = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    pr = make_problem(77, 4, 17, 2, B, 16, 3, 9)
    ts, dt = np.array([0., 3., 8.], np.float32), 1.0
    t0, t1, *_ = O.step_grid(ts, dt)
    dW = O.philox_dW(seed, 0, B, 16, t0, t1)
    ref, _ = O.solve_diffusion_model(pr['params'], 4, 17, pr['coeffs'], pr['times'], pr['y0'], ts, dt, dW,
                                     dtype=np.float32)
    np.testing.assert_array_equal(full, ref)
    assert slowest == 2.0              # MAX over ranks of (1 + rank)


def test_shard_rows_partition():
    import stable_neural_sdes_amd as S
    for n, w in ((1024, 8), (11, 2), (7, 4), (3, 8)):
        spans = [S.sharding.shard_rows(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        S.sharding.shard_rows(8, 2, 2)
