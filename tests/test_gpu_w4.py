"""GPU tests of the wave-owns-rows forward kernels (csrc/snsde_w4_kernel.h; run with -m gpu): H = 64 models with a diffusion net under
Euler - BASELINE config 4's model family.  Forward against the fp64 numpy oracle on replayed increments, against the 4-row-tile kernels,
Philox runs against the generic kernels, per-row outputs / interpolated outputs / ragged tiles, and the training-mode saves through the
unchanged adjoint + weight-gradient pass against fp64 autograd."""
import numpy as np
import pytest
import torch

import stable_neural_sdes_amd as S
from tests.helpers import assert_parity, draw_dW, make_problem
from tests.test_gpu_parity import _check_backward, hip_solve, oracle_solve

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'

W4_CASES = [
    # io, no, NL, B, C, L, ts, dt
    (3, 18, 2, 37, 5, 9, [0, 3.5, 8], 1.0),          # BASELINE config 4's model; ragged last tile, an interpolated output
    (1, 14, 1, 9, 3, 8, [0, 7], 0.5),                # one-layer drift, one-layer net, no time features in the drift
    (5, 19, 2, 21, 3, 9, [0, 8], 1.0),               # geometric drift, raw = net * y
    (3, 15, 1, 13, 4, 8, [0, 2.5, 7], 1.0),
    (1, 18, 2, 64, 3, 12, None, None),               # every knot an output, linspace grid
    (5, 14, 2, 8, 3, 8, [0, 7], 1.0),
]


@pytest.mark.parametrize('ci', range(len(W4_CASES)))
def test_w4_forward_vs_oracle_and_the_tile_kernels(ci):
    io, no, NL, B, C, L, ts, dt = W4_CASES[ci]
    H = 64
    times = np.linspace(0, 1, L).astype(np.float32) if ts is None else None
    pr = make_problem(8800 + ci, io, no, NL, B, H, C, L, times=times)
    ts = pr['times'] if ts is None else np.asarray(ts, np.float32)
    dt = dt or float(np.diff(pr['times']).min())
    model = S.engine.model_struct(C, H, H, NL, io, no)
    grid = S.engine.step_grid(ts, dt, pr['times'], torch.device(DEV))
    assert S.engine.forward_path(model, B, L, grid.N, kernel='w4') == 'w4'
    assert S.engine.forward_path(model, B, L, grid.N) == 'w4'                  # ... and `auto` takes it
    dW = draw_dW(8800 + ci, ts, dt, B, H)
    ref64, _ = oracle_solve(pr, ts, dt, dW, 'euler', np.float64)
    cpu32, _ = oracle_solve(pr, ts, dt, dW, 'euler', np.float32)
    ys, _ = hip_solve(pr, ts, dt, dW=dW, kernel='w4', save_traj=True)
    assert ys.shape == ref64.shape
    assert_parity(ys, ref64, cpu32, what=f'w4 case {ci}')
    y4, _ = hip_solve(pr, ts, dt, dW=dW, kernel='mfma4')
    assert np.abs(ys - y4).max() <= 2e-5 * (np.abs(y4).max() + 1.0)
    # in-kernel Philox: the same stream as every other kernel family (global row, step block, column)
    yp, _ = hip_solve(pr, ts, dt, seed=77, row_offset=5, kernel='w4')
    yg, _ = hip_solve(pr, ts, dt, seed=77, row_offset=5, kernel='generic')
    assert np.isfinite(yp).all() and np.abs(yp - yg).max() <= 2e-4 * (np.abs(yg).max() + 1.0)
    # row shards keep the global stream bit for bit
    half = B // 2
    ya, _ = hip_solve(pr, ts, dt, seed=77, row_offset=0, kernel='w4')
    yb0, _ = hip_solve(pr, ts, dt, seed=77, row_offset=0, rows=slice(0, half), kernel='w4')
    yb1, _ = hip_solve(pr, ts, dt, seed=77, row_offset=half, rows=slice(half, B), kernel='w4')
    np.testing.assert_array_equal(np.concatenate([yb0, yb1], axis=1), ya)


def test_w4_per_row_outputs_and_trajectory():
    io, no, NL, B, C, L, H = 3, 18, 2, 19, 5, 9, 64
    pr = make_problem(8900, io, no, NL, B, H, C, L)
    ts = pr['times'][[0, 2, 5, 8]]
    dW = draw_dW(8900, ts, 1.0, B, H)
    ref64, traj64 = oracle_solve(pr, ts, 1.0, dW, 'euler', np.float64)
    model = S.engine.model_struct(C, H, H, NL, io, no)
    from tests.test_gpu_parity import flat_params
    flat = flat_params(pr['params'], io, no, NL, C, H)
    grid = S.engine.step_grid(ts, 1.0, pr['times'], torch.device(DEV))
    row_out = torch.from_numpy(np.random.default_rng(1).integers(0, len(ts), size=B).astype(np.int32)).to(DEV)
    call = S.engine.SolveCall(model, flat, torch.from_numpy(pr['coeffs']).to(DEV), grid, torch.from_numpy(pr['y0']).to(DEV),
                              dW=torch.from_numpy(dW).to(DEV), kernel='w4', row_out=row_out, save_traj=True, save_dW=True)
    ys = call.launch()
    torch.cuda.synchronize()
    want = ref64[row_out.cpu().numpy(), np.arange(B)]
    assert_parity(ys.cpu().numpy(), want, what='w4 row_out')
    assert_parity(call.traj.cpu().numpy(), traj64, what='w4 trajectory')
    np.testing.assert_array_equal(call.dW_out.cpu().numpy(), dW)


BWD = [
    # io, no, NL, B, C, L, ts, dt
    (3, 18, 2, 21, 5, 9, [0, 3.5, 8], 1.0),
    (1, 14, 1, 9, 3, 8, [0, 7], 0.5),
    (5, 19, 2, 13, 3, 9, [0, 8], 1.0),
    (3, 15, 2, 11, 4, 8, [0, 2.5, 7], 1.0),
    (1, 18, 1, 10, 3, 8, [0, 7], 1.0),
]


@pytest.mark.parametrize('ci', range(len(BWD)))
def test_w4_training_saves_drive_the_fused_adjoint(ci):
    """Forward on the wave-owns-rows kernel in training mode (act_save slots, relu signs in the saved z, increments); backward: under
    'auto' / 'w4' the wave-pair adjoint (snsde_w4_euler_reverse_kernel), under 'mfma4' the tile adjoint on the SAME saves - both with the
    unchanged weight-gradient pass, against fp64 autograd through the tensor loop."""
    io, no, NL, B, C, L, ts, dt = BWD[ci]
    grid = S.engine.step_grid(np.asarray(ts, np.float32), dt, np.arange(L, dtype=np.float32), torch.device(DEV))
    model = S.engine.model_struct(C, 64, 64, NL, io, no)
    assert S.engine.forward_path(model, B, L, grid.N) == 'w4' and S.engine.backward_mode(model, B, L, grid, 'euler') == 1
    _check_backward(8950 + ci, io, no, NL, B, 64, C, L, ts, dt, 'euler', 'w4', strict=True)
    _check_backward(8950 + ci, io, no, NL, B, 64, C, L, ts, dt, 'euler', 'auto', strict=True)
    _check_backward(8950 + ci, io, no, NL, B, 64, C, L, ts, dt, 'euler', 'mfma4', strict=True)


# ---- SRK (SRID2) on the wave pair -------------------------------------------------------------------------------------------------
from oracle import sde_oracle as O      # noqa: E402

SRK_W4 = [
    # io, no, NL, B, C, L, ts, dt
    (3, 18, 2, 37, 5, 9, [0, 3.5, 8], 1.0),          # the README's neuralsde_3_18 under torch_ists' default method
    (1, 18, 2, 9, 3, 8, [0, 7], 0.5),
    (5, 19, 2, 21, 3, 9, [0, 8], 1.0),
    (3, 15, 1, 13, 4, 8, [0, 2.5, 7], 1.0),
    (1, 14, 2, 16, 3, 12, None, None),
]


def _levy(seed, dW, ts, dt):
    g0, g1 = O.step_grid(np.asarray(ts, np.float32), dt)[:2]
    hh = (g1 - g0).astype(np.float32).reshape(-1, 1, 1)
    xi = np.random.default_rng(seed).standard_normal(dW.shape).astype(np.float32)
    return (hh * (0.5 * dW + np.sqrt(hh / 12) * xi)).astype(np.float32)


@pytest.mark.parametrize('ci', range(len(SRK_W4)))
def test_w4_srk_forward_vs_oracle_and_the_tile_kernels(ci):
    io, no, NL, B, C, L, ts, dt = SRK_W4[ci]
    H = 64
    times = np.linspace(0, 1, L).astype(np.float32) if ts is None else None
    pr = make_problem(9300 + ci, io, no, NL, B, H, C, L, times=times)
    ts = pr['times'] if ts is None else np.asarray(ts, np.float32)
    dt = dt or float(np.diff(pr['times']).min())
    model = S.engine.model_struct(C, H, H, NL, io, no)
    grid = S.engine.step_grid(ts, dt, pr['times'], torch.device(DEV))
    assert S.engine.forward_path(model, B, L, grid.N, method='srk', kernel='w4') == 'w4'
    assert S.engine.forward_path(model, B, L, grid.N, method='srk') == 'w4'
    dW = draw_dW(9300 + ci, ts, dt, B, H)
    dU = _levy(9300 + ci, dW, ts, dt)
    ref64, _ = O.solve_diffusion_model(pr['params'], io, no, pr['coeffs'], pr['times'], pr['y0'], ts, dt, dW, method='srk', dtype=np.float64, dU=dU)
    cpu32, _ = O.solve_diffusion_model(pr['params'], io, no, pr['coeffs'], pr['times'], pr['y0'], ts, dt, dW, method='srk', dtype=np.float32, dU=dU)
    ys, _ = hip_solve(pr, ts, dt, dW=dW, dU=dU, method='srk', kernel='w4')
    assert_parity(ys, ref64, cpu32, what=f'w4 srk case {ci}')
    y4, _ = hip_solve(pr, ts, dt, dW=dW, dU=dU, method='srk', kernel='mfma4')
    assert np.abs(ys - y4).max() <= 5e-5 * (np.abs(y4).max() + 1.0)
    yp, _ = hip_solve(pr, ts, dt, seed=78, row_offset=3, method='srk', kernel='w4')
    yg, _ = hip_solve(pr, ts, dt, seed=78, row_offset=3, method='srk', kernel='generic')
    assert np.isfinite(yp).all() and np.abs(yp - yg).max() <= 5e-4 * (np.abs(yg).max() + 1.0)
    half = B // 2
    ya, _ = hip_solve(pr, ts, dt, seed=78, method='srk', kernel='w4')
    yb0, _ = hip_solve(pr, ts, dt, seed=78, rows=slice(0, half), method='srk', kernel='w4')
    yb1, _ = hip_solve(pr, ts, dt, seed=78, row_offset=half, rows=slice(half, B), method='srk', kernel='w4')
    np.testing.assert_array_equal(np.concatenate([yb0, yb1], axis=1), ya)


SRK_BWD = [
    (3, 18, 2, 21, 5, 9, [0, 3.5, 8], 1.0),
    (1, 14, 1, 9, 3, 8, [0, 7], 0.5),
    (5, 19, 2, 13, 3, 9, [0, 8], 1.0),
    (3, 15, 2, 11, 4, 8, [0, 2.5, 7], 1.0),
    (1, 18, 1, 10, 3, 8, [0, 7], 1.0),
]


@pytest.mark.parametrize('ci', range(len(SRK_BWD)))
def test_w4_srk_training_saves_drive_the_fused_srk_adjoint(ci):
    io, no, NL, B, C, L, ts, dt = SRK_BWD[ci]
    grid = S.engine.step_grid(np.asarray(ts, np.float32), dt, np.arange(L, dtype=np.float32), torch.device(DEV))
    model = S.engine.model_struct(C, 64, 64, NL, io, no)
    assert S.engine.forward_path(model, B, L, grid.N, method='srk') == 'w4' and S.engine.backward_mode(model, B, L, grid, 'srk') == 1
    """Forward SRID2 on the wave pair in training mode; backward: under 'w4' / 'auto' the wave-group SRK adjoint with the weight
    gradients inside (snsde_w4_srk_reverse_kernel: stage states from stage_save, no delta planes), under 'mfma4' the tile adjoint +
    weight-gradient pass on the same saves - all against fp64 autograd through the tensor loop."""
    _check_backward(9350 + ci, io, no, NL, B, 64, C, L, ts, dt, 'srk', 'w4', strict=True)
    _check_backward(9350 + ci, io, no, NL, B, 64, C, L, ts, dt, 'srk', 'auto', strict=True)
    _check_backward(9350 + ci, io, no, NL, B, 64, C, L, ts, dt, 'srk', 'mfma4', strict=True)


@pytest.mark.parametrize('case', [(3, 18, 2, 37, False), (5, 19, 2, 22, True), (1, 14, 1, 9, False), (3, 15, 2, 130, True), (5, 18, 1, 64, False)])
def test_w4_srk_adjoint_equals_the_tile_adjoint(case):
    """SRK training through sdeint with in-kernel Philox increments and (optionally) per-row outputs: the wave-group path ('auto':
    fused SRK adjoint, delta_slots == 0) against the 4-row-tile path ('mfma4': snsde_m4n_srk_reverse_kernel + the weight-gradient
    GEMMs) on the same key - states, dL/dy0 and every parameter gradient to round-off; ragged tails, gated drifts (io 5), raw = q y."""
    io, no, NL, B, row_out = case
    C, L, H = 5, 9, 64
    # (round 6: under the SRI2W1 rows the 22-row case drew a state whose hidden pre-activation sits within round-off of the relu
    #  kink in ONE row - tools/diag_w4_srk.py: row 12 alone differed, in all 64 components, by 2e-3 of the gradient scale, states
    #  equal to 1e-6; both gradients are valid one-sided derivatives.  That case draws from another seed.)
    pr = make_problem((9700 if B != 22 else 9800) + B, io, no, NL, B, H, C, L)
    ts = torch.from_numpy(pr['times'][[0, 2, 5, 8]]).to(DEV)
    ro = torch.from_numpy(np.random.default_rng(2).integers(0, 4, size=B).astype(np.int32)).to(DEV) if row_out else None
    model = S.engine.model_struct(C, H, H, NL, io, no)
    grid = S.engine.step_grid(pr['times'][[0, 2, 5, 8]], 0.5, pr['times'], torch.device(DEV))
    from tests.test_gpu_parity import flat_params
    flat = flat_params(pr['params'], io, no, NL, C, H)
    for kernel, fused in (('auto', True), ('w4', True), ('mfma4', False)):
        call = S.engine.SolveCall(model, flat, torch.from_numpy(pr['coeffs']).to(DEV), grid, torch.from_numpy(pr['y0']).to(DEV), seed=5,
                                  method='srk', kernel=kernel, save_traj=True, save_dW=True, save_act=True)
        assert (call.delta_slots == 0) == fused, (kernel, call.delta_slots)
    out = {}
    for kernel in ('auto', 'mfma4'):
        m = S.Diffusion_model(C, H, H, NL, input_option=io, noise_option=no)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()})
        m = m.to(DEV)
        m.set_X(torch.from_numpy(pr['coeffs']).to(DEV), torch.from_numpy(pr['times']).to(DEV))
        y0 = torch.from_numpy(pr['y0']).to(DEV).requires_grad_(True)
        opts = {'seed': 99, 'kernel': kernel, 'strict': True}
        if ro is not None:
            opts['row_out'] = ro
        ys = S.sdeint(m, y0, ts, dt=0.5, method='srk', options=opts)
        w = torch.from_numpy(np.random.default_rng(3).standard_normal(tuple(ys.shape)).astype(np.float32)).to(DEV)
        (ys * w).sum().backward()
        out[kernel] = (ys.detach(), y0.grad.clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    ya, ga, pa = out['auto']
    yb, gb, pb = out['mfma4']
    assert float((ya - yb).abs().max()) <= 2e-5 * (float(yb.abs().max()) + 1.0)
    assert float((ga - gb).abs().max()) <= 5e-5 * float(gb.abs().max())
    assert pa.keys() == pb.keys()
    for k in pb:
        assert float((pa[k] - pb[k]).abs().max()) <= 1e-4 * (float(pb[k].abs().max()) + 1e-12), k


@pytest.mark.parametrize('method', ['euler', 'srk'])
@pytest.mark.parametrize('case', [(4, [0, 1]), (5, [0, 1]), (4, [0, 2]), (7, [0, 1, 2]), (8, [0, 3])])
def test_w4_adjoints_on_the_shortest_solves_and_smallest_batches(method, case):
    """One-, two- and three-step solves (the adjoints' one-step-behind gradient waves and a-step-ahead fetches at their boundaries)
    on one tile, on one tile + a ragged second one (rows solved twice must not enter the sums twice), on two tiles."""
    B, ts = case
    io, no, NL, C, L = 3, 18, 2, 4, 5
    grid = S.engine.step_grid(np.asarray(ts, np.float32), 1.0, np.arange(L, dtype=np.float32), torch.device(DEV))
    model = S.engine.model_struct(C, 64, 64, NL, io, no)
    assert S.engine.forward_path(model, B, L, grid.N, method=method) == 'w4' and grid.N == int(ts[-1])
    _check_backward(9800 + 10 * B + len(ts), io, no, NL, B, 64, C, L, ts, 1.0, method, 'auto', strict=True)


@pytest.mark.parametrize('row_out', [False, True])
def test_w4_adjoint_with_in_kernel_philox_and_row_outputs_equals_the_tile_adjoint(row_out):
    """Training through sdeint with in-kernel Philox increments (the adjoint regenerates them or reads the forward's dW_out) and,
    optionally, per-row output selection: the wave-pair path ('auto') against the 4-row-tile path ('mfma4') on the same key - same
    Philox stream, so states and every gradient agree to round-off."""
    io, no, NL, B, C, L, H = 3, 18, 2, 37, 5, 9, 64
    pr = make_problem(9500, io, no, NL, B, H, C, L)
    ts = torch.from_numpy(pr['times'][[0, 2, 5, 8]]).to(DEV)
    ro = torch.from_numpy(np.random.default_rng(2).integers(0, 4, size=B).astype(np.int32)).to(DEV) if row_out else None
    out = {}
    for kernel in ('auto', 'mfma4'):
        m = S.Diffusion_model(C, H, H, NL, input_option=io, noise_option=no)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()})
        m = m.to(DEV)
        m.set_X(torch.from_numpy(pr['coeffs']).to(DEV), torch.from_numpy(pr['times']).to(DEV))
        y0 = torch.from_numpy(pr['y0']).to(DEV).requires_grad_(True)
        opts = {'seed': 99, 'kernel': kernel, 'strict': True}
        if ro is not None:
            opts['row_out'] = ro
        ys = S.sdeint(m, y0, ts, dt=0.5, method='euler', options=opts)
        w = torch.from_numpy(np.random.default_rng(3).standard_normal(tuple(ys.shape)).astype(np.float32)).to(DEV)
        (ys * w).sum().backward()
        out[kernel] = (ys.detach(), y0.grad.clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    ya, ga, pa = out['auto']
    yb, gb, pb = out['mfma4']
    assert float((ya - yb).abs().max()) <= 2e-5 * (float(yb.abs().max()) + 1.0)
    assert float((ga - gb).abs().max()) <= 2e-5 * float(gb.abs().max())
    assert pa.keys() == pb.keys()
    for k in pb:
        assert float((pa[k] - pb[k]).abs().max()) <= 5e-5 * (float(pb[k].abs().max()) + 1e-12), k


def test_fused_weight_gradients_need_no_delta_planes_and_fall_back_with_them():
    """snsde_save_layout reports delta_slots = 0 where the wave-pair adjoint sums the weight gradients itself (auto / w4, host or
    device-resident Philox key); the 4-row-tile selector keeps its delta planes.  Through the engine's two-call form
    (solve_backward + param_gradients) and the one-call form: same flat gradient as the tile path."""
    io, no, NL, B, C, L, H = 3, 18, 2, 37, 5, 9, 64
    pr = make_problem(9600, io, no, NL, B, H, C, L)
    from tests.test_gpu_parity import flat_params
    model = S.engine.model_struct(C, H, H, NL, io, no)
    flat = flat_params(pr['params'], io, no, NL, C, H)
    grid = S.engine.step_grid(pr['times'][[0, 3, 8]], 0.5, pr['times'], torch.device(DEV))
    args = (torch.from_numpy(pr['coeffs']).to(DEV), grid, torch.from_numpy(pr['y0']).to(DEV))
    g = torch.from_numpy(np.random.default_rng(1).standard_normal((3, B, H)).astype(np.float32)).to(DEV)
    grads = {}
    for kernel in ('auto', 'w4', 'mfma4'):
        call = S.engine.SolveCall(model, flat, *args, seed=11, kernel=kernel, save_traj=True, save_dW=True, save_act=True)
        call.launch()
        assert (call.delta_slots == 0) == (kernel != 'mfma4'), (kernel, call.delta_slots)
        adj, delta = S.engine.solve_backward(call, g, save_delta=True, adj0_only=True)
        assert (delta is None) == (kernel != 'mfma4')
        two = S.engine.param_gradients(call, adj, delta)
        adj1, one = S.engine.backward_with_gradients(call, g)
        assert torch.equal(adj, adj1) and torch.equal(one, two)
        grads[kernel] = one
    # device-resident key (graph replays): the same adjoint, reading the increments the forward left in dW_out
    seed_dev = torch.tensor([11], dtype=torch.int64, device=DEV)
    call = S.engine.SolveCall(model, flat, *args, seed=seed_dev, save_traj=True, save_dW=True, save_act=True)
    call.launch()
    assert call.delta_slots == 0
    grads['device key'] = S.engine.backward_with_gradients(call, g)[1]
    ref = grads['mfma4']
    assert float((grads['device key'] - ref).abs().max()) <= 5e-5 * float(ref.abs().max())
    for k in ('auto', 'w4'):
        assert float((grads[k] - ref).abs().max()) <= 5e-5 * float(ref.abs().max()), k
    assert float(ref.abs().max()) > 0
