"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI
(libsnsde.so via ctypes), against the oracle on identical seeded inputs and against the golden
vectors generated from the reference.  Tolerances are the ones stated in SURVEY.md 8c:
  single f/g evaluation:  allclose(rtol=1e-5, atol=2e-6) vs the reference's fp32 output
  trajectories (vs fp64 arbiter, identical dW): mean |err| <= 1e-5, |err| <= 1e-4 + 1e-4 |z| for
  >= 99.99 % of elements, max |err| <= 5e-3, and <= 4x the CPU-fp32-vs-fp64 error."""
import os

import numpy as np
import pytest
import torch

import stable_neural_sdes_amd as S
from oracle import sde_oracle as O
from tests.helpers import (assert_parity, draw_dW, group, load, make_problem, param_spec, params_of, unflatten)

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def flat_params(p, io, no, NL, C, H):
    return torch.from_numpy(np.concatenate([np.asarray(p[n], np.float32).reshape(-1)
                                            for n, _ in param_spec(io, no, NL, C, H)])).to(DEV)


def hip_solve(pr, ts, dt, dW=None, method='euler', seed=0, row_offset=0, kernel='auto', rows=None, save_traj=False,
              save_dW=False, dU=None):
    exact = kernel.endswith('x')          # 'mfma4x' = MFMA path keeping the reference's unfused emb order
    kernel = kernel[:-1] if exact else kernel
    io, no, NL, C, H = pr['io'], pr['no'], pr['NL'], pr['C'], pr['H']
    sl = slice(None) if rows is None else rows
    model = S.engine.model_struct(C, H, H, NL, io, no)
    flat = flat_params(pr['params'], io, no, NL, C, H)
    coeffs = torch.from_numpy(np.ascontiguousarray(pr['coeffs'][sl])).to(DEV)
    y0 = torch.from_numpy(np.ascontiguousarray(pr['y0'][sl])).to(DEV)
    grid = S.engine.step_grid(np.asarray(ts, np.float32), dt, pr['times'], torch.device(DEV))
    dWd = None if dW is None else torch.from_numpy(np.ascontiguousarray(dW[:, sl])).to(DEV)
    dUd = None if dU is None else torch.from_numpy(np.ascontiguousarray(dU[:, sl])).to(DEV)
    call = S.engine.SolveCall(model, flat, coeffs, grid, y0, dW=dWd, method=method, seed=seed, row_offset=row_offset,
                              kernel=kernel, save_traj=save_traj, save_dW=save_dW, exact_order=exact, dU=dUd)
    ys = call.launch()
    torch.cuda.synchronize()
    return ys.cpu().numpy(), call


def oracle_solve(pr, ts, dt, dW, method, dtype):
    ys, traj = O.solve_diffusion_model(pr['params'], pr['io'], pr['no'], pr['coeffs'], pr['times'], pr['y0'],
                                       np.asarray(ts, np.float32), dt, dW, method=method, dtype=dtype)
    return ys, traj


# ------------------------------------------------------------------------------------------------
SPL, FG, TRAJ = load('spline.npz'), load('fg.npz'), load('traj.npz')
G1_CASES = sorted({k.split('/')[1] for k in SPL.files if k.startswith('G1/')})


@pytest.mark.parametrize('case', G1_CASES)
def test_spline_evaluate_hip_bit_exact_vs_reference(case):
    c = group(SPL, f'G1/{case}/f32')
    g = group(SPL, f'G2/{case}/f32')
    coeffs = torch.from_numpy(np.concatenate([c['a'], c['b'], c['two_c'], c['three_d']], -1)).to(DEV)
    sp = S.torchcde.CubicSpline(coeffs, torch.from_numpy(c['times']).to(DEV))
    for i, t in enumerate(g['t']):
        np.testing.assert_array_equal(sp.evaluate(torch.tensor(t)).cpu().numpy(), g['evaluate'][i])
        np.testing.assert_array_equal(sp.derivative(torch.tensor(t)).cpu().numpy(), g['derivative'][i])


MODELS = FG['G3/models']


@pytest.mark.parametrize('mi', range(len(MODELS)))
def test_f_g_hip_vs_reference_golden(mi):
    io, no, NL = (int(v) for v in MODELS[mi])
    coeffs, times, y, tv = FG['G3/coeffs'], FG['G3/times'], FG['G3/y'], FG['G3/t']
    B, H = y.shape
    C = coeffs.shape[-1] // 4
    off = FG['G3/params_off']
    p = unflatten(FG['G3/params_flat'][off[mi]:off[mi + 1]], param_spec(io, no, NL, C, H))
    model = S.engine.model_struct(C, H, H, NL, io, no)
    flat = flat_params(p, io, no, NL, C, H)
    cd, yd = torch.from_numpy(coeffs).to(DEV), torch.from_numpy(y).to(DEV)
    exp32, exp64 = FG['G3/out32'][mi], FG['G3/out64'][mi]
    for ti, t in enumerate(tv):
        f, g = S.engine.eval_fg(model, flat, cd, times, np.float32(t), yd)
        f, g = f.cpu().numpy(), g.cpu().numpy()
        np.testing.assert_allclose(f, exp32[0, ti], rtol=1e-5, atol=2e-6, err_msg=f'f io={io} no={no} t={t}')
        np.testing.assert_allclose(g, exp32[1, ti], rtol=1e-5, atol=2e-6, err_msg=f'g io={io} no={no} t={t}')
        # and no further from the fp64 reference than fp32 round-off allows
        if float(t) == float(np.float32(t)):
            np.testing.assert_allclose(f, exp64[0, ti], rtol=2e-5, atol=4e-6)
            np.testing.assert_allclose(g, exp64[1, ti], rtol=2e-5, atol=4e-6)


G5_CASES = sorted({k.split('/')[1] for k in TRAJ.files if k.startswith('G5/')})


@pytest.mark.parametrize('case', G5_CASES)
def test_trajectory_hip_vs_reference_golden(case):
    """Golden trajectories = the reference's own f/g modules driven by the torchsde-style loop."""
    g = group(TRAJ, f'G5/{case}')
    io, no, NL = (int(v) for v in g['io_no_nl'])
    C, H = g['coeffs'].shape[-1] // 4, g['y0'].shape[1]
    pr = dict(params=params_of(TRAJ, f'G5/{case}'), coeffs=g['coeffs'], times=g['times'], y0=g['y0'], io=io, no=no,
              NL=NL, C=C, H=H)
    ys, _ = hip_solve(pr, g['ts'], float(g['dt']), dW=g['dW'], method=str(g['method']))
    assert ys.shape == g['ys64'].shape
    assert_parity(ys, g['ys64'], g['ys32'], what=case)


SEEDED = [
    # io, no, NL, B, H, C, L, ts, dt, method
    (4, 17, 2, 37, 32, 5, 13, [0, 12], 1.0, 'euler'),
    (6, 17, 2, 16, 64, 7, 21, [0, 3, 4, 11, 20], 1.0, 'euler'),
    (2, 16, 1, 9, 32, 2, 20, None, 0.02, 'euler'),           # ts = times = linspace(0,1,20): interpolated outputs
    (3, 18, 2, 24, 16, 9, 10, [0, 4.5, 9], 1.0, 'euler'),
    (1, 18, 3, 8, 24, 3, 8, [0, 7], 0.5, 'euler'),
    (1, 0, 2, 5, 10, 3, 8, [0, 2.5, 7], 1.0, 'euler'),        # H not a multiple of 4
    (0, 13, 2, 8, 12, 3, 8, [0, 7], 1.0, 'euler'),
    (5, 19, 2, 8, 16, 4, 8, [0, 7], 1.0, 'euler'),
    (2, 14, 2, 8, 16, 4, 8, [0, 7], 1.0, 'euler'),
    (4, 15, 4, 8, 16, 4, 8, [0, 7], 1.0, 'euler'),
    (4, 17, 2, 33, 32, 5, 13, [0, 5, 12], 1.0, 'milstein'),
    (6, 17, 2, 16, 16, 3, 9, None, None, 'milstein'),
    (2, 16, 2, 8, 16, 3, 9, [0, 8], 1.0, 'milstein'),
    (5, 8, 2, 8, 8, 3, 9, [0, 8], 0.5, 'milstein'),
    (3, 3, 2, 8, 8, 3, 9, [0, 8], 0.5, 'milstein'),
    (3, 6, 2, 8, 8, 3, 9, [0, 8], 0.5, 'milstein'),
    (3, 11, 2, 8, 8, 3, 9, [0, 8], 0.25, 'milstein'),
    (4, 17, 2, 64, 128, 21, 26, [0, 25], 1.0, 'euler'),      # headline model shape, short horizon
    (4, 17, 2, 12, 256, 14, 11, [0, 10], 1.0, 'euler'),
]


@pytest.mark.parametrize('ci', range(len(SEEDED)))
def test_trajectory_hip_vs_oracle_seeded(ci):
    io, no, NL, B, H, C, L, ts, dt, method = SEEDED[ci]
    times = np.linspace(0, 1, L).astype(np.float32) if ts is None else None
    pr = make_problem(100 + ci, io, no, NL, B, H, C, L, times=times)
    if ts is None:
        ts = pr['times']
        dt = dt or max(float(np.diff(pr['times']).min()), 1e-3)
    dW = draw_dW(100 + ci, ts, dt, B, H)
    ys, call = hip_solve(pr, ts, dt, dW=dW, method=method, save_traj=True)
    ref64, traj64 = oracle_solve(pr, ts, dt, dW, method, np.float64)
    cpu32, _ = oracle_solve(pr, ts, dt, dW, method, np.float32)
    assert_parity(ys, ref64, cpu32, what=f'case {ci}')
    assert_parity(call.traj.cpu().numpy(), traj64, what=f'traj {ci}')
    np.testing.assert_array_equal(ys[0], pr['y0'])


def test_all_noise_options_one_step_chain():
    """Every (io, no) pair through the SOLVER (not just the f/g probe), 6 Euler steps."""
    bad = []
    for io in range(7):
        for no in range(20):
            pr = make_problem(7 * io + no, io, no, 2, 6, 8, 3, 7)
            pr['y0'][0, 0] = -1.5
            dW = draw_dW(io * 20 + no, [0, 6], 1.0, 6, 8)
            ys, _ = hip_solve(pr, [0, 6], 1.0, dW=dW)
            ref64, _ = oracle_solve(pr, [0, 6], 1.0, dW, 'euler', np.float64)
            cpu32, _ = oracle_solve(pr, [0, 6], 1.0, dW, 'euler', np.float32)
            try:
                assert_parity(ys, ref64, cpu32, what=f'io={io} no={no}')
            except AssertionError as e:
                bad.append(str(e))
    assert not bad, bad


def test_milstein_refused_where_dg_dy_is_not_finite():
    """sqrt(y) (noise_option 7): nan_to_num clips the value, its derivative has no finite counterpart (torchsde's VJP
    returns NaN there).  The diffusion nets (dense dg/dy) are served by the generic Milstein kernel."""
    pr = make_problem(1, 1, 7, 2, 4, 8, 3, 5)
    with pytest.raises(S._lib.SnsdeError) as e:
        hip_solve(pr, [0, 4], 1.0, dW=draw_dW(1, [0, 4], 1.0, 4, 8), method='milstein')
    assert e.value.code == -4
    pr = make_problem(1, 1, 18, 2, 4, 8, 3, 5)
    ys, _ = hip_solve(pr, [0, 4], 1.0, dW=draw_dW(1, [0, 4], 1.0, 4, 8), method='milstein')
    assert np.isfinite(ys).all()


@pytest.mark.parametrize('kernel', ['auto', 'generic'])
@pytest.mark.parametrize('cfg', [(3, 18, 2, 33, 64, 5, 12), (1, 14, 1, 17, 32, 3, 9), (4, 19, 2, 21, 128, 21, 10), (6, 15, 3, 9, 48, 40, 8),
                                 (0, 18, 2, 12, 100, 6, 9), (5, 19, 2, 21, 64, 5, 9), (1, 19, 1, 9, 16, 3, 8), (2, 15, 2, 13, 128, 7, 9),
                                 (6, 19, 4, 7, 32, 3, 8), (4, 14, 2, 11, 64, 69, 9)])
def test_milstein_through_a_diffusion_net_vs_oracle(cfg, kernel):
    """Milstein with a dense dg/dy (noise_option 14/15/18/19): J_g^T (g (dW^2 - h)) per step, torchsde's VJP form, against the
    numpy restatement in float64 (oracle.diffusion_g_vjp) on replayed increments."""
    io, no, NL, B, H, C, L = cfg
    pr = make_problem(60 + no, io, no, NL, B, H, C, L)
    ts = [0, 2.5, L - 1]
    dW = draw_dW(7, ts, 0.5, B, H)
    if kernel == 'auto' and H in (16, 32, 64) and io != 0:      # instantiated: the MFMA net kernel, not the generic family
        N = S.engine.step_grid(np.asarray(ts, np.float32), 0.5, pr['times'], torch.device(DEV)).N
        assert S.engine.forward_path(S.engine.model_struct(C, H, H, NL, io, no), B, L, N, method='milstein') == 'mfma4'
    ys, _ = hip_solve(pr, ts, 0.5, dW=dW, method='milstein', kernel=kernel)
    ref64, _ = oracle_solve(pr, ts, 0.5, dW, 'milstein', np.float64)
    cpu32, _ = oracle_solve(pr, ts, 0.5, dW, 'milstein', np.float32)
    assert_parity(ys, ref64, cpu32, what=f'milstein io={io} no={no}')
    eu, _ = oracle_solve(pr, ts, 0.5, dW, 'euler', np.float64)
    assert np.abs(eu - ref64).max() > 1e-3          # the correction is not a no-op in these cases


def test_drift_only_control_is_a_quadrature():
    """io=0, no=0: f depends on X(t) only, so y_N = y0 + sum_n f(X(t_n)) h_n exactly as the oracle sums it."""
    pr = make_problem(5, 0, 0, 2, 10, 16, 4, 12, nan_frac=0.0)
    ys, _ = hip_solve(pr, [0, 11], 1.0, dW=None, seed=1)
    p64 = O.cast_params(pr['params'], np.float64)
    acc = pr['y0'].astype(np.float64)
    for n in range(11):
        Xt = O.spline_evaluate(pr['coeffs'].astype(np.float64), pr['times'].astype(np.float64), float(n))
        acc = acc + O.drift_f(p64, 0, float(n), acc, Xt) * 1.0
    np.testing.assert_allclose(ys[-1], acc, rtol=1e-5, atol=1e-5)


# ---- in-kernel Philox ------------------------------------------------------------------------------
def test_philox_increments_match_specification():
    pr = make_problem(11, 4, 17, 2, 40, 32, 5, 9)
    ts, dt = [0, 2.5, 8], 0.5
    ys, call = hip_solve(pr, ts, dt, dW=None, seed=0x1234_5678_9ABC_DEF0, row_offset=1000, save_dW=True)
    t0, t1, *_ = O.step_grid(np.asarray(ts, np.float32), dt)
    exp = O.philox_dW(0x1234_5678_9ABC_DEF0, 1000, 40, 32, t0, t1)
    got = call.dW_out.cpu().numpy()
    np.testing.assert_allclose(got, exp, rtol=2e-6, atol=2e-7)
    # and the trajectory is the Euler scheme on exactly those increments
    ref64, _ = oracle_solve(pr, ts, dt, got, 'euler', np.float64)
    cpu32, _ = oracle_solve(pr, ts, dt, got, 'euler', np.float32)
    assert_parity(ys, ref64, cpu32, what='philox')


@pytest.mark.parametrize('cfg', [(3, 18, 'milstein'), (1, 15, 'milstein'), (1, 18, 'srk'), (4, 19, 'euler')])
def test_philox_increments_of_the_generic_kernels_with_a_diffusion_net(cfg):
    """In-kernel increments of the kernels that serve the diffusion nets outside the MFMA path (Milstein kernel, generic SRK,
    generic Euler): the specified Philox stream, written out, and the trajectory is the scheme on exactly those increments."""
    io, no, method = cfg
    pr = make_problem(14, io, no, 2, 21, 24, 5, 9)
    ts, dt = [0, 2.5, 8], 0.5
    ys, call = hip_solve(pr, ts, dt, dW=None, seed=0xFEED_5EED, row_offset=300, save_dW=True, method=method, kernel='generic')
    t0, t1, *_ = O.step_grid(np.asarray(ts, np.float32), dt)
    got = call.dW_out.cpu().numpy()
    np.testing.assert_allclose(got, O.philox_dW(0xFEED_5EED, 300, 21, 24, t0, t1), rtol=2e-6, atol=2e-7)
    dU = call.dU_out.cpu().numpy() if method == 'srk' else None
    again, _ = hip_solve(pr, ts, dt, dW=got, dU=dU, method=method, kernel='generic')
    np.testing.assert_array_equal(ys, again)            # replaying the written increments: the same bits
    if method != 'srk':
        ref64, _ = oracle_solve(pr, ts, dt, got, method, np.float64)
        cpu32, _ = oracle_solve(pr, ts, dt, got, method, np.float32)
        assert_parity(ys, ref64, cpu32, what=f'philox {cfg}')


def test_philox_statistics_and_seed_sensitivity():
    pr = make_problem(12, 2, 16, 2, 512, 64, 3, 5)
    _, c1 = hip_solve(pr, [0, 4], 1.0, seed=1, save_dW=True)
    _, c2 = hip_solve(pr, [0, 4], 1.0, seed=2, save_dW=True)
    z = c1.dW_out.cpu().numpy()
    assert abs(z.mean()) < 1e-2 and abs(z.std() - 1) < 1e-2
    assert abs((z ** 4).mean() - 3) < 0.1
    assert not np.array_equal(z, c2.dW_out.cpu().numpy())
    # distinct steps / rows are uncorrelated
    assert abs(np.corrcoef(z[0].ravel(), z[1].ravel())[0, 1]) < 2e-2


def test_batch_shards_reproduce_the_global_solve_bitwise():
    """Rows are independent and Philox counters use the GLOBAL row: any sharding (1/2/4/8 GPUs) gives
    bit-identical trajectories."""
    pr = make_problem(13, 6, 17, 2, 96, 32, 5, 9)
    full, _ = hip_solve(pr, [0, 3, 8], 1.0, seed=77)
    for nshard in (2, 4, 8):
        per = 96 // nshard
        parts = [hip_solve(pr, [0, 3, 8], 1.0, seed=77, row_offset=r * per, rows=slice(r * per, (r + 1) * per))[0]
                 for r in range(nshard)]
        np.testing.assert_array_equal(np.concatenate(parts, axis=1), full)


# ---- drop-in API ------------------------------------------------------------------------------------
class _ReplayBM:
    def __init__(self, dW, dU=None):
        self.dW, self.dU, self.n = dW, dU, 0

    def __call__(self, ta, tb, return_U=False):
        out = self.dW[self.n]
        u = self.dU[self.n] if self.dU is not None else None
        self.n += 1
        return (out, u) if return_U else out


def test_sdeint_drop_in_on_cuda_dispatches_to_hip():
    pr = make_problem(21, 4, 17, 2, 20, 32, 5, 9)
    m = S.Diffusion_model(5, 32, 32, 2, input_option=4, noise_option=17)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()})
    m = m.to(DEV)
    m.set_X(torch.from_numpy(pr['coeffs']).to(DEV), torch.from_numpy(pr['times']).to(DEV))
    ts = torch.tensor([0., 3., 8.], device=DEV)
    dW = draw_dW(21, [0, 3, 8], 1.0, 20, 32)
    with torch.no_grad():
        ys = S.sdeint(sde=m, y0=torch.from_numpy(pr['y0']).to(DEV), ts=ts, dt=1.0, method='euler',
                      options={'dt': 1.0}, bm=_ReplayBM(torch.from_numpy(dW).to(DEV)))
    ref64, _ = oracle_solve(pr, [0, 3, 8], 1.0, dW, 'euler', np.float64)
    assert_parity(ys.cpu().numpy(), ref64, what='sdeint')
    # with grads enabled the same call goes through the differentiable fused solve (HIP forward + HIP adjoint)
    ys_g = S.sdeint(sde=m, y0=torch.from_numpy(pr['y0']).to(DEV), ts=ts, dt=1.0, method='euler',
                    bm=_ReplayBM(torch.from_numpy(dW).to(DEV)))
    assert ys_g.requires_grad and ys_g.grad_fn is not None
    np.testing.assert_allclose(ys_g.detach().cpu().numpy(), ys.cpu().numpy(), rtol=1e-6, atol=1e-6)
    # vector-field probes go through the HIP kernel too
    with torch.no_grad():
        f = m.f(torch.tensor(2.0), torch.from_numpy(pr['y0']).to(DEV)).cpu().numpy()
    p64 = O.cast_params(pr['params'], np.float64)
    Xt = O.spline_evaluate(pr['coeffs'].astype(np.float64), pr['times'].astype(np.float64), 2.0)
    np.testing.assert_allclose(f, O.drift_f(p64, 4, 2.0, pr['y0'].astype(np.float64), Xt), rtol=1e-5, atol=2e-6)
    # f and g of one (t, y) share a launch, but a parameter edit through .data (no version bump) between two f calls is seen
    with torch.no_grad():
        t2, y2 = torch.tensor(2.0), torch.from_numpy(pr['y0']).to(DEV)
        f0 = m.f(t2, y2)
        assert m._fg_cache[2] == {0} and m._fg_cache[3] is y2
        g0 = m.g(t2, y2)
        assert m._fg_cache is None                 # both halves of the launch handed out
        m.linear_out.bias.data.add_(0.25)
        f1 = m.f(t2, y2)
        assert not torch.equal(f1, f0) and torch.equal(m.g(t2, y2), g0)
        # temporaries: the allocator hands f's freed argument address to g's argument; the cached half must not be served
        for _ in range(4):
            d = torch.full_like(y2, 0.5)
            m.f(t2, y2 + d)
            g_tmp = m.g(t2, y2 - d)
            keep = y2 - d
            m.f(t2, y2 * 0 + 7.0)                  # evict
            assert torch.equal(g_tmp, m.g(t2, keep))


def test_neuralsde_forward_on_cuda():
    pr = make_problem(22, 6, 17, 2, 12, 16, 4, 9)
    torch.manual_seed(0)
    model, field = S.make_sde_model('neuralgsde', 4, 3, 16, 16, 2, initial=True)
    model = model.to(DEV).eval()
    times = torch.from_numpy(pr['times']).to(DEV)
    fi = torch.tensor([8, 3, 3, 5, 0, 8, 1, 2, 7, 8, 4, 4], device=DEV)
    with torch.no_grad():
        out = model(times, [torch.from_numpy(pr['coeffs']).to(DEV)], fi, options={'seed': 5})
        out2 = model(times, [torch.from_numpy(pr['coeffs']).to(DEV)], fi, options={'seed': 5})
    assert out.shape == (12, 3) and torch.isfinite(out).all()
    assert torch.equal(out, out2)


# ---- full-size configuration (BASELINE.json configs[1]) ---------------------------------------------
def test_k2_full_size_parity_and_properties():
    B, H, C, L, N = 1024, 128, 21, 101, 100
    pr = make_problem(1234, 4, 17, 2, B, H, C, L, nan_frac=0.3)
    ts, dt = [0, N], 1.0
    dW = draw_dW(2024, ts, dt, B, H)
    ys, _ = hip_solve(pr, ts, dt, dW=dW)
    ref64, _ = oracle_solve(pr, ts, dt, dW, 'euler', np.float64)
    cpu32, _ = oracle_solve(pr, ts, dt, dW, 'euler', np.float32)
    rep = assert_parity(ys, ref64, cpu32, what='K2')
    print('K2 parity', rep)
    # determinism + shard invariance at full size with in-kernel Philox
    a, _ = hip_solve(pr, ts, dt, seed=2024)
    b, _ = hip_solve(pr, ts, dt, seed=2024)
    np.testing.assert_array_equal(a, b)
    halves = [hip_solve(pr, ts, dt, seed=2024, row_offset=o, rows=slice(o, o + 512))[0] for o in (0, 512)]
    np.testing.assert_array_equal(np.concatenate(halves, axis=1), a)
    assert np.isfinite(a).all()


# ---- MFMA fast path (both tile flavours) ---------------------------------------------------------------
MFMA_CASES = [
    # io, no, NL, B, H, C, L, ts, dt, method
    (4, 17, 2, 37, 128, 21, 13, [0, 12], 1.0, 'euler'),
    (6, 17, 2, 50, 128, 21, 21, [0, 3, 4, 11, 20], 1.0, 'euler'),
    (2, 16, 2, 18, 128, 5, 20, None, 0.02, 'euler'),
    (4, 17, 1, 16, 128, 32, 9, [0, 8], 1.0, 'euler'),
    (6, 16, 1, 7, 128, 3, 9, [0, 2.5, 8], 0.5, 'euler'),
    (4, 17, 2, 33, 128, 14, 13, [0, 5, 12], 1.0, 'milstein'),
    (6, 13, 2, 20, 128, 17, 9, [0, 8], 1.0, 'milstein'),
    (2, 12, 2, 20, 128, 16, 9, [0, 8], 1.0, 'euler'),
    (4, 0, 2, 20, 128, 21, 9, [0, 8], 1.0, 'euler'),
    (4, 17, 2, 70, 64, 21, 13, [0, 12], 1.0, 'euler'),
    (6, 17, 2, 9, 64, 7, 9, None, None, 'milstein'),
    (2, 16, 1, 21, 64, 2, 20, None, 0.02, 'euler'),
    (4, 17, 2, 19, 32, 5, 13, [0, 12], 1.0, 'euler'),
    (2, 16, 1, 256, 32, 2, 20, None, 0.02, 'euler'),          # K1 shape on the GPU
    (1, 0, 2, 11, 64, 3, 8, [0, 2.5, 7], 1.0, 'euler'),
    (3, 17, 2, 11, 32, 3, 8, [0, 7], 1.0, 'euler'),
    (5, 16, 1, 11, 128, 3, 8, [0, 7], 0.5, 'euler'),
    (1, 13, 2, 11, 128, 3, 8, [0, 7], 0.5, 'milstein'),
    (3, 0, 1, 40, 64, 3, 8, [0, 7], 1.0, 'euler'),
    (4, 17, 3, 13, 128, 21, 9, [0, 8], 1.0, 'euler'),          # NL = 3, 4: more hidden `linears`
    (6, 17, 4, 13, 128, 5, 9, [0, 3.5, 8], 1.0, 'euler'),
    (2, 16, 4, 9, 64, 3, 9, [0, 8], 1.0, 'milstein'),
    (4, 17, 2, 21, 16, 3, 9, [0, 8], 1.0, 'euler'),            # H = 16
    (1, 0, 3, 7, 16, 3, 9, [0, 8], 0.5, 'euler'),
    (1, 18, 2, 19, 64, 3, 8, [0, 7], 1.0, 'euler'),            # naivesde: diffusion net on [t, y]
    (3, 18, 2, 23, 64, 69, 8, [0, 3, 7], 1.0, 'euler'),        # K4 model (C = 69 is unused by input_option 3)
    (3, 19, 1, 9, 128, 3, 8, [0, 7], 0.5, 'euler'),
    (1, 14, 2, 9, 32, 3, 8, [0, 7], 1.0, 'euler'),
    (3, 15, 3, 9, 16, 3, 8, [0, 7], 1.0, 'euler'),
    (4, 3, 2, 21, 64, 5, 9, [0, 3.5, 8], 1.0, 'milstein'),        # closed-form table noise: exp(sigma) y
    (2, 5, 1, 9, 32, 3, 8, [0, 7], 0.5, 'euler'),                 # exp(sigma_diag) t
    (6, 11, 2, 13, 128, 5, 9, [0, 8], 1.0, 'milstein'),           # t y
    (1, 1, 2, 9, 16, 3, 8, [0, 7], 1.0, 'euler'),
    (3, 6, 3, 9, 64, 3, 8, [0, 2.5, 7], 1.0, 'milstein'),
    (5, 2, 2, 9, 32, 3, 8, [0, 7], 0.5, 'euler'),
    (4, 4, 2, 9, 32, 3, 8, [0, 7], 1.0, 'euler'),
    (0, 17, 2, 13, 64, 5, 8, [0, 7], 1.0, 'milstein'),            # y-free drift: z = initial_network(X(t))
    (0, 0, 1, 9, 32, 3, 8, [0, 3, 7], 1.0, 'euler'),
    (0, 18, 3, 9, 128, 21, 8, [0, 7], 1.0, 'euler'),
    (0, 9, 2, 9, 16, 40, 8, [0, 7], 0.5, 'euler'),
    (2, 18, 2, 13, 64, 5, 8, [0, 7], 1.0, 'euler'),               # ... and on the time-free embedded drift
    (2, 15, 1, 9, 32, 3, 8, [0, 3, 7], 1.0, 'euler'),
    (4, 18, 2, 13, 64, 5, 8, [0, 7], 1.0, 'euler'),               # diffusion nets behind the control embedding (folded layer)
    (6, 15, 1, 9, 32, 3, 8, [0, 3, 7], 1.0, 'euler'),
    (4, 14, 3, 9, 128, 21, 8, [0, 7], 1.0, 'euler'),
    (6, 19, 2, 9, 16, 3, 8, [0, 7], 0.5, 'euler'),
    (5, 18, 2, 13, 64, 3, 8, [0, 7], 1.0, 'euler'),               # diffusion nets on the geometric drift
    (5, 14, 1, 9, 32, 3, 8, [0, 3, 7], 1.0, 'euler'),
    (4, 7, 2, 9, 32, 3, 8, [0, 7], 1.0, 'euler'),                 # y-only closed forms: sqrt y (NaN -> 0 for y < 0)
    (2, 8, 2, 9, 64, 3, 8, [0, 3, 7], 0.5, 'milstein'),           # y^3
    (6, 9, 1, 13, 128, 5, 8, [0, 7], 1.0, 'milstein'),            # sigmoid y
    (3, 10, 3, 9, 16, 3, 8, [0, 7], 1.0, 'euler'),                # relu y
    (4, 17, 2, 37, 128, 69, 9, [0, 3.5, 8], 1.0, 'euler'),        # wide control path (sepsis channel counts): 5 k-blocks
    (2, 16, 1, 21, 64, 35, 9, [0, 8], 1.0, 'milstein'),
    (6, 17, 3, 9, 32, 80, 7, [0, 6], 0.5, 'euler'),
    (4, 17, 2, 21, 256, 14, 11, [0, 4.5, 10], 1.0, 'milstein'),   # H = 256: weights streamed from L2 (K5 model)
    (6, 17, 1, 9, 256, 5, 9, [0, 8], 1.0, 'euler'),
    (1, 18, 2, 9, 256, 3, 8, [0, 7], 1.0, 'euler'),
    # H = 256 on the streamed-weight lean kernel (snsde_m4s_kernel.h): 0 / 1 / 2 hidden layers, no xt block (latent-only
    # drift), 1-3 xt blocks, ragged last tile, off-grid outputs, supplied and table diffusions
    (1, 16, 1, 7, 256, 3, 8, [0, 2.5, 7], 1.0, 'milstein'),
    (2, 12, 3, 10, 256, 20, 9, [0, 8], 0.5, 'euler'),
    (3, 5, 2, 5, 256, 3, 8, [0, 7], 1.0, 'milstein'),
    (4, 13, 3, 6, 256, 40, 8, [0, 3, 7], 1.0, 'euler'),
    (5, 3, 2, 9, 256, 3, 12, [0, 11], 1.0, 'milstein'),
    (6, 17, 2, 3, 256, 30, 9, [0, 8], 1.0, 'euler'),
    (4, 9, 1, 9, 256, 17, 8, [0, 7], 1.0, 'euler'),
]


@pytest.mark.parametrize('kernel', ['mfma16', 'mfma4', 'mfma16x', 'mfma4x'])
@pytest.mark.parametrize('ci', range(len(MFMA_CASES)))
def test_mfma_trajectory_vs_oracle(ci, kernel):
    io, no, NL, B, H, C, L, ts, dt, method = MFMA_CASES[ci]
    times = np.linspace(0, 1, L).astype(np.float32) if ts is None else None
    pr = make_problem(300 + ci, io, no, NL, B, H, C, L, times=times)
    if ts is None:
        ts = pr['times']
        dt = dt or max(float(np.diff(pr['times']).min()), 1e-3)
    dW = draw_dW(300 + ci, ts, dt, B, H)
    ys, call = hip_solve(pr, ts, dt, dW=dW, method=method, save_traj=True, kernel=kernel)
    ref64, traj64 = oracle_solve(pr, ts, dt, dW, method, np.float64)
    cpu32, _ = oracle_solve(pr, ts, dt, dW, method, np.float32)
    assert_parity(ys, ref64, cpu32, what=f'mfma case {ci} {kernel}')
    assert_parity(call.traj.cpu().numpy(), traj64, what=f'mfma traj {ci} {kernel}')
    np.testing.assert_array_equal(ys[0], pr['y0'])


def test_mfma_unsupported_configuration_is_refused_not_silently_rerouted():
    pr = make_problem(1, 0, 18, 2, 8, 64, 40, 5)     # diffusion net on a y-free drift with a wide control path: generic only
    with pytest.raises(S._lib.SnsdeError) as e:
        hip_solve(pr, [0, 4], 1.0, dW=draw_dW(1, [0, 4], 1.0, 8, 64), kernel='mfma')
    assert e.value.code == -4
    pr = make_problem(1, 4, 17, 2, 8, 48, 3, 5)     # H not instantiated
    with pytest.raises(S._lib.SnsdeError):
        hip_solve(pr, [0, 4], 1.0, dW=draw_dW(1, [0, 4], 1.0, 8, 48), kernel='mfma16')
    # 'auto' takes the generic kernel for those
    ys, _ = hip_solve(pr, [0, 4], 1.0, dW=draw_dW(1, [0, 4], 1.0, 8, 48), kernel='auto')
    assert np.isfinite(ys).all()


@pytest.mark.parametrize('H,method', [(32, 'euler'), (64, 'milstein'), (128, 'euler'), (256, 'milstein'), (256, 'euler')])
def test_long_solves_cross_the_step_table_chunks(H, method):
    """More than 128 solver steps: the kernels re-stage their step-table rows in LDS chunk by chunk (forward: lean /
    streamed kernels; backward: the adjoint kernel).  States vs the float64 oracle, gradients vs float64 autograd, with two
    off-grid outputs among the 300 steps."""
    io, no, NL, B, C, L = 4, 17, 2, 9, 5, 9
    pr = make_problem(700 + H, io, no, NL, B, H, C, L)
    ts = np.asarray([0.0, 3.1, 8.0], np.float32)
    dt = 8.0 / 300
    dW = draw_dW(700 + H, ts, dt, B, H)
    assert dW.shape[0] > 256
    ys, _ = hip_solve(pr, ts, dt, dW=dW, method=method, kernel='mfma4')
    ref64, _ = oracle_solve(pr, ts, dt, dW, method, np.float64)
    cpu32, _ = oracle_solve(pr, ts, dt, dW, method, np.float32)
    assert_parity(ys, ref64, cpu32, what=f'long solve H={H}')
    _check_backward(700 + H, io, no, NL, B, H, C, L, list(ts), dt, method, 'mfma4')


@pytest.mark.parametrize('H', [128, 256])       # 256: the streamed-weight kernel keeps its Philox normals in registers
@pytest.mark.parametrize('kernel', ['mfma16', 'mfma4'])
def test_mfma_philox_spec_and_shard_invariance(kernel, H):
    pr = make_problem(31, 4, 17, 2, 96, H, 21, 9)
    ts, dt = [0, 2.5, 8], 0.5
    ys, call = hip_solve(pr, ts, dt, dW=None, seed=0xABCDEF0123, row_offset=4096, save_dW=True, kernel=kernel)
    t0, t1, *_ = O.step_grid(np.asarray(ts, np.float32), dt)
    got = call.dW_out.cpu().numpy()
    np.testing.assert_allclose(got, O.philox_dW(0xABCDEF0123, 4096, 96, H, t0, t1), rtol=2e-6, atol=2e-7)
    ref64, _ = oracle_solve(pr, ts, dt, got, 'euler', np.float64)
    assert_parity(ys, ref64, what='mfma philox')
    # the increments are the same stream the generic kernel draws
    _, cg = hip_solve(pr, ts, dt, dW=None, seed=0xABCDEF0123, row_offset=4096, save_dW=True, kernel='generic')
    np.testing.assert_array_equal(got, cg.dW_out.cpu().numpy())
    full, _ = hip_solve(pr, ts, dt, seed=9, kernel=kernel)
    for nshard in (2, 4):
        per = 96 // nshard
        parts = [hip_solve(pr, ts, dt, seed=9, row_offset=k * per, rows=slice(k * per, (k + 1) * per), kernel=kernel)[0]
                 for k in range(nshard)]
        np.testing.assert_array_equal(np.concatenate(parts, axis=1), full)


@pytest.mark.parametrize('kernel', ['mfma16', 'mfma4', 'mfma4x'])
def test_k2_full_size_mfma(kernel):
    B, H, C, L, N = 1024, 128, 21, 101, 100
    pr = make_problem(1234, 4, 17, 2, B, H, C, L, nan_frac=0.3)
    ts, dt = [0, N], 1.0
    dW = draw_dW(2024, ts, dt, B, H)
    ys, _ = hip_solve(pr, ts, dt, dW=dW, kernel=kernel)
    ref64, _ = oracle_solve(pr, ts, dt, dW, 'euler', np.float64)
    cpu32, _ = oracle_solve(pr, ts, dt, dW, 'euler', np.float32)
    print('K2 parity', kernel, assert_parity(ys, ref64, cpu32, what='K2 ' + kernel))
    a, _ = hip_solve(pr, ts, dt, seed=2024, kernel=kernel)
    b, _ = hip_solve(pr, ts, dt, seed=2024, kernel=kernel)
    np.testing.assert_array_equal(a, b)
    halves = [hip_solve(pr, ts, dt, seed=2024, row_offset=o, rows=slice(o, o + 512), kernel=kernel)[0] for o in (0, 512)]
    np.testing.assert_array_equal(np.concatenate(halves, axis=1), a)


def test_solve_is_hip_graph_capturable_and_replayable():
    """The C ABI only enqueues kernels (no allocation, no sync): a prepared solve captures into a HIP graph and
    replays to the same result."""
    pr = make_problem(41, 4, 17, 2, 64, 128, 21, 9)
    io, no, NL, C, H = 4, 17, 2, 21, 128
    model = S.engine.model_struct(C, H, H, NL, io, no)
    flat = flat_params(pr['params'], io, no, NL, C, H)
    grid = S.engine.step_grid(np.array([0, 3, 8], np.float32), 1.0, pr['times'], torch.device(DEV))
    call = S.engine.SolveCall(model, flat, torch.from_numpy(pr['coeffs']).to(DEV), grid,
                              torch.from_numpy(pr['y0']).to(DEV), seed=3)
    eager = call.launch().clone()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        call.launch(side)                      # warm-up on the capture stream
        side.synchronize()
        with torch.cuda.graph(g, stream=side):
            call.launch(side)
    call.ys.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(call.ys, eager)


def test_solve_call_reuses_its_prepared_workspace_until_the_parameters_move():
    """SolveCall.launch(auto_reuse=True) sets SNSDE_FLAG_REUSE_PREPARED itself from the second launch on (no weight packing / table
    launch) while the parameter block's version counter stands; an in-place update of the block makes the next launch prepare
    again.  The default (auto_reuse=False) prepares on every launch - writes through `.data` aliases do not move the counter."""
    pr = make_problem(43, 4, 17, 2, 64, 128, 21, 9)
    io, no, NL, C, H = 4, 17, 2, 21, 128
    model = S.engine.model_struct(C, H, H, NL, io, no)
    flat = flat_params(pr['params'], io, no, NL, C, H)
    grid = S.engine.step_grid(np.array([0, 3, 8], np.float32), 1.0, pr['times'], torch.device(DEV))
    args = (torch.from_numpy(pr['coeffs']).to(DEV), grid, torch.from_numpy(pr['y0']).to(DEV))
    call = S.engine.SolveCall(model, flat, *args, seed=3)
    first = call.launch(auto_reuse=True).clone()
    assert not (call.desc.flags & S._lib.FLAG_REUSE_PREPARED)
    second = call.launch(auto_reuse=True).clone()
    assert call.desc.flags & S._lib.FLAG_REUSE_PREPARED and torch.equal(first, second)
    assert torch.equal(call.launch(), first) and not (call.desc.flags & S._lib.FLAG_REUSE_PREPARED)      # the default prepares
    flat.mul_(1.05)                                   # an optimizer step on the block
    third = call.launch(auto_reuse=True).clone()
    assert not (call.desc.flags & S._lib.FLAG_REUSE_PREPARED) and not torch.equal(third, first)
    fresh = S.engine.SolveCall(model, flat, *args, seed=3).launch()
    assert torch.equal(third, fresh)
    # a write through an alias with its own version counter (what an optimizer stepping on the arena's views does): the
    # default launch sees it, a stale auto_reuse launch would not
    flat.data[:16] += 0.25
    assert not torch.equal(call.launch(), third)
    assert torch.equal(call.launch(), S.engine.SolveCall(model, flat, *args, seed=3).launch())


def test_a_captured_launch_leaves_no_prepared_key_behind():
    """A launch recorded into a graph has not run: it must not mark the workspace as prepared for a later eager auto_reuse launch."""
    pr = make_problem(44, 4, 17, 2, 64, 128, 21, 9)
    io, no, NL, C, H = 4, 17, 2, 21, 128
    model = S.engine.model_struct(C, H, H, NL, io, no)
    flat = flat_params(pr['params'], io, no, NL, C, H)
    grid = S.engine.step_grid(np.array([0, 3, 8], np.float32), 1.0, pr['times'], torch.device(DEV))
    args = (model, flat, torch.from_numpy(pr['coeffs']).to(DEV), grid, torch.from_numpy(pr['y0']).to(DEV))
    call = S.engine.SolveCall(*args, seed=3)
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        S.engine.SolveCall(*args, seed=3).launch(side)      # code objects loaded by ANOTHER call object before the capture
        side.synchronize()
        with torch.cuda.graph(g, stream=side):
            call.launch(side, auto_reuse=True)
    assert getattr(call, '_prep_key', None) is None
    eager = call.launch(auto_reuse=True).clone()      # nothing prepared yet: this launch must prepare
    assert not (call.desc.flags & S._lib.FLAG_REUSE_PREPARED)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(call.ys, eager)


# ---- the other BASELINE.json configurations as full-size parity cases (forward) --------------------------
def test_k3_gsde_per_gpu_shard_full_size():
    """configs[2]: Neural GSDE (6,17), H=128, 200 steps, Hermite coefficients; 4096 rows over 8 GPUs = 512 per GPU."""
    B, H, C, L = 512, 128, 21, 201
    pr = make_problem(3003, 6, 17, 2, B, H, C, L, nan_frac=0.0, hermite=True)
    ts, dt = [0, L - 1], 1.0
    dW = draw_dW(3003, ts, dt, B, H)
    ref64, _ = oracle_solve(pr, ts, dt, dW, 'euler', np.float64)
    cpu32, _ = oracle_solve(pr, ts, dt, dW, 'euler', np.float32)
    for kernel in ('mfma4', 'mfma16', 'generic'):
        ys, _ = hip_solve(pr, ts, dt, dW=dW, kernel=kernel)
        # 200 GSDE steps amplify fp32 round-off on isolated rows (CPU fp32 is equally far from fp64): relative criterion
        print('K3', kernel, assert_parity(ys, ref64, cpu32, what='K3 ' + kernel, amplifying=True))


def test_bench_sized_single_gpu_solves_vs_oracle():
    """The sizes bench.py times on ONE GPU in its `extra` legs, at full size against the fp64 oracle on replayed increments:
    K3 with all 4096 rows (16-row tiles chosen by 'auto', 200 steps) and K5 with all 1024 rows (H = 256 Milstein, LDS-ring
    streamed weights on 4-row tiles, 50 outputs)."""
    B, H, C, L = 4096, 128, 21, 201
    pr = make_problem(3004, 6, 17, 2, B, H, C, L, nan_frac=0.0, hermite=True)
    ts, dt = [0, L - 1], 1.0
    dW = draw_dW(3004, ts, dt, B, H)
    ref64, _ = oracle_solve(pr, ts, dt, dW, 'euler', np.float64)
    cpu32, _ = oracle_solve(pr, ts, dt, dW, 'euler', np.float32)
    assert S.engine.forward_path(S.engine.model_struct(C, H, H, 2, 6, 17), B, L, L - 1) == 'mfma16'
    ys, _ = hip_solve(pr, ts, dt, dW=dW, kernel='auto')
    print('K3 4096 rows', assert_parity(ys, ref64, cpu32, what='K3 4096 rows auto', amplifying=True))
    B, H, C, L = 1024, 256, 14, 50
    pr = make_problem(5006, 4, 17, 2, B, H, C, L, nan_frac=0.3)
    ts, dt = pr['times'], 1.0
    dW = draw_dW(5006, ts, dt, B, H)
    ref64, _ = oracle_solve(pr, ts, dt, dW, 'milstein', np.float64)
    cpu32, _ = oracle_solve(pr, ts, dt, dW, 'milstein', np.float32)
    assert S.engine.forward_path(S.engine.model_struct(C, H, H, 2, 4, 17), B, L, L - 1, method='milstein') == 'lean-streamed'
    ys, _ = hip_solve(pr, ts, dt, dW=dW, method='milstein', kernel='auto')
    assert ys.shape == (50, B, H)
    print('K5 1024 rows', assert_parity(ys, ref64, cpu32, what='K5 1024 rows auto'))


def test_k4_sepsis_shaped_nsde_full_size():
    """configs[3]: Neural SDE (3,18), B=2048, H=64, C=69, times=linspace(1,72,72), per-row lengths, z0 supplied,
    ts = the distinct final times (T ~ 70)."""
    B, H, C, L = 2048, 64, 69, 72
    times = np.linspace(1, 72, 72).astype(np.float32)
    pr = make_problem(4004, 3, 18, 2, B, H, C, L, times=times, nan_frac=0.1)
    rng = np.random.default_rng(4)
    final_index = rng.integers(2, L, size=B)
    uniq = np.unique(final_index)
    uniq = uniq[(uniq != 0) & (uniq != L - 1)]
    ts = np.concatenate([times[:1], times[uniq], times[-1:]])
    dt = 1.0
    dW = draw_dW(4004, ts, dt, B, H)
    ref64, _ = oracle_solve(pr, ts, dt, dW, 'euler', np.float64)
    cpu32, _ = oracle_solve(pr, ts, dt, dW, 'euler', np.float32)
    model = S.engine.model_struct(C, H, H, 2, 3, 18)
    assert S.engine.forward_path(model, B, L, len(times) - 1) == 'w4'      # `auto` = the wave-pair kernels (snsde_w4_euler_kernel) since round 5
    for kernel in ('auto', 'w4', 'mfma4', 'mfma16', 'generic'):
        ys, _ = hip_solve(pr, ts, dt, dW=dW, kernel=kernel)
        assert ys.shape[0] == len(ts)
        print('K4', kernel, assert_parity(ys, ref64, cpu32, what='K4 ' + kernel))
    # ... and under torch_ists' default method (nsde_model.py:63-74), SRI2W1 through the net at the full K4 size
    rng2 = np.random.default_rng(4005)
    g0, g1 = O.step_grid(ts, dt)[:2]
    hh = (g1 - g0).astype(np.float32).reshape(-1, 1, 1)
    dU = (hh * (0.5 * dW + np.sqrt(hh / 12) * rng2.standard_normal(dW.shape).astype(np.float32))).astype(np.float32)
    ref64s, _ = O.solve_diffusion_model(pr['params'], 3, 18, pr['coeffs'], pr['times'], pr['y0'], ts, dt, dW, method='srk', dtype=np.float64, dU=dU)
    cpu32s, _ = O.solve_diffusion_model(pr['params'], 3, 18, pr['coeffs'], pr['times'], pr['y0'], ts, dt, dW, method='srk', dtype=np.float32, dU=dU)
    assert S.engine.forward_path(model, B, L, len(times) - 1, method='srk') == 'w4'
    for kernel in ('auto', 'mfma4'):
        ys, _ = hip_solve(pr, ts, dt, dW=dW, dU=dU, method='srk', kernel=kernel)
        print('K4 srk', kernel, assert_parity(ys, ref64s, cpu32s, what='K4 srk ' + kernel))


@pytest.mark.parametrize('case', [(4, 17, 2, 14, 37, 'milstein'), (4, 17, 2, 14, 128, 'euler'), (6, 16, 2, 21, 9, 'euler'), (1, 13, 1, 3, 21, 'milstein'),
                                  (3, 12, 2, 3, 5, 'euler'), (4, 9, 2, 40, 12, 'euler'), (2, 17, 3, 14, 8, 'euler')])
@pytest.mark.parametrize('train', [False, True])
def test_h256_two_tile_kernel_is_bit_identical_to_the_streamed_one(case, train):
    """H = 256 on 4-row tiles (round 6): eight waves of two tiles with the first 4 k-blocks of every layer in registers
    (snsde_m4s2_kernel.h) keep the k order and accumulator chains of the fully streamed sixteen-wave kernel (snsde_m4s_kernel.h, kept
    behind SNSDE_FLAG_STREAM_ALL): every output, the trajectory and every saved plane agree bit for bit - Philox and supplied
    increments, ragged tiles, interpolated outputs.  (NL = 3 and wide control blocks fall back to the streamed kernel where the
    two-tile instantiation would spill: equal by construction there.)"""
    io, no, NL, C, B, method = case
    pr = make_problem(6100 + B, io, no, NL, B, 256, C, 9)
    ts, dt = np.array([0., 2.5, 6., 8.], np.float32), 1.0
    model = S.engine.model_struct(C, 256, 256, NL, io, no)
    flat = flat_params(pr['params'], io, no, NL, C, 256)
    grid = S.engine.step_grid(ts, dt, pr['times'], torch.device(DEV))
    dW = draw_dW(6100 + B, ts, dt, B, 256)
    for supplied in (None, torch.from_numpy(dW).to(DEV)):
        outs = []
        for all_ in (True, False):
            call = S.engine.SolveCall(model, flat, torch.from_numpy(pr['coeffs']).to(DEV), grid, torch.from_numpy(pr['y0']).to(DEV),
                                      dW=supplied, method=method, seed=11, kernel='mfma4', stream_all=all_, save_traj=train, save_dW=train,
                                      save_act=train)
            ys = call.launch().clone()
            outs.append((ys, call.traj, call.act_save, call.dW_out))
        assert torch.isfinite(outs[0][0]).all()
        for x, y in zip(*outs):
            assert (x is None and y is None) or torch.equal(x, y)
    if no == 17 and io == 4:      # ... and the K5 model against the oracle, through the two-tile kernel
        ref64, _ = oracle_solve(pr, ts, dt, dW, method, np.float64)
        cpu32, _ = oracle_solve(pr, ts, dt, dW, method, np.float32)
        ys, _ = hip_solve(pr, ts, dt, dW=dW, method=method, kernel='mfma4')
        assert_parity(ys, ref64, cpu32, what='H=256 two-tile')


@pytest.mark.parametrize('train', [False, True])
@pytest.mark.parametrize('case', [(4, 17, 21, 64, 'euler'), (4, 17, 21, 37, 'milstein'), (6, 16, 5, 9, 'euler'), (3, 13, 3, 21, 'milstein')])
def test_h128_two_tile_kernel_is_bit_identical_to_the_lean_one(case, train):
    """SNSDE_FLAG_TWO_TILE (round 6 experiment): four waves of two tiles, one wave per SIMD, hidden / output weights pinned in
    AccVGPRs with asm-issued MFMAs - measured slower than the eight-wave lean kernel at K2 (214 vs 184 us), kept opt-in; its results are
    the lean kernel's bit for bit (same chains), which this pins together with the asm MFMAs' hand-placed wait states."""
    io, no, C, B, method = case
    pr = make_problem(6300 + B, io, no, 2, B, 128, C, 9)
    ts, dt = np.array([0., 2.5, 6., 8.], np.float32), 1.0
    model = S.engine.model_struct(C, 128, 128, 2, io, no)
    flat = flat_params(pr['params'], io, no, 2, C, 128)
    grid = S.engine.step_grid(ts, dt, pr['times'], torch.device(DEV))
    dW = torch.from_numpy(draw_dW(6300 + B, ts, dt, B, 128)).to(DEV)
    for supplied in (None, dW):
        outs = []
        for two in (False, True):
            call = S.engine.SolveCall(model, flat, torch.from_numpy(pr['coeffs']).to(DEV), grid, torch.from_numpy(pr['y0']).to(DEV), dW=supplied,
                                      method=method, seed=11, kernel='mfma4', two_tile=two, save_traj=train, save_dW=train, save_act=train)
            outs.append((call.launch().clone(), call.traj, call.act_save, call.dW_out))
        assert torch.isfinite(outs[0][0]).all()
        for x, y in zip(*outs):
            assert (x is None and y is None) or torch.equal(x, y)


@pytest.mark.parametrize('case', [(4, 17, 2, 14, 37, 'milstein', False), (4, 17, 2, 14, 128, 'euler', True), (6, 16, 2, 21, 9, 'euler', False),
                                  (1, 13, 1, 3, 21, 'milstein', True), (3, 9, 2, 3, 5, 'euler', False), (5, 3, 1, 4, 12, 'milstein', False)])
def test_h256_two_tile_adjoint_is_bit_identical_to_the_streamed_one(case):
    """The H = 256 adjoint on two tiles per wave (snsde_m4s2_rev_kernel.h, round 6) against the sixteen-wave streamed adjoint
    (SNSDE_FLAG_STREAM_ALL): dL/dy0, every adjoint state, every delta plane and the flat parameter gradient (which also sums the
    per-tile theta / table partials) bit for bit - supplied and Philox increments, per-row outputs, gated drifts (io 5 / 6), y-only and
    table diffusions."""
    io, no, NL, C, B, method, row_out = case
    pr = make_problem(6200 + B, io, no, NL, B, 256, C, 9)
    ts, dt = np.array([0., 2.5, 6., 8.], np.float32), 1.0
    model = S.engine.model_struct(C, 256, 256, NL, io, no)
    flat = flat_params(pr['params'], io, no, NL, C, 256)
    grid = S.engine.step_grid(ts, dt, pr['times'], torch.device(DEV))
    dW = torch.from_numpy(draw_dW(6200 + B, ts, dt, B, 256)).to(DEV)
    ro = torch.from_numpy(np.random.default_rng(5).integers(0, len(ts), size=B).astype(np.int32)).to(DEV) if row_out else None
    rng = np.random.default_rng(7)
    for supplied in (None, dW):
        outs = []
        for all_ in (True, False):
            call = S.engine.SolveCall(model, flat, torch.from_numpy(pr['coeffs']).to(DEV), grid, torch.from_numpy(pr['y0']).to(DEV), dW=supplied,
                                      method=method, seed=11, kernel='mfma4', stream_all=all_, save_traj=True, save_dW=supplied is None, save_act=True,
                                      row_out=ro)
            ys = call.launch()
            gy = torch.from_numpy(rng.standard_normal(tuple(ys.shape)).astype(np.float32)).to(DEV) if not outs else outs[0][-1]
            adj, delta = S.engine.solve_backward(call, gy, save_delta=True)
            grad = S.engine.param_gradients(call, adj, delta)
            outs.append((adj.clone(), delta.clone(), grad.clone(), gy))
        assert torch.isfinite(outs[0][2]).all() and float(outs[0][2].abs().max()) > 0
        for x, y in zip(outs[0][:3], outs[1][:3]):
            assert torch.equal(x, y)


def test_h256_two_tile_kernels_over_more_steps_than_one_table_chunk():
    """160 steps (> the 128-row chunk of the step table both kernels stage in LDS: fill_rows re-stages mid-solve), ragged batch: forward
    (training mode) and adjoint + gradients of the two-tile kernels bit for bit against the sixteen-wave ones."""
    io, no, NL, C, B, L = 4, 17, 2, 14, 23, 161
    pr = make_problem(6400, io, no, NL, B, 256, C, L, weight_scale=0.5)
    ts, dt = np.array([0., 77.5, 160.], np.float32), 1.0
    model = S.engine.model_struct(C, 256, 256, NL, io, no)
    flat = flat_params(pr['params'], io, no, NL, C, 256)
    grid = S.engine.step_grid(ts, dt, pr['times'], torch.device(DEV))
    assert grid.N == 160
    outs = []
    for all_ in (True, False):
        call = S.engine.SolveCall(model, flat, torch.from_numpy(pr['coeffs']).to(DEV), grid, torch.from_numpy(pr['y0']).to(DEV), method='milstein',
                                  seed=5, kernel='mfma4', stream_all=all_, save_traj=True, save_act=True)
        ys = call.launch().clone()
        gy = torch.ones_like(ys) if not outs else outs[0][-1]
        adj, grad = S.engine.backward_with_gradients(call, gy, adj0_only=True)
        outs.append((ys, call.traj.clone(), adj.clone(), grad.clone(), gy))
    assert torch.isfinite(outs[0][3]).all()
    for x, y in zip(outs[0][:4], outs[1][:4]):
        assert torch.equal(x, y)


def test_h256_two_tile_kernels_fuzz_against_the_streamed_ones():
    """60 random configurations (input / noise options, depth, channels, ragged batches, output grids with interpolated outputs, Euler /
    Milstein, supplied / Philox increments, per-row outputs): forward in training mode, adjoint and parameter gradients of the two-tile
    H = 256 kernels bit for bit against the sixteen-wave ones (tools/fuzz_h256.py; 300 cases were run when the kernels were written)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'fuzz_h256.py'), '60', '7'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert '60 cases, 0 mismatches' in r.stdout, r.stdout[-2000:]


def test_k5_milstein_h256_forecast_shaped():
    """configs[4] forward leg: (4,17) Milstein, H=256, MuJoCo-shaped L=50 C=14, ts = times (T=50), 128 rows per GPU."""
    B, H, C, L = 128, 256, 14, 50
    pr = make_problem(5005, 4, 17, 2, B, H, C, L, nan_frac=0.3)
    ts, dt = pr['times'], 1.0
    dW = draw_dW(5005, ts, dt, B, H)
    ref64, _ = oracle_solve(pr, ts, dt, dW, 'milstein', np.float64)
    cpu32, _ = oracle_solve(pr, ts, dt, dW, 'milstein', np.float32)
    for kernel in ('mfma4', 'mfma16', 'generic'):
        ys, _ = hip_solve(pr, ts, dt, dW=dW, method='milstein', kernel=kernel)
        assert ys.shape == (50, B, H)
        print('K5', kernel, assert_parity(ys, ref64, cpu32, what='K5 ' + kernel))


# ---- backward (adjoint) of the fused solve --------------------------------------------------------------
BWD_CASES = [
    # io, no, NL, B, H, C, L, ts, dt
    (4, 17, 2, 21, 32, 5, 9, [0, 8], 1.0),
    (4, 17, 2, 37, 128, 21, 13, [0, 5, 12], 1.0),
    (6, 17, 2, 18, 64, 7, 9, [0, 3, 8], 1.0),
    (2, 16, 1, 9, 32, 2, 12, None, 0.05),          # interpolated outputs at every knot
    (1, 0, 2, 11, 64, 3, 8, [0, 2.5, 7], 1.0),
    (3, 13, 2, 11, 32, 3, 8, [0, 7], 0.5),
    (5, 12, 1, 8, 128, 3, 8, [0, 7], 1.0),
    (4, 17, 4, 9, 64, 5, 9, [0, 8], 1.0),
    (6, 17, 3, 9, 16, 3, 9, [0, 8], 1.0),
    (4, 17, 2, 9, 256, 14, 9, [0, 3, 8], 1.0),
    (4, 17, 2, 9, 256, 14, 8, None, None, 'milstein'),        # K5 shape: H = 256, Milstein, ts = times
    (4, 17, 2, 21, 128, 5, 9, [0, 3.5, 8], 0.5, 'milstein'),
    (6, 17, 3, 9, 64, 3, 9, [0, 8], 1.0, 'milstein'),
    (3, 13, 2, 11, 32, 3, 8, [0, 7], 0.5, 'milstein'),
    (2, 16, 1, 9, 32, 2, 12, None, 0.05, 'milstein'),         # y-independent diffusion: Milstein term vanishes
    (4, 17, 2, 13, 64, 69, 9, [0, 3.5, 8], 1.0, 'euler'),     # wide control path (C = 69)
    (6, 17, 2, 9, 32, 35, 8, [0, 7], 1.0, 'milstein'),
    (3, 18, 2, 21, 64, 5, 9, [0, 3.5, 8], 1.0, 'euler'),      # diffusion nets on [tau, y] (BASELINE config 4's model)
    (1, 14, 2, 9, 32, 3, 8, [0, 7], 1.0, 'euler'),
    (3, 15, 1, 11, 128, 3, 8, [0, 7], 0.5, 'euler'),
    (1, 19, 3, 9, 16, 3, 8, [0, 2.5, 7], 1.0, 'euler'),
    (3, 19, 2, 9, 128, 3, 8, [0, 7], 1.0, 'euler'),
    (0, 17, 2, 9, 64, 5, 8, [0, 7], 1.0, 'milstein'),         # y-free drift (input_option 0)
    (0, 18, 2, 9, 32, 3, 8, [0, 3, 7], 1.0, 'euler'),
    (0, 4, 3, 9, 16, 40, 8, [0, 7], 1.0, 'euler'),
    (2, 18, 2, 9, 64, 5, 8, [0, 7], 1.0, 'euler'),            # time-free embedded drift + net: xaux keeps X first, tau after
    (2, 14, 1, 9, 32, 6, 8, [0, 3, 7], 1.0, 'euler'),
    (4, 18, 2, 9, 64, 5, 8, [0, 7], 1.0, 'euler'),            # diffusion nets behind the control embedding
    (6, 19, 2, 9, 32, 3, 8, [0, 3, 7], 1.0, 'euler'),
    (4, 15, 1, 9, 128, 21, 8, [0, 7], 0.5, 'euler'),
    (6, 14, 3, 9, 16, 3, 8, [0, 7], 1.0, 'euler'),
    (5, 18, 2, 9, 64, 3, 8, [0, 7], 1.0, 'euler'),            # geometric drift (tanh(y) gate) with a diffusion net
    (5, 15, 1, 9, 32, 3, 8, [0, 3, 7], 1.0, 'euler'),
]


GEN_BWD_CASES = [
    # io, no, NL, B, H, C, L, ts, dt, method    (generic adjoint kernel: any dims, Euler and Milstein)
    (4, 17, 2, 11, 24, 5, 9, [0, 3, 8], 1.0, 'euler'),
    (4, 17, 2, 9, 256, 14, 8, None, None, 'milstein'),        # K5 shape: H = 256, Milstein, ts = times
    (6, 17, 3, 9, 40, 3, 9, [0, 8], 1.0, 'milstein'),
    (2, 16, 1, 9, 20, 2, 12, None, 0.05, 'euler'),
    (0, 6, 2, 9, 12, 3, 8, [0, 7], 1.0, 'milstein'),
    (1, 8, 2, 9, 12, 3, 8, [0, 2.5, 7], 0.5, 'milstein'),
    (3, 9, 5, 9, 12, 3, 8, [0, 7], 0.5, 'milstein'),
    (5, 11, 2, 9, 12, 3, 8, [0, 7], 0.5, 'milstein'),
    (3, 3, 2, 9, 12, 3, 8, [0, 7], 0.5, 'euler'),
    (1, 10, 2, 9, 12, 3, 8, [0, 7], 0.5, 'euler'),
    (4, 17, 2, 11, 24, 5, 9, [0, 3.5, 8], 1.0, 'srk'),       # SRID2 adjoint (torch_ists default method)
    (6, 17, 3, 9, 40, 3, 9, [0, 8], 0.5, 'srk'),
    (2, 16, 1, 9, 20, 2, 12, None, 0.05, 'srk'),
    (1, 8, 2, 9, 12, 3, 8, [0, 2.5, 7], 0.5, 'srk'),
    (3, 13, 2, 10, 64, 3, 8, [0, 7], 1.0, 'srk'),
    (0, 6, 2, 9, 12, 3, 8, [0, 7], 1.0, 'srk'),
    (3, 18, 2, 11, 24, 5, 9, [0, 3.5, 8], 1.0, 'euler'),      # diffusion nets on [tau, y]: dense Jacobian of g
    (1, 14, 1, 9, 12, 3, 8, [0, 7], 0.5, 'euler'),
    (4, 19, 2, 9, 64, 69, 8, [0, 3, 7], 1.0, 'euler'),        # ... behind a wide control embedding (no MFMA variant)
    (2, 15, 3, 9, 40, 40, 8, [0, 7], 1.0, 'euler'),
    (1, 18, 2, 11, 24, 3, 9, [0, 3.5, 8], 1.0, 'srk'),        # torch_ists `neuralsde_1_18` under its default method
    (3, 14, 1, 9, 12, 3, 8, [0, 7], 0.5, 'srk'),
    (4, 15, 2, 9, 32, 5, 8, [0, 3, 7], 1.0, 'srk'),
    (6, 19, 3, 7, 64, 5, 8, [0, 7], 0.5, 'srk'),
    (0, 18, 2, 9, 128, 21, 8, [0, 7], 1.0, 'srk'),
    (3, 18, 2, 11, 24, 5, 9, [0, 3.5, 8], 1.0, 'milstein'),   # Milstein through a diffusion net: VJP of g per step, its
    (1, 14, 1, 9, 12, 3, 8, [0, 7], 0.5, 'milstein'),         # adjoint needs the net's tangent as well
    (4, 19, 2, 9, 64, 69, 8, [0, 3, 7], 1.0, 'milstein'),
    (2, 15, 3, 9, 40, 40, 8, [0, 7], 0.5, 'milstein'),
    (5, 19, 2, 9, 128, 3, 8, [0, 2.5, 7], 0.5, 'milstein'),
    (0, 15, 2, 7, 32, 5, 8, None, None, 'milstein'),
]


@pytest.mark.parametrize('ci', range(len(GEN_BWD_CASES)))
def test_generic_backward_matches_fp64_autograd(ci):
    io, no, NL, B, H, C, L, ts, dt, method = GEN_BWD_CASES[ci]
    _check_backward(900 + ci, io, no, NL, B, H, C, L, ts, dt, method, 'generic', strict=True)


@pytest.mark.parametrize('kernel', ['mfma4', 'mfma16'])
@pytest.mark.parametrize('ci', range(len(BWD_CASES)))
def test_backward_matches_fp64_autograd_through_the_unrolled_loop(ci, kernel):
    """dL/dy0 and dL/dtheta from the HIP adjoint + batched parameter pass vs float64 autograd through the unfused
    tensor-op loop (the reference's way of differentiating, common_sde.py:158-160) on identical increments."""
    io, no, NL, B, H, C, L, ts, dt = BWD_CASES[ci][:9]
    method = BWD_CASES[ci][9] if len(BWD_CASES[ci]) > 9 else 'euler'
    _check_backward(500 + ci, io, no, NL, B, H, C, L, ts, dt, method, kernel)


SRK_BWD_CASES = [
    # io, no, NL, B, H, C, L, ts, dt     (MFMA SRK forward + MFMA SRK adjoint + native parameter pass)
    (4, 17, 2, 11, 32, 5, 9, [0, 3.5, 8], 1.0),
    (6, 17, 3, 9, 64, 3, 9, [0, 8], 0.5),
    (2, 16, 1, 9, 16, 2, 12, None, 0.05),
    (1, 0, 2, 7, 32, 3, 8, [0, 7], 1.0),
    (3, 13, 2, 10, 64, 3, 8, [0, 2.5, 7], 1.0),
    (5, 12, 4, 6, 128, 3, 7, [0, 6], 1.0),
    (4, 17, 2, 21, 128, 21, 9, [0, 8], 1.0),
    (2, 3, 2, 9, 32, 3, 8, [0, 7], 1.0),             # closed-form table noise under SRK
    (6, 5, 1, 9, 64, 3, 8, [0, 3, 7], 0.5),
    (3, 11, 2, 9, 16, 3, 8, [0, 7], 1.0),
    (4, 1, 2, 9, 32, 5, 8, [0, 7], 1.0),
    (4, 9, 2, 9, 32, 5, 8, [0, 7], 1.0),             # y-only closed forms under SRK
    (1, 8, 2, 9, 16, 3, 8, [0, 3, 7], 0.5),
    (6, 10, 1, 9, 64, 3, 8, [0, 7], 1.0),
    (4, 17, 2, 9, 256, 14, 8, [0, 3, 7], 1.0),       # H = 256 (streamed weights): the torch_ists default method at the K5 width
    (6, 16, 1, 6, 256, 5, 7, [0, 6], 0.5),
    (4, 17, 2, 9, 64, 40, 8, [0, 3, 7], 1.0),        # wide control path (C > 32) under SRK
    (6, 13, 3, 7, 128, 69, 7, [0, 6], 1.0),
    (1, 18, 2, 9, 16, 3, 8, [0, 7], 0.5),            # SRK through a diffusion net: snsde_m4n_srk_reverse_kernel + weight-gradient
    (3, 15, 3, 8, 16, 4, 8, [0, 7], 1.0),            # jobs over the pass subsets / state planes of the four evaluations
    (1, 14, 1, 17, 32, 3, 9, [0, 2.5, 8], 0.5),
    (3, 18, 2, 33, 64, 5, 12, [0, 2.5, 11], 0.5),
    (5, 19, 2, 21, 64, 5, 9, [0, 8], 1.0),
    (4, 19, 2, 21, 128, 21, 10, [0, 9], 1.0),
    (2, 14, 2, 13, 32, 7, 9, [0, 3.5, 8], 0.5),
    (6, 15, 3, 9, 64, 40, 8, [0, 7], 1.0),
    (1, 18, 2, 37, 128, 5, 9, [0, 8], 1.0),
    (3, 18, 3, 11, 128, 5, 9, [0, 8], 0.5),
    (4, 18, 1, 11, 128, 69, 9, [0, 8], 1.0),
]


@pytest.mark.parametrize('kernel', ['mfma4', 'auto'])
@pytest.mark.parametrize('ci', range(len(SRK_BWD_CASES)))
def test_srk_backward_on_the_mfma_path(ci, kernel):
    io, no, NL, B, H, C, L, ts, dt = SRK_BWD_CASES[ci]
    if ts is not None:      # the fused MFMA adjoint, not the generic family
        grid = S.engine.step_grid(np.asarray(ts, np.float32), dt, np.arange(L, dtype=np.float32), torch.device(DEV))
        assert S.engine.backward_mode(S.engine.model_struct(C, H, H, NL, io, no), B, L, grid, 'srk', kernel) == 1
    _check_backward(4000 + ci, io, no, NL, B, H, C, L, ts, dt, 'srk', kernel, strict=True)


EULER_NET_CASES = [
    # io, no, NL, B, H, C, L     (Euler through a diffusion net on snsde_m4n_kernel.h: wide control paths behind the embedding - the
    (4, 18, 2, 19, 64, 69, 9),   #  sepsis channel count - and H = 128 on 4-row tiles)
    (6, 15, 3, 9, 128, 40, 8),
    (2, 14, 1, 13, 32, 33, 8),
    (1, 18, 2, 21, 128, 5, 9),
    (5, 19, 2, 11, 128, 3, 8),
]


@pytest.mark.parametrize('ci', range(len(EULER_NET_CASES)))
def test_euler_through_a_diffusion_net_on_the_net_kernels(ci):
    io, no, NL, B, H, C, L = EULER_NET_CASES[ci]
    pr = make_problem(800 + ci, io, no, NL, B, H, C, L)
    ts, dt = [0, 2.5, L - 1], 0.5
    model = S.engine.model_struct(C, H, H, NL, io, no)
    grid = S.engine.step_grid(np.asarray(ts, np.float32), dt, pr['times'], torch.device(DEV))
    assert S.engine.forward_path(model, B, L, grid.N) == 'mfma4' and S.engine.backward_mode(model, B, L, grid, 'euler') == 1
    dW = draw_dW(800 + ci, ts, dt, B, H)
    ys, _ = hip_solve(pr, ts, dt, dW=dW, kernel='auto')
    ref64, _ = oracle_solve(pr, ts, dt, dW, 'euler', np.float64)
    cpu32, _ = oracle_solve(pr, ts, dt, dW, 'euler', np.float32)
    assert_parity(ys, ref64, cpu32, what=f'euler net case {ci}')
    yg, _ = hip_solve(pr, ts, dt, dW=dW, kernel='generic')
    assert np.abs(ys - yg).max() <= 2e-4 * (np.abs(yg).max() + 1e-9)
    _check_backward(4700 + ci, io, no, NL, B, H, C, L, ts, dt, 'euler', 'auto', strict=True)


MIL_NET_BWD_CASES = [
    # io, no, NL, B, H, C, L, ts, dt     (Milstein through a diffusion net: snsde_m4n_mil_reverse_kernel - tangent + reverse pass
    (1, 18, 2, 9, 16, 3, 8, [0, 7], 0.5),            #  through the net per step - and the second-order weight-gradient jobs)
    (3, 15, 3, 8, 16, 4, 8, [0, 7], 1.0),
    (1, 14, 1, 17, 32, 3, 9, [0, 2.5, 8], 0.5),
    (3, 18, 2, 33, 64, 5, 12, [0, 2.5, 11], 0.5),
    (5, 19, 2, 21, 64, 5, 9, [0, 8], 1.0),
    (2, 14, 2, 13, 32, 7, 9, [0, 3.5, 8], 0.5),
    (6, 15, 3, 9, 64, 40, 8, [0, 7], 1.0),
    (6, 19, 4, 7, 32, 3, 8, [0, 7], 1.0),
    (4, 14, 1, 11, 128, 21, 9, [0, 8], 1.0),         # H = 128, one-layer net: matrices parked in LDS
    (4, 18, 2, 12, 64, 69, 9, [0, 4, 8], 1.0),       # the K4 channel count
]


@pytest.mark.parametrize('ci', range(len(MIL_NET_BWD_CASES)))
def test_milstein_backward_through_a_diffusion_net_on_the_mfma_path(ci):
    io, no, NL, B, H, C, L, ts, dt = MIL_NET_BWD_CASES[ci]
    grid = S.engine.step_grid(np.asarray(ts, np.float32), dt, np.arange(L, dtype=np.float32), torch.device(DEV))
    assert S.engine.backward_mode(S.engine.model_struct(C, H, H, NL, io, no), B, L, grid, 'milstein') == 1
    _check_backward(4500 + ci, io, no, NL, B, H, C, L, ts, dt, 'milstein', 'auto', strict=True)


@pytest.mark.parametrize('io', [0, 1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize('no', [0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 11, 12, 13, 16, 17])   # 7 (sqrt y): autograd's own gradient is NaN for y < 0
def test_backward_sweep_mfma_options(io, no):
    """Every (input_option, elementwise noise_option) pair of the MFMA path: adjoint kernel + native parameter pass vs
    float64 autograd, alternating depth, flavour and method."""
    k = io * 7 + no
    _check_backward(2000 + k, io, no, 1 + k % 3, 9, 16 if k % 2 else 32, 3, 7, [0, 2.5, 6], 1.0,
                    'milstein' if k % 3 == 0 else 'euler', 'mfma16' if k % 4 == 0 else 'mfma4')


@pytest.mark.parametrize('io', [0, 1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize('no', [0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 11, 12, 13, 16, 17])
def test_backward_sweep_generic_options(io, no):
    """Every input_option x elementwise noise_option on the generic adjoint kernels (Euler / Milstein / SRK in turn)."""
    k = io * 5 + no
    method = ('euler', 'milstein', 'srk')[k % 3]
    _check_backward(3000 + k, io, no, 1 + k % 2, 7, 12, 3, 6, [0, 5], 1.0 if k % 2 else 0.5, method, 'generic')


@pytest.mark.parametrize('io', [0, 1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize('no', [14, 15, 18, 19])
@pytest.mark.parametrize('method', ['euler', 'srk', 'milstein'])
def test_backward_sweep_generic_diffusion_nets(io, no, method):
    """Diffusion nets on [tau, y] (dense dg/dy) through the generic kernels, every input_option: Euler, SRK, Milstein."""
    k = io * 4 + no
    _check_backward(3500 + k, io, no, 1 + k % 2, 7, 12 if k % 2 else 20, 3, 6, [0, 5], 1.0 if k % 3 else 0.5, method, 'generic',
                    strict=True)


# Gradient tolerance per tensor: max|err| / max|ref| AND mean|err| / mean|ref| (round 5; it was 2e-3 on the maximum alone).  Measured over
# the ~5500 tensors of the cases below (profiles/r05_grad_margins_small.txt): all but 37 within 2.5e-5, the largest 2.3e-4.  The five
# cases above 5e-5 are listed with a 5e-4 bound: their fp64 gradients cancel to a small scalar (theta / sigma of 2009, 4006, 906) or the
# scheme amplifies round-off (4009: srk with raw = t y; 907: Milstein through y^3-like closed forms on the generic kernels).
GRAD_TOL_MAX, GRAD_TOL_MEAN = 1e-4, 1e-4
GRAD_TOL_LOOSE = {2009: 5e-4, 4006: 5e-4, 4009: 5e-4, 906: 5e-4, 907: 5e-4,
                  9354: 5e-4,      # (tests/test_gpu_w4.py, srk (1,18) NL = 1: theta's gradient cancels to -0.012, measured 1.3e-4 on either kernel family)
                  3035: 5e-4}      # (round 6, srk (6,5) on the generic kernels under the SRI2W1 rows: theta's gradient cancels to 0.0027, measured 2.5e-4)


def _check_backward(seed, io, no, NL, B, H, C, L, ts, dt, method, kernel, strict=False):
    times = np.linspace(0, 1, L).astype(np.float32) if ts is None else None
    pr = make_problem(seed, io, no, NL, B, H, C, L, times=times)
    ts = pr['times'] if ts is None else np.asarray(ts, np.float32)
    dt = dt or max(float(np.diff(pr['times']).min()), 1e-3)
    dW = draw_dW(seed, ts, dt, B, H)
    ci = seed
    wsum = np.random.default_rng(ci).standard_normal((len(ts), B, H)).astype(np.float32)
    dU = None
    if method == 'srk':      # space-time Levy integrals consistent in scale with the increments
        g0, g1 = O.step_grid(ts, dt)[:2]
        hh = (g1 - g0).astype(np.float32).reshape(-1, 1, 1)
        xi = np.random.default_rng(ci + 1).standard_normal(dW.shape).astype(np.float32)
        dU = (hh * (0.5 * dW + np.sqrt(hh / 12) * xi)).astype(np.float32)

    def build(dtype, device):
        m = S.Diffusion_model(C, H, H, NL, input_option=io, noise_option=no)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()})
        m = m.to(device=device, dtype=dtype)
        m.set_X(torch.from_numpy(pr['coeffs']).to(device=device, dtype=dtype), torch.from_numpy(pr['times']).to(device))
        y0 = torch.from_numpy(pr['y0']).to(device=device, dtype=dtype).requires_grad_(True)
        return m, y0

    m_ref, y0_ref = build(torch.float64, 'cpu')
    ys_ref = S.sdeint(m_ref, y0_ref, torch.from_numpy(ts),
                      bm=_ReplayBM(torch.from_numpy(dW).double(), None if dU is None else torch.from_numpy(dU).double()),
                      method=method, dt=dt, options={'backend': 'torch'})
    (ys_ref * torch.from_numpy(wsum).double()).sum().backward()

    m, y0 = build(torch.float32, DEV)
    ys = S.sdeint(m, y0, torch.from_numpy(ts).to(DEV),
                  bm=_ReplayBM(torch.from_numpy(dW).to(DEV), None if dU is None else torch.from_numpy(dU).to(DEV)),
                  method=method, dt=dt, options={'kernel': kernel, 'strict': strict})     # strict: no tensor-op fallback
    (ys * torch.from_numpy(wsum).to(DEV)).sum().backward()

    def close(got, ref, name):
        ref = ref.numpy()
        scale = np.abs(ref).max() + 1e-12
        e = np.abs(got.cpu().numpy().astype(np.float64) - ref)
        err = e.max() / scale
        mean_rel = e.mean() / (np.abs(ref).mean() + 1e-300)
        if os.environ.get('SNSDE_GRAD_MARGINS'):       # tools: measured margins of every case (profiles/r05_grad_margins_small.txt)
            with open(os.environ['SNSDE_GRAD_MARGINS'], 'a') as fh:
                fh.write(f'{seed} {io} {no} {H} {method} {kernel} {name} {err:.3e} {mean_rel:.3e}\n')
        assert err < GRAD_TOL_LOOSE.get(seed, GRAD_TOL_MAX), (name, err, scale)
        assert mean_rel < GRAD_TOL_LOOSE.get(seed, GRAD_TOL_MEAN), (name, mean_rel)

    fscale = float(ys_ref.detach().abs().max()) + 1e-12
    assert float((ys.detach().double().cpu() - ys_ref.detach()).abs().max()) / fscale < 2e-4, 'forward'
    close(y0.grad, y0_ref.grad, 'y0')
    ref_grads = dict(m_ref.named_parameters())
    for name, p in m.named_parameters():
        gref = ref_grads[name].grad
        if gref is None or float(gref.abs().max()) == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) < 1e-6, name
            continue
        close(p.grad, gref, name)


RECOMPUTE_CASES = [
    # io, no, NL, B, H, C, L, ts, dt, method, chunk, per-row outputs
    (4, 17, 2, 37, 128, 21, 13, [0, 12], 1.0, 'euler', 5, False),            # K2 model: one output at the end
    (4, 17, 2, 21, 256, 14, 12, None, None, 'milstein', 4, False),           # K5 model: Milstein, every knot an output
    (6, 17, 3, 9, 64, 5, 9, [0, 2.5, 4, 8], 0.5, 'milstein', 3, False),      # off-grid output inside a chunk, 16 half steps
    (2, 16, 1, 50, 32, 2, 11, [0, 10], 1.0, 'euler', 1, False),              # one step per chunk
    (4, 13, 2, 19, 64, 3, 10, None, None, 'euler', 4, True),                 # per-row output selection (classification wrapper)
    (3, 18, 2, 13, 64, 5, 9, [0, 3.5, 8], 1.0, 'euler', 3, False),           # diffusion net (BASELINE config 4's model)
    (4, 17, 2, 11, 128, 21, 9, [0, 8], 1.0, 'euler', 100, False),            # chunk longer than the solve
]


def test_recompute_mode_peaks_below_the_saved_activation_mode():
    """options={'recompute': K} is a CAPACITY mode: at the K2 size a step with 50- and 25-step chunks must peak below the saved-activation
    step (round 3 it peaked ABOVE at 50: two chunks' buffers were alive at once and every chunk copied its increments)."""
    io, no, NL, B, H, C, L = 4, 17, 2, 1024, 128, 21, 101
    pr = make_problem(7, io, no, NL, B, H, C, L, nan_frac=0.2)
    m = S.Diffusion_model(C, H, H, NL, input_option=io, noise_option=no)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()})
    m = m.to(DEV)
    times = torch.from_numpy(pr['times']).to(DEV)
    m.set_X(torch.from_numpy(pr['coeffs']).to(DEV), times)
    y0 = torch.from_numpy(pr['y0']).to(DEV)
    peak = {}
    for chunk in (0, 50, 25):
        def step():
            for p in m.parameters():
                p.grad = None
            yy = y0.clone().requires_grad_(True)
            S.sdeint(m, yy, times[[0, -1]], method='euler', dt=1.0, options={'seed': 1, 'recompute': chunk})[-1].square().mean().backward()
        step(); step()
        torch.cuda.synchronize()
        base = torch.cuda.memory_allocated()
        torch.cuda.reset_peak_memory_stats()
        step()
        torch.cuda.synchronize()
        peak[chunk] = torch.cuda.max_memory_allocated() - base
    assert peak[50] < 0.9 * peak[0], peak
    assert peak[25] < peak[50], peak


SIGN_CASES = [
    # io, no, NL, B, H, C, L, method, kernel: every Euler / Milstein forward kernel that feeds the MFMA adjoint
    (4, 17, 2, 37, 128, 21, 9, 'euler', 'auto'),        # lean 4-row tiles (K2 model)
    (6, 17, 4, 11, 64, 5, 9, 'milstein', 'auto'),       # lean, three hidden layers: four sign bits
    (4, 17, 2, 37, 128, 21, 9, 'euler', 'mfma16'),      # general kernel, 16-row tiles
    (4, 17, 2, 21, 256, 14, 8, 'milstein', 'auto'),     # streamed H = 256 (K5 model)
    (4, 17, 3, 21, 256, 14, 8, 'euler', 'mfma16'),      # H = 256 on 16-row tiles
    (3, 18, 2, 21, 64, 5, 9, 'euler', 'auto'),          # diffusion net, general 4-row tiles
    (3, 19, 2, 9, 128, 3, 8, 'euler', 'auto'),          # diffusion net at H = 128: the net kernels' Euler variant
    (4, 18, 2, 13, 64, 69, 9, 'euler', 'auto'),         # nets behind a wide control path
    (0, 17, 2, 9, 64, 5, 8, 'euler', 'auto'),           # y-free drift
    (3, 18, 2, 21, 64, 5, 9, 'srk', 'auto'),            # SRK through a two-layer net: every pass's z, + the net's hidden signs
    (1, 18, 2, 13, 128, 3, 8, 'srk', 'auto'),
    (1, 14, 3, 9, 32, 3, 8, 'srk', 'auto'),             # one-layer net: drift signs only
    (3, 18, 2, 21, 64, 5, 9, 'milstein', 'auto'),       # Milstein through a two-layer net: drift signs + the net's hidden sign
    (4, 17, 2, 37, 128, 21, 9, 'srk', 'auto'),          # SRK, elementwise diffusion (general kernel's SRK variant, 4-row tiles)
    (6, 17, 3, 21, 64, 5, 9, 'srk', 'mfma16'),          # ... on 16-row tiles (the SRK adjoint reads the same saves)
]


@pytest.mark.parametrize('ci', range(len(SIGN_CASES)))
def test_saved_drift_carries_the_relu_signs_of_its_step(ci):
    """Training-mode forward (include/snsde.h, act_save): the saved pre-tanh drift z (slot NL) carries [slot k > 0] of the same element
    in mantissa bit k, k < NL, and is otherwise the exact z (the adjoint takes its relu masks from these bits instead of re-reading the
    activation planes).  Checked on every forward kernel family against the activation planes of the same call, and z itself against
    an inference-mode-identical solve whose bits are cleared: the forward's states do not depend on the packing."""
    io, no, NL, B, H, C, L, method, kernel = SIGN_CASES[ci]
    pr = make_problem(300 + ci, io, no, NL, B, H, C, L, nan_frac=0.2)
    model = S.engine.model_struct(C, H, H, NL, io, no)
    flat = flat_params(pr['params'], io, no, NL, C, H)
    grid = S.engine.step_grid(np.asarray([0.0, float(L - 1)], np.float32), 1.0, pr['times'], torch.device(DEV))
    args = (model, flat, torch.from_numpy(pr['coeffs']).to(DEV), grid, torch.from_numpy(pr['y0']).to(DEV))
    call = S.engine.SolveCall(*args, method=method, seed=5, kernel=kernel, save_traj=True, save_dW=True, save_act=True)
    if S.engine.backward_supported(call) != 1:
        pytest.skip('not an MFMA-adjoint configuration')
    ys = call.launch().clone()
    plain = S.engine.SolveCall(*args, method=method, seed=5, kernel=kernel)
    assert torch.equal(ys, plain.launch()), 'training-mode states differ from the inference solve'
    act = call.act_save.cpu().numpy()                       # (passes, slots, B, H): passes = N, 3 N under SRK
    zbits = act[:, NL].view(np.uint32)
    for k in range(NL):
        want = act[:, k] > 0
        got = ((zbits >> k) & 1).astype(bool)
        assert np.array_equal(got, want), f'sign bit {k}'
        assert 0.02 < want.mean() < 0.98                    # (both signs occur: the check is not vacuous)
    nbits = NL
    if method == 'milstein' and no in (18, 19):
        nbits = NL + 1                                      # bit NL: the net's hidden layer (slot NL + 1)
        assert np.array_equal(((zbits >> NL) & 1).astype(bool), act[:, NL + 1] > 0)
    if method == 'srk' and no in (18, 19):
        # bit NL: the hidden layer of the net evaluation beside the pass (slot NL + 1); bit NL + 1 on the passes 3 n + 2: the hidden
        # layer of the step's fourth evaluation (slot NL + 3)
        nbits = NL + 2
        assert np.array_equal(((zbits >> NL) & 1).astype(bool), act[:, NL + 1] > 0)
        assert np.array_equal(((zbits[2::3] >> (NL + 1)) & 1).astype(bool), act[2::3, NL + 3] > 0)
    zc = (zbits & ~np.uint32((1 << nbits) - 1)).view(np.float32)
    z = act[:, NL]
    assert np.all(np.abs(z - zc) <= np.abs(zc) * 2.0 ** (-23 + nbits) + 1e-44)


@pytest.mark.parametrize('ci', range(len(RECOMPUTE_CASES)))
def test_recompute_mode_backward_equals_saved_activation_backward(ci):
    """options={'recompute': K}: states and increments kept, activations re-created chunk by chunk in backward
    (engine.backward_recompute).  dL/dy0 goes through the same kernels on the same numbers: bit-equal; the parameter
    gradients are the same sums taken chunk by chunk: equal to rounding of the accumulation order."""
    io, no, NL, B, H, C, L, ts, dt, method, chunk, per_row = RECOMPUTE_CASES[ci]
    times = np.linspace(0, 1, L).astype(np.float32) if ts is None else None
    pr = make_problem(1700 + ci, io, no, NL, B, H, C, L, times=times)
    ts = pr['times'] if ts is None else np.asarray(ts, np.float32)
    dt = dt or max(float(np.diff(pr['times']).min()), 1e-3)
    rng = np.random.default_rng(ci)
    row_out = torch.from_numpy(rng.integers(0, len(ts), B)).to(DEV) if per_row else None
    wsum = torch.from_numpy(rng.standard_normal((B, H) if per_row else (len(ts), B, H)).astype(np.float32)).to(DEV)
    out = {}
    for mode in ('saved', 'recompute'):
        m = S.Diffusion_model(C, H, H, NL, input_option=io, noise_option=no)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()})
        m = m.to(DEV)
        m.set_X(torch.from_numpy(pr['coeffs']).to(DEV), torch.from_numpy(pr['times']).to(DEV))
        y0 = torch.from_numpy(pr['y0']).to(DEV).requires_grad_(True)
        opts = {'seed': 11}
        if mode == 'recompute':
            opts['recompute'] = chunk
        if per_row:
            opts['row_out'] = row_out
        torch.cuda.reset_peak_memory_stats()
        ys = S.sdeint(m, y0, torch.from_numpy(ts).to(DEV), method=method, dt=dt, options=opts)
        (ys * wsum).sum().backward()
        out[mode] = (ys.detach(), y0.grad.detach(), {k: p.grad.detach() for k, p in m.named_parameters() if p.grad is not None})
    assert torch.equal(out['saved'][0], out['recompute'][0])
    assert torch.equal(out['saved'][1], out['recompute'][1]), 'dL/dy0 differs between the two modes'
    assert set(out['saved'][2]) == set(out['recompute'][2])
    for k, ref in out['saved'][2].items():
        got = out['recompute'][2][k]
        scale = float(ref.abs().max()) + 1e-12
        assert float((got - ref).abs().max()) <= 2e-5 * scale + 1e-7, (k, float((got - ref).abs().max()), scale)


@pytest.mark.parametrize('case', [(4, 17, 2, 300, 128, 21, 13, 'euler'), (6, 17, 3, 70, 64, 7, 9, 'milstein'),
                                  (3, 13, 2, 45, 32, 3, 8, 'milstein'), (1, 0, 1, 33, 16, 3, 8, 'euler'),
                                  (5, 12, 4, 40, 128, 3, 8, 'euler'), (2, 16, 2, 130, 256, 14, 9, 'milstein')])
def test_native_parameter_pass_matches_library_gemm_pass(case):
    """snsde_param_gradients (split-R MFMA GEMMs + reductions + first-layer algebra) against the same sums formed with
    library GEMMs / elementwise torch ops from the identical saved tensors."""
    io, no, NL, B, H, C, L, method = case
    pr = make_problem(77, io, no, NL, B, H, C, L)
    ts = np.asarray([0, (L - 1) / 2, L - 1], np.float32)
    dW = draw_dW(77, ts, 1.0, B, H)
    wsum = torch.from_numpy(np.random.default_rng(7).standard_normal((len(ts), B, H)).astype(np.float32)).to(DEV)
    out = {}
    for mode in ('hip', 'torch'):
        m = S.Diffusion_model(C, H, H, NL, input_option=io, noise_option=no)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()})
        m = m.to(DEV)
        m.set_X(torch.from_numpy(pr['coeffs']).to(DEV), torch.from_numpy(pr['times']).to(DEV))
        y0 = torch.from_numpy(pr['y0']).to(DEV).requires_grad_(True)
        ys = S.sdeint(m, y0, torch.from_numpy(ts).to(DEV), bm=_ReplayBM(torch.from_numpy(dW).to(DEV)), method=method, dt=1.0,
                      options={'param_pass': mode})
        (ys * wsum).sum().backward()
        out[mode] = {k: (None if p.grad is None else p.grad.detach().cpu().numpy()) for k, p in m.named_parameters()}
    for k, ref in out['torch'].items():
        got = out['hip'][k]
        if ref is None or np.abs(ref).max() == 0.0:
            assert got is None or np.abs(got).max() < 1e-6, k
            continue
        scale = np.abs(ref).max()
        assert np.abs(got - ref).max() / scale < 2e-4, (k, np.abs(got - ref).max() / scale)


@pytest.mark.parametrize('case', [(4, 17, 2, 300, 128, 21, 13, 'euler'), (6, 17, 3, 70, 64, 7, 9, 'milstein'), (1, 0, 1, 33, 16, 3, 8, 'euler'),
                                  (2, 16, 2, 130, 256, 14, 9, 'milstein'), (4, 13, 2, 37, 64, 5, 9, 'srk'), (3, 18, 2, 33, 64, 5, 12, 'srk'),
                                  (3, 18, 2, 21, 32, 69, 9, 'milstein'), (0, 16, 2, 19, 32, 5, 8, 'euler')])
def test_fused_backward_call_equals_the_two_separate_calls(case):
    """snsde_backward_with_gradients (adjoint + parameter pass in one C call, the forward-only parts of the parameter pass on the side
    stream beside the adjoint kernel) against snsde_solve_backward followed by snsde_param_gradients: bit-identical gradients."""
    io, no, NL, B, H, C, L, method = case
    pr = make_problem(78, io, no, NL, B, H, C, L)
    ts = np.asarray([0, (L - 1) / 2, L - 1], np.float32)
    wsum = torch.from_numpy(np.random.default_rng(8).standard_normal((len(ts), B, H)).astype(np.float32)).to(DEV)
    out = {}
    for mode in ('hip', 'split', 'hip'):
        m = S.Diffusion_model(C, H, H, NL, input_option=io, noise_option=no)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()})
        m = m.to(DEV)
        m.set_X(torch.from_numpy(pr['coeffs']).to(DEV), torch.from_numpy(pr['times']).to(DEV))
        y0 = torch.from_numpy(pr['y0']).to(DEV).requires_grad_(True)
        ys = S.sdeint(m, y0, torch.from_numpy(ts).to(DEV), method=method, dt=1.0, options={'param_pass': mode, 'seed': 5, 'strict': True})
        (ys * wsum).sum().backward()
        got = {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters() if p.grad is not None}
        got['y0'] = y0.grad.detach().cpu().numpy()
        if mode in out:
            for k in got:
                np.testing.assert_array_equal(got[k], out[mode][k], err_msg=f'{k}: fused call not reproducible')
        out[mode] = got
    assert out['hip'].keys() == out['split'].keys()
    for k in out['hip']:
        np.testing.assert_array_equal(out['hip'][k], out['split'][k], err_msg=k)


def test_backward_without_a_fused_adjoint_falls_back_to_the_tensor_loop_or_raises_when_strict():
    # Milstein with sqrt(y) (noise_option 7) is the one family without kernels: no finite dg/dy at the clipped values
    pr = make_problem(9, 1, 7, 2, 8, 64, 3, 5)
    m = S.Diffusion_model(3, 64, 64, 2, input_option=1, noise_option=7).to(DEV)
    m.set_X(torch.from_numpy(pr['coeffs']).to(DEV), torch.from_numpy(pr['times']).to(DEV))
    y0 = torch.from_numpy(pr['y0']).to(DEV).requires_grad_(True)
    ts = torch.tensor([0., 4.], device=DEV)
    with pytest.raises(NotImplementedError):
        S.sdeint(m, y0, ts, method='milstein', dt=1.0, options={'strict': True})
    # (round 2: every diffusion net has a fused backward - Euler, SRK and Milstein, wide control path included)
    pr2 = make_problem(9, 4, 18, 2, 8, 64, 40, 5)
    m2 = S.Diffusion_model(40, 64, 64, 2, input_option=4, noise_option=18).to(DEV)
    m2.set_X(torch.from_numpy(pr2['coeffs']).to(DEV), torch.from_numpy(pr2['times']).to(DEV))
    for method in ('euler', 'srk', 'milstein'):
        S.sdeint(m2, y0, ts, method=method, dt=1.0, options={'strict': True, 'seed': 1})[-1].sum().backward()
    # default: the reference's training loop keeps running — the call differentiates through the unfused tensor-op loop
    S.torchsde._UNFUSED_WARNED.clear()
    y0.grad = None
    with pytest.warns(UserWarning, match='no fused backward'):
        ys = S.sdeint(m, y0, ts, method='milstein', dt=1.0, options={'seed': 3})
    ys[-1].sum().backward()      # (autograd's own derivative of sqrt is not finite once a state goes negative: no value check)
    assert ys.shape == (2, 8, 64) and ys.grad_fn is not None and y0.grad is not None


@pytest.mark.parametrize('kernel,method,io,no', [('mfma4', 'euler', 4, 17), ('mfma16', 'milstein', 6, 17), ('generic', 'euler', 2, 7),
                                                  ('generic', 'srk', 4, 17), ('mfma4', 'euler', 3, 18),
                                                  ('mfma4', 'srk', 6, 17), ('auto', 'srk', 2, 16)])
def test_per_row_output_selection_equals_gather(kernel, method, io, no):
    """options['row_out'] (per-row output slot, fused into the solve) == solving for every output and gathering
    (the per-row selection of NeuralSDE.forward, neuralsde.py:115-116), forward and backward, bit for bit."""
    B, H, C, L = 37, 64, 5, 12
    torch.manual_seed(99)
    pr = make_problem(41, io, no, 2, B, H, C, L)
    ts = torch.tensor([0., 2.5, 4., 7., 11.], device=DEV)       # includes an interpolated output
    slot = torch.randint(0, 5, (B,), device=DEV)
    wsum = torch.randn(B, H, device=DEV)
    res = {}
    for mode in ('fused', 'gather'):
        m = S.Diffusion_model(C, H, H, 2, input_option=io, noise_option=no)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()})
        m = m.to(DEV)
        m.set_X(torch.from_numpy(pr['coeffs']).to(DEV), torch.from_numpy(pr['times']).to(DEV))
        grad = True
        y0 = torch.from_numpy(pr['y0']).to(DEV).requires_grad_(grad)
        opts = {'seed': 9, 'kernel': kernel}
        with torch.set_grad_enabled(grad):
            if mode == 'fused':
                z = S.sdeint(m, y0, ts, method=method, dt=1.0, options=dict(opts, row_out=slot))
            else:
                zt = S.sdeint(m, y0, ts, method=method, dt=1.0, options=opts)
                z = zt.gather(0, slot.reshape(1, -1, 1).expand(1, B, H)).squeeze(0)
            assert z.shape == (B, H)
            if grad:
                (z * wsum).sum().backward()
        res[mode] = (z.detach(), y0.grad, [p.grad for p in m.parameters()] if grad else [])
    assert torch.equal(res['fused'][0], res['gather'][0])
    if res['fused'][1] is not None:
        assert torch.equal(res['fused'][1], res['gather'][1])
        for a, b in zip(res['fused'][2], res['gather'][2]):
            assert (a is None and b is None) or torch.equal(a, b)


def test_neuralsde_wrapper_all_knot_outputs_equal_reference_output_time_selection():
    """NeuralSDE.forward on CUDA emits every knot and gathers each row's state; the reference solves on
    ts = [t0, times[unique(final_index)], t_end] (neuralsde.py:91-116).  Same grid, same interpolation: identical bits."""
    B, H, C, L = 96, 64, 5, 17
    pr = make_problem(31, 4, 17, 2, B, H, C, L)
    torch.manual_seed(3)
    model, field = S.make_sde_model('neurallnsde', C, 2, H, H, 2, initial=True)
    model = model.to(DEV).eval()
    times = torch.from_numpy(pr['times']).to(DEV)
    coeffs = torch.from_numpy(pr['coeffs']).to(DEV)
    fi = torch.randint(0, L, (B,), device=DEV)
    with torch.no_grad():
        got = model(times, [coeffs], fi, options={'seed': 5})
        field.set_X(coeffs, times)
        ts, slot = model.output_times(times, fi)
        # (initial state and readout through the same fused launches the wrapper uses in inference: the comparison is about
        # the output-time selection; fused vs tensor-op initial state / head: tests/test_gpu_wrappers.py)
        z_t = S.sdeint(field, torch.empty(B, H, device=DEV), ts, method='euler', dt=1.0,
                       options={'seed': 5, 'z0_linear': model.initial_network})
        want = model._readout(z_t.gather(0, slot.reshape(1, -1, 1).expand(1, B, H)).squeeze(0))
        z0 = model.initial_network(field.X.evaluate(times[0]))
        plain = model.linear(S.sdeint(field, z0, ts, method='euler', dt=1.0, options={'seed': 5})
                             .gather(0, slot.reshape(1, -1, 1).expand(1, B, H)).squeeze(0))
    assert torch.equal(got, want)
    assert float((got - plain).abs().max()) < 2e-4 * (float(plain.abs().max()) + 1e-6)


def test_training_step_recorded_into_a_graph_draws_fresh_noise_and_matches_eager():
    """A whole training step (NeuralSDE forward, fused solve + adjoint + parameter pass, Adam) recorded into one
    CUDA/HIP graph: every replay must integrate against fresh increments (device-resident Philox key) and leave the
    parameters exactly where the same steps run eagerly with those keys leave them."""
    from stable_neural_sdes_amd import torchsde as T
    B, H, C, L = 64, 32, 5, 9
    pr = make_problem(23, 4, 17, 2, B, H, C, L)
    times = torch.from_numpy(pr['times']).to(DEV)
    coeffs = torch.from_numpy(pr['coeffs']).to(DEV)
    fi = torch.randint(1, L, (B,), device=DEV)
    target = (torch.rand(B, device=DEV) > 0.5).float()

    def make():
        torch.manual_seed(1)
        model, _ = S.make_sde_model('neurallnsde', C, 1, H, H, 2, initial=True)
        model = model.to(DEV).train()
        model.linear[3].p = 0.0          # dropout off: its generator state is not part of this comparison
        return model, torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True)

    def step(model, opt, seed=None):
        opts = {} if seed is None else {'seed': seed}
        pred = model(times, [coeffs], fi, options=opts).squeeze(-1)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(pred, target)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    state = T.prepare_graph_capture(DEV)
    mg, og = make()
    me, oe = make()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for k in range(3):
            step(mg, og, seed=100 + k)
    torch.cuda.current_stream().wait_stream(side)
    for k in range(3):
        step(me, oe, seed=100 + k)
    torch.cuda.synchronize()
    base = int(state.item())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static_loss = step(mg, og)
    losses = []
    for _ in range(3):
        g.replay()
        losses.append(float(static_loss))
    assert int(state.item()) == base + 3
    assert len(set(losses)) == 3 and all(np.isfinite(losses))
    eager = [float(step(me, oe, seed=base + 1 + k)) for k in range(3)]
    assert losses == eager
    for (n1, p1), (n2, p2) in zip(mg.named_parameters(), me.named_parameters()):
        assert torch.equal(p1, p2), n1


def test_ists_neuralsde_trains_with_its_default_srk_method_on_the_fused_path():
    """torch_ists flavour (nsde_model.py:63-84): forward(coeffs, times) with the default method 'srk'; loss.backward()
    runs the fused SRK forward + SRK adjoint kernel + batched parameter pass (no tensor-op loop)."""
    B, H, C, L = 24, 32, 4, 9
    pr = make_problem(29, 4, 17, 2, B, H, C, L)
    torch.manual_seed(2)
    field = S.Diffusion_model(C, H, H, 2, input_option=4, noise_option=17)
    model = S.IstsNeuralSDE(field, C, H, 3).to(DEV).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    times = torch.from_numpy(pr['times']).to(DEV)
    coeffs = torch.from_numpy(pr['coeffs']).to(DEV)
    target = torch.randn(B, L, 3, device=DEV)
    before = field.linear_out.weight.detach().clone()
    for _ in range(2):
        out, z = model(coeffs, times, options={'seed': 3})
        assert out.shape == (B, L, 3) and z.shape == (B, L, H)
        loss = (out - target).square().mean()
        opt.zero_grad(); loss.backward(); opt.step()
    assert np.isfinite(float(loss.detach()))
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in field.parameters() if p.requires_grad)
    assert not torch.equal(before, field.linear_out.weight.detach())


def test_neuralsde_training_step_on_cuda():
    """One optimizer step of the reference's training recipe (Adam, BCE-with-logits) through the fused path."""
    pr = make_problem(23, 4, 17, 2, 64, 32, 5, 9)
    torch.manual_seed(1)
    model, field = S.make_sde_model('neurallnsde', 5, 1, 32, 32, 2, initial=True)
    model = model.to(DEV).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    times = torch.from_numpy(pr['times']).to(DEV)
    coeffs = torch.from_numpy(pr['coeffs']).to(DEV)
    fi = torch.randint(0, 9, (64,), device=DEV)
    target = (torch.rand(64, device=DEV) > 0.5).float()
    before = {k: v.detach().clone() for k, v in model.state_dict().items() if v.dtype.is_floating_point}
    losses = []
    for _ in range(3):
        pred = model(times, [coeffs], fi, options={'seed': 11}).squeeze(-1)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(pred, target)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(np.isfinite(losses))
    moved = [k for k, v in model.state_dict().items() if k in before and not torch.equal(v, before[k])]
    assert any(k.startswith('func.linear_in') for k in moved) and any(k.startswith('func.noise_t') for k in moved)
    assert any(k.startswith('initial_network') for k in moved)


# ---- SRK (SRID2) ---------------------------------------------------------------------------------------
SRK_CASES = [
    # io, no, NL, B, H, C, L, ts, dt
    (6, 17, 2, 19, 32, 5, 9, [0, 2.5, 8], 0.5),      # torch_ists / tutorial GSDE-SRK flavour
    (4, 17, 2, 37, 128, 21, 13, [0, 12], 1.0),
    (2, 16, 1, 9, 16, 2, 12, None, None),            # ts = times = linspace(0,1,12): interpolated outputs
    (1, 18, 2, 8, 24, 3, 8, [0, 7], 0.5),
    (3, 15, 3, 8, 16, 4, 8, [0, 7], 1.0),
    (0, 5, 2, 8, 12, 3, 8, [0, 7], 1.0),
    (5, 8, 2, 8, 10, 3, 8, [0, 3.5, 7], 0.25),
    (1, 0, 2, 5, 8, 3, 8, [0, 7], 1.0),
    (1, 12, 1, 9, 64, 3, 8, [0, 2.5, 7], 1.0),       # MFMA SRK variant: every drift family, H = 16 .. 128, NL 1 .. 4
    (3, 13, 3, 21, 32, 3, 9, [0, 8], 0.5),
    (5, 17, 2, 13, 16, 3, 8, [0, 7], 1.0),
    (4, 16, 4, 9, 64, 21, 9, [0, 4, 8], 1.0),
    (2, 0, 2, 7, 128, 32, 8, [0, 7], 1.0),
    (4, 6, 2, 9, 32, 5, 8, [0, 7], 1.0),
    (1, 2, 2, 9, 64, 3, 8, [0, 3, 7], 0.5),
    (2, 9, 2, 9, 32, 3, 8, [0, 7], 1.0),
    (5, 7, 2, 9, 16, 3, 8, [0, 7], 0.5),
    (4, 17, 2, 9, 256, 14, 9, [0, 3.5, 8], 1.0),     # H = 256 on the MFMA SRK variant (weights streamed)
    (1, 13, 1, 5, 256, 3, 8, [0, 7], 0.5),
    (0, 17, 2, 9, 64, 5, 8, [0, 7], 1.0),            # y-free drift and wide control paths on the MFMA SRK variant
    (0, 4, 1, 6, 128, 21, 8, [0, 3, 7], 0.5),
    (4, 17, 2, 9, 64, 40, 9, [0, 8], 1.0),
    (6, 16, 3, 7, 128, 69, 8, [0, 7], 1.0),
    (2, 12, 1, 9, 32, 33, 8, [0, 2.5, 7], 0.5),
    (1, 18, 2, 9, 16, 3, 8, [0, 7], 0.5),            # diffusion nets on the MFMA net kernels (snsde_m4n_kernel.h): H = 16 .. 128,
    (1, 14, 1, 17, 32, 3, 9, [0, 2.5, 8], 0.5),      # one- and two-layer nets, raw = net and net * y, every drift family
    (3, 18, 2, 33, 64, 5, 12, [0, 2.5, 11], 0.5),
    (5, 19, 2, 21, 64, 5, 9, [0, 8], 1.0),
    (4, 19, 2, 21, 128, 21, 10, [0, 9], 1.0),
    (2, 14, 2, 13, 32, 7, 9, [0, 3.5, 8], 0.5),
    (6, 15, 3, 9, 64, 40, 8, [0, 7], 1.0),
    (1, 18, 2, 37, 128, 5, 9, [0, 8], 1.0),          # H = 128: net matrices parked in the waves' LDS slices
    (3, 18, 3, 11, 128, 5, 9, [0, 8], 0.5),
    (4, 18, 1, 11, 128, 69, 9, [0, 8], 1.0),         # wide control path (C = 69) with a net
    (6, 19, 4, 7, 32, 3, 8, [0, 7], 1.0),
]
SRK_NET_ROWS = [i for i, c in enumerate(SRK_CASES) if c[1] in (14, 15, 18, 19) and c[4] in (16, 32, 64, 128)]


def _draw_dU(seed, dW, ts, dt):
    t0, t1, *_ = O.step_grid(np.asarray(ts, np.float32), dt)
    h = (t1 - t0).astype(np.float32)[:, None, None]
    xi = np.random.default_rng(seed + 7).standard_normal(dW.shape).astype(np.float32)
    return (h * (0.5 * dW + np.sqrt(h / 12) * xi)).astype(np.float32)


@pytest.mark.parametrize('kernel', ['auto', 'generic'])
@pytest.mark.parametrize('ci', range(len(SRK_CASES)))
def test_srk_trajectory_vs_oracle(ci, kernel):
    io, no, NL, B, H, C, L, ts, dt = SRK_CASES[ci]
    times = np.linspace(0, 1, L).astype(np.float32) if ts is None else None
    pr = make_problem(700 + ci, io, no, NL, B, H, C, L, times=times)
    if ts is None:
        ts = pr['times']
        dt = max(float(np.diff(pr['times']).min()), 1e-3) / 2      # torch_ists tutorial: dt = min gap / 2
    dW = draw_dW(700 + ci, ts, dt, B, H)
    dU = _draw_dU(700 + ci, dW, ts, dt)
    ys, call = hip_solve(pr, ts, dt, dW=dW, dU=dU, method='srk', save_traj=True, kernel=kernel)
    ref64, traj64 = O.solve_diffusion_model(pr['params'], io, no, pr['coeffs'], pr['times'], pr['y0'],
                                            np.asarray(ts, np.float32), dt, dW, method='srk', dtype=np.float64, dU=dU)
    cpu32, _ = O.solve_diffusion_model(pr['params'], io, no, pr['coeffs'], pr['times'], pr['y0'],
                                       np.asarray(ts, np.float32), dt, dW, method='srk', dtype=np.float32, dU=dU)
    assert_parity(ys, ref64, cpu32, what=f'srk case {ci}')
    assert_parity(call.traj.cpu().numpy(), traj64, what=f'srk traj {ci}')


@pytest.mark.parametrize('ci', SRK_NET_ROWS)
def test_srk_diffusion_nets_take_the_mfma_net_kernels(ci):
    """The diffusion-net rows of SRK_CASES at instantiated hidden sizes run on the MFMA path (not the generic VALU family) under
    kernel='auto', and the explicit 'mfma4' selector gives the same bits."""
    io, no, NL, B, H, C, L, ts, dt = SRK_CASES[ci]
    pr = make_problem(700 + ci, io, no, NL, B, H, C, L)
    model = S.engine.model_struct(C, H, H, NL, io, no)
    grid = S.engine.step_grid(np.asarray(ts, np.float32), dt, pr['times'], torch.device(DEV))
    path = S.engine.forward_path(model, B, L, grid.N, method='srk')
    assert path in ('mfma-srk', 'w4')       # (round 5: latent-only drifts at H = 64 take the wave-pair kernel, tests/test_gpu_w4.py)
    assert S.engine.backward_mode(model, B, L, grid, 'srk') == 1
    dW = draw_dW(700 + ci, ts, dt, B, H)
    dU = _draw_dU(700 + ci, dW, ts, dt)
    y_auto, _ = hip_solve(pr, ts, dt, dW=dW, dU=dU, method='srk', kernel='auto')
    y_m4, _ = hip_solve(pr, ts, dt, dW=dW, dU=dU, method='srk', kernel='mfma4')
    if path == 'mfma-srk':
        assert np.array_equal(y_auto, y_m4)
    else:
        assert np.abs(y_auto - y_m4).max() <= 5e-5 * (np.abs(y_m4).max() + 1.0)


SRK_M16_ROWS = [i for i, c in enumerate(SRK_CASES) if c[4] in (64, 128) and c[5] <= 32 and c[1] not in (14, 15, 18, 19) and c[7] is not None]


@pytest.mark.parametrize('ci', SRK_M16_ROWS)
def test_srk_on_16_row_tiles_vs_oracle_and_backward(ci):
    """SRID2 on the 16-row-tile flavour (large batches; H = 64 / 128, elementwise diffusions): forward against the fp64 oracle,
    identical Philox paths as the 4-row tiles, and a training step whose adjoint runs on 4-row tiles over the same saves."""
    io, no, NL, B, H, C, L, ts, dt = SRK_CASES[ci]
    pr = make_problem(700 + ci, io, no, NL, B, H, C, L)
    dW = draw_dW(700 + ci, ts, dt, B, H)
    dU = _draw_dU(700 + ci, dW, ts, dt)
    ys, _ = hip_solve(pr, ts, dt, dW=dW, dU=dU, method='srk', kernel='mfma16')
    ref64, _ = O.solve_diffusion_model(pr['params'], io, no, pr['coeffs'], pr['times'], pr['y0'], np.asarray(ts, np.float32), dt, dW,
                                       method='srk', dtype=np.float64, dU=dU)
    cpu32, _ = O.solve_diffusion_model(pr['params'], io, no, pr['coeffs'], pr['times'], pr['y0'], np.asarray(ts, np.float32), dt, dW,
                                       method='srk', dtype=np.float32, dU=dU)
    assert_parity(ys, ref64, cpu32, what=f'srk m16 case {ci}')
    y16, _ = hip_solve(pr, ts, dt, method='srk', kernel='mfma16', seed=5)
    y4, _ = hip_solve(pr, ts, dt, method='srk', kernel='mfma4', seed=5)
    assert np.abs(y16 - y4).max() <= 2e-4 * (np.abs(y4).max() + 1e-9)
    if io != 0:
        _check_backward(4800 + ci, io, no, NL, B, H, C, L, ts, dt, 'srk', 'mfma16', strict=True)


def test_srk_philox_levy_area_matches_specification_and_shards():
    pr = make_problem(41, 6, 17, 2, 24, 32, 5, 9)
    ts, dt = [0, 8], 0.5
    ys, call = hip_solve(pr, ts, dt, method='srk', seed=99, row_offset=64, save_dW=True)
    t0, t1, *_ = O.step_grid(np.asarray(ts, np.float32), dt)
    dW = call.dW_out.cpu().numpy()
    dU = call.dU_out.cpu().numpy()
    exp_dW = O.philox_dW(99, 64, 24, 32, t0, t1)
    np.testing.assert_allclose(dW, exp_dW, rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(dU, O.philox_dU(99, 64, 24, 32, t0, t1, dW), rtol=1e-5, atol=1e-6)
    ref64, _ = O.solve_diffusion_model(pr['params'], 6, 17, pr['coeffs'], pr['times'], pr['y0'], np.asarray(ts, np.float32),
                                       dt, dW, method='srk', dtype=np.float64, dU=dU)
    assert_parity(ys, ref64, what='srk philox')
    halves = [hip_solve(pr, ts, dt, method='srk', seed=99, row_offset=64 + o, rows=slice(o, o + 12))[0] for o in (0, 12)]
    np.testing.assert_array_equal(np.concatenate(halves, axis=1), ys)


def test_ists_neuralsde_default_srk_on_cuda():
    """torch_ists wrapper: default method srk, ts = times, dt = min gap (nsde_model.py:63-84)."""
    pr = make_problem(42, 6, 17, 2, 10, 16, 3, 9, times=np.linspace(0, 1, 9).astype(np.float32))
    torch.manual_seed(0)
    func = S.Diffusion_model(3, 16, 16, 2, input_option=6, noise_option=17)
    model = S.IstsNeuralSDE(func, 3, 16, 2, initial=True).to(DEV).eval()
    with torch.no_grad():
        out, z = model(torch.from_numpy(pr['coeffs']).to(DEV), torch.from_numpy(pr['times']).to(DEV), options={'seed': 3})
    assert out.shape == (10, 9, 2) and z.shape == (10, 9, 16) and torch.isfinite(z).all()


# ---- spline coefficient construction on the GPU (A11 / A12) -----------------------------------------------
@pytest.mark.parametrize('case', G1_CASES)
def test_natural_spline_coeffs_hip_vs_reference_golden(case):
    g = group(SPL, f'G1/{case}/f32')
    out = S.controldiffeq.natural_cubic_spline_coeffs(torch.from_numpy(g['times']).to(DEV), torch.from_numpy(g['X']).to(DEV))
    for got, name in zip(out, ('a', 'b', 'two_c', 'three_d')):
        assert got.shape == g[name].shape
        np.testing.assert_allclose(got.cpu().numpy(), g[name], rtol=2e-5, atol=2e-5, err_msg=name)


def test_spline_construction_hip_vs_oracle_large():
    rng = np.random.default_rng(8)
    B, L, C = 257, 72, 9
    times = np.cumsum(rng.uniform(0.5, 1.5, L)).astype(np.float32)
    X = (rng.standard_normal((B, L, C)) * 0.3).cumsum(1).astype(np.float32)
    X[rng.random((B, L, C)) < 0.35] = np.nan
    X[0, :, 0] = np.nan
    X[1, 1:, 1] = np.nan
    Xd, td = torch.from_numpy(X).to(DEV), torch.from_numpy(times).to(DEV)
    nat = torch.cat(S.controldiffeq.natural_cubic_spline_coeffs(td, Xd), dim=-1).cpu().numpy()
    ref = np.concatenate(O.natural_cubic_spline_coeffs(times.astype(np.float64), X.astype(np.float64)), axis=-1)
    np.testing.assert_allclose(nat, ref, rtol=2e-4, atol=2e-4)
    her = S.torchcde.hermite_cubic_coefficients_with_backward_differences(Xd, td).cpu().numpy()
    ref_h = O.hermite_cubic_coefficients_with_backward_differences(X.astype(np.float64), times.astype(np.float64))
    np.testing.assert_allclose(her, ref_h, rtol=2e-4, atol=2e-4)
    # the CPU tensor-op construction (host path for CPU tensors) agrees too
    nat_cpu = torch.cat(S.controldiffeq.natural_cubic_spline_coeffs(torch.from_numpy(times), torch.from_numpy(X)), dim=-1)
    np.testing.assert_allclose(nat, nat_cpu.numpy(), rtol=2e-4, atol=2e-4)


# ---- randomized cross-checks between the kernel families ------------------------------------------------
# SNSDE_FUZZ_H (exploration runs): hidden sizes drawn, e.g. SNSDE_FUZZ_H=256 to fuzz the streamed-weight kernels only
_FUZZ_H = [int(x) for x in os.environ.get('SNSDE_FUZZ_H', '16,32,64,128').split(',')]
def _fuzz_configs(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        io = int(rng.integers(0, 7))
        # (11 = t*y is left to the option sweeps: with t up to 10 its dynamics amplify float32 round-off beyond a fixed tolerance)
        no = int(rng.choice([0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 12, 13, 16, 17, 14, 15, 18, 19]))
        method = str(rng.choice(['euler', 'milstein', 'srk']))
        if (no in (14, 15, 18, 19) and method != 'euler') or (io == 0 and method == 'srk'):
            continue
        H = int(rng.choice(_FUZZ_H))
        C = int(rng.choice([2, 5, 21, 33, 40])) if method != 'srk' else int(rng.choice([2, 5, 21]))
        if no in (14, 15, 18, 19) and io in (0, 2, 4, 6) and C > 32:
            continue
        out.append((io, no, int(rng.integers(1, 5)), int(rng.integers(3, 40)), H, C, int(rng.integers(5, 12)), method))
    return out


# SNSDE_FUZZ_SEED shifts both fuzz streams (exploration runs; the default is what CI runs)
_FUZZ_SHIFT = int(os.environ.get('SNSDE_FUZZ_SEED', '0'))


@pytest.mark.parametrize('cfg', _fuzz_configs(120, 2026 + _FUZZ_SHIFT))
def test_fuzz_mfma_forward_matches_generic_forward(cfg):
    """Random supported configurations: the MFMA kernels (both tile flavours where instantiated) against the generic
    all-options kernel on identical increments — two independent implementations of the same scheme."""
    io, no, NL, B, H, C, L, method = cfg
    sd = sum(int(v) * (i + 3) for i, v in enumerate(cfg[:7]))       # deterministic across processes
    pr = make_problem(sd, io, no, NL, B, H, C, L)
    ts = [0, (L - 1) / 2 + 0.25, L - 1]
    dW = draw_dW(sd % 1000, ts, 1.0, B, H)
    dU = _draw_dU(sd % 1000, dW, ts, 1.0) if method == 'srk' else None
    ref, _ = hip_solve(pr, ts, 1.0, dW=dW, dU=dU, method=method, kernel='generic')
    for kern in (('mfma4',) if method == 'srk' else ('mfma4', 'mfma16')):
        ys, _ = hip_solve(pr, ts, 1.0, dW=dW, dU=dU, method=method, kernel=kern)
        err = np.abs(ys - ref) / (1.0 + np.abs(ref))
        assert np.isfinite(ys).all() and err.max() < 5e-4 and err.mean() < 2e-6, (cfg, kern, err.max(), err.mean())


@pytest.mark.parametrize('cfg', [c for c in _fuzz_configs(120, 7 + _FUZZ_SHIFT) if c[1] not in (14, 15, 18, 19)][:72])
def test_fuzz_mfma_backward_matches_generic_backward(cfg):
    """Random configurations with an elementwise diffusion: gradients from the MFMA adjoint + native parameter pass against
    the generic adjoint + batched autograd pass (both fused paths, different kernels and different parameter passes).
    (Exploration runs with SNSDE_FUZZ_SEED: 3 of ~3000 shifted configurations exceeded the tolerance, each through ONE relu
    unit whose pre-activation is ~0 at one (row, step): the two float32 forwards put it on different sides of the kink, so
    every parameter upstream of that unit differs by that sample's contribution, while the other kernel family matches
    float64 autograd to 1e-7 (twice the MFMA side, once the generic side) — tools/fuzz_debug.py prints the per-kernel
    comparison against float64 and whether a disagreement is confined to one unit.)"""
    io, no, NL, B, H, C, L, method = cfg
    sd = sum(int(v) * (i + 5) for i, v in enumerate(cfg[:7]))
    pr = make_problem(sd, io, no, NL, B, H, C, L)
    ts = np.asarray([0, (L - 1) / 2 + 0.25, L - 1], np.float32)
    dW = draw_dW(sd % 1000, ts, 1.0, B, H)
    dU = _draw_dU(sd % 1000, dW, ts, 1.0) if method == 'srk' else None
    wsum = torch.from_numpy(np.random.default_rng(3).standard_normal((3, B, H)).astype(np.float32)).to(DEV)
    res = {}
    for kern in ('mfma4', 'generic'):
        m = S.Diffusion_model(C, H, H, NL, input_option=io, noise_option=no)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()})
        m = m.to(DEV)
        m.set_X(torch.from_numpy(pr['coeffs']).to(DEV), torch.from_numpy(pr['times']).to(DEV))
        y0 = torch.from_numpy(pr['y0']).to(DEV).requires_grad_(True)
        bm = _ReplayBM(torch.from_numpy(dW).to(DEV), None if dU is None else torch.from_numpy(dU).to(DEV))
        ys = S.sdeint(m, y0, torch.from_numpy(ts).to(DEV), bm=bm, method=method, dt=1.0, options={'kernel': kern})
        (ys * wsum).sum().backward()
        res[kern] = [y0.grad] + [p.grad for p in m.parameters()]
    for a, b in zip(res['mfma4'], res['generic']):
        if b is None or float(b.abs().max()) == 0.0:
            assert a is None or float(a.abs().max()) < 1e-6
            continue
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) / scale < 2e-3, (cfg, float((a - b).abs().max()) / scale)


LONG_CASES = [
    # io, no, NL, B, H, C, L, method, dt      long grids: several step-table chunks (128 rows each), many outputs
    (4, 17, 2, 19, 64, 5, 150, 'euler', 1.0),
    (6, 17, 2, 9, 32, 3, 135, 'milstein', 1.0),
    (2, 16, 1, 13, 128, 21, 141, 'euler', 0.5),
    (4, 17, 2, 11, 32, 5, 60, 'srk', 1.0),          # 180 drift passes
    (3, 13, 3, 7, 64, 3, 101, 'srk', 1.0),
    (1, 18, 2, 9, 64, 3, 135, 'euler', 1.0),
]


@pytest.mark.parametrize('case', LONG_CASES)
@pytest.mark.parametrize('outputs', ['ends', 'knots', 'rows'])
def test_long_grids_mfma_vs_generic_forward_and_backward(case, outputs):
    """Grids longer than one LDS step-table chunk, with final-only / every-knot / per-row outputs: MFMA kernels against the
    generic kernels (forward on Philox increments from the same key; gradients where both adjoints exist)."""
    io, no, NL, B, H, C, L, method, dt = case
    torch.manual_seed(1234)
    pr = make_problem(L * 7 + io, io, no, NL, B, H, C, L)
    times = torch.from_numpy(pr['times']).to(DEV)
    ts = times if outputs != 'ends' else times[[0, L - 1]]
    opts = {'seed': 77}
    if outputs == 'rows':
        opts['row_out'] = torch.randint(0, L, (B,), device=DEV)
    wsum = None
    res = {}
    for kern in ('mfma4', 'generic'):
        m = S.Diffusion_model(C, H, H, NL, input_option=io, noise_option=no)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()})
        m = m.to(DEV)
        m.set_X(torch.from_numpy(pr['coeffs']).to(DEV), times)
        grad = no not in (14, 15, 18, 19)          # the generic adjoint has no diffusion nets
        y0 = torch.from_numpy(pr['y0']).to(DEV).requires_grad_(grad)
        with torch.set_grad_enabled(grad):
            ys = S.sdeint(m, y0, ts, method=method, dt=dt, options=dict(opts, kernel=kern))
            if wsum is None:
                wsum = torch.randn_like(ys)
            if grad:
                (ys * wsum).sum().backward()
        res[kern] = (ys.detach(), [y0.grad] + [p.grad for p in m.parameters()] if grad else [])
    a, b = res['mfma4'][0], res['generic'][0]
    err = (a - b).abs() / (1.0 + b.abs())
    assert torch.isfinite(a).all() and float(err.max()) < 2e-3 and float(err.mean()) < 1e-5, (float(err.max()), float(err.mean()))
    for ga, gb in zip(res['mfma4'][1], res['generic'][1]):
        if gb is None or float(gb.abs().max()) == 0.0:
            assert ga is None or float(ga.abs().max()) < 1e-5
            continue
        assert torch.isfinite(gb).all() and torch.isfinite(ga).all()
        assert float((ga - gb).abs().max()) / float(gb.abs().max()) < 5e-3


@pytest.mark.parametrize('case', [(4, 17, 2, 2500, 128, 21, 12, 'euler'), (6, 17, 2, 1100, 64, 5, 10, 'milstein'),
                                  (2, 16, 3, 3000, 32, 3, 9, 'euler'), (4, 17, 2, 700, 256, 14, 8, 'milstein'),
                                  (4, 17, 2, 1300, 64, 5, 9, 'srk')])
def test_wide_batches_auto_selection_vs_generic(case):
    """Batches beyond one round of workgroups (automatic M4 / M16 choice, many R-splits in the weight-gradient pass):
    forward and gradients against the generic kernels."""
    io, no, NL, B, H, C, L, method = case
    torch.manual_seed(4321)
    pr = make_problem(B + L, io, no, NL, B, H, C, L)
    times = torch.from_numpy(pr['times']).to(DEV)
    ts = times[[0, L // 2, L - 1]]
    res = {}
    wsum = None
    for kern in ('auto', 'generic'):
        m = S.Diffusion_model(C, H, H, NL, input_option=io, noise_option=no)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()})
        m = m.to(DEV)
        m.set_X(torch.from_numpy(pr['coeffs']).to(DEV), times)
        y0 = torch.from_numpy(pr['y0']).to(DEV).requires_grad_(True)
        ys = S.sdeint(m, y0, ts, method=method, dt=1.0, options={'seed': 5, 'kernel': kern})
        wsum = torch.randn_like(ys) if wsum is None else wsum
        (ys * wsum).sum().backward()
        res[kern] = (ys.detach(), [y0.grad] + [p.grad for p in m.parameters()])
    a, b = res['auto'][0], res['generic'][0]
    err = (a - b).abs() / (1.0 + b.abs())
    assert float(err.max()) < 2e-3 and float(err.mean()) < 1e-5
    for ga, gb in zip(res['auto'][1], res['generic'][1]):
        if gb is None or float(gb.abs().max()) == 0.0:
            assert ga is None or float(ga.abs().max()) < 1e-4
            continue
        assert float((ga - gb).abs().max()) / float(gb.abs().max()) < 5e-3


# ---- generic sde (arbitrary f / g modules): the hipGraph-replayed stepper -----------------------
class _LipSwish(torch.nn.Module):
    def forward(self, x):
        return 0.909 * torch.nn.functional.silu(x)


class _TutorialField(torch.nn.Module):
    """The tutorial's 'pure' LSDE vector field (tutorial/*.ipynb cell 7): LipSwish MLPs, control path through
    torchcde.CubicSpline, time-only diffusion."""
    sde_type, noise_type = 'ito', 'diagonal'

    def __init__(self, input_dim, hidden, sync=False):
        super().__init__()
        mlp = lambda i, o: torch.nn.Sequential(torch.nn.Linear(i, hidden), _LipSwish(), torch.nn.Linear(hidden, o))
        self.linear_X = torch.nn.Linear(input_dim, hidden)
        self.emb = torch.nn.Linear(2 * hidden, hidden)
        self.f_net, self.linear_out = mlp(hidden, hidden), torch.nn.Linear(hidden, hidden)
        self.noise_in, self.g_net = torch.nn.Linear(1, hidden), mlp(hidden, hidden)
        self.sync = sync

    def set_X(self, coeffs, times):
        self.X = S.torchcde.CubicSpline(coeffs, times)

    def f(self, t, y):
        if self.sync:
            float(t)                      # a device->host sync: such a step cannot be recorded
        z = self.emb(torch.cat([y, self.linear_X(self.X.evaluate(t))], dim=-1))
        return self.linear_out(self.f_net(z))

    def g(self, t, y):
        tt = torch.full_like(y[:, :1], 1.0) * t
        return self.g_net(self.noise_in(tt))


@pytest.mark.parametrize('method', ['euler', 'srk'])
@pytest.mark.parametrize('sync', [False, True])
def test_generic_sde_graph_replayed_stepper_equals_the_eager_loop(method, sync):
    torch.manual_seed(0)
    B, H, C, L = 16, 32, 2, 20
    field = _TutorialField(C, H, sync=sync).to(DEV)
    times = torch.linspace(0, 1, L, device=DEV)
    X = torch.cumsum(torch.randn(B, L, C, device=DEV) * 0.1, 1)
    field.set_X(S.torchcde.hermite_cubic_coefficients_with_backward_differences(X, times), times)
    y0 = torch.randn(B, H, device=DEV)
    with torch.no_grad():
        a = S.sdeint(field, y0, times, dt=0.05, method=method, options={'seed': 11})                     # graph (or fallback)
        b = S.sdeint(field, y0, times, dt=0.05, method=method, options={'seed': 11, 'graph': False})     # eager loop
    assert a.shape == (L, B, H) and torch.isfinite(a).all()
    assert torch.equal(a, b)
