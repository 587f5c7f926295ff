"""GPU: the fused HIP kernels (through the C ABI) against the exact-arithmetic known-answer vectors of tests/golden/exact.npz.

The vectors come from tests/golden/make_exact_golden.py: fractions.Fraction, the PUBLISHED SRI2W1 table (= torchsde srid2.py),
the definitions of Euler-Maruyama / Milstein, torchsde's fixed-step grid with float32 time accumulation - nothing of oracle/ or
of this package goes into them.  The K/* fields are ones the kernels evaluate as exactly-rational functions (relu MLPs with dyadic
weights, linear drift output, un-squashed diffusion, raw time feature: the variant switches of include/snsde.h), so a float32
kernel must land within round-off of the rational result.  Tolerance: |err| <= 2e-5 (1 + |y|) elementwise, float32 round-off
through <= 7 steps of three 16..64-wide layers (measured margin in the assertion message)."""
import numpy as np
import pytest
import torch

import stable_neural_sdes_amd as S
from tests.helpers import group, load

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
EX = load('exact.npz')
METHOD = {'srk': 'srk', 'srk2': 'srk', 'euler_mis': 'euler', 'milstein_mis': 'milstein'}
FIELDS = {          # prefix: (input_option, noise_option, diffusion_output, kernels to try)
    'K/tab': (4, 13, 1),        # g = s(t) y from a supplied table: lean kernel (Euler / Milstein), general SRK variant
    'K/net16': (3, 18, 2),      # two-layer diffusion net: snsde_m4n_kernel (Euler / Milstein / SRK through the net)
    'K/net64': (3, 18, 2),
}


def build(prefix, case):
    io, no, dout = FIELDS[prefix]
    g = group(EX, f'{prefix}/{case}')
    p = group(EX, f'{prefix}/param')
    H = p['linear_out.bias'].shape[0]
    if prefix == 'K/tab':
        times, coeffs = EX['K/tab/times'], EX['K/tab/coeffs']
    else:          # no control path inside the field: one zero channel over the solve's interval
        times = np.array([0.0, 1.0], np.float32)
        coeffs = np.zeros((g['y0'].shape[0], 1, 4), np.float32)
    C = coeffs.shape[-1] // 4
    model = S.engine.model_struct(C, H, H, 2, io, no, activation=0, drift_output=1, diffusion_output=dout, time_feature=1)
    layout, numel = S._lib.param_layout(model)
    flat = np.zeros(numel, np.float32)
    for name, off, shape in layout:
        if name in p:
            assert tuple(p[name].shape) == tuple(shape), (name, p[name].shape, shape)
            flat[off:off + p[name].size] = p[name].reshape(-1)
    return model, torch.from_numpy(flat).to(DEV), torch.from_numpy(coeffs).to(DEV), times, g


@pytest.mark.parametrize('kernel', ['auto', 'mfma4'])
@pytest.mark.parametrize('case', sorted(METHOD))
@pytest.mark.parametrize('prefix', sorted(FIELDS))
def test_kernels_reproduce_the_exact_rational_trajectories(prefix, case, kernel):
    method = METHOD[case]
    model, flat, coeffs, times, g = build(prefix, case)
    dev = torch.device(DEV)
    grid = S.engine.step_grid(g['ts'], float(g['dt']), times, dev)
    # the grid the library built is the generator's, bit for bit (fp32 accumulation, sliver step, interpolation weights)
    np.testing.assert_array_equal(grid.t0, g['t0'])
    np.testing.assert_array_equal(grid.t1, g['t1'])
    np.testing.assert_array_equal(np.asarray(grid.out_w).reshape(-1, 2)[:, 0], g['w0'])
    np.testing.assert_array_equal(np.asarray(grid.out_w).reshape(-1, 2)[:, 1], g['w1'])
    B = g['y0'].shape[0]
    path = S.engine.forward_path(model, B, len(times), grid.N, method=method, kernel=kernel, table='noise_table' in g)
    assert path not in ('none', 'generic', 'loop'), (prefix, case, kernel, path)
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
    tab = f32(g['noise_table']) if 'noise_table' in g else None
    call = S.engine.SolveCall(model, flat, coeffs, grid, f32(g['y0']), dW=f32(g['dW']), dU=f32(g['dU']) if 'dU' in g else None,
                              method=method, kernel=kernel, noise_table=tab, save_traj=True)
    ys = call.launch().double().cpu().numpy()
    traj = call.traj.double().cpu().numpy()
    for got, want, what in ((ys, g['ys'], 'ys'), (traj, g['traj'], 'traj')):
        err = np.abs(got - want) / (1.0 + np.abs(want))
        assert err.max() <= 2e-5, (prefix, case, kernel, path, what, float(err.max()))


@pytest.mark.parametrize('case', ['srk', 'srk2'])
@pytest.mark.parametrize('prefix', sorted(FIELDS))
def test_srk_kernels_do_not_step_with_the_sri1w1_rows_any_more(prefix, case):
    """Rounds 1 - 5 carried B(1) / beta(2) of SRI1W1 (torchsde srid1) inside SRI2W1: also order 1.5, but a different trajectory.
    The scalar case stores that trajectory; here: one more float64 evaluation of the K/* field with the old rows must be FAR from
    what the kernel returns (so the agreement above is not an accident of a degenerate field)."""
    from tests.test_exact_cpu import exact_field
    model, flat, coeffs, times, g = build(prefix, case)
    f, gf, _, _ = exact_field(prefix)
    B1_old = ((), (0.5,), (-1.0, 0.0), (-5.0, 3.0, 0.5))
    beta2_old = (-1.0, 4 / 3, -1 / 3, 0.0)
    from oracle import sde_oracle as O
    saved = O.SRK_B1, O.SRK_BETA2
    try:
        O.SRK_B1, O.SRK_BETA2 = B1_old, beta2_old
        ys_old, _ = O.integrate(f, gf, g['y0'], g['ts'], float(g['dt']), g['dW'], method='srk', dU=g['dU'])
    finally:
        O.SRK_B1, O.SRK_BETA2 = saved
    dev = torch.device(DEV)
    grid = S.engine.step_grid(g['ts'], float(g['dt']), times, dev)
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
    tab = f32(g['noise_table']) if 'noise_table' in g else None
    call = S.engine.SolveCall(model, flat, coeffs, grid, f32(g['y0']), dW=f32(g['dW']), dU=f32(g['dU']), method='srk', noise_table=tab)
    ys = call.launch().double().cpu().numpy()
    assert np.abs(ys - g['ys']).max() <= 2e-5 * (1 + np.abs(g['ys']).max())
    assert np.abs(ys - ys_old).max() > 1e-2


class _Poly(torch.nn.Module):
    noise_type, sde_type = 'diagonal', 'ito'

    def __init__(self):
        super().__init__()
        self.register_buffer('a', torch.from_numpy(EX['A/a']))
        self.register_buffer('b', torch.from_numpy(EX['A/b']))

    def f(self, t, y):
        a = self.a
        return a[0] + a[1] * t + a[2] * y + a[3] * t * y + a[4] * y * y

    def g(self, t, y):
        b = self.b
        return b[0] + b[1] * t + b[2] * y + b[3] * t * y


@pytest.mark.parametrize('case', sorted(METHOD))
def test_sdeint_on_an_unrecognised_module_on_the_gpu(case):
    """The scalar polynomial SDE through the public sdeint on cuda (graph-replayed tensor-op stepper), float64."""
    from tests.test_exact_cpu import ReplayBM
    g = group(EX, f'A/{case}')
    dU = g.get('dU')
    bm = ReplayBM(torch.from_numpy(g['dW']).to(DEV), None if dU is None else torch.from_numpy(dU).to(DEV))
    ys = S.torchsde.sdeint(_Poly().to(DEV), torch.from_numpy(g['y0']).to(DEV), torch.from_numpy(g['ts']).to(DEV), bm=bm,
                           method=METHOD[case], dt=float(g['dt']))
    np.testing.assert_allclose(ys.cpu().numpy(), g['ys'], rtol=1e-12, atol=1e-14)


GRAD_CASES = ['srk', 'euler_mis', 'milstein_mis']


@pytest.mark.parametrize('kernel', ['auto', 'mfma4'])
@pytest.mark.parametrize('case', GRAD_CASES)
@pytest.mark.parametrize('prefix', ['K/tab', 'K/net16'])
def test_adjoint_kernels_reproduce_the_exact_directional_derivatives(prefix, case, kernel):
    """The fused backward (adjoint kernel + native weight-gradient pass, one C call) against EXACT derivatives: the generator
    differentiates L = sum w . ys in forward mode over the rationals (make_exact_golden.Dual) along a random dyadic direction per
    parameter tensor, over y0, and per batch row of y0.  <gradient, direction> from the kernels must match each to float32 round-off:
    |err| <= 2e-4 max(|dL|, scale) with scale = the largest |<gradient, direction>| of the case (sums of ~1e3 products of O(1) terms)."""
    method = METHOD[case]
    model, flat, coeffs, times, g = build(prefix, case)
    dev = torch.device(DEV)
    grid = S.engine.step_grid(g['ts'], float(g['dt']), times, dev)
    B = g['y0'].shape[0]
    assert S.engine.backward_mode(model, B, len(times), grid, method, kernel=kernel, table='noise_table' in g) == 1
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
    tab = f32(g['noise_table']) if 'noise_table' in g else None
    call = S.engine.SolveCall(model, flat, coeffs, grid, f32(g['y0']), dW=f32(g['dW']), dU=f32(g['dU']) if 'dU' in g else None,
                              method=method, kernel=kernel, noise_table=tab, save_traj=True, save_act=True,
                              save_dW=method == 'srk')      # (the SRK adjoint reads the forward's dU_out)
    ys = call.launch()
    assert np.abs(ys.double().cpu().numpy() - g['ys']).max() <= 2e-5 * (1 + np.abs(g['ys']).max())
    # (Milstein through a diffusion net: its second-order weight-gradient jobs read every adjoint state - no adj0-only form)
    adj, grad = S.engine.backward_with_gradients(call, f32(g['grad/w']), adj0_only=S.engine.adj0_suffices(call))[:2]
    adj0 = adj[0].double().cpu().numpy()
    grad = grad.double().cpu().numpy()
    layout, _ = S._lib.param_layout(model)
    where = {name: (off, shape) for name, off, shape in layout}
    got, want = {}, {}
    for key in g:
        if not key.startswith('grad/dir/'):
            continue
        name = key[len('grad/dir/'):]
        v = g[key]
        want[name] = float(g['grad/dL/' + name])
        if name.startswith('y0'):
            got[name] = float((adj0 * v).sum())
        else:
            off, shape = where[name]
            got[name] = float((grad[off:off + v.size].reshape(v.shape) * v).sum())
    assert len(got) >= 7 + B
    scale = max(abs(x) for x in want.values())
    bad = {n: (got[n], want[n]) for n in got if abs(got[n] - want[n]) > 2e-4 * max(abs(want[n]), scale)}
    assert not bad, (prefix, case, kernel, scale, bad)
