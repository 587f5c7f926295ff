"""The numpy oracle must reproduce every golden vector generated from the reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import sde_oracle as O
from tests.helpers import group, load, param_spec, params_of, unflatten

SPL = load('spline.npz')
FG = load('fg.npz')
TRAJ = load('traj.npz')
WRAP = load('wrapper.npz')

G1_CASES = sorted({k.split('/')[1] for k in SPL.files if k.startswith('G1/')})


@pytest.mark.parametrize('case', G1_CASES)
@pytest.mark.parametrize('prec', ['f32', 'f64'])
def test_g1_natural_spline_coeffs(case, prec):
    g = group(SPL, f'G1/{case}/{prec}')
    out = O.natural_cubic_spline_coeffs(g['times'], g['X'])
    tol = dict(rtol=2e-5, atol=2e-5) if prec == 'f32' else dict(rtol=1e-11, atol=1e-11)
    for got, name in zip(out, ('a', 'b', 'two_c', 'three_d')):
        assert got.dtype == g[name].dtype
        np.testing.assert_allclose(got, g[name], err_msg=name, **tol)


@pytest.mark.parametrize('case', G1_CASES)
@pytest.mark.parametrize('prec', ['f32', 'f64'])
def test_g2_spline_evaluate(case, prec):
    c = group(SPL, f'G1/{case}/{prec}')
    g = group(SPL, f'G2/{case}/{prec}')
    coeffs = np.concatenate([c['a'], c['b'], c['two_c'], c['three_d']], axis=-1)
    for i, t in enumerate(g['t']):
        ev = O.spline_evaluate(coeffs, c['times'], t)
        de = O.spline_derivative(coeffs, c['times'], t)
        if prec == 'f32':
            # same operation order as the reference: bit-exact on CPU
            np.testing.assert_array_equal(ev, g['evaluate'][i])
            np.testing.assert_array_equal(de, g['derivative'][i])
        else:
            np.testing.assert_allclose(ev, g['evaluate'][i], rtol=1e-14, atol=1e-14)
            np.testing.assert_allclose(de, g['derivative'][i], rtol=1e-14, atol=1e-14)


MODELS = FG['G3/models']


@pytest.mark.parametrize('mi', range(len(MODELS)))
def test_g3_f_g_all_options(mi):
    io, no, NL = (int(v) for v in MODELS[mi])
    coeffs, times, y, tv = FG['G3/coeffs'], FG['G3/times'], FG['G3/y'], FG['G3/t']
    B, H = y.shape
    C = coeffs.shape[-1] // 4
    off = FG['G3/params_off']
    p32 = unflatten(FG['G3/params_flat'][off[mi]:off[mi + 1]], param_spec(io, no, NL, C, H))
    for prec, dtype, tol in (('32', np.float32, dict(rtol=1e-5, atol=2e-6)),
                             ('64', np.float64, dict(rtol=1e-12, atol=1e-13))):
        p = O.cast_params(p32, dtype)
        exp = FG[f'G3/out{prec}'][mi]
        for ti, t in enumerate(tv):
            t = dtype(t)
            Xt = O.spline_evaluate(coeffs.astype(dtype), times.astype(dtype), t)
            f = O.drift_f(p, io, t, y.astype(dtype), Xt)
            g = O.diffusion_g(p, no, t, y.astype(dtype))
            assert f.dtype == dtype and g.dtype == dtype
            np.testing.assert_allclose(f, exp[0, ti], err_msg=f'f io={io} no={no}', **tol)
            np.testing.assert_allclose(g, exp[1, ti], err_msg=f'g io={io} no={no}', **tol)


G5_CASES = sorted({k.split('/')[1] for k in TRAJ.files if k.startswith('G5/')})


@pytest.mark.parametrize('case', G5_CASES)
def test_g5_trajectories(case):
    g = group(TRAJ, f'G5/{case}')
    io, no, NL = (int(v) for v in g['io_no_nl'])
    method = str(g['method'])
    p = params_of(TRAJ, f'G5/{case}')
    ys64, _ = O.solve_diffusion_model(p, io, no, g['coeffs'], g['times'], g['y0'], g['ts'], float(g['dt']),
                                      g['dW'], method=method, dtype=np.float64)
    ys32, _ = O.solve_diffusion_model(p, io, no, g['coeffs'], g['times'], g['y0'], g['ts'], float(g['dt']),
                                      g['dW'], method=method, dtype=np.float32)
    assert ys64.shape == g['ys64'].shape
    scale = 1.0 + np.abs(g['ys64'])
    # fp64 oracle vs reference f/g in fp64 under the same step loop: only summation-order noise
    assert np.max(np.abs(ys64 - g['ys64']) / scale) < 1e-10
    # fp32 oracle vs fp32 reference modules: both within fp32 round-off of the fp64 arbiter
    err_ref = np.max(np.abs(g['ys32'] - g['ys64']) / scale)
    err_ora = np.max(np.abs(ys32 - g['ys64']) / scale)
    assert err_ora < max(4 * err_ref, 2e-6), (err_ora, err_ref)


def test_step_grid_matches_recorded_counts():
    for case in G5_CASES:
        g = group(TRAJ, f'G5/{case}')
        t0, t1, out_step, w0, w1 = O.step_grid(g['ts'], float(g['dt']))
        assert t0.shape[0] == g['dW'].shape[0], case
        assert out_step.shape[0] == g['ts'].shape[0] - 1
        assert t1[-1] == g['ts'][-1]


@pytest.mark.parametrize('name', ['arange', 'lin01_20', 'tiny', 'irr'])
def test_g4_dt_rule(name):
    """_prepare_sde_solver_kwargs: dt = max(min(diff(times)), 1e-3) (neuralsde.py:30-33)."""
    g = group(WRAP, f'G4/dt/{name}')
    times = g['times']
    dt = max(float((times[1:] - times[:-1]).min()), 1e-3)
    assert dt == float(g['dt'])
