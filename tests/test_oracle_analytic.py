"""Known-answer tests for the parts of the oracle that restate third-party behaviour whose source
is not in the reference tree (torchsde integrate/Euler/Milstein, torchcde Hermite, Philox):
SURVEY.md section 8c "parity unpinned" items (i)-(iv).  CPU only."""
import numpy as np
import pytest

from oracle import sde_oracle as O


def test_philox_random123_known_answers():
    kat = [((0, 0, 0, 0), (0, 0), '6627e8d5 e169c58d bc57ac4c 9b00dbd8'),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, '408f276d 41c83b0e a20bc7c6 6d5451fd'),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            'd16cfe09 94fdcceb 5001e420 24126ea1')]
    for c, k, exp in kat:
        r = O.philox4x32_10(*[np.array([v], dtype=np.uint32) for v in c], *k)
        assert ' '.join(f'{int(x[0]):08x}' for x in r) == exp


def test_philox_normals_statistics_and_sharding():
    z = O.philox_normals(2024, np.arange(4096), 3, 128)
    assert z.shape == (4096, 128) and z.dtype == np.float32
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1) < 5e-3
    assert abs((z ** 3).mean()) < 2e-2 and abs((z ** 4).mean() - 3) < 5e-2
    # counters use the GLOBAL row: any shard reproduces its slice bit-exactly
    z_hi = O.philox_normals(2024, np.arange(2048, 4096), 3, 128)
    np.testing.assert_array_equal(z[2048:], z_hi)
    # different step / seed -> different stream
    assert not np.array_equal(z, O.philox_normals(2024, np.arange(4096), 4, 128))
    assert not np.array_equal(z, O.philox_normals(2025, np.arange(4096), 3, 128))
    # H not a multiple of 4
    z5 = O.philox_normals(2024, np.arange(8), 3, 5)
    np.testing.assert_array_equal(z5, z[:8, :5])


def test_step_grid_integer_grid_is_exact():
    t0, t1, out_step, w0, w1 = O.step_grid(np.array([0., 100.], np.float32), 1.0)
    np.testing.assert_array_equal(t0, np.arange(100, dtype=np.float32))
    np.testing.assert_array_equal(t1, np.arange(1, 101, dtype=np.float32))
    assert list(out_step) == [99] and w0[0] == 0 and w1[0] == 1


def test_step_grid_multi_output_aligned():
    ts = np.array([1., 4., 6., 8.], np.float32)
    t0, t1, out_step, w0, w1 = O.step_grid(ts, 1.0)
    assert len(t0) == 7
    assert list(out_step) == [2, 4, 6]
    assert np.all(w0 == 0) and np.all(w1 == 1)


def test_step_grid_misaligned_interpolates_and_clamps():
    ts = np.linspace(0, 1, 20).astype(np.float32)
    t0, t1, out_step, w0, w1 = O.step_grid(ts, 0.05)
    # fp32 accumulation of 0.05 : 20 steps land at/near 1.0; the last is clamped to ts[-1]
    assert t1[-1] == ts[-1]
    assert np.all(t1 > t0)
    assert np.all(np.diff(out_step) >= 0)
    np.testing.assert_allclose(w0 + w1, 1.0, atol=1e-6)
    assert np.all((w0 >= 0) & (w1 >= 0))
    # every output time lies inside its step
    for k, n in enumerate(out_step):
        assert t0[n] <= ts[k + 1] <= t1[n]


def test_step_grid_two_outputs_in_one_step():
    ts = np.array([0., 0.3, 0.6, 2.0], np.float32)
    t0, t1, out_step, w0, w1 = O.step_grid(ts, 1.0)
    assert list(out_step) == [0, 0, 1]
    np.testing.assert_allclose(w1[:2], [0.3, 0.6], rtol=1e-6)


def test_step_grid_errors():
    with pytest.raises(ValueError):
        O.step_grid(np.array([0., 0.], np.float32), 1.0)
    with pytest.raises(ValueError):
        O.step_grid(np.array([1., 0.5], np.float32), 1.0)
    with pytest.raises(ValueError):
        O.step_grid(np.array([0., 1.], np.float32), 0.0)
    with pytest.raises(ValueError):   # 1e8 + 1 does not advance in fp32
        O.step_grid(np.array([1e8, 1e8 + 64], np.float32), 1.0)


def test_constant_drift_zero_noise_is_a_line():
    B, H = 3, 4
    c = np.linspace(-1, 1, B * H).reshape(B, H)
    ts = np.array([0., 0.25, 0.7, 2.0], np.float32)
    t0, *_ = O.step_grid(ts, 0.05)
    dW = np.random.default_rng(0).standard_normal((len(t0), B, H))
    ys, traj = O.integrate(lambda t, y: c, lambda t, y: np.zeros_like(y), np.zeros((B, H)), ts, 0.05, dW)
    for k, t in enumerate(ts):
        np.testing.assert_allclose(ys[k], c * np.float64(t), rtol=1e-6, atol=1e-7)


def test_ou_euler_closed_form_recursion():
    """dX = -a X dt + s dW: Euler gives X_{n+1} = (1 - a h) X_n + s dW_n exactly."""
    rng = np.random.default_rng(1)
    a, s, h, N = 0.7, 0.3, 0.125, 64
    x0 = rng.standard_normal((5, 2))
    dW = rng.standard_normal((N, 5, 2)) * np.sqrt(h)
    ts = np.array([0., N * h], np.float32)
    ys, traj = O.integrate(lambda t, y: -a * y, lambda t, y: s * np.ones_like(y), x0, ts, h, dW)
    x = x0.copy()
    for n in range(N):
        x = (1 - a * h) * x + s * dW[n]
    np.testing.assert_allclose(ys[-1], x, rtol=1e-12)


def _gbm_strong_error(method, N, rng_seed=3, paths=4000):
    mu, sig, T = 0.5, 0.8, 1.0
    fine = 2 ** 10
    rng = np.random.default_rng(rng_seed)
    dWf = rng.standard_normal((fine, paths, 1)) * np.sqrt(T / fine)
    W_T = dWf.sum(0)
    exact = np.exp((mu - 0.5 * sig ** 2) * T + sig * W_T)
    dW = dWf.reshape(N, fine // N, paths, 1).sum(1)
    ts = np.array([0., T], np.float32)
    ys, _ = O.integrate(lambda t, y: mu * y, lambda t, y: sig * y, np.ones((paths, 1)), ts, T / N, dW,
                        method=method, gdg=lambda t, y: (sig * y, sig * np.ones_like(y)))
    return np.mean(np.abs(ys[-1] - exact))


def test_strong_order_euler_half_milstein_one():
    e_eu = [_gbm_strong_error('euler', N) for N in (16, 64, 256)]
    e_mi = [_gbm_strong_error('milstein', N) for N in (16, 64, 256)]
    order_eu = np.log(e_eu[0] / e_eu[-1]) / np.log(16)
    order_mi = np.log(e_mi[0] / e_mi[-1]) / np.log(16)
    assert 0.35 < order_eu < 0.7, order_eu
    assert 0.85 < order_mi < 1.2, order_mi
    assert e_mi[-1] < e_eu[-1]


def test_hermite_backward_difference_properties():
    rng = np.random.default_rng(5)
    B, L, C = 3, 7, 2
    t = np.cumsum(rng.uniform(0.2, 1.0, L))
    X = rng.standard_normal((B, L, C)).cumsum(1)
    co = O.hermite_cubic_coefficients_with_backward_differences(X, t)
    assert co.shape == (B, L - 1, 4 * C)
    a, b, c2, d3 = (co[..., k * C:(k + 1) * C] for k in range(4))
    h = (t[1:] - t[:-1])[None, :, None]
    m = (X[:, 1:] - X[:, :-1]) / h
    # interpolates the knots: p(0) = x_k, p(h) = x_{k+1}
    np.testing.assert_allclose(a, X[:, :-1])
    np.testing.assert_allclose(a + b * h + 0.5 * c2 * h ** 2 + d3 * h ** 3 / 3, X[:, 1:], atol=1e-12)
    # p'(0) = previous secant slope (first interval: its own), p'(h) = this interval's secant slope
    np.testing.assert_allclose(b[:, 1:], m[:, :-1])
    np.testing.assert_allclose(b[:, 0], m[:, 0])
    np.testing.assert_allclose(b + c2 * h + d3 * h ** 2, m, atol=1e-12)
    # evaluation through the spline evaluator at knots and just before the next knot
    for k in range(L):
        np.testing.assert_allclose(O.spline_evaluate(co, t, t[k]), X[:, k], atol=1e-10)


def test_hermite_nan_fill_linear():
    t = np.arange(6, dtype=np.float64)
    X = np.array([np.nan, 1.0, np.nan, np.nan, 4.0, np.nan])[None, :, None]
    co = O.hermite_cubic_coefficients_with_backward_differences(X, t)
    np.testing.assert_allclose(co[0, :, 0], [1.0, 1.0, 2.0, 3.0, 4.0])


def test_milstein_closed_form_matches_autograd_style_fd():
    rng = np.random.default_rng(9)
    from tests.helpers import random_params
    B, H, C = 4, 8, 3
    y = rng.standard_normal((B, H))
    for no in O.MILSTEIN_ELEMENTWISE_NO:
        p = O.cast_params(random_params(rng, 4, no, 2, C, H), np.float64)
        g, dg = O.diffusion_g_dgdy(p, no, 0.7, y)
        eps = 1e-6
        fd = (O.diffusion_g(p, no, 0.7, y + eps) - O.diffusion_g(p, no, 0.7, y - eps)) / (2 * eps)
        if no == 10:
            mask = np.abs(y) > 1e-3
            np.testing.assert_allclose(dg[mask], fd[mask], rtol=1e-5, atol=1e-7)
        else:
            np.testing.assert_allclose(dg, fd, rtol=1e-5, atol=1e-7, err_msg=str(no))


def test_srk_strong_order_one_and_a_half_on_gbm():
    """SRID2 with exact (I_k, I_k0) pairs: strong order ~1.5 on geometric Brownian motion, far below Milstein's error."""
    mu, sig, T, paths, fine = 0.5, 0.8, 1.0, 4000, 2 ** 10
    rng = np.random.default_rng(11)
    hf = T / fine
    dWf = rng.standard_normal((fine, paths, 1)) * np.sqrt(hf)
    dUf = hf * (0.5 * dWf + np.sqrt(hf / 12) * rng.standard_normal((fine, paths, 1)))   # int (W_s - W_t0) ds on fine cells
    exact = np.exp((mu - 0.5 * sig ** 2) * T + sig * dWf.sum(0))
    errs = []
    for N in (8, 32, 128):
        m = fine // N
        dWc = dWf.reshape(N, m, paths, 1)
        dW = dWc.sum(1)
        # coarse I_k0 = sum_i [ dU_i + (W_i - W_0) hf ]  over the fine cells of the coarse step
        Wpre = np.cumsum(dWc, axis=1) - dWc
        dU = (dUf.reshape(N, m, paths, 1) + Wpre * hf).sum(1)
        ts = np.array([0., T], np.float32)
        ys, _ = O.integrate(lambda t, y: mu * y, lambda t, y: sig * y, np.ones((paths, 1)), ts, T / N, dW, method='srk', dU=dU)
        errs.append(np.mean(np.abs(ys[-1] - exact)))
    order = np.log(errs[0] / errs[-1]) / np.log(16)
    assert 1.25 < order < 1.9, (order, errs)
    assert errs[1] < 0.2 * _gbm_strong_error('milstein', 32, rng_seed=11)
