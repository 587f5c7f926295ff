"""Shared test helpers: fixture loading and the reference's parameter naming contract."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def param_spec(io, no, NL, C, H, HH=None):
    """state_dict names/shapes of the reference Diffusion_model, in state_dict order
    (/root/reference/benchmark_classification/models_sde/neuralsde.py:123-179)."""
    HH = H if HH is None else HH
    spec = [('theta', (1, 1))]
    if no in (1, 2, 3):
        spec.append(('sigma', (1,)))
    if no in (4, 5, 6):
        spec.append(('sigma_diag', (H,)))
    spec += [('initial_network.weight', (H, C)), ('initial_network.bias', (H,))]
    k_in = H + 2 if io in (3, 4, 5, 6) else H
    spec += [('linear_in.weight', (HH, k_in)), ('linear_in.bias', (HH,))]
    if io in (2, 4, 6):
        spec += [('emb.weight', (H, 2 * H)), ('emb.bias', (H,))]
    for i in range(NL - 1):
        spec += [(f'linears.{i}.weight', (HH, HH)), (f'linears.{i}.bias', (HH,))]
    spec += [('linear_out.weight', (H, HH)), ('linear_out.bias', (H,))]
    if no in (12, 13):
        spec += [('noise_t.weight', (H, 2)), ('noise_t.bias', (H,))]
    if no in (14, 15):
        spec += [('noise_y.weight', (H, H + 2)), ('noise_y.bias', (H,))]
    if no in (16, 17):
        spec += [('noise_t.0.weight', (H, 2)), ('noise_t.0.bias', (H,)),
                 ('noise_t.2.weight', (H, H)), ('noise_t.2.bias', (H,))]
    if no in (18, 19):
        spec += [('noise_y.0.weight', (H, H + 2)), ('noise_y.0.bias', (H,)),
                 ('noise_y.2.weight', (H, H)), ('noise_y.2.bias', (H,))]
    return spec


def unflatten(vec, spec):
    out, off = {}, 0
    for name, shape in spec:
        n = int(np.prod(shape))
        out[name] = np.asarray(vec[off:off + n]).reshape(shape)
        off += n
    assert off == len(vec), (off, len(vec))
    return out


def group(npz, prefix):
    """Sub-dict of an npz whose keys start with prefix/ (prefix stripped)."""
    pre = prefix + '/'
    return {k[len(pre):]: npz[k] for k in npz.files if k.startswith(pre)}


def params_of(npz, prefix):
    return group(npz, prefix + '/param')


def random_params(rng, io, no, NL, C, H, scale=None):
    """nn.Linear-like init (U(-1/sqrt(fan_in), 1/sqrt(fan_in))) with non-trivial theta/sigma."""
    p = {}
    for name, shape in param_spec(io, no, NL, C, H):
        if name == 'theta':
            p[name] = np.array([[0.8]], dtype=np.float32)
        elif name in ('sigma', 'sigma_diag'):
            p[name] = (-0.5 + 0.3 * rng.standard_normal(shape)).astype(np.float32)
        else:
            fan_in = shape[-1] if name.endswith('weight') else None
            if fan_in is None:
                wname = name[:-4] + 'weight'
                fan_in = dict(param_spec(io, no, NL, C, H))[wname][-1]
            b = (scale or 1.0) / np.sqrt(fan_in)
            p[name] = rng.uniform(-b, b, size=shape).astype(np.float32)
    return p


# ---------------------------------------------------------------------------------------------------
# synthetic problems + the stated fp32 parity criterion
# ---------------------------------------------------------------------------------------------------
def make_problem(seed, io, no, NL, B, H, C, L, times=None, nan_frac=0.2, y0_scale=0.5, hermite=False, weight_scale=None):
    """Seeded inputs for one solve: params (reference names), coeffs (B, L-1, 4C), times, y0."""
    import torch
    import stable_neural_sdes_amd as S
    rng = np.random.default_rng(seed)
    p = random_params(rng, io, no, NL, C, H, scale=weight_scale)
    if times is None:
        times = np.arange(L, dtype=np.float32)
    times = np.asarray(times, dtype=np.float32)
    X = (rng.standard_normal((B, L, C)) * 0.1).cumsum(1).astype(np.float32)
    X[:, :, 0] = times[None, :]          # channel 0 = time, as in the reference's datasets
    if nan_frac > 0:
        mask = rng.random((B, L, C)) < nan_frac
        mask[:, :, 0] = False
        X[mask] = np.nan
    # The coefficients of the parity inputs are built with THIS package's CPU construction (the batched tensor-op version of
    # controldiffeq.natural_cubic_spline_coeffs / the Hermite formula).  That is not circular for the solver tests: the same
    # coefficient array goes to the kernels and to the oracle, and the construction itself is pinned bit-level to the
    # reference's own vendored code by the G1 fixtures (tests/test_oracle_golden.py, test_gpu_parity.py::test_natural_spline_*).
    Xt, tt = torch.from_numpy(X), torch.from_numpy(times)
    if hermite:
        coeffs = S.torchcde.hermite_cubic_coefficients_with_backward_differences(Xt, tt)
    else:
        coeffs = torch.cat(S.controldiffeq.natural_cubic_spline_coeffs(tt, Xt), dim=-1)
    y0 = (y0_scale * rng.standard_normal((B, H))).astype(np.float32)
    return dict(params=p, coeffs=coeffs.numpy().astype(np.float32), times=times, y0=y0, io=io, no=no, NL=NL,
                B=B, H=H, C=C, L=L)


def draw_dW(seed, ts, dt, B, H):
    from oracle import sde_oracle as O
    t0, t1, *_ = O.step_grid(np.asarray(ts, np.float32), dt)
    rng = np.random.default_rng(seed + 1)
    Z = rng.standard_normal((len(t0), B, H)).astype(np.float32)
    return Z * np.sqrt(t1 - t0).astype(np.float32)[:, None, None]


def parity_report(got, ref64, cpu32=None):
    """SURVEY.md 8c criterion.  got: HIP fp32, ref64: fp64 arbiter on identical dW, cpu32: fp32 oracle."""
    got = np.asarray(got, dtype=np.float64)
    err = np.abs(got - ref64)
    rep = dict(mean=float(err.mean()), max=float(err.max()),
               frac_ok=float((err <= 1e-4 + 1e-4 * np.abs(ref64)).mean()))
    if cpu32 is not None:
        e32 = np.abs(np.asarray(cpu32, dtype=np.float64) - ref64)
        rep.update(cpu_mean=float(e32.mean()), cpu_max=float(e32.max()))
    return rep


def assert_parity(got, ref64, cpu32=None, what='', amplifying=False):
    """amplifying=True: for long horizons whose dynamics amplify round-off (fp32 on the CPU is itself far from the
    fp64 arbiter on isolated rows) only the relative criteria are applied: mean and 99.9th-percentile error within
    4x of the CPU-fp32 error on the same inputs."""
    assert np.all(np.isfinite(got)), f'{what}: non-finite output'
    rep = parity_report(got, ref64, cpu32)
    if amplifying:
        assert cpu32 is not None
        e = np.abs(np.asarray(got, np.float64) - ref64)
        e32 = np.abs(np.asarray(cpu32, np.float64) - ref64)
        rep.update(p999=float(np.quantile(e, 0.999)), cpu_p999=float(np.quantile(e32, 0.999)))
        assert rep['mean'] <= 4 * rep['cpu_mean'] + 1e-7, (what, rep)
        assert rep['p999'] <= 4 * rep['cpu_p999'] + 1e-6, (what, rep)
        # ... and SURVEY 8c's ABSOLUTE bounds on the rows where fp32 arithmetic itself stays on the fp64 trajectory.  Which rows
        # leave it is a property of the row, not of the rounding: at the K3 shape (512 rows, 200 GSDE steps) the CPU-fp32 run
        # leaves the tolerance band on 96 rows, a second fp32 run with every increment moved by one ulp on 95, 92 of them the same
        # (a saturating step late in the horizon turns an error of 1e-4 into 0.6; measured with the numpy oracle, round 4).  So:
        # on the rows the CPU-fp32 run keeps inside the band, mean <= 1e-5, at most 3 % of those rows may leave the band in the
        # kernel's run (its own few flips), and on the rows inside the band in both runs max <= 5e-3.
        tol = 1e-4 + 1e-4 * np.abs(ref64)
        other = tuple(i for i in range(e.ndim) if i != e.ndim - 2)
        stable = ~((e32 > tol).any(axis=other))
        got_out = (e > tol).any(axis=other)
        both = stable & ~got_out
        rep.update(stable_rows=int(stable.sum()), rows=int(stable.size), stable_rows_left_by_kernel=int((stable & got_out).sum()),
                   stable_mean=float(np.take(e, np.nonzero(both)[0], axis=e.ndim - 2).mean()) if both.any() else 0.0,
                   stable_max=float(np.take(e, np.nonzero(both)[0], axis=e.ndim - 2).max()) if both.any() else 0.0)
        assert rep['stable_rows'] >= 0.5 * rep['rows'], (what, rep)
        assert rep['stable_rows_left_by_kernel'] <= 0.03 * rep['stable_rows'] + 1, (what, rep)
        assert rep['stable_mean'] <= 1e-5 and rep['stable_max'] <= 5e-3, (what, rep)
        return rep
    assert rep['mean'] <= 1e-5, (what, rep)
    assert rep['frac_ok'] >= 0.9999, (what, rep)
    assert rep['max'] <= 5e-3, (what, rep)
    if cpu32 is not None:
        assert rep['mean'] <= 4 * rep['cpu_mean'] + 1e-7, (what, rep)
        assert rep['max'] <= 4 * rep['cpu_max'] + 1e-6, (what, rep)
    return rep


def grad_close(got, ref, name, tol, tag=''):
    """Gradient check per tensor (round 5): max |err| / max |ref| < tol AND mean |err| / mean |ref| < tol; with SNSDE_GRAD_MARGINS
    set the measured values are appended to that file (the tolerances in the tests are set from such a run)."""
    import torch
    g = got.detach().double().cpu() if torch.is_tensor(got) else torch.as_tensor(got).double()
    r = ref.detach().double().cpu() if torch.is_tensor(ref) else torch.as_tensor(ref).double()
    e = (g - r).abs()
    err = float(e.max()) / (float(r.abs().max()) + 1e-300)
    mean_rel = float(e.mean()) / (float(r.abs().mean()) + 1e-300)
    if os.environ.get('SNSDE_GRAD_MARGINS'):
        with open(os.environ['SNSDE_GRAD_MARGINS'], 'a') as fh:
            fh.write(f'{tag or "-"} - - - - - {name} {err:.3e} {mean_rel:.3e}\n')
    assert err < tol, (tag, name, err)
    assert mean_rel < tol, (tag, name, mean_rel)
