"""CPU tests of the host side: C-ABI surface, parameter layout, time grid, module API, spline
construction and the generic (tensor-op) sdeint loop - all against the oracle / golden vectors.
No GPU compute is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import stable_neural_sdes_amd as S
from oracle import sde_oracle as O
from stable_neural_sdes_amd import _lib
from tests.helpers import group, load, param_spec, params_of, unflatten

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPL, FG, TRAJ, WRAP = load('spline.npz'), load('fg.npz'), load('traj.npz'), load('wrapper.npz')


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, 'include', 'snsde.h')).read()
    declared = set(re.findall(r'\b(snsde_[a-z_]+)\s*\(', header))
    assert declared, 'no declarations parsed'
    assert declared == set(_lib.EXPORTS)
    lib = _lib.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.snsde_version() == 2
    assert lib.snsde_strerror(-7).decode().startswith('ts must')


def test_dynamic_symbol_table_is_the_header():
    """-fvisibility=hidden: the shared library exports the SNSDE_API entry points of include/snsde.h and nothing else (no mangled
    internal launchers / dispatchers)."""
    import subprocess
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = {line.split()[-1] for line in out.splitlines() if line.split() and line.split()[-2] in ('T', 'W', 'D', 'B', 'V', 'R')}
    ours = {n for n in names if 'snsde' in n}
    assert ours == set(_lib.EXPORTS), sorted(ours ^ set(_lib.EXPORTS))
    assert not [n for n in names if n.startswith('_Z')], [n for n in names if n.startswith('_Z')][:5]


def test_batches_beyond_the_32_bit_save_offsets_take_the_generic_kernels():
    """16 B H >= 2^32 exceeds the MFMA kernels' 32-bit uniform offset factors; the plan refuses it, so `auto` reports
    (and takes) the generic family for forward AND backward instead of failing at launch."""
    import ctypes as C
    lib = _lib.lib()

    def q(batch, kernel='auto'):
        s = _lib.Solve()
        s.model = S.engine.model_struct(21, 128, 128, 2, 4, 17)
        s.batch, s.knots, s.n_steps, s.n_out, s.method, s.kernel = batch, 101, 100, 2, _lib.EULER, _lib.KERNELS[kernel]
        return _lib.PATHS[lib.snsde_forward_path(C.byref(s))], lib.snsde_backward_supported(C.byref(s))
    assert q(1024) == ('lean', 1)
    assert q((1 << 21) - 4) == ('mfma16', 1)
    assert q(1 << 21) == ('generic', 2)
    assert q(1 << 21, 'mfma') == ('none', 2)


@pytest.mark.parametrize('pair', [(1, 0), (1, 18), (2, 16), (4, 17), (6, 17)])
def test_make_model_configurations_stay_on_the_mfma_families(pair):
    """The five (input_option, noise_option) pairs the reference's make_model builds (benchmark_classification/common_sde.py:301-342)
    x {euler, srk, milstein} x H in {16, 32, 64, 128} x the benchmark batch sizes x narrow / sepsis-wide control paths: the forward
    takes an MFMA kernel family and the backward is mode 1 (MFMA adjoint + native parameter pass).  Host-side queries of the
    library (snsde_forward_path / snsde_backward_supported), so a silent fall to the generic kernels fails CI on every round, GPU
    or not.  The one known hole is listed: Milstein through a TWO-layer diffusion net at H = 128 (its four net matrices + the drift
    exceed registers + LDS; the reference never uses Milstein: SURVEY 0.4)."""
    io, no = pair
    known_generic = {(1, 18, 'milstein', 128)}
    for method in ('euler', 'srk', 'milstein'):
        for H in (16, 32, 64, 128):
            for B in (256, 1024, 4096):
                for C_, L in ((21, 101), (69, 72)):
                    model = S.engine.model_struct(C_, H, H, 2, io, no)
                    path = S.engine.forward_path(model, B, L, L - 1, method=method)
                    grid = S.engine.step_grid(np.array([0.0, L - 1.0], np.float32), 1.0, np.arange(L, dtype=np.float32), None)
                    mode = S.engine.backward_mode(model, B, L, grid, method)
                    if (io, no, method, H) in known_generic:
                        assert (path, mode) == ('generic', 2), (io, no, method, H, B, C_, path, mode)
                    else:
                        assert path in ('lean', 'mfma4', 'mfma16', 'mfma-srk', 'w4') and mode == 1, (io, no, method, H, B, C_, path, mode)


def test_save_layout_reports_the_delta_free_adjoints():
    """snsde_save_layout (host-side query, include/snsde.h): delta_slots == 0 exactly where the wave-group adjoints accumulate the
    weight gradients themselves - H = 64 with a diffusion net, Euler and SRK, up to 6144 rows, `auto` / `w4`, host or
    device-resident Philox key; everything else keeps act_slots (+ the Milstein tangent factors) delta planes."""
    import ctypes as C
    lib = _lib.lib()

    def layout(io, no, H, B, method, kernel='auto', seed_dev=False, NL=2):
        s = _lib.Solve()
        s.model = S.engine.model_struct(5, H, H, NL, io, no)
        s.batch, s.knots, s.n_steps, s.n_out = B, 9, 8, 3
        s.method = {'euler': _lib.EULER, 'milstein': _lib.MILSTEIN, 'srk': _lib.SRK}[method]
        s.kernel = _lib.KERNELS[kernel]
        if seed_dev:
            s.seed_dev = C.c_void_p(64)      # (never dereferenced by the query)
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        assert lib.snsde_save_layout(C.byref(s), C.byref(a), C.byref(b), C.byref(c)) == 0
        return a.value, b.value, c.value
    for io, no in ((3, 18), (1, 14), (5, 19), (3, 15)):
        nn = 2 if no >= 18 else 1
        for method in ('euler', 'srk'):
            slots, planes, dslots = layout(io, no, 64, 2048, method)
            assert slots == 3 + nn * (2 if method == 'srk' else 1) and planes == (3 if method == 'srk' else 1) and dslots == 0
            assert layout(io, no, 64, 2048, method, 'w4')[2] == 0 and layout(io, no, 64, 37, method)[2] == 0
            assert layout(io, no, 64, 2048, method, seed_dev=True)[2] == 0
            assert layout(io, no, 64, 2048, method, 'mfma4')[2] == slots            # the tile adjoint + weight-gradient GEMMs
            assert layout(io, no, 64, 8192, method)[2] == slots                      # large batches stay on the tile adjoints
            assert layout(io, no, 128, 1024, method)[2] > 0 and layout(io, no, 64, 2048, method, NL=3)[2] > 0
        slots, _, dslots = layout(io, no, 64, 2048, 'milstein')
        assert dslots == slots + (3 if nn == 2 else 1)
    assert layout(4, 17, 128, 1024, 'euler')[2] == 3                                  # K2 (NL + 1 slots): no diffusion net, no fused gradients


def test_neural_sde_func_shapes_train_on_the_fused_path():
    """The NeuralSDEFunc mapping of fields.py ((3, 18) with the variant switches: smooth activation, linear drift output, the net's
    linear output as the diffusion, raw time) has a fused backward (mode 1) under Euler, SRK and Milstein at every instantiated width and
    one to three drift layers; under SRK the forward saves one more slot than under Euler for every pre-activation set it needs later
    (snsde_save_layout)."""
    import ctypes as C
    from stable_neural_sdes_amd import fields
    lib = _lib.lib()
    times = np.linspace(0, 1, 9).astype(np.float32)
    grid = S.engine.StepGrid(times, 0.125, times, None)
    for H in (16, 32, 64, 128):
        for NL in (1, 2, 3):
            for act in (0, 1, 2):
                m = S.engine.model_struct(3, H, H, NL, 3, 18, activation=act, drift_output=fields.DRIFT_LINEAR,
                                          diffusion_output=fields.DIFFUSION_RAW_NET, time_feature=fields.TIME_RAW)
                assert S.engine.backward_mode(m, 19, 9, grid, 'euler') == 1 and S.engine.backward_mode(m, 19, 9, grid, 'srk') == 1, (H, NL, act)
                # (Milstein through a two-layer net at H = 128 with more than one drift layer has no MFMA forward: four net matrices +
                #  the drift's exceed registers + LDS; such fields take the graph-replayed stepper)
                assert S.engine.backward_mode(m, 19, 9, grid, 'milstein') == (0 if H == 128 and NL >= 2 else 1), (H, NL, act)
                s = _lib.Solve()
                s.model = m
                s.batch, s.knots, s.n_steps, s.n_out = 19, 9, grid.N, grid.T
                s.method = _lib.SRK
                a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
                assert lib.snsde_save_layout(C.byref(s), C.byref(a), C.byref(b), C.byref(c)) == 0
                assert a.value == NL + 1 + 4 + (NL + 2 if act else 0) and b.value == 3, (H, NL, act, a.value)


def test_stale_binding_is_refused():
    """A descriptor whose struct_size is not the library's sizeof (a binding compiled against an older header) is refused with
    SNSDE_ERR_ABI before any field is read; snsde_abi_check verifies a binding's sizes at load time."""
    import ctypes as C
    lib = _lib.lib()
    s = _lib.Solve()
    assert s.struct_size == C.sizeof(_lib.Solve)
    s.model = S.engine.model_struct(3, 8, 8, 2, 4, 17)
    s.batch, s.knots, s.n_steps, s.n_out = 4, 5, 3, 2
    a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
    assert lib.snsde_save_layout(C.byref(s), C.byref(a), C.byref(b), C.byref(c)) == 0
    assert lib.snsde_workspace_bytes(C.byref(s)) > 0
    s.struct_size -= 8              # as if built before the last pointer field was added
    assert lib.snsde_save_layout(C.byref(s), C.byref(a), C.byref(b), C.byref(c)) == -10
    assert lib.snsde_workspace_bytes(C.byref(s)) == 0
    assert lib.snsde_solve_forward(C.byref(s), None) == -10
    assert lib.snsde_forward_path(C.byref(s)) == 0 and lib.snsde_backward_supported(C.byref(s)) == 0
    bw = _lib.Backward()
    bw.fwd = s
    assert lib.snsde_solve_backward(C.byref(bw), None) == -10
    bw.struct_size += 4
    assert lib.snsde_solve_backward(C.byref(bw), None) == -10
    h = _lib.Head()
    h.struct_size = 8
    assert lib.snsde_readout_head(C.byref(h), None) == -10
    sizes = (C.sizeof(_lib.Model), C.sizeof(_lib.Solve), C.sizeof(_lib.Backward), C.sizeof(_lib.Head))
    assert lib.snsde_abi_check(2, *sizes) == 0
    assert lib.snsde_abi_check(1, *sizes) == -10
    assert lib.snsde_abi_check(2, sizes[0], sizes[1] - 8, sizes[2], sizes[3]) == -10
    assert b'struct_size' in lib.snsde_strerror(-10)


@pytest.mark.parametrize('io', range(7))
@pytest.mark.parametrize('no', range(20))
def test_param_layout_matches_reference_state_dict(io, no):
    for NL, C_, H in ((1, 3, 8), (3, 5, 12)):
        m = S.engine.model_struct(C_, H, H, NL, io, no)
        layout, numel = _lib.param_layout(m)
        spec = param_spec(io, no, NL, C_, H)
        assert [(n, s) for n, _, s in layout] == [(n, s if len(s) == 2 else s) for n, s in spec]
        off = 0
        for (name, o, shape) in layout:
            assert o == off, name
            off += int(np.prod(shape))
        assert off == numel
        mod = S.Diffusion_model(C_, H, H, NL, input_option=io, noise_option=no)
        assert [(k, tuple(v.shape)) for k, v in mod.state_dict().items()] == spec


def test_model_validation_errors():
    lib = _lib.lib()
    assert lib.snsde_param_count(C.byref(S.engine.model_struct(3, 8, 8, 2, 7, 0))) == -3
    assert lib.snsde_param_count(C.byref(S.engine.model_struct(3, 8, 8, 2, 0, 20))) == -3
    assert lib.snsde_param_count(C.byref(S.engine.model_struct(0, 8, 8, 2, 0, 0))) == -2
    assert lib.snsde_param_count(C.byref(S.engine.model_struct(3, 8, 6, 2, 2, 0))) == -2   # emb needs HH == H
    assert lib.snsde_param_count(C.byref(S.engine.model_struct(3, 8, 6, 2, 1, 0))) > 0
    with pytest.raises(ValueError):
        S.Diffusion_model(3, 8, 8, 2, input_option=9)
    with pytest.raises(ValueError):
        S.Diffusion_model(3, 8, 8, 2, noise_option=20)


GRID_CASES = [
    (np.array([0., 100.]), 1.0, np.arange(101.)),
    (np.array([1., 4., 6., 8.]), 1.0, np.linspace(1, 8, 8)),
    (np.linspace(0, 1, 20), 0.05, np.linspace(0, 1, 20)),
    (np.linspace(0, 1, 20), 0.02, np.linspace(0, 1, 20)),
    (np.linspace(0, 1, 12), float(np.float32(1 / 11)), np.linspace(0, 1, 12)),
    (np.array([0., 0.3, 0.6, 2.0]), 1.0, np.array([0., 0.5, 1.0, 2.0])),
    (np.array([0., 7.]), 0.5, np.arange(8.)),
]


@pytest.mark.parametrize('ci', range(len(GRID_CASES)))
def test_grid_build_bit_exact_vs_oracle(ci):
    ts, dt, times = GRID_CASES[ci]
    ts32, times32 = ts.astype(np.float32), times.astype(np.float32)
    g = S.engine.StepGrid(ts32, dt, times32, None)
    t0, t1, out_step, w0, w1 = O.step_grid(ts32, dt)
    assert g.N == len(t0)
    np.testing.assert_array_equal(g.step_tab[:, 0], t0)
    np.testing.assert_array_equal(g.step_tab[:, 7], t1)
    np.testing.assert_array_equal(g.step_tab[:, 1], t1 - t0)
    np.testing.assert_array_equal(g.out_step, out_step)
    np.testing.assert_array_equal(g.out_w[:, 0], w0)
    np.testing.assert_array_equal(g.out_w[:, 1], w1)
    idx = g.step_tab[:, 5].view(np.int32)
    for n in range(g.N):
        i, frac = O.spline_index(times32, t0[n])
        assert idx[n] == i and g.step_tab[n, 4] == frac
    np.testing.assert_allclose(g.step_tab[:, 2], np.sin(t0.astype(np.float64)), atol=1e-7)
    np.testing.assert_allclose(g.step_tab[:, 3], np.cos(t0.astype(np.float64)), atol=1e-7)
    np.testing.assert_array_equal(g.step_tab[:, 6], np.sqrt(t1 - t0))


def test_grid_errors():
    with pytest.raises(ValueError):
        S.engine.StepGrid(np.array([0., 0.]), 1.0, np.array([0., 1.]), None)
    with pytest.raises(ValueError):
        S.engine.StepGrid(np.array([1., 0.5]), 1.0, np.array([0., 1.]), None)
    with pytest.raises(ValueError):
        S.engine.StepGrid(np.array([0., 1.]), -1.0, np.array([0., 1.]), None)
    with pytest.raises(ValueError):
        S.engine.StepGrid(np.array([1e8, 1e8 + 64]), 1.0, np.array([0., 1.]), None)
    with pytest.raises(ValueError):
        S.engine.StepGrid(np.array([0.]), 1.0, np.array([0., 1.]), None)


G1_CASES = sorted({k.split('/')[1] for k in SPL.files if k.startswith('G1/')})


@pytest.mark.parametrize('case', G1_CASES)
@pytest.mark.parametrize('prec', ['f32', 'f64'])
def test_natural_spline_coeffs_vs_golden(case, prec):
    g = group(SPL, f'G1/{case}/{prec}')
    out = S.controldiffeq.natural_cubic_spline_coeffs(torch.from_numpy(g['times']), torch.from_numpy(g['X']))
    tol = dict(rtol=5e-5, atol=5e-5) if prec == 'f32' else dict(rtol=1e-10, atol=1e-10)
    for got, name in zip(out, ('a', 'b', 'two_c', 'three_d')):
        assert got.shape == g[name].shape
        np.testing.assert_allclose(got.numpy(), g[name], err_msg=name, **tol)


@pytest.mark.parametrize('case', G1_CASES)
def test_spline_evaluate_cpu_vs_golden(case):
    c = group(SPL, f'G1/{case}/f32')
    g = group(SPL, f'G2/{case}/f32')
    coeffs = tuple(torch.from_numpy(c[n]) for n in ('a', 'b', 'two_c', 'three_d'))
    sp = S.controldiffeq.NaturalCubicSpline(torch.from_numpy(c['times']), coeffs)
    cs = S.torchcde.CubicSpline(torch.cat(coeffs, dim=-1), torch.from_numpy(c['times']))
    for i, t in enumerate(g['t']):
        np.testing.assert_allclose(sp.evaluate(torch.tensor(t)).numpy(), g['evaluate'][i], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(cs.derivative(torch.tensor(t)).numpy(), g['derivative'][i], rtol=1e-6, atol=1e-6)


def test_natural_spline_input_errors():
    t = torch.linspace(0, 1, 5)
    X = torch.randn(2, 5, 3)
    with pytest.raises(ValueError):
        S.controldiffeq.natural_cubic_spline_coeffs(t.long(), X)
    with pytest.raises(ValueError):
        S.controldiffeq.natural_cubic_spline_coeffs(t.flip(0), X)
    with pytest.raises(ValueError):
        S.controldiffeq.natural_cubic_spline_coeffs(t[:4], X)
    with pytest.raises(ValueError):
        S.controldiffeq.natural_cubic_spline_coeffs(t[:1], X[:, :1])


def test_hermite_vs_oracle():
    rng = np.random.default_rng(3)
    t = np.cumsum(rng.uniform(0.2, 1.0, 9))
    X = rng.standard_normal((4, 9, 3)).cumsum(1)
    X[0, 2, 1] = X[0, 3, 1] = np.nan
    X[1, 0, 0] = np.nan
    X[2, 8, 2] = np.nan
    X[3, :, 0] = np.nan
    got = S.torchcde.hermite_cubic_coefficients_with_backward_differences(torch.from_numpy(X), torch.from_numpy(t))
    exp = O.hermite_cubic_coefficients_with_backward_differences(X, t)
    np.testing.assert_allclose(got.numpy(), exp, rtol=1e-12, atol=1e-12)


MODELS = FG['G3/models']


@pytest.mark.parametrize('mi', range(0, len(MODELS), 3))
def test_module_fg_cpu_vs_golden(mi):
    io, no, NL = (int(v) for v in MODELS[mi])
    coeffs, times, y, tv = FG['G3/coeffs'], FG['G3/times'], FG['G3/y'], FG['G3/t']
    B, H = y.shape
    C_ = coeffs.shape[-1] // 4
    off = FG['G3/params_off']
    p = unflatten(FG['G3/params_flat'][off[mi]:off[mi + 1]], param_spec(io, no, NL, C_, H))
    m = S.Diffusion_model(C_, H, H, NL, input_option=io, noise_option=no)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in p.items()})
    m.set_X(torch.from_numpy(coeffs), torch.from_numpy(times))
    exp = FG['G3/out32'][mi]
    with torch.no_grad():
        for ti, t in enumerate(tv):
            f = m.f(torch.tensor(np.float32(t)), torch.from_numpy(y))
            g = m.g(torch.tensor(np.float32(t)), torch.from_numpy(y))
            np.testing.assert_allclose(f.numpy(), exp[0, ti], rtol=1e-5, atol=2e-6)
            np.testing.assert_allclose(g.numpy(), exp[1, ti], rtol=1e-5, atol=2e-6)


class _Recorder:
    def __init__(self):
        self.calls = []

    def __call__(self, sde, y0, ts, dt, **kw):
        self.calls.append(dict(ts=ts.clone(), dt=dt, kw=kw, y0=y0.clone()))
        return torch.stack([y0 + ts[k] for k in range(ts.shape[0])])


@pytest.mark.parametrize('case', ['int_grid', 'no_ends', 'all_last', 'lin01'])
def test_neuralsde_wrapper_bookkeeping_vs_golden(case, monkeypatch):
    g = group(WRAP, f'G4/{case}')
    times, fi, coeffs = (torch.from_numpy(g[k]) for k in ('times', 'final_index', 'coeffs'))
    C_, H = coeffs.shape[-1] // 4, g['y0'].shape[1]
    func = S.Diffusion_model(C_, H, H, 2, input_option=4, noise_option=17)
    model = S.NeuralSDE(func, C_, H, 2, initial=True).eval()
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params_of(WRAP, f'G4/{case}').items()})
    rec = _Recorder()
    monkeypatch.setattr(S.modules._torchsde, 'sdeint', rec)
    with torch.no_grad():
        pred = model(times, [coeffs], fi)
    call = rec.calls[0]
    np.testing.assert_array_equal(call['ts'].numpy(), g['ts'])
    assert call['dt'] == float(g['dt']) and call['kw']['method'] == 'euler'
    assert call['kw']['options']['dt'] == float(g['options_dt'])
    np.testing.assert_allclose(call['y0'].numpy(), g['y0'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pred.numpy(), g['pred'], rtol=1e-4, atol=1e-5)


def test_forecasting_and_ists_wrappers_vs_golden(monkeypatch):
    rec = _Recorder()
    monkeypatch.setattr(S.modules._torchsde, 'sdeint', rec)
    g = group(WRAP, 'G4/forecast')
    times = torch.from_numpy(g['times'])
    cs = [torch.from_numpy(g[k]) for k in ('a', 'b', 'two_c', 'three_d')]
    C_ = cs[0].shape[-1]
    sd = params_of(WRAP, 'G4/forecast')
    H = sd['initial_network.weight'].shape[0]
    func = S.Diffusion_model(C_, H, H, 2, input_option=2, noise_option=16)
    model = S.NeuralSDE_forecasting(func, C_, 3, H, 5, initial=True).eval()
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    with torch.no_grad():
        pred = model(times, cs, torch.zeros(cs[0].shape[0], dtype=torch.long))
    np.testing.assert_array_equal(rec.calls[-1]['ts'].numpy(), g['ts'])
    assert rec.calls[-1]['dt'] == float(g['dt'])
    np.testing.assert_allclose(pred.numpy(), g['pred'], rtol=1e-4, atol=1e-5)

    g = group(WRAP, 'G4/ists')
    times, coeffs = torch.from_numpy(g['times']), torch.from_numpy(g['coeffs'])
    sd = params_of(WRAP, 'G4/ists')
    H = sd['initial_network.weight'].shape[0]
    func = S.Diffusion_model(coeffs.shape[-1] // 4, H, H, 2, input_option=6, noise_option=17)
    model = S.IstsNeuralSDE(func, coeffs.shape[-1] // 4, H, 2, initial=True).eval()
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    with torch.no_grad():
        pred, z = model(coeffs, times)
    assert rec.calls[-1]['kw']['method'] == str(g['method']) == 'srk'
    assert rec.calls[-1]['dt'] == float(g['dt'])
    np.testing.assert_array_equal(rec.calls[-1]['ts'].numpy(), g['ts'])
    np.testing.assert_allclose(z.numpy(), g['z'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pred.numpy(), g['pred'], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('name', ['arange', 'lin01_20', 'tiny', 'irr'])
def test_prepare_solver_kwargs_dt(name):
    g = group(WRAP, f'G4/dt/{name}')
    kw, dt = S.prepare_sde_solver_kwargs(torch.from_numpy(g['times']), {}, default_method='euler',
                                         respect_euler_grid=False)
    assert dt == float(g['dt']) and kw == {'method': 'euler', 'options': {'dt': dt}}
    kw, _ = S.prepare_sde_solver_kwargs(torch.from_numpy(g['times']), {'options': {'step_size': 0.1}},
                                        default_method='euler', respect_euler_grid=True)
    assert 'dt' not in kw['options']


class _ReplayBM:
    def __init__(self, dW):
        self.dW, self.n = dW, 0

    def __call__(self, ta, tb):
        out = self.dW[self.n]
        self.n += 1
        return out


G5_CASES = sorted({k.split('/')[1] for k in TRAJ.files if k.startswith('G5/')})


@pytest.mark.parametrize('case', G5_CASES)
def test_sdeint_tensor_loop_cpu_vs_golden(case):
    g = group(TRAJ, f'G5/{case}')
    io, no, NL = (int(v) for v in g['io_no_nl'])
    coeffs = torch.from_numpy(g['coeffs'])
    C_, H = coeffs.shape[-1] // 4, g['y0'].shape[1]
    m = S.Diffusion_model(C_, H, H, NL, input_option=io, noise_option=no)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params_of(TRAJ, f'G5/{case}').items()})
    m.set_X(coeffs, torch.from_numpy(g['times']))
    with torch.no_grad():
        ys = S.sdeint(m, torch.from_numpy(g['y0']), torch.from_numpy(g['ts']), bm=_ReplayBM(torch.from_numpy(g['dW'])),
                      method=str(g['method']), dt=float(g['dt']))
    scale = 1.0 + np.abs(g['ys64'])
    err_ref = np.max(np.abs(g['ys32'] - g['ys64']) / scale)
    err = np.max(np.abs(ys.numpy() - g['ys64']) / scale)
    assert ys.shape == g['ys64'].shape
    assert err < max(4 * err_ref, 2e-6), (err, err_ref)


def test_sdeint_argument_errors():
    m = S.Diffusion_model(3, 8, 8, 2, input_option=4, noise_option=17)
    m.set_X(torch.zeros(2, 4, 12), torch.arange(5.))
    y0 = torch.zeros(2, 8)
    with pytest.raises(ValueError):
        S.sdeint(m, torch.zeros(8), torch.tensor([0., 1.]), method='euler')
    with pytest.raises(ValueError):
        S.sdeint(m, y0, torch.tensor([1., 0.]), method='euler', dt=1.0)
    with pytest.raises(ValueError):
        S.sdeint(m, y0, torch.tensor([0., 1.]), method='rk4', dt=1.0)
    with pytest.raises(ValueError):
        S.sdeint(m, y0, torch.tensor([0., 1.]), method='euler', dt=0.0)
    with pytest.raises(NotImplementedError):
        S.sdeint(m, y0, torch.tensor([0., 1.]), method='euler', dt=1.0, adaptive=True)
    with torch.no_grad():                                  # default method = srk (torchsde's default for Ito/diagonal)
        assert S.sdeint(m, y0, torch.tensor([0., 1.]), dt=0.5).shape == (2, 2, 8)
    with pytest.raises(ValueError):   # HIP engine requested on CPU tensors: loud, no fallback
        with torch.no_grad():
            S.sdeint(m, y0, torch.tensor([0., 1.]), method='euler', dt=1.0, options={'backend': 'hip'})


def test_install_registers_shims():
    import sys
    saved = {k: sys.modules.pop(k, None) for k in ('torchsde', 'torchcde', 'controldiffeq')}
    try:
        S.install()
        import torchcde
        import torchsde
        assert torchsde.sdeint is S.sdeint
        assert torchcde.CubicSpline is S.torchcde.CubicSpline
    finally:
        for k, v in saved.items():
            sys.modules.pop(k, None)
            if v is not None:
                sys.modules[k] = v


class _ReplayBMU:
    """bm(ta, tb, return_U=True) -> (dW_n, dU_n) replayed from arrays (SRK)."""
    levy_area_approximation = 'space-time'

    def __init__(self, dW, dU):
        self.dW, self.dU, self.n = dW, dU, 0

    def __call__(self, ta, tb, return_U=False):
        out = (self.dW[self.n], self.dU[self.n]) if return_U else self.dW[self.n]
        self.n += 1
        return out


@pytest.mark.parametrize('io,no', [(4, 17), (6, 17), (2, 16), (1, 18), (3, 8), (0, 5)])
def test_sdeint_srk_tensor_loop_vs_oracle(io, no):
    from tests.helpers import draw_dW, make_problem
    pr = make_problem(60 + io, io, no, 2, 5, 8, 3, 9)
    ts, dt = np.array([0., 2.5, 8.], np.float32), 0.5
    dW = draw_dW(60, ts, dt, 5, 8)
    dU = (0.5 * dt * dW + dt * np.sqrt(dt / 12) * np.random.default_rng(1).standard_normal(dW.shape)).astype(np.float32)
    m = S.Diffusion_model(3, 8, 8, 2, input_option=io, noise_option=no)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()})
    m.set_X(torch.from_numpy(pr['coeffs']), torch.from_numpy(pr['times']))
    with torch.no_grad():
        ys = S.sdeint(m, torch.from_numpy(pr['y0']), torch.from_numpy(ts), bm=_ReplayBMU(torch.from_numpy(dW), torch.from_numpy(dU)),
                      method='srk', dt=dt)
    ref, _ = O.solve_diffusion_model(pr['params'], io, no, pr['coeffs'], pr['times'], pr['y0'], ts, dt, dW, method='srk',
                                     dtype=np.float64, dU=dU)
    np.testing.assert_allclose(ys.numpy(), ref, rtol=2e-4, atol=2e-5)


# ---- host logic added with the fused training path -------------------------------------------------------
def _small_model(io=4, no=17, B=6, H=8, C_=3, L=7, seed=0):
    torch.manual_seed(seed)
    m = S.Diffusion_model(C_, H, H, 2, input_option=io, noise_option=no)
    times = torch.arange(L, dtype=torch.float32)
    m.set_X(torch.randn(B, L - 1, 4 * C_) * 0.1, times)
    return m, times, torch.randn(B, H)


def test_row_out_option_of_the_tensor_loop_equals_gather():
    """options['row_out'] has the same contract on every backend: (B, H), row b = the state at ts[row_out[b]]."""
    m, times, y0 = _small_model()
    ts = torch.tensor([0., 1.5, 3., 6.])
    slot = torch.tensor([0, 3, 1, 2, 3, 1])
    dW = torch.randn(6, 6, 8)
    with torch.no_grad():
        full = S.sdeint(m, y0, ts, bm=_ReplayBM(dW), method='euler', dt=1.0)
        sel = S.sdeint(m, y0, ts, bm=_ReplayBM(dW), method='euler', dt=1.0, options={'row_out': slot})
    assert sel.shape == (6, 8)
    assert torch.equal(sel, full.gather(0, slot.reshape(1, -1, 1).expand(1, 6, 8)).squeeze(0))


def test_neuralsde_forward_on_cpu_keeps_the_reference_output_time_selection():
    """CPU tensors take the reference's unique(final_index) route (neuralsde.py:91-116), CUDA tensors the fused
    row_out route; both must return the same rows (checked on GPU in test_gpu_parity)."""
    torch.manual_seed(1)
    model, field = S.make_sde_model('neurallnsde', 3, 2, 8, 8, 2, initial=True)
    model.eval()
    times = torch.arange(7, dtype=torch.float32)
    coeffs = torch.randn(5, 6, 12) * 0.1
    fi = torch.tensor([6, 2, 2, 0, 4])
    ts, slot = model.output_times(times, fi)
    assert ts.tolist() == [0.0, 2.0, 4.0, 6.0] and slot.tolist() == [3, 1, 1, 0, 2]
    with torch.no_grad():
        out = model(times, [coeffs], fi, options={'seed': 3})
    assert out.shape == (5, 2) and torch.isfinite(out).all()


def test_flatten_params_arena_tracks_in_place_updates_and_reallocation():
    from stable_neural_sdes_amd import engine
    m, _, _ = _small_model()
    model, layout, numel = engine.recognise(m)
    dev = torch.device('cpu')
    flat = engine.flatten_params(m, layout, numel, dev)
    assert flat.numel() == numel
    assert engine.flatten_params(m, layout, numel, dev) is flat          # no new buffer
    with torch.no_grad():
        m.linear_out.weight.add_(1.0)                                     # optimizer-style in-place step
    off = {n: o for n, o, _ in layout}['linear_out.weight']
    assert torch.equal(flat[off:off + m.linear_out.weight.numel()].view_as(m.linear_out.weight), m.linear_out.weight)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.linear_out.weight = torch.nn.Parameter(torch.zeros_like(m.linear_out.weight))   # re-allocated parameter
    flat2 = engine.flatten_params(m, layout, numel, dev)
    assert flat2 is not flat and float(flat2[off:off + 4].abs().sum()) == 0.0
    m.load_state_dict(sd)                                                 # copy_ into the (new) arena
    assert torch.equal(engine.flatten_params(m, layout, numel, dev)[off:off + 4], sd['linear_out.weight'].reshape(-1)[:4])


def test_host_time_cache_scalar_lookup():
    from stable_neural_sdes_amd.controldiffeq import _HostTimes
    t = torch.linspace(0, 1, 6)
    assert _HostTimes.scalar(t[3]) == float(t[3]) and _HostTimes.scalar(0.25) == 0.25
    assert np.array_equal(_HostTimes.get(t), t.numpy())


def test_srk_rows_step_equals_the_loop_step():
    """The batched SRID2 step used by the SRK parameter pass == the per-step scheme of the tensor-op loop."""
    from stable_neural_sdes_amd import engine, torchsde as T
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip('libsnsde.so not built')
    m, times, y0 = _small_model(io=4, no=17)
    m = m.double()
    m.set_X(m.coeffs.double(), times)
    B, H = y0.shape
    grid = engine.StepGrid(np.array([0., 6.], np.float32), 1.0, times.numpy(), None)
    grid.device = torch.device('cpu')
    tab = np.zeros((grid.N, 4, _lib.SNSDE_SRK_STRIDE), dtype=np.float32)
    _lib.check(_lib.lib().snsde_grid_srk_build(grid.step_tab.ctypes.data, grid.N, grid._times32.ctypes.data,
                                               grid._times32.shape[0], tab.ctypes.data), 'snsde_grid_srk_build')
    grid._d_srk = torch.from_numpy(tab)
    P = dict(m.named_parameters())
    n = 2
    Y = torch.randn(B, H, dtype=torch.float64)
    I_k = torch.randn(B, H, dtype=torch.float64)
    I_k0 = 0.5 * I_k + 0.1 * torch.randn(B, H, dtype=torch.float64)
    h = torch.ones(B, 1, dtype=torch.float64)
    with torch.no_grad():
        got = T._srk_rows(P, 4, 17, grid, n, n + 1, B, Y, I_k, I_k0, m.coeffs, h)
        want = T._srk_step(m.f, m.g, torch.tensor(float(grid.t0[n]), dtype=torch.float64), torch.tensor(1.0, dtype=torch.float64),
                           Y, I_k, I_k0)
    assert torch.allclose(got, want, rtol=1e-6, atol=1e-7)


def test_ctypes_structs_follow_the_header_field_order():
    """_lib.Solve / _lib.Backward and the stub printed in INTEGRATION.md must list the fields of the C structs in order."""
    hdr = open(os.path.join(ROOT, 'include', 'snsde.h')).read()

    def fields(struct):
        body = re.search(r'typedef struct %s \{(.*?)\} %s;' % (struct, struct), hdr, re.S).group(1)
        body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
        return [re.search(r'(\w+)\s*$', line.strip().rstrip(';')).group(1) for line in body.split('\n') if line.strip().endswith(';')]

    assert fields('snsde_solve') == [n for n, _ in _lib.Solve._fields_]
    assert fields('snsde_backward') == [n for n, _ in _lib.Backward._fields_]
    assert fields('snsde_model') == [n for n, _ in _lib.Model._fields_]


def integration_stub_namespace():
    """The python block INTEGRATION.md section 3 prints for a maintainer of the reference, executed as written (it loads
    libsnsde.so and runs snsde_abi_check on its own struct declarations)."""
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    block = re.search(r'```python\n(import ctypes as C, os, numpy as np, torch\n.*?)```', doc, re.S).group(1)
    ns = {'__name__': 'integration_stub'}
    cwd = os.getcwd()
    os.chdir(ROOT)                      # the stub names the library relative to the repository root
    try:
        exec(compile(block, 'INTEGRATION.md#stub', 'exec'), ns)
    finally:
        os.chdir(cwd)
    return ns


def test_integration_md_stub_declares_the_library_structs_field_for_field():
    """Round 4's stub lacked the four trailing snsde_solve fields and got SNSDE_ERR_ABI from the library; the test then only
    compared the pointer run.  Now: the stub is executed (its own snsde_abi_check must pass against the built library) and
    its WHOLE field lists, types, offsets and sizes are compared with the binding the package uses."""
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip('libsnsde.so not built')
    ns = integration_stub_namespace()
    for name in ('Model', 'Solve'):
        stub, own = ns[name], getattr(_lib, name)
        assert C.sizeof(stub) == C.sizeof(own), name

        def flat(struct):
            return [(n, t if not issubclass(t, C.Structure) else 'struct', getattr(struct, n).offset, getattr(struct, n).size)
                    for n, t in struct._fields_]
        assert flat(stub) == flat(own), name
    assert ns['SNSDE_VERSION'] == _lib.ABI_VERSION == _lib.lib().snsde_version()
    L = _lib.lib()
    assert L.snsde_abi_check(2, C.sizeof(ns['Model']), C.sizeof(ns['Solve']), 0, 0) == 0
    assert L.snsde_abi_check(2, C.sizeof(ns['Model']), C.sizeof(ns['Solve']) - 16, 0, 0) == -10      # round 4's stale stub
    assert L.snsde_abi_check(1, C.sizeof(ns['Model']), C.sizeof(ns['Solve']), 0, 0) == -10
    assert L.snsde_abi_check(2, 0, 0, C.sizeof(_lib.Backward) + 8, 0) == -10


def test_product_library_reads_no_environment():
    """The split heuristic's sweep knobs live in -DSNSDE_DEV_TUNING builds only (build.py devtuning)."""
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip('libsnsde.so not built')
    import subprocess
    und = subprocess.run(['nm', '-D', '--undefined-only', _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert 'getenv' not in und


# ---- torchsde / controldiffeq mirror: the names the reference's other call sites need -----------------------------------
def test_torchsde_mirror_exports_the_names_the_reference_imports():
    """torch-ists .../NSDE/latent_sde.py:31,134-141: `class LatentSDE(torchsde.SDEIto)`, `torchsde.sdeint_adjoint(...,
    names={'drift': 'f_aug', 'diffusion': 'g_aug'})`; torchsde.BrownianInterval is what user code passes as bm."""
    T = S.torchsde

    class Lat(T.SDEIto):
        def __init__(self):
            super().__init__(noise_type='diagonal')
            self.lin = torch.nn.Linear(3, 3)

        def f_aug(self, t, y):
            return self.lin(y).tanh()

        def g_aug(self, t, y):
            return 0.1 * torch.ones_like(y)

    m = Lat()
    assert m.sde_type == 'ito' and m.noise_type == 'diagonal' and isinstance(m, torch.nn.Module)
    assert T.SDEStratonovich('diagonal').sde_type == 'stratonovich'
    with pytest.raises(ValueError):
        T.SDEIto('banana')
    y0 = torch.zeros(4, 3, requires_grad=True)
    ys = T.sdeint_adjoint(m, y0, torch.tensor([0.0, 0.5, 1.0]), dt=0.25, method='euler',
                          names={'drift': 'f_aug', 'diffusion': 'g_aug'}, adjoint_method='euler')
    assert ys.shape == (3, 4, 3)
    ys.sum().backward()
    assert m.lin.weight.grad is not None and y0.grad is not None
    bm = T.BrownianInterval(t0=0.0, t1=1.0, size=(4, 3), entropy=3)
    assert bm(0.0, 0.25).shape == (4, 3)
    # the default name mapping keeps a Diffusion_model on whatever path plain sdeint takes (same numbers)
    pr_ts = torch.tensor([0.0, 1.0, 2.0])
    d = S.Diffusion_model(2, 8, 8, 2, input_option=4, noise_option=17)
    d.set_X(torch.zeros(3, 4, 8), torch.arange(5.0))
    a = T.sdeint(d, torch.ones(3, 8), pr_ts, dt=1.0, method='euler', options={'seed': 1})
    b = T.sdeint(d, torch.ones(3, 8), pr_ts, dt=1.0, method='euler', options={'seed': 1}, names={'drift': 'f', 'diffusion': 'g'})
    assert torch.equal(a, b)


def test_controldiffeq_mirror_delegates_unknown_names_to_a_vendored_package(tmp_path, monkeypatch):
    """After install() `import controldiffeq` resolves to the mirror; the reference's CDE baselines built by the same
    common_sde.make_model call controldiffeq.cdeint (benchmark_classification/models_sde/metamodel.py), which the mirror
    serves from the vendored package when one is importable."""
    import sys
    from stable_neural_sdes_amd import controldiffeq as M
    pkg = tmp_path / 'controldiffeq'
    pkg.mkdir()
    (pkg / '__init__.py').write_text('from .solver import cdeint\nMARK = "vendored"\n')
    (pkg / 'solver.py').write_text('def cdeint(*a, **k):\n    return "vendored-cdeint"\n')
    monkeypatch.syspath_prepend(str(tmp_path))
    monkeypatch.setattr(M, '_VENDORED', None)
    try:
        assert M.cdeint() == 'vendored-cdeint' and M.MARK == 'vendored'
        assert M.natural_cubic_spline_coeffs.__module__.endswith('controldiffeq')      # own names stay the mirror's
        with pytest.raises(AttributeError):
            M.does_not_exist
    finally:
        sys.modules.pop('_snsde_vendored_controldiffeq', None)
        sys.modules.pop('_snsde_vendored_controldiffeq.solver', None)
        M._VENDORED = None
    monkeypatch.setattr(M, '_VENDORED', False)      # nothing to delegate to: a clear AttributeError
    with pytest.raises(AttributeError, match='mirror'):
        M.cdeint


def test_dropin_fixture_wrappers_reproduce_the_reference_outputs_on_cpu():
    """tests/golden/dropin.npz = the reference's own wrapper + Diffusion_model code over this package's mirrors
    (tools/check_reference_dropin.py).  The mirror's wrappers and Diffusion_model, loaded with the reference's state_dict,
    reproduce those outputs on the CPU tensor-op path."""
    from tests.helpers import load
    F = load('dropin.npz')

    class Replay:
        levy_area_approximation = 'space-time'

        def __init__(self, dW, dU=None):
            self.dW, self.dU, self.n = dW, dU, 0

        def __call__(self, ta, tb, return_U=False):
            i, self.n = self.n, self.n + 1
            return (self.dW[i], self.dU[i]) if return_U else self.dW[i]

    def sd(prefix):
        pre = prefix + '/sd/'
        return {k[len(pre):]: torch.from_numpy(F[k].copy()) for k in F.files if k.startswith(pre)}

    B, H, C, L, NL, io, no, oc = (int(v) for v in F['cls/dims'])
    m = S.NeuralSDE(S.Diffusion_model(C, H, H, NL, input_option=io, noise_option=no), C, H, oc, initial=True)
    m.load_state_dict(sd('cls'))
    m.eval()
    with torch.no_grad():
        out = m(torch.from_numpy(F['cls/times']), (torch.from_numpy(F['cls/coeffs']),), torch.from_numpy(F['cls/final_index']),
                bm=Replay(torch.from_numpy(F['cls/dW'])))
    np.testing.assert_allclose(out.numpy(), F['cls/out'], rtol=1e-5, atol=1e-6)
    B, H, C, L, NL, io, no, oc = (int(v) for v in F['ists/dims'])
    m = S.IstsNeuralSDE(S.Diffusion_model(C, H, H, NL, input_option=io, noise_option=no), C, H, oc, initial=True)
    m.load_state_dict(sd('ists'))
    m.eval()
    with torch.no_grad():
        out, z = m(torch.from_numpy(F['ists/coeffs']), torch.from_numpy(F['ists/times']),
                   bm=Replay(torch.from_numpy(F['ists/dW']), torch.from_numpy(F['ists/dU'])))
    np.testing.assert_allclose(z.numpy(), F['ists/z'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out.numpy(), F['ists/out'], rtol=1e-5, atol=1e-6)


def test_bench_self_launch_builds_a_torchrun_command(monkeypatch):
    """`python bench.py --gpus 4` (no torchrun around it) re-executes under torch.distributed.run with one rank per GPU."""
    import importlib
    import subprocess
    import sys
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    bench = importlib.import_module('bench')
    seen = {}

    def fake_call(cmd, env=None):
        seen['cmd'], seen['env'] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, 'call', fake_call)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '7'])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen['cmd']
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1'] and '--nproc-per-node=4' in cmd
    assert '127.0.0.1' in cmd and cmd[-4:] == ['--gpus', '4', '--steps', '7']
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' or 'HSA_ENABLE_IPC_MODE_LEGACY' in os.environ


# ---- tutorial-style fields (fields.py): structural recognition and weight composition, no GPU ----------------------------

def _variant_drift(cf, flat, t, y, Xt):
    """float64 evaluation of the C ABI's variant semantics (include/snsde.h SNSDE_ACT_* / SNSDE_DRIFT_* / SNSDE_TIME_RAW)
    on a composed parameter block: what the lean kernel computes for the drift."""
    import torch
    P = {n: flat[o:o + int(np.prod(sh))].view(*sh).double() for n, o, sh in cf.layout}
    m = cf.model
    act = {0: torch.relu, 1: lambda x: 0.909 * torch.nn.functional.silu(x), 2: torch.nn.functional.silu}[m.activation]
    if m.input_option == 4:
        tf = torch.cat([torch.full_like(y[:, :1], t), torch.zeros_like(y[:, :1])], 1) if m.time_feature == 1 else \
            torch.cat([torch.sin(torch.full_like(y[:, :1], t)), torch.cos(torch.full_like(y[:, :1], t))], 1)
        yy = torch.cat([tf, y], 1) @ P['linear_in.weight'].T + P['linear_in.bias']
    else:
        yy = y @ P['linear_in.weight'].T + P['linear_in.bias']
    xt = Xt @ P['initial_network.weight'].T + P['initial_network.bias']
    z = act(torch.cat([yy, xt], 1) @ P['emb.weight'].T + P['emb.bias'])
    for i in range(m.num_hidden_layers - 1):
        z = act(z @ P[f'linears.{i}.weight'].T + P[f'linears.{i}.bias'])
    z = z @ P['linear_out.weight'].T + P['linear_out.bias']
    return {0: torch.tanh(z), 1: z, 2: z * y}[m.drift_output]


@pytest.mark.parametrize('kind', ['lsde', 'lnsde', 'lnsde_additive', 'gsde'])
@pytest.mark.parametrize('layers,act', [(1, 'lipswish'), (3, 'relu'), (2, 'silu')])
def test_tutorial_field_composition_matches_module(kind, layers, act):
    import torch
    from stable_neural_sdes_amd import fields
    from tests.tutorial_fields import TutorialField
    from tests.helpers import make_problem
    torch.manual_seed(3)
    B, H, C_, L = 5, 32, 3, 6
    pr = make_problem(1, 4, 17, 2, B, H, C_, L, times=np.linspace(0, 1, L))
    field = TutorialField(kind, C_, H, layers, act).double()
    field.set_X(torch.from_numpy(pr['coeffs']).double(), torch.from_numpy(pr['times']).double())
    cf = fields.compose(field)
    assert cf is not None
    assert cf.model.activation == {'relu': 0, 'lipswish': 1, 'silu': 2}[act]
    assert cf.model.input_option == (2 if kind == 'lsde' else 4) and cf.model.time_feature == (0 if kind == 'lsde' else 1)
    assert cf.model.noise_option == (12 if kind in ('lsde', 'lnsde_additive') else 13) and cf.model.diffusion_output == 1
    assert cf.model.num_hidden_layers == layers
    cf.model.drift_output = fields.DRIFT_TIMES_Y if kind == 'gsde' else fields.DRIFT_LINEAR     # (the GPU probe decides this)
    flat = cf.flat(torch.device('cpu'))
    assert flat.dtype == torch.float32 and flat.numel() == cf.numel
    t = torch.tensor(0.4, dtype=torch.float64)
    y = torch.rand(B, H, dtype=torch.float64) + 0.2
    with torch.no_grad():
        want = field.f(t, y)
        got = _variant_drift(cf, flat, 0.4, y, field.X.evaluate(t))
        assert float((got - want).abs().max()) <= 2e-6 * (1 + float(want.abs().max()))
        # time-only diffusion factor over a column of times == g(t, ones) row by row
        ts = torch.tensor([0.0, 0.3, 0.9], dtype=torch.float64)
        field32 = field.float()
        tab = cf.noise_table(ts, torch.device('cpu'))
        for n in range(3):
            assert torch.allclose(tab[n], field32.g(ts[n].float(), torch.ones(1, H))[0], rtol=1e-5, atol=1e-6)


def test_tutorial_field_structural_rejects():
    import torch
    from stable_neural_sdes_amd import fields
    from tests.tutorial_fields import TutorialField
    f = TutorialField('lsde', 2, 32, 1)
    f.f_net._model.append(torch.nn.Tanh())
    assert fields.compose(f) is None                       # Tanh-terminated MLP
    f = TutorialField('lsde', 2, 32, 1)
    f.f_net._model[1] = torch.nn.Tanh()
    assert fields.compose(f) is None                       # activation the kernel does not implement
    f = TutorialField('lnsde', 2, 32, 1)
    f.g = lambda t, y: torch.sin(y)                        # diffusion neither additive nor proportional to y
    assert fields.compose(f) is None
    assert fields.compose(S.Diffusion_model(2, 32, 32, 2, input_option=4, noise_option=17)) is None   # own fast path instead
    assert fields._classify_activation(torch.nn.SiLU()) == fields.ACT_SILU
    assert fields._classify_activation(torch.nn.ReLU()) == fields.ACT_RELU


def test_c_abi_variant_fields_validate():
    import ctypes as C
    from stable_neural_sdes_amd import _lib, engine
    assert C.sizeof(_lib.Model) == 40
    L = _lib.lib()
    ok = engine.model_struct(3, 32, 32, 2, 4, 13, activation=1, drift_output=2, diffusion_output=1, time_feature=1)
    assert L.snsde_param_count(C.byref(ok)) > 0
    for kw in (dict(activation=3), dict(drift_output=3), dict(diffusion_output=2), dict(diffusion_output=3), dict(time_feature=2),
               dict(activation=-1)):
        assert L.snsde_param_count(C.byref(engine.model_struct(3, 32, 32, 2, 4, 13, **kw))) < 0
    # SNSDE_DIFFUSION_RAW_NET (the un-rectified two-layer net of the tutorial's NeuralSDEFunc): noise_option 18 / 19 only
    assert L.snsde_param_count(C.byref(engine.model_struct(3, 32, 32, 1, 3, 18, activation=1, drift_output=1, diffusion_output=2,
                                                           time_feature=1))) > 0


def test_forward_path_query_names_the_kernel_family():
    """snsde_forward_path (host-side): the BASELINE configurations land on the kernels DESIGN.md says they do."""
    import ctypes as C
    from stable_neural_sdes_amd import _lib, engine
    L = _lib.lib()

    def path(H, Cn, io, no, B, method, NL=2, kernel='auto', **variant):
        s = _lib.Solve()
        s.model = engine.model_struct(Cn, H, H, NL, io, no, **variant)
        s.batch, s.knots, s.n_steps, s.n_out = B, 51, 50, 2
        s.method, s.kernel = {'euler': _lib.EULER, 'milstein': _lib.MILSTEIN, 'srk': _lib.SRK}[method], _lib.KERNELS[kernel]
        return _lib.PATHS[L.snsde_forward_path(C.byref(s))]

    assert path(128, 21, 4, 17, 1024, 'euler') == 'lean'                 # K2
    assert path(128, 21, 6, 17, 512, 'euler') == 'lean'                  # K3 shard
    assert path(64, 69, 3, 18, 2048, 'euler') == 'w4'                    # K4: diffusion net at H = 64: a wave pair per 4 rows
    assert path(64, 69, 3, 18, 2048, 'srk') == 'w4' and path(64, 69, 3, 18, 16384, 'euler') == 'mfma16'
    assert path(64, 69, 3, 18, 2048, 'milstein') == 'mfma4'
    assert path(256, 14, 4, 17, 128, 'milstein') == 'lean-streamed'      # K5 shard
    assert path(256, 14, 4, 17, 16384, 'milstein') == 'mfma16'           # large batch: 16-row tiles
    assert path(128, 21, 4, 17, 1024, 'srk') == 'mfma-srk'
    assert path(48, 5, 4, 17, 64, 'euler') == 'generic'                  # no MFMA instantiation for H = 48
    assert path(128, 21, 3, 18, 64, 'milstein') == 'generic'             # Milstein through a diffusion net: generic kernels
    assert path(128, 21, 3, 7, 64, 'milstein') == 'none'                 # sqrt(y): no finite dg/dy at the clipped values
    assert path(128, 21, 4, 17, 64, 'euler', kernel='generic') == 'generic'
    assert path(32, 2, 4, 13, 256, 'euler', NL=1, activation=1, drift_output=1, diffusion_output=1, time_feature=1) == 'lean'
    assert path(48, 2, 4, 13, 256, 'euler', NL=1, activation=1, drift_output=1, diffusion_output=1, time_feature=1) == 'none'


@pytest.mark.parametrize('io,no,H,HH', [(4, 17, 48, 48), (3, 18, 40, 24), (6, 13, 20, 20), (1, 5, 24, 40), (2, 16, 12, 12), (5, 15, 10, 30)])
def test_zero_padded_model_reproduces_the_original_on_its_real_components(io, no, H, HH):
    """engine.padded_flat: the padded parameter block, loaded into a Diffusion_model of the padded width, gives the same
    f and g on the real state components (and exact zeros of the drift on the padded ones) - the premise of
    engine.padding_plan.  Pure host logic (CPU tensors)."""
    import torch
    from stable_neural_sdes_amd import _lib, engine
    from tests.helpers import make_problem
    C_, NL, B, L, P = 3, 3, 6, 7, 64
    torch.manual_seed(io * 100 + no)
    pr = make_problem(3, io, no, NL, B, H, C_, L)
    small = S.Diffusion_model(C_, H, HH, NL, input_option=io, noise_option=no)
    with torch.no_grad():
        for p in small.parameters():
            p.mul_(1.5)
    times, coeffs = torch.from_numpy(pr['times']), torch.from_numpy(pr['coeffs'])
    small.set_X(coeffs, times)
    layout, _ = _lib.param_layout(engine.model_struct(C_, H, HH, NL, io, no))
    mp = engine.model_struct(C_, P, P, NL, io, no)
    layout_p, numel_p = _lib.param_layout(mp)
    flat = engine.padded_flat(small, layout, layout_p, H, P, torch.device('cpu'), grad=False)
    assert flat.numel() == numel_p
    big = S.Diffusion_model(C_, P, P, NL, input_option=io, noise_option=no)
    big.load_state_dict({name: flat[off:off + int(np.prod(shape))].view(*shape) for name, off, shape in layout_p})
    big.set_X(coeffs, times)
    y = torch.rand(B, H) + 0.1
    yp = torch.nn.functional.pad(y, (0, P - H))
    t = torch.tensor(2.5)
    with torch.no_grad():
        assert torch.allclose(big.f(t, yp)[:, :H], small.f(t, y), rtol=1e-5, atol=1e-6)
        assert torch.allclose(big.g(t, yp)[:, :H], small.g(t, y), rtol=1e-5, atol=1e-6)
        assert float(big.f(t, yp)[:, H:].abs().max()) == 0.0
    # differentiable: the gradient of a function of the padded block reaches the module's parameters
    small.zero_grad()
    engine.padded_flat(small, layout, layout_p, H, P, torch.device('cpu'), grad=True).square().sum().backward()
    w = small.linear_out.weight
    assert torch.allclose(w.grad, 2 * w.detach())


@pytest.mark.parametrize('ts,dt,chunk', [([0, 12], 1.0, 5), ([0, 2.5, 4, 8], 0.5, 3), (list(range(10)), 1.0, 4), ([0, 0.3, 0.35, 1.0], 0.25, 1)])
def test_recompute_chunks_partition_the_step_table(ts, dt, chunk):
    """engine.sub_grid / chunk_plan (recompute-mode backward): the chunks' step rows are the parent's rows, every parent output
    belongs to exactly one chunk with its interpolation weights, each chunk ends with an on-grid output (the hand-over state)
    and the per-step output bookkeeping (count, first index) is consistent with the chunk's own out_step list."""
    from stable_neural_sdes_amd import engine
    times = np.linspace(0, max(ts), 9).astype(np.float32)
    g = engine.StepGrid(np.asarray(ts, np.float32), dt, times, None)
    plan = engine.chunk_plan(g, chunk)
    assert [(a, b) for a, b, _, _ in plan] == [(n0, min(n0 + chunk, g.N)) for n0 in range(0, g.N, chunk)]
    seen = []
    for n0, n1, sub, ks in plan:
        assert sub.N == n1 - n0 and sub.T == len(ks) + 2
        np.testing.assert_array_equal(sub.step_tab[:, :8], g.step_tab[n0:n1, :8])          # same times / sizes / spline intervals
        for j, k in enumerate(ks):                       # parent output k (ys index) = sub output j + 1
            assert sub.out_step[j] == g.out_step[k - 1] - n0
            np.testing.assert_array_equal(sub.out_w[j], g.out_w[k - 1])
        assert sub.out_step[-1] == sub.N - 1 and tuple(sub.out_w[-1]) == (0.0, 1.0)
        nout = sub.step_tab[:, 8].view(np.int32)
        first = sub.step_tab[:, 9].view(np.int32)
        for n in range(sub.N):
            idx = [j for j, st in enumerate(sub.out_step) if st == n]
            assert nout[n] == len(idx) and (not idx or first[n] == idx[0])
        seen += ks
    assert seen == list(range(1, g.T))


def test_readout_head_recognition_gate():
    """engine.head_layers: the fused inference head only stands in for the wrappers' readout structures in evaluation mode
    (neuralsde.py:59-61: Linear, BatchNorm1d, ReLU, Dropout, Linear)."""
    nn = torch.nn
    head = nn.Sequential(nn.Linear(8, 8), nn.BatchNorm1d(8), nn.ReLU(), nn.Dropout(0.1), nn.Linear(8, 2))
    engine = S.engine
    assert engine.head_layers(head) is None                     # training mode: batch statistics and dropout
    head.eval()
    tanh, lin1, bn, lin2 = engine.head_layers(head)
    assert not tanh and lin1 is head[0] and bn is head[1] and lin2 is head[4]
    assert engine.head_layers(nn.Sequential(nn.Tanh(), nn.Linear(8, 8), nn.ReLU(), nn.Linear(8, 2)))[0] is True
    assert engine.head_layers(nn.Sequential(nn.Linear(8, 8), nn.Tanh(), nn.Linear(8, 2))) is None
    assert engine.head_layers(nn.Sequential(nn.Linear(8, 8), nn.ReLU(), nn.Linear(4, 2))) is None
    assert engine.head_layers(nn.Linear(8, 2)) is None



@pytest.mark.parametrize('no', [14, 15, 18, 19])
def test_oracle_diffusion_net_vjp_equals_autograd_on_the_module(no):
    """oracle.diffusion_g_vjp (the Milstein term of the diffusion nets) against torch autograd through Diffusion_model.g in
    float64, and the oracle's Milstein trajectory against the tensor-op loop (torchsde's VJP form) on the same increments."""
    torch.manual_seed(no)
    B, H, C, L = 6, 10, 3, 7
    m = S.Diffusion_model(C, H, H, 2, input_option=1, noise_option=no).double()
    p = {k: v.detach().numpy() for k, v in m.state_dict().items()}
    y = torch.randn(B, H, dtype=torch.float64, requires_grad=True)
    cot = torch.randn(B, H, dtype=torch.float64)
    t = torch.tensor(0.7, dtype=torch.float64)
    g = m.g(t, y)
    want, = torch.autograd.grad(g, y, grad_outputs=cot)
    g_o, vjp_o = O.diffusion_g_vjp(p, no, 0.7, y.detach().numpy(), cot.numpy())
    np.testing.assert_allclose(g_o, g.detach().numpy(), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(vjp_o, want.numpy(), rtol=1e-10, atol=1e-12)
    rng = np.random.default_rng(no)
    times = np.arange(L, dtype=np.float32)
    X = rng.standard_normal((B, L, C)).cumsum(1)
    coeffs = S.controldiffeq.natural_cubic_spline_coeffs(torch.from_numpy(times), torch.from_numpy(X).float())
    coeffs = torch.cat(coeffs, dim=-1).double()
    m.set_X(coeffs, torch.from_numpy(times))
    y0 = rng.standard_normal((B, H))
    ts = np.array([0, 2.5, L - 1], np.float32)
    t0s = O.step_grid(ts, 0.5)[0]
    dW = rng.standard_normal((len(t0s), B, H)) * np.sqrt(0.5)

    class Replay:
        n = 0
        def __call__(self, ta, tb):
            self.n += 1
            return torch.from_numpy(dW[self.n - 1])
    with torch.no_grad():
        loop = S.sdeint(m, torch.from_numpy(y0), torch.from_numpy(ts), bm=Replay(), method='milstein', dt=0.5,
                        options={'backend': 'torch'})
    ora, _ = O.solve_diffusion_model(p, 1, no, coeffs.numpy(), times, y0, ts, 0.5, dW, method='milstein')
    np.testing.assert_allclose(ora, loop.numpy(), rtol=1e-9, atol=1e-10)


TUT = load('tutorial.npz')
T1_CASES = sorted({k.split('/')[1] for k in TUT.files if k.startswith('T1/')})


@pytest.mark.parametrize('case', T1_CASES)
def test_tutorial_field_class_reproduces_the_notebooks_fields(case):
    """tests/golden/tutorial.npz was produced by executing the notebooks' vector-field cell (tutorial/*.ipynb cell 7).  The
    test-side class must load those state_dicts strictly and give the same f / g; the host tensor-op loop over the mirror's
    CubicSpline must reproduce the float64 trajectory; the composition (fields.compose) must accept the module."""
    from stable_neural_sdes_amd import fields
    from tests.tutorial_fields import TutorialField
    g = group(TUT, f'T1/{case}')
    C, H, layers = (int(v) for v in g['meta'])
    sd = {k: torch.from_numpy(v.copy()) for k, v in params_of(TUT, f'T1/{case}').items()}
    field = TutorialField(str(g['kind']), C, H, layers, str(g['activation']))
    field.load_state_dict(sd, strict=True)
    times = torch.from_numpy(g['times'])
    field.set_X(torch.from_numpy(g['coeffs']), times)
    y0 = torch.from_numpy(g['y0'])
    with torch.no_grad():
        for i, t in enumerate(torch.from_numpy(g['probe_t'])):
            np.testing.assert_allclose(field.f(t, y0).numpy(), g['f'][i], rtol=2e-5, atol=2e-6)
            np.testing.assert_allclose(field.g(t, y0).numpy(), g['g'][i], rtol=2e-5, atol=2e-6)
    assert fields.compose(field) is not None
    f64 = TutorialField(str(g['kind']), C, H, layers, str(g['activation'])).double()
    f64.load_state_dict({k: v.double() for k, v in sd.items()}, strict=True)
    f64.set_X(torch.from_numpy(g['coeffs']).double(), times)

    class Replay:
        n = 0
        def __call__(self, ta, tb, return_U=False):
            self.n += 1
            w = torch.from_numpy(g['dW'][self.n - 1]).double()
            return (w, torch.from_numpy(g['dU'][self.n - 1]).double()) if return_U else w
    with torch.no_grad():
        ys = S.sdeint(f64, y0.double(), times, bm=Replay(), dt=float(g['dt']), method=str(g['method']),
                      options={'backend': 'torch'})
    np.testing.assert_allclose(ys.numpy(), g['ys64'], rtol=1e-7, atol=1e-8)


# ---- torch-ists' LatentSDE (tests/golden/latent.npz: generated by importing the reference class) ------------------------------
LAT = load('latent.npz')
L1_CASES = sorted({k.split('/')[1] for k in LAT.files if k.startswith('L1/')})


def _latent_case(case, dtype=torch.float32):
    from tests.latent_field import LatentField
    g = group(LAT, f'L1/{case}')
    C, H, HH, NL = (int(v) for v in g['meta'])
    sd = {k: torch.from_numpy(v.copy()) for k, v in params_of(LAT, f'L1/{case}').items()}
    m = LatentField(C, H, HH, NL).to(dtype)
    m.load_state_dict({k: v.to(dtype) for k, v in sd.items()}, strict=True)
    return g, m


@pytest.mark.parametrize('case', L1_CASES)
def test_latent_field_class_reproduces_the_reference_latent_sde(case):
    """The test-side LatentField loads the reference LatentSDE's state_dict strictly (buffers included), gives the same
    f_aug / g_aug, and the tensor-op loop through names={'drift': 'f_aug', 'diffusion': 'g_aug'} reproduces the float64
    trajectory of the augmented system; the latent composition (fields.compose_latent) accepts the module."""
    from stable_neural_sdes_amd import fields
    g, m = _latent_case(case)
    y0 = torch.from_numpy(g['y0'])
    with torch.no_grad():
        for i, t in enumerate(torch.from_numpy(g['probe_t'])):
            np.testing.assert_allclose(m.f_aug(t, y0).numpy(), g['f_aug'][i], rtol=2e-5, atol=2e-5)
            np.testing.assert_allclose(m.g_aug(t, y0).numpy(), g['g_aug'][i], rtol=0, atol=0)
    names = {'drift': 'f_aug', 'diffusion': 'g_aug'}
    cf = fields.compose_latent(m, names, y0.shape[1])
    assert cf is not None and cf.parts['latent'] == y0.shape[1] - 1
    assert cf.model.hidden_channels in (16, 32, 64, 128) and cf.model.hidden_channels >= y0.shape[1] - 1
    flat = cf.flat(torch.device('cpu'))
    assert flat.numel() == cf.numel and bool(torch.isfinite(flat).all())
    assert fields.compose_latent(m, {'drift': 'f', 'diffusion': 'g'}, y0.shape[1]) is None
    _, m64 = _latent_case(case, torch.float64)

    class Replay:
        n = 0
        def __call__(self, ta, tb, return_U=False):
            self.n += 1
            w = torch.from_numpy(g['dW'][self.n - 1]).double()
            return (w, torch.from_numpy(g['dU'][self.n - 1]).double()) if return_U else w
    with torch.no_grad():
        ys = S.sdeint(m64, y0.double(), torch.from_numpy(g['ts']), bm=Replay(), dt=float(g['dt']), method=str(g['method']), names=names)
    np.testing.assert_allclose(ys.numpy(), g['ys64'], rtol=1e-7, atol=1e-7)


def test_latent_composition_rejects_modules_whose_accumulator_feeds_back():
    from stable_neural_sdes_amd import fields
    from tests.latent_field import LatentField

    class Coupled(LatentField):
        def f_aug(self, t, y):
            out = super().f_aug(t, y)
            return torch.cat([out[:, :-1] + 0.1 * y[:, -1:], out[:, -1:]], dim=1)
    names = {'drift': 'f_aug', 'diffusion': 'g_aug'}
    assert fields.compose_latent(Coupled(2, 9, 16, 2), names, 9) is None
    assert fields.compose_latent(LatentField(2, 9, 16, 2), names, 9) is not None
    assert fields.compose_latent(LatentField(2, 9, 16, 2), names, 10) is None          # state width != latent + 1


def test_memoised_mapping_does_not_travel_and_follows_rebound_functions():
    """the latent mapping is memoised on the module, keyed on the layer objects and the functions the structural
    probes looked at - re-binding f_aug to something the split cannot express drops the mapping - and copy.deepcopy / pickle of
    the module carry an empty cache (no device tensors or ctypes structs ride along)."""
    import copy, pickle, types
    from stable_neural_sdes_amd import fields
    from tests.latent_field import LatentField
    names = {'drift': 'f_aug', 'diffusion': 'g_aug'}
    m = LatentField(2, 9, 16, 2)
    cf = fields.compose_latent(m, names, 9)
    assert cf is not None and fields.compose_latent(m, names, 9) is cf                 # memoised
    for clone in (copy.deepcopy(m), pickle.loads(pickle.dumps(m))):
        slot = clone.__dict__['_snsde_cache']
        assert slot.latent is None and slot.composed is None
        cf2 = fields.compose_latent(clone, names, 9)
        assert cf2 is not None and cf2 is not cf and cf2.sde.sde is clone
    orig = type(m).f_aug

    def coupled(self, t, y):          # the accumulator feeds back into the latent drift: not splittable
        out = orig(self, t, y)
        return torch.cat([out[:, :-1] + 0.1 * y[:, -1:], out[:, -1:]], dim=1)
    m.f_aug = types.MethodType(coupled, m)
    assert fields.compose_latent(m, names, 9) is None
    del m.f_aug
    assert fields.compose_latent(m, names, 9) is not None
    m.linears[0] = torch.nn.Linear(16, 16)                                            # a replaced layer: recognised again
    cf3 = fields.compose_latent(m, names, 9)
    assert cf3 is not None and cf3 is not cf and cf3.parts['linears'][0] is m.linears[0]


def test_param_index_follows_replaced_layers_and_added_parameters():
    """engine.param_index caches the parameter lists on the module and validates them against the module TREE: a replaced
    submodule (its old _parameters dict still holds the old tensors), a re-assigned, an added or a removed parameter rebuild it."""
    m = S.Diffusion_model(3, 16, 16, 2, input_option=3, noise_option=18)
    layout, _ = S.engine._lib.param_layout(S.engine.model_struct(3, 16, 16, 2, 3, 18))
    idx = S.engine.param_index(m, layout)
    assert S.engine.param_index(m, layout) is idx
    m.linears[0] = torch.nn.Linear(16, 16)
    idx2 = S.engine.param_index(m, layout)
    assert idx2 is not idx and any(p is m.linears[0].weight for p in idx2.params)
    m.linear_out.weight = torch.nn.Parameter(torch.zeros_like(m.linear_out.weight))
    idx3 = S.engine.param_index(m, layout)
    assert idx3 is not idx2 and any(p is m.linear_out.weight for p in idx3.params)
    assert idx3.valid(m, layout)
    m.register_parameter('extra', torch.nn.Parameter(torch.zeros(2)))      # (such a module is no longer the reference's: recognise() refuses it)
    assert not idx3.valid(m, layout)
    del m._parameters['extra']
    assert idx3.valid(m, layout)
    m.add_module('aux', torch.nn.Linear(2, 2))
    assert not idx3.valid(m, layout)
    del m._modules['aux']
    assert idx3.valid(m, layout) and S.engine.param_index(m, layout) is idx3


def test_every_step_grid_outputs_every_state_of_the_same_steps():
    """engine.every_step_grid: the caller's solver steps, one exact output after each (the LatentSDE split solve reads the whole
    trajectory through it)."""
    ts = np.array([0.0, 0.13, 0.4, 0.77, 1.0], dtype=np.float32)
    grid = S.engine.StepGrid(ts, 0.06, np.array([0.0, 1.0], dtype=np.float32), None)
    full = S.engine.every_step_grid(grid)
    assert full is S.engine.every_step_grid(grid)                       # memoised
    assert full.N == grid.N and full.T == grid.N + 1
    np.testing.assert_array_equal(full.t0, grid.t0)
    np.testing.assert_array_equal(full.t1, grid.t1)
    np.testing.assert_array_equal(full.step_tab[:, :8], grid.step_tab[:, :8])
    np.testing.assert_array_equal(full.out_step, np.arange(grid.N))
    np.testing.assert_array_equal(full.out_w, np.tile([[0.0, 1.0]], (grid.N, 1)).astype(np.float32))
    np.testing.assert_array_equal(full.step_tab[:, 8].view(np.int32), np.ones(grid.N, dtype=np.int32))
    np.testing.assert_array_equal(full.step_tab[:, 9].view(np.int32), np.arange(grid.N, dtype=np.int32))


def test_scalar_noise_euler_in_the_tensor_loop():
    """torchsde's scalar noise (g (B, H, 1), one Brownian motion per row): Euler y + f h + g[..., 0] I with I of shape (B, 1);
    the other schemes refuse (they need the dense Jacobian-vector product of g).  The tutorial's Neural ODE notebook uses this
    shape with g = 0."""
    class Scalar(torch.nn.Module):
        sde_type, noise_type = 'ito', 'scalar'

        def f(self, t, y):
            return -0.5 * y + t

        def g(self, t, y):
            return (0.3 * y).unsqueeze(-1)
    B, H = 4, 3
    ts = torch.tensor([0.0, 0.5, 1.0], dtype=torch.float64)
    y0 = torch.linspace(0.1, 1.2, B * H, dtype=torch.float64).reshape(B, H)
    rng = np.random.default_rng(0)
    dW = torch.from_numpy(rng.standard_normal((4, B, 1)) * 0.5)

    class Replay:
        n = 0
        def __call__(self, ta, tb):
            self.n += 1
            return dW[self.n - 1]
    ys = S.sdeint(Scalar(), y0, ts, bm=Replay(), dt=0.25, method='euler')
    y, want, t = y0, [y0], 0.0
    for n in range(4):
        y = y + (-0.5 * y + t) * 0.25 + 0.3 * y * dW[n]
        t += 0.25
        if n % 2 == 1:
            want.append(y)
    np.testing.assert_allclose(ys.numpy(), torch.stack(want).numpy(), rtol=1e-12, atol=1e-12)
    free = S.sdeint(Scalar(), y0, ts, dt=0.25, method='euler', options={'seed': 5})
    assert tuple(free.shape) == (3, B, H) and bool(torch.isfinite(free).all())
    with pytest.raises(NotImplementedError):
        S.sdeint(Scalar(), y0, ts, dt=0.25, method='srk')
