#!/usr/bin/env python
"""Generate golden vectors by IMPORTING the reference (build container only).

Run:  python tests/golden/make_golden.py          (needs /root/reference; writes tests/golden/*.npz)

The fixtures hold inputs + expected outputs only (no reference source / bytecode travels).
Import route = SURVEY.md Appendix A: the vendored ``controldiffeq`` spline code is imported
as-is, ``torchcde.CubicSpline`` is adapted onto it (same arithmetic) and
``torchsde`` is a stub (its ``sdeint`` is replaced by a recorder or by the fixed-step loop
below), because neither third-party package is installed here.

Fixture groups:
  G1 spline_coeffs.npz   natural_cubic_spline_coeffs (regular/irregular grid, NaN patterns, L=2)
  G2 spline_eval.npz     NaturalCubicSpline.evaluate / derivative
  G3 fg.npz              Diffusion_model.f / g for every (input_option, noise_option), fp32 + fp64
  G4 wrapper.npz         NeuralSDE / NeuralSDE_forecasting bookkeeping around sdeint (recorded calls)
  G5 traj.npz            fixed-step Euler / Milstein trajectories driving the reference f/g with
                         supplied dW (fp32 and fp64 modules)
"""
import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = '/root/reference'


def load_reference(which='benchmark_classification'):
    ref_dir = os.path.join(REF_ROOT, which)
    for name in list(sys.modules):
        if name == 'controldiffeq' or name.startswith('controldiffeq.'):
            del sys.modules[name]
    pkg = types.ModuleType('controldiffeq')
    pkg.__path__ = [os.path.join(ref_dir, 'controldiffeq')]
    sys.modules['controldiffeq'] = pkg
    interp = importlib.import_module('controldiffeq.interpolate')

    class CubicSplineAdapter:
        def __init__(self, coeffs, t):
            C = coeffs.size(-1) // 4
            self._sp = interp.NaturalCubicSpline(t, tuple(coeffs[..., k * C:(k + 1) * C] for k in range(4)))

        def evaluate(self, t):
            return self._sp.evaluate(t)

        def derivative(self, t):
            return self._sp.derivative(t)

    tc = types.ModuleType('torchcde')
    tc.CubicSpline = CubicSplineAdapter
    sys.modules['torchcde'] = tc
    tsde = types.ModuleType('torchsde')
    sys.modules['torchsde'] = tsde
    spec = importlib.util.spec_from_file_location('ref_neuralsde_' + which,
                                                  os.path.join(ref_dir, 'models_sde', 'neuralsde.py'))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    return ref, interp, tsde


def load_torch_ists():
    path = os.path.join(REF_ROOT, 'torch-ists', 'torch_ists', 'diff_module', 'NSDE', 'nsde_model.py')
    spec = importlib.util.spec_from_file_location('ref_torch_ists_nsde', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def npy(t):
    return t.detach().cpu().numpy()


def make_path(gen, B, L, C, dtype=torch.float32):
    x = torch.randn(B, L, C, generator=gen, dtype=torch.float64).cumsum(1) * 0.3
    return x.to(dtype)


# ------------------------------------------------------------------------------------------------
def g1_g2(interp, out):
    gen = torch.Generator().manual_seed(101)
    cases = {}
    B, C = 4, 3
    for name, L, irregular in (('reg7', 7, False), ('irr8', 8, True), ('reg9', 9, False), ('L2', 2, False),
                               ('L3', 3, True)):
        if irregular:
            times = torch.rand(L, generator=gen, dtype=torch.float64).cumsum(0).float()
        else:
            times = torch.linspace(0, L - 1, L)
        X = make_path(gen, B, L, C)
        cases[name + '_nonan'] = (times, X)
        if L >= 7:
            Xn = X.clone()
            Xn[0, 2, 0] = float('nan')           # interior
            Xn[0, 3, 0] = float('nan')           # two adjacent interior
            Xn[1, 0, 1] = float('nan')           # leading
            Xn[1, 1, 1] = float('nan')
            Xn[2, L - 1, 2] = float('nan')       # trailing
            Xn[3, :, 0] = float('nan')           # all-NaN channel
            Xn[3, 1:, 1] = float('nan')          # single observation
            mask = torch.rand(B, L, C, generator=gen) < 0.3
            Xn2 = X.clone()
            Xn2[mask] = float('nan')
            cases[name + '_nan'] = (times, Xn)
            cases[name + '_nan30'] = (times, Xn2)
    eval_ts = {}
    for name, (times, X) in cases.items():
        for dt_name, dtype in (('f32', torch.float32), ('f64', torch.float64)):
            t_, X_ = times.to(dtype), X.to(dtype)
            coeffs = interp.natural_cubic_spline_coeffs(t_, X_)
            out[f'G1/{name}/{dt_name}/times'] = npy(t_)
            out[f'G1/{name}/{dt_name}/X'] = npy(X_)
            for k, nm in enumerate(('a', 'b', 'two_c', 'three_d')):
                out[f'G1/{name}/{dt_name}/{nm}'] = npy(coeffs[k])
            spline = interp.NaturalCubicSpline(t_, coeffs)
            lo, hi = float(t_[0]), float(t_[-1])
            pts = [lo - 0.5, lo, hi, hi + 0.7] + [float(v) for v in t_[1:-1]]
            pts += [lo + (hi - lo) * f for f in (0.13, 0.5, 0.77, 0.999)]
            pts = torch.tensor(pts, dtype=dtype)
            ev = torch.stack([spline.evaluate(p) for p in pts])
            de = torch.stack([spline.derivative(p) for p in pts])
            out[f'G2/{name}/{dt_name}/t'] = npy(pts)
            out[f'G2/{name}/{dt_name}/evaluate'] = npy(ev)
            out[f'G2/{name}/{dt_name}/derivative'] = npy(de)
    return cases


# ------------------------------------------------------------------------------------------------
def randomise(model, gen):
    """Non-default but bounded parameters so every branch matters (theta, sigma included)."""
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name == 'theta':
                p.copy_(torch.tensor([[0.7]]) + 0.3 * torch.randn(1, 1, generator=gen))
            elif name in ('sigma', 'sigma_diag'):
                p.copy_(-0.5 + 0.4 * torch.randn(p.shape, generator=gen))
            else:
                p.add_(0.2 * torch.randn(p.shape, generator=gen))


def g3(ref, ref_fc, ists, interp, out):
    gen = torch.Generator().manual_seed(202)
    B, H, C, L = 4, 6, 3, 6
    times = torch.linspace(0, 2.5, L)
    X = make_path(gen, B, L, C)
    X[1, 2, 0] = float('nan')
    coeffs = torch.cat(interp.natural_cubic_spline_coeffs(times, X), dim=-1)
    out['G3/times'] = npy(times)
    out['G3/coeffs'] = npy(coeffs)
    y = torch.randn(B, H, generator=gen) * 1.5
    y[0, 0] = 0.0
    y[0, 1] = -2.0
    y[2, 3] = 40.0      # saturates tanh
    y[3, 4] = -1e-3
    out['G3/y'] = npy(y)
    tvals = [0.0, 1.3, 7.25]
    out['G3/t'] = np.array(tvals, dtype=np.float64)
    n_models = 0
    models, flat, offs, o32, o64 = [], [], [0], [], []
    for io in range(7):
        for no in range(20):
            for NL in ((1, 2, 3) if (io, no) in ((2, 16), (4, 17), (6, 17), (1, 0), (1, 18), (3, 18)) else (2,)):
                torch.manual_seed(1000 + 100 * io + no + 7 * NL)
                m = ref.Diffusion_model(C, H, H, NL, theta=1.0, sigma=1.0, input_option=io, noise_option=no)
                randomise(m, gen)
                sd = m.state_dict()
                # parameters are stored flat, concatenated in state_dict order (names/shapes are a
                # function of (io, no, NL, C, H): SURVEY 8b "Fast-path recognition")
                vec = torch.cat([v.reshape(-1) for v in sd.values()])
                m.set_X(coeffs, times)
                with torch.no_grad():
                    f32 = torch.stack([m.f(torch.tensor(t), y) for t in tvals])
                    g32 = torch.stack([m.g(torch.tensor(t), y) for t in tvals])
                    md = ref.Diffusion_model(C, H, H, NL, input_option=io, noise_option=no).double()
                    md.load_state_dict({k: v.double() for k, v in sd.items()})
                    md.set_X(coeffs.double(), times.double())
                    yd = y.double()
                    f64 = torch.stack([md.f(torch.tensor(t, dtype=torch.float64), yd) for t in tvals])
                    g64 = torch.stack([md.g(torch.tensor(t, dtype=torch.float64), yd) for t in tvals])
                models.append((io, no, NL))
                flat.append(npy(vec))
                offs.append(offs[-1] + vec.numel())
                o32.append(np.stack([npy(f32), npy(g32)]))
                o64.append(np.stack([npy(f64), npy(g64)]))
                if n_models == 0:
                    out['G3/example_keys'] = np.array(list(sd.keys()))
                # the three copies in the reference are numerically identical (its own test enforces 1e-6)
                if NL == 2 and no in (0, 16, 17, 18):
                    for other in (ref_fc, ists):
                        mo = other.Diffusion_model(C, H, H, NL, input_option=io, noise_option=no)
                        mo.load_state_dict(sd)
                        mo.set_X(coeffs, times)
                        with torch.no_grad():
                            fo = torch.stack([mo.f(torch.tensor(t), y) for t in tvals])
                            go = torch.stack([mo.g(torch.tensor(t), y) for t in tvals])
                        assert torch.equal(fo, f32) and torch.equal(go, g32), (io, no)
                n_models += 1
    out['G3/models'] = np.array(models, dtype=np.int32)
    out['G3/params_flat'] = np.concatenate(flat)
    out['G3/params_off'] = np.array(offs, dtype=np.int64)
    out['G3/out32'] = np.stack(o32)     # (M, 2[f,g], T, B, H)
    out['G3/out64'] = np.stack(o64)
    print('G3 models:', n_models)


# ------------------------------------------------------------------------------------------------
def g4(ref, ref_fc, ists, interp, tsde, out):
    gen = torch.Generator().manual_seed(303)
    B, H, C, L = 6, 4, 3, 8
    rec = {}

    def fake_sdeint(sde, y0, ts, dt, **kw):
        rec['ts'] = ts.clone()
        rec['dt'] = dt
        rec['kw'] = {k: (dict(v) if isinstance(v, dict) else v) for k, v in kw.items()}
        rec['y0'] = y0.clone()
        T = ts.shape[0]
        # deterministic fake trajectory: z[k] = y0 + ts[k]
        return torch.stack([y0 + ts[k] for k in range(T)])

    for mod in (ref, ref_fc, ists):
        mod.torchsde.sdeint = fake_sdeint
    for case, times, fi in (
            ('int_grid', torch.linspace(1, 8, L), torch.tensor([7, 3, 3, 5, 0, 7])),
            ('no_ends', torch.linspace(0, 7, L), torch.tensor([2, 3, 3, 5, 4, 2])),
            ('all_last', torch.linspace(0, 7, L), torch.tensor([7, 7, 7, 7, 7, 7])),
            ('lin01', torch.linspace(0, 1, L), torch.tensor([1, 6, 2, 5, 7, 0]))):
        X = make_path(gen, B, L, C)
        coeffs = torch.cat(interp.natural_cubic_spline_coeffs(times, X), dim=-1)
        torch.manual_seed(5)
        func = ref.Diffusion_model(C, H, H, 2, input_option=4, noise_option=17)
        model = ref.NeuralSDE(func, C, H, 2, initial=True).eval()
        with torch.no_grad():
            pred = model(times, [coeffs], fi)
            z0 = model.initial_network(func.X.evaluate(times[0]))
        k = f'G4/{case}'
        out[f'{k}/times'] = npy(times)
        out[f'{k}/final_index'] = npy(fi)
        out[f'{k}/coeffs'] = npy(coeffs)
        out[f'{k}/ts'] = npy(rec['ts'])
        out[f'{k}/dt'] = np.float64(rec['dt'])
        out[f'{k}/options_dt'] = np.float64(rec['kw']['options']['dt'])
        assert rec['kw']['method'] == 'euler'
        out[f'{k}/y0'] = npy(rec['y0'])
        out[f'{k}/z0'] = npy(z0)
        out[f'{k}/pred'] = npy(pred)
        for kk, v in model.state_dict().items():
            out[f'{k}/param/{kk}'] = npy(v)
    # forecasting wrapper: 4 natural-spline tensors concatenated, ts = times, decode last output_time
    times = torch.linspace(0, 7, L)
    X = make_path(gen, B, L, C)
    cs = interp.natural_cubic_spline_coeffs(times, X)
    torch.manual_seed(6)
    func = ref_fc.Diffusion_model(C, H, H, 2, input_option=2, noise_option=16)
    model = ref_fc.NeuralSDE_forecasting(func, C, 3, H, 5, initial=True).eval()
    with torch.no_grad():
        pred = model(times, list(cs), torch.zeros(B, dtype=torch.long))
    k = 'G4/forecast'
    out[f'{k}/times'] = npy(times)
    for i, nm in enumerate(('a', 'b', 'two_c', 'three_d')):
        out[f'{k}/{nm}'] = npy(cs[i])
    out[f'{k}/ts'] = npy(rec['ts'])
    out[f'{k}/dt'] = np.float64(rec['dt'])
    out[f'{k}/pred'] = npy(pred)
    for kk, v in model.state_dict().items():
        out[f'{k}/param/{kk}'] = npy(v)
    # torch_ists wrapper: default method srk, ts = times
    times = torch.linspace(0, 1, L)
    X = make_path(gen, B, L, C)
    coeffs = torch.cat(interp.natural_cubic_spline_coeffs(times, X), dim=-1)
    torch.manual_seed(7)
    func = ists.Diffusion_model(C, H, H, 2, input_option=6, noise_option=17)
    model = ists.NeuralSDE(func, C, H, 2, initial=True).eval()
    with torch.no_grad():
        pred, z = model(coeffs, times)
    k = 'G4/ists'
    out[f'{k}/times'] = npy(times)
    out[f'{k}/coeffs'] = npy(coeffs)
    out[f'{k}/ts'] = npy(rec['ts'])
    out[f'{k}/dt'] = np.float64(rec['dt'])
    out[f'{k}/method'] = np.array(rec['kw']['method'])
    out[f'{k}/pred'] = npy(pred)
    out[f'{k}/z'] = npy(z)
    for kk, v in model.state_dict().items():
        out[f'{k}/param/{kk}'] = npy(v)
    # dt rule of _prepare_sde_solver_kwargs
    for nm, t in (('arange', torch.arange(0., 9.)), ('lin01_20', torch.linspace(0, 1, 20)),
                  ('tiny', torch.tensor([0.0, 1e-4, 2e-4])), ('irr', torch.tensor([0.0, 0.3, 0.35, 1.0]))):
        kw, dt = ref._prepare_sde_solver_kwargs(t, {}, default_method='euler', respect_euler_grid=False)
        out[f'G4/dt/{nm}/times'] = npy(t)
        out[f'G4/dt/{nm}/dt'] = np.float64(dt)


# ------------------------------------------------------------------------------------------------
def _srid2_step(model, t0, h, y, I_k, I_k0):
    """One SRID2 step (torchsde's `srk` for diagonal noise; tableau restated in oracle/sde_oracle.py) on the module's f / g."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))) if os.path.dirname(os.path.dirname(HERE)) not in sys.path else None
    from oracle import sde_oracle as O
    rdt = h ** 0.5
    I_kk = (I_k * I_k - h) / 2
    I_kkk = (I_k ** 3 - 3 * h * I_k) / 6
    fs, gs = [], []
    y1 = y
    for s_ in range(4):
        H0, H1 = y, y
        for j in range(s_):
            H0 = H0 + O.SRK_A0[s_][j] * fs[j] * h + O.SRK_B0[s_][j] * gs[j] * I_k0 / h
            H1 = H1 + O.SRK_A1[s_][j] * fs[j] * h + O.SRK_B1[s_][j] * gs[j] * rdt
        fs.append(model.f(t0 + O.SRK_C0[s_] * h, H0))
        gs.append(model.g(t0 + O.SRK_C1[s_] * h, H1))
        gw = O.SRK_BETA1[s_] * I_k + O.SRK_BETA2[s_] * I_kk / rdt + O.SRK_BETA3[s_] * I_k0 / h + O.SRK_BETA4[s_] * I_kkk / h
        y1 = y1 + O.SRK_ALPHA[s_] * fs[s_] * h + gw * gs[s_]
    return y1


def torch_step_grid_and_solve(model, y0, ts, dt, dW, method, dU=None):
    """torchsde 0.2.5 fixed-step integrate semantics written with torch ops,
    driving the REFERENCE module's f/g.  dW[n] plays bm(t0_n, t1_n) (SRK: dU[n] the space-time Levy integral)."""
    curr_t = ts[0]
    prev_t = ts[0]
    curr_y = prev_y = y0
    ys = [y0]
    n = 0
    for out_t in ts[1:]:
        while curr_t < out_t:
            next_t = min(curr_t + dt, ts[-1])
            prev_t, prev_y = curr_t, curr_y
            h = next_t - curr_t
            I = dW[n]
            if method == 'euler':
                f = model.f(curr_t, curr_y)
                g = model.g(curr_t, curr_y)
                curr_y = curr_y + f * h + g * I
            elif method == 'srk':
                curr_y = _srid2_step(model, curr_t, h, curr_y, I, dU[n])
            else:
                with torch.enable_grad():
                    yy = curr_y.detach().requires_grad_(True)
                    g = model.g(curr_t, yy)
                    v = I ** 2 - h
                    gdg, = torch.autograd.grad(g, yy, grad_outputs=g.detach() * v, allow_unused=True)
                    if gdg is None:   # torchsde: misc.vjp(..., allow_unused=True) -> zeros
                        gdg = torch.zeros_like(curr_y)
                g = g.detach()
                f = model.f(curr_t, curr_y)
                curr_y = curr_y + f * h + g * I + 0.5 * gdg
            curr_t = next_t
            n += 1
        ys.append((curr_t - out_t) / (curr_t - prev_t) * prev_y + (out_t - prev_t) / (curr_t - prev_t) * curr_y)
    return torch.stack(ys), n


def count_steps(ts, dt):
    curr_t = ts[0]
    n = 0
    for out_t in ts[1:]:
        while curr_t < out_t:
            curr_t = min(curr_t + dt, ts[-1])
            n += 1
    return n


def g5(ref, interp, out):
    gen = torch.Generator().manual_seed(404)
    cases = [
        # name, io, no, NL, B, H, C, times, ts, dt, method
        ('lnsde_int', 4, 17, 2, 6, 16, 3, torch.arange(0., 13.), torch.tensor([0., 12.]), 1.0, 'euler'),
        ('gsde_int_multi', 6, 17, 2, 5, 16, 4, torch.arange(0., 21.), torch.tensor([0., 3., 4., 11., 20.]), 1.0, 'euler'),
        ('lsde_lin01', 2, 16, 1, 7, 8, 2, torch.linspace(0, 1, 20), torch.linspace(0, 1, 20), 0.02, 'euler'),
        ('lsde_lin01_dtgrid', 2, 16, 2, 4, 8, 2, torch.linspace(0, 1, 12), torch.linspace(0, 1, 12), None, 'euler'),
        ('nsde_3_18', 3, 18, 2, 6, 8, 5, torch.linspace(1, 10, 10), torch.tensor([1., 4., 7., 10.]), 1.0, 'euler'),
        ('naive_1_18', 1, 18, 2, 4, 8, 3, torch.arange(0., 8.), torch.tensor([0., 7.]), 0.5, 'euler'),
        ('static_1_0', 1, 0, 3, 4, 8, 3, torch.arange(0., 8.), torch.tensor([0., 2.5, 7.]), 1.0, 'euler'),
        ('ctrl_0_13', 0, 13, 2, 4, 8, 3, torch.arange(0., 8.), torch.tensor([0., 7.]), 1.0, 'euler'),
        ('lnsde_long', 4, 17, 2, 3, 16, 3, torch.arange(0., 101.), torch.tensor([0., 100.]), 1.0, 'euler'),
        ('lnsde_milstein', 4, 17, 2, 6, 16, 3, torch.arange(0., 13.), torch.tensor([0., 5., 12.]), 1.0, 'milstein'),
        ('gsde_milstein', 6, 17, 2, 5, 8, 3, torch.linspace(0, 1, 9), torch.linspace(0, 1, 9), None, 'milstein'),
        ('lsde_milstein', 2, 16, 2, 5, 8, 3, torch.arange(0., 9.), torch.tensor([0., 8.]), 1.0, 'milstein'),
        ('y_8_milstein', 5, 8, 2, 5, 8, 3, torch.arange(0., 9.), torch.tensor([0., 8.]), 0.5, 'milstein'),
        # (round 2, appended: the generator stream of the cases above is unchanged)  Milstein through the diffusion nets:
        # torchsde's VJP of g with cotangent g (dW^2 - h), taken by autograd on the reference's g
        ('nsde_3_18_milstein', 3, 18, 2, 6, 8, 5, torch.linspace(1, 10, 10), torch.tensor([1., 4., 7., 10.]), 1.0, 'milstein'),
        ('naive_1_14_milstein', 1, 14, 1, 4, 8, 3, torch.arange(0., 8.), torch.tensor([0., 7.]), 0.5, 'milstein'),
        ('net_4_19_milstein', 4, 19, 2, 5, 16, 3, torch.arange(0., 9.), torch.tensor([0., 3.5, 8.]), 1.0, 'milstein'),
        ('net_2_15_milstein', 2, 15, 2, 4, 8, 3, torch.arange(0., 8.), torch.tensor([0., 7.]), 0.5, 'milstein'),
    ]
    for (name, io, no, NL, B, H, C, times, ts, dt, method) in cases:
        L = times.shape[0]
        X = make_path(gen, B, L, C)
        mask = torch.rand(B, L, C, generator=gen) < 0.2
        X[mask] = float('nan')
        X[:, 0, :] = torch.nan_to_num(X[:, 0, :])
        coeffs = torch.cat(interp.natural_cubic_spline_coeffs(times, X), dim=-1)
        if dt is None:
            _, dt = ref._prepare_sde_solver_kwargs(times, {}, default_method='euler', respect_euler_grid=False)
        torch.manual_seed(11)
        m = ref.Diffusion_model(C, H, H, NL, input_option=io, noise_option=no)
        randomise(m, gen)
        sd = m.state_dict()
        m.set_X(coeffs, times)
        y0 = 0.5 * torch.randn(B, H, generator=gen)
        N = count_steps(ts, dt)
        # increments with the right variance for each step (t1 - t0 in fp32)
        Z = torch.randn(N, B, H, generator=gen)
        hs = []
        curr = ts[0]
        for out_t in ts[1:]:
            while curr < out_t:
                nxt = min(curr + dt, ts[-1])
                hs.append(nxt - curr)
                curr = nxt
        dW = Z * torch.stack(hs).sqrt().view(N, 1, 1)
        with torch.no_grad():
            ys32, n32 = torch_step_grid_and_solve(m, y0, ts, dt, dW, method)
        md = ref.Diffusion_model(C, H, H, NL, input_option=io, noise_option=no).double()
        md.load_state_dict({k: v.double() for k, v in sd.items()})
        md.set_X(coeffs.double(), times.double())
        with torch.no_grad():
            # same fp32 time grid, fp64 arithmetic: feed fp32 ts; f/g see fp32 0-dim t promoted inside
            ys64, n64 = torch_step_grid_and_solve(_F64Times(md), y0.double(), ts, dt, dW.double(), method)
        assert n32 == N and n64 == N
        k = f'G5/{name}'
        out[f'{k}/io_no_nl'] = np.array([io, no, NL])
        out[f'{k}/method'] = np.array(method)
        out[f'{k}/times'] = npy(times)
        out[f'{k}/ts'] = npy(ts)
        out[f'{k}/dt'] = np.float64(dt)
        out[f'{k}/coeffs'] = npy(coeffs)
        out[f'{k}/y0'] = npy(y0)
        out[f'{k}/dW'] = npy(dW)
        out[f'{k}/ys32'] = npy(ys32)
        out[f'{k}/ys64'] = npy(ys64)
        for kk, v in sd.items():
            out[f'{k}/param/{kk}'] = npy(v)
        print(f'G5 {name}: N={N} T={ts.shape[0]} |ys|max={ys64.abs().max():.3f} '
              f'f32-f64 max={float((ys32.double() - ys64).abs().max()):.2e}')


class _F64Times:
    """Feed the fp64 module fp64 copies of the fp32 step times (the step grid itself stays fp32)."""

    def __init__(self, m):
        self.m = m

    def f(self, t, y):
        return self.m.f(t.double(), y)

    def g(self, t, y):
        return self.m.g(t.double(), y)


def save(name, d):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **d)
    print(f'wrote {path}: {len(d)} arrays, {os.path.getsize(path) / 1024:.1f} KiB')


def main():
    torch.set_num_threads(1)
    ref, interp, tsde = load_reference('benchmark_classification')
    ists = load_torch_ists()
    ref_fc, _, _ = load_reference('benchmark_forecasting')
    # reload classification last so sys.modules['controldiffeq'] matches `interp`
    ref, interp, tsde = load_reference('benchmark_classification')
    d = {}
    g1_g2(interp, d)
    save('spline.npz', d)
    d = {}
    g3(ref, ref_fc, ists, interp, d)
    save('fg.npz', d)
    d = {}
    g4(ref, ref_fc, ists, interp, tsde, d)
    save('wrapper.npz', d)
    d = {}
    g5(ref, interp, d)
    save('traj.npz', d)


if __name__ == '__main__':
    main()
