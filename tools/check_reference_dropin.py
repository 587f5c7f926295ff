#!/usr/bin/env python3
"""Build-container check of the zero-diff route, and generator of the G6 fixture tests/golden/dropin.npz.

Runs HERE (where /root/reference exists; CPU only): registers the engine's `torchsde` / `torchcde` / `controldiffeq`
mirrors (`stable_neural_sdes_amd.install()`, no stubs), then imports the reference's OWN source on top of them

    benchmark_classification/models_sde/neuralsde.py     (NeuralSDE, Diffusion_model)
    benchmark_forecasting/models_sde/neuralsde.py        (NeuralSDE_forecasting)
    torch-ists/torch_ists/diff_module/NSDE/              (package __init__: nsde_model.NeuralSDE + latent_sde.LatentSDE,
                                                          which subclasses torchsde.SDEIto and calls sdeint_adjoint)

by file path (their parent packages import torchdiffeq / signatory, which this image lacks), checks that the engine
recognises the reference's Diffusion_model as a fast-path module, and runs the three wrappers' forward() with a replayed
Brownian motion (`bm=` is forwarded to sdeint by the wrappers' **kwargs).  On CPU tensors `sdeint` is the tensor-op loop
calling the REFERENCE's f / g, so the recorded outputs are: reference wrapper bookkeeping + reference vector field +
this repo's restatement of torchsde's fixed-step scheme.  The fixture holds inputs, state_dicts, increments and outputs
only (no reference source); tests/test_gpu_parity.py::test_dropin_fixture_* loads each state_dict into a locally defined
module with the reference's attribute / parameter names and requires the HIP path to reproduce the outputs.

usage: python tools/check_reference_dropin.py [--write]
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = '/root/reference'

import stable_neural_sdes_amd as S  # noqa: E402


def load(name, path, package=False):
    kw = dict(submodule_search_locations=[os.path.dirname(path)]) if package else {}
    spec = importlib.util.spec_from_file_location(name, path, **kw)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class ReplayBM:
    """bm(ta, tb[, return_U]) replayed from arrays, in call order."""
    levy_area_approximation = 'space-time'

    def __init__(self, dW, dU=None):
        self.dW, self.dU, self.n = dW, dU, 0

    def __call__(self, ta, tb, return_U=False):
        i, self.n = self.n, self.n + 1
        return (self.dW[i], self.dU[i]) if return_U else self.dW[i]


def make_inputs(rng, B, L, C, times):
    X = (rng.standard_normal((B, L, C)) * 0.3).cumsum(1).astype(np.float32)
    X[:, :, 0] = times[None, :]
    mask = rng.random((B, L, C)) < 0.2
    mask[:, :, 0] = False
    X[mask] = np.nan
    coeffs = torch.cat(S.controldiffeq.natural_cubic_spline_coeffs(torch.from_numpy(times), torch.from_numpy(X)), dim=-1)
    return coeffs


def trainloop_problem():
    """Small synthetic binary classification problem for the training-loop replay (also imported by tests/test_train_cpu.py):
    (coeffs, label, final_index) loaders in common_sde's batch format."""
    n, L, C, H = 24, 8, 3, 8
    rng = np.random.default_rng(5)
    times = np.arange(L, dtype=np.float32)
    slope = rng.standard_normal((n, 1, C)).astype(np.float32) * 0.2
    X = slope * times[None, :, None] + 0.05 * rng.standard_normal((n, L, C)).astype(np.float32)
    X[:, :, 0] = times[None]
    y = torch.from_numpy((slope[:, 0, 1] > 0).astype(np.float32))
    coeffs = torch.cat(S.controldiffeq.natural_cubic_spline_coeffs(torch.from_numpy(times), torch.from_numpy(X)), dim=-1)
    fi = torch.from_numpy(rng.integers(3, L, size=n).astype(np.int64))
    mk = lambda lo, hi: torch.utils.data.DataLoader(torch.utils.data.TensorDataset(coeffs[lo:hi], y[lo:hi], fi[lo:hi]),
                                                    batch_size=8, shuffle=False)
    return dict(times=torch.from_numpy(times), train=mk(0, 16), val=mk(16, 24), C=C, H=H)


def main():
    S.install()
    import torchsde
    assert torchsde.sdeint is S.sdeint and hasattr(torchsde, 'SDEIto') and hasattr(torchsde, 'sdeint_adjoint')
    cls = load('ref_cls_neuralsde', f'{REF}/benchmark_classification/models_sde/neuralsde.py')
    fc = load('ref_fc_neuralsde', f'{REF}/benchmark_forecasting/models_sde/neuralsde.py')
    nsde = load('ref_ists_nsde', f'{REF}/torch-ists/torch_ists/diff_module/NSDE/__init__.py', package=True)
    assert issubclass(nsde.LatentSDE, torchsde.SDEIto)

    out = {}
    rng = np.random.default_rng(20260928)
    B, H, C, L, NL = 6, 32, 5, 9, 2

    # ---- classification wrapper: (4, 17) LNSDE, integer grid, per-row final index ---------------------------------
    torch.manual_seed(11)
    times = np.arange(L, dtype=np.float32)
    func = cls.Diffusion_model(C, H, H, NL, input_option=4, noise_option=17)
    assert S.engine.recognise(func) is None or True      # (coeffs are attached by set_X inside forward)
    model = cls.NeuralSDE(func, C, H, 3, initial=True).eval()
    coeffs = make_inputs(rng, B, L, C, times)
    final_index = torch.tensor([8, 3, 3, 5, 0, 8])
    dW = (rng.standard_normal((L - 1, B, H)).astype(np.float32))          # dt = 1: N = L - 1 unit steps
    with torch.no_grad():
        y = model(torch.from_numpy(times), (coeffs,), final_index, bm=ReplayBM(torch.from_numpy(dW)))
    rec = S.engine.recognise(func)
    assert rec is not None, 'the reference Diffusion_model must satisfy the fast-path contract'
    out.update({'cls/times': times, 'cls/coeffs': coeffs.numpy(), 'cls/final_index': final_index.numpy(), 'cls/dW': dW,
                'cls/out': y.numpy(), 'cls/dims': np.array([B, H, C, L, NL, 4, 17, 3])})
    out.update({'cls/sd/' + k: v.numpy() for k, v in model.state_dict().items()})

    # ---- forecasting wrapper: every knot an output, decoder on the last output_time states -------------------------
    torch.manual_seed(12)
    func = fc.Diffusion_model(C, H, H, NL, input_option=4, noise_option=17)
    model = fc.NeuralSDE_forecasting(func, C, 3, H, 2, initial=True).eval()
    a, b, c2, d3 = (coeffs[..., k * C:(k + 1) * C] for k in range(4))
    dW = rng.standard_normal((L - 1, B, H)).astype(np.float32)
    with torch.no_grad():
        y = model(torch.from_numpy(times), (a, b, c2, d3), None, bm=ReplayBM(torch.from_numpy(dW)))
    out.update({'fc/times': times, 'fc/coeffs': coeffs.numpy(), 'fc/dW': dW, 'fc/out': y.numpy(),
                'fc/dims': np.array([B, H, C, L, NL, 4, 17, 2, 3])})
    out.update({'fc/sd/' + k: v.numpy() for k, v in model.state_dict().items()})

    # ---- torch_ists wrapper: linspace grid, default method srk (I_k and I_k0 replayed), outputs at every knot --------
    torch.manual_seed(13)
    times2 = np.linspace(0, 1, L).astype(np.float32)
    func = nsde.Diffusion_model(C, H, H, NL, input_option=6, noise_option=17)
    model = nsde.NeuralSDE(func, C, H, 2, initial=True).eval()
    coeffs2 = make_inputs(rng, B, L, C, times2)
    grid = S.engine.StepGrid(times2, max(float(np.diff(times2).min()), 1e-3), times2, None)
    h = (grid.t1 - grid.t0).astype(np.float32)
    dW = rng.standard_normal((grid.N, B, H)).astype(np.float32) * np.sqrt(h)[:, None, None]
    xi = rng.standard_normal((grid.N, B, H)).astype(np.float32)
    dU = h[:, None, None] * (0.5 * dW + np.sqrt(h / 12)[:, None, None] * xi)
    with torch.no_grad():
        y, z = model(coeffs2, torch.from_numpy(times2), bm=ReplayBM(torch.from_numpy(dW), torch.from_numpy(dU)))
    out.update({'ists/times': times2, 'ists/coeffs': coeffs2.numpy(), 'ists/dW': dW, 'ists/dU': dU, 'ists/out': y.numpy(),
                'ists/z': z.numpy(), 'ists/dims': np.array([B, H, C, L, NL, 6, 17, 2])})
    out.update({'ists/sd/' + k: v.numpy() for k, v in model.state_dict().items()})

    # ---- LatentSDE imports and integrates over the mirrors (names= / sdeint_adjoint; generic tensor-op path) ---------
    torch.manual_seed(14)
    lat = nsde.LatentSDE(C, 8, 8, 2)
    assert lat.sde_type == 'ito' and lat.noise_type == 'diagonal'
    ys = torchsde.sdeint_adjoint(lat, torch.zeros(3, 8), torch.tensor([0.0, 0.5, 1.0]), dt=0.25, method='euler',
                                 names={'drift': 'f_aug', 'diffusion': 'g_aug'})
    assert ys.shape == (3, 3, 8) and bool(torch.isfinite(ys).all())
    # the reference class itself is what fields.compose_latent recognises (the split solve of DESIGN 3.8b on CUDA), and its own
    # forward (spline start, sdeint_adjoint with names, KL) runs over the mirrors
    from stable_neural_sdes_amd import fields as _fields
    cf = _fields.compose_latent(lat, {'drift': 'f_aug', 'diffusion': 'g_aug'}, 8)
    assert cf is not None and cf.parts['latent'] == 7 and cf.model.hidden_channels == 16
    lt = torch.linspace(0, 1, 6)
    lx = torch.cumsum(0.2 * torch.randn(3, 6, C), dim=1)
    lcoeffs = S.torchcde.hermite_cubic_coefficients_with_backward_differences(lx, lt)
    lout, llat, lkl = lat(lcoeffs, lt, method='euler')
    assert lout.shape == (3, 6, 8) and llat.shape == (3, 6, 7) and lkl.dim() == 0 and bool(torch.isfinite(lkl))

    # ---- the reference's TRAINING LOOP itself (benchmark_classification/common_sde.py:_train_loop / _evaluate_metrics) over the
    #      mirrors: two epochs on a small synthetic binary problem, CPU.  common_sde.py does `import models_sde`, whose package
    #      __init__ pulls torchdiffeq-based baselines this image lacks, so the name is bound to a module holding the classes of
    #      models_sde/neuralsde.py loaded above; the loop, the metrics and the model are the reference's own code.
    import types
    ms = types.ModuleType('models_sde')
    ms.NeuralSDE, ms.Diffusion_model = cls.NeuralSDE, cls.Diffusion_model
    sys.modules['models_sde'] = ms
    common = load('ref_common_sde', f'{REF}/benchmark_classification/common_sde.py')
    tl = trainloop_problem()
    torch.manual_seed(21)
    func = cls.Diffusion_model(tl['C'], tl['H'], tl['H'], 2, input_option=4, noise_option=17)
    model = common._SqueezeEnd(cls.NeuralSDE(func, tl['C'], tl['H'], 1, initial=True))
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-2, weight_decay=1e-2 * 0.01)
    loss_fn = common._add_weight_regularisation(torch.nn.functional.binary_cross_entropy_with_logits, func)
    torch.manual_seed(22)          # the Brownian increments of the CPU path are drawn from torch's global generator
    hist = common._train_loop(tl['train'], tl['val'], model, tl['times'], optimizer, loss_fn, 2, 2, 'cpu', {}, 'trainloss')
    trace = np.array([[h.train_metrics.loss, h.train_metrics.accuracy, h.train_metrics.auroc, h.val_metrics.loss,
                       h.val_metrics.accuracy, h.val_metrics.auroc] for h in hist], dtype=np.float64)
    assert trace.shape == (2, 6) and np.isfinite(trace).all()
    out['trainloop/trace'] = trace
    out.update({'trainloop/final_sd/' + k: v.numpy() for k, v in model.state_dict().items()})
    print('reference common_sde._train_loop over install(): epochs', len(hist), 'train loss', trace[:, 0])

    path = os.path.join(ROOT, 'tests', 'golden', 'dropin.npz')
    if '--write' in sys.argv:
        np.savez_compressed(path, **out)
        print('wrote', path, os.path.getsize(path), 'bytes')
    else:
        old = np.load(path)
        for k, v in out.items():
            np.testing.assert_array_equal(old[k], v, err_msg=k)
        print('fixture reproduced bit-for-bit:', len(out), 'arrays')
    print('reference drop-in check ok: neuralsde.py x3 + NSDE package import over the mirrors; fast-path contract recognised')


if __name__ == '__main__':
    main()
