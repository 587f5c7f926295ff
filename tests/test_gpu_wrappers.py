"""Inference-time fusion around the solve in the wrappers (modules._SDEHead): the initial state evaluated inside the solve's
prepare launch (snsde_solve.z0_weight / z0_bias; reference: NeuralSDE._prepare_initial_state, benchmark_classification/
models_sde/neuralsde.py:63-69) and the readout head in one launch (snsde_readout_head; neuralsde.py:59-61,119,
benchmark_forecasting/models_sde/neuralsde.py:153-155,186, torch_ists nsde_model.py).  Reference for both: the same module
evaluated with tensor ops."""
import numpy as np
import pytest
import torch

import stable_neural_sdes_amd as S
from stable_neural_sdes_amd import engine
from tests.helpers import make_problem

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0') if torch.cuda.is_available() else None


def _heads(H, Hh, O):
    nn = torch.nn
    return {
        'classification': nn.Sequential(nn.Linear(H, Hh), nn.BatchNorm1d(Hh), nn.ReLU(), nn.Dropout(0.1), nn.Linear(Hh, O)),
        'forecasting': nn.Sequential(nn.Linear(H, Hh), nn.ReLU(), nn.Linear(Hh, O)),
        'ists': nn.Sequential(nn.Tanh(), nn.Linear(H, Hh), nn.ReLU(), nn.Linear(Hh, O)),
        'plain_bn': nn.Sequential(nn.Linear(H, Hh), nn.BatchNorm1d(Hh, affine=False), nn.ReLU(), nn.Linear(Hh, O, bias=False)),
    }


@pytest.mark.parametrize('kind', ['classification', 'forecasting', 'ists', 'plain_bn'])
@pytest.mark.parametrize('shape', [((1024,), 128, 128, 1), ((37,), 64, 64, 3), ((5, 10), 256, 256, 14), ((9,), 40, 72, 5),
                                   ((3,), 16, 300, 2), ((130,), 200, 512, 7), ((2050,), 128, 128, 20)])
def test_readout_head_matches_the_module(kind, shape):
    lead, H, Hh, O = shape
    torch.manual_seed(hash((kind, H, Hh)) % 1000)
    head = _heads(H, Hh, O)[kind].to(DEV)
    for m in head:
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_(); m.running_var.uniform_(0.3, 2.0)
            if m.weight is not None:
                m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_()
    head.eval()
    layers = engine.head_layers(head)
    assert (layers is not None) == (Hh <= 128 and H <= 256)
    mods = [m for m in head if isinstance(m, (torch.nn.Linear, torch.nn.BatchNorm1d))]
    layers = (isinstance(head[0], torch.nn.Tanh), mods[0], mods[1] if len(mods) == 3 else None, mods[-1])   # the kernel itself: any size
    x = torch.randn(*lead, H, device=DEV) * 2
    with torch.no_grad():
        got = engine.readout_head(x, layers)
        ref = head(x.reshape(-1, H)).reshape(*lead, O)
        ref64 = head.double()(x.reshape(-1, H).double()).reshape(*lead, O)
    scale = float(ref64.abs().max()) + 1e-6
    assert got.shape == ref.shape
    assert float((got.double() - ref64).abs().max()) / scale < 2e-6
    # and no further from the float64 value than the library-GEMM float32 evaluation is (x4)
    assert float((got.double() - ref64).abs().max()) <= 4 * float((ref.double() - ref64).abs().max()) + 1e-6 * scale


def test_head_structure_gate():
    nn = torch.nn
    head = _heads(8, 8, 2)['classification']
    assert engine.head_layers(head) is None                     # training mode: batch statistics and dropout
    head.eval()
    assert engine.head_layers(head) is not None
    assert engine.head_layers(nn.Sequential(nn.Linear(8, 8), nn.Tanh(), nn.Linear(8, 2))) is None
    assert engine.head_layers(nn.Sequential(nn.Linear(8, 8), nn.ReLU(), nn.Linear(8, 8), nn.ReLU(), nn.Linear(8, 2))) is None
    assert engine.head_layers(nn.Linear(8, 2)) is None
    assert engine.head_layers(nn.Sequential(nn.Linear(8, 130), nn.ReLU(), nn.Linear(130, 2))) is None      # wide heads: library GEMMs


@pytest.mark.parametrize('cfg', [(4, 17, 21, 32, 5, 9, 'euler'), (1, 18, 13, 64, 3, 8, 'euler'), (6, 17, 9, 24, 3, 8, 'euler'),
                                 (2, 16, 11, 32, 2, 12, 'srk'), (4, 17, 9, 128, 21, 9, 'milstein'), (0, 4, 9, 16, 40, 8, 'euler')])
def test_initial_state_inside_the_solve(cfg):
    """options['z0_linear']: y0 = initial_network(X(ts[0])) computed by the solve's prepare launch (MFMA families) or a
    stand-alone launch (generic / padded-free shapes) - same trajectory as handing the solve the tensor-op value."""
    io, no, B, H, C, L, method = cfg
    pr = make_problem(77, io, no, 2, B, H, C, L)
    m = S.Diffusion_model(C, H, H, 2, input_option=io, noise_option=no)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pr['params'].items()})
    m = m.to(DEV)
    times = torch.from_numpy(pr['times']).to(DEV)
    m.set_X(torch.from_numpy(pr['coeffs']).to(DEV), times)
    torch.manual_seed(3)
    lin = torch.nn.Linear(C, H).to(DEV)
    with torch.no_grad():
        z0 = lin(m.X.evaluate(times[0]))
        ref = S.sdeint(m, z0, times, method=method, dt=0.5, options={'seed': 11})
        got = S.sdeint(m, torch.empty(B, H, device=DEV), times, method=method, dt=0.5, options={'seed': 11, 'z0_linear': lin})
    assert float((got[0] - z0).abs().max()) < 1e-5 * (float(z0.abs().max()) + 1)
    scale = float(ref.abs().max()) + 1e-6
    assert float((got - ref).abs().max()) / scale < 2e-4
    # with gradients enabled the same option is honoured through tensor ops (differentiable w.r.t. initial_network)
    ys = S.sdeint(m, torch.empty(B, H, device=DEV), times, method=method, dt=0.5, options={'seed': 11, 'z0_linear': lin})
    ys[-1].sum().backward()
    assert lin.weight.grad is not None and float(lin.weight.grad.abs().max()) > 0


@pytest.mark.parametrize('name', ['neurallnsde', 'naivesde', 'neuralgsde', 'neurallsde'])
def test_classification_wrapper_inference_equals_its_tensor_op_form(name, monkeypatch):
    B, H, C, L = 50, 32, 5, 12
    pr = make_problem(5, 4, 17, 2, B, H, C, L, nan_frac=0.2)
    times = torch.from_numpy(pr['times']).to(DEV)
    coeffs = torch.from_numpy(pr['coeffs']).to(DEV)
    fi = torch.randint(0, L, (B,), device=DEV)
    torch.manual_seed(1)
    model, _ = S.make_sde_model(name, C, 3, H, H, 2, initial=True)
    model = model.to(DEV)
    model.linear[1].running_mean.normal_(); model.linear[1].running_var.uniform_(0.5, 2.0)
    model.eval()
    calls = []
    orig = engine.readout_head
    monkeypatch.setattr(engine, 'readout_head', lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    with torch.no_grad():
        fused = model(times, [coeffs], fi, options={'seed': 9})
    assert calls, 'the fused head did not run'
    monkeypatch.setattr(engine, 'head_layers', lambda seq: None)
    monkeypatch.setattr(type(model), '_initial_state', lambda self, t, z0, kw: (self._prepare_initial_state(t, z0), kw))
    with torch.no_grad():
        plain = model(times, [coeffs], fi, options={'seed': 9})
    scale = float(plain.abs().max()) + 1e-6
    assert float((fused - plain).abs().max()) / scale < 2e-4


def test_forecasting_and_ists_wrappers_inference_equal_their_tensor_op_form(monkeypatch):
    B, H, C, L = 20, 64, 6, 10
    pr = make_problem(6, 4, 17, 2, B, H, C, L)
    times = torch.from_numpy(pr['times']).to(DEV)
    coeffs = torch.from_numpy(pr['coeffs']).to(DEV)
    torch.manual_seed(2)
    f1 = S.Diffusion_model(C, H, H, 2, input_option=4, noise_option=17)
    fc = S.NeuralSDE_forecasting(f1, C, 4, H, C).to(DEV).eval()
    f2 = S.Diffusion_model(C, H, H, 2, input_option=6, noise_option=17)
    ists = S.IstsNeuralSDE(f2, C, H, 3).to(DEV).eval()
    four = [coeffs[..., k * C:(k + 1) * C].contiguous() for k in range(4)]
    with torch.no_grad():
        a = fc(times, four, None, options={'seed': 4})
        b, zb = ists(coeffs, times, method='euler', options={'seed': 4})
    monkeypatch.setattr(engine, 'head_layers', lambda seq: None)
    for cls in (type(fc), type(ists)):
        monkeypatch.setattr(cls, '_initial_state', lambda self, t, z0, kw: (self._prepare_initial_state(t, z0), kw))
    with torch.no_grad():
        a0 = fc(times, four, None, options={'seed': 4})
        b0, zb0 = ists(coeffs, times, method='euler', options={'seed': 4})
    for x, y in ((a, a0), (b, b0), (zb, zb0)):
        assert x.shape == y.shape
        assert float((x - y).abs().max()) / (float(y.abs().max()) + 1e-6) < 2e-4
