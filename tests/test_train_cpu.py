"""The training driver (stable-neural-sdes_amd/train.py = the engine's equivalent of the reference's common_sde.py) on
CPU: regulariser and ranking metrics against their definitions (sklearn), the epoch loop end to end on a small synthetic
classification / regression problem through the tensor-op path, and the one-process-per-rank variant on two gloo ranks."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stable_neural_sdes_amd as S  # noqa: E402
from stable_neural_sdes_amd import train as T  # noqa: E402


def synthetic_loader(n, L, C, classes, batch, seed=0, shard=None):
    """(coeffs, y, final_index) batches: the label is a function of the series' drift."""
    rng = np.random.default_rng(seed)
    times = np.arange(L, dtype=np.float32)
    slope = rng.standard_normal((n, 1, C)).astype(np.float32) * 0.2
    X = slope * times[None, :, None] + 0.05 * rng.standard_normal((n, L, C)).astype(np.float32)
    X[:, :, 0] = times[None]
    s = slope[:, 0, 1]
    if classes is None:
        y = torch.from_numpy(np.stack([s, -s], 1).astype(np.float32))
    elif classes == 2:
        y = torch.from_numpy((s > 0).astype(np.float32))
    else:
        y = torch.from_numpy(np.digitize(s, np.quantile(s, np.linspace(0, 1, classes + 1)[1:-1])).astype(np.int64))
    coeffs = torch.cat(S.controldiffeq.natural_cubic_spline_coeffs(torch.from_numpy(times), torch.from_numpy(X)), dim=-1)
    fi = torch.full((n,), L - 1, dtype=torch.int64)
    if shard is not None:
        lo, hi = S.sharding.shard_rows(n, shard[1], shard[0])
        coeffs, y, fi = coeffs[lo:hi], y[lo:hi], fi[lo:hi]
    ds = torch.utils.data.TensorDataset(coeffs, y, fi)
    return torch.from_numpy(times), torch.utils.data.DataLoader(ds, batch_size=batch, shuffle=False)


def test_weight_regularisation_is_the_sum_of_per_tensor_norms():
    torch.manual_seed(0)
    field = S.Diffusion_model(3, 8, 8, 2, input_option=4, noise_option=17)
    base = torch.nn.functional.mse_loss
    a, b = torch.randn(5, 2), torch.randn(5, 2)
    for mode, p in (('l2', 2), ('l1', 1)):
        got = T.add_weight_regularisation(base, field, scaling=0.01, mode=mode)(a, b)
        want = base(a, b) + sum(0.01 * q.norm(p) for q in field.parameters() if q.requires_grad)   # common_sde.py:16-23
        assert torch.allclose(got, want, rtol=1e-6)
    field.zero_grad()
    T.add_weight_regularisation(base, field)(a, b).backward()
    w = field.linear_out.weight
    assert torch.allclose(w.grad, 0.01 * w.detach() / w.detach().norm(), rtol=1e-5, atol=1e-8)


def test_ranking_metrics_match_sklearn_including_ties():
    import sklearn.metrics
    rng = np.random.default_rng(3)
    for n, ties in ((50, False), (200, True), (7, True)):
        s = rng.standard_normal(n)
        if ties:
            s = np.round(s, 1)
        y = (rng.random(n) < 0.4).astype(np.float64)
        y[0], y[1] = 0.0, 1.0
        auc, ap = T.binary_ranking_metrics(torch.from_numpy(s), torch.from_numpy(y))
        assert abs(auc - sklearn.metrics.roc_auc_score(y, s)) < 1e-12
        assert abs(ap - sklearn.metrics.average_precision_score(y, s)) < 1e-12
    assert np.isnan(T.binary_ranking_metrics(torch.zeros(4), torch.zeros(4))[0])


@pytest.mark.parametrize('classes', [2, 3, None])
def test_main_trains_and_reports_like_the_reference_loop(classes, tmp_path):
    torch.manual_seed(1)
    L, C, H = 6, 3, 8
    times, train = synthetic_loader(48, L, C, classes, 16, seed=1)
    _, val = synthetic_loader(24, L, C, classes, 24, seed=2)
    _, test = synthetic_loader(24, L, C, classes, 24, seed=3)
    out_ch = 1 if classes == 2 else (2 if classes is None else classes)
    factory = T.make_model('neurallnsde', C, out_ch, H, H, 2, use_intensity=False, initial=True)
    lines = []
    res = T.main('unit', 'neurallnsde', times, train, val, test, 'cpu', factory, classes, 3, 5e-3,
                 dict(method='euler', options={'seed': 7}), 'valloss', results_dir=str(tmp_path), log=lines.append)
    assert len(res.history) == 3 and [h.epoch for h in res.history] == [0, 1, 2]
    assert res.train_metrics.dataset_size == 48 and res.test_metrics.dataset_size == 24
    assert np.isfinite(res.train_metrics.loss) and res.parameters == T.count_parameters(res.model)
    if classes is not None:
        assert res.val_metrics.confusion.shape == (classes, classes) and res.val_metrics.confusion.sum() == 24
        assert 0.0 <= res.val_metrics.accuracy <= 1.0
    if classes == 2:
        assert 0.0 <= res.test_metrics.auroc <= 1.0 and 'average_precision' in res.test_metrics
    assert any(line.startswith('Epoch: 2') for line in lines)
    assert os.path.exists(res.saved_to) and res.saved_to.endswith(os.path.join('unit', '0'))
    # the model holds the parameters of its best validation epoch: re-evaluating reproduces that epoch's validation score
    best = max(res.history, key=lambda h: h.val_metrics.accuracy if classes is not None else -h.val_metrics.loss)
    if classes is None:
        assert abs(res.val_metrics.loss - best.val_metrics.loss) < 1e-6
    with pytest.raises(ValueError):
        T.train_loop(train, val, res.model, times, None, None, 1, classes, 'cpu', {}, 'bogus')


def test_plateau_scheduler_follows_the_chosen_metric(monkeypatch):
    torch.manual_seed(2)
    times, train = synthetic_loader(16, 5, 2, 2, 16, seed=4)
    model, field = S.make_sde_model('neurallsde', 2, 1, 4, 4, 1)
    model = T.SqueezeEnd(model)
    opt = torch.optim.SGD(model.parameters(), lr=1e-3)
    loss = T.add_weight_regularisation(torch.nn.BCEWithLogitsLoss(), field)
    flat = T.AttrDict(loss=1.0, accuracy=0.5, auroc=0.5, dataset_size=16)       # a metric that never improves
    monkeypatch.setattr(T, 'evaluate_metrics', lambda *a, **k: T.AttrDict(flat))
    hist = T.train_loop(train, train, model, times, opt, loss, 9, 2, 'cpu', dict(method='euler', options={'seed': 1}),
                        'valauc', plateau_terminate=100)
    lrs = [h.lr for h in hist]
    # ReduceLROnPlateau(patience=5, mode='max'): best at epoch 0, epochs 1..6 do not improve -> cut by 10 at epoch 6
    assert len(hist) == 9 and lrs[:6] == [1e-3] * 6 and abs(lrs[6] - 1e-4) < 1e-12 and lrs[8] == lrs[6]
    # ... and the loop stops `plateau_terminate` epochs after the last improvement of the training loss
    hist = T.train_loop(train, train, model, times, opt, loss, 9, 2, 'cpu', dict(method='euler', options={'seed': 1}),
                        'none', plateau_terminate=3)
    assert [h.epoch for h in hist] == [0, 1, 2, 3, 4]


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _ddp_main(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.manual_seed(5)                               # identical initialisation on every rank
        times, train = synthetic_loader(40, 6, 3, 2, 10, seed=1, shard=(rank, world))
        _, val = synthetic_loader(20, 6, 3, 2, 10, seed=2, shard=(rank, world))
        factory = T.make_model('neurallnsde', 3, 1, 8, 8, 2)
        res = T.main(None, 'neurallnsde', times, train, val, val, 'cpu', factory, 2, 2, 5e-3, dict(method='euler'),
                     'valauc', log=None)
        flat = torch.cat([p.detach().reshape(-1) for p in res.model.parameters()])
        q.put((rank, res.train_metrics.dataset_size, res.val_metrics.dataset_size, res.train_metrics.loss,
               res.val_metrics.auroc, float(res.val_metrics.confusion.sum()), flat.numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_main_under_two_gloo_ranks_trains_one_model_and_reduces_metrics():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = got
    assert a[1] == b[1] == 40 and a[2] == b[2] == 20 and a[5] == b[5] == 20.0       # global sizes on every rank
    assert a[3] == b[3] and a[4] == b[4]                                            # reduced metrics agree
    assert np.array_equal(a[6], b[6])                                               # one model: DDP kept the ranks in step


def test_package_training_loop_reproduces_the_reference_common_sde_loop_trace():
    """tests/golden/dropin.npz `trainloop/*` was recorded by running the REFERENCE's own benchmark_classification/
    common_sde.py:_train_loop / _evaluate_metrics (and its NeuralSDE / Diffusion_model) over this package's torchsde /
    torchcde / controldiffeq mirrors on CPU (tools/check_reference_dropin.py).  The package's own model classes and
    training loop, on the same problem with the same seeds, must land on the same per-epoch metrics and final weights."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('check_reference_dropin', os.path.join(ROOT, 'tools', 'check_reference_dropin.py'))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    gold = np.load(os.path.join(ROOT, 'tests', 'golden', 'dropin.npz'))
    tl = tool.trainloop_problem()
    torch.manual_seed(21)
    func = S.Diffusion_model(tl['C'], tl['H'], tl['H'], 2, input_option=4, noise_option=17)
    model = T.SqueezeEnd(S.NeuralSDE(func, tl['C'], tl['H'], 1, initial=True))
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-2, weight_decay=1e-2 * 0.01)
    loss_fn = T.add_weight_regularisation(torch.nn.functional.binary_cross_entropy_with_logits, func)
    torch.manual_seed(22)
    hist = T.train_loop(tl['train'], tl['val'], model, tl['times'], optimizer, loss_fn, 2, 2, 'cpu', {}, 'trainloss')
    trace = np.array([[h.train_metrics.loss, h.train_metrics.accuracy, h.train_metrics.auroc, h.val_metrics.loss,
                       h.val_metrics.accuracy, h.val_metrics.auroc] for h in hist], dtype=np.float64)
    np.testing.assert_allclose(trace, gold['trainloop/trace'], rtol=2e-5, atol=2e-6)
    sd = model.state_dict()
    for k in sd:
        np.testing.assert_allclose(sd[k].numpy(), gold['trainloop/final_sd/' + k], rtol=2e-4, atol=2e-6, err_msg=k)
