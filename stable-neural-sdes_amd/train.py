"""Training driver around the fused solver: the engine's equivalent of the reference's `common_sde.py`
(benchmark_classification/common_sde.py:16-23 regulariser, :49-92 metrics, :107-216 loop, :248-298 main;
benchmark_forecasting/common_sde.py for the regression variant).  The reference's own file also runs unchanged on
`stable_neural_sdes_amd.install()`; this module is the same loop written for one-process-per-GPU training:

  * the weight regulariser  scaling * sum_p ||p||  is ONE multi-tensor norm launch over the vector field's parameters
    (torch._foreach_norm) instead of one reduction + one add per parameter tensor per batch;
  * evaluation keeps every running sum (loss, correct count, confusion matrix, the scores for AUROC / average precision)
    on the device and synchronises with the host ONCE per pass over a dataloader instead of once per batch; AUROC and
    average precision are computed on the device from one sort (ties handled as sklearn does);
  * under an initialised process group (one rank per GPU) the model is wrapped in DistributedDataParallel (RCCL
    all-reduce of gradients, bucketed by DDP), metric sums are all-reduced, and the solver's Philox row offsets default
    to rank * local batch (torchsde._sdeint_hip), so ranks never integrate against the same Brownian rows;
  * the best model is restored with in-place copies, which keeps the engine's parameter arena (engine.flatten_params)
    valid; the reference re-points `parameter.data`, which the arena also survives (address check) at the price of one
    re-flatten.

Same entry points and semantics as the reference: `main(...)`, `make_model(...)`, `_train_loop`, `_evaluate_metrics`,
`_add_weight_regularisation` (Adam with weight_decay = 0.01 lr, ReduceLROnPlateau(patience=5) on the chosen metric,
termination after 50 epochs without improvement of the training loss / accuracy, best model by validation accuracy).
"""
import copy
import json
import math
import os

import numpy as np
import torch

from . import modules


class AttrDict(dict):
    def __setattr__(self, key, value):
        self[key] = value

    def __getattr__(self, item):
        try:
            return self[item]
        except KeyError as exc:
            raise AttributeError(item) from exc


def add_weight_regularisation(loss_fn, regularise_parameters, scaling=0.01, mode='l2'):
    """loss + scaling * sum over trainable tensors of ||p||_2 (classification benchmark) or ||p||_1 (`mode='l1'`,
    forecasting benchmark).  One multi-tensor norm launch."""
    if mode not in ('l1', 'l2', None):
        raise ValueError("mode must be 'l1', 'l2' or None")

    def new_loss_fn(pred_y, true_y):
        total = loss_fn(pred_y, true_y)
        ps = [p for p in regularise_parameters.parameters() if p.requires_grad]
        if mode is None or not ps:
            return total
        norms = torch._foreach_norm(ps, 1 if mode == 'l1' else 2)
        return total + scaling * torch.stack(norms).sum()
    return new_loss_fn


class SqueezeEnd(torch.nn.Module):
    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, *args, **kwargs):
        return self.model(*args, **kwargs).squeeze(-1)


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def binary_ranking_metrics(scores, labels):
    """(AUROC, average precision) of binary labels from one descending sort on the scores' device; tied scores form one
    threshold (the convention of sklearn.metrics.roc_auc_score / average_precision_score)."""
    scores = scores.reshape(-1).double()
    labels = labels.reshape(-1).double()
    n_pos, n = labels.sum(), labels.numel()
    n_neg = n - n_pos
    if n == 0 or float(n_pos) == 0.0 or float(n_neg) == 0.0:
        return float('nan'), float('nan')
    order = torch.argsort(scores, descending=True, stable=True)
    s, y = scores[order], labels[order]
    last = torch.ones_like(s, dtype=torch.bool)          # last element of every run of equal scores
    last[:-1] = s[1:] != s[:-1]
    tp = torch.cumsum(y, 0)[last]
    fp = torch.cumsum(1.0 - y, 0)[last]
    zero = torch.zeros(1, dtype=tp.dtype, device=tp.device)
    tp0, fp0 = torch.cat([zero, tp]), torch.cat([zero, fp])
    auroc = torch.trapz(tp0 / n_pos, fp0 / n_neg)
    precision = tp / (tp + fp)
    ap = ((tp0[1:] - tp0[:-1]) / n_pos * precision).sum()
    return float(auroc), float(ap)


def _world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist
    return None


def evaluate_metrics(dataloader, model, times, loss_fn, num_classes, device, kwargs):
    """One pass over `dataloader` (batches = (*coeffs, true_y, lengths)): loss, accuracy, confusion matrix and, for two
    classes, AUROC / average precision.  num_classes=None: regression (loss only).  One host synchronisation."""
    dist = _world()
    with torch.no_grad():
        loss_sum = torch.zeros((), device=device, dtype=torch.float64)
        correct = torch.zeros((), device=device, dtype=torch.float64)
        size = 0
        confusion = None if num_classes is None else torch.zeros(num_classes * num_classes, device=device, dtype=torch.int64)
        scores, labels = [], []
        for batch in dataloader:
            batch = tuple(b.to(device, non_blocking=True) for b in batch)
            *coeffs, true_y, lengths = batch
            pred_y = model(times, coeffs, lengths, **kwargs)
            bs = true_y.size(0)
            size += bs
            loss_sum += loss_fn(pred_y, true_y).double() * bs
            if num_classes is None:
                continue
            guess = (pred_y > 0).to(torch.int64) if num_classes == 2 else torch.argmax(pred_y, dim=1)
            truth = true_y.to(torch.int64)
            correct += (guess == truth).sum()
            confusion += torch.bincount(truth.reshape(-1) * num_classes + guess.reshape(-1), minlength=num_classes * num_classes)
            if num_classes == 2:
                scores.append(pred_y.detach().reshape(-1))
                labels.append(true_y.detach().reshape(-1))
        sums = torch.stack([loss_sum, correct, torch.tensor(float(size), device=device, dtype=torch.float64)])
        if dist is not None:
            dist.all_reduce(sums)
            if confusion is not None:
                dist.all_reduce(confusion)
        metrics = AttrDict(dataset_size=int(sums[2].item()))
        total = max(metrics.dataset_size, 1)
        metrics.loss = float(sums[0]) / total            # assumes 'mean' reduction in the loss function
        if num_classes is not None:
            metrics.accuracy = float(sums[1]) / total
            metrics.confusion = confusion.reshape(num_classes, num_classes).cpu().numpy().astype(np.float64)
        if num_classes == 2 and scores:
            sc, lb = torch.cat(scores), torch.cat(labels)
            if dist is not None:       # ragged all-gather through padding to the largest shard
                n = torch.tensor([sc.numel()], device=device)
                ns = [torch.zeros_like(n) for _ in range(dist.get_world_size())]
                dist.all_gather(ns, n)
                m = int(max(int(x) for x in ns))
                pad = lambda t: torch.cat([t, t.new_zeros(m - t.numel())])
                gs = [sc.new_zeros(m) for _ in ns]
                gl = [lb.new_zeros(m) for _ in ns]
                dist.all_gather(gs, pad(sc))
                dist.all_gather(gl, pad(lb))
                sc = torch.cat([g[:int(k)] for g, k in zip(gs, ns)])
                lb = torch.cat([g[:int(k)] for g, k in zip(gl, ns)])
            metrics.auroc, metrics.average_precision = binary_ranking_metrics(sc, lb)
        return metrics


class GraphedStep:
    """The training step of the loop below - forward, loss, backward, optimizer step - recorded into one CUDA/HIP graph per
    batch shape and replayed with each batch copied into the recording's input buffers (the step is launch-bound: ~60
    kernels; replaying takes about two thirds of the eager time).  The solver draws fresh Brownian increments
    on every replay from a device-resident Philox key (torchsde.prepare_graph_capture).  The first `warmup` batches of a
    shape run eagerly (on a side stream, as capture requires); a learning-rate change drops the recordings (the rate is a
    constant of the recorded optimizer kernel).  Needs an optimizer created with capturable=True."""

    def __init__(self, model, times, optimizer, loss_fn, kwargs, device, warmup=3, log=None):
        from . import torchsde as _T
        _T.prepare_graph_capture(device)
        self.model, self.times, self.opt, self.loss_fn, self.kwargs = model, times, optimizer, loss_fn, kwargs
        self.device, self.warmup = device, warmup
        self.entries, self.lrs = {}, None
        self.side = torch.cuda.Stream(device)
        self.replays = 0
        self.log = log or (lambda msg: None)
        self.disabled = None          # reason the recordings were given up (a model that is not capture-safe): eager steps from then on

    def _step(self, coeffs, y, lengths):
        pred = self.model(self.times, coeffs, lengths, **self.kwargs)
        loss = self.loss_fn(pred, y)
        loss.backward()
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)

    def _eager(self, coeffs, y, lengths):
        """One eager step with the reference loop's handling of a failing batch (common_sde.py:157-166: an AssertionError of a
        batch is reported and the loop goes on)."""
        try:
            self._step(list(coeffs), y, lengths)
        except AssertionError as exc:
            self.log('Caught AssertionError: ' + str(exc))
            self.opt.zero_grad(set_to_none=True)

    def __call__(self, coeffs, y, lengths):
        if self.disabled is not None:
            return self._eager(coeffs, y, lengths)
        lrs = tuple(float(g['lr']) for g in self.opt.param_groups)
        if lrs != self.lrs:
            self.entries, self.lrs = {}, lrs
        batch = tuple(coeffs) + (y, lengths)
        key = tuple((tuple(b.shape), b.dtype) for b in batch)
        entry = self.entries.setdefault(key, {'seen': 0, 'graph': None, 'static': None})
        if entry['graph'] is None:
            if entry['seen'] < self.warmup:          # eager steps of this shape, on the side stream
                entry['seen'] += 1
                self.side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(self.side):
                    self._eager(coeffs, y, lengths)
                torch.cuda.current_stream(self.device).wait_stream(self.side)
                return
            static = tuple(b.clone() for b in batch)
            torch.cuda.synchronize(self.device)
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    self._step(list(static[:-2]), static[-2], static[-1])
            except Exception as exc:                  # a host sync, .item(), an allocation the capture cannot take, a failing batch ...
                self.disabled = f'{type(exc).__name__}: {exc}'
                self.log('GraphedStep: the training step cannot be recorded (' + self.disabled.splitlines()[0] +
                         '); eager steps from here on')
                self.entries = {}
                torch.cuda.synchronize(self.device)
                self.opt.zero_grad(set_to_none=True)  # (nothing of the failed recording ran; drop gradients it may have allocated)
                return self._eager(coeffs, y, lengths)
            entry['graph'], entry['static'] = graph, static
            graph.replay()                            # (the capture itself does not execute: run this batch's step now)
            self.replays += 1
            return
        for dst, src in zip(entry['static'], batch):
            dst.copy_(src, non_blocking=True)
        entry['graph'].replay()
        self.replays += 1


def train_loop(train_dataloader, val_dataloader, model, times, optimizer, loss_fn, max_epochs, num_classes, device, kwargs,
               step_mode, log=None, plateau_terminate=None, graph_steps=False):
    """The reference's epoch loop (common_sde.py:107-216).  Returns the history; `model` ends with the best parameters
    (validation accuracy for classification, validation loss for regression).  graph_steps=True replays the training step
    from a CUDA/HIP graph (GraphedStep; one process, CUDA, optimizer with capturable=True)."""
    modes = {'trainloss': 'min', 'valloss': 'min', 'valaccuracy': 'max', 'valauc': 'max', 'none': None}
    if step_mode not in modes:
        raise ValueError(f'step_mode must be one of {sorted(modes)}')
    scheduler = None
    if modes[step_mode] is not None:
        scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, patience=5, mode=modes[step_mode])
    if plateau_terminate is None:
        plateau_terminate = 100 if step_mode == 'none' else 50
    log = log or (lambda msg: None)
    model.train()
    best_state = copy.deepcopy(model.state_dict())
    best_train_loss, best_train_loss_epoch = math.inf, 0
    best_train_accuracy, best_train_accuracy_epoch = 0.0, 0
    best_val = -math.inf
    history = []
    graphed = GraphedStep(model, times, optimizer, loss_fn, kwargs, torch.device(device), log=log) if graph_steps else None
    for epoch in range(max_epochs):
        sampler = getattr(train_dataloader, 'sampler', None)
        if hasattr(sampler, 'set_epoch'):
            sampler.set_epoch(epoch)
        for batch in train_dataloader:
            batch = tuple(b.to(device, non_blocking=True) for b in batch)
            *train_coeffs, train_y, lengths = batch
            if graphed is not None:
                graphed(train_coeffs, train_y, lengths)
                continue
            try:
                pred_y = model(times, train_coeffs, lengths, **kwargs)
                loss = loss_fn(pred_y, train_y)
                loss.backward()
                optimizer.step()
                optimizer.zero_grad(set_to_none=True)
            except AssertionError as exc:          # the reference swallows assertion failures of a batch and goes on
                log('Caught AssertionError: ' + str(exc))
                optimizer.zero_grad(set_to_none=True)
        model.eval()
        train_metrics = evaluate_metrics(train_dataloader, model, times, loss_fn, num_classes, device, kwargs)
        val_metrics = evaluate_metrics(val_dataloader, model, times, loss_fn, num_classes, device, kwargs)
        model.train()
        if train_metrics.loss * 1.0001 < best_train_loss:
            best_train_loss, best_train_loss_epoch = train_metrics.loss, epoch
        if num_classes is not None and train_metrics.accuracy > best_train_accuracy * 1.001:
            best_train_accuracy, best_train_accuracy_epoch = train_metrics.accuracy, epoch
        score = val_metrics.accuracy if num_classes is not None else -val_metrics.loss
        if score > best_val:
            best_val = score
            best_state = copy.deepcopy(model.state_dict())
        line = f'Epoch: {epoch}  Train loss: {train_metrics.loss:.3}'
        if num_classes is not None:
            line += f'  Train accuracy: {train_metrics.accuracy:.3}'
        if 'auroc' in train_metrics:
            line += f'  Train auroc: {train_metrics.auroc:.3}'
        line += f'  Val loss: {val_metrics.loss:.3}'
        if num_classes is not None:
            line += f'  Val accuracy: {val_metrics.accuracy:.3}'
        if 'auroc' in val_metrics:
            line += f'  Val auroc: {val_metrics.auroc:.3}'
        log(line)
        if scheduler is not None:
            scheduler.step({'trainloss': train_metrics.loss, 'valloss': val_metrics.loss,
                            'valaccuracy': val_metrics.get('accuracy', 0.0), 'valauc': val_metrics.get('auroc', 0.0)}[step_mode])
        history.append(AttrDict(epoch=epoch, train_metrics=train_metrics, val_metrics=val_metrics,
                                lr=optimizer.param_groups[0]['lr']))
        if epoch > best_train_loss_epoch + plateau_terminate:
            log(f'Breaking because of no improvement in training loss for {plateau_terminate} epochs.')
            break
        if num_classes is not None and epoch > best_train_accuracy_epoch + plateau_terminate:
            log(f'Breaking because of no improvement in training accuracy for {plateau_terminate} epochs.')
            break
    model.load_state_dict(best_state)       # in-place copies: the solver's parameter arena stays where it is
    return history


class _Encoder(json.JSONEncoder):
    def default(self, o):
        if isinstance(o, (torch.Tensor, np.ndarray)):
            return o.tolist()
        return super().default(o)


def save_results(directory, name, result):
    """results-sde/<name>/<run number> (JSON), as the reference's _save_results."""
    loc = os.path.join(directory, name)
    os.makedirs(loc, exist_ok=True)
    num = max([int(f) for f in os.listdir(loc) if f.isdigit()] + [-1]) + 1
    out = {k: v for k, v in result.items() if not k.endswith('_dataloader')}
    out['model'] = str(out['model'])
    path = os.path.join(loc, str(num))
    with open(path, 'w') as f:
        json.dump(out, f, cls=_Encoder)
    return path


def make_model(name, input_channels, output_channels, hidden_channels, hidden_hidden_channels, num_hidden_layers,
               use_intensity=False, initial=True):
    """common_sde.make_model for the SDE entries (common_sde.py:301-342): returns a factory () -> (model, vector field);
    `use_intensity` belongs to the reference's non-SDE baselines (GRU-ODE etc.) and is ignored."""
    def factory():
        return modules.make_sde_model(name, input_channels, output_channels, hidden_channels, hidden_hidden_channels,
                                      num_hidden_layers, initial=initial)
    return factory


def main(name, model_name, times, train_dataloader, val_dataloader, test_dataloader, device, make_model, num_classes,
         max_epochs, lr, kwargs, step_mode, pos_weight=torch.tensor(1), results_dir=None, log=print, regularise='l2',
         graph_steps=None):
    """common_sde.main (common_sde.py:248-298): build, train, evaluate; `num_classes=None` trains a regression model with
    the mean-squared error (the forecasting benchmark).  Results are written only when `name` and `results_dir` are set.
    graph_steps: replay the training step from a CUDA/HIP graph (GraphedStep) - the step of these models is bound by the host
    (whole-model neurallnsde step: 1.43 ms eager, 0.88 ms replayed, 0.61 ms for its solve).  None (default) = wherever it is eligible:
    one process, CUDA device; False = eager steps; True = required where eligible (still eager under DDP / on the CPU)."""
    device = torch.device(device)
    times = times.to(device)
    on_gpu = device.type == 'cuda'
    baseline_memory = None
    if on_gpu:
        torch.cuda.reset_peak_memory_stats(device)
        baseline_memory = torch.cuda.memory_allocated(device)
    model, regularise_parameters = make_model()
    if num_classes == 2:
        model = SqueezeEnd(model)
        base_loss = torch.nn.BCEWithLogitsLoss(pos_weight=pos_weight.to(device))
    elif num_classes is None:
        base_loss = torch.nn.functional.mse_loss
    else:
        base_loss = torch.nn.functional.cross_entropy
    loss_fn = add_weight_regularisation(base_loss, regularise_parameters, mode=regularise)
    model.to(device)
    dist = _world()
    net = model
    if dist is not None:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[device.index] if on_gpu else None)
    # same update rule as the reference's Adam (common_sde.py:287); on the GPU the single-launch implementation instead of
    # the default multi-tensor one (~15 launches per step for these models)
    graph_steps = (graph_steps is None or bool(graph_steps)) and on_gpu and dist is None
    optimizer = torch.optim.Adam(net.parameters(), lr=lr, weight_decay=lr * 0.01, fused=True if on_gpu else None,
                                 capturable=graph_steps)
    history = train_loop(train_dataloader, val_dataloader, net, times, optimizer, loss_fn, max_epochs, num_classes, device,
                         kwargs, step_mode, log=log if (dist is None or dist.get_rank() == 0) else None, graph_steps=graph_steps)
    net.eval()
    train_metrics = evaluate_metrics(train_dataloader, net, times, loss_fn, num_classes, device, kwargs)
    val_metrics = evaluate_metrics(val_dataloader, net, times, loss_fn, num_classes, device, kwargs)
    test_metrics = evaluate_metrics(test_dataloader, net, times, loss_fn, num_classes, device, kwargs)
    memory_usage = torch.cuda.max_memory_allocated(device) - baseline_memory if on_gpu else None
    result = AttrDict(name=name, model_name=model_name, times=times, memory_usage=memory_usage,
                      baseline_memory=baseline_memory, num_classes=num_classes, train_dataloader=train_dataloader,
                      val_dataloader=val_dataloader, test_dataloader=test_dataloader, model=model,
                      parameters=count_parameters(model), history=history, train_metrics=train_metrics,
                      val_metrics=val_metrics, test_metrics=test_metrics)
    if name is not None and results_dir is not None and (dist is None or dist.get_rank() == 0):
        result.saved_to = save_results(results_dir, name, result)
    return result


# the reference's private names
_add_weight_regularisation = add_weight_regularisation
_evaluate_metrics = evaluate_metrics
_train_loop = train_loop
_SqueezeEnd = SqueezeEnd
_AttrDict = AttrDict
_count_parameters = count_parameters
