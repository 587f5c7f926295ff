#!/usr/bin/env python3
"""The reference's tutorial workflow (tutorial/simple OU process - Neural LSDE / LNSDE / GSDE .ipynb) on the engine:
simulate Ornstein-Uhlenbeck paths, fit a Neural LSDE / LNSDE / GSDE whose vector field is written the tutorial's way
(nn.Sequential MLPs with LipSwish, control embedding, time-only diffusion net), report the test MSE.

After `stable_neural_sdes_amd.install()` the script's own `import torchsde, torchcde` resolve to the engine; on a GPU every
`torchsde.sdeint` of training and evaluation then runs on the fused HIP kernels (fields.py: the module is recognised by its
structure, its weights composed onto the C ABI's parameter block) - `--backend torch` forces the tensor-op loop for
comparison.

usage: python examples/tutorial_ou_process.py [--field lsde|lnsde|gsde] [--epochs 20] [--batch 16] [--backend auto|torch]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stable_neural_sdes_amd  # noqa: E402

stable_neural_sdes_amd.install()
import torchcde  # noqa: E402  (the engine's mirrors from here on)
import torchsde  # noqa: E402


def ou_paths(num, T, N, theta, mu, sigma, x0, rng):
    dt = T / N
    x = np.full((num, N), x0, dtype=np.float64)
    for i in range(1, N):
        x[:, i] = x[:, i - 1] + theta * (mu - x[:, i - 1]) * dt + sigma * rng.normal(0.0, np.sqrt(dt), num)
    t = np.broadcast_to(np.linspace(0, T, N), (num, N))
    return torch.tensor(np.stack([t, x], axis=-1), dtype=torch.float32)        # (num, N, 2): [time, value]


class LipSwish(nn.Module):
    def forward(self, x):
        return 0.909 * torch.nn.functional.silu(x)


def mlp(n_in, n_out, width, depth):
    mods = [nn.Linear(n_in, width), LipSwish()]
    for _ in range(depth - 1):
        mods += [nn.Linear(width, width), LipSwish()]
    return nn.Sequential(*mods, nn.Linear(width, n_out))


class Field(nn.Module):
    """lsde: f on [y | X], additive g(t);  lnsde: f on [t, y | X], g(t) y;  gsde: f y, g(t) y."""

    def __init__(self, kind, input_dim, hidden_dim, num_layers):
        super().__init__()
        self.kind, self.sde_type, self.noise_type = kind, 'ito', 'diagonal'
        if kind != 'lsde':
            self.linear_in = nn.Linear(hidden_dim + 1, hidden_dim)
        self.linear_X = nn.Linear(input_dim, hidden_dim)
        self.emb = nn.Linear(2 * hidden_dim, hidden_dim)
        self.f_net = mlp(hidden_dim, hidden_dim, hidden_dim, num_layers)
        self.linear_out = nn.Linear(hidden_dim, hidden_dim)
        self.noise_in = nn.Linear(1, hidden_dim)
        self.g_net = mlp(hidden_dim, hidden_dim, hidden_dim, num_layers)

    def set_X(self, coeffs, times):
        self.coeffs, self.times = coeffs, times
        self.X = torchcde.CubicSpline(coeffs, times)

    @staticmethod
    def _col(t, y):
        return torch.full_like(y[:, :1], float(t)) if t.dim() == 0 else t

    def f(self, t, y):
        Xt = self.linear_X(self.X.evaluate(t))
        yy = y if self.kind == 'lsde' else self.linear_in(torch.cat((self._col(t, y), y), dim=-1))
        z = self.linear_out(self.f_net(self.emb(torch.cat([yy, Xt], dim=-1))))
        return z * y if self.kind == 'gsde' else z

    def g(self, t, y):
        s = self.g_net(self.noise_in(self._col(t, y)))
        return s if self.kind == 'lsde' else s * y


class Model(nn.Module):
    def __init__(self, kind, input_dim, hidden_dim, output_dim, num_layers, options):
        super().__init__()
        self.func = Field(kind, input_dim, hidden_dim, num_layers)
        self.initial = nn.Linear(input_dim, hidden_dim)
        self.decoder = nn.Linear(hidden_dim, output_dim)
        self.kind, self.options = kind, options

    def forward(self, coeffs, times):
        self.func.set_X(coeffs, times)
        y0 = self.initial(self.func.X.evaluate(times[0]))
        if self.kind == 'gsde':
            y0 = torch.nn.functional.softplus(y0) + 1e-4
        z = torchsde.sdeint(self.func, y0=y0, ts=times, dt=0.05, method='euler', options=self.options)
        return self.decoder(z.permute(1, 0, 2))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--field', default='lsde', choices=['lsde', 'lnsde', 'gsde'])
    ap.add_argument('--epochs', type=int, default=20)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--samples', type=int, default=1000)
    ap.add_argument('--hidden', type=int, default=32)
    ap.add_argument('--layers', type=int, default=1)
    ap.add_argument('--backend', default='auto', choices=['auto', 'torch'])
    ap.add_argument('--device', default='cuda' if torch.cuda.is_available() else 'cpu')
    ap.add_argument('--seed', type=int, default=42)
    args = ap.parse_args(argv)
    dev = torch.device(args.device)
    torch.manual_seed(args.seed)
    rng = np.random.default_rng(args.seed)
    data = ou_paths(args.samples, 10.0, 20, 0.2, 0.0, 0.1, 1.0, rng)
    times = torch.linspace(0, 1, data.shape[1])
    coeffs = torchcde.hermite_cubic_coefficients_with_backward_differences(data, times)
    n_train = int(0.8 * args.samples)
    perm = torch.from_numpy(rng.permutation(args.samples))
    tr, te = perm[:n_train], perm[n_train:]
    data, coeffs, times = data.to(dev), coeffs.to(dev), times.to(dev)
    model = Model(args.field, 2, args.hidden, 1, args.layers, {'backend': args.backend}).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    mse = nn.MSELoss()

    def evaluate():
        model.eval()
        with torch.no_grad():
            losses = [mse(model(coeffs[idx], times).squeeze(-1), data[idx, :, 1]).item()
                      for idx in te.split(args.batch)]
        return float(np.mean(losses))

    history = [evaluate()]
    print(f'field={args.field} hidden={args.hidden} device={dev} backend={args.backend}: test MSE before training {history[0]:.5f}')
    for epoch in range(1, args.epochs + 1):
        model.train()
        t0, total = time.perf_counter(), 0.0
        order = tr[torch.randperm(n_train)]
        for idx in order.split(args.batch):
            opt.zero_grad()
            loss = mse(model(coeffs[idx], times).squeeze(-1), data[idx, :, 1])
            loss.backward()
            opt.step()
            total += loss.item()
        if dev.type == 'cuda':
            torch.cuda.synchronize()
        el = time.perf_counter() - t0
        history.append(evaluate())
        print(f'epoch {epoch:3d}  train MSE {total / len(order.split(args.batch)):.5f}  test MSE {history[-1]:.5f}  '
              f'{len(order.split(args.batch)) / el:7.1f} training steps/s')
    return history


if __name__ == '__main__':
    main()
