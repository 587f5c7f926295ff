"""Vector fields shaped like the reference tutorial's (tutorial/*.ipynb, cell 7): an MLP written as
nn.Sequential(Linear, act, ..., Linear), a control embedding, a time-only diffusion net.  Written here for the tests
(same attribute names and call structure as the notebooks, so a notebook user's module takes the same path)."""
import torch
from torch import nn

import stable_neural_sdes_amd as S


class LipSwish(nn.Module):
    def forward(self, x):
        return 0.909 * torch.nn.functional.silu(x)


ACTS = {'lipswish': LipSwish, 'relu': nn.ReLU, 'silu': nn.SiLU}


class MLP(nn.Module):
    def __init__(self, n_in, n_out, width, depth, activation='lipswish'):
        super().__init__()
        act = ACTS[activation]()
        mods = [nn.Linear(n_in, width), act]
        for _ in range(depth - 1):
            mods += [nn.Linear(width, width), act]
        mods.append(nn.Linear(width, n_out))
        self._model = nn.Sequential(*mods)

    def forward(self, x):
        return self._model(x)


class TutorialField(nn.Module):
    """kind: 'lsde' (f on [y | X], additive g(t)), 'lnsde' (f on [t, y | X], g(t) * y with the saturating time
    feature), 'lnsde_additive', 'gsde' (f * y, g(t) * y), 'nsde' (the Neural SDE notebook's NeuralSDEFunc: f and g both MLPs
    of [t, y], no control path in the field), 'ode' (the Neural ODE notebook's NeuralODEFunc: f = MLP of [t, y], noise_type
    'scalar' with g = zeros (B, H, 1))."""

    def __init__(self, kind, input_dim, hidden_dim, num_layers, activation='lipswish'):
        super().__init__()
        self.kind = kind
        self.sde_type, self.noise_type = 'ito', ('scalar' if kind == 'ode' else 'diagonal')
        if kind != 'lsde':
            self.linear_in = nn.Linear(hidden_dim + 1, hidden_dim)
        if kind == 'ode':            # the Neural ODE notebook's NeuralODEFunc: drift only, scalar-noise shape with g = 0
            self.f_net = MLP(hidden_dim, hidden_dim, hidden_dim, num_layers, activation)
            return
        if kind != 'nsde':
            self.linear_X = nn.Linear(input_dim, hidden_dim)
            self.emb = nn.Linear(2 * hidden_dim, hidden_dim)
        self.f_net = MLP(hidden_dim, hidden_dim, hidden_dim, num_layers, activation)
        self.linear_out = nn.Linear(hidden_dim, hidden_dim)
        self.noise_in = nn.Linear(hidden_dim + 1 if kind == 'nsde' else 1, hidden_dim)
        self.g_net = MLP(hidden_dim, hidden_dim, hidden_dim, num_layers, activation)
        if kind.startswith('lnsde'):
            self.time_rate = nn.Parameter(torch.tensor(1.0))

    def set_X(self, coeffs, times):
        self.coeffs, self.times = coeffs, times
        self.X = S.torchcde.CubicSpline(coeffs, times)

    def _t(self, t, y):
        if t.dim() == 0:
            t = torch.full_like(y[:, 0], fill_value=float(t)).unsqueeze(-1)
        return t

    def f(self, t, y):
        if self.kind in ('nsde', 'ode'):
            return self.f_net(self.linear_in(torch.cat((self._t(t, y), y), dim=-1)))
        Xt = self.linear_X(self.X.evaluate(t))
        if self.kind == 'lsde':
            yy = y
        else:
            yy = self.linear_in(torch.cat((self._t(t, y), y), dim=-1))
        z = self.linear_out(self.f_net(self.emb(torch.cat([yy, Xt], dim=-1))))
        return z * y if self.kind == 'gsde' else z

    def g(self, t, y):
        if self.kind == 'ode':
            return torch.zeros(y.size(0), y.size(1), 1, dtype=y.dtype, device=y.device)
        t = self._t(t, y)
        if self.kind == 'nsde':
            return self.g_net(self.noise_in(torch.cat((t, y), dim=-1)))
        if self.kind.startswith('lnsde'):
            t = 1.0 - torch.exp(-torch.nn.functional.softplus(self.time_rate) * t)
        s = self.g_net(self.noise_in(t))
        return s if self.kind in ('lsde', 'lnsde_additive') else s * y
