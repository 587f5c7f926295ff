"""Test-side module with the attribute / parameter names of torch-ists' LatentSDE (reference
torch-ists/torch_ists/diff_module/NSDE/latent_sde.py:31-89) so that the fixture's state_dict loads: posterior drift = relu MLP
of [sin t, cos t, y] on the latent channels, constant shared diffusion sigma, OU prior drift theta (mu - y); the augmented
system appends the running KL term 0.5 |(f - h) / g|^2 as one more state channel.  Written for the tests, not the product."""
import math

import torch
from torch import distributions, nn

import stable_neural_sdes_amd as S


def _safe_ratio(num, den, eps=1e-7):
    den = torch.where(den.abs().detach() > eps, den, torch.full_like(den, eps) * den.sign())
    return num / den


class LatentField(nn.Module):
    sde_type, noise_type = 'ito', 'diagonal'

    def __init__(self, input_channels, hidden_channels, hidden_hidden_channels, num_hidden_layers, theta=1.0, mu=0.0, sigma=0.5):
        super().__init__()
        for name, v in (('theta', theta), ('mu', mu), ('sigma', sigma)):
            self.register_buffer(name, torch.tensor([[float(v)]]))
        logvar = math.log(sigma ** 2 / (2.0 * theta))      # stationary variance of the OU prior
        self.register_buffer('py0_mean', torch.tensor([[float(mu)]]))
        self.register_buffer('py0_logvar', torch.tensor([[logvar]]))
        self.qy0_mean = nn.Parameter(torch.tensor([[float(mu)]]))
        self.qy0_logvar = nn.Parameter(torch.tensor([[logvar]]))
        latent = hidden_channels - 1
        self.initial_network = nn.Sequential(nn.Linear(input_channels, latent))
        self.linear_in = nn.Linear(latent + 2, hidden_hidden_channels)
        self.linears = nn.ModuleList(nn.Linear(hidden_hidden_channels, hidden_hidden_channels) for _ in range(num_hidden_layers - 1))
        self.linear_out = nn.Linear(hidden_hidden_channels, latent)
        self.embedding = nn.Linear(latent, hidden_channels)

    def f(self, t, y):
        if t.dim() == 0:
            t = t.expand(y.shape[0], 1)
        z = self.linear_in(torch.cat([t.sin(), t.cos(), y], dim=-1)).relu()
        for lin in self.linears:
            z = lin(z).relu()
        return self.linear_out(z)

    def g(self, t, y):
        return self.sigma.expand(y.shape[0], y.shape[1])

    def h(self, t, y):
        return self.theta * (self.mu - y)

    def f_aug(self, t, y):
        y = y[:, :-1]
        f = self.f(t, y)
        u = _safe_ratio(f - self.h(t, y), self.g(t, y))
        return torch.cat([f, 0.5 * (u * u).sum(dim=1, keepdim=True)], dim=1)

    def g_aug(self, t, y):
        y = y[:, :-1]
        return torch.cat([self.g(t, y), torch.zeros(y.shape[0], 1, dtype=y.dtype, device=y.device)], dim=1)

    def forward(self, coeffs, times, **kwargs):
        """The reference's forward (latent_sde.py:91-150) over this package's mirrors: (readout, latent path, KL)."""
        X = S.torchcde.CubicSpline(coeffs, times)
        aug_y0 = self.initial_network(X.evaluate(times[0]))
        aug_y0 = torch.cat([aug_y0, torch.zeros(coeffs.shape[0], 1).to(aug_y0)], dim=1)
        dt = max(float((times[1:] - times[:-1]).min()), 1e-3)
        kwargs.setdefault('method', 'srk')
        aug = S.torchsde.sdeint_adjoint(sde=self, y0=aug_y0, ts=times, dt=dt, names={'drift': 'f_aug', 'diffusion': 'g_aug'}, **kwargs)
        aug = aug.permute(1, 0, 2)
        latent = aug[:, :, :-1]
        q0 = distributions.Normal(self.qy0_mean, torch.exp(0.5 * self.qy0_logvar))
        p0 = distributions.Normal(self.py0_mean, torch.exp(0.5 * self.py0_logvar))
        kl0 = distributions.kl_divergence(q0, p0).sum(dim=1)
        return self.embedding(latent), latent, (kl0 + aug[:, -1, -1]).mean(dim=0)
