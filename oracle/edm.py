"""CPU oracle: EDM parameterisation used by the inpainting sampler.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned by tests/golden/edm_schedule.npz and
tests/golden/sampler_*.npz (captured from the reference's own EDM / Sampler classes).

Follows reference diff_params/edm.py: get_gamma :38-53, create_schedule :55-64, sample_prior :87-95,
cskip/cout/cin/cnoise :97-128, denoiser :133-148.
"""
from __future__ import annotations

import torch


class OracleEDM:
    def __init__(self, sigma_data=0.063, sigma_min=1e-4, sigma_max=1.0, ro=13, Schurn=10, Stmin=0, Stmax=50, Snoise=1.0):
        self.sigma_data, self.sigma_min, self.sigma_max, self.ro = sigma_data, sigma_min, sigma_max, ro
        self.Schurn, self.Stmin, self.Stmax, self.Snoise = Schurn, Stmin, Stmax, Snoise

    def create_schedule(self, nb_steps: int) -> torch.Tensor:
        # edm.py:61-64 -- int64 arange / python int -> float32 division, python-float powers
        i = torch.arange(0, nb_steps + 1)
        t = (self.sigma_max ** (1 / self.ro) + i / (nb_steps - 1) *
             (self.sigma_min ** (1 / self.ro) - self.sigma_max ** (1 / self.ro))) ** self.ro
        t[-1] = 0
        return t

    def get_gamma(self, t: torch.Tensor) -> torch.Tensor:
        # edm.py:44-53
        N = t.shape[0]
        gamma = torch.zeros(t.shape)
        idx = torch.logical_and(t > self.Stmin, t < self.Stmax)
        gamma[idx] = gamma[idx] + torch.min(torch.Tensor([self.Schurn / N, 2 ** (1 / 2) - 1]))
        return gamma

    def cskip(self, s): return self.sigma_data ** 2 * (s ** 2 + self.sigma_data ** 2) ** -1
    def cout(self, s): return s * self.sigma_data * (self.sigma_data ** 2 + s ** 2) ** (-0.5)
    def cin(self, s): return (self.sigma_data ** 2 + s ** 2) ** (-0.5)
    def cnoise(self, s): return (1 / 4) * torch.log(s)

    def denoiser(self, xn, net, sigma):
        # edm.py:141-148
        if len(sigma.shape) == 1:
            sigma = sigma.unsqueeze(-1)
        return self.cskip(sigma) * xn + self.cout(sigma) * net(self.cin(sigma) * xn, self.cnoise(sigma))
