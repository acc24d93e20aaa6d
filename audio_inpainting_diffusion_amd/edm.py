"""EDM diffusion parameterisation (host side) -- mirror of the reference's ``diff_params.edm.EDM``.

Same constructor ``EDM(args)`` and the methods the sampling path uses: ``create_schedule`` (reference
diff_params/edm.py:55-64), ``get_gamma`` (:38-53), ``sample_prior`` (:87-95), ``cskip/cout/cin/cnoise``
(:97-128) and ``denoiser`` (:133-148).  Everything here is O(T) or O(B) scalar work that stays on the host in
float32 torch arithmetic, bit-identical to the reference (pinned by tests/golden/edm_schedule.npz); the
per-sample tensor work is done by the HIP kernels (the network's fused ``denoise`` entry folds c_in into the
CQT analysis and c_skip / c_out into the synthesis spectrum).  The training members ``sample_ptrain_safe`` (:76-85),
``prepare_train_preconditioning`` (:150-163) and ``loss_fn`` (:166-193) are mirrored too: with the MI355X network in
``train()`` mode its forward is differentiable w.r.t. the parameters (autograd.TrainFn), so the reference's own
``loss.backward(); optimizer.step()`` works unchanged; ``training.Trainer`` is the all-HIP path (no torch.optim).
"""
from __future__ import annotations

import torch


class EDM:
    def __init__(self, args):
        self.args = args
        dp = args.diff_params
        self.sigma_min, self.sigma_max = dp.sigma_min, dp.sigma_max
        self.P_mean, self.P_std = dp.P_mean, dp.P_std
        self.ro, self.ro_train = dp.ro, dp.ro_train
        self.sigma_data = dp.sigma_data
        self.Schurn, self.Stmin, self.Stmax, self.Snoise = dp.Schurn, dp.Stmin, dp.Stmax, dp.Snoise
        if dp.aweighting.use_aweighting:
            raise NotImplementedError("A-weighted training loss is outside the sampling hot path")

    def get_gamma(self, t):
        N = t.shape[0]
        gamma = torch.zeros(t.shape).to(t.device)
        indexes = torch.logical_and(t > self.Stmin, t < self.Stmax)
        gamma[indexes] = gamma[indexes] + torch.min(torch.Tensor([self.Schurn / N, 2 ** (1 / 2) - 1]))
        return gamma

    def create_schedule(self, nb_steps):
        i = torch.arange(0, nb_steps + 1)
        t = (self.sigma_max ** (1 / self.ro) + i / (nb_steps - 1) *
             (self.sigma_min ** (1 / self.ro) - self.sigma_max ** (1 / self.ro))) ** self.ro
        t[-1] = 0
        return t

    def sample_prior(self, shape, sigma):
        return torch.randn(shape).to(sigma.device) * sigma

    def cskip(self, sigma):
        return self.sigma_data ** 2 * (sigma ** 2 + self.sigma_data ** 2) ** -1

    def cout(self, sigma):
        return sigma * self.sigma_data * (self.sigma_data ** 2 + sigma ** 2) ** (-0.5)

    def cin(self, sigma):
        return (self.sigma_data ** 2 + sigma ** 2) ** (-0.5)

    def cnoise(self, sigma):
        return (1 / 4) * torch.log(sigma)

    def denoiser(self, xn, net, sigma):
        """Generic preconditioned denoiser (works with any ``net``); our Sampler uses ``net.denoise`` instead
        when the network is the MI355X one."""
        if len(sigma.shape) == 1:
            sigma = sigma.unsqueeze(-1)
        return self.cskip(sigma) * xn + self.cout(sigma) * net(self.cin(sigma) * xn, self.cnoise(sigma))

    # ---- training (edm.py:76-85, :150-193) ----------------------------------------------------------------------------------
    def sample_ptrain_safe(self, N):
        a = torch.rand(N)
        return (self.sigma_max ** (1 / self.ro_train) + a * (self.sigma_min ** (1 / self.ro_train) - self.sigma_max ** (1 / self.ro_train))) ** self.ro_train

    def prepare_train_preconditioning(self, x, sigma):
        noise = self.sample_prior(x.shape, sigma)
        cskip, cout, cin, cnoise = self.cskip(sigma), self.cout(sigma), self.cin(sigma), self.cnoise(sigma)
        target = (1 / cout) * (x - cskip * (x + noise))
        return cin * (x + noise), target, cnoise

    def loss_fn(self, net, x):
        """(error**2, sigma) exactly as the reference returns them (the caller takes ``.mean()`` and calls ``backward()``)."""
        sigma = self.sample_ptrain_safe(x.shape[0]).unsqueeze(-1).to(x.device)
        inp, target, cnoise = self.prepare_train_preconditioning(x, sigma)
        error = net(inp, cnoise) - target
        try:                                   # as the reference (:180-187): it reads args.net.use_cqt_DC_correction inside a bare
            if self.args.net.use_cqt_DC_correction:    # try -- no shipped config has an `args.net`, so the correction never runs there either
                error = net.CQTransform.apply_hpf_DC(error)
        except Exception:
            pass
        return error ** 2, sigma
