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


class _FirFn(torch.autograd.Function):
    """training.FirFilter (aid_resample_poly at ratio 1:1) with its exact adjoint as the backward."""

    @staticmethod
    def forward(ctx, x, fir):
        ctx.fir = fir
        return fir.apply(x.detach())

    @staticmethod
    def backward(ctx, g):
        return ctx.fir.adjoint(g.contiguous()), None


class AWeighting:
    """The reference's ``FIRFilter(filter_type="aw")`` (utils/training_utils.py:55-137) as the EDM loss uses it: A-weighting taps
    (training.a_weighting_taps, golden-pinned against the reference's) applied along time on the HIP FIR kernel, differentiable.
    GPU rows only -- like every operator of this package it has no CPU path."""

    def __init__(self, fs, ntaps):
        from .training import a_weighting_taps
        self.taps = a_weighting_taps(fs, ntaps)
        self._fir = {}

    def __call__(self, err):
        from . import _lib
        from .training import FirFilter
        if not err.is_cuda:
            raise _lib.AidError("AWeighting runs on the GPU only (no CPU fallback)")
        fir = self._fir.get(err.device)
        if fir is None:
            fir = self._fir[err.device] = FirFilter(self.taps, err.device)
        shape = err.shape
        return _FirFn.apply(err.reshape(-1, shape[-1]).contiguous().float(), fir).reshape(shape)


class EDM:
    def __init__(self, args):
        self.args = args
        dp = args.diff_params
        self.sigma_min, self.sigma_max = dp.sigma_min, dp.sigma_max
        self.P_mean, self.P_std = dp.P_mean, dp.P_std
        self.ro, self.ro_train = dp.ro, dp.ro_train
        self.sigma_data = dp.sigma_data
        self.Schurn, self.Stmin, self.Stmax, self.Snoise = dp.Schurn, dp.Stmin, dp.Stmax, dp.Snoise
        self.AW = None
        if dp.aweighting.use_aweighting:              # perceptual error filter (edm.py:33-34, applied at :189-190)
            self.AW = AWeighting(args.exp.sample_rate, dp.aweighting.ntaps)

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

    def lambda_w(self, sigma):
        """Karras et al.'s loss weight 1 / c_out(sigma)^2 (edm.py:130-131; the reference defines it and never calls it)."""
        return (self.sigma_data ** 2 + sigma ** 2) / (sigma * self.sigma_data) ** 2

    # ---- training (edm.py:67-85, :150-193) ----------------------------------------------------------------------------------
    def sample_ptrain(self, N):
        """Log-normal noise levels exp(N(P_mean, P_std^2)) clipped to [sigma_min, sigma_max] (edm.py:67-75, numpy's global generator as there;
        'not used' by the reference's own training, which draws sample_ptrain_safe)."""
        import numpy as np
        return np.clip(np.exp(self.P_mean + self.P_std * np.random.randn(N)), self.sigma_min, self.sigma_max)

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
        if self.AW is not None:                # (:189-190)
            error = self.AW(error)
        return error ** 2, sigma
