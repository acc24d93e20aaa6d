"""CPU oracle: the EDM inpainting sampling loop (Heun / stochastic churn / reconstruction guidance /
data-consistency projection).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned by tests/golden/sampler_*.npz (trajectories of
the reference's own ``Sampler`` driving a toy denoiser).

Follows reference testing/edm_sampler_inpainting.py: get_score_rec_guidance :57-113, get_score :115-153,
predict :178-262, apply_mask :264-269, apply_spectral_mask :271-290, prepare_smooth_mask :302-325,
predict_inpainting :327-346, predict_spectrogram_inpainting :348-364.

Batch semantics.  The reference is only ever run at B=1 (its guided branch raises for B>1 with norm=2,
:75-:78).  Here every quantity the reference reduces over the whole batch (the guidance norm :75, the
gradient norm ``normguide`` :83, the mask row used for smoothing :307) is reduced PER ITEM, so item b of a
batch equals the reference's B=1 run on that item.  RNG: when ``seeds`` is None the global torch CPU
generator is consumed exactly like the reference (one ``randn(shape)`` for the prior, edm.py:94, then one
``randn(shape)`` per churned step, :212); with ``seeds`` each item owns a CPU generator seeded
``seeds[b]`` and draws ``randn([1,L])`` in the same order, so results do not depend on how items are
batched or sharded.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch


def smooth_mask_rows(mask: torch.Tensor, size: int) -> torch.Tensor:
    """prepare_smooth_mask (:302-325) applied to every row of mask[R,L] (the reference uses row 0 for all)."""
    hann = torch.hann_window(size * 2)
    left, right = hann[0:size], hann[size:]
    out = mask.clone().to(torch.float32)
    for r in range(mask.shape[0]):
        m = mask[r].cpu().numpy()
        prev = np.concatenate(([1.0], m[:-1]))
        for i in np.nonzero(m != prev)[0]:
            if m[i] == 0:
                out[r, i - size:i] = right
            if m[i] == 1:
                out[r, i:i + size] = left
    return out


def spectral_mask_apply(x: torch.Tensor, mask: torch.Tensor, n_fft=1024, hop=256, win_length=1024) -> torch.Tensor:
    """apply_spectral_mask (:271-290): zero-pad to a multiple of n_fft, stft (Hann), times mask[F,T] (or [B,F,T]),
    istft, crop.  Differentiable (the reference takes the guidance gradient through it with autograd)."""
    window = torch.hann_window(win_length, dtype=x.dtype)
    L = x.shape[-1]
    xp = torch.nn.functional.pad(x, (0, n_fft - L % n_fft), mode="constant", value=0)
    X = torch.stft(xp, n_fft, hop, win_length, window, return_complex=True)
    X = X * (mask.unsqueeze(0) if mask.dim() == 2 else mask)
    y = torch.istft(X, n_fft, hop, win_length, window, return_complex=False)
    return y[..., 0:L]


class OracleSampler:
    def __init__(self, model, edm, T=35, order=2, xi=0.25, norm=2, data_consistency=True, smooth=True,
                 hann_size=50, filter_out_cqt_DC_Nyq=True, audio_len=None, dc_type="always", smoothl1_beta=1.0):
        self.model, self.edm = model, edm
        self.nb_steps, self.order, self.xi, self.norm = T, order, xi, norm
        self.smoothl1_beta = smoothl1_beta              # tester.posterior_sampling.smoothl1_beta (norm == "smoothl1", :72-73)
        # (:22-24) data_consistency.use and type == "always" | "end"
        self.data_consistency = bool(data_consistency) and dc_type == "always"
        self.data_consistency_end = bool(data_consistency) and dc_type == "end"
        self.smooth, self.hann_size = smooth, hann_size
        self.filter_hpf = filter_out_cqt_DC_Nyq
        self.audio_len = audio_len
        self.trace = None
        self.stop_after = None                          # tests: leave the loop after this many steps (the state is returned as it is)
        self.states = None                              # tests: set to [] to record the state after every step
        self.rid = None                                 # set by predict_*(rid=True): per-step debug buffers (:185-191, :217-226, :255)
        self.degradation = lambda x: self.mask * x                                          # apply_mask (:264-269)
        self.project = lambda x: self.smask * self.y + (1 - self.smask) * x                 # (:343)

    # -- one denoiser evaluation (:115-153) ------------------------------------------------------
    def get_score(self, x, t_i):
        B = x.shape[0]
        sig = t_i.reshape(1, 1).expand(B, 1)
        if self.y is None:                                  # unconditional sampling (:115-125): denoise, DC/Nyquist projector, no projection
            with torch.no_grad():
                x_hat = self.edm.denoiser(x, self.model, sig)
                if self.filter_hpf:
                    x_hat = self.model.CQTransform.apply_hpf_DC(x_hat)
            if self.trace is not None:
                self.trace.append(x_hat.detach().clone())
            return (x_hat - x) / t_i ** 2
        if self.xi > 0:
            x = x.detach().requires_grad_()
            x_hat = self.edm.denoiser(x, self.model, sig)
            if self.filter_hpf:
                x_hat = self.model.CQTransform.apply_hpf_DC(x_hat)
            if self.norm == "smoothl1":                                                     # (:72-73; 'sum' over the item)
                norm = torch.nn.functional.smooth_l1_loss(self.y, self.degradation(x_hat), reduction="none", beta=self.smoothl1_beta).reshape(B, -1).sum(dim=1)
            else:                                                                           # (:67-70: dim = 1 for [B, N] observations; dim = (1, 2) for 3-D ones,
                diff = self.y - self.degradation(x_hat)                                     #  where ord = 2 / 1 is the INDUCED matrix norm: spectral / max column sum)
                norm = torch.linalg.norm(diff, dim=(1, 2) if diff.dim() == 3 else 1, ord=self.norm)   # [B]  (:65,:75)
            g = torch.autograd.grad(norm.sum(), x)[0]                                       # per-item grads
            L = self.audio_len if self.audio_len is not None else x.shape[-1]
            normguide = torch.linalg.norm(g, dim=1, keepdim=True) / L ** 0.5                # (:83) per item
            s = t_i * self.xi / (normguide + 1e-6)                                          # (:87)
            x_hat_old = x_hat.detach().clone()
            x_hat = (x_hat - s * g).detach()                                                # (:97)
            x = x.detach()
            self._rid_last = (x_hat_old, (s * g).detach(), x_hat.clone())
        else:
            with torch.no_grad():
                x_hat = self.edm.denoiser(x, self.model, sig)
        if self.data_consistency or self.xi == 0:
            # guided branch: only with type "always" (:100); replacement branch: at EVERY evaluation, whatever the
            # type (:141-147) -- the reference raises AttributeError there when no projection was ever defined
            if not (self.data_consistency or self.data_consistency_end):
                raise AttributeError("proj_convex_set is undefined: data_consistency.use is False (:338)")
            x_hat = self.project(x_hat)                                                     # (:100, :146, :343 / :360)
        self._rid_pocs = x_hat.detach().clone()
        if self.trace is not None:
            self.trace.append(x_hat.detach().clone())
        return (x_hat - x) / t_i ** 2                                                       # (:105)

    def _randn(self, shape, gens):
        if gens is None:
            return torch.randn(shape)
        return torch.cat([torch.randn([1, shape[1]], generator=g) for g in gens], dim=0)

    # -- the loop (:178-262) -----------------------------------------------------------------------
    def predict_unconditional(self, shape, seeds: Optional[List[int]] = None, record: bool = False):
        """(:155-162) y = None, degradation = None"""
        self.y = self.degradation = None
        self._shape = tuple(shape)
        return self._predict(seeds, record)

    def predict_inpainting(self, y_masked, mask, seeds: Optional[List[int]] = None, record: bool = False, rid: bool = False):
        self._want_rid = rid
        self.y, self.mask = y_masked, mask
        self.degradation = lambda x: self.mask * x
        self.project = lambda x: self.smask * self.y + (1 - self.smask) * x
        if self.data_consistency or self.data_consistency_end:
            self.smask = smooth_mask_rows(mask, self.hann_size) if self.smooth else mask
        return self._predict(seeds, record)

    def predict_resample(self, y, shape, degradation, seeds: Optional[List[int]] = None, record: bool = False):
        """(:164-173) generic entry point: observations y[B, ...], signal shape (B, L), degradation = any differentiable torch callable.
        No projection exists for it (proj_convex_set is only defined by the two inpainting entry points)."""
        self.y, self.degradation, self._shape = y, degradation, tuple(shape)

        def no_projection(x):
            raise AttributeError("proj_convex_set is undefined for predict_resample")
        self.project = no_projection
        return self._predict(seeds, record, shape=tuple(shape))

    def predict_spectrogram_inpainting(self, y_masked, mask, stft=(1024, 256, 1024), seeds: Optional[List[int]] = None,
                                       record: bool = False):
        """(:348-364) degradation = STFT-domain mask, projection = y + x - A(x)."""
        self.y, self.mask = y_masked, mask
        n_fft, hop, win = stft
        self.degradation = lambda x: spectral_mask_apply(x, self.mask, n_fft, hop, win)
        self.project = lambda x: self.y + x - self.degradation(x)
        return self._predict(seeds, record)

    def _predict(self, seeds, record, shape=None):
        y_masked = self.y
        self.trace = [] if record else None
        if shape is None:
            shape = y_masked.shape if y_masked is not None else self._shape
        rid = getattr(self, "_want_rid", False)
        self._want_rid = False
        if rid:
            assert self.xi > 0, "the reference's rid buffers only exist on the guided branch (:217)"
            R = {k: torch.zeros((self.nb_steps,) + tuple(shape)) for k in ("denoised", "grads", "grad_update", "pocs", "xt", "xt2")}
        gens = None if seeds is None else [torch.Generator().manual_seed(int(s)) for s in seeds]
        t = self.edm.create_schedule(self.nb_steps)
        x = self._randn(shape, gens) * t[0]
        gamma = self.edm.get_gamma(t)
        for i in range(self.nb_steps):
            if gamma[i] == 0:
                t_hat = t[i]
            else:
                t_hat = t[i] + gamma[i] * t[i]
                eps = self._randn(shape, gens) * self.edm.Snoise
                x = x + ((t_hat ** 2 - t[i] ** 2) ** (1 / 2)) * eps
            if rid:
                R["xt"][i] = x
            score = self.get_score(x, t_hat)
            if rid:
                R["denoised"][i], R["grads"][i], R["grad_update"][i] = self._rid_last
                R["pocs"][i] = self._rid_pocs
            d = -t_hat * score
            h = t[i + 1] - t_hat
            if t[i + 1] != 0 and self.order == 2:
                x_prime = x + h * d
                score = self.get_score(x_prime, t[i + 1])
                d_prime = -t[i + 1] * score
                x = x + h * ((1 / 2) * d + (1 / 2) * d_prime)
            else:
                x = x + h * d
            if rid:
                R["xt2"][i] = x
            if self.states is not None:
                self.states.append(x.detach().clone())
            if self.stop_after is not None and i + 1 >= self.stop_after:
                return x.detach()
        if self.data_consistency_end and self.y is not None:                                # (:252)
            x = self.project(x)
        if rid:                                                                             # (:260)
            return x.detach(), R["denoised"], R["grads"], R["grad_update"], R["pocs"], R["xt"], R["xt2"], t
        return x.detach()
