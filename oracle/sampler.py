"""CPU oracle: the EDM inpainting sampling loop (Heun / stochastic churn / reconstruction guidance /
data-consistency projection).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned by tests/golden/sampler_*.npz (trajectories of
the reference's own ``Sampler`` driving a toy denoiser).

Follows reference testing/edm_sampler_inpainting.py: get_score_rec_guidance :57-113, get_score :115-153,
predict :178-262, apply_mask :264-269, prepare_smooth_mask :302-325, predict_inpainting :327-346.

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


class OracleSampler:
    def __init__(self, model, edm, T=35, order=2, xi=0.25, norm=2, data_consistency=True, smooth=True,
                 hann_size=50, filter_out_cqt_DC_Nyq=True, audio_len=None):
        self.model, self.edm = model, edm
        self.nb_steps, self.order, self.xi, self.norm = T, order, xi, norm
        self.data_consistency, self.smooth, self.hann_size = data_consistency, smooth, hann_size
        self.filter_hpf = filter_out_cqt_DC_Nyq
        self.audio_len = audio_len
        self.trace = None

    # -- one denoiser evaluation (:115-153) ------------------------------------------------------
    def get_score(self, x, t_i):
        B = x.shape[0]
        sig = t_i.reshape(1, 1).expand(B, 1)
        if self.xi > 0:
            x = x.detach().requires_grad_()
            x_hat = self.edm.denoiser(x, self.model, sig)
            if self.filter_hpf:
                x_hat = self.model.CQTransform.apply_hpf_DC(x_hat)
            norm = torch.linalg.norm(self.y - self.mask * x_hat, dim=1, ord=self.norm)      # [B]  (:75)
            g = torch.autograd.grad(norm.sum(), x)[0]                                       # per-item grads
            L = self.audio_len if self.audio_len is not None else x.shape[-1]
            normguide = torch.linalg.norm(g, dim=1, keepdim=True) / L ** 0.5                # (:83) per item
            s = t_i * self.xi / (normguide + 1e-6)                                          # (:87)
            x_hat = (x_hat - s * g).detach()                                                # (:97)
            x = x.detach()
        else:
            with torch.no_grad():
                x_hat = self.edm.denoiser(x, self.model, sig)
        if self.data_consistency:
            x_hat = self.smask * self.y + (1 - self.smask) * x_hat                          # (:343)
        if self.trace is not None:
            self.trace.append(x_hat.detach().clone())
        return (x_hat - x) / t_i ** 2                                                       # (:105)

    def _randn(self, shape, gens):
        if gens is None:
            return torch.randn(shape)
        return torch.cat([torch.randn([1, shape[1]], generator=g) for g in gens], dim=0)

    # -- the loop (:178-262) -----------------------------------------------------------------------
    def predict_inpainting(self, y_masked, mask, seeds: Optional[List[int]] = None, record: bool = False):
        self.y, self.mask = y_masked, mask
        if self.data_consistency:
            self.smask = smooth_mask_rows(mask, self.hann_size) if self.smooth else mask
        self.trace = [] if record else None
        shape = y_masked.shape
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
            score = self.get_score(x, t_hat)
            d = -t_hat * score
            h = t[i + 1] - t_hat
            if t[i + 1] != 0 and self.order == 2:
                x_prime = x + h * d
                score = self.get_score(x_prime, t[i + 1])
                d_prime = -t[i + 1] * score
                x = x + h * ((1 / 2) * d + (1 / 2) * d_prime)
            else:
                x = x + h * d
        return x.detach()
