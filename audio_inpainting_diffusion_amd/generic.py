"""GenericModelAdapter -- drives ANY torch denoiser through the MI355X sampler loop.  **Not the HIP network path.**

``sampler.Sampler`` talks to its model through the fused entry points of the MI355X network (``denoise`` /
``denoise_guided``, network.py).  This adapter gives a foreign ``model(x[B,L], cnoise[B,1]) -> [B,L]`` (anything the
reference's ``EDM.denoiser`` accepts, diff_params/edm.py:133-148) the same two entry points, evaluated with torch
eager ops and ``torch.autograd`` exactly as the reference's ``get_score`` / ``get_score_rec_guidance`` do
(testing/edm_sampler_inpainting.py:57-153).  It exists so that

  * the sampler loop itself (churn, Heun, guidance scaling, projection, RNG order) can be pinned on the GPU against
    the reference's own trajectories of a toy denoiser (tests/golden/sampler_*.npz, tests/test_gpu_generic.py);
  * users can sample from a model that has no MI355X implementation.

It is never selected implicitly: ``Sampler`` raises for a model without the fused entry points, the caller wraps it.
"""
from __future__ import annotations

import torch


class _OperatorFn(torch.autograd.Function):
    """A(x) with A^T as its backward, for linear degradation objects exposing ``apply`` / ``adjoint`` (stft.SpectralMask)."""

    @staticmethod
    def forward(ctx, x, op):
        ctx.op = op
        return op.apply(x.detach().contiguous())

    @staticmethod
    def backward(ctx, g):
        return ctx.op.adjoint(g.contiguous()), None


class GenericModelAdapter:
    def __init__(self, model, diff_params):
        self.model, self.diff_params = model, diff_params
        self.CQTransform = getattr(model, "CQTransform", None)

    @staticmethod
    def _col(v, B, device):
        import numbers
        if isinstance(v, numbers.Real) or (torch.is_tensor(v) and v.numel() == 1) or (not torch.is_tensor(v) and getattr(v, "size", 2) == 1):
            return torch.full((B, 1), float(v), dtype=torch.float32, device=device)
        if not torch.is_tensor(v) or v.numel() != B:
            raise ValueError(f"EDM scalar must be a number or a tensor of {B} elements")
        return v.to(device=device, dtype=torch.float32).reshape(B, 1)

    def _eval(self, x, cnoise, cin, cskip, cout, hpf):
        B = x.shape[0]
        c = [self._col(v, B, x.device) for v in (cnoise, cin, cskip, cout)]
        x_hat = c[2] * x + c[3] * self.model(c[1] * x, c[0]).to(x.dtype)          # EDM.denoiser (edm.py:145-148)
        if hpf:
            x_hat = self.model.CQTransform.apply_hpf_DC(x_hat)                    # (:62-63, :122-123)
        return x_hat

    @torch.no_grad()
    def denoise(self, x, cnoise, cin, cskip, cout, hpf: bool):
        return self._eval(x, cnoise, cin, cskip, cout, hpf)

    def denoise_guided(self, x, cnoise, cin, cskip, cout, hpf: bool, y, mask, degradation=None, norm_type=2, beta=1.0):
        """(x_hat, d norm / d x, norm[B]) by torch.autograd (:57-81), norms reduced per item."""
        x = x.detach().requires_grad_()
        with torch.enable_grad():
            x_hat = self._eval(x, cnoise, cin, cskip, cout, hpf)
            den = mask * x_hat if degradation is None else _OperatorFn.apply(x_hat, degradation)
            B = x.shape[0]
            if norm_type == "smoothl1":
                norm = torch.nn.functional.smooth_l1_loss(y, den, reduction="none", beta=beta).reshape(B, -1).sum(dim=1)
            else:
                diff = y - den                                  # (:67-70: 3-D observations take the induced matrix norm over dim = (1, 2), as in the reference)
                norm = torch.linalg.norm(diff, dim=(1, 2) if diff.dim() == 3 else 1, ord=norm_type)
            g = torch.autograd.grad(outputs=norm.sum(), inputs=x)[0]
        return x_hat.detach(), g.detach().contiguous(), norm.detach()
