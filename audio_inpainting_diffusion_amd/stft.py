"""STFT-domain masking operator of spectrogram inpainting on MI355X.

``SpectralMask(mask[F,T], L, n_fft, hop, win_length)`` is the linear operator of
``Sampler.apply_spectral_mask`` (testing/edm_sampler_inpainting.py:271-290): zero-pad to a multiple of ``n_fft``
(:283 -- a full extra block when L already is one), ``torch.stft`` (centred, reflect padding, periodic Hann),
multiply by the mask, ``torch.istft``, crop.  ``apply`` runs it with two kernels (``aid_stft_frames``,
``aid_stft_ola``), ``adjoint`` runs A^T -- the same kernels with the border handling transposed -- which is what
the reference obtains from ``torch.autograd`` inside the guidance gradient (:65-81).  Host tables (window,
twiddles, istft envelope) are float64 numpy rounded once to fp32.  GPU only; no fallback.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


def stft_tables(audio_len: int, n_fft: int, hop: int, win_length: int):
    """(window[n_fft], istft envelope[Lp], twiddles[n_fft/2, 2]) in float64; the window is rounded to fp32 first so
    the envelope matches what the kernels multiply with."""
    Lp = audio_len + (n_fft - audio_len % n_fft)
    n_frames = 1 + Lp // hop
    w = np.zeros(n_fft, dtype=np.float64)                     # torch.stft centre-pads a shorter window
    off = (n_fft - win_length) // 2
    w[off:off + win_length] = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(win_length) / win_length)   # periodic Hann
    w = w.astype(np.float32).astype(np.float64)
    env = np.zeros(Lp + n_fft, dtype=np.float64)              # istft: overlap-added squared window, trimmed n_fft/2
    for n in range(n_frames):
        env[n * hop:n * hop + n_fft] += w * w
    env = env[n_fft // 2:n_fft // 2 + Lp]
    if np.abs(env).min() < 1e-11:
        raise _lib.AidError("the istft window envelope vanishes for this n_fft / hop / window")
    m = np.arange(n_fft // 2)
    tw = np.stack([np.cos(2 * np.pi * m / n_fft), -np.sin(2 * np.pi * m / n_fft)], axis=1)
    return w, env, tw


class SpectralMask:
    def __init__(self, mask: torch.Tensor, audio_len: int, n_fft: int = 1024, hop_length: int = 256,
                 win_length: int = 1024, window: str = "hann", device="cuda"):
        if window != "hann":
            raise NotImplementedError("Only hann window is implemented for now")       # as the reference (:276)
        if n_fft & (n_fft - 1) or not 16 <= n_fft <= 8192:
            raise _lib.AidError("n_fft must be a power of two in [16, 8192]")
        self.L, self.n_fft, self.hop, self.win_length = int(audio_len), int(n_fft), int(hop_length), int(win_length)
        self.Lp = self.L + (n_fft - self.L % n_fft)
        if self.Lp % self.hop:
            raise _lib.AidError("hop_length must divide the padded length")
        self.n_frames = 1 + self.Lp // self.hop
        self.F = n_fft // 2 + 1
        dev = torch.device(device)
        if dev.type != "cuda":
            raise _lib.AidError("SpectralMask runs on the GPU only (no CPU fallback)")
        w, env, tw = stft_tables(self.L, n_fft, hop_length, win_length)
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        self.window, self.inv_env, self.twiddle = f32(w), f32(1.0 / env), f32(tw)
        self.set_mask(mask, dev)
        self._frames = {}

    def set_mask(self, mask, dev=None):
        dev = self.window.device if dev is None else dev
        m = mask.detach().to(dev, torch.float32).contiguous()
        if m.dim() == 2:
            m = m.unsqueeze(0)
        if m.shape[1] != self.F or m.shape[2] != self.n_frames:
            raise _lib.AidError(f"spectral mask must be [{self.F}, {self.n_frames}], got {tuple(m.shape[1:])}")
        self.mask = m

    def _params(self, x, out, adjoint, c0=1.0, add1=None, add2=None):
        B = x.shape[0]
        key = (B, torch.cuda.current_stream().cuda_stream)          # (sub-batches on concurrent streams must not share the frame scratch)
        fr = self._frames.get(key)
        if fr is None:
            fr = self._frames[key] = torch.empty(B, self.n_frames, self.n_fft, device=x.device, dtype=torch.float32)
        if self.mask.shape[0] not in (1, B):
            raise _lib.AidError("per-item spectral masks must match the batch size")
        return _lib.StftParams(x.data_ptr(), fr.data_ptr(), out.data_ptr(), self.window.data_ptr(), self.twiddle.data_ptr(),
                               self.inv_env.data_ptr(), self.mask.data_ptr(), self.mask.stride(0) if self.mask.shape[0] > 1 else 0,
                               self.mask.stride(1), _lib.ptr(add1), _lib.ptr(add2), float(c0), B, self.L, self.Lp,
                               self.n_fft, self.hop, self.n_frames, int(adjoint))

    def _run(self, x, adjoint, c0=1.0, add1=None, add2=None):
        if x.dim() != 2 or x.shape[1] != self.L or x.dtype != torch.float32 or not x.is_cuda:
            raise _lib.AidError(f"expected a float32 GPU tensor [B, {self.L}]")
        x = x.contiguous()
        out = torch.empty_like(x)
        p = self._params(x, out, adjoint, c0, add1, add2)
        _lib.call("aid_stft_frames", p)
        _lib.call("aid_stft_ola", p)
        return out

    @property
    def shared_mask(self) -> bool:
        """one mask for every item (sub-batches can then use this operator concurrently)"""
        return self.mask.shape[0] == 1

    def apply(self, x):
        """A(x)"""
        return self._run(x, 0)

    def adjoint(self, g):
        """A^T(g)"""
        return self._run(g, 1)

    def project(self, x, y):
        """y + x - A(x)      (predict_spectrogram_inpainting's proj_convex_set, :360)"""
        return self._run(x, 0, c0=-1.0, add1=y.contiguous(), add2=x)

    __call__ = apply
