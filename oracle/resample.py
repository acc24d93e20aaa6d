"""CPU oracle: the waveform resampling the reference's tester applies before sampling.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  **Parity unpinned**: the reference delegates to
``torchaudio.functional.resample`` (utils/training_utils.py:148,152,160,162,176,187,198,202 ...), torchaudio is not installed
in this image and not vendored under /root/reference.  This file restates torchaudio's published default algorithm
(``sinc_interp_hann``, ``lowpass_filter_width=6``, ``rolloff=0.99``: a Hann-windowed sinc evaluated per output phase, applied as a
strided conv1d over the zero-padded waveform) and the case logic of ``resample_batch`` (:140-212).
"""
from __future__ import annotations

import math

import torch


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99, dtype=torch.float64):
    """(kernel [new, 1, 2*width + orig], width) for frequencies already divided by their gcd."""
    base_freq = min(orig_freq, new_freq) * rolloff
    width = math.ceil(lowpass_filter_width * orig_freq / base_freq)
    idx = torch.arange(-width, width + orig_freq, dtype=dtype)[None, None] / orig_freq
    t = torch.arange(0, -new_freq, -1, dtype=dtype)[:, None, None] / new_freq + idx
    t = (t * base_freq).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base_freq / orig_freq
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=dtype), t.sin() / t) * window * scale
    return kernels, width


def resample(x: torch.Tensor, orig_freq: int, new_freq: int) -> torch.Tensor:
    """x [B, L] -> [B, ceil(new*L/orig)]"""
    g = math.gcd(int(orig_freq), int(new_freq))
    o, n = int(orig_freq) // g, int(new_freq) // g
    if o == n:
        return x
    k, width = sinc_resample_kernel(o, n, dtype=x.dtype)
    L = x.shape[-1]
    xp = torch.nn.functional.pad(x, (width, width + o))
    y = torch.nn.functional.conv1d(xp[:, None], k, stride=o)                # [B, n, frames]
    y = y.transpose(1, 2).reshape(x.shape[0], -1)
    return y[..., :math.ceil(n * L / o)]


def resample_batch(audio: torch.Tensor, fs, fs_target: int, length_target: int) -> torch.Tensor:
    """resample_batch (:140-212) for batches whose items share one sampling rate (the only branch that returns a full batch:
    the mixed-rate loops of the reference return after their first item)."""
    fs = int(fs[0]) if hasattr(fs, "__len__") else int(fs)
    if fs_target == 22050 and fs == 44100:
        return resample(audio, 2, 1)[:, :length_target]
    if fs_target == 22050 and fs == 48000:
        return resample(audio, 160 * 2, 147)[:, :length_target]
    if fs_target == 44100 and fs == 44100:
        return audio[:, :length_target]
    if fs_target == 44100 and fs == 48000:
        return resample(audio, 160, 147)[:, :length_target]
    return resample(audio, fs, fs_target)[:, :length_target]
