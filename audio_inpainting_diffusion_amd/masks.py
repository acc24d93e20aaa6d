"""Inpainting masks as the reference tester builds them (testing/tester_inpainting.py:231-254, ``prepare_mask``):
ones[1,L] with zeros over the gap(s).  'long': one gap of ``int(gap_ms*fs/1000)`` samples, centred unless a start
is given; 'short': ``num_gaps`` gaps at uniformly random starts (``torch.randint(0, L-gap, (num_gaps,))``).
``spectral_mask``: the rectangular time-frequency mask of ``prepare_spectral_mask`` (:256-294)."""
from __future__ import annotations

import torch


def long_gap_mask(audio_len: int, sample_rate: float, gap_ms: float, start_ms=None) -> torch.Tensor:
    mask = torch.ones((1, audio_len))
    gap = int(gap_ms * sample_rate / 1000)
    start = int(audio_len // 2 - gap // 2) if start_ms is None else int(start_ms * sample_rate / 1000)
    mask[..., start:start + gap] = 0
    return mask


def short_gaps_mask(audio_len: int, sample_rate: float, gap_ms: float, num_gaps: int = 4, generator=None) -> torch.Tensor:
    mask = torch.ones((1, audio_len))
    gap = int(gap_ms * sample_rate / 1000)
    starts = torch.randint(0, audio_len - gap, (num_gaps,), generator=generator)
    for i in range(num_gaps):
        mask[..., starts[i]:starts[i] + gap] = 0
    return mask


def mask_from_args(args, generator=None) -> torch.Tensor:
    inp = args.tester.inpainting
    if inp.mask_mode == "long":
        s = inp.long.start_gap_idx
        return long_gap_mask(args.exp.audio_len, args.exp.sample_rate, inp.long.gap_length, None if s in ("None", None) else s)
    if inp.mask_mode == "short":
        return short_gaps_mask(args.exp.audio_len, args.exp.sample_rate, inp.short.gap_length, int(inp.short.num_gaps), generator)
    raise NotImplementedError(inp.mask_mode)


def spectral_mask(audio_len: int, sample_rate: float, n_fft: int = 1024, hop_length: int = 256, time_mask_ms: float = 2000,
                  fmin_hz: float = 300, fmax_hz: float = 2000, start_ms=None) -> torch.Tensor:
    """ones[F, T] with zeros over [fmin, fmax) x the (centred) time gap; F = n_fft/2+1, T = 1 + Lp/hop where Lp is
    the length after the reference's zero-padding to a multiple of n_fft (tester_inpainting.py:267-291)."""
    Lp = audio_len + (n_fft - audio_len % n_fft)
    F, T = n_fft // 2 + 1, 1 + Lp // hop_length
    A = torch.ones((F, T))
    freqs = torch.fft.fftfreq(n_fft, d=1 / sample_rate)
    fmin_idx = int(torch.argmin(torch.abs(freqs - fmin_hz)))
    fmax_idx = int(torch.argmin(torch.abs(freqs - fmax_hz)))
    gap = int(time_mask_ms * sample_rate / 1000)
    if start_ms is None:
        start = int(audio_len // 2 - gap // 2) // hop_length
    else:
        start = int(start_ms * sample_rate / 1000) // hop_length
    end = start + gap // hop_length
    A[fmin_idx:fmax_idx, start:end] = 0
    return A


def spectral_mask_from_args(args) -> torch.Tensor:
    sp = args.tester.spectrogram_inpainting
    s = sp.time_start_idx
    return spectral_mask(args.exp.audio_len, args.exp.sample_rate, sp.stft.n_fft, sp.stft.hop_length, sp.time_mask_length,
                         sp.min_masked_freq, sp.max_masked_freq, None if s in ("None", None) else s)
