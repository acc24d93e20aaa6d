"""Inpainting masks as the reference tester builds them (testing/tester_inpainting.py:231-254, ``prepare_mask``):
ones[1,L] with zeros over the gap(s).  'long': one gap of ``int(gap_ms*fs/1000)`` samples, centred unless a start
is given; 'short': ``num_gaps`` gaps at uniformly random starts (``torch.randint(0, L-gap, (num_gaps,))``)."""
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
