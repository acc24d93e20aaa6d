"""Seeded, platform-independent parameter initialiser.

The reference draws its random-init weights from torch's global RNG at construction time
(networks/unet_cqt_oct_with_projattention_adaLN_2.py:20-34,599-600); those bits depend on the torch build.
Benchmarks and parity fixtures need identical weights in the dev container, on the GPU box and inside the
imported reference, so weights are produced by a counter-based generator (splitmix64 on
``(seed, crc32(state_dict key), element index)``) with the reference's distribution family:

  * conv / linear ``weight``  : U(-1,1) * sqrt(3/fan_in) * sqrt(1/3)        (kaiming_uniform, init_weight sqrt(1/3), :599)
  * ``gate*.weight``          : U(-1,1) * sqrt(3/fan_in) * gate_scale        (reference gate_scale = 1e-7, :600)
  * ``affine*.weight``        : as a plain weight, times ``affine_scale``    (reference 1.0)
  * ``embedding.RFF_freq``    : 16 * N(0,1)                                   (:176-177)
  * ``*.gamma`` = 1, ``*.bias`` = 0 (init_bias=0, :34), resampler ``kernel`` buffers = cubic taps (:514-515)
  * optional switches: ``attn_block.qk.bias`` (bias_qkv) U(-0.5,0.5); ``rel_pos.relative_attention_bias.weight`` (use_rel_pos) as a
    plain weight; ``freq_encodings.*.RFF_freq`` 16*N(0,1) and ``.embeddings`` U(-1,1) (use_fencoding)

Parity tests use gate_scale ~ 10 and affine_scale ~ 10 so that every conv / attention branch contributes
O(1) to the output (with the reference's 1e-7 gates a broken kernel would be invisible at 1e-4 rel-L2).
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Iterable, Tuple

import numpy as np
import torch

_CUBIC = (-0.01171875, -0.03515625, 0.11328125, 0.43359375, 0.43359375, 0.11328125, -0.03515625, -0.01171875)
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _uniform01(seed: int, stream: int, n: int, offset: int = 0) -> np.ndarray:
    """n float64 values in (0,1), a pure function of (seed, stream, offset+index)."""
    key = _splitmix64(np.array([(seed * 0x632BE59BD9B4E019 + stream * 0xD1342543DE82EF95) & 0xFFFFFFFFFFFFFFFF],
                               dtype=np.uint64))[0]
    out = np.empty(n, dtype=np.float64)
    CH = 1 << 22
    for s in range(0, n, CH):
        e = min(n, s + CH)
        with np.errstate(over="ignore"):
            ctr = (np.arange(s + offset, e + offset, dtype=np.uint64) + key) & _M64
        z = _splitmix64(ctr)
        out[s:e] = ((z >> np.uint64(40)).astype(np.float64) + 0.5) / float(1 << 24)
    return out


def seeded_normal(seed: int, stream: int, n: int) -> np.ndarray:
    """n float32 N(0,1) values (Box-Muller on the counter-based uniforms); used for synthetic waveforms."""
    u1, u2 = _uniform01(seed, stream, n), _uniform01(seed, stream, n, offset=n)
    return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)).astype(np.float32)


def seeded_tensor(name: str, shape: Tuple[int, ...], index: int, seed: int, gate_scale: float = 1e-7,
                  affine_scale: float = 1.0) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    leaf = name.split(".")[-1]
    if leaf == "gamma":
        return torch.ones(shape, dtype=torch.float32)
    if leaf == "bias":
        if name.split(".")[-2:-1] == ["qk"]:          # attention_dict.bias_qkv: a zero bias would make the switch untestable
            return torch.from_numpy((0.5 * (2.0 * _uniform01(seed, index, n) - 1.0)).astype(np.float32)).reshape(shape)
        return torch.zeros(shape, dtype=torch.float32)
    if leaf == "embeddings":                          # use_fencoding: sin / cos tables (any values in [-1, 1] exercise the path)
        return torch.from_numpy((2.0 * _uniform01(seed, index, n) - 1.0).astype(np.float32)).reshape(shape)
    if leaf == "kernel":
        return torch.tensor(_CUBIC, dtype=torch.float32).reshape(shape)
    if leaf == "RFF_freq":
        u1, u2 = _uniform01(seed, index, n), _uniform01(seed, index, n, offset=n)
        z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
        return torch.from_numpy((16.0 * z).astype(np.float32)).reshape(shape)
    assert leaf == "weight", name
    fan_in = int(np.prod(shape[1:]))
    u = 2.0 * _uniform01(seed, index, n) - 1.0
    scale = math.sqrt(3.0 / fan_in)
    parent = name.split(".")[-2] if "." in name else ""
    owner = name.split(".")[-3] if name.count(".") >= 2 else ""
    if parent.startswith("gate") or owner.startswith("gate"):
        scale *= gate_scale
    else:
        scale *= math.sqrt(1.0 / 3.0)
        if parent.startswith("affine") or owner.startswith("affine"):
            scale *= affine_scale
    return torch.from_numpy((u * scale).astype(np.float32)).reshape(shape)


def seeded_state_dict(shapes: Iterable[Tuple[str, Tuple[int, ...]]], seed: int = 0, gate_scale: float = 1e-7,
                      affine_scale: float = 1.0) -> Dict[str, torch.Tensor]:
    """shapes: (key, shape) pairs -> {key: tensor}.  The per-tensor stream is crc32(key), so the result does not
    depend on the order in which a module happens to register its parameters."""
    return {k: seeded_tensor(k, tuple(s), zlib.crc32(k.encode()), seed, gate_scale, affine_scale) for k, s in shapes}


@torch.no_grad()
def seeded_init_(module: torch.nn.Module, seed: int = 0, gate_scale: float = 1e-7, affine_scale: float = 1.0):
    """Fill ``module`` in place (each tensor's stream is derived from its state_dict key)."""
    sd = module.state_dict()
    new = seeded_state_dict([(k, tuple(v.shape)) for k, v in sd.items()], seed, gate_scale, affine_scale)
    module.load_state_dict(new, strict=True)
    return module
