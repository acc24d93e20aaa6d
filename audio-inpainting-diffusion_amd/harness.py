"""Host-side harness pieces around the sampler.

* ``load_checkpoint(net, path_or_state)`` -- what ``Tester.load_checkpoint`` does for the inpainting tester
  (testing/tester_inpainting.py:195-202 -> utils/training_utils.py:214-289 with ``ema=network``): load
  ``state['ema']`` strictly, else non-strictly, else key-by-key where the shapes agree.  Our network carries the
  reference's state_dict keys, so released checkpoints go through the first strategy; kernel-side weight packs are
  rebuilt from the parameter versions on the next evaluation.
* ``centre_gap_window`` / ``inpaint_long`` -- the long-file path of the tester (:382-418): the gap sits at the centre of
  the file, a window of ``audio_len`` samples centred on it goes through the sampler, the result is stitched back.
  The reference handles one file per sampler call; here B files form one batch with per-item masks.
No device code here; everything numerical happens in the sampler / network.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch


def load_checkpoint(net, state, key: str = "ema") -> Tuple[int, str]:
    """Returns (iteration stored in the checkpoint or 0, strategy used: 'strict' | 'non-strict' | 'shape-matched')."""
    if isinstance(state, (str, bytes)) or hasattr(state, "__fspath__"):
        state = torch.load(state, map_location="cpu")
    it = int(state["it"]) if "it" in state else 0
    sd = state[key]
    try:
        net.load_state_dict(sd)                                      # attempt 1 (:229-236)
        return it, "strict"
    except Exception:
        pass
    try:
        net.load_state_dict(sd, strict=False)                        # attempt 2 (:242-249): missing / unexpected keys tolerated
        return it, "non-strict"
    except Exception:
        pass
    own = net.state_dict()                                           # attempt 3 (:256-283): copy what matches by name and shape
    n = 0
    for name, param in sd.items():
        if name in own and own[name].shape == param.shape:
            own[name] = param
            n += 1
    if n == 0:
        raise ValueError("No parameters were loaded")
    net.load_state_dict(own)
    return it, "shape-matched"


def centre_gap_window(length: int, audio_len: int, gap: int) -> Tuple[int, int]:
    """(start of the gap, start of the audio_len window) for a file of ``length`` samples (:399-411)."""
    if length < audio_len:
        raise ValueError(f"file shorter ({length}) than the model's segment length ({audio_len})")
    return int(length // 2 - gap // 2), int(length // 2 - audio_len // 2)


def inpaint_long(sampler, files: Sequence[torch.Tensor], gap_ms: float, sample_rate: float, audio_len: int,
                 device="cuda") -> List[torch.Tensor]:
    """Inpaint a centred gap of ``gap_ms`` in every 1-D waveform of ``files`` (any lengths >= audio_len) and return the
    full-length results.  One sampler call for the whole batch."""
    gap = int(gap_ms * sample_rate / 1000)
    segs, masks, where = [], [], []
    for x in files:
        x = x.reshape(-1).float()
        g0, s0 = centre_gap_window(x.numel(), audio_len, gap)
        m = torch.ones(x.numel())
        m[g0:g0 + gap] = 0
        segs.append(x[s0:s0 + audio_len])
        masks.append(m[s0:s0 + audio_len])
        where.append(s0)
    seg, mask = torch.stack(segs).to(device), torch.stack(masks).to(device)
    pred = sampler.predict_inpainting(seg * mask, mask).cpu()
    out = []
    for x, s0, p in zip(files, where, pred):
        x = x.reshape(-1).float()
        out.append(torch.cat((x[:s0], p, x[s0 + audio_len:])))       # (:415)
    return out
