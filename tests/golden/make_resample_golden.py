#!/usr/bin/env python
"""Dump ``torchaudio.functional.resample`` (absent from this repository's build image) for the rate pairs the reference's
``resample_batch`` uses (utils/training_utils.py:148,152,176,187,198,202), so that oracle/resample.py and the HIP polyphase
resampler (harness.resample_batch -> aid_resample_poly) can be pinned to the real thing.

    python tests/golden/make_resample_golden.py        # on a machine with torchaudio; writes tests/golden/resample_ref.npz

tests/test_resample_conformance.py consumes the file (and skips, saying so, while it does not exist)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
PAIRS = [(2, 1), (320, 147), (160, 147), (44100, 16000), (48000, 16000)]


def main():
    try:
        import torchaudio
    except ImportError:
        sys.exit("torchaudio is not installed here: run this script on a machine that has it")
    rng = np.random.Generator(np.random.PCG64(77))
    x = torch.from_numpy(rng.standard_normal((2, 30000)).astype(np.float32))
    d = {"seed": np.array(77), "shape": np.array(x.shape), "torchaudio_version": np.array(torchaudio.__version__),
         "pairs": np.array(PAIRS, dtype=np.int64)}
    for o, n in PAIRS:
        d[f"y_{o}_{n}"] = torchaudio.functional.resample(x, o, n).numpy()
    np.savez_compressed(os.path.join(HERE, "resample_ref.npz"), **d)
    print("wrote resample_ref.npz", {k: v.shape for k, v in d.items() if k.startswith("y_")})


if __name__ == "__main__":
    main()
