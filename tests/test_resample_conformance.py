"""Conformance of the waveform resampler with ``torchaudio.functional.resample``, which the
reference calls and this image lacks.  Consumes tests/golden/resample_ref.npz (tests/golden/make_resample_golden.py, run
where torchaudio exists); skips -- loudly -- while that fixture is absent: the resampler's parity is then UNPINNED."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2

PATH = os.path.join(GOLDEN, "resample_ref.npz")
NO_FIXTURE = "no tests/golden/resample_ref.npz: resampler parity vs torchaudio is UNPINNED -- run tests/golden/make_resample_golden.py where torchaudio is installed"


def _load():
    if not os.path.exists(PATH):
        pytest.skip(NO_FIXTURE)
    z = np.load(PATH)
    rng = np.random.Generator(np.random.PCG64(int(z["seed"])))
    x = torch.from_numpy(rng.standard_normal(tuple(z["shape"])).astype(np.float32))
    return z, x


def test_oracle_resampler_reproduces_torchaudio():
    from oracle.resample import resample
    z, x = _load()
    for o, n in z["pairs"]:
        ref = z[f"y_{o}_{n}"]
        got = resample(x.double(), int(o), int(n))
        assert got.shape == ref.shape and rel_l2(got, ref) < 1e-5, (o, n)


@pytest.mark.gpu
def test_hip_resampler_reproduces_torchaudio():
    from audio_inpainting_diffusion_amd.harness import resample
    z, x = _load()
    for o, n in z["pairs"]:
        ref = z[f"y_{o}_{n}"]
        got = resample(x.cuda(), int(o), int(n)).cpu()
        assert got.shape == ref.shape and rel_l2(got, ref) < 1e-5, (o, n)
