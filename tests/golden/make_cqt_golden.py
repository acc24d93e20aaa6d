#!/usr/bin/env python
"""Dump the reference's REAL constant-Q transform (the external package ``cqt_nsgt_pytorch``, which this repository's
build image does not have) into fixtures that pin our CQT to it.

Run this on any machine where ``pip install cqt_nsgt_pytorch`` works (CPU is enough), from the repository root:

    python tests/golden/make_cqt_golden.py            # writes tests/golden/cqt_ref_<cfg>.npz

then run ``python -m pytest tests/test_cqt_conformance.py`` (CPU) and, on the MI355X, ``-m gpu``: the tests pick up
every ``cqt_ref_*.npz`` (they skip, loudly, while none exists), report which rule set of
``audio_inpainting_diffusion_amd/cqt.py::RULE_PRESETS`` reproduces the package's frame (window lengths, centre bins,
window samples, octave lengths, DC/Nyquist projector) and then hold ``fwd`` / ``bwd`` / ``apply_hpf_DC`` to 1e-5.

Calls are exactly the reference's: constructor networks/unet_cqt_oct_with_projattention_adaLN_2.py:620, ``fwd`` :743,
``bwd`` :841, ``apply_hpf_DC`` testing/edm_sampler_inpainting.py:63.  Only DATA is written (inputs are regenerated from a
numpy PCG64 seed, outputs stored as float32 / complex64)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
CONFIGS = [("small", 4, 8, 22050, 4096), ("cfgA_22k", 7, 64, 22050, 184184), ("cfgA_16k", 7, 64, 16000, 184184),
           ("cfgB_44k_4s", 8, 64, 44100, 184184), ("cfgB_44k_8s", 8, 64, 44100, 368368)]


def _attr(obj, *names):
    for n in names:
        if hasattr(obj, n):
            return getattr(obj, n)
    return None


PINNED_VERSION = "0.0.8"     # the version the reference's own notebook output records: /root/reference/notebooks/demo_inpainting_spectrogram.ipynb
                             # cell 4: "Downloading cqt-nsgt-pytorch-0.0.8.tar.gz (12 kB)" (wheel built there: cqt_nsgt_pytorch-0.0.8-py3-none-any.whl)


def _package_version(mod):
    try:
        from importlib.metadata import version
        return version("cqt_nsgt_pytorch")
    except Exception:
        return str(getattr(mod, "__version__", "unknown"))


def _package_sha256(mod):
    """sha256 over the package's .py files in sorted order (identifies the exact source the dump was made with)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(os.path.dirname(mod.__file__), "**", "*.py"), recursive=True)):
        h.update(os.path.relpath(f, os.path.dirname(mod.__file__)).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def _arr(v):
    if v is None:
        return np.zeros(0)
    if isinstance(v, torch.Tensor):
        return v.detach().cpu().numpy()
    return np.asarray(v)


def main():
    try:
        from cqt_nsgt_pytorch import CQT_nsgt
    except ImportError:
        sys.exit("cqt_nsgt_pytorch is not installed here: run this script on a machine that has it (pip install cqt_nsgt_pytorch)")
    import cqt_nsgt_pytorch
    version = _package_version(cqt_nsgt_pytorch)
    if version != PINNED_VERSION:
        msg = (f"cqt_nsgt_pytorch {version} is installed, the reference author ran {PINNED_VERSION} "
               f"(notebooks/demo_inpainting_spectrogram.ipynb cell 4 output): pip install cqt_nsgt_pytorch=={PINNED_VERSION}")
        if "--any-version" not in sys.argv:
            sys.exit(msg + "   (or pass --any-version: the dump then records the version it was made with and the conformance test says so)")
        print("WARNING:", msg)
    for tag, numocts, binsoct, fs, L in CONFIGS:
        cqt = CQT_nsgt(numocts, binsoct, mode="oct", window=("kaiser", 1), fs=fs, audio_len=L, dtype=torch.float32, device="cpu")
        rng = np.random.Generator(np.random.PCG64(1234))
        x = torch.from_numpy((0.063 * rng.standard_normal((2, L))).astype(np.float32))
        with torch.no_grad():
            C = cqt.fwd(x.unsqueeze(1))
            rt = cqt.bwd(C)
            hp = cqt.apply_hpf_DC(x)
            Cr = [torch.complex(torch.from_numpy(rng.standard_normal(tuple(c.shape)).astype(np.float32)),
                                torch.from_numpy(rng.standard_normal(tuple(c.shape)).astype(np.float32))) for c in C]
            yr = cqt.bwd(Cr)
        d = {"cfg": np.array([numocts, binsoct, fs, L], dtype=np.float64), "seed": np.array(1234),
             "package_version": np.array(version), "package_file_sha256": np.array(_package_sha256(cqt_nsgt_pytorch)),
             "n_oct": np.array(len(C)), "roundtrip": _arr(rt), "hpf": _arr(hp), "bwd_random": _arr(yr)}
        for o, c in enumerate(C):
            d[f"fwd_{o}"] = _arr(c).astype(np.complex64)
        # frame design, under whichever attribute names this version of the package uses
        g = _attr(cqt, "g")
        if g is not None:
            d["g_len"] = np.array([len(w) for w in g], dtype=np.int64)
            d["g_cat"] = np.concatenate([_arr(w).astype(np.float64) for w in g])
        gd = _attr(cqt, "gd")
        if gd is not None and not isinstance(gd, torch.Tensor):
            d["gd_cat"] = np.concatenate([_arr(w).astype(np.float64) for w in gd])
        d["M"] = _arr(_attr(cqt, "M")).astype(np.int64)
        wins = _attr(cqt, "wins")
        if wins is not None:
            d["win_first_bin"] = np.array([int(_arr(w)[0]) for w in wins], dtype=np.int64)       # first DFT bin each window covers
        d["frqs"] = _arr(_attr(cqt, "frqs")).astype(np.float64)
        d["q"] = _arr(_attr(cqt, "q")).astype(np.float64)
        d["Hhpf"] = _arr(_attr(cqt, "Hhpf")).astype(np.float64)
        d["size_per_oct"] = _arr(_attr(cqt, "size_per_oct")).astype(np.int64)
        out = os.path.join(HERE, f"cqt_ref_{tag}.npz")
        np.savez_compressed(out, **d)
        print("wrote", out, {k: v.shape for k, v in d.items() if hasattr(v, "shape") and v.ndim}, flush=True)


if __name__ == "__main__":
    main()
