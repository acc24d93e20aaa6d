"""CQT conformance kit (the reference's CQT is the un-vendored package ``cqt_nsgt_pytorch``).

Two layers:
 * always on: every rule preset of ``cqt.RULE_PRESETS`` is a working frame (perfect reconstruction through the
   DC/Nyquist projector, adjointness) in the oracle, and the product's vectorised plan agrees with the oracle's
   band-by-band design for every preset the HIP gather supports;
 * when ``tests/golden/cqt_ref_*.npz`` exist (dumped from the real package by tests/golden/make_cqt_golden.py):
   find the preset that reproduces the package's frame and hold fwd / bwd / apply_hpf_DC to 1e-5 -- on the CPU for the
   oracle, on the MI355X (``-m gpu``) for the HIP kernels.  Without fixtures these tests SKIP and say why: CQT parity
   stays "unpinned" until someone with the package runs the dump script.
"""
import dataclasses
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2

FIXTURES = sorted(glob.glob(os.path.join(GOLDEN, "cqt_ref_*.npz")))
PINNED_CQT_VERSION = "0.0.8"
NO_FIXTURE = ("no tests/golden/cqt_ref_*.npz: CQT parity vs cqt_nsgt_pytorch is UNPINNED -- run tests/golden/make_cqt_golden.py "
              "on a machine that has the package")


def _presets():
    from audio_inpainting_diffusion_amd.cqt import RULE_PRESETS
    return {k: dataclasses.asdict(v) for k, v in RULE_PRESETS.items()}


@pytest.mark.parametrize("preset", ["default", "nsgt_f_over_q", "nsgt_f_over_q_periodic", "nsgt_midpoint", "nsgt_midpoint_periodic"])
def test_every_rule_preset_is_a_frame_in_the_oracle(preset):
    from oracle.nsgt_cqt import OracleCQT
    L, fs = 16384, 22050
    q = OracleCQT(7, 8, "oct", ("kaiser", 1), fs, L, dtype=torch.float64, rules=_presets()[preset])
    assert all(a * 2 == b for a, b in zip(q.size_per_oct, q.size_per_oct[1:])), q.size_per_oct
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, L, dtype=torch.float64, generator=g)
    c = q.fwd(x[:, None])
    assert rel_l2(q.bwd(c)[:, 0], q.apply_hpf_DC(x)) < 1e-10


@pytest.mark.parametrize("preset", ["default", "nsgt_f_over_q", "nsgt_f_over_q_periodic", "nsgt_midpoint", "nsgt_midpoint_periodic"])
@pytest.mark.parametrize("cfg", [(7, 64, 22050, 184184), (8, 64, 44100, 368368), (3, 8, 22050, 2048)])
def test_plan_matches_oracle_design_for_every_supported_preset(preset, cfg):
    from audio_inpainting_diffusion_amd.cqt import CQTPlan
    from oracle.nsgt_cqt import OracleCQT
    no, bpo, fs, L = cfg
    try:
        P = CQTPlan(no, bpo, fs, L, ("kaiser", 1.0), rules=preset)
    except NotImplementedError as e:         # (the midpoint presets push the last band's window across Nyquist on some grids: oracle only)
        assert "midpoint" in preset, e
        pytest.skip(f"{preset}: {e}")
    O = OracleCQT(no, bpo, "oct", ("kaiser", 1), fs, L, rules=_presets()[preset])
    K = no * bpo
    assert P.T_oct == O.size_per_oct
    assert np.array_equal(P.Lg, O.Lg[1:K + 1]) and np.array_equal(P.rc, O.rc[1:K + 1])
    assert np.allclose(P.g, np.concatenate(O.g[1:K + 1]), rtol=1e-6)
    assert np.allclose(P.gdM, np.concatenate([O.gd[k] * O.M[k] for k in range(1, K + 1)]), rtol=1e-5)
    assert np.allclose(P.hpf, O.Hhpf.numpy(), atol=1e-6)


def test_rule_presets_differ_where_they_should():
    """The switches are live: the sliCQ-style length rule changes the edge bands only, the periodic sampling only odd windows."""
    from oracle.nsgt_cqt import OracleCQT
    a = OracleCQT(7, 64, "oct", ("kaiser", 1), 22050, 184184)
    b = OracleCQT(7, 64, "oct", ("kaiser", 1), 22050, 184184, rules=_presets()["nsgt_f_over_q"])
    c = OracleCQT(7, 64, "oct", ("kaiser", 1), 22050, 184184, rules=_presets()["nsgt_f_over_q_periodic"])
    K = 7 * 64
    assert np.array_equal(a.Lg[2:K], b.Lg[2:K]) and a.size_per_oct == b.size_per_oct == [32, 64, 128, 256, 512, 1024, 2048]
    assert b.Lg[K + 1] != a.Lg[K + 1]          # (the last constant-Q band happens to round to the same length here)
    odd = [k for k in range(K + 2) if b.Lg[k] % 2]
    even = [k for k in range(K + 2) if b.Lg[k] % 2 == 0]
    assert odd and all(not np.allclose(b.g[k], c.g[k]) for k in odd) and all(np.allclose(b.g[k], c.g[k]) for k in even)


def test_band_rule_that_breaks_octave_halving_is_refused():
    """band0_len='to_dc' (the textbook neighbours rule) makes the lowest band ~45x longer than its octave mates: a valid frame
    (the oracle builds it) whose lowest octave is 1024 long instead of 32 -- the U-Net needs octave lengths to double
    (unet...py:768-774,786) and the HIP plan refuses it instead of producing something silently different."""
    from audio_inpainting_diffusion_amd.cqt import CQTPlan, CQTRules
    from oracle.nsgt_cqt import OracleCQT
    O = OracleCQT(7, 64, "oct", ("kaiser", 1), 22050, 184184, rules=dict(band0_len="to_dc"))
    assert O.size_per_oct[:2] == [1024, 64]
    with pytest.raises(NotImplementedError):
        CQTPlan(7, 64, 22050, 184184, ("kaiser", 1.0), rules=CQTRules(band0_len="to_dc"))


def test_rules_can_be_selected_from_the_config():
    from audio_inpainting_diffusion_amd.config import small_args
    from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
    a = small_args()
    a.network.cqt.rules = "nsgt_f_over_q_periodic"
    net = Unet_CQT_oct_with_attention(a, torch.device("cpu"))
    assert net.CQTransform.plan.rules.window_sampling == "half_sample_odd"
    assert Unet_CQT_oct_with_attention(small_args(), torch.device("cpu")).CQTransform.plan.rules.window_sampling == "integer"


# ---- consuming a dump of the real package ----------------------------------------------------------------------------
def _input_of(z):
    no, bpo, fs, L = (int(v) if i != 2 else float(v) for i, v in enumerate(z["cfg"]))
    rng = np.random.Generator(np.random.PCG64(int(z["seed"])))
    x = torch.from_numpy((0.063 * rng.standard_normal((2, L))).astype(np.float32))
    shapes = [z[f"fwd_{o}"].shape for o in range(int(z["n_oct"]))]
    Cr = [torch.complex(torch.from_numpy(rng.standard_normal(s).astype(np.float32)), torch.from_numpy(rng.standard_normal(s).astype(np.float32)))
          for s in shapes]
    return (no, bpo, fs, L), x, Cr


def _matching_preset(z):
    """Name of the rule preset whose frame equals the dumped one (window lengths, first covered bins, window samples)."""
    from oracle.nsgt_cqt import OracleCQT
    (no, bpo, fs, L), _, _ = _input_of(z)
    K = no * bpo
    report = {}
    for name, rules in _presets().items():
        O = OracleCQT(no, bpo, "oct", ("kaiser", 1), fs, L, dtype=torch.float64, rules=rules)
        ok = True
        if "g_len" in z.files:
            ref_len = z["g_len"][:K + 2]
            ok &= np.array_equal(ref_len, O.Lg)
            if ok and "g_cat" in z.files:
                # the package stores windows in FFT order (centre at index 0); ours are centred
                off = np.concatenate(([0], np.cumsum(z["g_len"])))
                for k in range(K + 2):
                    w = np.fft.fftshift(z["g_cat"][off[k]:off[k + 1]]) if True else None
                    ok &= bool(np.allclose(np.roll(w, 0), O.g[k], atol=1e-6) or np.allclose(z["g_cat"][off[k]:off[k + 1]], O.g[k], atol=1e-6))
        if ok and "win_first_bin" in z.files:
            ok &= np.array_equal(z["win_first_bin"][:K + 2] % L, (O.rc - O.Lg // 2) % L)
        if ok and z["size_per_oct"].size:
            ok &= list(z["size_per_oct"]) == O.size_per_oct
        report[name] = bool(ok)
    return [n for n, ok in report.items() if ok], report


@pytest.mark.parametrize("path", FIXTURES or [None])
def test_oracle_reproduces_the_real_package(path):
    if path is None:
        pytest.skip(NO_FIXTURE)
    from oracle.nsgt_cqt import OracleCQT
    from audio_inpainting_diffusion_amd.cqt import RULES_DEFAULT, RULE_PRESETS
    z = np.load(path)
    # the version the reference author ran is on record (notebooks/demo_inpainting_spectrogram.ipynb cell 4 output: cqt-nsgt-pytorch-0.0.8.tar.gz):
    # a dump made with another version pins "some version", not "the version" -- refuse it
    assert "package_version" in z.files and str(z["package_version"]) == PINNED_CQT_VERSION, \
        f"{os.path.basename(path)} was dumped from cqt_nsgt_pytorch {z['package_version'] if 'package_version' in z.files else '?'}; " \
        f"the reference's notebook records {PINNED_CQT_VERSION}: pip install cqt_nsgt_pytorch=={PINNED_CQT_VERSION} and re-run tests/golden/make_cqt_golden.py"
    match, report = _matching_preset(z)
    assert match, f"no rule preset reproduces the frame of cqt_nsgt_pytorch for {os.path.basename(path)}: {report}; extend CQTRules"
    assert RULE_PRESETS[match[0]] == RULES_DEFAULT or any(RULE_PRESETS[m] == RULES_DEFAULT for m in match), \
        f"the package's frame is preset {match}, but cqt.RULES_DEFAULT is another one: set RULES_DEFAULT = RULE_PRESETS['{match[0]}']"
    (no, bpo, fs, L), x, Cr = _input_of(z)
    O = OracleCQT(no, bpo, "oct", ("kaiser", 1), fs, L, rules=_presets()[match[0]])
    C = O.fwd(x[:, None])
    for o, c in enumerate(C):
        assert rel_l2(torch.view_as_real(c), torch.view_as_real(torch.from_numpy(z[f"fwd_{o}"]))) < 1e-5
    assert rel_l2(O.bwd(Cr), z["bwd_random"]) < 1e-5
    assert rel_l2(O.apply_hpf_DC(x), z["hpf"]) < 1e-5
    assert rel_l2(O.bwd(C), z["roundtrip"]) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES or [None])
def test_hip_cqt_reproduces_the_real_package(path):
    if path is None:
        pytest.skip(NO_FIXTURE)
    from audio_inpainting_diffusion_amd.cqt import CQTransform
    z = np.load(path)
    match, report = _matching_preset(z)
    assert match, report
    (no, bpo, fs, L), x, Cr = _input_of(z)
    tr = CQTransform(no, bpo, "oct", ("kaiser", 1.0), fs, L, device="cuda", rules=match[0])
    C = tr.fwd(x[:, None].cuda())
    for o, c in enumerate(C):
        assert rel_l2(torch.view_as_real(c.cpu()), torch.view_as_real(torch.from_numpy(z[f"fwd_{o}"]))) < 1e-5
    assert rel_l2(tr.bwd([c.cuda() for c in Cr]).cpu(), z["bwd_random"]) < 1e-5
    assert rel_l2(tr.apply_hpf_DC(x.cuda()).cpu(), z["hpf"]) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("preset", ["nsgt_f_over_q", "nsgt_f_over_q_periodic", "nsgt_midpoint"])
@pytest.mark.parametrize("cfg", [(4, 8, 16000, 4096), (7, 64, 22050, 184184)])
def test_hip_cqt_under_alternative_rules_vs_oracle(preset, cfg):
    """The non-default rule sets run through the same HIP kernels (only the host tables change)."""
    from audio_inpainting_diffusion_amd.cqt import CQTransform
    from oracle.nsgt_cqt import OracleCQT
    no, bpo, fs, Ls = cfg
    orc = OracleCQT(no, bpo, "oct", ("kaiser", 1), fs, Ls, rules=_presets()[preset])
    try:
        tr = CQTransform(no, bpo, "oct", ("kaiser", 1.0), fs, Ls, device="cuda", rules=preset)
    except NotImplementedError as e:
        assert "midpoint" in preset, e
        pytest.skip(f"{preset}: {e}")
    x = torch.randn(2, Ls, generator=torch.Generator().manual_seed(60)) * 0.063
    for a, b in zip(tr.fwd(x[:, None].cuda()), orc.fwd(x[:, None])):
        assert rel_l2(torch.view_as_real(a.cpu()), torch.view_as_real(b)) < 1e-5
    hp = tr.apply_hpf_DC(x.cuda())
    assert rel_l2(hp.cpu(), orc.apply_hpf_DC(x)) < 1e-5
    assert rel_l2(tr.bwd(tr.fwd(x[:, None].cuda()))[:, 0].cpu(), hp.cpu()) < 1e-5
