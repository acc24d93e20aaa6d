"""CPU tests (no GPU): host-side logic of the product package and the C-ABI surface.

 * libaid_hip.so loads here and exports every symbol include/aid_kernels.h declares (no compute calls);
 * the product's vectorised CQT plan equals the oracle's band-by-band design, and a numpy emulation of the
   device algorithm (gather * window -> power-of-two inverse FFT; FFT * dual window -> overlap-add) built
   from the plan's tables reproduces the oracle transform;
 * drop-in surface: state_dict keys/shapes equal the reference module's (recorded in the golden fixtures),
   the plugin classes resolve through the dotted-string mechanism the reference uses;
 * EDM host schedule == golden; smooth-mask construction == oracle; the product fails loudly without a GPU.
"""
import ast
import importlib
import os
import re

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT, rel_l2


def test_c_abi_exports_every_declared_symbol():
    import ctypes
    hdr = open(os.path.join(ROOT, "include", "aid_kernels.h")).read()
    declared = set(re.findall(r"^\s*(?:int|int64_t|void|const char\*)\s+(aid_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 15
    so = os.path.join(ROOT, "audio_inpainting_diffusion_amd", "libaid_hip.so")
    if not os.path.exists(so):
        import build
        build.build()
    lib = ctypes.CDLL(so)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in aid_kernels.h but not exported"
    lib.aid_abi_version.restype = ctypes.c_int
    assert lib.aid_abi_version() == int(re.search(r"#define AID_ABI_VERSION (\d+)", hdr).group(1)) == 14
    from audio_inpainting_diffusion_amd import _lib
    assert set(_lib.EXPORTS) == declared
    a, b = _lib.pack_dims(2, 96)
    assert (a, b) == (32, 96)
    assert _lib.pack_dims(256, 256) == (256, 256) and _lib.pack_dims(40, 5120) == (64, 5120) and _lib.pack_dims(64, 192) == (64, 192)


@pytest.mark.parametrize("cfg", [(7, 64, 22050, 184184), (8, 64, 44100, 368368), (7, 8, 22050, 16384), (3, 8, 22050, 2048)])
def test_cqt_plan_matches_oracle_design(cfg):
    from audio_inpainting_diffusion_amd.cqt import CQTPlan
    from oracle.nsgt_cqt import OracleCQT
    no, bpo, fs, L = cfg
    P, O = CQTPlan(no, bpo, fs, L, ("kaiser", 1.0)), OracleCQT(no, bpo, "oct", ("kaiser", 1), fs, L)
    K = no * bpo
    assert P.T_oct == O.size_per_oct
    assert np.array_equal(P.Lg, O.Lg[1:K + 1]) and np.array_equal(P.rc, O.rc[1:K + 1])
    assert np.allclose(P.g, np.concatenate(O.g[1:K + 1]), rtol=1e-6)
    assert np.allclose(P.gdM, np.concatenate([O.gd[k] * O.M[k] for k in range(1, K + 1)]), rtol=1e-5)
    assert np.allclose(P.hpf, O.Hhpf.numpy(), atol=1e-6)


def _emulate_device_cqt(P, x):
    """numpy restatement of csrc/aid_cqt.hip driven by the plan tables (float64)."""
    B, L = x.shape
    X = np.fft.rfft(x, axis=-1)
    coefs, ws = [], np.zeros((B, P.ws_per_b), dtype=complex)
    for k in range(P.K):
        T, Lg, rc = int(P.Tk[k]), int(P.Lg[k]), int(P.rc[k])
        j = np.arange(Lg) - Lg // 2
        buf = np.zeros((B, T), dtype=complex)
        buf[:, (j + T) & (T - 1)] = X[:, rc + j] * P.g[P.goff[k]:P.goff[k] + Lg]
        c = np.fft.ifft(buf, axis=-1)
        coefs.append(c)
        ws[:, P.woff[k]:P.woff[k] + T] = np.fft.fft(c, axis=-1)
    Y = np.zeros((B, P.Lh), dtype=complex)
    for v in range(P.Lh):
        for k in range(P.kfirst[v], P.kfirst[v] + P.kcount[v]):
            jj = v - int(P.rc[k])
            idx = jj + int(P.Lg[k]) // 2
            if 0 <= idx < P.Lg[k]:
                T = int(P.Tk[k])
                Y[:, v] += ws[:, P.woff[k] + ((jj + T) & (T - 1))] * P.gdM[P.goff[k] + idx]
    return coefs, np.fft.irfft(Y, n=L, axis=-1)


def test_cqt_device_algorithm_emulation_matches_oracle():
    from audio_inpainting_diffusion_amd.cqt import CQTPlan
    from oracle.nsgt_cqt import OracleCQT
    no, bpo, fs, L = 3, 8, 22050, 2048
    P, O = CQTPlan(no, bpo, fs, L, ("kaiser", 1.0)), OracleCQT(no, bpo, "oct", ("kaiser", 1), fs, L, dtype=torch.float64)
    x = torch.randn(2, L, dtype=torch.float64, generator=torch.Generator().manual_seed(0))
    coefs, y = _emulate_device_cqt(P, x.numpy())
    ref = O.fwd(x[:, None])
    for o in range(no):
        got = np.stack(coefs[o * bpo:(o + 1) * bpo], axis=1)
        assert rel_l2(torch.view_as_real(torch.from_numpy(got)), torch.view_as_real(ref[o][:, 0])) < 1e-6
    assert rel_l2(y, O.apply_hpf_DC(x)) < 1e-6          # bwd(fwd(x)) == apply_hpf_DC(x)


def test_oracle_cqt_properties():
    """CQT parity is unpinned (no reference source): the oracle is held to the NSGT properties instead."""
    from oracle.nsgt_cqt import OracleCQT
    L, fs = 16384, 22050
    q = OracleCQT(7, 8, "oct", ("kaiser", 1), fs, L, dtype=torch.float64)
    g = torch.Generator().manual_seed(1)
    x, x2 = torch.randn(1, L, dtype=torch.float64, generator=g), torch.randn(1, L, dtype=torch.float64, generator=g)
    c = q.fwd(x[:, None])
    assert [ci.shape[-1] for ci in c] == q.size_per_oct and all(a * 2 == b for a, b in zip(q.size_per_oct, q.size_per_oct[1:]))
    h = q.apply_hpf_DC(x)
    assert rel_l2(q.bwd(c)[:, 0], h) < 1e-10                      # perfect reconstruction
    assert rel_l2(q.apply_hpf_DC(h), h) < 0.2 and rel_l2(q.apply_hpf_DC(h), h) >= 0   # smooth projector (not idempotent by design)
    c2 = q.fwd((2 * x + 3 * x2)[:, None])
    cx2 = q.fwd(x2[:, None])
    assert all(rel_l2(torch.view_as_real(a), torch.view_as_real(2 * b + 3 * d)) < 1e-10 for a, b, d in zip(c2, c, cx2))
    # pure tone lands in the expected octave / bin
    k, o = 3, 4
    f0 = (fs / 2) / 2 ** 7 * 2 ** ((o * 8 + k) / 8)
    tone = torch.sin(2 * np.pi * f0 * torch.arange(L, dtype=torch.float64) / fs)[None]
    e = torch.stack([ci.abs().pow(2).sum(-1)[0, 0] * (q.size_per_oct[-1] / ci.shape[-1]) for ci in q.fwd(tone[:, None])])
    assert int(e.flatten().argmax()) == o * 8 + k


def test_state_dict_surface_matches_reference():
    from audio_inpainting_diffusion_amd.config import make_args, small_args
    from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
    for tag in ("a", "b"):
        z = np.load(os.path.join(GOLDEN, f"unet_small_{tag}.npz"))
        kw = ast.literal_eval(str(z["cfg"]))
        net = Unet_CQT_oct_with_attention(small_args(**kw), torch.device("cpu"))
        sd = net.state_dict()
        assert list(sd.keys()) == list(z["keys"])
        assert [repr(tuple(v.shape)) for v in sd.values()] == list(z["shapes"])
    # the product refuses to compute without a GPU instead of falling back
    net = Unet_CQT_oct_with_attention(small_args(), torch.device("cpu"))
    from audio_inpainting_diffusion_amd._lib import AidError
    with pytest.raises(AidError):
        net(torch.zeros(1, 4096), torch.zeros(1, 1))


def test_plugin_strings_resolve_like_the_reference_does():
    """utils/dnnlib/util.py:235-273 resolves `callable` strings with importlib.import_module on dotted prefixes.
    Both spellings of the package (the importable one and the repository's hyphenated alias) must resolve to the
    SAME class objects -- one identity for isinstance / pickling, one loaded libaid_hip.so."""
    for tail in ("network.Unet_CQT_oct_with_attention", "sampler.Sampler", "edm.EDM"):
        objs = []
        for pkg in ("audio_inpainting_diffusion_amd", "audio-inpainting-diffusion_amd"):
            mod, _, attr = (pkg + "." + tail).rpartition(".")
            objs.append(getattr(importlib.import_module(mod), attr))
        assert callable(objs[0]) and objs[0] is objs[1]
        assert objs[0].__module__.startswith("audio_inpainting_diffusion_amd.")
    import audio_inpainting_diffusion_amd._lib as a
    b = importlib.import_module("audio-inpainting-diffusion_amd._lib")
    assert a is b


def test_edm_host_schedule_matches_golden():
    from audio_inpainting_diffusion_amd.config import make_args
    from audio_inpainting_diffusion_amd.edm import EDM
    from audio_inpainting_diffusion_amd.sampler import Sampler
    z = np.load(os.path.join(GOLDEN, "edm_schedule.npz"))
    args = make_args()
    edm = EDM(args)
    Sampler(model=None, diff_params=edm, args=args)          # applies tester.diff_params like the reference (:43-53)
    for T in (35, 36, 70, 128):
        t = edm.create_schedule(T)
        assert np.array_equal(t.numpy(), z[f"t{T}"]) and np.array_equal(edm.get_gamma(t).numpy(), z[f"gamma{T}"])
    s = torch.from_numpy(z["sigma"])
    for n in ("cskip", "cout", "cin", "cnoise"):
        assert np.array_equal(getattr(edm, n)(s).numpy(), z[n])


def test_smooth_mask_matches_oracle_and_reference_geometry():
    from audio_inpainting_diffusion_amd.sampler import prepare_smooth_mask
    from oracle.sampler import smooth_mask_rows
    L = 184184
    gap = int(300 * 22050 / 1000)
    start = L // 2 - gap // 2
    assert (gap, start) == (6615, 88785)                      # SURVEY.md section 8c closed form
    m = torch.ones(2, L)
    m[0, start:start + gap] = 0
    m[1, 1000:1400] = 0
    m[1, 90000:90800] = 0
    a, b = prepare_smooth_mask(m, 50), smooth_mask_rows(m, 50)
    assert torch.equal(a, b)
    assert float(a[0, start - 1]) < 1e-3 and float(a[0, start - 50]) == 1.0 and float(a[0, start + gap]) == 0.0


def test_masks_match_reference_prepare_mask_geometry():
    """Closed forms quoted in SURVEY.md section 8c: 300 ms @22050 -> 6615 samples from 88785; 1.5 s @44100 -> 66150;
    25/50/100 ms @16 kHz -> 400/800/1600 samples, 4 gaps."""
    from audio_inpainting_diffusion_amd.config import make_args
    from audio_inpainting_diffusion_amd.masks import long_gap_mask, mask_from_args, short_gaps_mask
    from audio_inpainting_diffusion_amd.sampler import prepare_smooth_mask
    L = 184184
    m = long_gap_mask(L, 22050, 300)
    z = (m[0] == 0).nonzero().flatten()
    assert (len(z), int(z[0])) == (6615, 88785)
    assert int((long_gap_mask(L, 44100, 1500)[0] == 0).sum()) == 66150
    for ms, n in ((25, 400), (50, 800), (100, 1600)):
        g = torch.Generator().manual_seed(ms)
        m = short_gaps_mask(L, 16000, ms, 4, g)
        assert (m == 0).sum() <= 4 * n and (m == 0).sum() >= n
        sm = prepare_smooth_mask(m, 100)
        assert float(sm.min()) == 0.0 and float(sm.max()) == 1.0 and torch.all(sm >= m * 0)   # cross-fades stay in [0,1]
    a = make_args("librispeech16k", T=70, gap_ms=50.0)
    assert a.tester.T == 70 and a.tester.data_consistency.hann_size == 100
    assert mask_from_args(a, torch.Generator().manual_seed(0)).shape == (1, L)
    assert torch.equal(mask_from_args(make_args("maestro22k", gap_ms=300.0)), long_gap_mask(L, 22050, 300))


def _emulate_stft_kernels(x, mask, L, n_fft, hop, win, adjoint):
    """numpy transcription of csrc/aid_stft.hip (frames kernel + overlap-add gather) in float64."""
    from audio_inpainting_diffusion_amd.stft import stft_tables
    w, env, _ = stft_tables(L, n_fft, hop, win)
    Lp, N, half = len(env), n_fft, n_fft // 2
    nfr = 1 + Lp // hop
    B = x.shape[0]
    fr = np.zeros((B, nfr, N))
    f = np.arange(N)
    fm = np.where(f <= half, f, N - f)
    for n in range(nfr):
        q = n * hop - half + np.arange(N)
        if adjoint:
            inside, qq = (q >= 0) & (q < Lp), np.clip(q, 0, Lp - 1)
        else:
            qq = np.where(q < 0, -q, q)
            qq = np.where(qq >= Lp, 2 * (Lp - 1) - qq, qq)
            inside = np.ones(N, bool)
        ok = inside & (qq < L)
        v = np.where(ok, x[:, np.minimum(qq, L - 1)], 0.0)
        if adjoint:
            v = v * np.where(ok, 1.0 / env[qq], 0.0)
        fr[:, n] = np.fft.ifft(np.fft.fft(v * w, axis=-1) * mask[fm, n], axis=-1).real * w

    def cover(pos):
        q = pos + half
        n1 = min(q // hop, nfr - 1)
        n0 = 0 if q - (N - 1) < 0 else -(-(q - (N - 1)) // hop)
        return sum((fr[:, n, q - n * hop] for n in range(n0, n1 + 1)), np.zeros(B))

    out = np.zeros((B, L))
    for t in range(L):
        v = cover(t)
        if adjoint:
            if 1 <= t <= half:
                v = v + cover(-t)
            r = 2 * (Lp - 1) - t
            if Lp <= r < Lp + half:
                v = v + cover(r)
        else:
            v = v / env[t]
        out[:, t] = v
    return out


@pytest.mark.parametrize("case", [(1000, 64, 16, 64), (1024, 64, 16, 64), (777, 128, 32, 128), (1500, 64, 16, 32)])
def test_stft_mask_device_algorithm_and_adjoint_match_oracle(case):
    """The frame/overlap-add decomposition the HIP kernels use (and its transposed border handling) equals
    torch.stft -> mask -> torch.istft and the gradient autograd takes through it."""
    from oracle.sampler import spectral_mask_apply
    L, n_fft, hop, win = case
    Lp = L + (n_fft - L % n_fft)
    rng = np.random.default_rng(0)
    mask = (rng.random((n_fft // 2 + 1, 1 + Lp // hop)) > 0.3).astype(np.float64)
    x, g = rng.standard_normal((2, L)), rng.standard_normal((2, L))
    xt = torch.from_numpy(x).requires_grad_()
    ref = spectral_mask_apply(xt, torch.from_numpy(mask), n_fft, hop, win)
    (ref * torch.from_numpy(g)).sum().backward()
    assert rel_l2(_emulate_stft_kernels(x, mask, L, n_fft, hop, win, 0), ref.detach()) < 1e-7
    assert rel_l2(_emulate_stft_kernels(g, mask, L, n_fft, hop, win, 1), xt.grad) < 1e-7


def test_spectral_mask_matches_reference_prepare_spectral_mask_geometry():
    """tester_inpainting.py:256-294 with conf/tester/inpainting_tester.yaml:78-87 at 22.05 kHz, L=184184."""
    from audio_inpainting_diffusion_amd.config import make_args
    from audio_inpainting_diffusion_amd.masks import spectral_mask_from_args
    A = spectral_mask_from_args(make_args("maestro22k"))
    assert tuple(A.shape) == (513, 721)
    zr, zc = np.nonzero((A == 0).any(1).numpy())[0], np.nonzero((A == 0).any(0).numpy())[0]
    assert (zr[0], zr[-1] + 1) == (14, 93)              # 300 Hz .. 2 kHz in 21.53 Hz bins
    assert (zc[0], zc[-1] + 1) == (273, 445)            # (92092 - 22050)//256, + 44100//256
    assert int((A == 0).sum()) == (93 - 14) * (445 - 273)


def test_checkpoint_loading_strategies_and_long_file_window():
    """harness.load_checkpoint follows utils/training_utils.py:214-289 (strict -> non-strict -> shape-matched);
    centre_gap_window reproduces tester_inpainting.py:399-411."""
    from audio_inpainting_diffusion_amd.config import small_args
    from audio_inpainting_diffusion_amd.harness import centre_gap_window, load_checkpoint
    from audio_inpainting_diffusion_amd.init import seeded_init_
    from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
    src = seeded_init_(Unet_CQT_oct_with_attention(small_args(), torch.device("cpu")), 1)
    sd = {k: v.clone() for k, v in src.state_dict().items()}
    dst = Unet_CQT_oct_with_attention(small_args(), torch.device("cpu"))
    from audio_inpainting_diffusion_amd.harness import UnpinnedCQTError
    with pytest.raises(UnpinnedCQTError, match="NOT pinned to cqt_nsgt_pytorch"):    # trained weights on an unpinned CQT: refused by default
        load_checkpoint(dst, {"it": 750000, "ema": sd}, cqt_pinned=False)
    with pytest.warns(RuntimeWarning, match="NOT pinned to cqt_nsgt_pytorch"):      # ... and loud when the caller opts out
        assert load_checkpoint(dst, {"it": 750000, "ema": sd}, cqt_pinned=False, allow_unpinned_cqt=True) == (750000, "strict")
    assert all(torch.equal(a, b) for a, b in zip(dst.state_dict().values(), sd.values()))
    k0 = next(k for k in sd if k.endswith("H.0.weight"))
    extra = dict(sd, **{"not.a.parameter": torch.zeros(3)})
    del extra[k0]
    dst = Unet_CQT_oct_with_attention(small_args(), torch.device("cpu"))
    assert load_checkpoint(dst, {"ema": extra}, allow_unpinned_cqt=True) == (0, "non-strict")
    bad = dict(sd)
    bad[k0] = torch.zeros(1, 2, 3)                                   # wrong shape: only the third strategy survives
    dst = Unet_CQT_oct_with_attention(small_args(), torch.device("cpu"))
    before = dst.state_dict()[k0].clone()
    assert load_checkpoint(dst, {"ema": bad}, allow_unpinned_cqt=True)[1] == "shape-matched"
    assert torch.equal(dst.state_dict()[k0], before)
    k1 = next(k for k in sd if k.endswith("H.1.weight"))
    assert torch.equal(dst.state_dict()[k1], sd[k1])
    assert load_checkpoint(dst, {"ema": {"nothing": torch.zeros(1)}}, allow_unpinned_cqt=True)[1] == "non-strict"      # (as the reference: attempt 2 accepts it)
    with pytest.raises(ValueError):
        load_checkpoint(dst, {"ema": {k0: torch.zeros(1, 2, 3)}}, allow_unpinned_cqt=True)                            # nothing matches by name AND shape
    # 6 s file at 22.05 kHz, 1.5 s gap, 184184-sample model window
    assert centre_gap_window(132300 * 2, 184184, 33075) == (132300 - 16537, 132300 - 92092)
    with pytest.raises(ValueError):
        centre_gap_window(1000, 184184, 10)


def test_resampling_kernel_tables_wav_io_and_case_logic(tmp_path):
    """harness._sinc_kernel == the oracle's restatement of torchaudio's default kernel; write_audio_file (utils/logging.py:295-319)
    round-trips through read_wav; oracle resample: tone in -> same tone out, lengths ceil(new*L/orig)."""
    from audio_inpainting_diffusion_amd.harness import _sinc_kernel, read_wav, write_audio_file
    from oracle.resample import resample, resample_batch, sinc_resample_kernel
    for o, n in ((2, 1), (320, 147), (160, 147), (441, 160), (147, 320)):
        k, w = sinc_resample_kernel(o, n)
        k2, w2 = _sinc_kernel(o, n)
        assert w == w2 and k.shape[-1] == 2 * w + o and float(np.abs(k[:, 0].numpy() - k2).max()) < 1e-15
    t = torch.arange(48000, dtype=torch.float64) / 48000
    x = torch.sin(2 * np.pi * 1000 * t)[None]
    y = resample(x, 320, 147)
    assert y.shape[-1] == int(np.ceil(147 * 48000 / 320))
    tt = torch.arange(y.shape[-1], dtype=torch.float64) * 320 / 147 / 48000
    assert float((y[0, 200:-200] - torch.sin(2 * np.pi * 1000 * tt)[200:-200]).abs().max()) < 2e-3
    assert resample_batch(torch.zeros(2, 100000), torch.tensor([44100, 44100]), 22050, 40000).shape == (2, 40000)
    assert resample_batch(torch.zeros(1, 100000), 48000, 44100, 50000).shape == (1, 50000)
    sig = 0.5 * torch.sin(2 * np.pi * 440 * torch.arange(2205) / 22050)
    fn = write_audio_file(sig.reshape(1, -1), 22050, "tone", path=str(tmp_path))
    back, sr = read_wav(fn)
    assert sr == 22050 and back.shape == (1, 2205) and float((back[0] - sig).abs().max()) < 1.0 / 32767
    loud = write_audio_file(4.0 * sig.reshape(1, -1), 22050, "loud", path=str(tmp_path))      # max >= 1 -> rescaled to peak 1
    assert abs(float(read_wav(loud)[0].max()) - 32767 / 32768) < 1e-4


def test_graft_entry_build_checks_the_header_abi_version():
    """__graft_entry__.build() is what the driver runs: it must compile, load the library and agree with the header's ABI version."""
    import __graft_entry__ as g
    g.build()


def test_header_compiles_as_c_and_struct_sizes_match_the_binding(tmp_path):
    """include/aid_kernels.h is the C ABI: it must compile as plain C99, and every parameter struct must have the size the ctypes mirror in
    _lib.py gives it (a drifted field list would shift every later field silently)."""
    import ctypes
    import shutil
    import subprocess
    from audio_inpainting_diffusion_amd import _lib
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    pairs = {"aid_view": "View", "aid_group_stats_params": "GroupStatsParams", "aid_conv2d_params": "Conv2dParams", "aid_resample_params": "ResampleParams",
             "aid_attention_params": "AttentionParams", "aid_embed_params": "EmbedParams", "aid_modulation_params": "ModulationParams",
             "aid_cqt_tables": "CqtTables", "aid_cqt_params": "CqtParams", "aid_cqt_gather_params": "CqtGatherParams", "aid_fft_pass_params": "FftPassParams",
             "aid_axpby_params": "AxpbyParams", "aid_score_step_params": "ScoreStepParams", "aid_group_dot_params": "GroupDotParams",
             "aid_norm_bwd_params": "NormBwdParams", "aid_attention_bwd_params": "AttentionBwdParams", "aid_guidance_seed_params": "GuidanceSeedParams",
             "aid_row_norm_params": "RowNormParams", "aid_guidance_step_params": "GuidanceStepParams", "aid_set_rows_params": "SetRowsParams", "aid_resample_poly_params": "ResamplePolyParams", "aid_stft_params": "StftParams",
             "aid_scale_act_params": "ScaleActParams", "aid_add2_params": "Add2Params", "aid_conv2d_wgrad_params": "WgradParams",
             "aid_wino_gy_params": "WinoGyParams", "aid_pack_conv_weight_params": "PackConvWeightParams", "aid_wgrad_reduce_params": "WgradReduceParams",
             "aid_channel_dot_params": "ChannelDotParams", "aid_relpos_bwd_params": "RelposBwdParams", "aid_scale_bwd_params": "ScaleBwdParams", "aid_modulation_bwd_params": "ModulationBwdParams",
             "aid_embed_bwd_params": "EmbedBwdParams", "aid_wino2d_gemm_params": "Wino2dGemmParams", "aid_adam_params": "AdamParams", "aid_ema_params": "EmaParams", "aid_sumsq_params": "SumsqParams"}
    hdr = open(os.path.join(ROOT, "include", "aid_kernels.h")).read()
    declared = set(re.findall(r"\}\s*(aid_\w+)\s*;", hdr))
    assert declared == set(pairs), declared ^ set(pairs)
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "aid_kernels.h"\nint main(void) {\n' +
                   "".join(f'    printf("{c} %zu\\n", sizeof({c}));\n' for c in pairs) + "    return 0;\n}\n")
    exe = tmp_path / "sizes"
    subprocess.check_call([gcc, "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)]).decode().splitlines())
    for c, py in pairs.items():
        assert int(out[c]) == ctypes.sizeof(getattr(_lib, py)), (c, out[c], ctypes.sizeof(getattr(_lib, py)))


def test_a_weighting_taps_match_the_reference_filter():
    """training.a_weighting_taps restates FIRFilter("aw") (utils/training_utils.py:94-120); golden from the imported reference (make_golden.py --only aweighting)."""
    from audio_inpainting_diffusion_amd.training import a_weighting_taps
    z = np.load(os.path.join(GOLDEN, "aweighting.npz"))
    for fs in (22050, 44100, 16000):
        t = a_weighting_taps(fs, 101)
        assert t.dtype == np.float32 and np.allclose(t, z[f"taps{fs}"], rtol=0, atol=1e-7)
        y = torch.nn.functional.conv1d(torch.from_numpy(z["x"]).unsqueeze(1), torch.from_numpy(t).view(1, 1, -1), padding=50).squeeze(1)
        assert rel_l2(y, z[f"y{fs}"]) < 1e-6
    with pytest.raises(ValueError):
        a_weighting_taps(22050, 100)


def test_edm_scalar_normalisation_of_the_fused_entry_points():
    """numpy.float32 / int / 0-d / 1-element values are host scalars, device vectors need B elements, anything else is refused
    before it can reach a kernel as a pointer."""
    from audio_inpainting_diffusion_amd._lib import AidError
    from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention as U
    dev = torch.device("cpu")
    out = U._norm_scalars(3, dev, (np.float32(0.5), 1, torch.tensor(0.25), torch.tensor([2.0])))
    assert out == (0.5, 1.0, 0.25, 2.0) and all(isinstance(v, float) for v in out)
    out = U._norm_scalars(2, dev, (torch.tensor([1.0, 2.0]), torch.tensor([[3.0], [4.0]]), torch.tensor([5.0, 6.0], dtype=torch.float64), 7.0))
    assert all(torch.is_tensor(v) and v.dtype == torch.float32 and v.shape == (2,) for v in out) and out[3].tolist() == [7.0, 7.0] and out[1].tolist() == [3.0, 4.0]
    with pytest.raises(AidError):
        U._norm_scalars(3, dev, (torch.zeros(2), 1.0, 1.0, 1.0))
    with pytest.raises(AidError):
        U._norm_scalars(3, dev, ("0.5", 1.0, 1.0, 1.0))


def test_cpu_thread_placements_and_numa_binding_are_well_formed():
    """bench.cpu_thread_configs (the placements the CPU baseline sweeps) and dist.bind_rank_to_gpu_numa (per-rank CPU shares): disjoint, non-empty
    shares of the cores this process may use; without a GPU / sysfs entry the split is even."""
    import bench
    from audio_inpainting_diffusion_amd import dist as D
    avail = sorted(os.sched_getaffinity(0))
    cfgs = bench.cpu_thread_configs()
    assert 1 <= len(cfgs) <= 3 and all(cpus and set(cpus) <= set(avail) for _, cpus in cfgs)
    assert len({tuple(c) for _, c in cfgs}) == len(cfgs) and len(cfgs[0][1]) <= len(cfgs[-1][1])
    assert D._cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    threads0 = torch.get_num_threads()
    try:
        shares = []
        for r in range(2):
            n = D.bind_rank_to_gpu_numa(r, 2)
            shares.append(sorted(os.sched_getaffinity(0)))
            os.sched_setaffinity(0, avail)
            assert n == len(shares[-1]) >= 1
        assert not (set(shares[0]) & set(shares[1])) or len(avail) == 1
        assert set(shares[0]) | set(shares[1]) <= set(avail)
    finally:
        os.sched_setaffinity(0, avail)
        torch.set_num_threads(threads0)


def test_f83_matrices_header_and_python_twin_agree():
    """csrc/aid_wino8.h (generated by tools/gen_wino8.py, compiled into the kernels) and _lib.wino8_matrices() (weight packs, tests) describe the same
    F(8,3): y = A^T [(G w) * (B^T d)] reproduces the 3-tap correlation of 10 samples exactly in fp64, the header's fp32 constants are the rounded
    matrices, and the +-a rows have the even / odd structure the kernels' transforms rely on."""
    from audio_inpainting_diffusion_amd import _lib
    AT, G, BT = _lib.wino8_matrices()
    rng = np.random.default_rng(3)
    for _ in range(4):
        d, w = rng.standard_normal(10), rng.standard_normal(3)
        y = AT @ ((G @ w) * (BT @ d))
        ref = np.array([sum(w[k] * d[o + k] for k in range(3)) for o in range(8)])
        assert np.abs(y - ref).max() < 1e-10
    sg = np.array([(-1.0) ** k for k in range(10)])
    for j in range(4):
        assert np.abs(BT[2 + 2 * j] - sg * BT[1 + 2 * j]).max() < 1e-12 and BT[1 + 2 * j][0] == 0 and BT[1 + 2 * j][9] == 0
    hdr = open(os.path.join(ROOT, "audio_inpainting_diffusion_amd", "csrc", "aid_wino8.h")).read()

    def macro(name):
        body = re.search(r"#define %s (\{.*\})" % name, hdr).group(1)
        return np.array(ast.literal_eval(body.replace("{", "[").replace("}", "]").replace("f,", ",").replace("f]", "]")), dtype=np.float64)
    assert np.abs(macro("AID_W8_G") - G).max() < 1e-12
    assert np.abs(macro("AID_W8_BT0") - BT[0][0::2]).max() < 1e-7
    assert np.abs(macro("AID_W8_BTE") - np.stack([BT[1 + 2 * j][2::2] for j in range(4)])).max() < 1e-7
    assert np.abs(macro("AID_W8_BTO") - np.stack([BT[1 + 2 * j][1:9:2] for j in range(4)])).max() < 1e-7
    at = macro("AID_W8_AT")
    assert np.abs(at - np.stack([AT[:, 1 + 2 * j] for j in range(4)])).max() < 1e-4 * np.abs(at).max() and at.shape == (4, 8)


def test_winograd_form_choice_is_a_function_of_the_launch_shape():
    """aid_conv2d_wino_form / aid_conv2d_wino8_supported (pure host functions of the library): a batch of one takes F(8,3) -- the K-group instances --
    on the levels whose launch is at most one 512-position tile per CU and keeps F(4,3) on the 96-channel levels (384 tiles) and the deepest one (112),
    sub-batches of four take F(8,3) wherever its 512-position tiles quantise, six rows per residue class have no F(8,3) tile, and the answer never
    depends on anything but the launch shape."""
    from audio_inpainting_diffusion_amd import _lib
    L = _lib.lib()
    levels = [(64, 64, 2048, 2), (96, 128, 1024, 3), (96, 192, 512, 4), (128, 256, 256, 5), (128, 320, 128, 6), (256, 384, 64, 7), (256, 448, 32, 7)]
    for C, F, T, nd in levels:
        for k in range(nd):
            f1 = L.aid_conv2d_wino_form(1, C, C, F, T, 1 << k)
            tiles8 = (C // 64) * F * T // 512
            assert f1 == (8 if (C != 96 and 128 < tiles8 <= 256 and L.aid_conv2d_wino8_supported(C, C, F, T, 1 << k)) else 4), (C, F, T, k, f1)
            f4 = L.aid_conv2d_wino_form(4, C, C, F, T, 1 << k)
            assert f4 in (4, 8) and f4 == L.aid_conv2d_wino_form(4, C, C, F, T, 1 << k)
            if f4 == 8:
                assert L.aid_conv2d_wino8_supported(C, C, F, T, 1 << k)
    assert L.aid_conv2d_wino_form(4, 256, 256, 384, 64, 1) == 8 and L.aid_conv2d_wino_form(4, 256, 256, 448, 32, 4) == 8 and L.aid_conv2d_wino_form(4, 64, 64, 64, 2048, 2) == 8
    assert L.aid_conv2d_wino_form(4, 128, 128, 320, 128, 1) == 4                      # 640 tiles of 512 positions = 2.5 per CU: F(4,3) quantises no worse
    assert not L.aid_conv2d_wino8_supported(256, 256, 384, 64, 64) and L.aid_conv2d_wino_form(8, 256, 256, 384, 64, 64) == 4     # 6 rows per class
    assert not L.aid_conv2d_wino8_supported(256, 256, 64, 16, 1) and L.aid_conv2d_wino_form(8, 256, 256, 64, 16, 1) == 4         # T = 16: F(4,3) tiles only
    assert L.aid_conv2d_wino_form(8, 2, 64, 64, 1024, 1) == 0 and L.aid_conv2d_wino_form(8, 64, 2, 64, 1024, 1) == 0              # few-channel layers: no Winograd input
    # fin_mode (the last tile of a sample folds the epilogue partials) is honoured by the row-shared kernels only
    assert L.aid_conv2d_fin_supported(4, 128, 128, 256, 256, 2, 2) == 1 and L.aid_conv2d_fin_supported(4, 128, 128, 256, 256, 2, 1) == 1
    assert L.aid_conv2d_fin_supported(2, 96, 96, 20, 256, 4, 1) == 0          # 5 rows per class: the row-shared kernel declines, the 96 x 512 tiles take the launch
    assert L.aid_conv2d_fin_supported(8, 256, 256, 384, 64, 64, 2) == 0       # no F(8,3) tile for 6 rows per class
    assert L.aid_conv2d_fin_supported(4, 128, 128, 256, 256, 2, 0) == 0 and L.aid_conv2d_fin_supported(4, 2, 64, 64, 1024, 1, 1) == 0


def test_two_dimensional_winograd_choice_is_a_function_of_the_launch_shape():
    """aid_conv2d_wino2d_supported / _wanted / _positions (pure host functions): K = 256 levels always (padding up to 4/3), K = 128 levels with T <= 256 up to
    1.2 padding (longer rows only for launches of at most two samples), nothing below 128 channels; positions = B * ceil(F / (4 dil)) * dil * T / 4."""
    from audio_inpainting_diffusion_amd import _lib
    L = _lib.lib()
    for B in (1, 2, 4, 8):
        for d in (1, 2, 4, 8, 16, 32, 64):
            assert L.aid_conv2d_wino2d_wanted(B, 256, 256, 448, 32, d) == 1 and L.aid_conv2d_wino2d_wanted(B, 256, 256, 384, 64, d) == 1
        for d in (1, 2, 4, 8, 16, 32):
            assert L.aid_conv2d_wino2d_wanted(B, 128, 128, 320, 128, d) == 1                 # (d = 32: 10 rows per class -> 12, padding 1.2)
        assert L.aid_conv2d_wino2d_wanted(B, 128, 128, 384, 64, 64) == 0                    # 6 rows per class -> 8: 1.33 does not pay at K = 128
        assert L.aid_conv2d_wino2d_wanted(B, 128, 128, 256, 256, 2) == 1 and L.aid_conv2d_wino2d_wanted(B, 128, 128, 128, 512, 2) == (1 if B <= 2 else 0)
        assert L.aid_conv2d_wino2d_wanted(B, 96, 96, 192, 512, 2) == 1 and L.aid_conv2d_wino2d_wanted(B, 96, 96, 64, 2048, 1) == 0 and L.aid_conv2d_wino2d_wanted(B, 64, 64, 64, 2048, 1) == 0
        assert L.aid_conv2d_wino2d_wanted(B, 128, 128, 320, 128, 4) == L.aid_conv2d_wino2d_wanted(B, 128, 128, 320, 128, 4)
        assert L.aid_conv2d_wino2d_positions(B, 448, 32, 64) == B * 2 * 64 * 8 and L.aid_conv2d_wino2d_positions(B, 384, 64, 1) == B * 96 * 16
    assert L.aid_conv2d_wino2d_supported(256, 256, 448, 32, 1) == 1 and L.aid_conv2d_wino2d_supported(128, 128, 256, 256, 16) == 1
    assert L.aid_conv2d_wino2d_supported(96, 96, 192, 512, 2) == 1 and L.aid_conv2d_wino2d_supported(64, 64, 64, 2048, 1) == 0 and L.aid_conv2d_wino2d_supported(256, 256, 448, 24, 1) == 0      # Cout % 128 or % 96, T % 16
    assert L.aid_conv2d_wino2d_supported(256, 256, 450, 32, 4) == 0 and L.aid_conv2d_wino2d_supported(128, 128, 64, 4096, 1) == 0    # F % dil, T <= 2048


def test_failed_launch_re_zeroes_the_scratch_that_must_be_found_zero():
    """Plan._fail: arrival counters / split-K flags have the contract "zero before a launch, left zero by it"; a launch that fails must not leave them
    half-way for the next run."""
    import ctypes as C
    from audio_inpainting_diffusion_amd import _lib
    from audio_inpainting_diffusion_amd.plan import Op, Plan
    pl = Plan()
    cnt = torch.ones(4, dtype=torch.int32)
    flags = torch.ones(4096, dtype=torch.float32)
    pl.zero_on_fail.extend([cnt, flags])
    holder = C.c_int(0)
    pl.ops.append(Op(lambda addr, stream: 2, C.addressof(holder), "aid_conv2d", 0, 0, "fake", 0, [], [], holder))
    pl.run = Plan.run.__get__(pl)
    with pytest.raises(_lib.AidError, match="aid_conv2d failed rc=2"):
        pl._fail(pl.ops[0], 2)
    assert int(cnt.abs().sum()) == 0 and float(flags[:1024].abs().sum()) == 0.0 and float(flags[1024:].sum()) == 3072.0
    # the flag region's size is the header's constant, not a literal of plan.py
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "aid_kernels.h")).read()
    assert int(re.search(r"#define AID_CONV2D_SPLIT_FLAG_BYTES (\d+)", hdr).group(1)) == _lib.AID_CONV2D_SPLIT_FLAG_BYTES == 4096
    # a run ABANDONED half-way for any other reason (an exception between launches) marks the plan dirty: the next run re-zeroes before its first launch
    pl2 = Plan()
    cnt2 = torch.zeros(4, dtype=torch.int32)
    pl2.zero_on_fail.append(cnt2)
    seen = []

    def first(addr, stream):
        cnt2.fill_(7)                                     # "a kernel left its counter half-way"
        return 0

    def boom(addr, stream):
        raise KeyboardInterrupt
    pl2.ops.append(Op(first, C.addressof(holder), "aid_add2", 0, 0, "fake", 0, [], [], holder))
    pl2.ops.append(Op(boom, C.addressof(holder), "aid_add2", 0, 0, "fake", 0, [], [], holder))
    import unittest.mock as mock
    with mock.patch("torch.cuda.current_stream", lambda: type("S", (), {"cuda_stream": 0})()), mock.patch("torch.cuda.synchronize", lambda: None):
        with pytest.raises(KeyboardInterrupt):
            pl2.run()
        assert pl2._dirty and int(cnt2.sum()) == 28
        pl2.ops[0] = Op(lambda addr, stream: seen.append(int(cnt2.sum())) or 0, C.addressof(holder), "aid_add2", 0, 0, "fake", 0, [], [], holder)
        pl2.ops.pop()
        pl2.run()
    assert seen == [0] and not pl2._dirty                 # zeroed BEFORE the first launch of the next run


def test_bench_reports_pmc_traffic_only_for_the_same_build_and_names_kernels_as_rocprof_does():
    """bench.py: roofline.traffic comes from a committed PMC profile ONLY when that profile was taken on the same kernel sources (no stale
    numbers); the per-kernel tables aggregate by the device kernel's name as rocprofv3 lists it; executed-FLOP fractions per Winograd form."""
    import bench
    assert bench.kernel_base("w2d_gemm_kernel<128x256,kc16,nb3>") == "w2d_gemm_kernel" and bench.kernel_base("conv53_wino8r_kernel(64+32)") == "conv53_wino8r_kernel"
    assert bench.kernel_base("conv11_dma_kernel+splitk") == "conv11_dma_kernel" and bench.kernel_base("w2d_gemm_s6_kernel<128x128,nb3,wpc3>") == "w2d_gemm_s6_kernel"
    assert bench.WINO_EXEC["w2d_gemm_kernel"] == 0.2 and abs(bench.WINO_EXEC["conv53_wino8r_kernel"] - 5 / 12) < 1e-12 and bench.WINO_EXEC["conv53_wino4r_kernel"] == 0.5
    f = bench._traffic_fields("conv53_wino8r_kernel", {"algorithmic_mb_per_launch": 500.0})
    tp = f["traffic_from_profile"]
    if tp is None:
        assert f["traffic"] is None and f["traffic_over_algorithmic"] is None
    else:
        same = tp.get("kernel_sources_sha16") == bench._kernel_sources_sha16()
        assert tp["used"] == same and (f["traffic"] is None or same) and (f["traffic_source"] is None) == (not same)
    assert len(bench._kernel_sources_sha16()) == 16


def test_lambda_degradation_is_the_jacobian_transpose_product_at_x_hat():
    """sampler.LambdaDegradation (Sampler.predict_resample, edm_sampler_inpainting.py:164-173): apply = the callable on a detached copy, adjoint = its VJP at the
    same point -- what the reference's torch.autograd.grad forms through `degradation(x_hat)` (:65-81) -- for a linear operator that changes the length and for a
    non-linear one; a second adjoint without an apply is refused, a constant degradation has no gradient."""
    from audio_inpainting_diffusion_amd._lib import AidError
    from audio_inpainting_diffusion_amd.sampler import LambdaDegradation
    from degradations import resample_degradations
    g0 = torch.Generator().manual_seed(0)
    x = torch.randn(2, 512, generator=g0)
    for which in (0, 1):
        fn = resample_degradations([0.1, 0.2, 0.4, 0.2, 0.1])[which]
        op = LambdaDegradation(fn)
        den = op.apply(x)
        assert not den.requires_grad and torch.equal(den, fn(x))
        seed = torch.randn(den.shape, generator=g0)
        got = op.adjoint(seed)
        xl = x.clone().requires_grad_()
        (ref,) = torch.autograd.grad(fn(xl), xl, seed)
        assert got.shape == x.shape and torch.allclose(got, ref, rtol=0, atol=1e-7)
        with pytest.raises(AidError):
            op.adjoint(seed)
    op = LambdaDegradation(lambda v: torch.zeros_like(v))
    op.apply(x)
    with pytest.raises(RuntimeError):
        op.adjoint(torch.ones(2, 512))
    with pytest.raises(AttributeError):
        op.project(x, x)


def test_sampler_and_edm_expose_the_reference_classes_method_names():
    """Every method of testing/edm_sampler_inpainting.Sampler (:8-364) and diff_params/edm.EDM (:7-193) exists under the same name with the same
    positional parameters (the names are the reference's API, listed here; the mirrors' behaviour is pinned by the golden / GPU tests)."""
    import inspect
    from audio_inpainting_diffusion_amd.edm import EDM
    from audio_inpainting_diffusion_amd.sampler import Sampler
    sampler_api = {"__init__": ["model", "diff_params", "args", "rid"], "update_diff_params": [], "get_score_rec_guidance": ["x", "y", "t_i", "degradation"],
                   "get_score": ["x", "y", "t_i", "degradation"], "predict_unconditional": ["shape", "device"], "predict_resample": ["y", "shape", "degradation"],
                   "predict": ["shape", "device"], "apply_mask": ["x", "mask"], "apply_spectral_mask": ["x"], "prepare_smooth_mask": ["mask", "size"],
                   "predict_inpainting": ["y_masked", "mask"], "predict_spectrogram_inpainting": ["y_masked", "mask"]}
    edm_api = {"__init__": ["args"], "get_gamma": ["t"], "create_schedule": ["nb_steps"], "sample_ptrain": ["N"], "sample_ptrain_safe": ["N"],
               "sample_prior": ["shape", "sigma"], "cskip": ["sigma"], "cout": ["sigma"], "cin": ["sigma"], "cnoise": ["sigma"], "lambda_w": ["sigma"],
               "denoiser": ["xn", "net", "sigma"], "prepare_train_preconditioning": ["x", "sigma"], "loss_fn": ["net", "x"]}
    for cls, api in ((Sampler, sampler_api), (EDM, edm_api)):
        for name, params in api.items():
            fn = getattr(cls, name, None)
            assert callable(fn), f"{cls.__name__}.{name} is missing"
            got = [p for p in inspect.signature(fn).parameters if p != "self"]
            assert got[:len(params)] == params, (cls.__name__, name, got)


def test_three_dimensional_observations_take_the_induced_matrix_norm_or_are_rejected():
    """torch.linalg.norm(y - den, dim=(1, 2), ord=2) over 3-D observations (edm_sampler_inpainting.py:67-75) is the SPECTRAL norm, not the
    Frobenius norm of the flattened item.  The oracle keeps the reference's semantics; the HIP sampler (whose guidance seed is element-wise)
    refuses such a call instead of silently computing another norm."""
    from audio_inpainting_diffusion_amd import _lib
    from audio_inpainting_diffusion_amd.config import small_args
    from audio_inpainting_diffusion_amd.edm import EDM
    from audio_inpainting_diffusion_amd.sampler import Sampler
    from oracle.edm import OracleEDM
    from oracle.sampler import OracleSampler
    from test_oracle_golden import _Toy
    L = 2048
    deg = lambda v: v.reshape(v.shape[0], 4, L // 4)[:, :3, ::2]                # noqa: E731  (a linear map onto [B, 3, L/8] observations)
    torch.manual_seed(3)
    y = deg(0.063 * torch.randn(1, L))
    runs = {}
    for norm in (2, "fro"):
        s = OracleSampler(_Toy(L), OracleEDM(), T=2, xi=0.25, data_consistency=False, audio_len=L, norm=norm)
        torch.manual_seed(4)
        runs[norm] = s.predict_resample(y, (1, L), deg)
    assert torch.isfinite(runs[2]).all() and rel_l2(runs[2], runs["fro"]) > 1e-3        # ord = 2 is NOT the flattened (Frobenius) norm
    args = small_args()
    args.tester.posterior_sampling.norm = 2

    class _Fused:                                                                   # (stands for the MI355X network: never evaluated here)
        denoise = denoise_guided = None
    smp = Sampler(model=_Fused(), diff_params=EDM(args), args=args)
    with pytest.raises(_lib.AidError, match="induced matrix norm"):
        smp.predict_resample(y, (1, L), deg)
