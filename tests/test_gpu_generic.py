"""The HIP sampler loop (churn, Heun combine, guidance step, projection: csrc/aid_sampler.hip) pinned DIRECTLY against the REFERENCE's own
trajectories (tests/golden/sampler_*.npz, produced by importing testing/edm_sampler_inpainting.Sampler around a toy denoiser): the toy
runs as a torch module on the GPU behind generic.GenericModelAdapter (explicitly; the Sampler has no implicit eager path), the DC/Nyquist
projector it owns is the HIP CQTransform.  Also: the norm = 1 / "smoothl1" guidance seeds of the HIP network (aid_guidance_seed) against
torch.autograd over the oracle."""
import ast
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


class _ToyGPU(torch.nn.Module):
    """Same toy denoiser as tests/golden/make_golden.py::_ToyNet (a test fixture), on the GPU, with the HIP CQT for apply_hpf_DC."""

    def __init__(self, L):
        super().__init__()
        from audio_inpainting_diffusion_amd.cqt import CQTransform
        self.CQTransform = CQTransform(3, 8, mode="oct", window=("kaiser", 1.0), fs=22050, audio_len=L, dtype=torch.float32, device=torch.device(DEV))
        self.k = torch.tensor([0.02, -0.05, 0.1, 0.25, 0.4, 0.25, 0.1, -0.05, 0.02], device=DEV).view(1, 1, 9)

    def forward(self, x, cnoise):
        y = torch.nn.functional.conv1d(x.unsqueeze(1), self.k, padding=4).squeeze(1)
        return y * torch.tanh(cnoise) + 0.1 * torch.sin(3.0 * x)


def _sampler(L, T, xi, **over):
    from audio_inpainting_diffusion_amd.config import make_args
    from audio_inpainting_diffusion_amd.edm import EDM
    from audio_inpainting_diffusion_amd.generic import GenericModelAdapter
    from audio_inpainting_diffusion_amd.sampler import Sampler
    args = make_args(audio_len=L, T=T, xi=xi)
    args.tester.data_consistency.hann_size = 20
    for k, v in over.items():
        node = args.tester
        *path, leaf = k.split(".")
        for q in path:
            node = getattr(node, q)
        setattr(node, leaf, v)
    edm = EDM(args)
    return Sampler(model=GenericModelAdapter(_ToyGPU(L), edm), diff_params=edm, args=args, rid=bool(over.get("_rid", False)))


def test_sampler_refuses_a_model_without_the_fused_entry_points():
    from audio_inpainting_diffusion_amd import _lib
    from audio_inpainting_diffusion_amd.config import make_args
    from audio_inpainting_diffusion_amd.edm import EDM
    from audio_inpainting_diffusion_amd.sampler import Sampler
    args = make_args(audio_len=2048, T=2, xi=0.0)
    smp = Sampler(model=_ToyGPU(2048), diff_params=EDM(args), args=args)
    with pytest.raises(_lib.AidError, match="GenericModelAdapter"):
        smp.predict_inpainting(torch.zeros(1, 2048, device=DEV), torch.ones(1, 2048, device=DEV))


@pytest.mark.parametrize("tag", ["g_s0", "g_s1", "g_s2_nosmooth", "r_s0", "r_b2_s1", "r_b2_nosmooth"])
def test_hip_sampler_loop_vs_reference_trajectories(tag):
    z = np.load(os.path.join(GOLDEN, "sampler_toy.npz"))
    L, T = int(z["L"]), int(z["T"])
    xi, B, smooth, seed = z[tag + ".meta"]
    smp = _sampler(L, T, float(xi), **{"data_consistency.smooth": bool(smooth)})
    torch.manual_seed(int(seed))
    out = smp.predict_inpainting(torch.from_numpy(z[tag + ".y"]).to(DEV), torch.from_numpy(z[tag + ".mask"]).to(DEV))
    e = rel_l2(out.cpu(), z[tag + ".out"])
    print(f"HIP sampler loop vs reference trajectory {tag}: {e:.2e}")
    assert e < 2e-5


@pytest.mark.parametrize("tag", ["g_end", "r_end", "g_always"])
def test_hip_sampler_data_consistency_types_vs_reference(tag):
    z = np.load(os.path.join(GOLDEN, "sampler_dc.npz"))
    L, T = int(z["L"]), int(z["T"])
    xi, seed, is_end = z[tag + ".meta"]
    smp = _sampler(L, T, float(xi), **{"data_consistency.type": "end" if is_end else "always"})
    torch.manual_seed(int(seed))
    out = smp.predict_inpainting(torch.from_numpy(z[tag + ".y"]).to(DEV), torch.from_numpy(z[tag + ".mask"]).to(DEV))
    assert rel_l2(out.cpu(), z[tag + ".out"]) < 2e-5


@pytest.mark.parametrize("tag", ["u_b1", "u_b2", "u_b2_nohpf"])
def test_hip_sampler_unconditional_vs_reference(tag):
    z = np.load(os.path.join(GOLDEN, "sampler_uncond.npz"))
    L, T = int(z["L"]), int(z["T"])
    B, hpf, seed = z[tag + ".meta"]
    smp = _sampler(L, T, 0.25, filter_out_cqt_DC_Nyq=bool(hpf))
    torch.manual_seed(int(seed))
    out = smp.predict_unconditional((int(B), L), torch.device(DEV))
    assert rel_l2(out.cpu(), z[tag + ".out"]) < 2e-5


@pytest.mark.parametrize("tag", ["l1", "sl1_small", "sl1_large"])
def test_hip_sampler_guidance_norm_variants_vs_reference(tag):
    """norm = 1 / "smoothl1" (edm_sampler_inpainting.py:72-75) through the adapter's torch.autograd seed and the HIP loop kernels."""
    z = np.load(os.path.join(GOLDEN, "sampler_norms.npz"))
    L, T = int(z["L"]), int(z["T"])
    kind, beta, seed = z[tag + ".meta"]
    smp = _sampler(L, T, 0.25, **{"posterior_sampling.norm": 1 if kind == 1 else "smoothl1", "posterior_sampling.smoothl1_beta": float(beta)})
    torch.manual_seed(int(seed))
    out = smp.predict_inpainting(torch.from_numpy(z[tag + ".y"]).to(DEV), torch.from_numpy(z[tag + ".mask"]).to(DEV))
    assert rel_l2(out.cpu(), z[tag + ".out"]) < 2e-5


@pytest.mark.parametrize("tag", ["lpf_s0", "lpf_s1_l1", "clip_s2", "clip_s3_sl1"])
def test_hip_sampler_predict_resample_vs_reference(tag):
    """predict_resample (edm_sampler_inpainting.py:164-173) -- guidance through a generic degradation lambda (low-pass + decimate by 2: observations
    half as long as the signal; a non-linear soft clipper) -- on the HIP loop kernels against the REFERENCE's own trajectories."""
    from degradations import resample_degradations
    z = np.load(os.path.join(GOLDEN, "sampler_resample.npz"))
    L, T = int(z["L"]), int(z["T"])
    B, kind, seed, which = z[tag + ".meta"]
    smp = _sampler(L, T, 0.25, **{"data_consistency.use": False, "posterior_sampling.norm": {2: 2, 1: 1, 3: "smoothl1"}[int(kind)],
                                  "posterior_sampling.smoothl1_beta": 0.02})
    torch.manual_seed(int(seed))
    out = smp.predict_resample(torch.from_numpy(z[tag + ".y"]).to(DEV), (int(B), L), resample_degradations(z["k"])[int(which)])
    e = rel_l2(out.cpu(), z[tag + ".out"])
    print(f"HIP sampler loop, predict_resample {tag}: {e:.2e} vs the reference trajectory")
    assert e < 2e-5


@pytest.mark.parametrize("which,norm", [(0, 2), (1, 2), (0, "smoothl1")])
def test_hip_network_predict_resample_vs_oracle(which, norm):
    """The same entry point on the HIP NETWORK: aid_guidance_seed on the degraded estimate (its own length), the lambda's torch VJP at x_hat, then the
    hand-written input-VJP of the whole denoiser -- against the oracle's autograd through network and lambda (B = 2, per-item seeds)."""
    import ast
    from audio_inpainting_diffusion_amd.config import small_args
    from audio_inpainting_diffusion_amd.edm import EDM
    from audio_inpainting_diffusion_amd.init import seeded_init_, seeded_normal
    from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
    from audio_inpainting_diffusion_amd.sampler import Sampler
    from degradations import resample_degradations
    from oracle.edm import OracleEDM
    from oracle.nsgt_cqt import OracleCQT
    from oracle.sampler import OracleSampler
    from oracle.unet import OracleUnet
    z = np.load(os.path.join(GOLDEN, "unet_small_a.npz"))
    kw = ast.literal_eval(str(z["cfg"]))
    args = small_args(**kw)
    args.tester.T, args.tester.posterior_sampling.xi = 3, 0.25
    args.tester.data_consistency.use = False
    args.tester.posterior_sampling.norm, args.tester.posterior_sampling.smoothl1_beta = norm, 0.02
    net = Unet_CQT_oct_with_attention(args, torch.device(DEV))
    seeded_init_(net, int(z["seed"]), gate_scale=10.0, affine_scale=10.0)
    Ls = kw["audio_len"]
    deg = resample_degradations(np.load(os.path.join(GOLDEN, "sampler_resample.npz"))["k"])[which]
    clean = torch.stack([torch.from_numpy(seeded_normal(41, g, Ls)) for g in range(2)]) * 0.063
    y = deg(clean)
    smp = Sampler(model=net, diff_params=EDM(args), args=args)
    smp.seeds = [7, 8]
    out = smp.predict_resample(y.to(DEV), (2, Ls), deg)
    cqt = OracleCQT(kw["num_octs"], kw["bins_per_oct"], "oct", ("kaiser", 1), kw["fs"], Ls)
    orc = OracleUnet(kw["num_octs"], kw["bins_per_oct"], cqt).load_state_dict(net.state_dict())
    osmp = OracleSampler(orc, OracleEDM(), T=3, xi=0.25, data_consistency=False, audio_len=Ls, norm=norm, smoothl1_beta=0.02)
    ref = osmp.predict_resample(y, (2, Ls), deg, seeds=[7, 8])
    e = rel_l2(out.cpu(), ref)
    print(f"HIP network, predict_resample (degradation {which}, norm {norm}, y {tuple(y.shape)}): rel-L2 vs oracle = {e:.2e}")
    assert e < 5e-4


def test_predict_resample_has_no_projection_like_the_reference():
    """data_consistency.use with predict_resample: the reference has no proj_convex_set there (AttributeError at the first evaluation)."""
    smp = _sampler(2048, 2, 0.25)
    with pytest.raises(AttributeError):
        smp.predict_resample(torch.zeros(1, 2048, device=DEV), (1, 2048), lambda x: x)


@pytest.mark.parametrize("mode", ["guided", "guided_rid", "replacement", "uncond", "lambda"])
def test_get_score_single_evaluation_vs_oracle(mode):
    """Sampler.get_score / get_score_rec_guidance (edm_sampler_inpainting.py:57-153) as stand-alone calls: one evaluation of the score on every branch,
    against the oracle's get_score (itself pinned by the reference trajectories); rid=True returns the reference's 5-tuple (:106-108)."""
    from oracle.edm import OracleEDM
    from oracle.sampler import OracleSampler, smooth_mask_rows
    from test_oracle_golden import _Toy
    L, t = 2048, 1.7
    xi = 0.0 if mode == "replacement" else 0.25
    smp = _sampler(L, 3, xi, _rid=(mode == "guided_rid"), **({"data_consistency.use": False} if mode == "lambda" else {}))
    g = torch.Generator().manual_seed(11)
    y, x = torch.randn(1, L, generator=g) * 0.063, torch.randn(1, L, generator=g) * 1.5
    mask = torch.ones(1, L)
    mask[:, 600:900] = 0
    osmp = OracleSampler(_Toy(L), OracleEDM(), T=3, xi=xi, hann_size=20, audio_len=L, data_consistency=(mode != "lambda"))
    if mode == "lambda":
        deg = lambda v: 0.05 * torch.tanh(v / 0.05)
        out = smp.get_score(x.to(DEV), deg(y).to(DEV), torch.tensor(t), deg)
        osmp.y, osmp.degradation = deg(y), deg
        ref = osmp.get_score(x, torch.tensor(t))
        assert smp.y is None and smp.spectral is None                       # the call installed its operator for its own duration only
    else:
        smp.setup_inpainting((y * mask).to(DEV), mask.to(DEV))
        osmp.y, osmp.mask, osmp.smask = (y * mask if mode != "uncond" else None), mask, smooth_mask_rows(mask, 20)
        if mode == "uncond":
            out = smp.get_score(x.to(DEV), None, torch.tensor(t), None)
        elif mode == "replacement":
            out = smp.get_score(x.to(DEV), smp.y, torch.tensor(t), smp.degradation)
        else:
            out = smp.get_score_rec_guidance(x.to(DEV), smp.y, torch.tensor(t), smp.degradation)
        ref = osmp.get_score(x, torch.tensor(t))
        assert smp.y is not None and smp.mask is not None                   # installed state untouched
    if mode == "guided_rid":
        assert isinstance(out, tuple) and len(out) == 5
        for name, a, b in zip(("score", "denoised", "s*grads", "after guidance", "after projection"), out, (ref,) + tuple(osmp._rid_last) + (osmp._rid_pocs,)):
            assert rel_l2(a.cpu(), b) < 2e-5, name
        out = out[0]
    e = rel_l2(out.cpu(), ref)
    print(f"get_score [{mode}]: rel-L2 vs oracle = {e:.2e}")
    assert e < 2e-5
    assert torch.equal(smp.apply_mask(x.to(DEV), mask.to(DEV)).cpu(), mask * x)
    assert torch.equal(smp.prepare_smooth_mask(mask, 20), smooth_mask_rows(mask, 20))


def test_hip_sampler_rid_buffers_vs_reference():
    z = np.load(os.path.join(GOLDEN, "sampler_rid.npz"))
    L, T = int(z["L"]), int(z["T"])
    smp = _sampler(L, T, 0.25, _rid=True)
    torch.manual_seed(3)
    res = smp.predict_inpainting(torch.from_numpy(z["y"]).to(DEV), torch.from_numpy(z["mask"]).to(DEV))
    assert len(res) == 8
    for name, r in zip(("out", "denoised", "grads", "grad_update", "pocs", "xt", "xt2", "t"), res):
        assert tuple(r.shape) == z[name].shape, name
        assert rel_l2(r.cpu(), z[name]) < 2e-5, name


@pytest.mark.parametrize("norm,beta", [(1, 1.0), ("smoothl1", 0.02), ("smoothl1", 0.5)])
def test_network_guided_evaluation_norm_variants_vs_oracle_autograd(norm, beta):
    """network.denoise_guided with the analytic L1 / smooth-L1 seeds (aid_guidance_seed norm_type 1 / 3) and the hand-written input-VJP vs
    torch.autograd over the CPU oracle; then three guided sampler steps vs the oracle sampler."""
    from audio_inpainting_diffusion_amd.config import small_args
    from audio_inpainting_diffusion_amd.edm import EDM
    from audio_inpainting_diffusion_amd.init import seeded_init_
    from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
    from audio_inpainting_diffusion_amd.sampler import Sampler
    from oracle.edm import OracleEDM
    from oracle.nsgt_cqt import OracleCQT
    from oracle.sampler import OracleSampler
    from oracle.unet import OracleUnet
    z = np.load(os.path.join(GOLDEN, "unet_small_a.npz"))
    kw = ast.literal_eval(str(z["cfg"]))
    args = small_args(**kw)
    Ls = kw["audio_len"]
    net = seeded_init_(Unet_CQT_oct_with_attention(args, torch.device(DEV)), int(z["seed"]), gate_scale=10.0, affine_scale=10.0)
    orc = OracleUnet(kw["num_octs"], kw["bins_per_oct"], OracleCQT(kw["num_octs"], kw["bins_per_oct"], "oct", ("kaiser", 1), kw["fs"], Ls)).load_state_dict(net.state_dict())
    x = torch.from_numpy(z["x"])
    B = x.shape[0]
    y = x * 0.126
    mask = torch.ones(1, Ls)
    mask[:, 1800:2300] = 0
    edm = OracleEDM()
    sig = torch.tensor(0.4)
    xr = x.clone().requires_grad_()
    xh = orc.CQTransform.apply_hpf_DC(edm.denoiser(xr, orc, sig.reshape(1, 1).expand(B, 1)))
    if norm == "smoothl1":
        nr = torch.nn.functional.smooth_l1_loss(y * mask, mask * xh, reduction="none", beta=beta).sum(dim=1)
    else:
        nr = torch.linalg.norm(y * mask - mask * xh, dim=1, ord=1)
    gref = torch.autograd.grad(nr.sum(), xr)[0]
    s1 = sig.reshape(1)
    sc = [float(f(s1)) for f in (edm.cnoise, edm.cin, edm.cskip, edm.cout)]
    xh_d, g_d, n_d = net.denoise_guided(x.to(DEV), *sc, True, (y * mask).to(DEV), mask.to(DEV), norm_type=norm, beta=beta)
    e = (rel_l2(xh_d.cpu(), xh.detach()), rel_l2(g_d.cpu(), gref), rel_l2(n_d.cpu(), nr.detach()))
    print(f"norm={norm} beta={beta}: x_hat {e[0]:.2e} gradient {e[1]:.2e} norm {e[2]:.2e}")
    assert e[0] < 1e-4 and e[1] < 2e-4 and e[2] < 1e-4
    args.tester.T, args.tester.posterior_sampling.xi = 3, 0.25
    args.tester.posterior_sampling.norm, args.tester.posterior_sampling.smoothl1_beta = norm, beta
    args.tester.data_consistency.hann_size = 20
    smp = Sampler(model=net, diff_params=EDM(args), args=args)
    smp.seeds, smp.trace = [5, 6], []
    smp.predict_inpainting((y * mask).to(DEV), mask.to(DEV))
    osmp = OracleSampler(orc, OracleEDM(), T=3, xi=0.25, hann_size=20, audio_len=Ls, norm=norm, smoothl1_beta=beta)
    osmp.predict_inpainting(y * mask, mask, seeds=[5, 6], record=True)
    errs = [rel_l2(a.cpu(), b) for a, b in zip(smp.trace, osmp.trace)]
    print("  per-evaluation x_hat rel-L2:", ["%.2e" % v for v in errs])
    assert errs[0] < 1e-4 and max(errs) < 1e-3
