"""CPU: the oracle restatement (oracle/) against the golden vectors captured from the reference itself
(tests/golden/make_golden.py).  Tolerance: 2e-6 rel-L2 (fp32 op-order noise between two torch-CPU
formulations of the same arithmetic)."""
import ast
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2
from oracle import unet as OU
from oracle.edm import OracleEDM
from oracle.nsgt_cqt import OracleCQT
from oracle.sampler import OracleSampler

TOL = 2e-6


@pytest.fixture(scope="module")
def ops():
    return np.load(os.path.join(GOLDEN, "ops_small.npz"))


def _sd(z, prefix):
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}


def test_group_norm(ops):
    y = OU.group_std_norm(torch.from_numpy(ops["gn.x"]), torch.from_numpy(ops["gn.gamma"]))
    assert rel_l2(y, ops["gn.y"]) < TOL


def test_resamplers(ops):
    x = torch.from_numpy(ops["rs.x"])
    assert rel_l2(OU.resample_down(x), ops["rs.down"]) < TOL
    assert rel_l2(OU.resample_up(x), ops["rs.up"]) < TOL


def test_embedding(ops):
    sd = _sd(ops, "emb.sd.")
    assert rel_l2(OU.embed(sd, torch.from_numpy(ops["emb.sigma"])), ops["emb.y"]) < TOL


def test_time_attention(ops):
    sd = _sd(ops, "ta.sd.")
    y = OU.time_attention(sd, "", torch.from_numpy(ops["ta.x"]), 8)
    assert rel_l2(y, ops["ta.y"]) < TOL


@pytest.mark.parametrize("tag", ["rb_plain", "rb_attn", "rb_out", "rb_init", "rb_dec"])
def test_resnet_block(ops, tag):
    sd = _sd(ops, tag + ".sd.")
    y = OU.resnet_block(sd, "", torch.from_numpy(ops[tag + ".x"]), torch.from_numpy(ops[tag + ".emb"]), 8)
    assert rel_l2(y, ops[tag + ".y"]) < TOL


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_unet_small(tag):
    from audio_inpainting_diffusion_amd.init import seeded_state_dict
    z = np.load(os.path.join(GOLDEN, f"unet_small_{tag}.npz"))
    kw = ast.literal_eval(str(z["cfg"]))
    shapes = [(k, ast.literal_eval(s)) for k, s in zip(z["keys"], z["shapes"])]
    sd = seeded_state_dict(shapes, int(z["seed"]), gate_scale=10.0, affine_scale=10.0)
    cqt = OracleCQT(kw["num_octs"], kw["bins_per_oct"], "oct", ("kaiser", 1), kw["fs"], kw["audio_len"])
    net = OU.OracleUnet(kw["num_octs"], kw["bins_per_oct"], cqt).load_state_dict(sd)
    with torch.no_grad():
        y = net(torch.from_numpy(z["x"]), torch.from_numpy(z["cnoise"]))
    assert rel_l2(y, z["y"]) < 5e-6
    # the fixture is sensitive to every branch: dropping the attention output must move the result
    sd2 = dict(sd)
    for k in sd2:
        if k.endswith("attn_block.proj_out.weight"):
            sd2[k] = torch.zeros_like(sd2[k])
    with torch.no_grad():
        y2 = OU.OracleUnet(kw["num_octs"], kw["bins_per_oct"], cqt).load_state_dict(sd2)(
            torch.from_numpy(z["x"]), torch.from_numpy(z["cnoise"]))
    assert rel_l2(y2, z["y"]) > 3e-3


def test_edm_schedule():
    z = np.load(os.path.join(GOLDEN, "edm_schedule.npz"))
    e = OracleEDM()
    for T in (35, 36, 70, 128):
        t = e.create_schedule(T)
        assert np.array_equal(t.numpy(), z[f"t{T}"])
        assert np.array_equal(e.get_gamma(t).numpy(), z[f"gamma{T}"])
    s = torch.from_numpy(z["sigma"])
    for n in ("cskip", "cout", "cin", "cnoise"):
        assert np.array_equal(getattr(e, n)(s).numpy(), z[n])
    # closed-form anchors quoted in SURVEY.md section 8c
    assert abs(float(z["t35"][1]) - 0.82238) < 1e-5 and abs(float(z["gamma36"][0]) - 0.27027) < 1e-5


class _Toy(torch.nn.Module):
    """Same toy denoiser as tests/golden/make_golden.py::_ToyNet (a test fixture, not reference code)."""

    def __init__(self, L):
        super().__init__()
        self.CQTransform = OracleCQT(3, 8, "oct", ("kaiser", 1), 22050, L)
        self.k = torch.tensor([0.02, -0.05, 0.1, 0.25, 0.4, 0.25, 0.1, -0.05, 0.02]).view(1, 1, 9)

    def forward(self, x, cnoise):
        y = torch.nn.functional.conv1d(x.unsqueeze(1), self.k, padding=4).squeeze(1)
        return y * torch.tanh(cnoise) + 0.1 * torch.sin(3.0 * x)


@pytest.mark.parametrize("tag", ["g_s0", "g_s1", "g_s2_nosmooth", "r_s0", "r_b2_s1", "r_b2_nosmooth"])
def test_sampler_trajectories(tag):
    z = np.load(os.path.join(GOLDEN, "sampler_toy.npz"))
    L, T = int(z["L"]), int(z["T"])
    xi, B, smooth, seed = z[tag + ".meta"]
    net = _Toy(L)
    s = OracleSampler(net, OracleEDM(), T=T, xi=float(xi), smooth=bool(smooth), hann_size=20, audio_len=L)
    torch.manual_seed(int(seed))
    out = s.predict_inpainting(torch.from_numpy(z[tag + ".y"]), torch.from_numpy(z[tag + ".mask"]))
    assert rel_l2(out, z[tag + ".out"]) < 1e-5


@pytest.mark.parametrize("tag", ["u_b1", "u_b2", "u_b2_nohpf"])
def test_unconditional_sampling_matches_reference(tag):
    """predict_unconditional (edm_sampler_inpainting.py:155-162): no observations, no projection, optional DC/Nyquist projector (:121-122)."""
    z = np.load(os.path.join(GOLDEN, "sampler_uncond.npz"))
    L, T = int(z["L"]), int(z["T"])
    B, hpf, seed = z[tag + ".meta"]
    s = OracleSampler(_Toy(L), OracleEDM(), T=T, xi=0.25, filter_out_cqt_DC_Nyq=bool(hpf), audio_len=L)
    torch.manual_seed(int(seed))
    out = s.predict_unconditional((int(B), L))
    assert rel_l2(out, z[tag + ".out"]) < 1e-5


@pytest.mark.parametrize("tag", ["g_end", "r_end", "g_always"])
def test_sampler_data_consistency_types_match_reference(tag):
    """data_consistency.type 'end' vs 'always' (edm_sampler_inpainting.py:22-24): the guided branch projects per
    evaluation only with 'always' (:100), the replacement branch ALWAYS (:141-147), 'end' projects after the loop (:252)."""
    z = np.load(os.path.join(GOLDEN, "sampler_dc.npz"))
    L, T = int(z["L"]), int(z["T"])
    xi, seed, is_end = z[tag + ".meta"]
    s = OracleSampler(_Toy(L), OracleEDM(), T=T, xi=float(xi), hann_size=20, audio_len=L, dc_type="end" if is_end else "always")
    torch.manual_seed(int(seed))
    out = s.predict_inpainting(torch.from_numpy(z[tag + ".y"]), torch.from_numpy(z[tag + ".mask"]))
    assert rel_l2(out, z[tag + ".out"]) < 1e-5


@pytest.mark.parametrize("tag", ["l1", "sl1_small", "sl1_large"])
def test_sampler_guidance_norm_variants_match_reference(tag):
    """tester.posterior_sampling.norm = 1 and "smoothl1" (edm_sampler_inpainting.py:72-75)."""
    z = np.load(os.path.join(GOLDEN, "sampler_norms.npz"))
    L, T = int(z["L"]), int(z["T"])
    kind, beta, seed = z[tag + ".meta"]
    s = OracleSampler(_Toy(L), OracleEDM(), T=T, xi=0.25, hann_size=20, audio_len=L, norm=1 if kind == 1 else "smoothl1", smoothl1_beta=float(beta))
    torch.manual_seed(int(seed))
    out = s.predict_inpainting(torch.from_numpy(z[tag + ".y"]), torch.from_numpy(z[tag + ".mask"]))
    assert rel_l2(out, z[tag + ".out"]) < 1e-5


@pytest.mark.parametrize("tag", ["lpf_s0", "lpf_s1_l1", "clip_s2", "clip_s3_sl1"])
def test_predict_resample_generic_degradation_matches_reference(tag):
    """predict_resample (edm_sampler_inpainting.py:164-173): guidance through a generic degradation lambda -- a low-pass + decimate-by-2
    (observations half as long as the signal) and a non-linear soft clipper -- against the reference's own trajectories."""
    from degradations import resample_degradations
    z = np.load(os.path.join(GOLDEN, "sampler_resample.npz"))
    L, T = int(z["L"]), int(z["T"])
    B, kind, seed, which = z[tag + ".meta"]
    s = OracleSampler(_Toy(L), OracleEDM(), T=T, xi=0.25, data_consistency=False, audio_len=L, norm={2: 2, 1: 1, 3: "smoothl1"}[int(kind)], smoothl1_beta=0.02)
    torch.manual_seed(int(seed))
    out = s.predict_resample(torch.from_numpy(z[tag + ".y"]), (int(B), L), resample_degradations(z["k"])[int(which)])
    assert rel_l2(out, z[tag + ".out"]) < 1e-5


def test_predict_resample_has_no_projection_like_the_reference():
    L = 2048
    s = OracleSampler(_Toy(L), OracleEDM(), T=2, xi=0.25, data_consistency=True, audio_len=L)
    with pytest.raises(AttributeError):
        s.predict_resample(torch.zeros(1, L), (1, L), lambda x: x)


def test_replacement_branch_without_projection_raises_like_the_reference():
    L = 2048
    s = OracleSampler(_Toy(L), OracleEDM(), T=2, xi=0.0, data_consistency=False, audio_len=L)
    with pytest.raises(AttributeError):
        s.predict_inpainting(torch.zeros(1, L), torch.ones(1, L))


def test_sampler_batch_items_are_independent():
    """Per-item semantics: a B=2 run with per-item seeds equals two B=1 runs (guided branch included)."""
    L, T = 2048, 4
    net = _Toy(L)
    y = torch.randn(2, L, generator=torch.Generator().manual_seed(3)) * 0.063
    mask = torch.ones(2, L)
    mask[0, 700:900] = 0
    mask[1, 1200:1300] = 0
    s = OracleSampler(net, OracleEDM(), T=T, xi=0.25, hann_size=20, audio_len=L)
    both = s.predict_inpainting(y * mask, mask, seeds=[11, 12])
    for b in range(2):
        one = s.predict_inpainting((y * mask)[b:b + 1], mask[b:b + 1], seeds=[11 + b])
        assert rel_l2(both[b:b + 1], one) < 1e-5


@pytest.mark.parametrize("tag", ["sg_s0", "sg_s1", "sr_s0"])
def test_spectrogram_inpainting_operator_and_trajectories(tag):
    """apply_spectral_mask (edm_sampler_inpainting.py:271-290) and predict_spectrogram_inpainting (:348-364)."""
    from oracle.sampler import spectral_mask_apply
    z = np.load(os.path.join(GOLDEN, "sampler_spectral.npz"))
    L, T = int(z["L"]), int(z["T"])
    stft = tuple(int(v) for v in z["stft"])
    xi, seed = z[tag + ".meta"]
    mask = torch.from_numpy(z["mask"])
    ym = spectral_mask_apply(torch.from_numpy(z[tag + ".y0"]), mask, *stft)
    assert rel_l2(ym, z[tag + ".y"]) < 1e-6
    s = OracleSampler(_Toy(L), OracleEDM(), T=T, xi=float(xi), audio_len=L)
    torch.manual_seed(int(seed))
    out = s.predict_spectrogram_inpainting(torch.from_numpy(z[tag + ".y"]), mask, stft=stft)
    assert rel_l2(out, z[tag + ".out"]) < 1e-5


def test_sampler_rid_debug_buffers():
    """rid=True (edm_sampler_inpainting.py:185-191, :217-226, :255-260): the 8-tuple of per-step buffers."""
    z = np.load(os.path.join(GOLDEN, "sampler_rid.npz"))
    L, T = int(z["L"]), int(z["T"])
    s = OracleSampler(_Toy(L), OracleEDM(), T=T, xi=0.25, hann_size=20, audio_len=L)
    torch.manual_seed(3)
    res = s.predict_inpainting(torch.from_numpy(z["y"]), torch.from_numpy(z["mask"]), rid=True)
    assert len(res) == 8
    for name, r in zip(("out", "denoised", "grads", "grad_update", "pocs", "xt", "xt2", "t"), res):
        assert tuple(r.shape) == z[name].shape, name
        assert rel_l2(r, z[name]) < 1e-5, name


def test_training_iterations_vs_reference_trainer():
    """tests/golden/train_small.npz holds three iterations of the REFERENCE's Trainer.train_step + update_ema (training/trainer.py:253-304, its
    EDM.loss_fn and setup_optimizer's Adam).  The same loop over the oracle network with our EDM restatement (sample_ptrain_safe, loss_fn) and
    the reference's update rules restated here: sigma bit-identical, losses, parameters and EMA to fp32 op-order noise."""
    from audio_inpainting_diffusion_amd.config import small_args
    from audio_inpainting_diffusion_amd.edm import EDM
    from audio_inpainting_diffusion_amd.init import seeded_init_, seeded_normal
    z = np.load(os.path.join(GOLDEN, "train_small.npz"))
    kw, hp = ast.literal_eval(str(z["cfg"])), ast.literal_eval(str(z["hp"]))
    args = small_args(**kw)
    edm = EDM(args)
    cqt = OracleCQT(kw["num_octs"], kw["bins_per_oct"], "oct", ("kaiser", 1), kw["fs"], kw["audio_len"])
    orc = OU.OracleUnet(kw["num_octs"], kw["bins_per_oct"], cqt)
    proto = _reference_shaped_state_dict(args, int(z["seed"]))
    orc.load_state_dict(proto)
    keys = list(orc.sd.keys())
    params = [torch.nn.Parameter(orc.sd[k].clone()) for k in keys]
    names = set(str(n) for n in z["names"])
    trainable = [p for k, p in zip(keys, params) if k in names]
    opt = torch.optim.Adam(trainable, lr=hp["lr"], betas=(0.9, 0.999), eps=1e-8)
    ema = {k: orc.sd[k].clone() for k in keys}
    B, L = hp["batch"], kw["audio_len"]
    for it in range(int(z["n_it"])):
        audio = torch.from_numpy(seeded_normal(41, it, B * L)).reshape(B, L) * 0.063
        orc.sd = {k: p for k, p in zip(keys, params)}
        torch.manual_seed(500 + it)
        opt.zero_grad()
        err2, sigma = edm.loss_fn(orc, audio)
        assert np.array_equal(sigma.reshape(-1).numpy(), z[f"sigma.{it}"])
        loss = err2.mean()
        assert abs(float(loss.detach()) - float(z["loss"][it])) < 2e-6 * abs(float(z["loss"][it]))
        loss.backward()
        if it <= hp["lr_rampup_it"]:
            for g in opt.param_groups:
                g["lr"] = hp["lr"] * min(it / max(hp["lr_rampup_it"], 1e-8), 1)
        assert abs(opt.param_groups[0]["lr"] - float(z[f"lr.{it}"])) < 1e-12
        torch.nn.utils.clip_grad_norm_(trainable, hp["max_grad_norm"])
        opt.step()
        t = it * hp["batch"]
        s = float(np.clip(t / hp["ema_rampup"], 0.0, hp["ema_rate"])) if t < hp["ema_rampup"] else hp["ema_rate"]
        with torch.no_grad():
            for k, p in zip(keys, params):
                if k in names:
                    ema[k].copy_(ema[k] * s + p * (1 - s))
    from conftest import projected_rel_error
    ep = projected_rel_error(z, "p", {k: p for k, p in zip(keys, params)})
    ee = projected_rel_error(z, "ema", ema)
    print(f"oracle loop vs reference trainer: parameters {ep:.2e}, EMA {ee:.2e}")
    assert ep < 2e-5 and ee < 2e-5


def _reference_shaped_state_dict(args, seed):
    """the seeded weights every U-Net fixture uses (make_golden.py::_seed_module): built on the parameter container of our module on the CPU"""
    from audio_inpainting_diffusion_amd.init import seeded_init_
    from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
    net = Unet_CQT_oct_with_attention(args, torch.device("cpu"))
    return seeded_init_(net, seed, gate_scale=10.0, affine_scale=10.0).state_dict()


@pytest.mark.parametrize("tag", ["a", "c"])
def test_guided_chain_matches_reference_sampler_and_unet(tag):
    """sampler_guided_unet.npz: the REFERENCE's Sampler + EDM driving the REFERENCE's own U-Net (make_golden.py --only guided).  The oracle chain
    (denoiser -> apply_hpf_DC -> masked 2-norm -> autograd) reproduces x_hat, rec_grads and norm of EVERY evaluation from the recorded input
    state (teacher-forced), and the oracle sampler reproduces the whole trajectory from the same global-generator seed."""
    from audio_inpainting_diffusion_amd.init import seeded_state_dict
    from oracle.edm import OracleEDM
    from oracle.sampler import OracleSampler
    z = np.load(os.path.join(GOLDEN, "sampler_guided_unet.npz"))
    zu = np.load(os.path.join(GOLDEN, f"unet_small_{tag}.npz"))
    kw = ast.literal_eval(str(z[f"{tag}.cfg"]))
    assert kw == ast.literal_eval(str(zu["cfg"]))
    shapes = [(k, ast.literal_eval(s)) for k, s in zip(zu["keys"], zu["shapes"])]
    sd = seeded_state_dict(shapes, int(z[f"{tag}.seed"]), gate_scale=10.0, affine_scale=10.0)
    cqt = OracleCQT(kw["num_octs"], kw["bins_per_oct"], "oct", ("kaiser", 1), kw["fs"], kw["audio_len"])
    net = OU.OracleUnet(kw["num_octs"], kw["bins_per_oct"], cqt).load_state_dict(sd)
    edm = OracleEDM()
    for seed in (0, 1):
        k = f"{tag}.s{seed}"
        y, mask = torch.from_numpy(z[k + ".y"]), torch.from_numpy(z[k + ".mask"])
        n = int(z[k + ".n_eval"])
        assert n == 5
        for e in range(n):
            x = torch.from_numpy(z[f"{k}.e{e}.x"]).clone().requires_grad_()
            sig = torch.full((1, 1), float(z[f"{k}.e{e}.t"]))
            xh = cqt.apply_hpf_DC(edm.denoiser(x, net, sig))
            norm = torch.linalg.norm(y - mask * xh, dim=1, ord=2)
            g = torch.autograd.grad(norm.sum(), x)[0]
            assert rel_l2(xh.detach(), z[f"{k}.e{e}.x_hat"]) < 2e-5
            assert rel_l2(g, z[f"{k}.e{e}.rec_grads"]) < 1e-4
            assert abs(float(norm) - float(z[f"{k}.e{e}.norm"][0])) < 2e-5 * float(norm)
        smp = OracleSampler(net, edm, T=3, xi=0.25, hann_size=20, audio_len=kw["audio_len"])
        torch.manual_seed(seed)
        out = smp.predict_inpainting(y, mask)
        assert rel_l2(out, z[k + ".out"]) < 2e-3          # (five chained evaluations of an O(1)-gate random network amplify rounding)


def test_guided_whole_trajectory_matches_reference():
    """sampler_guided_traj.npz: a WHOLE T = 10 trajectory of the reference's Sampler + EDM + U-Net (make_golden.py --only guided_traj) with churn
    inside a window only -- deterministic head, stochastic middle, deterministic tail, Heun steps and the final Euler step onto t = 0
    (edm_sampler_inpainting.py:204, :236-251).  The oracle reproduces the schedule / gamma bit for bit, every evaluation teacher-forced, and the
    free-running trajectory state by state."""
    from audio_inpainting_diffusion_amd.init import seeded_state_dict
    from oracle.edm import OracleEDM
    from oracle.sampler import OracleSampler
    z = np.load(os.path.join(GOLDEN, "sampler_guided_traj.npz"))
    zu = np.load(os.path.join(GOLDEN, "unet_small_a.npz"))
    kw = ast.literal_eval(str(z["cfg"]))
    assert kw == ast.literal_eval(str(zu["cfg"]))
    shapes = [(k, ast.literal_eval(s)) for k, s in zip(zu["keys"], zu["shapes"])]
    sd = seeded_state_dict(shapes, int(z["seed"]), gate_scale=10.0, affine_scale=10.0)
    cqt = OracleCQT(kw["num_octs"], kw["bins_per_oct"], "oct", ("kaiser", 1), kw["fs"], kw["audio_len"])
    net = OU.OracleUnet(kw["num_octs"], kw["bins_per_oct"], cqt).load_state_dict(sd)
    T = int(z["T"])
    edm = OracleEDM(Schurn=float(z["Schurn"]), Stmin=float(z["Stmin"]), Stmax=float(z["Stmax"]))
    t = edm.create_schedule(T)
    gamma = edm.get_gamma(t)
    assert np.array_equal(t.numpy(), z["t"]) and np.array_equal(gamma.numpy(), z["gamma"])
    g = z["gamma"][:T]
    assert g[0] == 0 and g[T - 1] == 0 and (g > 0).sum() >= 3 and (g == 0).sum() >= 3 and z["t"][-1] == 0 and int(z["n_eval"]) == 2 * T - 1
    y, mask = torch.from_numpy(z["y"]), torch.from_numpy(z["mask"])
    for e in range(int(z["n_eval"])):
        x = torch.from_numpy(z[f"e{e}.x"]).clone().requires_grad_()
        sig = torch.full((1, 1), float(z[f"e{e}.t"]))
        xh = cqt.apply_hpf_DC(edm.denoiser(x, net, sig))
        norm = torch.linalg.norm(y - mask * xh, dim=1, ord=2)
        gr = torch.autograd.grad(norm.sum(), x)[0]
        assert rel_l2(xh.detach(), z[f"e{e}.x_hat"]) < 2e-5, e
        assert rel_l2(gr, z[f"e{e}.rec_grads"]) < 1e-4, e
        assert abs(float(norm) - float(z[f"e{e}.norm"][0])) < 2e-5 * float(norm), e
    smp = OracleSampler(net, edm, T=T, xi=0.25, hann_size=20, audio_len=kw["audio_len"])
    torch.manual_seed(int(z["noise_seed"]))
    res = smp.predict_inpainting(y, mask, rid=True)
    errs = [rel_l2(res[6][i], z["xt2"][i]) for i in range(T)]
    print("oracle free-running trajectory vs reference, state after every step:", " ".join(f"{e:.1e}" for e in errs))
    assert max(errs) < 1e-5 and rel_l2(res[0], z["out"]) < 1e-5       # (same torch CPU kernels as the reference: 3e-7 measured)



def test_full_size_guided_evaluation_oracle_vs_reference_fixture():
    """unet_full_cfgA_guided.npz (one guided evaluation of the reference's Sampler + EDM + full-size cfg-A U-Net): the oracle chain at FULL size
    (186 M parameters; ~1.5 min and ~18 GB of host RAM) reproduces norm, the strided samples and the seeded projections of x_hat and rec_grads."""
    from audio_inpainting_diffusion_amd.config import make_args
    from audio_inpainting_diffusion_amd.init import seeded_normal, seeded_state_dict
    from audio_inpainting_diffusion_amd.masks import long_gap_mask
    from oracle.edm import OracleEDM
    z = np.load(os.path.join(GOLDEN, "unet_full_cfgA_guided.npz"))
    zk = np.load(os.path.join(GOLDEN, "unet_full_cfgA_keys.npz")) if os.path.exists(os.path.join(GOLDEN, "unet_full_cfgA_keys.npz")) else None
    args = make_args("maestro22k")
    Ls = args.exp.audio_len
    if zk is not None:
        shapes = [(k, ast.literal_eval(s)) for k, s in zip(zk["keys"], zk["shapes"])]
    else:                                                   # key / shape list of the full network from the product class on the meta device (no memory)
        from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
        shapes = [(k, tuple(v.shape)) for k, v in Unet_CQT_oct_with_attention(args, torch.device("meta")).state_dict().items()]
    sd = seeded_state_dict(shapes, 0, gate_scale=10.0, affine_scale=10.0)
    cqt = OracleCQT(7, 64, "oct", ("kaiser", 1), 22050, Ls)
    net = OU.OracleUnet(7, 64, cqt).load_state_dict(sd)
    edm = OracleEDM()
    x = (torch.from_numpy(seeded_normal(21, 0, Ls)).reshape(1, Ls) * 0.3).requires_grad_()
    y = torch.from_numpy(seeded_normal(22, 0, Ls)).reshape(1, Ls) * 0.063
    mask = long_gap_mask(Ls, 22050, 300)
    xh = cqt.apply_hpf_DC(edm.denoiser(x, net, torch.full((1, 1), float(z["t"]))))
    norm = torch.linalg.norm(y * mask - mask * xh, dim=1, ord=2)
    g = torch.autograd.grad(norm.sum(), x)[0]
    assert abs(float(norm) - float(z["norm"][0])) < 2e-5 * float(norm)
    for name, t, stream in (("x_hat", xh.detach(), 1), ("rec_grads", g, 2)):
        tv = t.double().reshape(-1).numpy()
        ref_p = z[name + "_proj"]
        probes = np.stack([seeded_normal(7000 + j, stream, tv.size) for j in range(8)]).astype(np.float64)
        assert rel_l2(t[:, ::97], z[name + "_s"]) < 2e-5, name
        assert np.abs(probes @ tv - ref_p[:8]).max() < 1e-4 * np.sqrt(ref_p[-1]), name
