"""Training step on the HIP kernels against torch.autograd / torch.optim over the CPU oracle.

 * parameter gradients of mean((net(input, c_noise) - target)**2) for EVERY tensor of the state_dict, vs autograd through the
   oracle U-Net (which is pinned to the reference by tests/golden/unet_small_*.npz);
 * aid_sumsq / aid_adam / aid_ema vs clip_grad_norm_ + torch.optim.Adam + the reference's EMA rule;
 * two whole iterations (EDM preconditioning, loss, backward, lr ramp-up, clipping, Adam, EMA) vs the same loop on the oracle."""
import ast
import math
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _setup(tag="a"):
    from audio_inpainting_diffusion_amd.config import small_args
    from audio_inpainting_diffusion_amd.init import seeded_init_
    from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
    from oracle.nsgt_cqt import OracleCQT
    from oracle.unet import OracleUnet
    z = np.load(os.path.join(GOLDEN, f"unet_small_{tag}.npz"))
    kw = ast.literal_eval(str(z["cfg"]))
    args = small_args(**kw)
    net = seeded_init_(Unet_CQT_oct_with_attention(args, torch.device(DEV)), int(z["seed"]), gate_scale=10.0, affine_scale=10.0)
    cqt = OracleCQT(kw["num_octs"], kw["bins_per_oct"], "oct", ("kaiser", 1), kw["fs"], kw["audio_len"])
    orc = OracleUnet(kw["num_octs"], kw["bins_per_oct"], cqt).load_state_dict(net.state_dict())
    return net, orc, z, kw, args


def _oracle_grads(orc, x, cn, target, hpf=False):
    for v in orc.sd.values():
        v.requires_grad_(True)
        v.grad = None
    y = orc(x, cn)
    err = y - target
    if hpf:
        err = orc.CQTransform.apply_hpf_DC(err)
    loss = (err ** 2).mean()
    loss.backward()
    g = {k: (v.grad.clone() if v.grad is not None else torch.zeros_like(v)) for k, v in orc.sd.items()}
    for v in orc.sd.values():
        v.requires_grad_(False)
    return float(loss), g


@pytest.mark.parametrize("tag,hpf", [("a", False), ("b", False), ("a", True), ("c", False)])     # c: use_fencoding + bias_qkv + use_rel_pos
def test_parameter_gradients_vs_oracle_autograd(tag, hpf):
    net, orc, z, kw, _ = _setup(tag)
    x, cn = torch.from_numpy(z["x"]), torch.from_numpy(z["cnoise"])
    target = torch.randn(x.shape, generator=torch.Generator().manual_seed(5)) * 0.2
    loss, err2 = net.loss_and_grads(x.to(DEV), cn.to(DEV), target.to(DEV), hpf_error=hpf)
    ref_loss, ref = _oracle_grads(orc, x, cn, target, hpf)
    assert abs(float(loss) - ref_loss) < 1e-5 * abs(ref_loss)
    got = {k: v.detach().cpu() for k, v in net.train_state(x.shape[0])["builder"].pgrad.items()}
    tot_num = tot_den = 0.0
    worst = ("", 0.0)
    gmax = max(float(v.norm()) for v in ref.values())
    for k, r in ref.items():
        if k.endswith("RFF_freq") or k.endswith("kernel") or k.endswith("embeddings"):
            assert float(got[k].abs().max()) == 0.0              # frozen tensors
            continue
        d = float((got[k].double() - r.double()).norm())
        tot_num += d * d
        tot_den += float(r.double().norm()) ** 2
        e = d / (float(r.norm()) + 1e-6 * gmax)
        if e > worst[1]:
            worst = (k, e)
    tot = math.sqrt(tot_num / tot_den)
    print(f"parameter gradients ({tag}, hpf={hpf}): all tensors together rel-L2 = {tot:.2e}; worst tensor {worst[0]} {worst[1]:.2e}; loss {float(loss):.6f}")
    assert tot < 1e-4 and worst[1] < 1e-3


def test_optimizer_and_ema_kernels_vs_torch():
    from audio_inpainting_diffusion_amd import _lib as L
    n = 100003
    g0 = torch.Generator().manual_seed(1)
    p0 = torch.randn(n, generator=g0)
    ema0 = p0.clone()
    pd, m, v = p0.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    ws = torch.zeros(L.AID_SUMSQ_BLOCKS, device=DEV, dtype=torch.float64)
    gst = torch.zeros(2, device=DEV)
    pr = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([pr], lr=2e-4, betas=(0.9, 0.999), eps=1e-8)
    for t in range(1, 5):
        g = torch.randn(n, generator=g0) * (3.0 if t % 2 else 1e-3)        # clipped on odd steps only
        gd = g.to(DEV)
        L.call("aid_sumsq", L.SumsqParams(gd.data_ptr(), ws.data_ptr(), gst.data_ptr(), n, 1.0))
        L.call("aid_adam", L.AdamParams(pd.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), gst[1:].data_ptr(), n, 2e-4, 0.9, 0.999, 1e-8,
                                        1.0 - 0.9 ** t, math.sqrt(1.0 - 0.999 ** t)))
        pr.grad = g.clone()
        nrm = torch.nn.utils.clip_grad_norm_([pr], 1.0)
        opt.step()
        assert abs(float(gst[0]) - float(nrm)) < 1e-5 * float(nrm)
        assert rel_l2(pd.cpu(), pr.detach()) < 1e-6 and rel_l2(pd.cpu() - p0, pr.detach() - p0) < 1e-4     # (the deltas are ~1e-4 of the values)
    ema = ema0.to(DEV)
    L.call("aid_ema", L.EmaParams(ema.data_ptr(), pd.data_ptr(), n, 0.75))
    assert rel_l2(ema.cpu(), ema0 * 0.75 + pd.cpu() * 0.25) < 1e-6


@pytest.mark.parametrize("tag", ["a", "c"])          # c: use_fencoding + bias_qkv + use_rel_pos (frozen tables, qk bias and relative-position embedding gradients)
def test_two_training_iterations_vs_oracle_loop(tag):
    """trainer.py:253-304 on the small network: same sigma / noise on both sides; loss values, EMA and parameters after two steps."""
    from audio_inpainting_diffusion_amd.edm import EDM
    from audio_inpainting_diffusion_amd.training import Trainer, prepare_train_preconditioning
    net, orc, z, kw, args = _setup(tag)
    edm = EDM(args)
    tr = Trainer(net, edm, lr=2e-3, lr_rampup_it=2, max_grad_norm=1.0, ema_rate=0.9, ema_rampup=8, batch=2)
    B, Ls = 2, kw["audio_len"]
    g0 = torch.Generator().manual_seed(9)
    keys = list(orc.sd.keys())
    params = [torch.nn.Parameter(orc.sd[k].clone()) for k in keys]
    trainable = [p for k, p in zip(keys, params) if not (k.endswith("RFF_freq") or k.endswith("kernel") or k.endswith("embeddings"))]
    for k, p in zip(keys, params):
        if k.endswith("embeddings"):
            p.requires_grad_(False)
    opt = torch.optim.Adam(trainable, lr=2e-3, betas=(0.9, 0.999), eps=1e-8)
    ema_ref = {k: orc.sd[k].clone() for k in keys}
    losses, ref_losses = [], []
    for it in range(3):
        audio = torch.randn(B, Ls, generator=g0) * 0.063
        sigma = (torch.rand(B, 1, generator=g0) * 0.5 + 0.05)
        noise = torch.randn(B, Ls, generator=g0) * sigma
        losses.append(float(tr.train_step(audio.to(DEV), sigma, noise.to(DEV))))
        # ---- the reference's iteration on the oracle --------------------------------------------------------------------
        orc.sd = {k: p for k, p in zip(keys, params)}
        inp, target, cnoise = prepare_train_preconditioning(edm, audio, sigma, noise)
        opt.zero_grad()
        loss = ((orc(inp, cnoise) - target) ** 2).mean()
        loss.backward()
        for gpar in opt.param_groups:
            gpar["lr"] = 2e-3 * min(it / max(2, 1e-8), 1)
        torch.nn.utils.clip_grad_norm_(trainable, 1.0)
        opt.step()
        t = it * 2
        s = float(np.clip(t / 8, 0.0, 0.9)) if t < 8 else 0.9
        with torch.no_grad():
            for k, p in zip(keys, params):
                ema_ref[k].copy_(ema_ref[k] * s + p * (1 - s))
        ref_losses.append(float(loss))
    print("losses", losses, "oracle", ref_losses)
    for a_, b_ in zip(losses, ref_losses):
        assert abs(a_ - b_) < 1e-4 * abs(b_)
    sd = net.state_dict()
    num = math.sqrt(sum(float((sd[k].cpu() - params[i].detach()).norm()) ** 2 for i, k in enumerate(keys)))
    den = math.sqrt(sum(float(params[i].detach().norm()) ** 2 for i in range(len(keys))))
    print(f"parameters after 3 iterations: rel-L2 vs oracle loop = {num / den:.2e}")
    assert num / den < 1e-4
    ema = tr.ema_state_dict()
    num = math.sqrt(sum(float((ema[k].cpu() - ema_ref[k]).norm()) ** 2 for k in keys))
    assert num / den < 1e-4


def test_full_size_parameter_gradients_vs_oracle_autograd():
    """Full-size 22.05 kHz network (186 M parameters), B=1: loss and a spread of parameter gradients (first / deep 5x3 weights,
    qk projection, 1x1 projections, norm gammas, affine / gate Linears, embedding MLP) vs torch.autograd over the CPU oracle."""
    from audio_inpainting_diffusion_amd.config import make_args
    from audio_inpainting_diffusion_amd.init import seeded_init_, seeded_normal
    from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
    from oracle.nsgt_cqt import OracleCQT
    from oracle.unet import OracleUnet
    args = make_args("maestro22k")
    net = seeded_init_(Unet_CQT_oct_with_attention(args, torch.device(DEV)), 0, gate_scale=10.0, affine_scale=10.0)
    Ls = args.exp.audio_len
    x = torch.from_numpy(seeded_normal(61, 0, Ls)).reshape(1, Ls) * 0.5
    target = torch.from_numpy(seeded_normal(62, 0, Ls)).reshape(1, Ls) * 0.3
    cn = torch.tensor([[-0.45]])
    loss, _ = net.loss_and_grads(x.to(DEV), cn.to(DEV), target.to(DEV))
    got = net.train_state(1)["builder"].pgrad
    orc = OracleUnet(7, 64, OracleCQT(7, 64, "oct", ("kaiser", 1), 22050, Ls)).load_state_dict(net.state_dict())
    ref_loss, ref = _oracle_grads(orc, x, cn, target)
    assert abs(float(loss) - ref_loss) < 1e-5 * abs(ref_loss)
    keys = ["downs.0.0.proj_in.weight", "downs.0.2.H.1.weight", "downs.0.1.weight", "downs.3.2.H.4.weight", "downs.6.2.H.6.weight",
            "downs.5.2.attn_block.qk.weight", "downs.4.2.attn_block.proj_in.weight", "downs.4.2.attn_block.proj_out.weight",
            "middle.0.1.H.3.weight", "middle.0.0.proj_out.weight", "middle.0.0.res_conv.weight", "ups.0.1.H.0.weight", "ups.6.1.H.1.weight",
            "ups.3.0.proj_out.weight", "downs.2.2.norm.1.gamma", "downs.6.2.norm2.gamma", "downs.1.2.affine.2.weight", "downs.1.2.affine.2.bias",
            "ups.2.1.gate.3.weight", "ups.2.1.gate.3.bias", "downs.6.2.gate2.weight", "embedding.MLP.0.weight", "embedding.MLP.2.bias"]
    worst = 0.0
    for k in keys:
        e = rel_l2(got[k].cpu(), ref[k])
        worst = max(worst, e)
        print(f"  {k:44s} rel-L2 {e:.2e}")
        assert e < 1e-4, k
    train = [k for k in ref if not (k.endswith("RFF_freq") or k.endswith("kernel"))]      # (frozen in the reference: requires_grad=False / buffers)
    assert all(float(got[k].abs().max()) == 0.0 for k in ref if k not in train)
    num = math.sqrt(sum(float((got[k].cpu().double() - ref[k].double()).norm()) ** 2 for k in train))
    den = math.sqrt(sum(float(ref[k].double().norm()) ** 2 for k in train))
    print(f"full-size parameter gradients: loss {float(loss):.6f} (oracle {ref_loss:.6f}); all trainable tensors together rel-L2 = {num / den:.2e}; worst listed {worst:.2e}")
    assert num / den < 1e-4


def test_reference_style_training_step_runs_unchanged():
    """The reference's iteration, verbatim (trainer.py:253-281): ``error, sigma = diff_params.loss_fn(network, audio); loss = error.mean();
    loss.backward(); clip_grad_norm_; optimizer.step()`` with torch.optim.Adam on OUR network's parameters -- the parameter gradients
    come from the HIP backward plan through autograd.TrainFn -- against the same lines run on the oracle."""
    from audio_inpainting_diffusion_amd.edm import EDM
    net, orc, z, kw, args = _setup("a")
    edm = EDM(args)
    net.train()
    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    keys = list(orc.sd.keys())
    params = [torch.nn.Parameter(orc.sd[k].clone()) for k in keys]
    trainable = [p for k, p in zip(keys, params) if not (k.endswith("RFF_freq") or k.endswith("kernel"))]
    opt_ref = torch.optim.Adam(trainable, lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    B, Ls = 2, kw["audio_len"]
    for it in range(2):
        audio = torch.randn(B, Ls, generator=torch.Generator().manual_seed(20 + it)) * 0.063
        torch.manual_seed(100 + it)                            # loss_fn draws sigma and the noise from the global CPU generator
        opt.zero_grad()
        error, sigma = edm.loss_fn(net, audio.to(DEV))
        loss = error.mean()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
        opt.step()
        # ---- the same lines on the oracle -------------------------------------------------------------------------------------
        orc.sd = {k: p for k, p in zip(keys, params)}
        torch.manual_seed(100 + it)
        opt_ref.zero_grad()
        error_r, sigma_r = edm.loss_fn(orc, audio)
        loss_r = error_r.mean()
        loss_r.backward()
        torch.nn.utils.clip_grad_norm_(trainable, 1.0)
        opt_ref.step()
        assert torch.equal(sigma.cpu(), sigma_r) and abs(float(loss) - float(loss_r)) < 1e-4 * float(loss_r)
    sd = net.state_dict()
    num = math.sqrt(sum(float((sd[k].cpu() - params[i].detach()).norm()) ** 2 for i, k in enumerate(keys)))
    den = math.sqrt(sum(float(params[i].detach().norm()) ** 2 for i in range(len(keys))))
    print(f"reference-style loop, 2 iterations: parameters rel-L2 vs oracle = {num / den:.2e}")
    assert num / den < 1e-4
    # the guidance branch still takes the input-VJP path while the module sits in train() mode (the reference's tester never calls eval())
    x = torch.randn(1, Ls, generator=torch.Generator().manual_seed(3)).to(DEV).requires_grad_()
    y = net(x, torch.tensor([[0.1]], device=DEV))
    gx = torch.autograd.grad(y.sum(), x)[0]
    assert torch.isfinite(gx).all() and all(p.grad is None or True for p in net.parameters())


def test_three_iterations_vs_reference_trainer_golden():
    """tests/golden/train_small.npz: three iterations of the REFERENCE's Trainer.train_step + update_ema (its EDM.loss_fn, setup_optimizer's
    Adam, lr ramp-up, clipping, EMA) on the small network -- our all-HIP Trainer with the same hyper-parameters, segments, sigma and noise."""
    from audio_inpainting_diffusion_amd.config import small_args
    from audio_inpainting_diffusion_amd.edm import EDM
    from audio_inpainting_diffusion_amd.init import seeded_init_, seeded_normal
    from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
    from audio_inpainting_diffusion_amd.training import Trainer
    z = np.load(os.path.join(GOLDEN, "train_small.npz"))
    kw, hp = ast.literal_eval(str(z["cfg"])), ast.literal_eval(str(z["hp"]))
    args = small_args(**kw)
    net = seeded_init_(Unet_CQT_oct_with_attention(args, torch.device(DEV)), int(z["seed"]), gate_scale=10.0, affine_scale=10.0)
    edm = EDM(args)
    tr = Trainer(net, edm, lr=hp["lr"], lr_rampup_it=hp["lr_rampup_it"], use_grad_clip=hp["use_grad_clip"], max_grad_norm=hp["max_grad_norm"],
                 ema_rate=hp["ema_rate"], ema_rampup=hp["ema_rampup"], batch=hp["batch"])
    B, Ls = hp["batch"], kw["audio_len"]
    for it in range(int(z["n_it"])):
        audio = torch.from_numpy(seeded_normal(41, it, B * Ls)).reshape(B, Ls) * 0.063
        torch.manual_seed(500 + it)                          # the reference's draws: sigma = f(torch.rand(B)), then noise = torch.randn(B, L) * sigma
        sigma = edm.sample_ptrain_safe(B).unsqueeze(-1)
        assert np.array_equal(sigma.reshape(-1).numpy(), z[f"sigma.{it}"])
        noise = torch.randn(B, Ls) * sigma
        loss = float(tr.train_step(audio.to(DEV), sigma, noise.to(DEV)))
        assert abs(loss - float(z["loss"][it])) < 1e-4 * abs(float(z["loss"][it])), (it, loss, float(z["loss"][it]))
    from conftest import projected_rel_error
    ep = projected_rel_error(z, "p", net.state_dict())
    ee = projected_rel_error(z, "ema", tr.ema_state_dict())
    print(f"after 3 iterations vs the reference trainer: parameters rel-L2 (projected) = {ep:.2e}, EMA = {ee:.2e}")
    assert ep < 1e-4 and ee < 1e-4


def test_trainer_checkpoint_roundtrip_in_the_reference_shape(tmp_path):
    """Trainer.state_dict() has the reference checkpoint's shape (training/trainer.py:186-199): its 'optimizer' entry loads into the
    torch.optim.Adam that the reference's setup_optimizer builds (utils/setup.py:55-58) and a state_dict of THAT optimiser loads back;
    resuming from a saved checkpoint continues bit-identically (Adam moments, step count -> bias correction, lr ramp-up, EMA, it)."""
    from audio_inpainting_diffusion_amd.edm import EDM
    from audio_inpainting_diffusion_amd.training import Trainer
    net, orc, z, kw, args = _setup("a")
    edm = EDM(args)
    kwt = dict(lr=2e-3, lr_rampup_it=5, max_grad_norm=1.0, ema_rate=0.9, ema_rampup=8, batch=2)
    tr = Trainer(net, edm, **kwt)
    B, Ls = 2, kw["audio_len"]
    g0 = torch.Generator().manual_seed(11)
    data = []
    for _ in range(4):
        audio = torch.randn(B, Ls, generator=g0) * 0.063
        sigma = torch.rand(B, 1, generator=g0) * 0.5 + 0.05
        data.append((audio.to(DEV), sigma, (torch.randn(B, Ls, generator=g0) * sigma).to(DEV)))
    for d in data[:2]:
        tr.train_step(*d)
    ck = tr.state_dict(args="cfg")
    assert set(ck) == {"it", "network", "optimizer", "ema", "args"} and ck["it"] == 2
    assert list(ck["network"]) == list(net.state_dict()) and list(ck["ema"]) == list(net.state_dict())
    # the reference's optimiser accepts it, and what it saves comes back
    ref_opt = torch.optim.Adam(net.parameters(), lr=2e-3, betas=(0.9, 0.999), eps=1e-8)
    ref_opt.load_state_dict(ck["optimizer"])
    names = [k for k, _ in net.named_parameters()]
    i_w = names.index("downs.0.2.H.0.weight")
    st = ref_opt.state[list(net.parameters())[i_w]]
    assert float(st["step"]) == 2.0 and torch.equal(st["exp_avg"], ck["optimizer"]["state"][i_w]["exp_avg"])
    assert names.index("embedding.RFF_freq") not in ck["optimizer"]["state"]          # frozen parameter: no Adam state, like torch
    path = str(tmp_path / "ck.pt")
    tr.save_checkpoint(path, args="cfg")
    # continue the original run
    ref_losses = [float(tr.train_step(*d)) for d in data[2:]]
    ref_params = {k: v.clone() for k, v in net.state_dict().items()}
    ref_ema = {k: v.clone() for k, v in tr.ema_state_dict().items()}
    # a fresh network + trainer resumed from the file
    net2, *_ = _setup("a")
    tr2 = Trainer(net2, edm, **kwt)
    assert tr2.load_state_dict(torch.load(path, map_location=DEV, weights_only=False))
    assert tr2.it == 2 and tr2.steps == 2
    losses = [float(tr2.train_step(*d)) for d in data[2:]]
    assert losses == ref_losses
    for k, v in net2.state_dict().items():
        assert torch.equal(v, ref_params[k]), k
    for k, v in tr2.ema_state_dict().items():
        assert torch.equal(v, ref_ema[k]), k
    # a checkpoint whose optimiser state was written by torch.optim.Adam itself (the reference trainer's) restores the same moments
    ck2 = dict(ck, optimizer=ref_opt.state_dict())
    tr3 = Trainer(_setup("a")[0], edm, **kwt)
    tr3.load_state_dict(ck2)
    m_ck = torch.cat([ck["optimizer"]["state"][i]["exp_avg"].reshape(-1) for i in sorted(ck["optimizer"]["state"])])
    m_3 = torch.cat([tr3._views(tr3.m)[names[i]].reshape(-1) for i in sorted(ck["optimizer"]["state"])])
    assert tr3.steps == 2 and torch.equal(m_3, m_ck)


def test_trainer_follows_rehomed_parameters():
    """If the parameters are re-homed after the Trainer was built (here: a fresh flat buffer), optimiser and EMA must keep updating the LIVE weights."""
    from audio_inpainting_diffusion_amd.edm import EDM
    from audio_inpainting_diffusion_amd.training import Trainer
    net, orc, z, kw, args = _setup("a")
    tr = Trainer(net, EDM(args), lr=2e-3, lr_rampup_it=0, batch=2)
    for p in list(net.parameters()) + list(net.buffers()):      # what load_state_dict(assign=True) / .to() do: new storage per tensor
        p.data = p.data.clone()
    net._aid_flat = None
    B, Ls = 2, kw["audio_len"]
    audio = torch.randn(B, Ls, generator=torch.Generator().manual_seed(1)) * 0.063
    before = net.state_dict()["downs.0.2.H.0.weight"].clone()
    tr.it = 1
    tr.train_step(audio.to(DEV))
    after = net.state_dict()["downs.0.2.H.0.weight"]
    assert float((after - before).abs().max()) > 0, "the optimiser step did not reach the live parameters"


def test_accumulation_rounds_and_a_weighted_loss_vs_oracle_loop():
    """trainer.py:259-266 with num_accumulation_rounds = 2 (two loss.backward() into the same gradients) and the A-weighted, DC-corrected error of
    edm.py:180-190 (FirFilter = the reference's FIRFilter("aw"), golden-pinned taps): loss, parameters and EMA after two iterations vs the same loop in
    torch autograd over the CPU oracle."""
    from audio_inpainting_diffusion_amd.edm import EDM
    from audio_inpainting_diffusion_amd.training import Trainer, a_weighting_taps, prepare_train_preconditioning
    net, orc, z, kw, args = _setup("a")
    edm = EDM(args)
    taps = a_weighting_taps(kw["fs"], 101)
    tr = Trainer(net, edm, lr=2e-3, lr_rampup_it=2, max_grad_norm=1.0, ema_rate=0.9, ema_rampup=8, batch=2, use_cqt_DC_correction=True, aweighting_taps=taps)
    B, Ls = 2, kw["audio_len"]
    g0 = torch.Generator().manual_seed(21)
    keys = list(orc.sd.keys())
    params = [torch.nn.Parameter(orc.sd[k].clone()) for k in keys]
    trainable = [p for k, p in zip(keys, params) if not (k.endswith("RFF_freq") or k.endswith("kernel"))]
    opt = torch.optim.Adam(trainable, lr=2e-3, betas=(0.9, 0.999), eps=1e-8)
    wt = torch.from_numpy(taps).view(1, 1, -1)
    for it in range(3):
        rounds = []
        for _ in range(2):
            audio = torch.randn(B, Ls, generator=g0) * 0.063
            sigma = torch.rand(B, 1, generator=g0) * 0.5 + 0.05
            rounds.append((audio, sigma, torch.randn(B, Ls, generator=g0) * sigma))
        loss = float(tr.train_step([a.to(DEV) for a, _, _ in rounds], [s for _, s, _ in rounds], [n.to(DEV) for _, _, n in rounds]))
        orc.sd = {k: p for k, p in zip(keys, params)}
        opt.zero_grad()
        for audio, sigma, noise in rounds:
            inp, target, cnoise = prepare_train_preconditioning(edm, audio, sigma, noise)
            err = orc.CQTransform.apply_hpf_DC(orc(inp, cnoise) - target)
            err = torch.nn.functional.conv1d(err.unsqueeze(1), wt, padding=50).squeeze(1)
            ref_loss = (err ** 2).mean()
            ref_loss.backward()
        for gpar in opt.param_groups:                    # lr ramp-up (trainer.py:270-274): 0 at it = 0
            gpar["lr"] = 2e-3 * min(it / max(2, 1e-8), 1)
        torch.nn.utils.clip_grad_norm_(trainable, 1.0)
        opt.step()
        ref_loss = ref_loss.detach()
        print(f"iteration {it}: loss of the last round {loss:.6f} vs oracle {float(ref_loss):.6f}")
        assert abs(loss - float(ref_loss)) < 1e-4 * abs(float(ref_loss))
    sd = net.state_dict()
    num = math.sqrt(sum(float((sd[k].cpu() - params[i].detach()).norm()) ** 2 for i, k in enumerate(keys)))
    den = math.sqrt(sum(float(params[i].detach().norm()) ** 2 for i in range(len(keys))))
    print(f"parameters after 3 iterations x 2 rounds, A-weighted loss: rel-L2 vs oracle loop = {num / den:.2e}")
    assert num / den < 1e-4


def test_edm_mirror_a_weighted_loss_vs_reference_filter_golden():
    """EDM(args) with diff_params.aweighting.use_aweighting (edm.py:33-34, :189-190): the mirror's ``AW`` is the reference's FIRFilter("aw") on the HIP FIR
    kernel -- output against the reference's own filter output (tests/golden/aweighting.npz), gradient against torch's conv1d, and ``loss_fn`` applies it."""
    from audio_inpainting_diffusion_amd import _lib
    from audio_inpainting_diffusion_amd.config import make_args
    from audio_inpainting_diffusion_amd.edm import EDM
    z = np.load(os.path.join(GOLDEN, "aweighting.npz"))
    args = make_args(audio_len=4096, T=2, xi=0.0)
    args.diff_params.aweighting.use_aweighting, args.diff_params.aweighting.ntaps = True, 101
    assert args.exp.sample_rate == 22050
    edm = EDM(args)
    x = torch.from_numpy(z["x"]).to(DEV).requires_grad_()
    y = edm.AW(x)
    assert rel_l2(y.detach().cpu(), z["y22050"]) < 1e-6
    w = torch.randn(y.shape, generator=torch.Generator().manual_seed(2)).to(DEV)
    (gx,) = torch.autograd.grad((y * w).sum(), x)
    xc = torch.from_numpy(z["x"]).requires_grad_()
    yc = torch.nn.functional.conv1d(xc.unsqueeze(1), torch.from_numpy(z["taps22050"]).view(1, 1, -1), padding=50).squeeze(1)
    (gc,) = torch.autograd.grad((yc * w.cpu()).sum(), xc)
    assert rel_l2(gx.cpu(), gc) < 1e-6
    with pytest.raises(_lib.AidError):
        edm.AW(torch.zeros(1, 4096))

    class _Net(torch.nn.Module):                       # loss_fn with a trivial net: error = net(inp) - target, filtered, squared
        def forward(self, inp, cnoise):
            return 0.5 * inp
    torch.manual_seed(4)
    audio = torch.from_numpy(z["x"]).to(DEV) * 0.063
    err2, sigma = edm.loss_fn(_Net(), audio)
    torch.manual_seed(4)
    plain = EDM(make_args(audio_len=4096, T=2, xi=0.0))
    s2 = plain.sample_ptrain_safe(audio.shape[0]).unsqueeze(-1).to(DEV)
    inp, target, _ = plain.prepare_train_preconditioning(audio, s2)
    err = (0.5 * inp - target).cpu()
    ref = torch.nn.functional.conv1d(err.unsqueeze(1), torch.from_numpy(z["taps22050"]).view(1, 1, -1), padding=50).squeeze(1) ** 2
    assert torch.equal(sigma, s2) and rel_l2(err2.cpu(), ref) < 1e-5
    assert abs(float(edm.lambda_w(torch.tensor(0.7))) * float(plain.cout(torch.tensor(0.7)) ** 2) - 1) < 1e-5
    assert edm.sample_ptrain(5).shape == (5,) and float(edm.sample_ptrain(100).min()) >= edm.sigma_min
