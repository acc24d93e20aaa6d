#!/usr/bin/env python
"""Generate the golden fixtures in tests/golden/ by IMPORTING the reference (dev container only).

    python tests/golden/make_golden.py [--full]

The reference at /root/reference is imported, never copied.  Two throw-away stub modules are written to a
temp dir so the import succeeds in this image:
  * ``torchaudio``       -- imported by the reference but never used on this path;
  * ``cqt_nsgt_pytorch`` -- the un-vendored CQT package; the stub exposes ``CQT_nsgt`` = our oracle
                            NSGT-CQT (oracle/nsgt_cqt.py), so whole-network goldens = reference U-Net
                            body + our CQT definition (CQT parity itself is unpinned, see oracle/__init__.py).
Weights come from the seeded counter-based initialiser (audio_inpainting_diffusion_amd/init.py) with O(1)
gates, loaded into the reference modules through ``load_state_dict``.

Fixtures written (inputs + reference outputs only -- data, not code):
  ops_small.npz      per-op: BiasFreeGroupNorm, UpDownResample up/down, RFF_MLP_Block, TimeAttentionBlock,
                     ResnetBlock in the five structural variants the U-Net uses
  unet_small_a.npz / unet_small_b.npz   whole network (reduced widths): input + output + state_dict key/shape list
                     (weights are regenerated from the seeded initialiser, seed stored)
  edm_schedule.npz   create_schedule / get_gamma for T in {35,36,70,128} with the tester parameters
  sampler_toy.npz    full sampler trajectories (reference Sampler + EDM driving a toy denoiser)
  sampler_resample.npz  predict_resample trajectories through generic degradation lambdas (--only resample)
  sampler_dc.npz         data_consistency.type = 'end' / 'always' on both branches (guided, replacement)
  sampler_rid.npz        rid=True: the sampler's per-step debug buffers (denoised, grads, grad_update, pocs, xt, xt2, t)
  sampler_spectral.npz   spectrogram inpainting: apply_spectral_mask + full trajectories (guided / replacement)
  unet_full_cfgA.npz (--full) full-size 22.05 kHz network output for the seeded weights/input (B=1)
  sampler_guided_unet.npz (--only guided)   reference Sampler + EDM + reference U-Net (small a / c): x_hat, rec_grads, norm of every evaluation
  sampler_guided_traj.npz (--only guided_traj)   a WHOLE T = 10 guided trajectory of the reference chain (churn window, Heun + final Euler step)
  unet_full_cfgA_guided.npz (--only full_guided)   one guided evaluation of the reference chain at full size (projections + strided samples)
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def _setup_imports():
    stub = tempfile.mkdtemp(prefix="aid_stubs_")
    os.makedirs(os.path.join(stub, "torchaudio"))
    open(os.path.join(stub, "torchaudio", "__init__.py"), "w").close()
    os.makedirs(os.path.join(stub, "cqt_nsgt_pytorch"))
    with open(os.path.join(stub, "cqt_nsgt_pytorch", "__init__.py"), "w") as f:
        f.write("from oracle.nsgt_cqt import OracleCQT as CQT_nsgt\n")
    # import-only stand-ins for the logging / audio packages training/trainer.py pulls in at import time (none is called by train_step / update_ema)
    for mod in ("librosa", "wandb", "omegaconf", "soundfile", "plotly", "plotly/express", "plotly/graph_objects"):
        os.makedirs(os.path.join(stub, mod), exist_ok=True)
        open(os.path.join(stub, mod, "__init__.py"), "w").close()
    for p in (stub, REF, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)


def _np(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


def _seed_module(mod, seed, gate_scale=10.0, affine_scale=10.0):
    from audio_inpainting_diffusion_amd.init import seeded_init_
    return seeded_init_(mod, seed, gate_scale, affine_scale)


def gen_ops(out):
    import networks.unet_cqt_oct_with_projattention_adaLN_2 as R
    from audio_inpainting_diffusion_amd.config import Cfg
    from audio_inpainting_diffusion_amd.init import seeded_normal
    d = {}
    rnd = lambda stream, *shape: torch.from_numpy(seeded_normal(1234, stream, int(np.prod(shape)))).reshape(*shape)
    init = dict(init_mode="kaiming_uniform", init_weight=np.sqrt(1 / 3))
    init_zero = dict(init_mode="kaiming_uniform", init_weight=1e-7)
    attn = Cfg(num_heads=8, attn_dropout=0.0, bias_qkv=False, N=0, rel_pos_num_buckets=32,
               rel_pos_max_distance=64, use_rel_pos=False, Nproj=8)
    with torch.no_grad():
        # group norm
        gn = R.BiasFreeGroupNorm(16, 8)
        gn.gamma.copy_(1.0 + 0.3 * rnd(1, 1, 16, 1, 1))
        x = rnd(2, 2, 16, 5, 12) * 3.0 + 0.5
        d["gn.x"], d["gn.gamma"], d["gn.y"] = x.numpy(), gn.gamma.numpy(), gn(x).numpy()
        # resamplers
        x = rnd(3, 2, 3, 4, 16)
        d["rs.x"] = x.numpy()
        d["rs.down"] = R.UpDownResample(down=True, mode_resample="T")(x).numpy()
        d["rs.up"] = R.UpDownResample(up=True, mode_resample="T")(x).numpy()
        # embedding
        emb = _seed_module(R.RFF_MLP_Block(emb_dim=32, init=init), 5)
        sig = torch.tensor([[-2.1], [-0.3], [0.0]])
        for k, v in _np(emb.state_dict()).items():
            d["emb.sd.embedding." + k] = v
        d["emb.sigma"], d["emb.y"] = sig.numpy(), emb(sig).numpy()
        # attention block alone
        ta = _seed_module(R.TimeAttentionBlock(16, attn, init, init_zero, 12), 6)
        x = rnd(4, 2, 16, 12, 8)
        for k, v in _np(ta.state_dict()).items():
            d["ta.sd." + k] = v
        d["ta.x"], d["ta.y"] = x.numpy(), ta(x).numpy()
        # resnet block variants: (tag, dim, dim_out, num_dils, kernel, proj_place, attention, F, T)
        variants = [("rb_plain", 16, 16, 3, (5, 3), "before", False, 24, 16),
                    ("rb_attn", 8, 16, 2, (5, 3), "before", True, 12, 8),
                    ("rb_out", 16, 2, 1, (1, 1), "after", False, 10, 8),
                    ("rb_init", 2, 8, 1, (1, 1), "before", False, 8, 32),
                    ("rb_dec", 32, 16, 4, (5, 3), "before", True, 20, 4)]
        for si, (tag, dim, dout, nd, ks, pp, at, Fd, T) in enumerate(variants):
            rb = R.ResnetBlock(dim, dout, True, num_dils=nd, bias=False, kernel_size=ks, emb_dim=32,
                               proj_place=pp, init=init, init_zero=init_zero,
                               attention_dict=attn if at else None, Fdim=Fd)
            _seed_module(rb, 10 + si)
            x = rnd(20 + si, 2, dim, Fd, T)
            e = torch.relu(rnd(40 + si, 2, 32)) * 0.2
            for k, v in _np(rb.state_dict()).items():
                d[f"{tag}.sd.{k}"] = v
            d[f"{tag}.x"], d[f"{tag}.emb"], d[f"{tag}.y"] = x.numpy(), e.numpy(), rb(x, e).numpy()
    np.savez_compressed(os.path.join(out, "ops_small.npz"), **d)
    print("ops_small.npz", len(d), "arrays")


def gen_unet_small(out):
    import networks.unet_cqt_oct_with_projattention_adaLN_2 as R
    from audio_inpainting_diffusion_amd.config import small_args
    from audio_inpainting_diffusion_amd.init import seeded_normal
    cfgs = {
        "a": dict(num_octs=4, bins_per_oct=8, Ns=(8, 8, 16, 16), num_dils=(1, 2, 2, 3), attention=(0, 0, 1, 1, 1),
                  audio_len=4096, fs=22050, emb_dim=32),
        "b": dict(num_octs=7, bins_per_oct=8, Ns=(8, 8, 16, 16, 16, 32, 32), num_dils=(2, 3, 4, 5, 6, 7, 7),
                  attention=(0, 0, 0, 0, 1, 1, 1, 1), audio_len=16384, fs=22050, emb_dim=64),
        # the optional switches no shipped configuration turns on (unet...py:213-312,321,364,625-632,754-756)
        "c": dict(num_octs=4, bins_per_oct=8, Ns=(8, 8, 16, 16), num_dils=(1, 2, 2, 3), attention=(0, 0, 1, 1, 1),
                  audio_len=4096, fs=22050, emb_dim=32, use_fencoding=True, bias_qkv=True, use_rel_pos=True),
    }
    for tag, kw in cfgs.items():
        args = small_args(**kw)
        net = R.Unet_CQT_oct_with_attention(args, torch.device("cpu"))
        _seed_module(net, 100 + ord(tag))
        L = kw["audio_len"]
        x = torch.from_numpy(seeded_normal(77, ord(tag), 2 * L)).reshape(2, L) * 0.5
        cn = torch.tensor([[-0.9], [0.2]])
        with torch.no_grad():
            y = net(x, cn)
        # weights are NOT stored: both sides regenerate them with seeded_init_(seed, gate_scale=10, affine_scale=10)
        d = dict(x=x.numpy(), cnoise=cn.numpy(), y=y.numpy(), cfg=np.array(repr(kw)), seed=np.array(100 + ord(tag)),
                 keys=np.array(list(net.state_dict().keys())),
                 shapes=np.array([repr(tuple(v.shape)) for v in net.state_dict().values()]))
        np.savez_compressed(os.path.join(out, f"unet_small_{tag}.npz"), **d)
        print(f"unet_small_{tag}.npz  y rms {float(y.pow(2).mean().sqrt()):.4g}  x rms {float(x.pow(2).mean().sqrt()):.4g}")


def gen_training(out):
    """Three iterations of the reference's own Trainer.train_step + update_ema (training/trainer.py:253-304) with its EDM.loss_fn
    (diff_params/edm.py:166-193) and setup_optimizer's Adam (utils/setup.py:55-58) on the small reference network 'a' (its CQT = the oracle
    CQT, as in every U-Net fixture): per-iteration loss, sigma, and the network / EMA parameters after the last iteration."""
    import copy
    import diff_params.edm as E
    import networks.unet_cqt_oct_with_projattention_adaLN_2 as R
    import training.trainer as TR
    from audio_inpainting_diffusion_amd.config import small_args
    from audio_inpainting_diffusion_amd.init import seeded_normal
    kw = dict(num_octs=4, bins_per_oct=8, Ns=(8, 8, 16, 16), num_dils=(1, 2, 2, 3), attention=(0, 0, 1, 1, 1), audio_len=4096, fs=22050, emb_dim=32)
    args = small_args(**kw)
    hp = dict(lr=2e-3, lr_rampup_it=2, use_grad_clip=True, max_grad_norm=1.0, ema_rate=0.9, ema_rampup=8, batch=2, num_accumulation_rounds=1)
    for k, v in hp.items():
        setattr(args.exp, k, v)
    args.exp.optimizer = dict(beta1=0.9, beta2=0.999, eps=1e-8)
    args.logging = dict(log=False)
    net = R.Unet_CQT_oct_with_attention(args, torch.device("cpu"))
    _seed_module(net, 100 + ord("a"))
    tr = TR.Trainer.__new__(TR.Trainer)
    tr.args, tr.network, tr.ema, tr.it = args, net, copy.deepcopy(net), 0
    tr.diff_params = E.EDM(args)
    tr.optimizer = torch.optim.Adam(net.parameters(), lr=args.exp.lr, betas=(0.9, 0.999), eps=1e-8)        # = utils/setup.py::setup_optimizer
    B, L = hp["batch"], kw["audio_len"]
    d = dict(cfg=np.array(repr(kw)), hp=np.array(repr(hp)), seed=np.array(100 + ord("a")), n_it=np.array(3))
    losses = []
    for it in range(3):
        audio = torch.from_numpy(seeded_normal(41, it, B * L)).reshape(B, L) * 0.063
        tr.get_batch = lambda a_=audio: a_
        torch.manual_seed(500 + it)                      # loss_fn draws sigma (torch.rand) and the noise (torch.randn) from the global CPU generator
        tr.train_step()
        tr.update_ema()
        tr.it += 1
        torch.manual_seed(500 + it)
        with torch.no_grad():                            # the sigma that iteration drew (first draw after the seed)
            d[f"sigma.{it}"] = tr.diff_params.sample_ptrain_safe(B).numpy()
        d[f"lr.{it}"] = np.array(tr.optimizer.param_groups[0]["lr"])
    # the loss of each iteration, recomputed from the recorded print is fragile: re-run the three iterations' losses deterministically instead
    net2 = R.Unet_CQT_oct_with_attention(args, torch.device("cpu"))
    _seed_module(net2, 100 + ord("a"))
    tr2 = TR.Trainer.__new__(TR.Trainer)
    tr2.args, tr2.network, tr2.ema, tr2.it = args, net2, copy.deepcopy(net2), 0
    tr2.diff_params = tr.diff_params
    tr2.optimizer = torch.optim.Adam(net2.parameters(), lr=args.exp.lr, betas=(0.9, 0.999), eps=1e-8)
    for it in range(3):
        audio = torch.from_numpy(seeded_normal(41, it, B * L)).reshape(B, L) * 0.063
        torch.manual_seed(500 + it)
        with torch.no_grad():
            err, sg = tr2.diff_params.loss_fn(net2, audio)
        losses.append(float(err.mean()))
        tr2.get_batch = lambda a_=audio: a_
        torch.manual_seed(500 + it)
        tr2.train_step()
        tr2.update_ema()
        tr2.it += 1
    d["loss"] = np.array(losses)
    sd, esd, sd2 = net.state_dict(), tr.ema.state_dict(), net2.state_dict()
    assert all(torch.equal(sd[k], sd2[k]) for k in sd), "the reference loop is not deterministic"
    # Every trained tensor is pinned by four seeded random projections + its squared norm (fp64): <delta, probe> ~ N(0, |delta|^2), so the
    # projections of a wrong tensor differ by about its error norm; tensors of <= 4096 elements are stored whole as well.  (The full set is 5 MB.)
    from audio_inpainting_diffusion_amd.init import seeded_normal as _sn
    names = [k for k in sd if not (k.endswith("RFF_freq") or k.endswith("kernel") or not sd[k].dtype.is_floating_point)]
    d["names"] = np.array(names)
    for i, k in enumerate(names):
        n = sd[k].numel()
        probes = np.stack([_sn(9000 + j, i, n) for j in range(4)]).astype(np.float64)
        for tag, t in (("p", sd[k]), ("ema", esd[k])):
            v = t.double().reshape(-1).numpy()
            d[f"proj.{tag}.{k}"] = np.concatenate([probes @ v, [float(v @ v)]])
            if n <= 4096:
                d[f"{tag}.{k}"] = t.numpy()
    np.savez_compressed(os.path.join(out, "train_small.npz"), **d)
    print("train_small.npz: losses", losses, "lr", [float(d[f"lr.{i}"]) for i in range(3)])


def gen_edm(out):
    import diff_params.edm as E
    from audio_inpainting_diffusion_amd.config import make_args
    args = make_args()
    edm = E.EDM(args)
    tp = args.tester.diff_params
    edm.sigma_min, edm.sigma_max, edm.ro, edm.sigma_data = tp.sigma_min, tp.sigma_max, tp.ro, tp.sigma_data
    edm.Schurn, edm.Stmin, edm.Stmax, edm.Snoise = tp.Schurn, tp.Stmin, tp.Stmax, tp.Snoise
    d = {}
    for T in (35, 36, 70, 128):
        t = edm.create_schedule(T)
        d[f"t{T}"], d[f"gamma{T}"] = t.numpy(), edm.get_gamma(t).numpy()
    s = torch.tensor([[1.2], [0.5], [1e-3]])
    d["sigma"] = s.numpy()
    for n in ("cskip", "cout", "cin", "cnoise"):
        d[n] = getattr(edm, n)(s).numpy()
    np.savez_compressed(os.path.join(out, "edm_schedule.npz"), **d)
    print("edm_schedule.npz t35[:3]", d["t35"][:3], "gamma35", d["gamma35"][0], "gamma36", d["gamma36"][0])


class _ToyNet(torch.nn.Module):
    """Differentiable stand-in denoiser: fixed 9-tap conv times tanh(cnoise); owns an oracle CQT for apply_hpf_DC."""

    def __init__(self, L):
        super().__init__()
        from oracle.nsgt_cqt import OracleCQT
        self.CQTransform = OracleCQT(3, 8, "oct", ("kaiser", 1), 22050, L)
        k = torch.tensor([0.02, -0.05, 0.1, 0.25, 0.4, 0.25, 0.1, -0.05, 0.02])
        self.register_buffer("k", k.view(1, 1, 9))

    def forward(self, x, cnoise):
        y = torch.nn.functional.conv1d(x.unsqueeze(1), self.k, padding=4).squeeze(1)
        return y * torch.tanh(cnoise) + 0.1 * torch.sin(3.0 * x)


def gen_sampler(out):
    import diff_params.edm as E
    import testing.edm_sampler_inpainting as S
    from audio_inpainting_diffusion_amd.config import make_args
    from audio_inpainting_diffusion_amd.init import seeded_normal
    L, T = 2048, 6
    d = {"L": np.array(L), "T": np.array(T)}
    net = _ToyNet(L)
    cases = [("g_s0", 0.25, 1, True, 0), ("g_s1", 0.25, 1, True, 1), ("g_s2_nosmooth", 0.25, 1, False, 2),
             ("r_s0", 0.0, 1, True, 0), ("r_b2_s1", 0.0, 2, True, 1), ("r_b2_nosmooth", 0.0, 2, False, 2)]
    for tag, xi, B, smooth, seed in cases:
        args = make_args(audio_len=L, T=T, xi=xi)
        args.tester.data_consistency.smooth = smooth
        args.tester.data_consistency.hann_size = 20
        edm = E.EDM(args)
        smp = S.Sampler(model=net, diff_params=edm, args=args, rid=False)
        y = torch.from_numpy(seeded_normal(5, seed, B * L)).reshape(B, L) * 0.063
        mask = torch.ones(1, L)
        mask[:, 900:1150] = 0
        torch.manual_seed(seed)
        x = smp.predict_inpainting(y * mask, mask)
        d[f"{tag}.y"], d[f"{tag}.mask"], d[f"{tag}.out"] = (y * mask).numpy(), mask.numpy(), x.numpy()
        d[f"{tag}.meta"] = np.array([xi, B, int(smooth), seed], dtype=np.float64)
        print("sampler", tag, "out rms", float(x.pow(2).mean().sqrt()))
    np.savez_compressed(os.path.join(out, "sampler_toy.npz"), **d)


def gen_uncond(out):
    """predict_unconditional (edm_sampler_inpainting.py:155-162, :115-125) on the toy denoiser: B = 1 and B = 2, with and without the DC/Nyquist projector."""
    import diff_params.edm as E
    import testing.edm_sampler_inpainting as S
    from audio_inpainting_diffusion_amd.config import make_args
    L, T = 2048, 5
    d = {"L": np.array(L), "T": np.array(T)}
    net = _ToyNet(L)
    for tag, B, hpf, seed in (("u_b1", 1, True, 3), ("u_b2", 2, True, 4), ("u_b2_nohpf", 2, False, 5)):
        args = make_args(audio_len=L, T=T, xi=0.25)
        args.tester.filter_out_cqt_DC_Nyq = hpf
        smp = S.Sampler(model=net, diff_params=E.EDM(args), args=args, rid=False)
        torch.manual_seed(seed)
        x = smp.predict_unconditional((B, L), torch.device("cpu"))
        d[f"{tag}.out"], d[f"{tag}.meta"] = x.numpy(), np.array([B, int(hpf), seed], dtype=np.float64)
        print("uncond", tag, "out rms", float(x.pow(2).mean().sqrt()))
    np.savez_compressed(os.path.join(out, "sampler_uncond.npz"), **d)


def resample_degradations(k):
    """The two degradation callables of sampler_resample.npz (tests rebuild them from the stored FIR taps `k`): a low-pass + decimate-by-2
    (linear, observations HALF as long as the signal -- the bandwidth-extension shape of the reference's generic sampler) and a soft clipper
    (non-linear: the guidance needs the Jacobian-transpose product at x_hat, not an adjoint)."""
    kk = k.view(1, 1, -1)
    return {"lpf_dec2": lambda x: torch.nn.functional.conv1d(x.unsqueeze(1), kk.to(x.device), stride=2, padding=kk.shape[-1] // 2).squeeze(1),
            "softclip": lambda x: 0.05 * torch.tanh(x / 0.05)}


def gen_resample(out):
    """predict_resample (edm_sampler_inpainting.py:164-173): the reference's Sampler + EDM around the toy denoiser, reconstruction guidance through
    a GENERIC degradation lambda, no projection (data_consistency.use = False: the reference defines proj_convex_set for inpainting only)."""
    import diff_params.edm as E
    import testing.edm_sampler_inpainting as S
    from audio_inpainting_diffusion_amd.config import make_args
    from audio_inpainting_diffusion_amd.init import seeded_normal
    L, T = 2048, 5
    k = torch.tensor([-0.01, 0.0, 0.03, 0.0, -0.07, 0.0, 0.3, 0.5, 0.3, 0.0, -0.07, 0.0, 0.03, 0.0, -0.01])
    d = {"L": np.array(L), "T": np.array(T), "k": k.numpy()}
    net = _ToyNet(L)
    degs = resample_degradations(k)
    for tag, name, B, norm, seed in (("lpf_s0", "lpf_dec2", 1, 2, 0), ("lpf_s1_l1", "lpf_dec2", 1, 1, 1), ("clip_s2", "softclip", 1, 2, 2), ("clip_s3_sl1", "softclip", 1, "smoothl1", 3)):
        args = make_args(audio_len=L, T=T, xi=0.25)
        args.tester.data_consistency.use = False
        args.tester.posterior_sampling.norm = norm
        args.tester.posterior_sampling.smoothl1_beta = 0.02
        smp = S.Sampler(model=net, diff_params=E.EDM(args), args=args, rid=False)
        clean = torch.from_numpy(seeded_normal(8, seed, B * L)).reshape(B, L) * 0.063
        with torch.no_grad():
            y = degs[name](clean)
        torch.manual_seed(seed)
        x = smp.predict_resample(y, (B, L), degs[name])
        d[f"{tag}.y"], d[f"{tag}.out"] = y.numpy(), x.numpy()
        d[f"{tag}.meta"] = np.array([B, {2: 2.0, 1: 1.0, "smoothl1": 3.0}[norm], seed, 0 if name == "lpf_dec2" else 1], dtype=np.float64)   # (B = 1: the reference's guidance needs a scalar norm)
        print("resample", tag, "y", tuple(y.shape), "out rms", float(x.pow(2).mean().sqrt()))
    np.savez_compressed(os.path.join(out, "sampler_resample.npz"), **d)


def gen_norms(out):
    """tester.posterior_sampling.norm variants of the reference's guidance (edm_sampler_inpainting.py:72-75): 1 (L1) and "smoothl1"
    (reduction 'sum', smoothl1_beta), B = 1, on the toy denoiser."""
    import diff_params.edm as E
    import testing.edm_sampler_inpainting as S
    from audio_inpainting_diffusion_amd.config import make_args
    from audio_inpainting_diffusion_amd.init import seeded_normal
    L, T = 2048, 5
    d = {"L": np.array(L), "T": np.array(T)}
    net = _ToyNet(L)
    for tag, norm, beta, seed in (("l1", 1, 1.0, 0), ("sl1_small", "smoothl1", 0.01, 1), ("sl1_large", "smoothl1", 0.2, 2)):
        args = make_args(audio_len=L, T=T, xi=0.25)
        args.tester.posterior_sampling.norm = norm
        args.tester.posterior_sampling.smoothl1_beta = beta
        args.tester.data_consistency.hann_size = 20
        smp = S.Sampler(model=net, diff_params=E.EDM(args), args=args, rid=False)
        y = torch.from_numpy(seeded_normal(6, seed, L)).reshape(1, L) * 0.063
        mask = torch.ones(1, L)
        mask[:, 700:1000] = 0
        torch.manual_seed(seed)
        x = smp.predict_inpainting(y * mask, mask)
        d[f"{tag}.y"], d[f"{tag}.mask"], d[f"{tag}.out"] = (y * mask).numpy(), mask.numpy(), x.numpy()
        d[f"{tag}.meta"] = np.array([1.0 if norm == 1 else 3.0, beta, seed], dtype=np.float64)
        print("norms", tag, "out rms", float(x.pow(2).mean().sqrt()))
    np.savez_compressed(os.path.join(out, "sampler_norms.npz"), **d)


def gen_aweighting(out):
    """The A-weighting error filter of the reference's loss (utils/training_utils.py:55-137 FIRFilter("aw"), used at diff_params/edm.py:33-34, :189-190):
    taps for the three shipped sampling rates and the filter applied to a seeded signal."""
    import utils.training_utils as TU
    from audio_inpainting_diffusion_amd.init import seeded_normal
    d = {}
    x = torch.from_numpy(seeded_normal(8, 0, 2 * 4096)).reshape(2, 4096)
    d["x"] = x.numpy()
    for fs in (22050, 44100, 16000):
        f = TU.FIRFilter(filter_type="aw", fs=fs, ntaps=101)
        d[f"taps{fs}"] = f.fir.weight.data.reshape(-1).numpy()
        d[f"y{fs}"] = f(x).numpy()
    np.savez_compressed(os.path.join(out, "aweighting.npz"), **d)
    print("aweighting taps centre", d["taps22050"][50])


def gen_dc(out):
    """data_consistency.type variants of the reference sampler (edm_sampler_inpainting.py:22-24, :100, :141-147, :252):
    'end' projects only after the loop on the guided branch, but the replacement branch (xi = 0) projects at EVERY
    evaluation whatever the type."""
    import diff_params.edm as E
    import testing.edm_sampler_inpainting as S
    from audio_inpainting_diffusion_amd.config import make_args
    from audio_inpainting_diffusion_amd.init import seeded_normal
    L, T = 2048, 5
    d = {"L": np.array(L), "T": np.array(T)}
    net = _ToyNet(L)
    for tag, xi, dctype, seed in (("g_end", 0.25, "end", 0), ("r_end", 0.0, "end", 1), ("g_always", 0.25, "always", 2)):
        args = make_args(audio_len=L, T=T, xi=xi)
        args.tester.data_consistency.type = dctype
        args.tester.data_consistency.hann_size = 20
        smp = S.Sampler(model=net, diff_params=E.EDM(args), args=args, rid=False)
        y = torch.from_numpy(seeded_normal(4, seed, L)).reshape(1, L) * 0.063
        mask = torch.ones(1, L)
        mask[:, 800:1100] = 0
        torch.manual_seed(seed)
        x = smp.predict_inpainting(y * mask, mask)
        d[f"{tag}.y"], d[f"{tag}.mask"], d[f"{tag}.out"] = (y * mask).numpy(), mask.numpy(), x.numpy()
        d[f"{tag}.meta"] = np.array([xi, seed, 1.0 if dctype == "end" else 0.0], dtype=np.float64)
        print("dc", tag, "out rms", float(x.pow(2).mean().sqrt()))
    np.savez_compressed(os.path.join(out, "sampler_dc.npz"), **d)


def gen_rid(out):
    """rid=True: the reference sampler's per-step debug buffers (edm_sampler_inpainting.py:185-191, :217-226, :255-260)."""
    import diff_params.edm as E
    import testing.edm_sampler_inpainting as S
    from audio_inpainting_diffusion_amd.config import make_args
    from audio_inpainting_diffusion_amd.init import seeded_normal
    L, T = 2048, 4
    net = _ToyNet(L)
    args = make_args(audio_len=L, T=T, xi=0.25)
    args.tester.data_consistency.hann_size = 20
    smp = S.Sampler(model=net, diff_params=E.EDM(args), args=args, rid=True)
    y = torch.from_numpy(seeded_normal(8, 0, L)).reshape(1, L) * 0.063
    mask = torch.ones(1, L)
    mask[:, 700:1000] = 0
    torch.manual_seed(3)
    res = smp.predict_inpainting(y * mask, mask)
    names = ("out", "denoised", "grads", "grad_update", "pocs", "xt", "xt2", "t")
    d = {n: r.numpy() for n, r in zip(names, res)}
    d.update(y=(y * mask).numpy(), mask=mask.numpy(), L=np.array(L), T=np.array(T))
    np.savez_compressed(os.path.join(out, "sampler_rid.npz"), **d)
    print("rid", {n: tuple(r.shape) for n, r in zip(names, res)})


def gen_spectral(out):
    """Spectrogram inpainting: the reference's apply_spectral_mask operator and full sampler trajectories."""
    import diff_params.edm as E
    import testing.edm_sampler_inpainting as S
    from audio_inpainting_diffusion_amd.config import make_args
    from audio_inpainting_diffusion_amd.init import seeded_normal
    L, T = 2000, 5
    n_fft, hop = 128, 32
    d = {"L": np.array(L), "T": np.array(T), "stft": np.array([n_fft, hop, n_fft])}
    net = _ToyNet(L)
    Lp = L + (n_fft - L % n_fft)
    mask = torch.ones(n_fft // 2 + 1, 1 + Lp // hop)
    mask[6:30, 20:41] = 0
    d["mask"] = mask.numpy()
    for tag, xi, seed in (("sg_s0", 0.25, 0), ("sg_s1", 0.25, 1), ("sr_s0", 0.0, 0)):
        args = make_args(audio_len=L, T=T, xi=xi)
        st = args.tester.spectrogram_inpainting.stft
        st.n_fft, st.hop_length, st.win_length = n_fft, hop, n_fft
        smp = S.Sampler(model=net, diff_params=E.EDM(args), args=args, rid=False)
        y = torch.from_numpy(seeded_normal(6, seed, L)).reshape(1, L) * 0.063
        smp.mask = mask
        ym = smp.apply_spectral_mask(y)
        torch.manual_seed(seed)
        x = smp.predict_spectrogram_inpainting(ym, mask)
        d[f"{tag}.y0"], d[f"{tag}.y"], d[f"{tag}.out"] = y.numpy(), ym.numpy(), x.numpy()
        d[f"{tag}.meta"] = np.array([xi, seed], dtype=np.float64)
        print("spectral", tag, "masked rms", float(ym.pow(2).mean().sqrt()), "out rms", float(x.pow(2).mean().sqrt()))
    np.savez_compressed(os.path.join(out, "sampler_spectral.npz"), **d)


class _GuidanceTap:
    """Records what the reference's get_score_rec_guidance (edm_sampler_inpainting.py:57-113) computes at every evaluation WITHOUT re-stating
    it: the sampler is built with rid=True, whose return tuple already carries the denoised estimate before the gradient step; `norm` and
    `rec_grads` are seen by wrapping torch.autograd.grad (outputs = norm, result = rec_grads) while the sampler runs."""

    def __init__(self, smp):
        self.smp, self.rows = smp, []
        self._inner = smp.get_score_rec_guidance
        smp.get_score_rec_guidance = self._call

    def _call(self, x, y, t_i, degradation):
        real = torch.autograd.grad
        seen = {}

        def grad(outputs, inputs, *a, **k):
            r = real(outputs, inputs, *a, **k)
            seen["norm"], seen["g"] = outputs.detach().clone(), r[0].detach().clone()
            return r
        torch.autograd.grad = grad
        try:
            x_in = x.detach().clone()
            res = self._inner(x, y, t_i, degradation)
        finally:
            torch.autograd.grad = real
        self.rows.append(dict(x=x_in.numpy(), t=np.array(float(t_i)), x_hat=res[1].numpy(), rec_grads=seen["g"].numpy(),
                              norm=seen["norm"].reshape(-1).numpy(), x_hat_proj=res[4].detach().numpy()))
        return res


def gen_guided(out):
    """The guided chain pinned to the reference DIRECTLY (VERDICT r3 missing-3): the reference's Sampler + EDM driving the reference's own
    Unet_CQT_oct_with_attention (small configs a and c, O(1) gates, CQT = the oracle CQT as in every U-Net fixture), xi = 0.25, T = 3, B = 1,
    seeds {0, 1}: input state, t, x_hat (after apply_hpf_DC, before the gradient step), rec_grads and norm of EVERY evaluation, and the output."""
    import diff_params.edm as E
    import networks.unet_cqt_oct_with_projattention_adaLN_2 as R
    import testing.edm_sampler_inpainting as S
    from audio_inpainting_diffusion_amd.config import small_args
    from audio_inpainting_diffusion_amd.init import seeded_normal
    cfgs = {
        "a": dict(num_octs=4, bins_per_oct=8, Ns=(8, 8, 16, 16), num_dils=(1, 2, 2, 3), attention=(0, 0, 1, 1, 1),
                  audio_len=4096, fs=22050, emb_dim=32),
        "c": dict(num_octs=4, bins_per_oct=8, Ns=(8, 8, 16, 16), num_dils=(1, 2, 2, 3), attention=(0, 0, 1, 1, 1),
                  audio_len=4096, fs=22050, emb_dim=32, use_fencoding=True, bias_qkv=True, use_rel_pos=True),
    }
    d = {}
    for tag, kw in cfgs.items():
        L = kw["audio_len"]
        args = small_args(**kw, T=3, xi=0.25)
        args.tester.data_consistency.hann_size = 20
        net = R.Unet_CQT_oct_with_attention(args, torch.device("cpu"))
        _seed_module(net, 100 + ord(tag))
        d[f"{tag}.cfg"], d[f"{tag}.seed"] = np.array(repr(kw)), np.array(100 + ord(tag))
        for seed in (0, 1):
            smp = S.Sampler(model=net, diff_params=E.EDM(args), args=args, rid=True)
            tap = _GuidanceTap(smp)
            y = torch.from_numpy(seeded_normal(12, seed, L)).reshape(1, L) * 0.063
            mask = torch.ones(1, L)
            mask[:, 1800:2300] = 0
            torch.manual_seed(seed)
            res = smp.predict_inpainting(y * mask, mask)
            k = f"{tag}.s{seed}"
            d[k + ".y"], d[k + ".mask"], d[k + ".out"] = (y * mask).numpy(), mask.numpy(), res[0].detach().numpy()
            d[k + ".n_eval"] = np.array(len(tap.rows))
            for i, r in enumerate(tap.rows):
                for name, v in r.items():
                    d[f"{k}.e{i}.{name}"] = v
            print("guided", k, "evaluations", len(tap.rows), "norms", [float(r["norm"][0]) for r in tap.rows])
    np.savez_compressed(os.path.join(out, "sampler_guided_unet.npz"), **d)

def gen_guided_traj(out):
    """A WHOLE trajectory of the reference chain: the reference's Sampler + EDM driving the reference's own U-Net (small config
    a, O(1) gates), xi = 0.25, T = 10, churn only inside a window (tester.diff_params Stmin = 2e-3 < t < Stmax = 0.3, Schurn = 4: the first steps deterministic, the middle ones
    stochastic, the tail deterministic again -- both branches of edm_sampler_inpainting.py:204), Heun steps and the final Euler step onto t = 0
    (:236-251).  Stored: schedule, gamma, the state before / after every step (rid_xt / rid_xt2), the denoised estimate of every first evaluation,
    the input state + x_hat + rec_grads + norm of EVERY evaluation (for teacher-forced checks), and the output."""
    import diff_params.edm as E
    import networks.unet_cqt_oct_with_projattention_adaLN_2 as R
    import testing.edm_sampler_inpainting as S
    from audio_inpainting_diffusion_amd.config import small_args
    from audio_inpainting_diffusion_amd.init import seeded_normal
    kw = dict(num_octs=4, bins_per_oct=8, Ns=(8, 8, 16, 16), num_dils=(1, 2, 2, 3), attention=(0, 0, 1, 1, 1), audio_len=4096, fs=22050, emb_dim=32)
    T, seed, L = 10, 3, kw["audio_len"]
    args = small_args(**kw, T=T, xi=0.25)
    args.tester.data_consistency.hann_size = 20
    dp = args.tester.diff_params                           # (same_as_training = False: the Sampler overrides the EDM object with these, edm_sampler_inpainting.py:26-36)
    dp.Stmin, dp.Stmax, dp.Schurn = 2e-3, 0.3, 4.0
    net = R.Unet_CQT_oct_with_attention(args, torch.device("cpu"))
    _seed_module(net, 100 + ord("a"))
    edm = E.EDM(args)
    smp = S.Sampler(model=net, diff_params=edm, args=args, rid=True)
    tap = _GuidanceTap(smp)
    y = torch.from_numpy(seeded_normal(12, seed, L)).reshape(1, L) * 0.063
    mask = torch.ones(1, L)
    mask[:, 1700:2400] = 0
    torch.manual_seed(seed)
    res = smp.predict_inpainting(y * mask, mask)
    t = res[7]
    gamma = edm.get_gamma(t)
    assert len(tap.rows) == 2 * T - 1 and float(t[-1]) == 0.0
    assert int((gamma[:T] == 0).sum()) >= 3 and int((gamma[:T] > 0).sum()) >= 3 and float(gamma[0]) == 0.0 and float(gamma[T - 1]) == 0.0
    d = dict(cfg=np.array(repr(kw)), seed=np.array(100 + ord("a")), T=np.array(T), noise_seed=np.array(seed), Stmin=np.array(2e-3), Stmax=np.array(0.3), Schurn=np.array(4.0),
             y=(y * mask).numpy(), mask=mask.numpy(), out=res[0].numpy(), t=t.numpy(), gamma=gamma.numpy(),
             denoised=res[1].numpy(), xt=res[5].numpy(), xt2=res[6].numpy(), n_eval=np.array(len(tap.rows)))
    for i, r in enumerate(tap.rows):
        for name in ("x", "t", "x_hat", "rec_grads", "norm"):
            d[f"e{i}.{name}"] = r[name]
    print("guided_traj: T", T, "gamma", [round(float(g), 4) for g in gamma[:T]], "evaluations", len(tap.rows), "|out|", float(res[0].norm()))
    np.savez_compressed(os.path.join(out, "sampler_guided_traj.npz"), **d)


def _proj(t, stream, n=8):
    """seeded random projections + squared norm (fp64) of a tensor: <delta, probe> ~ N(0, |delta|^2) (the train_small.npz trick)"""
    from audio_inpainting_diffusion_amd.init import seeded_normal
    v = t.detach().double().reshape(-1).numpy()
    probes = np.stack([seeded_normal(7000 + j, stream, v.size) for j in range(n)]).astype(np.float64)
    return np.concatenate([probes @ v, [float(v @ v)]])


def gen_full_guided(out):
    """ONE guided evaluation of the reference's Sampler + EDM + full-size cfg-A Unet (186 M parameters; ~1 min and ~18 GB here): x_hat,
    rec_grads (strided samples + seeded projections + squared norm) and norm.  Inputs are regenerated from seeds on both sides."""
    import diff_params.edm as E
    import networks.unet_cqt_oct_with_projattention_adaLN_2 as R
    import testing.edm_sampler_inpainting as S
    from audio_inpainting_diffusion_amd.config import make_args
    from audio_inpainting_diffusion_amd.init import seeded_normal
    from audio_inpainting_diffusion_amd.masks import long_gap_mask
    args = make_args("maestro22k")
    t0 = time.time()
    net = R.Unet_CQT_oct_with_attention(args, torch.device("cpu"))
    _seed_module(net, 0)
    L = args.exp.audio_len
    smp = S.Sampler(model=net, diff_params=E.EDM(args), args=args, rid=True)
    tap = _GuidanceTap(smp)
    x = torch.from_numpy(seeded_normal(21, 0, L)).reshape(1, L) * 0.3
    y = torch.from_numpy(seeded_normal(22, 0, L)).reshape(1, L) * 0.063
    mask = long_gap_mask(L, 22050, 300)
    smp.data_consistency = False                          # (the projection after the gradient step is not part of this fixture)
    t_i = torch.tensor(0.4)
    smp.get_score_rec_guidance(x, y * mask, t_i, lambda v: v * mask)
    r = tap.rows[0]
    xh, g = torch.from_numpy(r["x_hat"]), torch.from_numpy(r["rec_grads"])
    print("full guided evaluation done in %.1fs: norm %.6g, |x_hat| %.4g, |g| %.4g" % (time.time() - t0, float(r["norm"][0]), float(xh.norm()), float(g.norm())))
    np.savez_compressed(os.path.join(out, "unet_full_cfgA_guided.npz"), t=np.array(0.4), norm=r["norm"],
                        x_hat_proj=_proj(xh, 1), rec_grads_proj=_proj(g, 2), x_hat_s=r["x_hat"][:, ::97].astype(np.float32), rec_grads_s=r["rec_grads"][:, ::97].astype(np.float32),
                        recipe=np.array("weights: seeded_init_(seed=0, gate_scale=10, affine_scale=10); x = 0.3*seeded_normal(21,0,L); y = 0.063*seeded_normal(22,0,L); "
                                        "mask = long_gap_mask(L, 22050, 300); t = 0.4; filter_out_cqt_DC_Nyq as shipped; projections: 8 x seeded_normal(7000+j, stream) + squared norm, "
                                        "streams 1 (x_hat) / 2 (rec_grads); *_s = every 97th sample"))


def gen_full(out):
    import networks.unet_cqt_oct_with_projattention_adaLN_2 as R
    from audio_inpainting_diffusion_amd.config import make_args
    from audio_inpainting_diffusion_amd.init import seeded_normal
    args = make_args("maestro22k")
    t0 = time.time()
    net = R.Unet_CQT_oct_with_attention(args, torch.device("cpu"))
    _seed_module(net, 0)
    L = args.exp.audio_len
    x = torch.from_numpy(seeded_normal(2024, 0, L)).reshape(1, L) * 0.5
    cn = torch.tensor([[-0.35]])
    with torch.no_grad():
        y = net(x, cn)
    print("full cfgA forward done in %.1fs, y rms %.4g" % (time.time() - t0, float(y.pow(2).mean().sqrt())))
    np.savez_compressed(os.path.join(out, "unet_full_cfgA.npz"), y=y.numpy().astype(np.float32), cnoise=cn.numpy(),
                        recipe=np.array("weights: seeded_init_(seed=0, gate_scale=10, affine_scale=10); "
                                        "x = 0.5*seeded_normal(2024, 0, L)"))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    _setup_imports()
    torch.set_grad_enabled(True)
    todo = a.only.split(",") if a.only else ["ops", "unet", "edm", "sampler", "spectral", "rid", "dc", "training", "uncond", "norms", "aweighting"] + (["full"] if a.full else [])
    if "ops" in todo: gen_ops(HERE)
    if "unet" in todo: gen_unet_small(HERE)
    if "edm" in todo: gen_edm(HERE)
    if "sampler" in todo: gen_sampler(HERE)
    if "spectral" in todo: gen_spectral(HERE)
    if "rid" in todo: gen_rid(HERE)
    if "dc" in todo: gen_dc(HERE)
    if "training" in todo: gen_training(HERE)
    if "uncond" in todo: gen_uncond(HERE)
    if "norms" in todo: gen_norms(HERE)
    if "aweighting" in todo: gen_aweighting(HERE)
    if "full" in todo: gen_full(HERE)
    if "resample" in todo: gen_resample(HERE)
    if "guided" in todo: gen_guided(HERE)
    if "guided_traj" in todo: gen_guided_traj(HERE)
    if "full_guided" in todo: gen_full_guided(HERE)
