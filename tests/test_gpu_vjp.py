"""GPU parity of the input-VJP (reconstruction-guidance branch, reference testing/edm_sampler_inpainting.py:57-105):
the hand-written backward kernels against torch.autograd over the CPU oracle."""
import ast
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def L():
    from audio_inpainting_diffusion_amd import _lib
    _lib.lib()
    return _lib


def _rand(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("case", [(2, 16, 24, 16, 5, 3, 2), (1, 64, 20, 256, 5, 3, 8), (2, 32, 12, 8, 1, 1, 1), (2, 128, 16, 32, 5, 3, 4),
                                  (2, 64, 16, 64, 1, 1, 1)])
def test_fused_step_vjp(L, case):
    """One ResnetBlock step  y = (x + conv(gelu(norm(x)*(1+g))) * s)/sqrt2  (unet...py:472-482): dgrad conv with
    dGELU epilogue + group-dot + normalisation backward vs torch.autograd."""
    from oracle.unet import group_std_norm
    B, N, Fd, T, KH, KW, dil = case
    x = _rand(B, N, Fd, T, seed=1, scale=2.0) + 0.3
    w = _rand(N, N, KH, KW, seed=2, scale=1 / math.sqrt(N * KH * KW))
    gamma, g, s = 1 + 0.3 * _rand(N, seed=3), 0.4 * _rand(B, N, seed=4), _rand(B, N, seed=5)
    gy = _rand(B, N, Fd, T, seed=6)
    xr = x.clone().requires_grad_()
    h = group_std_norm(xr, gamma.view(1, -1, 1, 1)) * (1 + g)[:, :, None, None]
    y = (xr + F.conv2d(F.gelu(h), w, padding="same", dilation=(dil, 1) if KH > 1 else 1) * s[:, :, None, None]) / math.sqrt(2)
    ref = torch.autograd.grad((y * gy).sum(), xr)[0]

    xd, gyd, gd_, gam, gm, sd = (t.to(DEV).contiguous() for t in (x, gy, torch.zeros_like(x), gamma, g, s))
    scale, stats = torch.empty(B, N, device=DEV), torch.empty(B, 8, 2, device=DEV)
    ws = torch.empty(B * 8 * L.AID_STATS_SPLIT * 2, device=DEV, dtype=torch.float64)
    L.call("aid_group_stats", L.GroupStatsParams(L.view4(xd), B, N, Fd, T, 8, gam.data_ptr(), gm.data_ptr(), gm.stride(0), 1e-7,
                                                 scale.data_ptr(), stats.data_ptr(), ws.data_ptr()))
    wT = L.pack_conv_weight(w.to(DEV), transpose=True)
    p = L.Conv2dParams()
    p.x, p.y, p.res, p.aux = L.view4(gyd), L.view4(gd_), L.view4(None), L.view4(xd)
    p.wp = wT.data_ptr()
    p.in_scale, p.in_scale_ld = sd.data_ptr(), sd.stride(0)
    p.out_scale, p.out_scale_ld = scale.data_ptr(), scale.stride(0)
    p.aux_scale, p.aux_scale_ld = scale.data_ptr(), scale.stride(0)
    p.B, p.Cin, p.Cout, p.F, p.T = B, N, N, Fd, T
    p.Cin_pad, p.Cout_pad = wT.shape[1], wT.shape[2]
    p.KH, p.KW, p.dilF, p.act, p.epi = KH, KW, dil, 0, 1
    p.alpha, p.res_scale = 1 / math.sqrt(2), 1.0
    L.call("aid_conv2d", p)
    L.call("aid_group_dot", L.GroupDotParams(L.view4(gd_), L.view4(xd), B, N, Fd, T, 8, ws.data_ptr()))
    out = torch.full_like(xd, 0.25)      # accumulate on top of an existing gradient
    L.call("aid_norm_bwd", L.NormBwdParams(L.view4(gd_), L.view4(xd), L.view4(gyd), L.view4(out), B, N, Fd, T, 8, stats.data_ptr(),
                                           ws.data_ptr(), 1e-7, 1 / math.sqrt(2), 1))
    assert rel_l2(out.cpu() - 0.25, ref) < 2e-5


@pytest.mark.parametrize("shape", [(2, 8, 12, 8), (1, 8, 40, 32), (1, 8, 320, 128), (1, 4, 56, 64)])
def test_attention_bwd(L, shape):
    B, H, Fd, T = shape
    qk = _rand(B, H * 2 * Fd, T, seed=30, scale=1.5).requires_grad_()
    v = _rand(B, H, Fd, T, seed=31).requires_grad_()
    go = _rand(B, H, Fd, T, seed=32)
    q4 = qk.reshape(B, H, 2 * Fd, T).permute(0, 1, 3, 2)
    sim = torch.einsum("bhnd,bhmd->bhnm", q4[..., :Fd], q4[..., Fd:]) * (float(Fd) ** -0.5)
    out = torch.einsum("bhnm,bhmd->bhnd", sim.softmax(-1), v.permute(0, 1, 3, 2)).permute(0, 1, 3, 2)
    gq_ref, gv_ref = torch.autograd.grad((out * go).sum(), (qk, v))
    qd, vd, god = qk.detach().to(DEV), v.detach().to(DEV), go.to(DEV)
    o = torch.empty(B, H, Fd, T, device=DEV)
    probs = torch.empty(B, H, T, T, device=DEV)
    sc = float(Fd) ** -0.5
    L.call("aid_time_attention", L.AttentionParams(qd.data_ptr(), vd.data_ptr(), o.data_ptr(), probs.data_ptr(), B, H, Fd, T, sc))
    gq = torch.empty_like(qd)
    gv = torch.full_like(vd, 0.5)
    ws = torch.empty(B, H, T, T, device=DEV)
    L.call("aid_time_attention_bwd", L.AttentionBwdParams(qd.data_ptr(), vd.data_ptr(), probs.data_ptr(), god.data_ptr(), gq.data_ptr(),
                                                          gv.data_ptr(), B, H, Fd, T, sc, 1, ws.data_ptr()))
    assert rel_l2(gq.cpu(), gq_ref) < 2e-5
    assert rel_l2(gv.cpu() - 0.5, gv_ref) < 2e-5


def test_cqt_adjoints(L):
    """<A x, c> == <x, A^H c> for analysis and synthesis (real inner products), via the device kernels."""
    from audio_inpainting_diffusion_amd.cqt import CQTransform
    no, bpo, fs, Ls = 4, 8, 22050, 4096
    tr = CQTransform(no, bpo, "oct", ("kaiser", 1.0), fs, Ls, device=DEV)
    tab = tr._tables(torch.device(DEV))
    x = _rand(2, Ls, seed=1).to(DEV)
    outs = tr.alloc_octaves(2, DEV)
    tr.analysis(x, outs)
    gc = [_rand(*o.shape, seed=10 + i).to(DEV) for i, o in enumerate(outs)]
    lhs = sum(float((a.double() * b.double()).sum()) for a, b in zip(outs, gc))
    gx = torch.fft.irfft(tr.analysis_adjoint(gc), n=Ls)
    rhs = float((x.double() * gx.double()).sum())
    assert abs(lhs - rhs) < 1e-5 * max(1.0, abs(lhs))
    # synthesis: c -> y = irfft(gather(...)) ;  adjoint: g_y -> g_c
    c = [_rand(*o.shape, seed=20 + i).to(DEV) for i, o in enumerate(outs)]
    y = torch.fft.irfft(tr.synthesis_spectrum(c), n=Ls)
    gy = _rand(2, Ls, seed=30).to(DEV)
    gco = [torch.empty_like(o) for o in outs]
    tr.synthesis_adjoint(tr.spectrum_scale(torch.fft.rfft(gy), tab["w_over_L"]), gco)
    lhs = float((y.double() * gy.double()).sum())
    rhs = sum(float((a.double() * b.double()).sum()) for a, b in zip(c, gco))
    assert abs(lhs - rhs) < 1e-5 * max(1.0, abs(lhs))


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


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_network_vjp_vs_oracle_autograd(tag):
    """torch.autograd.grad through the drop-in network (autograd bridge -> hand-written VJP plan) vs the oracle."""
    net, orc, z, kw, _ = _setup(tag)
    x, cn = torch.from_numpy(z["x"]), torch.from_numpy(z["cnoise"])
    g = _rand(*x.shape, seed=3)
    xr = x.clone().requires_grad_()
    ref = torch.autograd.grad((orc(xr, cn) * g).sum(), xr)[0]
    xd = x.to(DEV).requires_grad_()
    y = net(xd, cn.to(DEV))
    got = torch.autograd.grad((y * g.to(DEV)).sum(), xd)[0]
    e = rel_l2(got.cpu(), ref)
    print(f"network input-VJP ({tag}): rel-L2 vs oracle autograd = {e:.3e}")
    assert e < 1e-4


def test_guided_evaluation_and_sampler_vs_oracle():
    from audio_inpainting_diffusion_amd.edm import EDM
    from audio_inpainting_diffusion_amd.sampler import Sampler
    from oracle.edm import OracleEDM
    from oracle.sampler import OracleSampler
    net, orc, z, kw, args = _setup("a")
    Ls = kw["audio_len"]
    args.tester.T, args.tester.posterior_sampling.xi = 3, 0.25
    args.tester.data_consistency.hann_size = 20
    y = torch.from_numpy(z["x"]) * 0.126
    mask = torch.ones(1, Ls)
    mask[:, 1800:2300] = 0
    smp = Sampler(model=net, diff_params=EDM(args), args=args)
    smp.seeds, smp.trace = [5, 6], []
    out = smp.predict_inpainting((y * mask).to(DEV), mask.to(DEV))
    osmp = OracleSampler(orc, OracleEDM(), T=3, xi=0.25, hann_size=20, audio_len=Ls)
    ref = osmp.predict_inpainting(y * mask, mask, seeds=[5, 6], record=True)
    errs = [rel_l2(a.cpu(), b) for a, b in zip(smp.trace, osmp.trace)]
    print("guided per-evaluation x_hat rel-L2:", ["%.2e" % e for e in errs], " final:", "%.2e" % rel_l2(out.cpu(), ref))
    assert len(errs) == 5 and errs[0] < 1e-4 and max(errs) < 5e-4


def test_full_size_guided_evaluation_vs_oracle_autograd():
    """Full cfg-A network (186 M parameters, L=184184), B=1: x_hat and the reconstruction-guidance gradient of
    network.denoise_guided against torch.autograd over the CPU oracle (~60 s of CPU, ~18 GB host RAM)."""
    from audio_inpainting_diffusion_amd.config import make_args
    from audio_inpainting_diffusion_amd.init import seeded_init_, seeded_normal
    from audio_inpainting_diffusion_amd.masks import long_gap_mask
    from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
    from oracle.edm import OracleEDM
    from oracle.nsgt_cqt import OracleCQT
    from oracle.unet import OracleUnet
    args = make_args("maestro22k")
    net = seeded_init_(Unet_CQT_oct_with_attention(args, torch.device(DEV)), 0, gate_scale=10.0, affine_scale=10.0)
    Ls = args.exp.audio_len
    edm = OracleEDM()
    x = torch.from_numpy(seeded_normal(21, 0, Ls)).reshape(1, Ls) * 0.3
    y = torch.from_numpy(seeded_normal(22, 0, Ls)).reshape(1, Ls) * 0.063
    mask = long_gap_mask(Ls, 22050, 300)
    s = torch.full((1, 1), 0.4)
    v = lambda t: t.reshape(-1).to(DEV).contiguous()
    xh, g, nrm = net.denoise_guided(x.to(DEV), v(edm.cnoise(s)), v(edm.cin(s)), v(edm.cskip(s)), v(edm.cout(s)), True,
                                    (y * mask).to(DEV), mask.to(DEV))
    orc = OracleUnet(7, 64, OracleCQT(7, 64, "oct", ("kaiser", 1), 22050, Ls)).load_state_dict(net.state_dict())
    xr = x.clone().requires_grad_()
    xh_ref = orc.CQTransform.apply_hpf_DC(edm.denoiser(xr, orc, s))
    norm = torch.linalg.norm(y * mask - mask * xh_ref, dim=1, ord=2)
    g_ref = torch.autograd.grad(norm.sum(), xr)[0]
    e1, e2 = rel_l2(xh.cpu(), xh_ref.detach()), rel_l2(g.cpu(), g_ref)
    print(f"full-size guided evaluation: x_hat rel-L2 = {e1:.3e}, rec_grads rel-L2 = {e2:.3e}, |norm diff| = {abs(float(nrm.cpu()) - float(norm)):.2e}")
    assert e1 < 1e-4 and e2 < 1e-4


@pytest.mark.parametrize("case", [(2, 64, 64, 16, 64, 2, 1), (1, 96, 96, 16, 128, 4, 0), (1, 128, 256, 16, 32, 8, 1), (2, 256, 128, 32, 256, 1, 1),
                                  (2, 64, 64, 56, 32, 4, 1), (1, 128, 128, 40, 64, 4, 1), (1, 64, 64, 56, 32, 8, 1),
                                  (2, 96, 96, 32, 128, 2, 1), (1, 96, 96, 20, 256, 4, 1),
                                  # xw = 2: F(8,3) row-shared tiles (single-class, multi-class, T = 32, 64+32 pair)
                                  (2, 64, 64, 16, 64, 2, 2), (1, 128, 128, 40, 128, 2, 2), (2, 256, 128, 32, 32, 1, 2), (2, 64, 64, 56, 32, 8, 2),
                                  (2, 96, 96, 32, 128, 2, 2), (1, 96, 96, 16, 256, 4, 2),
                                  # more than 256 F(8,3) tiles: the plain instances (the small cases above run the K-group instances)
                                  (3, 128, 128, 64, 512, 2, 2), (4, 96, 96, 128, 256, 2, 2)])
def test_conv_epilogue_dot_partials(L, case):
    """dgrad conv with the dGELU epilogue + dot_ws: the per-tile partials of <y, aux> per (sample, channel group) that
    replace the aid_group_dot pass, on the in-kernel-transform and the Winograd-domain-input F(4,3) kernels and on the F(8,3) kernel (xw = 2)."""
    B, Cin, Cout, Fd, T, dil, xw = case
    P = L.lib().aid_conv2d_dot_partials(B, Cin, Cout, Fd, T, dil, xw)
    assert P > 0
    g = _rand(B, Cin, Fd, T, seed=40)
    w = _rand(Cout, Cin, 5, 3, seed=41, scale=1.0 / math.sqrt(Cin * 15))
    aux = _rand(B, Cout, Fd, T, seed=42)
    asc = 1.0 + 0.3 * _rand(B, Cout, seed=43)
    gd, wd, auxd, ascd = g.to(DEV), w.to(DEV), aux.to(DEV), asc.to(DEV)
    wp, wpw = L.pack_conv_weight(wd), (L.pack_conv_weight_wino8(wd) if xw == 2 else L.pack_conv_weight_wino(wd))
    y = torch.empty(B, Cout, Fd, T, device=DEV)
    ws = torch.full((B * 8 * (P + 1),), float("nan"), device=DEV, dtype=torch.float64)
    p = L.Conv2dParams()
    xin = gd
    if xw:
        assert xw == 1 or L.lib().aid_conv2d_wino8_supported(Cin, Cout, Fd, T, dil)
        xin = torch.empty(B, Cin, Fd, (10 * (T // 8)) if xw == 2 else (6 * (T // 4)), device=DEV)
        L.call("aid_scale_act", L.ScaleActParams(L.view4(gd), L.view4(xin), None, 0, B, Cin, Fd, T, 0, xw))
    p.x, p.y, p.res, p.aux = L.view4(xin), L.view4(y), L.view4(None), L.view4(auxd)
    p.wp, p.wp_wino, p.wino_taps, p.x_wino = wp.data_ptr(), wpw.data_ptr(), wpw.shape[0], int(xw)
    p.out_scale, p.out_scale_ld = ascd.data_ptr(), ascd.stride(0)
    p.aux_scale, p.aux_scale_ld = ascd.data_ptr(), ascd.stride(0)
    p.B, p.Cin, p.Cout, p.F, p.T = B, Cin, Cout, Fd, T
    p.Cin_pad, p.Cout_pad = wp.shape[1], wp.shape[2]
    p.KH, p.KW, p.dilF, p.act, p.epi = 5, 3, dil, 0, 1
    p.alpha, p.res_scale = 0.7, 1.0
    p.dot_ws, p.dot_n = ws.data_ptr(), P
    L.call("aid_conv2d", p)
    torch.cuda.synchronize()
    yc = y.cpu().double()
    u = aux.double() * asc.double()[:, :, None, None]
    dg = 0.5 * (1 + torch.erf(u / math.sqrt(2))) + u * torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi)
    ref = 0.7 * F.conv2d(g.double(), w.double(), padding="same", dilation=(dil, 1)) * asc.double()[:, :, None, None] * dg
    assert rel_l2(yc, ref) < 1e-5
    got = ws[:B * 8 * P].cpu().reshape(B, 8, P).sum(-1)
    want = (yc * aux.double()).reshape(B, 8, Cout // 8, Fd, T).sum((2, 3, 4))
    assert torch.isfinite(got).all()
    assert float((got - want).abs().max()) < 1e-5 * float((yc.abs() * aux.double().abs()).reshape(B, 8, -1).sum(-1).max())
    if not (xw and L.lib().aid_conv2d_fin_supported(B, Cin, Cout, Fd, T, dil, xw)):
        return
    # fin_mode = 2: the last tile of each sample also writes the coefficients aid_norm_bwd's first kernel would compute from the partials -- same bits
    stats = torch.stack([_rand(B, 8, seed=44), 0.5 + _rand(B, 8, seed=45).abs()], -1).contiguous().to(DEV)      # (mean, 1 / (std + eps)) of the forward
    out = torch.empty_like(y)
    npar = L.NormBwdParams(L.view4(y), L.view4(auxd), L.view4(None), L.view4(out), B, Cout, Fd, T, 8, stats.data_ptr(), ws.data_ptr(), 1e-7, 1.0, 0, P)
    L.call("aid_norm_bwd", npar)
    torch.cuda.synchronize()
    coef_ref = ws[B * 8 * P:].view(torch.float32)[:B * 8].clone()
    out_ref = out.clone()
    assert bool(torch.isfinite(coef_ref).all()) and float(coef_ref.abs().max()) > 0
    cnt = torch.zeros(B, device=DEV, dtype=torch.int32)
    for _ in range(2):
        ws.fill_(float("nan"))
        p.fin_mode, p.fin_count, p.fin_eps, p.fin_stats, p.fin_scale = 2, cnt.data_ptr(), 1e-7, stats.data_ptr(), ws.data_ptr() + 8 * B * 8 * P
        L.call("aid_conv2d", p)
        npar.coef_ready = 1
        out.fill_(float("nan"))
        L.call("aid_norm_bwd", npar)
        torch.cuda.synchronize()
        assert torch.equal(ws[B * 8 * P:].view(torch.float32)[:B * 8], coef_ref) and torch.equal(out, out_ref) and int(cnt.abs().sum()) == 0


@pytest.mark.parametrize("case", [(2, 64, 64, 16, 256, True), (1, 96, 96, 8, 128, False), (3, 128, 128, 64, 32, True), (2, 256, 256, 16, 64, True), (1, 64, 128, 4, 512, False)])
def test_conv1x1_epilogue_dot_partials(L, case):
    """The same <y, aux> partials from the direct-to-LDS 1x1 kernel (dgrad of the 1x1 ResnetBlock steps of the init / out blocks, with the gate
    as per-(b,ci) prologue scale): y and the per-(sample, group) sums against fp64 torch."""
    B, Cin, Cout, Fd, T, gate = case
    P = L.lib().aid_conv2d_dot_partials_1x1(B, Cin, Cout, Fd, T)
    assert P > 0
    g = _rand(B, Cin, Fd, T, seed=50)
    w = _rand(Cout, Cin, 1, 1, seed=51, scale=1.0 / math.sqrt(Cin))
    aux = _rand(B, Cout, Fd, T, seed=52)
    asc = 1.0 + 0.3 * _rand(B, Cout, seed=53)
    isc = (1.0 + 0.5 * _rand(B, Cin, seed=54)) if gate else None
    gd, wd, auxd, ascd = g.to(DEV), w.to(DEV), aux.to(DEV), asc.to(DEV)
    iscd = None if isc is None else isc.to(DEV)
    wp = L.pack_conv_weight(wd)
    y = torch.empty(B, Cout, Fd, T, device=DEV)
    ws = torch.full((B * 8 * (P + 1),), float("nan"), device=DEV, dtype=torch.float64)
    p = L.Conv2dParams()
    p.x, p.y, p.res, p.aux = L.view4(gd), L.view4(y), L.view4(None), L.view4(auxd)
    p.wp = wp.data_ptr()
    p.in_scale, p.in_scale_ld = L.ptr(iscd), (0 if iscd is None else iscd.stride(0))
    p.out_scale, p.out_scale_ld = ascd.data_ptr(), ascd.stride(0)
    p.aux_scale, p.aux_scale_ld = ascd.data_ptr(), ascd.stride(0)
    p.B, p.Cin, p.Cout, p.F, p.T = B, Cin, Cout, Fd, T
    p.Cin_pad, p.Cout_pad = wp.shape[1], wp.shape[2]
    p.KH, p.KW, p.dilF, p.act, p.epi = 1, 1, 1, 0, 1
    p.alpha, p.res_scale = 0.7, 1.0
    p.dot_ws, p.dot_n = ws.data_ptr(), P
    L.call("aid_conv2d", p)
    assert L.lib().aid_last_kernel().decode() in ("conv11_dma_kernel", "conv11_rs_kernel")
    yc = y.cpu().double()
    u = aux.double() * asc.double()[:, :, None, None]
    dg = 0.5 * (1 + torch.erf(u / math.sqrt(2))) + u * torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi)
    gin = g.double() if isc is None else g.double() * isc.double()[:, :, None, None]
    ref = 0.7 * F.conv2d(gin, w.double()) * asc.double()[:, :, None, None] * dg
    assert rel_l2(yc, ref) < 1e-5
    got = ws[:B * 8 * P].cpu().reshape(B, 8, P).sum(-1)
    want = (yc * aux.double()).reshape(B, 8, Cout // 8, Fd, T).sum((2, 3, 4))
    assert torch.isfinite(got).all()
    assert float((got - want).abs().max()) < 1e-5 * float((yc.abs() * aux.double().abs()).reshape(B, 8, -1).sum(-1).max())


def test_sampler_rid_debug_buffers_vs_oracle():
    """Sampler(rid=True).predict_inpainting returns the reference's 8-tuple (edm_sampler_inpainting.py:185-191, :260)."""
    from audio_inpainting_diffusion_amd.edm import EDM
    from audio_inpainting_diffusion_amd.sampler import Sampler
    from oracle.edm import OracleEDM
    from oracle.sampler import OracleSampler
    net, orc, z, kw, args = _setup("a")
    Ls = kw["audio_len"]
    args.tester.T, args.tester.posterior_sampling.xi = 3, 0.25
    args.tester.data_consistency.hann_size = 20
    y = torch.from_numpy(z["x"]) * 0.126
    mask = torch.ones(1, Ls)
    mask[:, 1500:2100] = 0
    smp = Sampler(model=net, diff_params=EDM(args), args=args, rid=True)
    smp.seeds = [7, 8]
    res = smp.predict_inpainting((y * mask).to(DEV), mask.to(DEV))
    ref = OracleSampler(orc, OracleEDM(), T=3, xi=0.25, hann_size=20, audio_len=Ls).predict_inpainting(y * mask, mask, seeds=[7, 8], rid=True)
    assert len(res) == 8
    names = ("out", "denoised", "grads", "grad_update", "pocs", "xt", "xt2", "t")
    for n, a, b in zip(names, res, ref):
        assert tuple(a.shape) == tuple(b.shape), n
        if n in ("denoised", "grads", "grad_update", "pocs", "xt"):
            # first step is teacher-forced identical input; later steps inherit the (chaotic) drift of the trajectory
            assert rel_l2(a[0].cpu(), b[0]) < 1e-4, n
        assert rel_l2(a.cpu(), b) < 2e-3, n
    smp0 = Sampler(model=net, diff_params=EDM(small_xi0(args)), args=small_xi0(args), rid=True)
    with pytest.raises(L_AidError()):
        smp0.predict_inpainting((y * mask).to(DEV), mask.to(DEV))


def small_xi0(args):
    import copy
    a = copy.deepcopy(args)
    a.tester.posterior_sampling.xi = 0.0
    return a


def L_AidError():
    from audio_inpainting_diffusion_amd._lib import AidError
    return AidError


@pytest.mark.parametrize("wform", [1, 2])
@pytest.mark.parametrize("shape,with_gy", [((2, 16, 5, 32), True), ((1, 64, 7, 256), False), ((2, 8, 3, 16), True), ((1, 8, 2, 4096), True)])
def test_norm_bwd_with_winograd_domain_copy(L, shape, with_gy, wform):
    """aid_norm_bwd(wout): the plain output equals the plain call's, and wout is the F(4,3) (wform = 1) / F(8,3) (wform = 2) input transform of
    out * wscale[b,c] (= what aid_scale_act(wino = wform) writes from it)."""
    B, C, Fd, T = shape
    gd, x, gy = _rand(B, C, Fd, T, seed=60), _rand(B, C, Fd, T, seed=61), _rand(B, C, Fd, T, seed=62)
    ws_ = (1.0 + 0.5 * _rand(B, C, seed=63)).to(DEV)
    gdd, xd, gyd = gd.to(DEV), x.to(DEV), gy.to(DEV)
    stats = torch.empty(B, 8, 2, device=DEV)
    scale = torch.empty(B, C, device=DEV)
    gam = torch.ones(C, device=DEV)
    sws = torch.zeros(B * 8 * (L.AID_STATS_SPLIT + 1) * 2, device=DEV, dtype=torch.float64)
    L.call("aid_group_stats", L.GroupStatsParams(L.view4(xd), B, C, Fd, T, 8, gam.data_ptr(), None, 0, 1e-7, scale.data_ptr(), stats.data_ptr(), sws.data_ptr()))
    L.call("aid_group_dot", L.GroupDotParams(L.view4(gdd), L.view4(xd), B, C, Fd, T, 8, sws.data_ptr()))
    outs = []
    for fused in (0, 1):
        out = torch.empty(B, C, Fd, T, device=DEV)
        wout = torch.full((B, C, Fd, (10 * (T // 8)) if wform == 2 else (6 * (T // 4))), float("nan"), device=DEV)
        p = L.NormBwdParams(L.view4(gdd), L.view4(xd), L.view4(gyd if with_gy else None), L.view4(out), B, C, Fd, T, 8, stats.data_ptr(), sws.data_ptr(),
                            1e-7, 0.7, 0, 0)
        if fused:
            p.wout, p.wscale, p.wscale_ld, p.wform = L.view4(wout), ws_.data_ptr(), ws_.stride(0), wform
        L.call("aid_norm_bwd", p)
        torch.cuda.synchronize()
        outs.append((out, wout))
    assert torch.equal(outs[0][0], outs[1][0])
    ref = torch.empty_like(outs[1][1])
    L.call("aid_scale_act", L.ScaleActParams(L.view4(outs[0][0]), L.view4(ref), ws_.data_ptr(), ws_.stride(0), B, C, Fd, T, 0, wform))
    torch.cuda.synchronize()
    assert rel_l2(outs[1][1].cpu(), ref.cpu()) < 1e-6            # (same arithmetic; the compiler contracts the two forms differently)


@pytest.mark.parametrize("wform", [3, 4])
@pytest.mark.parametrize("shape,dil,with_gy", [((2, 16, 24, 64), 2, True), ((1, 64, 20, 32), 4, False), ((2, 8, 28, 96), 1, True), ((1, 8, 8, 544), 1, True),
                                                ((2, 16, 24, 32), 8, True), ((1, 8, 12, 512), 2, True)])
def test_norm_bwd_folded_into_the_2d_input_pass(L, shape, dil, with_gy, wform):
    """aid_norm_bwd(wform = 3 | 4): ONE pass that is both the normalisation backward (out = what the plain call writes, to rounding: the two kernels'
    FMA contraction differs) and the 2-D Winograd input pass of the layer below -- V [48 | 80][C][N] = what aid_scale_act(wino = 3 | 4, scale = wscale,
    dilF = wdil) writes from that result (reverse sweep of the 2-D layers, round 6; ragged row tiles, several residue classes, rows across wave boundaries)."""
    B, C, Fd, T = shape
    nxi, tf = (48, 4) if wform == 3 else (80, 8)
    N = int(L.lib().aid_conv2d_wino2d_positions(B, Fd, T, dil)) * 4 // tf
    gd, x, gy = _rand(B, C, Fd, T, seed=60), _rand(B, C, Fd, T, seed=61), _rand(B, C, Fd, T, seed=62)
    ws_ = (1.0 + 0.5 * _rand(B, C, seed=63)).to(DEV)
    gdd, xd, gyd = gd.to(DEV), x.to(DEV), gy.to(DEV)
    stats, scale, gam = torch.empty(B, 8, 2, device=DEV), torch.empty(B, C, device=DEV), torch.ones(C, device=DEV)
    sws = torch.zeros(B * 8 * (L.AID_STATS_SPLIT + 1) * 2, device=DEV, dtype=torch.float64)
    L.call("aid_group_stats", L.GroupStatsParams(L.view4(xd), B, C, Fd, T, 8, gam.data_ptr(), None, 0, 1e-7, scale.data_ptr(), stats.data_ptr(), sws.data_ptr()))
    L.call("aid_group_dot", L.GroupDotParams(L.view4(gdd), L.view4(xd), B, C, Fd, T, 8, sws.data_ptr()))
    out0, out1 = torch.empty(B, C, Fd, T, device=DEV), torch.full((B, C, Fd, T), float("nan"), device=DEV)
    V0, V1 = torch.empty(nxi * C * N, device=DEV), torch.full((nxi * C * N + 16,), 7.0, device=DEV)
    p = L.NormBwdParams(L.view4(gdd), L.view4(xd), L.view4(gyd if with_gy else None), L.view4(out0), B, C, Fd, T, 8, stats.data_ptr(), sws.data_ptr(), 1e-7, 0.7, 0, 0)
    L.call("aid_norm_bwd", p)
    L.call("aid_scale_act", L.ScaleActParams(L.view4(out0), L.View(V0.data_ptr(), 0, 0, 0), ws_.data_ptr(), ws_.stride(0), B, C, Fd, T, 0, wform, dil))
    p.out = L.view4(out1)
    p.wout, p.wscale, p.wscale_ld, p.wform, p.wdil = L.View(V1.data_ptr(), 0, 0, 0), ws_.data_ptr(), ws_.stride(0), wform, dil
    L.call("aid_norm_bwd", p)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out1).all()) and rel_l2(out1.cpu(), out0.cpu()) < 1e-6
    assert float(V1[nxi * C * N:].min()) == 7.0 and rel_l2(V1[:nxi * C * N].cpu(), V0.cpu()) < 1e-6


def test_fused_passes_match_the_unfused_schedule_at_full_size():
    """Statistics from the conv epilogue (aid_conv2d stat_ws) and the Winograd-domain copy written by aid_norm_bwd (wout) against the same
    network with both switched off (separate read / gate passes): a full-size guided evaluation agrees to rounding, and the fused forms
    really are in the plans."""
    import ctypes
    from audio_inpainting_diffusion_amd.config import make_args
    from audio_inpainting_diffusion_amd.init import seeded_init_, seeded_normal
    from audio_inpainting_diffusion_amd.masks import long_gap_mask
    from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
    from oracle.edm import OracleEDM
    args = make_args("maestro22k")
    Ls = args.exp.audio_len
    edm = OracleEDM()
    x = torch.from_numpy(seeded_normal(23, 0, 2 * Ls)).reshape(2, Ls) * 0.3
    y = torch.from_numpy(seeded_normal(24, 0, 2 * Ls)).reshape(2, Ls) * 0.063
    mask = long_gap_mask(Ls, 22050, 300)
    s = torch.tensor([[0.4], [2.0]])
    v = lambda t: t.reshape(-1).to(DEV).contiguous()
    res = []
    for fused in (True, False):
        net = seeded_init_(Unet_CQT_oct_with_attention(args, torch.device(DEV)), 0, gate_scale=10.0, affine_scale=10.0)
        net.epilogue_stats = net.fuse_norm_bwd_wino = fused
        net.use_graphs = False
        xh, g, nrm = net.denoise_guided(x.to(DEV), v(edm.cnoise(s)), v(edm.cin(s)), v(edm.cskip(s)), v(edm.cout(s)), True,
                                        (y * mask).to(DEV), mask.to(DEV))
        st = net._state(2)
        # (statistics from a conv epilogue: folded by aid_group_stats(ws_n) or, with fin_mode = 1, by the last tile of the conv itself)
        n_stat = sum(1 for k in st["plan_body"].keep if (isinstance(k, L_mod().GroupStatsParams) and k.ws_n > 0)
                     or (isinstance(k, L_mod().Conv2dParams) and k.stat_n > 0 and k.fin_mode == 1))
        n_nb = sum(1 for k in st["plan_bwd"].keep if isinstance(k, L_mod().NormBwdParams) and k.wout.p)
        res.append((xh.cpu(), g.cpu(), nrm.cpu(), n_stat, n_nb))
        del net
        torch.cuda.empty_cache()
    # (round 6: the layers that take the 2-D Winograd form -- at this batch of two every C >= 128 level -- fold the normalisation backward into the input
    #  pass of the layer below (aid_norm_bwd wform = 3 | 4) as the F(4,3) / F(8,3) layers have their Winograd-domain copy written by it: 16 + ~45 here)
    assert res[0][3] >= 40 and res[0][4] >= 50 and res[1][3] == 0 and res[1][4] == 0, (res[0][3:], res[1][3:])
    e1, e2 = rel_l2(res[0][0], res[1][0]), rel_l2(res[0][1], res[1][1])
    print(f"fused vs unfused passes: x_hat rel-L2 = {e1:.2e}, rec_grads rel-L2 = {e2:.2e}; statistics from epilogues: {res[0][3]}, fused norm_bwd: {res[0][4]}")
    assert e1 < 2e-6 and e2 < 2e-5


def L_mod():
    from audio_inpainting_diffusion_amd import _lib
    return _lib


@pytest.mark.parametrize("tag", ["a", "c"])
def test_guided_chain_vs_reference_sampler_and_unet_fixture(tag):
    """sampler_guided_unet.npz = the REFERENCE's Sampler + EDM + U-Net (tests/golden/make_golden.py --only guided).  The HIP network's fused
    guided evaluation, teacher-forced on the recorded input state of every evaluation: x_hat, rec_grads, norm <= 1e-4; then the HIP sampler's
    whole trajectory from the same global-generator seed."""
    from audio_inpainting_diffusion_amd.edm import EDM
    from audio_inpainting_diffusion_amd.sampler import Sampler
    from oracle.edm import OracleEDM
    net, _, zu, kw, args = _setup(tag)
    z = np.load(os.path.join(GOLDEN, "sampler_guided_unet.npz"))
    assert ast.literal_eval(str(z[f"{tag}.cfg"])) == kw and int(z[f"{tag}.seed"]) == int(zu["seed"])
    edm = OracleEDM()
    v = lambda t: t.reshape(-1).to(DEV).contiguous()
    worst = [0.0, 0.0, 0.0]
    for seed in (0, 1):
        k = f"{tag}.s{seed}"
        y, mask = torch.from_numpy(z[k + ".y"]), torch.from_numpy(z[k + ".mask"])
        for e in range(int(z[k + ".n_eval"])):
            x = torch.from_numpy(z[f"{k}.e{e}.x"])
            s = torch.full((1, 1), float(z[f"{k}.e{e}.t"]))
            xh, g, nrm = net.denoise_guided(x.to(DEV), v(edm.cnoise(s)), v(edm.cin(s)), v(edm.cskip(s)), v(edm.cout(s)), True, y.to(DEV), mask.to(DEV))
            e1, e2 = rel_l2(xh.cpu(), z[f"{k}.e{e}.x_hat"]), rel_l2(g.cpu(), z[f"{k}.e{e}.rec_grads"])
            e3 = abs(float(nrm.cpu()) - float(z[f"{k}.e{e}.norm"][0])) / float(z[f"{k}.e{e}.norm"][0])
            worst = [max(a, b) for a, b in zip(worst, (e1, e2, e3))]
        args.tester.T, args.tester.posterior_sampling.xi = 3, 0.25
        args.tester.data_consistency.hann_size = 20
        smp = Sampler(model=net, diff_params=EDM(args), args=args)
        torch.manual_seed(seed)
        out = smp.predict_inpainting(y.to(DEV), mask.to(DEV))
        et = rel_l2(out.cpu(), z[k + ".out"])
        print(f"guided chain vs reference fixture ({k}): trajectory {et:.2e}")
        assert et < 2e-3
    print(f"guided chain vs reference fixture ({tag}): worst x_hat {worst[0]:.2e}, rec_grads {worst[1]:.2e}, norm {worst[2]:.2e}")
    assert worst[0] < 1e-4 and worst[1] < 1e-4 and worst[2] < 1e-4


def test_guided_whole_trajectory_vs_reference_fixture():
    """sampler_guided_traj.npz = a WHOLE T = 10 trajectory of the REFERENCE's Sampler + EDM + U-Net (make_golden.py --only guided_traj): churn only
    inside a window (deterministic head, five stochastic steps, deterministic tail: both branches of edm_sampler_inpainting.py:204), nine Heun steps
    and the final Euler step onto t = 0 (:236-251).  The HIP sampler over the HIP network, free-running from the same global-generator seed: the
    state after EVERY step and the output; and all 19 evaluations teacher-forced on the recorded input states."""
    from audio_inpainting_diffusion_amd.edm import EDM
    from audio_inpainting_diffusion_amd.sampler import Sampler
    from oracle.edm import OracleEDM
    net, _, zu, kw, args = _setup("a")
    z = np.load(os.path.join(GOLDEN, "sampler_guided_traj.npz"))
    assert ast.literal_eval(str(z["cfg"])) == kw and int(z["seed"]) == int(zu["seed"])
    T = int(z["T"])
    g = z["gamma"][:T]
    assert g[0] == 0 and g[T - 1] == 0 and (g > 0).sum() >= 3 and (g == 0).sum() >= 3 and z["t"][-1] == 0 and int(z["n_eval"]) == 2 * T - 1
    y, mask = torch.from_numpy(z["y"]), torch.from_numpy(z["mask"])
    edm = OracleEDM()
    v = lambda t: t.reshape(-1).to(DEV).contiguous()
    worst = [0.0, 0.0, 0.0]
    for e in range(int(z["n_eval"])):
        x = torch.from_numpy(z[f"e{e}.x"])
        s = torch.full((1, 1), float(z[f"e{e}.t"]))
        xh, gr, nrm = net.denoise_guided(x.to(DEV), v(edm.cnoise(s)), v(edm.cin(s)), v(edm.cskip(s)), v(edm.cout(s)), True, y.to(DEV), mask.to(DEV))
        e1, e2 = rel_l2(xh.cpu(), z[f"e{e}.x_hat"]), rel_l2(gr.cpu(), z[f"e{e}.rec_grads"])
        e3 = abs(float(nrm.cpu()) - float(z[f"e{e}.norm"][0])) / float(z[f"e{e}.norm"][0])
        worst = [max(a, b) for a, b in zip(worst, (e1, e2, e3))]
    print(f"whole trajectory, 19 evaluations teacher-forced vs the reference: worst x_hat {worst[0]:.2e}, rec_grads {worst[1]:.2e}, norm {worst[2]:.2e}")
    assert worst[0] < 1e-4 and worst[1] < 1e-4 and worst[2] < 1e-4
    args.tester.T, args.tester.posterior_sampling.xi = T, 0.25
    args.tester.data_consistency.hann_size = 20
    dp = args.tester.diff_params
    dp.Stmin, dp.Stmax, dp.Schurn = float(z["Stmin"]), float(z["Stmax"]), float(z["Schurn"])
    smp = Sampler(model=net, diff_params=EDM(args), args=args, rid=True)
    torch.manual_seed(int(z["noise_seed"]))
    res = smp.predict_inpainting(y.to(DEV), mask.to(DEV))
    assert np.array_equal(res[7].cpu().numpy(), z["t"])
    errs = [rel_l2(res[6][i].cpu(), z["xt2"][i]) for i in range(T)]
    errs_in = [rel_l2(res[5][i].cpu(), z["xt"][i]) for i in range(T)]
    print("HIP free-running trajectory vs the reference, state after every step:", " ".join(f"{e:.1e}" for e in errs), "| out", f"{rel_l2(res[0].cpu(), z['out']):.1e}")
    assert max(errs) < 2e-5 and max(errs_in) < 2e-5 and rel_l2(res[0].cpu(), z["out"]) < 2e-5      # (measured 7e-8 growing to 4e-7 over the ten steps)
    # the same run without the debug buffers returns the same output
    smp2 = Sampler(model=net, diff_params=EDM(args), args=args)
    torch.manual_seed(int(z["noise_seed"]))
    out2 = smp2.predict_inpainting(y.to(DEV), mask.to(DEV))
    assert torch.equal(out2, res[0])



def test_full_size_guided_evaluation_vs_reference_fixture():
    """unet_full_cfgA_guided.npz = ONE guided evaluation of the reference's Sampler + EDM + full-size cfg-A U-Net (make_golden.py --only full_guided):
    norm, every 97th sample and eight seeded projections (+ squared norm) of x_hat and rec_grads."""
    from audio_inpainting_diffusion_amd.config import make_args
    from audio_inpainting_diffusion_amd.init import seeded_init_, seeded_normal
    from audio_inpainting_diffusion_amd.masks import long_gap_mask
    from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
    from oracle.edm import OracleEDM
    z = np.load(os.path.join(GOLDEN, "unet_full_cfgA_guided.npz"))
    args = make_args("maestro22k")
    net = seeded_init_(Unet_CQT_oct_with_attention(args, torch.device(DEV)), 0, gate_scale=10.0, affine_scale=10.0)
    Ls = args.exp.audio_len
    edm = OracleEDM()
    x = torch.from_numpy(seeded_normal(21, 0, Ls)).reshape(1, Ls) * 0.3
    y = torch.from_numpy(seeded_normal(22, 0, Ls)).reshape(1, Ls) * 0.063
    mask = long_gap_mask(Ls, 22050, 300)
    s = torch.full((1, 1), float(z["t"]))
    v = lambda t: t.reshape(-1).to(DEV).contiguous()
    xh, g, nrm = net.denoise_guided(x.to(DEV), v(edm.cnoise(s)), v(edm.cin(s)), v(edm.cskip(s)), v(edm.cout(s)), True, (y * mask).to(DEV), mask.to(DEV))
    assert abs(float(nrm.cpu()) - float(z["norm"][0])) < 1e-4 * float(z["norm"][0])
    for name, t, stream in (("x_hat", xh, 1), ("rec_grads", g, 2)):
        tv = t.cpu().double().reshape(-1).numpy()
        ref_p, ref_s = z[name + "_proj"], z[name + "_s"]
        scale = np.sqrt(ref_p[-1])                                                   # |reference tensor|
        probes = np.stack([seeded_normal(7000 + j, stream, tv.size) for j in range(8)]).astype(np.float64)
        dp = np.abs(probes @ tv - ref_p[:8]).max() / scale                           # ~ |delta| / |ref| for a delta uncorrelated with the probes
        ds = rel_l2(t.cpu()[:, ::97], ref_s)
        print(f"full-size guided evaluation vs REFERENCE fixture: {name}: strided rel-L2 {ds:.2e}, projections {dp:.2e}, |.|^2 {abs(tv @ tv - ref_p[-1]) / ref_p[-1]:.2e}")
        assert ds < 1e-4 and dp < 3e-4 and abs(tv @ tv - ref_p[-1]) < 2e-4 * ref_p[-1]
