"""GPU parity of the non-fused 2-D Winograd path F(4,5) x F(4,3) of the dilated 5x3 convolution (csrc/aid_wino2d.hip), through the C ABI:
aid_scale_act(wino=3) -> aid_conv2d(x_wino=3) = aid_wino2d_gemm + output pass, against the torch-CPU form of the plain convolution
(reference: Conv2d.forward / ResnetBlock.forward, networks/unet_cqt_oct_with_projattention_adaLN_2.py:79-88, :472-482).

fp32 error budget of the form: 1e-5 rel-L2 per layer (tools/wino2d_fm5_error.py: 2.2e-6 ... 4.3e-6).
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def L():
    from audio_inpainting_diffusion_amd import _lib
    _lib.lib()
    return _lib


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize("shape", [(128, 128, 256), (64, 128, 1000), (256, 256, 2052), (32, 128, 4), (96, 96, 1000), (64, 192, 260)])
def test_wino2d_gemm_vs_torch(L, shape):
    """M[xi] = U[xi]^T V[xi] for 48 planes; ragged last column tile, K of 2 .. 16 chunks"""
    cin, cout, N = shape
    cip, cop = L.pack_dims(cin, cout)
    U = torch.zeros(48, cip, cop)
    U[:, :cin, :cout] = _rand(48, cin, cout, seed=1, scale=1 / math.sqrt(cin))
    V = _rand(48, cin, N, seed=2)
    Ud, Vd = U.to(DEV), V.to(DEV)
    M = torch.full((48, cout, N + 8), 7.0, device=DEV)[:, :, :N].contiguous()
    tail = torch.full((64,), 7.0, device=DEV)
    p = L.Wino2dGemmParams(Ud.data_ptr(), Vd.data_ptr(), M.data_ptr(), 48, cin, cout, cip, cop, N, 0)
    L.call("aid_wino2d_gemm", p)
    torch.cuda.synchronize()
    ref = torch.bmm(U[:, :cin, :cout].transpose(1, 2).double(), V.double())
    assert rel_l2(M.cpu(), ref) < 2e-6
    assert float(tail.min()) == 7.0
    # the labelled split-precision variant (three bf16 pieces per fp32 operand, six products on the bf16 matrix pipe, fp32 accumulation): the pieces are an
    # exact decomposition; the dropped cross terms p1 q2 + p2 q1 are ~2^-24 of a product each (measured max 2^-21.3, median 2^-25.2: tools/bf16split_error.py
    # --per-product) -- the order of one fp32 rounding, NOT below 2^-32 -- and the bf16 MFMA sums 16 products per instruction, so over a K-long sum the
    # variant measures at least as close to the fp64 product as the fp32-MFMA kernel (an empirical bar, asserted below, not an exactness claim)
    e32 = rel_l2(M.cpu(), ref)
    if cout % 128 == 0:
        for variant in (100, 101):
            M.fill_(7.0)
            p.variant = variant
            L.call("aid_wino2d_gemm", p)
            torch.cuda.synchronize()
            assert "w2d_gemm_s6" in L.lib().aid_last_kernel().decode()
            es = rel_l2(M.cpu(), ref)
            print(f"GEMM {shape}: fp32 MFMA {e32:.2e}, bf16 x 6 split (variant {variant}) {es:.2e}")
            assert es < 2e-6 and es < 1.5 * e32 + 1e-8 and float(tail.min()) == 7.0


def _conv_ref(x, w, dil, in_scale, act, out_scale, res, res_scale, alpha):
    h = x * in_scale[:, :, None, None]
    if act:
        h = F.gelu(h)
    y = F.conv2d(h, w, padding="same", dilation=(dil, 1))
    if out_scale is not None:
        y = y * out_scale[:, :, None, None]
    if res is not None:
        y = y + res_scale * res
    return alpha * y


W2D_CASES = [
    # B, C, F, T, dil
    (2, 128, 16, 32, 1),         # R = 16: 4 row tiles
    (1, 128, 24, 64, 2),         # R = 12, two residue classes
    (2, 128, 20, 16, 4),         # R = 5: ragged (second tile holds one real row), T = 16
    (1, 128, 28, 32, 1),         # R = 28
    (2, 128, 24, 32, 8),         # R = 3: a single ragged tile per class
    (1, 256, 12, 48, 2),         # two Cout tiles, T = 48 (TG = 12)
    (3, 128, 8, 528, 1),         # TG = 132 > 64 lanes: neighbour samples across wave boundaries
    (2, 96, 24, 64, 2),          # the 96-channel levels: 96 x 128 GEMM tiles (three interleaved row fragments per lane), K = 6 chunks
    (1, 192, 16, 32, 1),         # two 96-row tiles
    (1, 128, 12, 512, 2),        # T = 512: the input pass works in two T segments of 256 samples with a float4 of halo on either side
]


def _input_ref(h, dil, tf=4):
    """V[xf*NTP+xt][c][b*NB + (j*dil + r)*TG + g] from its definition, float64 on the CPU (tf = 4: F(4,3) along T, 6 planes; tf = 8: F(8,3), 10 planes)"""
    from audio_inpainting_diffusion_amd._lib import wino45_matrices, wino8_matrices
    BF = torch.from_numpy(wino45_matrices()[2])
    BT = torch.from_numpy(wino45_matrices()[5] if tf == 4 else wino8_matrices()[2])
    ntp = tf + 2
    B, C, Fd, T = h.shape
    R, TG = Fd // dil, T // tf
    J = (R + 3) // 4
    hs = h.double().reshape(B, C, R, dil, T)                              # row f = jj * dil + r
    hp = F.pad(hs, (1, tf, 0, 0, 2, 4 * J + 4 - R))                          # jj: 2 before, up to 4J+5 after; t: 1 before, tf after
    pt = hp.unfold(-1, ntp, tf)[..., :TG, :]                                # [B, C, JJ, dil, TG, ntp]
    pf = pt.unfold(2, 8, 4)[:, :, :J]                                       # [B, C, J, dil, TG, ntp, 8]
    V = torch.einsum("xi,yk,bcjrgki->xybjrgc", BF, BT, pf)                  # [8, ntp, B, J, dil, TG, C]
    return V.reshape(8 * ntp, B * J * dil * TG, C).permute(0, 2, 1).contiguous()  # [nxi, C, N]


W2D8_CASES = [
    # the F(8,3)-along-T variant (x_wino = 4): T % 32 == 0
    (2, 128, 16, 32, 1),         # TG = 4
    (1, 128, 24, 64, 2),         # two residue classes
    (2, 128, 20, 32, 4),         # R = 5: ragged row tiles
    (2, 128, 24, 32, 8),         # R = 3: a single ragged tile per class
    (1, 256, 12, 96, 2),         # two Cout tiles, TG = 12
    (3, 128, 8, 544, 1),         # TG = 68 > 64 lanes: neighbour samples across wave boundaries
    (2, 96, 24, 64, 2),          # 96-channel level
    (1, 96, 8, 1024, 4),         # T = 1024: four T segments in the input pass, four residue classes
    (2, 128, 20, 512, 1),        # T = 512, ragged row tiles
]


@pytest.mark.parametrize("case", [c + (4,) for c in W2D_CASES] + [c + (8,) for c in W2D8_CASES])
def test_conv2d_wino2d(L, case):
    """tf = 4: F(4,5) x F(4,3) (x_wino = 3, 48 planes); tf = 8: F(4,5) x F(8,3) (x_wino = 4, 80 planes over groups of eight samples; fp32 error budget
    of that form 2e-5 per layer: tools/wino2d_fm5_error.py measures 1.0e-5 at Cin = 128, 1.3e-5 at Cin = 256)"""
    B, C, Fd, T, dil, tf = case
    XW, NXI = (3, 48) if tf == 4 else (4, 80)
    tol = 1e-5 if tf == 4 else 2.5e-5
    assert L.lib().aid_conv2d_wino2d_supported(C, C, Fd, T, dil)
    N = int(L.lib().aid_conv2d_wino2d_positions(B, Fd, T, dil))
    R = Fd // dil
    assert N == B * ((R + 3) // 4) * dil * (T // 4)
    N = N * 4 // tf
    pack2d = L.pack_conv_weight_wino2d if tf == 4 else L.pack_conv_weight_wino2d8
    x = _rand(B, C, Fd, T, seed=40)
    w = _rand(C, C, 5, 3, seed=41, scale=1.0 / math.sqrt(C * 15))
    in_scale = 1.0 + 0.5 * _rand(B, C, seed=42)
    out_scale = _rand(B, C, seed=43)
    res = _rand(B, C, Fd, T, seed=44)
    alpha, res_scale = 1 / math.sqrt(2), 1.5
    xd, wd, isd, osd, resd = x.to(DEV), w.to(DEV), in_scale.to(DEV), out_scale.to(DEV), res.to(DEV)
    # (1) the input transform against its definition (GELU prologue)
    V = torch.full((NXI * C * N + 16,), 7.0, device=DEV)
    sp = L.ScaleActParams(L.view4(xd), L.View(V.data_ptr(), 0, 0, 0), isd.data_ptr(), isd.stride(0), B, C, Fd, T, 1, XW, dil)
    L.call("aid_scale_act", sp)
    torch.cuda.synchronize()
    assert float(V[NXI * C * N:].min()) == 7.0
    Vref = _input_ref(F.gelu(x * in_scale[:, :, None, None]), dil, tf)
    assert rel_l2(V[:NXI * C * N].cpu().reshape(NXI, C, N), Vref) < 1e-6
    # (2) forward: the convolution on it, gate + residual epilogue, (sum, sum of squares) partials
    wp, wpw = L.pack_conv_weight(wd), pack2d(wd)
    y = torch.full((B, C, Fd, T), float("nan"), device=DEV)
    ws = torch.full((NXI * C * N + 16,), 7.0, device=DEV)
    nst = int(L.lib().aid_conv2d_stat_partials(B, C, C, Fd, T, dil, XW))
    assert nst > 0 and nst == int(L.lib().aid_conv2d_dot_partials(B, C, C, Fd, T, dil, XW))
    assert L.lib().aid_conv2d_fin_supported(B, C, C, Fd, T, dil, XW) == 1
    sws = torch.zeros(B * 8 * nst * 2, device=DEV, dtype=torch.float64)
    p = L.Conv2dParams()
    p.x, p.y, p.res, p.aux = L.View(V.data_ptr(), 0, 0, 0), L.view4(y), L.view4(resd), L.view4(None)
    p.wp, p.wp_wino, p.wino_taps, p.x_wino = wp.data_ptr(), wpw.data_ptr(), NXI, XW
    p.out_scale, p.out_scale_ld = osd.data_ptr(), osd.stride(0)
    p.B, p.Cin, p.Cout, p.F, p.T = B, C, C, Fd, T
    p.Cin_pad, p.Cout_pad = wp.shape[1], wp.shape[2]
    p.KH, p.KW, p.dilF, p.act, p.epi = 5, 3, dil, 0, 0
    p.alpha, p.res_scale = alpha, res_scale
    p.ws, p.ws_bytes = ws.data_ptr(), NXI * C * N * 4
    p.stat_ws, p.stat_n = sws.data_ptr(), nst
    L.call("aid_conv2d", p)
    torch.cuda.synchronize()
    assert "w2d_gemm" in L.lib().aid_last_kernel().decode()
    ref = _conv_ref(x, w, dil, in_scale, 1, out_scale, res, res_scale, alpha)
    err = rel_l2(y.cpu(), ref)
    assert err < tol, err
    assert float(ws[NXI * C * N:].min()) == 7.0
    part = sws.cpu().reshape(B, 8, nst, 2).sum(2)
    yg = y.cpu().double().reshape(B, 8, -1)
    assert rel_l2(part[..., 0], yg.sum(-1)) < 1e-5 and rel_l2(part[..., 1], (yg * yg).sum(-1)) < 1e-6
    # aid_group_stats folds these partials like those of the fused kernels
    gamma = (1.0 + 0.3 * _rand(C, seed=45)).to(DEV)
    sc1, sc2 = torch.empty(B, C, device=DEV), torch.empty(B, C, device=DEV)
    st1, st2 = torch.empty(B, 8, 2, device=DEV), torch.empty(B, 8, 2, device=DEV)
    gws = torch.empty(B * 8 * L.AID_STATS_SPLIT * 2, device=DEV, dtype=torch.float64)
    L.call("aid_group_stats", L.GroupStatsParams(L.view4(y), B, C, Fd, T, 8, gamma.data_ptr(), None, 0, 1e-7, sc1.data_ptr(), st1.data_ptr(), gws.data_ptr(), 0))
    L.call("aid_group_stats", L.GroupStatsParams(L.view4(y), B, C, Fd, T, 8, gamma.data_ptr(), None, 0, 1e-7, sc2.data_ptr(), st2.data_ptr(), sws.data_ptr(), nst))
    assert rel_l2(sc2.cpu(), sc1.cpu()) < 1e-6 and rel_l2(st2.cpu(), st1.cpu()) < 1e-5
    # three runs bit-identical
    y2 = torch.empty_like(y)
    p.y = L.view4(y2)
    L.call("aid_conv2d", p)
    torch.cuda.synchronize()
    assert torch.equal(y, y2)
    # fin_mode = 1: the LAST block of each sample's output pass folds the partials itself -- the same bits as aid_group_stats(ws_n = nst), counters back to zero;
    # also through the two separate entry points (GEMM, then output pass)
    cnt = torch.zeros(B + 2, device=DEV, dtype=torch.int32)
    cnt[B:] = 7
    for rep in range(2):
        scf, stf = torch.full((B, C), float("nan"), device=DEV), torch.full((B, 8, 2), float("nan"), device=DEV)
        sws.fill_(float("nan"))
        y2.fill_(float("nan"))
        p.fin_mode, p.fin_count, p.fin_gamma, p.fin_eps, p.fin_scale, p.fin_stats = 1, cnt.data_ptr(), gamma.data_ptr(), 1e-7, scf.data_ptr(), stf.data_ptr()
        if rep == 0:
            L.call("aid_conv2d", p)
        else:
            L.call("aid_conv2d_wino2d_gemm", p)
            assert "w2d_gemm" in L.lib().aid_last_kernel().decode()
            L.call("aid_conv2d_wino2d_output", p)
            assert L.lib().aid_last_kernel().decode() == "w2d_output_kernel"
        torch.cuda.synchronize()
        assert torch.equal(y2, y) and int(cnt[:B].abs().sum()) == 0 and int(cnt[B]) == 7
        assert torch.equal(scf, sc2) and torch.equal(stf, st2)
    p.fin_mode = 0
    # (3) reverse sweep: the transposed operator on the gated gradient (no activation), dGELU epilogue, <y, aux> partials
    gy = _rand(B, C, Fd, T, seed=46)
    gyd = gy.to(DEV)
    sp = L.ScaleActParams(L.view4(gyd), L.View(V.data_ptr(), 0, 0, 0), osd.data_ptr(), osd.stride(0), B, C, Fd, T, 0, XW, dil)
    L.call("aid_scale_act", sp)
    wpT, wpwT = L.pack_conv_weight(wd, transpose=True), pack2d(wd, transpose=True)
    gd = torch.full((B, C, Fd, T), float("nan"), device=DEV)
    dws = torch.zeros(B * 8 * nst, device=DEV, dtype=torch.float64)
    q = L.Conv2dParams()
    q.x, q.y, q.res, q.aux = L.View(V.data_ptr(), 0, 0, 0), L.view4(gd), L.view4(None), L.view4(xd)
    q.wp, q.wp_wino, q.wino_taps, q.x_wino = wpT.data_ptr(), wpwT.data_ptr(), NXI, XW
    q.out_scale, q.out_scale_ld = isd.data_ptr(), isd.stride(0)
    q.aux_scale, q.aux_scale_ld = isd.data_ptr(), isd.stride(0)
    q.B, q.Cin, q.Cout, q.F, q.T = B, C, C, Fd, T
    q.Cin_pad, q.Cout_pad = wpT.shape[1], wpT.shape[2]
    q.KH, q.KW, q.dilF, q.act, q.epi = 5, 3, dil, 0, 1
    q.alpha, q.res_scale = alpha, 1.0
    q.ws, q.ws_bytes = ws.data_ptr(), NXI * C * N * 4
    q.dot_ws, q.dot_n = dws.data_ptr(), nst
    L.call("aid_conv2d", q)
    torch.cuda.synchronize()
    xr = x.clone().requires_grad_(True)
    hr = F.gelu(xr * in_scale[:, :, None, None])
    yr = alpha * F.conv2d(hr, w, padding="same", dilation=(dil, 1)) * out_scale[:, :, None, None]
    gref, = torch.autograd.grad(yr, xr, gy)                              # = alpha * in_scale * gelu'(x in_scale) * conv^T(gy * out_scale)
    assert rel_l2(gd.cpu(), gref) < tol
    dref = (gd.cpu().double() * x.double()).reshape(B, 8, -1).sum(-1)
    assert rel_l2(dws.cpu().reshape(B, 8, nst).sum(-1), dref) < 1e-5
    # fin_mode = 2: the last block also writes the coefficients aid_norm_bwd's first kernel would compute from the partials -- same bits
    dwsf = torch.full((B * 8 * (nst + 1),), float("nan"), device=DEV, dtype=torch.float64)
    dwsf[:B * 8 * nst] = dws
    out = torch.empty_like(gd)
    npar = L.NormBwdParams(L.view4(gd), L.view4(xd), L.view4(None), L.view4(out), B, C, Fd, T, 8, st2.data_ptr(), dwsf.data_ptr(), 1e-7, 1.0, 0, nst)
    L.call("aid_norm_bwd", npar)
    torch.cuda.synchronize()
    coef_ref, out_ref = dwsf[B * 8 * nst:].view(torch.float32)[:B * 8].clone(), out.clone()
    assert bool(torch.isfinite(coef_ref).all()) and float(coef_ref.abs().max()) > 0
    gd2 = torch.empty_like(gd)
    q.y = L.view4(gd2)
    for _ in range(2):
        dwsf.fill_(float("nan"))
        q.dot_ws = dwsf.data_ptr()
        q.fin_mode, q.fin_count, q.fin_eps, q.fin_stats, q.fin_scale = 2, cnt.data_ptr(), 1e-7, st2.data_ptr(), dwsf.data_ptr() + 8 * B * 8 * nst
        L.call("aid_conv2d", q)
        npar.coef_ready = 1
        out.fill_(float("nan"))
        L.call("aid_norm_bwd", npar)
        torch.cuda.synchronize()
        assert torch.equal(gd2, gd) and torch.equal(dwsf[B * 8 * nst:].view(torch.float32)[:B * 8], coef_ref) and torch.equal(out, out_ref) and int(cnt[:B].abs().sum()) == 0
    # (4) the pack kernel writes the same 2-D packs as the torch helper
    outs = [torch.empty_like(wp), torch.empty_like(wpT), torch.empty_like(wpw), torch.empty_like(wpwT)]
    pk = (outs[2].data_ptr(), outs[3].data_ptr(), None, None) if tf == 4 else (None, None, outs[2].data_ptr(), outs[3].data_ptr())
    pp = L.PackConvWeightParams(wd.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(), None, None, C, C, 5, 3, wp.shape[1], wp.shape[2],
                                wpT.shape[1], wpT.shape[2], None, None, *pk)
    L.call("aid_pack_conv_weight", pp)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], wp) and torch.equal(outs[1], wpT)
    assert rel_l2(outs[2].cpu(), wpw.cpu()) < 1e-7 and rel_l2(outs[3].cpu(), wpwT.cpu()) < 1e-7


@pytest.mark.parametrize("case", [(4, 128, 128, 80, 512, 1, "<128x64,foldM,kc32"), (4, 96, 96, 96, 512, 2, "<96x64,foldM,kc24"),
                                  (4, 112, 128, 80, 512, 1, "<128x64,foldM,kc16"), (4, 96, 128, 80, 512, 1, "<128x64,foldM,kc32"),
                                  (4, 112, 96, 96, 512, 2, "<128x64(96),foldM,kc16"), (4, 128, 96, 96, 512, 1, "<128x64(96),foldM,kc32")])
def test_conv2d_wino2d_folded_gemm_on_large_launches(L, case):
    """Launches with >= 768 folded workgroups and Cin <= 128 on the 80-plane form take w2d_gemm_fold_kernel (the row-axis output transform folded into the GEMM: M has
    40 planes) and the FOLD output pass -- the per-shape cases above are too small for it.  Forward with gate + residual and the (sum, sum of squares) partials, through
    aid_conv2d and through the two separate entry points, against the fp64 convolution on the CPU; the kernel that ran is asserted by name: 128-channel panels on
    the four-wave instance with K chunks of 32 (16 where Cin is no multiple of 32), 96-channel panels on the three-wave instance with chunks of 24 or, for other Cin,
    on the four-wave instance with clamped weight columns."""
    B, Ci, C, Fd, T, dil, inst = case
    N = int(L.lib().aid_conv2d_wino2d_positions(B, Fd, T, dil)) // 2
    assert 10 * ((N + 63) // 64) >= 768
    x = _rand(B, Ci, Fd, T, seed=70)
    w = _rand(C, Ci, 5, 3, seed=71, scale=1.0 / math.sqrt(Ci * 15))
    in_scale, out_scale, res = 1.0 + 0.5 * _rand(B, Ci, seed=72), _rand(B, C, seed=73), _rand(B, C, Fd, T, seed=74)
    alpha, res_scale = 1 / math.sqrt(2), 1.5
    xd, wd, isd, osd, resd = x.to(DEV), w.to(DEV), in_scale.to(DEV), out_scale.to(DEV), res.to(DEV)
    V = torch.empty(80 * Ci * N, device=DEV)
    L.call("aid_scale_act", L.ScaleActParams(L.view4(xd), L.View(V.data_ptr(), 0, 0, 0), isd.data_ptr(), isd.stride(0), B, Ci, Fd, T, 1, 4, dil))
    wp, wpw = L.pack_conv_weight(wd), L.pack_conv_weight_wino2d8(wd)
    ws = torch.full((80 * C * N + 16,), 7.0, device=DEV)
    nst = int(L.lib().aid_conv2d_stat_partials(B, Ci, C, Fd, T, dil, 4))
    sws = torch.zeros(B * 8 * nst * 2, device=DEV, dtype=torch.float64)
    ys = []
    for split_calls in (False, True):
        y = torch.full((B, C, Fd, T), float("nan"), device=DEV)
        p = L.Conv2dParams()
        p.x, p.y, p.res, p.aux = L.View(V.data_ptr(), 0, 0, 0), L.view4(y), L.view4(resd), L.view4(None)
        p.wp, p.wp_wino, p.wino_taps, p.x_wino = wp.data_ptr(), wpw.data_ptr(), 80, 4
        p.out_scale, p.out_scale_ld = osd.data_ptr(), osd.stride(0)
        p.B, p.Cin, p.Cout, p.F, p.T = B, Ci, C, Fd, T
        p.Cin_pad, p.Cout_pad = wp.shape[1], wp.shape[2]
        p.KH, p.KW, p.dilF, p.act, p.epi = 5, 3, dil, 0, 0
        p.alpha, p.res_scale = alpha, res_scale
        p.ws, p.ws_bytes = ws.data_ptr(), 80 * C * N * 4
        p.stat_ws, p.stat_n = sws.data_ptr(), nst
        if split_calls:
            L.call("aid_conv2d_wino2d_gemm", p)
            assert inst in L.lib().aid_last_kernel().decode(), L.lib().aid_last_kernel().decode()
            L.call("aid_conv2d_wino2d_output", p)
        else:
            L.call("aid_conv2d", p)
        torch.cuda.synchronize()
        ys.append(y)
    assert torch.equal(ys[0], ys[1]) and float(ws[80 * C * N:].min()) == 7.0
    ref = _conv_ref(x.double(), w.double(), dil, in_scale.double(), 1, out_scale.double(), res.double(), res_scale, alpha)
    err = rel_l2(ys[0].cpu(), ref)
    print(f"folded GEMM {case}: rel-L2 vs the fp64 convolution = {err:.2e}")
    assert err < 2.5e-5, err
    part = sws.cpu().reshape(B, 8, nst, 2).sum(2)
    yg = ys[0].cpu().double().reshape(B, 8, -1)
    assert rel_l2(part[..., 0], yg.sum(-1)) < 1e-5 and rel_l2(part[..., 1], (yg * yg).sum(-1)) < 1e-6
