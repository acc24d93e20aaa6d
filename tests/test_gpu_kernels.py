"""GPU parity, kernel by kernel: every C-ABI entry point of libaid_hip.so against the CPU oracle (oracle/) or
the plain torch-CPU fp32 form of the same op, on the same seeded inputs.

Tolerances (fp32 path, stated per test): 2e-6 rel-L2 for pure data movement / FIR, 1e-5 for reductions and
GEMM-shaped kernels (summation order differs from oneDNN), well inside BASELINE.json's 1e-4 budget.
"""
import ctypes
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


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(2, 16, 5, 12), (3, 64, 7, 256), (1, 96, 128, 1024), (2, 8, 3, 4)])
def test_group_stats(L, shape):
    from oracle.unet import group_std_norm
    B, Cc, Fd, T = shape
    x = _rand(*shape, seed=1, scale=3.0) + 0.5
    gamma = 1.0 + 0.3 * _rand(Cc, seed=2)
    mod = 0.5 * _rand(B, Cc + 5, seed=3)
    xd = x.to(DEV)
    scale = torch.empty(B, Cc, device=DEV)
    stats = torch.empty(B, 8, 2, device=DEV)
    ws = torch.empty(B * 8 * L.AID_STATS_SPLIT * 2, device=DEV, dtype=torch.float64)
    gd, md = gamma.to(DEV), mod.to(DEV)
    p = L.GroupStatsParams(L.view4(xd), B, Cc, Fd, T, 8, gd.data_ptr(), md[:, 3:].data_ptr(), md.stride(0), 1e-7,
                           scale.data_ptr(), stats.data_ptr(), ws.data_ptr())
    L.call("aid_group_stats", p)
    ref = group_std_norm(x, gamma.view(1, -1, 1, 1)) * (1 + mod[:, 3:3 + Cc])[:, :, None, None]
    got = xd * scale[:, :, None, None]
    assert rel_l2(got.cpu(), ref) < 1e-5
    xg = x.reshape(B, 8, -1)
    assert rel_l2(stats[:, :, 0].cpu(), xg.mean(-1)) < 1e-4 + 1e-5
    assert rel_l2(stats[:, :, 1].cpu(), 1.0 / (xg.std(-1) + 1e-7)) < 1e-5


# ---------------------------------------------------------------------------------------------------------
def _conv_ref(x, w, dil, in_scale, act, out_scale, res, res_scale, alpha):
    h = x
    if in_scale is not None:
        h = h * in_scale[:, :, None, None]
    if act:
        h = F.gelu(h)
    kh = w.shape[2]
    y = F.conv2d(h, w, padding="same", dilation=(dil, 1) if kh > 1 else 1)
    if out_scale is not None:
        y = y * out_scale[:, :, None, None]
    if res is not None:
        y = y + res_scale * res
    return alpha * y


CONV_CASES = [
    # B, Cin, Cout, F, T, KH, KW, dil, prologue, epilogue
    (2, 16, 16, 24, 16, 5, 3, 1, True, True),
    (2, 16, 16, 24, 16, 5, 3, 4, True, True),
    (1, 64, 64, 64, 512, 5, 3, 2, True, True),       # level-0 shape class (M=64 config, TT=256)
    (1, 96, 96, 20, 256, 5, 3, 8, True, True),       # M=96 config
    (1, 128, 128, 40, 128, 5, 3, 16, True, True),    # M=128 config, ROWS=2
    (2, 256, 256, 24, 32, 5, 3, 64, True, True),     # two M tiles, ROWS=8, dilation larger than F
    (2, 2, 64, 16, 64, 5, 3, 1, False, True),        # pyramid projection 2 -> C
    (2, 2, 8, 8, 32, 1, 1, 1, False, False),         # init-block proj_in
    (2, 48, 2, 10, 8, 1, 1, 1, False, True),         # out-block projection C -> 2
    (2, 24, 8, 12, 8, 1, 1, 1, True, False),         # attention proj_in (scale prologue, no GELU)
    (2, 8, 24, 12, 8, 1, 1, 1, False, True),         # attention proj_out
    (2, 96, 192, 1, 8, 1, 1, 1, False, False),       # qk GEMM, F = 1
    (1, 320, 640, 1, 128, 1, 1, 1, False, False),    # qk GEMM, larger K (split-K through the scratch buffer)
    (2, 1024, 512, 1, 64, 1, 1, 1, False, True),     # qk-sized GEMM with gate + residual epilogue after the split-K reduction
    (1, 2560, 5120, 1, 128, 1, 1, 1, False, False),  # the reference's batch: N = 128 columns -> skinny GEMM (4 column tiles per wave, K split 8 ways)
    (1, 512, 1024, 1, 32, 1, 1, 1, False, True),     # N = 32: one column tile per wave, epilogue through the reduction
    (3, 256, 640, 1, 36, 1, 1, 1, False, True),      # N = 108: ragged last column tile (columns >= B*T masked), T not a power of two
    (4, 768, 384, 1, 64, 1, 1, 1, False, False),     # N = 256 > 128: direct-to-LDS tiles + split-K
    (3, 40, 40, 9, 12, 5, 3, 2, True, True),         # ragged: T not a power of two, odd row count
    # few-channel layers -> VALU streaming kernels (aid_conv_small.hip)
    (2, 96, 2, 12, 64, 5, 3, 4, False, True),        # pyramid-projection input gradient C -> 2
    (2, 96, 2, 12, 64, 5, 3, 1, False, True),        # ... at dilation 1: sliding-row kernel (4 rows per lane, channels split over the waves)
    (1, 256, 2, 8, 32, 5, 3, 1, False, True),        # deep level: 8 lanes per row, 8 row groups per wave
    (3, 64, 2, 20, 512, 5, 3, 1, False, False),      # two t-tiles per row, no residual
    (2, 40, 2, 4, 8, 5, 3, 1, False, True),          # F = R: every input row of the halo is out of range
    (2, 2, 96, 12, 64, 5, 3, 8, False, True),        # pyramid projection 2 -> C, dilation > F/2
    (2, 64, 2, 10, 32, 1, 1, 1, True, True),         # out-block projection with prologue scale
    (2, 96, 8, 6, 16, 1, 1, 1, True, False),         # attention proj_in
    (2, 8, 64, 6, 16, 1, 1, 1, False, True),         # attention proj_out
    (3, 2, 40, 5, 12, 1, 1, 1, False, False),        # ragged rows, Cout not a multiple of anything
    # 1x1 with enough positions -> streaming kernel (aid_conv1x1.hip)
    (2, 64, 192, 16, 64, 1, 1, 1, True, True),       # two 96-wide Cout slices, per-(b,ci) prologue scale
    (2, 96, 64, 8, 32, 1, 1, 1, False, True),        # 64-wide slice, a wave spans two rows
    (1, 128, 128, 4, 512, 1, 1, 1, False, False),    # 128-wide slice
    (2, 192, 96, 12, 16, 1, 1, 1, True, True),       # a wave spans four rows
    (3, 64, 40, 16, 64, 1, 1, 1, False, True),       # Cout padded to 64: masked rows
    (2, 512, 256, 8, 32, 1, 1, 1, False, True),      # K = 512, two 128-wide slices
    (4, 64, 64, 18, 32, 1, 1, 1, True, True),        # F*T = 9 * 64: the 64-positions-per-wave variant
    # 1x1 with K >= 128 -> direct-to-LDS kernel (aid_conv1x1_dma.hip)
    (2, 128, 96, 16, 64, 1, 1, 1, True, True),       # 96-wide tile (4 waves), prologue scale on the weight fragment
    (1, 256, 192, 8, 128, 1, 1, 1, True, True),      # three 64-wide Cout tiles, two rows per tile
    (2, 144, 128, 8, 256, 1, 1, 1, False, True),     # K = 9 chunks (odd chunk count), 128-wide tile
    # no prologue -> direct-to-LDS (global_load_lds) kernel
    (1, 64, 64, 24, 1024, 5, 3, 2, False, True),     # 64 x 512 tile, two t-tiles per row
    (2, 96, 96, 20, 256, 5, 3, 8, False, True),      # 96 x 256 tile (weight rows padded to 128 in LDS)
    (1, 128, 128, 40, 128, 5, 3, 16, False, True),   # 128 x 256 tile, ROWS = 2
    (2, 256, 256, 24, 32, 5, 3, 64, False, True),    # two M tiles, ROWS = 8, dilation larger than F
    (3, 64, 128, 9, 12, 5, 3, 1, False, False),      # ragged T (zero-page columns), odd rows, Cin != Cout
    (1, 128, 64, 7, 8, 5, 3, 2, False, True),        # ROWS = 32 (halo area full)
    (2, 64, 64, 128, 512, 5, 3, 2, False, True),     # 256 tiles of 512 positions: the full-grid 64 x 512 configuration (smaller launches take 64 x 256)
]


def _wino_applicable(case):
    """the Winograd path of aid_conv2d needs a plain-copy 5x3 conv with Cin % 4 == 0 and Cout >= 64"""
    B, Cin, Cout, Fd, T, KH, KW, dil, pro, epi = case
    return (KH, KW) == (5, 3) and not pro and Cin % 4 == 0 and Cout >= 64


# every case in its direct form; the applicable ones also with the F(4,3) pack (only those are generated: a skipped parametrisation is not a test)
CONV_PARAMS = [(c, 0) for c in CONV_CASES] + [(c, 30) for c in CONV_CASES if _wino_applicable(c)]


@pytest.mark.parametrize("case,wino", CONV_PARAMS)
def test_conv2d(L, case, wino):
    B, Cin, Cout, Fd, T, KH, KW, dil, pro, epi = case
    x = _rand(B, Cin, Fd, T, seed=10)
    w = _rand(Cout, Cin, KH, KW, seed=11, scale=1.0 / math.sqrt(Cin * KH * KW))
    in_scale = (1.0 + 0.5 * _rand(B, Cin, seed=12)) if pro else None
    act = 1 if (pro and KH > 1) else 0
    out_scale = _rand(B, Cout + 3, seed=13) if epi else None
    res = _rand(B, Cout, Fd, T, seed=14) if epi else None
    alpha, res_scale = (1 / math.sqrt(2), 1.5) if epi else (1.0, 1.0)
    ref = _conv_ref(x, w, dil, in_scale, act, None if out_scale is None else out_scale[:, 1:1 + Cout], res, res_scale, alpha)

    xd, wd = x.to(DEV), w.to(DEV)
    wp = L.pack_conv_weight(wd)
    # write into a channel slice of a larger buffer to exercise the strided-view path
    ybig = torch.full((B, Cout + 4, Fd, T), 7.0, device=DEV)
    y = ybig[:, 2:2 + Cout]
    p = L.Conv2dParams()
    resd = None if res is None else res.to(DEV)
    isd = None if in_scale is None else in_scale.to(DEV)
    osd = None if out_scale is None else out_scale.to(DEV)
    p.x, p.y, p.res, p.aux = L.view4(xd), L.view4(y), L.view4(resd), L.view4(None)
    p.wp = wp.data_ptr()
    p.in_scale, p.in_scale_ld = L.ptr(isd), (0 if isd is None else isd.stride(0))
    p.out_scale, p.out_scale_ld = (None if osd is None else osd[:, 1:].data_ptr()), (0 if osd is None else osd.stride(0))
    p.B, p.Cin, p.Cout, p.F, p.T = B, Cin, Cout, Fd, T
    p.Cin_pad, p.Cout_pad = wp.shape[1], wp.shape[2]
    p.KH, p.KW, p.dilF, p.act, p.epi = KH, KW, dil, act, 0
    p.alpha, p.res_scale = alpha, res_scale
    if Fd == 1:                                          # scratch for the split-K path (NULL -> single pass, also valid)
        ws = torch.empty(8 * B * Cout * T, device=DEV)
        p.ws, p.ws_bytes = ws.data_ptr(), ws.numel() * 4
    wpw = None
    if wino:
        wpw = L.pack_conv_weight_wino(wd)
        p.wp_wino, p.wino_taps = wpw.data_ptr(), wino
    L.call("aid_conv2d", p)
    torch.cuda.synchronize()
    assert rel_l2(y.cpu(), ref) < 1e-5
    assert float(ybig[:, :2].min()) == 7.0 and float(ybig[:, 2 + Cout:].min()) == 7.0, "wrote outside its channel slice"


WINO_V_CASES = [
    # B, Cin, Cout, F, T, dil, act, epilogue
    (1, 64, 64, 24, 1024, 2, 1, True),       # two t-tiles per row, GELU prologue
    (2, 96, 64, 20, 256, 8, 0, True),        # Cin != Cout, two rows per tile
    (1, 128, 128, 40, 128, 16, 1, True),     # two M tiles, ROWS = 4
    (2, 256, 256, 24, 32, 64, 1, True),      # ROWS = 16, dilation larger than F
    (3, 64, 128, 9, 48, 1, 0, False),        # T not a power of two (zero-page groups), odd rows
    (2, 64, 64, 128, 512, 4, 1, True),       # 256 tiles of 512 positions: the full-grid 64 x 512 configuration (smaller launches take 64 x 256)
    # row-shared kernel, several residue classes per tile (few rows per class): <TT, NC>
    (2, 64, 64, 56, 32, 2, 1, True),         # <32, 2>: 28 rows per class, 4 rows x 2 classes
    (1, 128, 64, 56, 32, 4, 0, True),        # <32, 4>: 14 rows per class, 2 rows x 4 classes
    (2, 64, 128, 56, 32, 8, 1, True),        # <32, 8>: 7 rows per class, one row of 8 classes
    (1, 64, 64, 40, 64, 4, 1, False),        # <64, 2>: 10 rows per class
    (2, 64, 64, 20, 128, 4, 0, True),        # <64, 4>: 5 rows per class, two t tiles
    (1, 256, 256, 448, 32, 16, 1, True),     # the deepest level of the shipped network (28 rows per class)
    # T = 16 (deepest level of the 8-octave 44.1 kHz network at L = 184184): <16, NC> tiles, 16 / NC rows per class
    (2, 256, 256, 64, 16, 1, 1, True),       # <16, 1>
    (1, 64, 128, 64, 16, 8, 0, True),        # <16, 2>: 8 rows per class
    (2, 128, 64, 32, 16, 8, 1, False),       # <16, 4>: 4 rows per class
    (1, 64, 64, 32, 16, 16, 1, True),        # <16, 8>: 2 rows per class
    # 96 output channels: 64-channel tiles + 32-channel x 512-position remainder tiles (two launches)
    (1, 96, 96, 32, 256, 2, 1, True),
    (2, 64, 96, 48, 64, 2, 0, False),        # remainder tile over two residue classes <64, 2, 1>
    (2, 96, 96, 20, 256, 4, 1, True),        # 5 rows per class: the row-shared kernel declines -> 96 x 512 tiles
    (2, 64, 96, 16, 32, 2, 0, False),        # T = 32: no remainder-tile instance -> 96 x 512 tiles
    # batch 1, few tiles: the split-K instances (two workgroups per tile) when `ws` is given (T >= 64 tile shapes; T <= 32: the K-group instances)
    (1, 256, 128, 384, 64, 2, 1, True),      # 192 tiles (the first up-path conv of level 5)
    (1, 128, 128, 160, 128, 32, 0, True),    # 160 tiles, <64, 4>
    (1, 256, 256, 448, 32, 64, 1, False),    # <32, 8>
    (1, 96, 64, 36, 128, 1, 1, True),        # 24 chunks per half
]


@pytest.mark.parametrize("case", WINO_V_CASES)
def test_conv2d_winograd_domain_input(L, case):
    """aid_scale_act(wino=1) -> aid_conv2d(x_wino=1): the input transform of F(4,3) done by the producer pass."""
    B, Cin, Cout, Fd, T, dil, act, epi = case
    assert L.lib().aid_conv2d_wino_input_ok(B, Cin, Cout, Fd, T, dil)
    assert bool(L.lib().aid_conv2d_wino_input_supported(Cin, Cout, T)) == (T >= 32)      # (the geometry-free form only vouches for T >= 32)
    x = _rand(B, Cin, Fd, T, seed=30)
    w = _rand(Cout, Cin, 5, 3, seed=31, scale=1.0 / math.sqrt(Cin * 15))
    in_scale = 1.0 + 0.5 * _rand(B, Cin, seed=32)
    out_scale = _rand(B, Cout, seed=33) if epi else None
    res = _rand(B, Cout, Fd, T, seed=34) if epi else None
    alpha, res_scale = (1 / math.sqrt(2), 1.5) if epi else (1.0, 1.0)
    ref = _conv_ref(x, w, dil, in_scale, act, out_scale, res, res_scale, alpha)
    xd, wd, isd = x.to(DEV), w.to(DEV), in_scale.to(DEV)
    G = T // 4
    # (1) the transform itself against its definition
    xv = torch.full((B, Cin, Fd, 6 * G + 4), 7.0, device=DEV)[..., :6 * G]          # padded rows: strided view
    L.call("aid_scale_act", L.ScaleActParams(L.view4(xd), L.view4(xv), isd.data_ptr(), isd.stride(0), B, Cin, Fd, T, act, 1))
    h = x * in_scale[:, :, None, None]
    if act:
        h = F.gelu(h)
    d = F.pad(h, (1, 4)).unfold(-1, 6, 4)[..., :G, :]                               # d[..., g, :] = h[4g-1 .. 4g+4]
    BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0],
                       [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float32)
    Vref = torch.einsum("xk,bcfgk->bcfxg", BT, d).reshape(B, Cin, Fd, 6 * G)
    assert rel_l2(xv.cpu(), Vref) < 1e-6
    # (2) the convolution on it
    wp, wpw = L.pack_conv_weight(wd), L.pack_conv_weight_wino(wd)
    y = torch.empty(B, Cout, Fd, T, device=DEV)
    p = L.Conv2dParams()
    resd = None if res is None else res.to(DEV)
    osd = None if out_scale is None else out_scale.to(DEV)
    p.x, p.y, p.res, p.aux = L.view4(xv), L.view4(y), L.view4(resd), L.view4(None)
    p.wp, p.wp_wino, p.wino_taps, p.x_wino = wp.data_ptr(), wpw.data_ptr(), 30, 1
    p.out_scale, p.out_scale_ld = L.ptr(osd), (0 if osd is None else osd.stride(0))
    p.B, p.Cin, p.Cout, p.F, p.T = B, Cin, Cout, Fd, T
    p.Cin_pad, p.Cout_pad = wp.shape[1], wp.shape[2]
    p.KH, p.KW, p.dilF, p.act, p.epi = 5, 3, dil, 0, 0
    p.alpha, p.res_scale = alpha, res_scale
    L.call("aid_conv2d", p)
    torch.cuda.synchronize()
    assert rel_l2(y.cpu(), ref) < 1e-5
    # (3) split-K (batch 1, few tiles): same result up to the order of one addition, deterministic, flags left zero
    need = int(L.lib().aid_conv2d_wino_split_ws_bytes(B, Cin, Cout, Fd, T, dil))
    tiles = B * Fd * T * (Cout // 64) // 256
    kern = L.lib().aid_last_kernel().decode()
    # (launches of at most 256 tiles on the T <= 32 tile shapes take the K-group instance -- eight waves per tile, the K halves summed through LDS -- and need no scratch)
    assert ("wino4r_ks" in kern) == (Cout % 64 == 0 and T <= 32 and tiles <= 256 and "wino4r" in kern), (kern, tiles)
    assert (need > 0) == (B == 1 and Cout % 64 == 0 and tiles <= 230 and "wino4r" in kern and "wino4r_ks" not in kern), (need, kern)
    if need:
        ws = torch.zeros(need // 4 + 8, device=DEV)
        ws[need // 4:] = 7.0
        p.ws, p.ws_bytes = ws.data_ptr(), need
        ys = []
        for _ in range(3):
            y2 = torch.full_like(y, float("nan"))
            p.y = L.view4(y2)
            L.call("aid_conv2d", p)
            assert L.lib().aid_last_kernel().decode() == "conv53_wino4r_kernel(split-K)"
            ys.append(y2)
        torch.cuda.synchronize()
        assert rel_l2(ys[0].cpu(), ref) < 1e-5 and rel_l2(ys[0].cpu(), y.cpu()) < 5e-6
        assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2])
        assert float(ws[:1024].abs().max()) == 0.0 and float(ws[need // 4:].min()) == 7.0


WINO8_CASES = [
    # B, Cin, Cout, F, T, dil, act, epilogue          F(8,3) row-shared tiles <TT, NC>: 64 groups of 8 outputs = RA rows x NC classes x TT samples
    (1, 64, 64, 24, 1024, 2, 1, True),       # <64, 1>: 12 rows per class -> two quads of 8 (the second half empty), 16 t-tiles
    (2, 64, 128, 32, 64, 1, 1, True),        # <64, 1>: RA = 8, four quads, two Cout tiles
    (2, 128, 64, 40, 128, 2, 0, True),       # <64, 2>: 20 rows per class (RA = 8 would pad 20 -> 24): RA = 4 of two classes
    (1, 256, 256, 448, 32, 1, 1, True),      # <32, 1>: RA = 16, the deepest level of the shipped network
    (2, 64, 64, 56, 32, 2, 1, True),         # <32, 1>: 28 rows per class -> 2 quads of 16 (14 % padding rows, still fewer MFMAs than F(4,3))
    (2, 64, 128, 56, 32, 8, 1, True),        # <32, 2>: 7 rows per class -> RA = 8 of two classes
    (2, 64, 128, 32, 32, 8, 0, True),        # <32, 4>: 4 rows per class -> RA = 4 of four classes
    (3, 64, 64, 16, 96, 1, 0, False),        # T = 96: TT = 32, three t-tiles
    (1, 256, 128, 384, 64, 64, 1, True),     # 6 rows per class: no F(8,3) tile fits -> the library answers F(4,3)
    (4, 64, 64, 96, 1024, 1, 1, True),       # 768 tiles: 1.5 rounds of the 512 resident workgroups -> stream-K cuts every third tile
    (5, 128, 128, 64, 512, 2, 0, True),      # 640 tiles x 2 Cout tiles = 1280: 2.5 rounds
    # 96 output channels: pair instance (64-channel tiles + 32-channel x 1024-position remainder tiles)
    (1, 96, 96, 32, 256, 2, 1, True),        # <64, 1> + <64, 1>: 16 rows per class
    (2, 64, 96, 48, 64, 2, 0, False),        # 24 rows per class: <64, 1> + remainder over two classes (RA = 8)
    (2, 96, 96, 16, 128, 4, 1, True),        # 4 rows per class: <64, 2> + <64, 4>
    # more than 256 tiles: the plain instances (two four-wave workgroups per CU) of the tile shapes the small cases above run as K-group instances
    (5, 128, 128, 448, 32, 1, 1, True),      # <32, 1>, 280 tiles
    (5, 64, 128, 448, 32, 64, 1, True),      # <32, 2>, 320 tiles
    (9, 64, 128, 256, 32, 64, 0, True),      # <32, 4>, 288 tiles
    (4, 96, 96, 128, 256, 2, 1, True),       # pair <64, 1> + <64, 1>, 384 tiles
    (30, 64, 96, 48, 64, 2, 0, False),       # pair <64, 1> + <64, 2>, 270 tiles
    (44, 96, 96, 16, 128, 4, 1, True),       # pair <64, 2> + <64, 4>, 264 tiles
]
# launches of at most 256 tiles (one per CU) take the K-group instances: one eight-wave workgroup per tile, two K groups (conv53_wino8r_ks_kernel)
WINO8_PLAIN = {(4, 64, 64, 96, 1024), (5, 128, 128, 64, 512), (5, 128, 128, 448, 32), (5, 64, 128, 448, 32), (9, 64, 128, 256, 32), (4, 96, 96, 128, 256),
               (30, 64, 96, 48, 64), (44, 96, 96, 16, 128)}


@pytest.mark.parametrize("case", WINO8_CASES)
def test_conv2d_winograd8_domain_input(L, case):
    """aid_scale_act(wino=2) -> aid_conv2d(x_wino=2): F(8,3) along T on the row-shared tiles (10 MFMAs per 8 outputs), against the torch-CPU
    reference of the plain convolution.  fp32 error budget of the form: 1e-5 per layer (measured 3e-6 ... 6e-6, tools/wino_fm3_error.py)."""
    B, Cin, Cout, Fd, T, dil, act, epi = case
    if (Cin, Cout, Fd, T, dil) == (256, 128, 384, 64, 64):
        assert not L.lib().aid_conv2d_wino8_supported(Cin, Cout, Fd, T, dil) and L.lib().aid_conv2d_wino_form(B, Cin, Cout, Fd, T, dil) == 4
        return
    assert L.lib().aid_conv2d_wino8_supported(Cin, Cout, Fd, T, dil)
    assert L.lib().aid_conv2d_wino_form(B, Cin, Cout, Fd, T, dil) in (4, 8) and L.lib().aid_conv2d_wino_form(8 * B, Cin, Cout, Fd, T, dil) in (4, 8)
    x = _rand(B, Cin, Fd, T, seed=40)
    w = _rand(Cout, Cin, 5, 3, seed=41, scale=1.0 / math.sqrt(Cin * 15))
    in_scale = 1.0 + 0.5 * _rand(B, Cin, seed=42)
    out_scale = _rand(B, Cout, seed=43) if epi else None
    res = _rand(B, Cout, Fd, T, seed=44) if epi else None
    alpha, res_scale = (1 / math.sqrt(2), 1.5) if epi else (1.0, 1.0)
    ref = _conv_ref(x, w, dil, in_scale, act, out_scale, res, res_scale, alpha)
    xd, wd, isd = x.to(DEV), w.to(DEV), in_scale.to(DEV)
    G = T // 8
    # (1) the input transform against its definition
    xv = torch.full((B, Cin, Fd, 10 * G + 4), 7.0, device=DEV)[..., :10 * G]        # padded rows: strided view
    L.call("aid_scale_act", L.ScaleActParams(L.view4(xd), L.view4(xv), isd.data_ptr(), isd.stride(0), B, Cin, Fd, T, act, 2))
    h = x * in_scale[:, :, None, None]
    if act:
        h = F.gelu(h)
    d = F.pad(h, (1, 8)).unfold(-1, 10, 8)[..., :G, :]                              # d[..., g, :] = h[8g-1 .. 8g+8]
    AT, Gm, BT = L.wino8_matrices()
    Vref = torch.einsum("xk,bcfgk->bcfxg", torch.from_numpy(BT), d.double()).reshape(B, Cin, Fd, 10 * G)
    assert rel_l2(xv.cpu().double(), Vref) < 1e-6
    # (2) the convolution on it
    wp, wpw = L.pack_conv_weight(wd), L.pack_conv_weight_wino8(wd)
    y = torch.empty(B, Cout, Fd, T, device=DEV)
    p = L.Conv2dParams()
    resd = None if res is None else res.to(DEV)
    osd = None if out_scale is None else out_scale.to(DEV)
    p.x, p.y, p.res, p.aux = L.view4(xv), L.view4(y), L.view4(resd), L.view4(None)
    p.wp, p.wp_wino, p.wino_taps, p.x_wino = wp.data_ptr(), wpw.data_ptr(), 50, 2
    p.out_scale, p.out_scale_ld = L.ptr(osd), (0 if osd is None else osd.stride(0))
    p.B, p.Cin, p.Cout, p.F, p.T = B, Cin, Cout, Fd, T
    p.Cin_pad, p.Cout_pad = wp.shape[1], wp.shape[2]
    p.KH, p.KW, p.dilF, p.act, p.epi = 5, 3, dil, 0, 0
    p.alpha, p.res_scale = alpha, res_scale
    L.call("aid_conv2d", p)
    torch.cuda.synchronize()
    kern = L.lib().aid_last_kernel().decode()
    assert "wino8r" in kern and ("wino8r_ks" in kern) == ((B, Cin, Cout, Fd, T) not in WINO8_PLAIN), kern
    err = rel_l2(y.cpu(), ref)
    assert err < 1e-5, err
    # (3) the pack kernel writes the same F(8,3) packs as the torch helper
    outs = [torch.empty_like(wp), torch.empty(15, *L.pack_dims(Cout, Cin), device=DEV), torch.empty(50, *wp.shape[1:], device=DEV),
            torch.empty(50, *L.pack_dims(Cout, Cin), device=DEV)]
    pp = L.PackConvWeightParams(wd.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(), None, None, Cout, Cin, 5, 3, wp.shape[1], wp.shape[2],
                                outs[1].shape[1], outs[1].shape[2], outs[2].data_ptr(), outs[3].data_ptr())
    L.call("aid_pack_conv_weight", pp)
    assert torch.equal(outs[2], wpw) and torch.equal(outs[3], L.pack_conv_weight_wino8(wd, transpose=True))


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(2, 3, 4, 16), (1, 2, 5, 8), (2, 4, 3, 256), (1, 1, 2, 2048)])
def test_resample_and_adjoints(L, shape):
    from oracle.unet import resample_down, resample_up
    B, Cc, Fd, T = shape
    x = _rand(*shape, seed=20)
    xd = x.to(DEV)

    def run(inp, Tout, up, adjoint):
        out = torch.empty(B, Cc, Fd, Tout, device=DEV)
        p = L.ResampleParams(L.view4(inp), L.view4(out), B, Cc, Fd, inp.shape[-1], up, adjoint)
        L.call("aid_resample", p)
        return out

    dn = run(xd, T // 2, 0, 0)
    up = run(xd, 2 * T, 1, 0)
    assert rel_l2(dn.cpu(), resample_down(x)) < 2e-6
    assert rel_l2(up.cpu(), resample_up(x)) < 2e-6
    # adjoints: <A x, g> == <x, A^T g>
    g1 = _rand(B, Cc, Fd, T // 2, seed=21).to(DEV)
    g2 = _rand(B, Cc, Fd, 2 * T, seed=22).to(DEV)
    a1 = run(g1, T, 0, 1)
    a2 = run(g2, T, 1, 1)
    assert abs(float((dn.double() * g1.double()).sum() - (xd.double() * a1.double()).sum())) < 1e-4 * float(dn.norm() * g1.norm())
    assert abs(float((up.double() * g2.double()).sum() - (xd.double() * a2.double()).sum())) < 1e-4 * float(up.norm() * g2.norm())


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(2, 8, 12, 8), (1, 8, 40, 32), (2, 8, 320, 128), (1, 4, 56, 64), (1, 8, 9, 4)])
def test_time_attention(L, shape):
    B, H, Fd, T = shape
    qk = _rand(B, H * 2 * Fd, T, seed=30, scale=2.0)
    v = _rand(B, H, Fd, T, seed=31)
    q4 = qk.reshape(B, H, 2 * Fd, T).permute(0, 1, 3, 2)
    q, k = q4[..., :Fd], q4[..., Fd:]
    sim = torch.einsum("bhnd,bhmd->bhnm", q, k) * (float(Fd) ** -0.5)
    attn = sim.softmax(-1)
    ref = torch.einsum("bhnm,bhmd->bhnd", attn, v.permute(0, 1, 3, 2)).permute(0, 1, 3, 2)
    qd, vd = qk.to(DEV), v.to(DEV)
    out = torch.empty(B, H, Fd, T, device=DEV)
    probs = torch.empty(B, H, T, T, device=DEV)
    p = L.AttentionParams(qd.data_ptr(), vd.data_ptr(), out.data_ptr(), probs.data_ptr(), B, H, Fd, T, float(Fd) ** -0.5)
    L.call("aid_time_attention", p)
    assert rel_l2(probs.cpu(), attn) < 1e-5
    assert rel_l2(out.cpu(), ref) < 1e-5


# ---------------------------------------------------------------------------------------------------------
def test_embed_and_modulation(L):
    from oracle.unet import embed
    E, B = 48, 5
    sd = {"embedding.RFF_freq": 16 * _rand(1, 32, seed=40)}
    dims = [(128, 64), (256, 128), (E, 256)]
    for i, (o, ii) in enumerate(dims):
        sd[f"embedding.MLP.{i}.weight"] = _rand(o, ii, seed=41 + i, scale=1 / math.sqrt(ii))
        sd[f"embedding.MLP.{i}.bias"] = 0.1 * _rand(o, seed=45 + i)
    sig = torch.tensor([[-2.1], [-0.3], [0.0], [-1.0], [0.4]])
    ref = embed(sd, sig)
    d = {k: v.to(DEV).contiguous() for k, v in sd.items()}
    sg = sig.reshape(-1).to(DEV).contiguous()
    emb = torch.empty(B, E, device=DEV)
    p = L.EmbedParams(sg.data_ptr(), d["embedding.RFF_freq"].data_ptr(),
                      d["embedding.MLP.0.weight"].data_ptr(), d["embedding.MLP.0.bias"].data_ptr(),
                      d["embedding.MLP.1.weight"].data_ptr(), d["embedding.MLP.1.bias"].data_ptr(),
                      d["embedding.MLP.2.weight"].data_ptr(), d["embedding.MLP.2.bias"].data_ptr(),
                      emb.data_ptr(), B, 32, 128, 256, E)
    L.call("aid_embed", p)
    assert rel_l2(emb.cpu(), ref) < 2e-5     # sin/cos of ~1e2 rad arguments: device vs host libm
    N = 1000
    Wm, bm = _rand(N, E, seed=50), _rand(N, seed=51)
    Wd, bd_ = Wm.to(DEV), bm.to(DEV)
    for Bm in (5, 19):
        e = _rand(Bm, E, seed=52).to(DEV)
        mod = torch.empty(Bm, N, device=DEV)
        L.call("aid_modulation", L.ModulationParams(e.data_ptr(), Wd.data_ptr(), bd_.data_ptr(), mod.data_ptr(), Bm, E, N))
        assert rel_l2(mod.cpu(), e.cpu() @ Wm.t() + bm) < 1e-5


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [(3, 8, 22050, 2048), (7, 8, 22050, 16384), (4, 8, 16000, 4096), (7, 64, 22050, 184184)])
def test_cqt_against_oracle(L, cfg):
    from audio_inpainting_diffusion_amd.cqt import CQTransform
    from oracle.nsgt_cqt import OracleCQT
    no, bpo, fs, Ls = cfg
    orc = OracleCQT(no, bpo, "oct", ("kaiser", 1), fs, Ls)
    tr = CQTransform(no, bpo, "oct", ("kaiser", 1.0), fs, Ls, device=DEV)
    x = _rand(2, Ls, seed=60, scale=0.063)
    c_ref = orc.fwd(x[:, None])
    c = tr.fwd(x[:, None].to(DEV))
    for a, b in zip(c, c_ref):
        assert a.shape == b.shape and a.dtype == torch.complex64
        assert rel_l2(torch.view_as_real(a.cpu()), torch.view_as_real(b)) < 1e-5
    # synthesis of ARBITRARY coefficients (not in the range of fwd)
    cr = [torch.complex(_rand(*ci.shape, seed=61 + i), _rand(*ci.shape, seed=71 + i)) for i, ci in enumerate(c_ref)]
    y_ref = orc.bwd(cr)
    y = tr.bwd([ci.to(DEV) for ci in cr])
    assert rel_l2(y.cpu(), y_ref) < 1e-5
    # perfect reconstruction up to the DC/Nyquist projector (BASELINE's "CQT round-trip", <= 1e-5)
    rt = tr.bwd(c)[:, 0]
    hp = tr.apply_hpf_DC(x.to(DEV))
    assert rel_l2(hp.cpu(), orc.apply_hpf_DC(x)) < 1e-5
    assert rel_l2(rt.cpu(), hp.cpu()) < 1e-5


# ---------------------------------------------------------------------------------------------------------
def test_sampler_elementwise(L):
    B, Ls = 3, 1000
    x, xh, y = _rand(B, Ls, seed=80), _rand(B, Ls, seed=81), _rand(B, Ls, seed=82)
    sm = torch.rand(B, Ls, generator=torch.Generator().manual_seed(83))
    t, h = torch.tensor([0.5, 0.7, 0.9]), torch.tensor([-0.1, -0.2, -0.3])
    xd, xhd, yd, smd, td, hd = (a.to(DEV) for a in (x, xh, y, sm, t, h))
    xn, dd, xo = torch.empty_like(xd), torch.empty_like(xd), torch.empty_like(xd)
    p = L.ScoreStepParams(xd.data_ptr(), xhd.data_ptr(), yd.data_ptr(), smd.data_ptr(), smd.stride(0), None, None,
                          td.data_ptr(), hd.data_ptr(), xn.data_ptr(), dd.data_ptr(), xo.data_ptr(), B, Ls, 0)
    L.call("aid_score_step", p)
    xp = sm * y + (1 - sm) * xh
    score = (xp - x) / t[:, None] ** 2
    d = -t[:, None] * score
    assert rel_l2(xo.cpu(), xp) < 2e-6 and rel_l2(dd.cpu(), d) < 2e-6 and rel_l2(xn.cpu(), x + h[:, None] * d) < 2e-6
    # Heun combine
    x2 = torch.empty_like(xd)
    p = L.ScoreStepParams(xn.data_ptr(), xhd.data_ptr(), yd.data_ptr(), smd.data_ptr(), smd.stride(0), xd.data_ptr(), dd.data_ptr(),
                          td.data_ptr(), hd.data_ptr(), x2.data_ptr(), None, None, B, Ls, 1)
    L.call("aid_score_step", p)
    xpr = x + h[:, None] * d
    d2 = -(t[:, None]) * ((xp - xpr) / t[:, None] ** 2)
    assert rel_l2(x2.cpu(), x + h[:, None] * (0.5 * d + 0.5 * d2)) < 2e-6
    out = torch.empty_like(xd)
    L.call("aid_axpby", L.AxpbyParams(xd.data_ptr(), yd.data_ptr(), out.data_ptr(), None, td.data_ptr(), B, Ls))
    assert rel_l2(out.cpu(), x + t[:, None] * y) < 2e-6


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("Ls", [2048, 4096, 16384, 65536, 184184, 368368, 2 * 3 * 5 * 7 * 11 * 13])
def test_mixed_radix_fft_vs_rocfft(L, Ls):
    """aid_fft_pass (own mixed-radix Stockham) against torch.fft (rocFFT) as an independent GPU implementation and
    against numpy fp64 on the host: rfft, irfft, round trip."""
    from audio_inpainting_diffusion_amd.cqt import CQTransform, CQTPlan
    tr = CQTransform.__new__(CQTransform)
    tr.plan = type("P", (), {})()
    from audio_inpainting_diffusion_amd.cqt import fft_radices
    tr.plan.L, tr.plan.Lh, tr.plan.radices = Ls, Ls // 2 + 1, fft_radices(Ls)
    tw = np.stack([np.cos(2 * np.pi * np.arange(Ls) / Ls), -np.sin(2 * np.pi * np.arange(Ls) / Ls)], axis=-1).astype(np.float32).reshape(-1)
    tr._dev = dict(device=torch.device(DEV, 0), twiddle_L=torch.from_numpy(tw).to(DEV))
    tr._tables = lambda device: tr._dev
    x = _rand(3, Ls, seed=5)
    xd = x.to(DEV)
    X = tr.rfft(xd)
    ref = np.fft.rfft(x.double().numpy(), axis=-1)
    assert rel_l2(torch.view_as_real(X.cpu()), torch.view_as_real(torch.from_numpy(ref))) < 2e-6
    assert rel_l2(torch.view_as_real(X.cpu()), torch.view_as_real(torch.fft.rfft(xd).cpu())) < 2e-6
    Y = torch.complex(_rand(3, Ls // 2 + 1, seed=6), _rand(3, Ls // 2 + 1, seed=7)).to(DEV)
    y = tr.irfft(Y)
    assert rel_l2(y.cpu(), torch.fft.irfft(Y, n=Ls).cpu()) < 2e-6
    assert rel_l2(tr.irfft(X).cpu(), x) < 2e-6


def test_conv2d_dispatch_fuzz(L):
    """Seeded random sweep over the aid_conv2d dispatch space (F(4,3) with and without Winograd-domain input, direct
    5x3, tiled / streaming / direct-to-LDS 1x1, few-channel VALU kernels, split-K) with strided views, residuals that
    alias the output, dGELU epilogues and the dot partials: every launch against the fp64 reference."""
    rng = np.random.default_rng(20260928)
    n_checked, paths = 0, set()
    for it in range(70):
        k53 = bool(rng.integers(0, 2))
        KH, KW = (5, 3) if k53 else (1, 1)
        Cin = int(rng.choice([2, 8, 16, 40, 64, 96, 128, 144, 256]))
        Cout = int(rng.choice([2, 8, 24, 64, 96, 128, 192, 256]))
        T = int(rng.choice([8, 12, 16, 32, 48, 64, 128, 256]))
        Fd = int(rng.choice([1, 3, 8, 12, 16])) if not k53 else int(rng.choice([5, 8, 12, 16, 24]))
        B = int(rng.integers(1, 4))
        if B * Cin * Fd * T * (15 if k53 else 1) * Cout > 4e9:
            continue
        dil = int(rng.choice([1, 2, 4, 16])) if k53 else 1
        use_in = bool(rng.integers(0, 2))
        act = int(use_in and rng.integers(0, 2))
        use_out = bool(rng.integers(0, 2))
        use_res = bool(rng.integers(0, 2))
        epi = int(rng.integers(0, 3) == 0)
        alias = use_res and bool(rng.integers(0, 2))            # residual = the output buffer (gradient accumulation)
        wino = k53 and not use_in and Cin % 4 == 0 and Cout >= 64 and bool(rng.integers(0, 2))
        xw = wino and bool(L.lib().aid_conv2d_wino_input_supported(Cin, Cout, T)) and bool(rng.integers(0, 2))
        g = torch.Generator().manual_seed(1000 + it)
        x = torch.randn(B, Cin, Fd, T, generator=g)
        w = torch.randn(Cout, Cin, KH, KW, generator=g) / math.sqrt(Cin * KH * KW)
        in_scale = (1.0 + 0.5 * torch.randn(B, Cin, generator=g)) if use_in else None
        out_scale = torch.randn(B, Cout, generator=g) if use_out else None
        res = torch.randn(B, Cout, Fd, T, generator=g) if use_res else None
        aux = torch.randn(B, Cout, Fd, T, generator=g) if epi else None
        asc = (1.0 + 0.3 * torch.randn(B, Cout, generator=g)) if epi else None
        alpha, res_scale = float(rng.choice([1.0, 0.7071])), float(rng.choice([1.0, 1.5]))
        h = x.double() * (in_scale.double()[:, :, None, None] if use_in else 1.0)
        if act:
            h = F.gelu(h)
        ref = F.conv2d(h, w.double(), padding="same", dilation=(dil, 1) if k53 else 1)
        if use_out:
            ref = ref * out_scale.double()[:, :, None, None]
        if epi:
            u = aux.double() * asc.double()[:, :, None, None]
            ref = ref * (0.5 * (1 + torch.erf(u / math.sqrt(2))) + u * torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi))
        if use_res:
            ref = ref + res_scale * res.double()
        ref = alpha * ref
        dev = lambda t: None if t is None else t.to(DEV)
        xd, wd = x.to(DEV), w.to(DEV)
        wp = L.pack_conv_weight(wd)
        ybig = torch.full((B, Cout + 8, Fd, T), 7.0, device=DEV)
        y = ybig[:, 4:4 + Cout]                                  # strided view (16-byte aligned channel offset)
        resd = dev(res)
        if alias:
            y.copy_(resd)
            resd = y
        p = L.Conv2dParams()
        keep = [dev(in_scale), dev(out_scale), dev(aux), dev(asc)]
        xin = xd
        if xw:
            xin = torch.empty(B, Cin, Fd, 6 * (T // 4), device=DEV)
            L.call("aid_scale_act", L.ScaleActParams(L.view4(xd), L.view4(xin), None, 0, B, Cin, Fd, T, 0, 1))
        p.x, p.y, p.res, p.aux = L.view4(xin), L.view4(y), L.view4(resd), L.view4(keep[2])
        p.wp = wp.data_ptr()
        p.in_scale, p.in_scale_ld = L.ptr(keep[0]), (0 if keep[0] is None else keep[0].stride(0))
        p.out_scale, p.out_scale_ld = L.ptr(keep[1]), (0 if keep[1] is None else keep[1].stride(0))
        p.aux_scale, p.aux_scale_ld = L.ptr(keep[3]), (0 if keep[3] is None else keep[3].stride(0))
        p.B, p.Cin, p.Cout, p.F, p.T = B, Cin, Cout, Fd, T
        p.Cin_pad, p.Cout_pad = wp.shape[1], wp.shape[2]
        p.KH, p.KW, p.dilF, p.act, p.epi = KH, KW, dil, act, epi
        p.alpha, p.res_scale = alpha, res_scale
        p.x_wino = int(xw)
        if wino:
            wpw = L.pack_conv_weight_wino(wd)
            p.wp_wino, p.wino_taps = wpw.data_ptr(), 30
        ws = None
        if Fd == 1 and not k53:
            ws = torch.empty(16 * B * Cout * T, device=DEV)
            p.ws, p.ws_bytes = ws.data_ptr(), ws.numel() * 4
        P = 0
        dws = None
        if wino and epi and not use_res:
            P = int(L.lib().aid_conv2d_dot_partials(B, Cin, Cout, Fd, T, dil, int(xw)))
            if P:
                dws = torch.full((B * 8 * (P + 1),), float("nan"), device=DEV, dtype=torch.float64)
                p.dot_ws, p.dot_n = dws.data_ptr(), P
        L.call("aid_conv2d", p)
        torch.cuda.synchronize()
        e = rel_l2(y.cpu().double(), ref)
        tag = f"it{it} {KH}x{KW} B{B} Cin{Cin} Cout{Cout} F{Fd} T{T} d{dil} in{int(use_in)} act{act} out{int(use_out)} res{int(use_res)} alias{int(alias)} epi{epi} wino{int(wino)} xw{int(xw)} P{P}"
        assert e < 2e-5, tag + f": rel-L2 {e:.2e}"
        assert float(ybig[:, :4].min()) == 7.0 and float(ybig[:, 4 + Cout:].min()) == 7.0, tag + ": wrote outside its channel slice"
        if P:
            got = dws[:B * 8 * P].cpu().reshape(B, 8, P).sum(-1)
            want = (y.cpu().double() * aux.double()).reshape(B, 8, Cout // 8, Fd, T).sum((2, 3, 4))
            scale = float((y.cpu().double().abs() * aux.double().abs()).reshape(B, 8, -1).sum(-1).max())
            assert float((got - want).abs().max()) < 1e-5 * scale, tag + ": dot partials"
        n_checked += 1
        paths.add((KH, wino, xw, Cin <= 8 or Cout <= 8, Fd == 1, bool(P)))
    print(f"conv dispatch fuzz: {n_checked} launches, {len(paths)} distinct path classes")
    assert n_checked >= 50 and len(paths) >= 8


@pytest.mark.parametrize("ratio", [(2, 1), (320, 147), (160, 147), (44100, 16000), (22050, 44100)])
def test_polyphase_resampler_vs_oracle(L, ratio):
    """aid_resample_poly through harness.resample against the oracle's restatement of torchaudio's default resampler, and the
    resample_batch case logic (utils/training_utils.py:140-212)."""
    from audio_inpainting_diffusion_amd.harness import resample, resample_batch
    from oracle import resample as O
    o, n = ratio
    x = _rand(3, 50001, seed=60)
    ref = O.resample(x.double(), o, n)
    got = resample(x.to(DEV), o, n).cpu()
    assert got.shape == ref.shape
    assert rel_l2(got, ref) < 2e-6
    if ratio == (2, 1):
        a = _rand(2, 200000, seed=61)
        out = resample_batch(a.to(DEV), torch.tensor([44100, 44100]), 22050, 65536).cpu()
        assert out.shape == (2, 65536) and rel_l2(out, O.resample_batch(a.double(), 44100, 22050, 65536)) < 2e-6
        out = resample_batch(a.to(DEV), torch.tensor([48000, 48000]), 22050, 65536).cpu()
        assert rel_l2(out, O.resample_batch(a.double(), 48000, 22050, 65536)) < 2e-6
        # mixed rates in one batch (training_utils.py:156-167): every row at its own rate (the reference fills item 0 only)
        mix = resample_batch(a.to(DEV), torch.tensor([48000, 44100]), 22050, 65536).cpu()
        assert rel_l2(mix[0:1], O.resample_batch(a[0:1].double(), 48000, 22050, 65536)) < 2e-6
        assert rel_l2(mix[1:2], O.resample_batch(a[1:2].double(), 44100, 22050, 65536)) < 2e-6
        # a row shorter than length_target raises (the reference's row assignment would), a strange rate is passed through with a warning (:165)
        with pytest.raises(ValueError, match="fewer than length_target"):
            resample_batch(a.to(DEV), torch.tensor([48000, 44100]), 22050, 10 ** 6)
        with pytest.warns(UserWarning, match="strange fs"):
            odd = resample_batch(a.to(DEV), torch.tensor([32000, 44100]), 22050, 65536).cpu()
        assert torch.equal(odd[0], a[0, :65536]) and torch.equal(odd[1], mix[1])
        with pytest.warns(UserWarning, match="strange fs"):          # the SAME rule when the whole batch has that rate (the reference's all()-tests fail there too: ADVICE r5)
            odd2 = resample_batch(a.to(DEV), torch.tensor([32000, 32000]), 22050, 65536).cpu()
        assert torch.equal(odd2, a[:, :65536])


WGRAD_CASES = [
    # B, Cin, Cout, F, T, KH, KW, dil, S
    (2, 64, 64, 12, 64, 5, 3, 2, 1),        # one chunk, chains of 6 rows
    (2, 64, 96, 10, 128, 5, 3, 1, 3),       # 96 output channels (half-empty second tile), splits cutting chains, two chunks
    (1, 40, 72, 9, 50, 5, 3, 4, 2),         # ragged channels, T % 4 != 0 (scalar loads), F % dil != 0
    (2, 32, 64, 6, 32, 5, 3, 16, 2),        # T = 32 (short chunk), dilation larger than F
    (1, 128, 128, 14, 200, 5, 3, 2, 5),     # T = 3 chunks + remainder, four tiles
    (3, 96, 8, 7, 64, 1, 1, 1, 2),          # 1x1, few output channels
    (2, 2, 64, 8, 96, 5, 3, 2, 8),          # pyramid projection 2 -> C, S = F
    (1, 64, 64, 28, 32, 5, 3, 8, 4),        # deep-level shape: several residue classes per split
    (2, 64, 192, 8, 128, 1, 1, 1, 3),       # 1x1, position-split 128 x 64 blocks (second Cout block half empty)
    (1, 96, 64, 5, 50, 1, 1, 1, 2),         # 1x1, 64 x 64 blocks, ragged Cin block, T % 4 != 0
    (2, 256, 128, 6, 64, 1, 1, 1, 1),       # 1x1, four Cin blocks
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv2d_wgrad_partials(L, case):
    """aid_conv2d_wgrad: per-(sample, split) partial weight gradients; their sum over the splits against the fp64 definition
    dW[b,co,ci,kh,kw] = alpha * sum_{f,t} gy[b,co,f,t] * x[b,ci,f+(kh-KH/2)*dil,t+kw-KW/2]."""
    B, Cin, Cout, Fd, T, KH, KW, dil, S = case
    gy = _rand(B, Cout, Fd, T, seed=70)
    x = _rand(B, Cin, Fd, T, seed=71)
    alpha = 0.75
    ph, pw = (KH // 2) * dil, KW // 2
    xp = F.pad(x.double(), (pw, pw, ph, ph))
    ref = torch.zeros(B, Cout, Cin, KH, KW, dtype=torch.float64)
    for kh in range(KH):
        for kw in range(KW):
            ref[:, :, :, kh, kw] = alpha * torch.einsum("boft,bift->boi", gy.double(), xp[:, :, kh * dil:kh * dil + Fd, kw:kw + T])
    # strided views (the network hands over channel slices of wider buffers)
    gbig = torch.full((B, Cout + 3, Fd, T), 7.0, device=DEV)
    xbig = torch.full((B, Cin + 5, Fd, T), 7.0, device=DEV)
    gd, xd = gbig[:, 2:2 + Cout], xbig[:, 4:4 + Cin]
    gd.copy_(gy.to(DEV)); xd.copy_(x.to(DEV))
    K = KH * KW
    P = torch.full((B * S * Cout * Cin * K + 16,), float("nan"), device=DEV)
    p = L.WgradParams(L.view4(gd), L.view4(xd), P.data_ptr(), B, Cin, Cout, Fd, T, KH, KW, dil, S, alpha)
    L.call("aid_conv2d_wgrad", p)
    torch.cuda.synchronize()
    assert bool(torch.isnan(P[B * S * Cout * Cin * K:]).all()), "wrote past the partial buffer"
    got = P[:B * S * Cout * Cin * K].cpu().double().reshape(B, S, Cout, KH, KW, Cin).sum(1).permute(0, 1, 4, 2, 3)   # [co][tap][ci] -> [co][ci][kh][kw]
    assert bool(torch.isfinite(got).all())
    assert rel_l2(got, ref) < 2e-6


@pytest.mark.parametrize("case", [(2, 3, 64, 64, 15), (3, 1, 96, 40, 15), (2, 2, 8, 300, 1), (1, 4, 256, 256, 15)])
def test_wgrad_reduce_and_gate_gradient(L, case):
    """aid_wgrad_reduce on partials in the kernel's [co][tap][ci] layout: dW[co,ci,tap] = sum_b gate*in_scale*sum_s P and
    dgate[b,co] = sum_{ci,tap} W*in_scale*sum_s P, against fp64."""
    B, S, Cout, Cin, K = case
    P = _rand(B, S, Cout, K, Cin, seed=80)
    W = _rand(Cout, Cin, K, seed=81)
    gate = _rand(B, Cout, seed=82)
    isc = 1.0 + 0.3 * _rand(B, Cin, seed=83)
    Ps = P.double().sum(1).permute(0, 1, 3, 2)                                  # [B, co, ci, tap]
    dW_ref = torch.einsum("bo,bi,boik->oik", gate.double(), isc.double(), Ps)
    dg_ref = torch.einsum("oik,bi,boik->bo", W.double(), isc.double(), Ps)
    Pd, Wd, gd_, id_ = P.to(DEV).contiguous(), W.to(DEV).contiguous(), gate.to(DEV), isc.to(DEV)
    dW = torch.full((Cout, Cin, K), 0.5, device=DEV)
    dg = torch.empty(B, Cout, device=DEV)
    p = L.WgradReduceParams(Pd.data_ptr(), Wd.data_ptr(), gd_.data_ptr(), gd_.stride(0), id_.data_ptr(), id_.stride(0),
                            dW.data_ptr(), dg.data_ptr(), dg.stride(0), B, S, Cout, Cin, K, 1)
    L.call("aid_wgrad_reduce", p)
    torch.cuda.synchronize()
    assert rel_l2(dW.cpu().double() - 0.5, dW_ref) < 2e-6
    assert rel_l2(dg.cpu().double(), dg_ref) < 2e-6


@pytest.mark.parametrize("shape", [(2, 16, 5, 12), (1, 64, 7, 256), (2, 8, 3, 10)])
def test_channel_dot(L, shape):
    B, C, Fd, T = shape
    u, v = _rand(B, C, Fd, T, seed=84), _rand(B, C, Fd, T, seed=85)
    big = torch.full((B, C + 2, Fd, T), 7.0, device=DEV)
    ud = big[:, 1:1 + C]
    ud.copy_(u.to(DEV))
    vd = v.to(DEV)
    out = torch.empty(B, C, device=DEV)
    L.call("aid_channel_dot", L.ChannelDotParams(L.view4(ud), L.view4(vd), out.data_ptr(), out.stride(0), B, C, Fd, T))
    torch.cuda.synchronize()
    ref = (u.double() * v.double()).sum((2, 3))
    assert float((out.cpu().double() - ref).abs().max()) < 1e-5 * float(ref.abs().max() + 1)


@pytest.mark.parametrize("xw", [1, 2])
@pytest.mark.parametrize("case", [(2, 64, 64, 24, 128, 2), (1, 128, 256, 16, 64, 4), (2, 96, 96, 32, 128, 2), (1, 64, 128, 56, 32, 8),
                                  # more than 256 F(8,3) tiles: the plain instances (the small cases run the K-group instances)
                                  (3, 128, 128, 64, 512, 2), (4, 96, 96, 128, 256, 2)])
def test_group_stats_from_conv_epilogue(L, case, xw):
    """aid_conv2d(stat_ws) on the row-shared F(4,3) (xw = 1) / F(8,3) (xw = 2) kernel + aid_group_stats(ws_n): the per-tile (sum, sum of squares)
    partials of the conv output replace the read pass; scale and (mean, 1/(std+eps)) agree with the plain two-kernel statistics of the same tensor."""
    B, Cin, Cout, Fd, T, dil = case
    if xw == 2 and not L.lib().aid_conv2d_wino8_supported(Cin, Cout, Fd, T, dil):
        pytest.skip("no F(8,3) tile fits this shape")
    P = int(L.lib().aid_conv2d_stat_partials(B, Cin, Cout, Fd, T, dil, xw))
    assert P > 0
    x = _rand(B, Cin, Fd, T, seed=90)
    w = _rand(Cout, Cin, 5, 3, seed=91, scale=1.0 / math.sqrt(Cin * 15))
    res = _rand(B, Cout, Fd, T, seed=92)
    gate = _rand(B, Cout, seed=93)
    xd, wd = x.to(DEV), w.to(DEV)
    xv = torch.empty(B, Cin, Fd, (10 * (T // 8)) if xw == 2 else (6 * (T // 4)), device=DEV)
    L.call("aid_scale_act", L.ScaleActParams(L.view4(xd), L.view4(xv), None, 0, B, Cin, Fd, T, 0, xw))
    wp, wpw = L.pack_conv_weight(wd), (L.pack_conv_weight_wino8(wd) if xw == 2 else L.pack_conv_weight_wino(wd))
    y = torch.empty(B, Cout, Fd, T, device=DEV)
    resd, gd = res.to(DEV), gate.to(DEV)
    ws = torch.full((B * 8 * P * 2 + 4,), float("nan"), device=DEV, dtype=torch.float64)
    p = L.Conv2dParams()
    p.x, p.y, p.res, p.aux = L.view4(xv), L.view4(y), L.view4(resd), L.view4(None)
    p.wp, p.wp_wino, p.wino_taps, p.x_wino = wp.data_ptr(), wpw.data_ptr(), wpw.shape[0], xw
    p.out_scale, p.out_scale_ld = gd.data_ptr(), gd.stride(0)
    p.B, p.Cin, p.Cout, p.F, p.T = B, Cin, Cout, Fd, T
    p.Cin_pad, p.Cout_pad = wp.shape[1], wp.shape[2]
    p.KH, p.KW, p.dilF, p.act, p.epi = 5, 3, dil, 0, 0
    p.alpha, p.res_scale = 1 / math.sqrt(2), 1.0
    p.stat_ws, p.stat_n = ws.data_ptr(), P
    L.call("aid_conv2d", p)
    torch.cuda.synchronize()
    assert bool(torch.isnan(ws[B * 8 * P * 2:]).all()) and bool(torch.isfinite(ws[:B * 8 * P * 2]).all())
    gamma = (1.0 + 0.1 * _rand(Cout, seed=94)).to(DEV)
    mod = (0.2 * _rand(B, Cout, seed=95)).to(DEV)
    out = []
    for n_, w_ in ((P, ws), (0, torch.empty(B * 8 * L.AID_STATS_SPLIT * 2, device=DEV, dtype=torch.float64))):
        scale = torch.empty(B, Cout, device=DEV)
        st = torch.empty(B, 8, 2, device=DEV)
        sp = L.GroupStatsParams(L.view4(y), B, Cout, Fd, T, 8, gamma.data_ptr(), mod.data_ptr(), mod.stride(0), 1e-7, scale.data_ptr(),
                                st.data_ptr(), w_.data_ptr(), n_)
        L.call("aid_group_stats", sp)
        torch.cuda.synchronize()
        out.append((scale.cpu(), st.cpu()))
    assert rel_l2(out[0][0], out[1][0]) < 1e-6
    assert float((out[0][1][..., 0] - out[1][1][..., 0]).abs().max()) < 1e-6 * float(out[1][1][..., 0].abs().max() + 1)
    assert rel_l2(out[0][1][..., 1], out[1][1][..., 1]) < 1e-6
    # fin_mode = 1: the last tile of each sample folds the partials itself -- the same bits as aid_group_stats(ws_n = P), run after run, counters back to zero,
    # with and without the saved statistics / the modulation
    assert L.lib().aid_conv2d_fin_supported(B, Cin, Cout, Fd, T, dil, xw)
    cnt = torch.zeros(B + 2, device=DEV, dtype=torch.int32)
    cnt[B:] = 7
    for rep in range(3):
        scale = torch.full((B, Cout), float("nan"), device=DEV)
        st = torch.full((B, 8, 2), float("nan"), device=DEV)
        ws.fill_(float("nan"))
        y2 = torch.full_like(y, float("nan"))
        p.y = L.view4(y2)
        p.fin_mode, p.fin_count, p.fin_gamma, p.fin_eps, p.fin_scale = 1, cnt.data_ptr(), gamma.data_ptr(), 1e-7, scale.data_ptr()
        p.fin_mod, p.fin_mod_ld = (mod.data_ptr(), mod.stride(0)) if rep != 1 else (None, 0)
        p.fin_stats = st.data_ptr() if rep != 2 else None
        L.call("aid_conv2d", p)
        torch.cuda.synchronize()
        assert torch.equal(y2, y) and cnt[:B].abs().sum().item() == 0 and int(cnt[B]) == 7
        want = out[0][0] if rep != 1 else out[0][0] / (1.0 + mod.cpu())
        assert torch.equal(scale.cpu(), out[0][0]) if rep != 1 else rel_l2(scale.cpu(), want) < 1e-6
        if rep != 2:
            assert torch.equal(st.cpu(), out[0][1])
    # the option is refused where no kernel honours it
    p.x_wino, p.x = 0, L.view4(xd)
    p.stat_ws, p.stat_n = None, 0
    assert L.lib().aid_conv2d(ctypes.addressof(p), None) != 0 and b"fin_mode" in L.lib().aid_last_error()


@pytest.mark.parametrize("shape", [(64, 64, 5, 3), (96, 128, 5, 3), (2, 96, 5, 3), (256, 40, 1, 1), (192, 96, 1, 3)])
def test_pack_conv_weight_kernel_matches_the_torch_packs(L, shape):
    """aid_pack_conv_weight: the tap-major pack, its input-gradient operator and (5x3) both F(4,3) packs in one launch, bit-identical to
    _lib.pack_conv_weight / pack_conv_weight_wino."""
    co, ci, kh, kw = shape
    w = _rand(co, ci, kh, kw, seed=97).to(DEV)
    wino = (kh, kw) == (5, 3) and co >= 64 and ci >= 64
    (cip, cop), (cipT, copT) = L.pack_dims(ci, co), L.pack_dims(co, ci)
    outs = [torch.full((kh * kw, cip, cop), float("nan"), device=DEV), torch.full((kh * kw, cipT, copT), float("nan"), device=DEV),
            torch.full((30, cip, cop), float("nan"), device=DEV) if wino else None, torch.full((30, cipT, copT), float("nan"), device=DEV) if wino else None]
    p = L.PackConvWeightParams(w.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(), L.ptr(outs[2]), L.ptr(outs[3]), co, ci, kh, kw, cip, cop, cipT, copT)
    L.call("aid_pack_conv_weight", p)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], L.pack_conv_weight(w)) and torch.equal(outs[1], L.pack_conv_weight(w, transpose=True))
    if wino:
        assert torch.equal(outs[2], L.pack_conv_weight_wino(w)) and torch.equal(outs[3], L.pack_conv_weight_wino(w, transpose=True))


WGRAD_WINO_CASES = [
    # B, Cin, Cout, F, T, dil, S
    (2, 64, 64, 12, 64, 2, 1),
    (2, 96, 96, 20, 128, 4, 3),          # half-empty second Cout tile, two chunks, splits cutting chains
    (1, 40, 72, 9, 48, 4, 2),            # ragged channels, a 12-group chunk, F % dil != 0
    (1, 128, 64, 28, 32, 8, 4),          # T = 32: 8 groups per chunk, several residue classes per split
    (3, 32, 64, 6, 80, 16, 2),           # dilation larger than F
]


@pytest.mark.parametrize("case", WGRAD_WINO_CASES)
def test_conv2d_wgrad_winograd_form(L, case):
    """aid_wino_gy + aid_scale_act(wino=1) -> aid_conv2d_wgrad(wino=1) -> aid_wgrad_reduce(wino=1): the F(4,3) form of the 5x3 weight gradient
    (half the MFMAs) and the gate gradient recovered from its U-domain partials, against the fp64 definition."""
    B, Cin, Cout, Fd, T, dil, S = case
    gy, x = _rand(B, Cout, Fd, T, seed=110), _rand(B, Cin, Fd, T, seed=111)
    w = _rand(Cout, Cin, 5, 3, seed=112, scale=1.0 / math.sqrt(Cin * 15))
    gate, isc = _rand(B, Cout, seed=113), 1.0 + 0.3 * _rand(B, Cin, seed=114)
    alpha = 0.75
    xp = F.pad(x.double(), (1, 1, 2 * dil, 2 * dil))
    dWb = torch.zeros(B, Cout, Cin, 5, 3, dtype=torch.float64)
    for kh in range(5):
        for kw in range(3):
            dWb[:, :, :, kh, kw] = alpha * torch.einsum("boft,bift->boi", gy.double(), xp[:, :, kh * dil:kh * dil + Fd, kw:kw + T])
    dW_ref = torch.einsum("bo,bi,boihw->oihw", gate.double(), isc.double(), dWb)
    dg_ref = torch.einsum("oihw,bi,boihw->bo", w.double(), isc.double(), dWb)
    gd, xd, wd = gy.to(DEV), x.to(DEV), w.to(DEV)
    G = T // 4
    gyw = torch.empty(B, Cout, Fd, 6 * G, device=DEV)
    xw = torch.empty(B, Cin, Fd, 6 * G, device=DEV)
    L.call("aid_wino_gy", L.WinoGyParams(L.view4(gd), L.view4(gyw), B, Cout, Fd, T))
    L.call("aid_scale_act", L.ScaleActParams(L.view4(xd), L.view4(xw), None, 0, B, Cin, Fd, T, 0, 1))
    P = torch.full((B * S * Cout * Cin * 30 + 16,), float("nan"), device=DEV)
    L.call("aid_conv2d_wgrad", L.WgradParams(L.view4(gyw), L.view4(xw), P.data_ptr(), B, Cin, Cout, Fd, T, 5, 3, dil, S, alpha, 1))
    torch.cuda.synchronize()
    assert bool(torch.isnan(P[B * S * Cout * Cin * 30:]).all()) and bool(torch.isfinite(P[:B * S * Cout * Cin * 30]).all())
    wpw = L.pack_conv_weight_wino(wd)
    gated, iscd = gate.to(DEV), isc.to(DEV)
    dW = torch.zeros(Cout, Cin, 15, device=DEV)
    dg = torch.empty(B, Cout, device=DEV)
    rp = L.WgradReduceParams(P.data_ptr(), wd.data_ptr(), gated.data_ptr(), gated.stride(0), iscd.data_ptr(), iscd.stride(0), dW.data_ptr(), dg.data_ptr(),
                             dg.stride(0), B, S, Cout, Cin, 15, 0, 1, wpw.data_ptr(), wpw.shape[1], wpw.shape[2])
    L.call("aid_wgrad_reduce", rp)
    torch.cuda.synchronize()
    e1, e2 = rel_l2(dW.cpu().double().reshape(Cout, Cin, 5, 3), dW_ref), rel_l2(dg.cpu().double(), dg_ref)
    print(f"F(4,3) weight gradient: dW rel-L2 {e1:.2e}, dgate rel-L2 {e2:.2e}")
    assert e1 < 1e-5 and e2 < 1e-5


@pytest.mark.parametrize("case", [(2, 64, 64, 192, 16, 64, True), (1, 96, 96, 256, 8, 256, False), (3, 128, 128, 64, 9, 32, True), (2, 16, 48, 96, 12, 16, False)])
def test_conv1x1_with_the_k_axis_in_two_tensors(L, case):
    """aid_conv2d x2 / Cin1: y = alpha * (res + W[:, :Cin1] x + W[:, Cin1:] x2) -- the merged input gradient of a ResnetBlock's proj_in and
    res_conv (unet...py:414-415, :488-491) -- against F.conv2d on the concatenated input (fp64 reference)."""
    B, c1, c2, cout, Fd, T, acc = case
    assert L.lib().aid_conv2d_x2_supported(c1 + c2, c1, cout, Fd, T)
    g = torch.Generator().manual_seed(7)
    big = torch.randn(B, c1 + 8, Fd, T, generator=g)                     # x is a channel slice of a wider buffer: strided view
    x1, x2 = big[:, 4:4 + c1], torch.randn(B, c2, Fd, T, generator=g)
    w = torch.randn(cout, c1 + c2, 1, 1, generator=g) / math.sqrt(c1 + c2)
    res = torch.randn(B, cout, Fd, T, generator=g)
    ref = F.conv2d(torch.cat([x1, x2], 1).double(), w.double())
    if acc:
        ref = ref + res.double()
    bigd = big.to(DEV)
    x1d, x2d, yd = bigd[:, 4:4 + c1], x2.to(DEV), res.to(DEV).clone()
    wp = L.pack_conv_weight(w.to(DEV))
    p = L.Conv2dParams()
    p.x, p.x2, p.Cin1, p.y, p.res, p.aux = L.view4(x1d), L.view4(x2d), c1, L.view4(yd), L.view4(yd if acc else None), L.view4(None)
    p.wp = wp.data_ptr()
    p.B, p.Cin, p.Cout, p.F, p.T = B, c1 + c2, cout, Fd, T
    p.Cin_pad, p.Cout_pad = wp.shape[1], wp.shape[2]
    p.KH, p.KW, p.dilF, p.act, p.epi = 1, 1, 1, 0, 0
    p.alpha, p.res_scale = 1.0, 1.0
    L.call("aid_conv2d", p)
    assert L.lib().aid_last_kernel().decode() in ("conv11_dma_kernel", "conv11_rs_kernel")
    e = rel_l2(yd.cpu(), ref)
    assert e < 1e-5, e
    # unsupported shapes are refused loudly, not routed elsewhere
    p.Cin1 = c1 + 1
    with pytest.raises(L.AidError):
        L.call("aid_conv2d", p)


RS_CASES = [
    # B, Cin, Cout, F, T, act (GELU prologue + in_scale), in_scale, epi (dGELU(aux)), res + out_scale, dot partials, Cin1 (x2), strided x
    (2, 64, 64, 8, 64, 1, 1, 0, 1, 0, 0, 1),
    (1, 96, 96, 6, 32, 0, 0, 1, 0, 1, 0, 0),        # a 64-position tile spans two rows
    (3, 128, 96, 4, 48, 0, 1, 1, 0, 1, 0, 0),       # T not a power of two
    (2, 256, 96, 2, 96, 0, 0, 0, 0, 0, 0, 1),       # K = 256: eight blocks of 16 k-steps
    (2, 128, 128, 8, 64, 0, 0, 1, 0, 1, 0, 0),      # 128 output channels: the direct-to-LDS tile kernel keeps the layer
    (2, 128, 64, 8, 64, 0, 0, 0, 1, 0, 0, 0),       # ... and 64 output channels with K > 64
    (1, 192, 96, 8, 16, 0, 0, 0, 1, 0, 64, 0),      # K axis in two tensors
    (8, 64, 96, 16, 128, 0, 0, 0, 1, 0, 0, 0),      # more tiles than waves: the persistent loop
    (1, 64, 64, 2, 2048, 1, 1, 0, 0, 0, 0, 0),
]


@pytest.mark.parametrize("case", RS_CASES)
def test_conv1x1_register_streamed_kernel(L, case):
    """conv11_rs_kernel (K <= 256 1x1 layers: weights resident in LDS, activations HBM -> registers) with every prologue / epilogue option
    against an fp64 reference of  y = alpha * (res_scale * res + out_scale * dGELU(aux * aux_scale) * (W (act(x * in_scale)))) ."""
    B, Cin, Cout, Fd, T, act, isc_on, epi, res_on, dot_on, cin1, strided = case
    g = torch.Generator().manual_seed(11)
    rn = lambda *s: torch.randn(*s, generator=g)
    big = rn(B, Cin + 8, Fd, T)
    x = big[:, 4:4 + Cin] if strided else big[:, :Cin].contiguous()
    w = rn(Cout, Cin, 1, 1) / math.sqrt(Cin)
    isc = (1.0 + 0.5 * rn(B, Cin)) if isc_on else None
    osc = (1.0 + 0.3 * rn(B, Cout)) if (res_on or epi) else None
    res = rn(B, Cout, Fd, T) if res_on else None
    aux, asc = (rn(B, Cout, Fd, T), 1.0 + 0.3 * rn(B, Cout)) if epi else (None, None)
    alpha, res_scale = 0.7, 1.5
    h = x.double()
    if isc is not None:
        h = h * isc.double()[:, :, None, None]
    if act:
        h = F.gelu(h)
    ref = F.conv2d(h, w.double())
    if osc is not None:
        ref = ref * osc.double()[:, :, None, None]
    if epi:
        u = aux.double() * asc.double()[:, :, None, None]
        ref = ref * (0.5 * (1 + torch.erf(u / math.sqrt(2))) + u * torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi))
    if res is not None:
        ref = ref + res_scale * res.double()
    ref = alpha * ref
    bigd = big.to(DEV)
    xd = bigd[:, 4:4 + Cin] if strided else bigd[:, :Cin].contiguous()
    dv = lambda t: None if t is None else t.to(DEV)
    iscd, oscd, resd, auxd, ascd = dv(isc), dv(osc), dv(res), dv(aux), dv(asc)
    wp = L.pack_conv_weight(w.to(DEV))
    y = torch.full((B, Cout + 2, Fd, T), 7.0, device=DEV)
    p = L.Conv2dParams()
    if cin1:
        x1d, x2d = xd[:, :cin1].contiguous(), xd[:, cin1:].contiguous()
        p.x, p.x2, p.Cin1 = L.view4(x1d), L.view4(x2d), cin1
    else:
        p.x = L.view4(xd)
    p.y, p.res, p.aux = L.view4(y[:, 1:1 + Cout]), L.view4(resd), L.view4(auxd)
    p.wp = wp.data_ptr()
    p.in_scale, p.in_scale_ld = L.ptr(iscd), (0 if iscd is None else iscd.stride(0))
    p.out_scale, p.out_scale_ld = L.ptr(oscd), (0 if oscd is None else oscd.stride(0))
    p.aux_scale, p.aux_scale_ld = L.ptr(ascd), (0 if ascd is None else ascd.stride(0))
    p.B, p.Cin, p.Cout, p.F, p.T = B, Cin, Cout, Fd, T
    p.Cin_pad, p.Cout_pad = wp.shape[1], wp.shape[2]
    p.KH, p.KW, p.dilF, p.act, p.epi = 1, 1, 1, act, epi
    p.alpha, p.res_scale = alpha, res_scale
    if dot_on:
        P = int(L.lib().aid_conv2d_dot_partials_1x1(B, Cin, Cout, Fd, T))
        assert P == (Fd * T // 64 if (Cout == 96 or (Cout == 64 and Cin <= 64)) else Fd * T // 256)
        ws = torch.full((B * 8 * P + 4,), float("nan"), device=DEV, dtype=torch.float64)
        p.dot_ws, p.dot_n = ws.data_ptr(), P
    L.call("aid_conv2d", p)
    assert L.lib().aid_last_kernel().decode() == ("conv11_rs_kernel" if (Cout == 96 or (Cout == 64 and Cin <= 64)) else "conv11_dma_kernel")
    torch.cuda.synchronize()
    yc = y[:, 1:1 + Cout].cpu().double()
    e = rel_l2(yc, ref)
    assert e < 1e-5, e
    assert float(y[:, 0].min()) == 7.0 and float(y[:, 1 + Cout].max()) == 7.0, "wrote outside its channel slice"
    if dot_on:
        assert bool(torch.isnan(ws[B * 8 * P:]).all())
        got = ws[:B * 8 * P].cpu().reshape(B, 8, P).sum(-1)
        want = (yc * aux.double()).reshape(B, 8, Cout // 8, Fd, T).sum((2, 3, 4))
        assert float((got - want).abs().max()) < 1e-5 * float((yc.abs() * aux.double().abs()).reshape(B, 8, -1).sum(-1).max())


def test_qk_gemm_is_bit_identical_across_batch_sizes(L):
    """The qk projections take the skinny GEMM when N = B*T <= 128 and direct-to-LDS tiles above; both split K by the same batch-independent
    partition and add the same k-pairs in the same order, so a segment's result does not depend on the batch it is evaluated in."""
    Cin, Cout, T = 2560, 5120, 64
    w = _rand(Cout, Cin, 1, 1, seed=71, scale=1.0 / math.sqrt(Cin)).to(DEV)
    wp = L.pack_conv_weight(w)
    x8 = _rand(8, Cin, 1, T, seed=72).to(DEV)
    outs = {}
    for B in (1, 2, 8):
        xb = x8[:B].contiguous()
        y = torch.empty(B, Cout, 1, T, device=DEV)
        ws = torch.empty(8 * B * Cout * T, device=DEV)
        p = L.Conv2dParams()
        p.x, p.y, p.res, p.aux = L.view4(xb), L.view4(y), L.view4(None), L.view4(None)
        p.wp = wp.data_ptr()
        p.B, p.Cin, p.Cout, p.F, p.T = B, Cin, Cout, 1, T
        p.Cin_pad, p.Cout_pad = wp.shape[1], wp.shape[2]
        p.KH, p.KW, p.dilF, p.act, p.epi = 1, 1, 1, 0, 0
        p.alpha, p.res_scale = 1.0, 1.0
        p.ws, p.ws_bytes = ws.data_ptr(), ws.numel() * 4
        L.call("aid_conv2d", p)
        outs[B] = (y, L.lib().aid_last_kernel().decode())
    assert outs[1][1].startswith("gemm_skinny") and outs[2][1].startswith("gemm_skinny") and outs[8][1].startswith("conv11_dma")
    assert torch.equal(outs[1][0], outs[8][0][:1]) and torch.equal(outs[2][0], outs[8][0][:2])
