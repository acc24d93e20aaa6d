#!/usr/bin/env python
"""Per-layer A/B of the row-shared 5x3 kernels: F(4,3) (x_wino = 1) against F(8,3) (x_wino = 2) on the shapes of the shipped 22.05 kHz network.
   python tools/wino8_probe.py [B ...]      prints time per launch, algorithmic TFLOP/s and the rel-L2 of both against each other."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_inpainting_diffusion_amd import _lib as L

SHAPES = [  # C, F, T, dilations
    (64, 64, 2048, (1, 2)), (96, 128, 1024, (1, 4)), (96, 192, 512, (1, 8)), (128, 256, 256, (1, 16)),
    (128, 320, 128, (1, 4, 16)), (256, 384, 64, (1, 8, 32)), (256, 448, 32, (1, 4, 16, 32, 64)),
]


def run(B, C, F, T, dil, form, reps=10, epi=0, sk=False):
    dev = "cuda"
    x = torch.randn(B, C, F, T, device=dev)
    y = torch.empty(B, C, F, T, device=dev)
    w = torch.randn(C, C, 5, 3, device=dev) / math.sqrt(C * 15)
    wp = L.pack_conv_weight(w)
    wpw = L.pack_conv_weight_wino8(w) if form == 8 else L.pack_conv_weight_wino(w)
    osc = torch.randn(B, C, device=dev)
    cols = 10 * (T // 8) if form == 8 else 6 * (T // 4)
    xv = torch.empty(B, C, F, cols, device=dev)
    L.call("aid_scale_act", L.ScaleActParams(L.view4(x), L.view4(xv), None, 0, B, C, F, T, 0, 2 if form == 8 else 1))
    p = L.Conv2dParams()
    p.x, p.y, p.res, p.aux = L.view4(xv), L.view4(y), L.view4(x), L.view4(x if epi else None)
    if epi:
        p.aux_scale, p.aux_scale_ld = osc.data_ptr(), osc.stride(0)
    p.wp, p.wp_wino, p.wino_taps, p.x_wino = wp.data_ptr(), wpw.data_ptr(), wpw.shape[0], 2 if form == 8 else 1
    p.out_scale, p.out_scale_ld = osc.data_ptr(), osc.stride(0)
    p.B, p.Cin, p.Cout, p.F, p.T = B, C, C, F, T
    p.Cin_pad, p.Cout_pad = wp.shape[1], wp.shape[2]
    p.KH, p.KW, p.dilF, p.act, p.epi = 5, 3, dil, 0, epi
    p.alpha, p.res_scale = 1 / math.sqrt(2), 1.0
    if sk:           # the stream-K instance left the library in round 5 (profiles/r05_streamk_variant.patch.txt restores it)
        return None, None, "plain tiles"
    torch.manual_seed(0)
    for _ in range(2):
        L.call("aid_conv2d", p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.call("aid_conv2d", p)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, y, L.lib().aid_last_kernel().decode()


def main():
    Bs = [int(a) for a in sys.argv[1:]] or [8]
    epi = int(os.environ.get("PROBE_EPI", "0"))
    for B in Bs:
        t4s = t8s = tks = 0.0
        for C, F, T, dils in SHAPES:
            for dil in dils:
                torch.manual_seed(1)
                t4, y4, k4 = run(B, C, F, T, dil, 4, epi=epi)
                form = 8 if L.lib().aid_conv2d_wino8_supported(C, C, F, T, dil) else 4
                pick = int(L.lib().aid_conv2d_wino_form(B, C, C, F, T, dil))
                fl = 2.0 * B * F * T * C * C * 15
                if form != 8:
                    print(f"B{B} C{C} F{F} T{T} d{dil}: F(4,3) {t4:7.1f} us {fl / t4 / 1e6:6.1f} TF/s   [library keeps F(4,3)]")
                    t4s += t4; t8s += t4; tks += t4
                    continue
                torch.manual_seed(1)
                t8, y8, k8 = run(B, C, F, T, dil, 8, epi=epi)
                err = float((y8 - y4).norm() / y4.norm())
                torch.manual_seed(1)
                tk, yk, kk = run(B, C, F, T, dil, 8, epi=epi, sk=True)
                t4s += t4; t8s += t8; tks += (tk if tk is not None else t8)
                sks = "stream-K: plain tiles kept" if tk is None else f"stream-K {tk:7.1f} us {fl / tk / 1e6:6.1f} TF/s x{t8 / tk:.3f} diff {float((yk - y8).norm() / y8.norm()):.1e}"
                print(f"B{B} C{C} F{F} T{T} d{dil}: F(4,3) {t4:7.1f} us {fl / t4 / 1e6:6.1f} TF/s | F(8,3) {t8:7.1f} us {fl / t8 / 1e6:6.1f} TF/s  x{t4 / t8:.3f}  diff {err:.1e}  {k8} | {sks} | library picks F({pick},3)")
        print(f"B{B} sum: F(4,3) {t4s / 1e3:.2f} ms, F(8,3) {t8s / 1e3:.2f} ms (x{t4s / t8s:.3f}), F(8,3) with stream-K where taken {tks / 1e3:.2f} ms (x{t4s / tks:.3f})")


if __name__ == "__main__":
    main()
