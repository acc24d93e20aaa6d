#!/usr/bin/env python
"""aid_conv2d_wgrad on the main 5x3 shapes of the full-size network at the training batch (4): time per launch, direct-form TFLOP/s
and fraction of the fp32 MFMA peak (157.3).  usage: wgrad_probe.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_inpainting_diffusion_amd import _lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda")
SHAPES = [  # C, F, T, dil
    (64, 64, 1024, 1), (96, 128, 1024, 2), (96, 192, 512, 4), (128, 256, 256, 4), (128, 320, 128, 8),
    (256, 384, 64, 8), (256, 448, 32, 4), (256, 448, 32, 32),
]
for C, F, T, dil in SHAPES:
    gy = torch.randn(B, C, F, T, device=dev)
    x = torch.randn(B, C, F, T, device=dev)
    tiles = int(L.lib().aid_conv2d_wgrad_tiles(C, C, 5, 3, 1))
    S = max(1, min(F, 256 // (tiles * B)))
    P = torch.empty(B * S * C * C * 15, device=dev)
    p = L.WgradParams(L.view4(gy), L.view4(x), P.data_ptr(), B, C, C, F, T, 5, 3, dil, S, 1.0)
    for _ in range(2):
        L.call("aid_conv2d_wgrad", p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n):
        L.call("aid_conv2d_wgrad", p)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * B * F * T * C * C * 15
    print(f"wgrad C={C:3d} F={F:3d} T={T:4d} dil={dil:2d} S={S:3d}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:6.1f} TFLOP/s  {fl / ms / 1e9 / 157.3:.3f} of peak", end="")
    # F(4,3) form: operands in the Winograd domain (the two transform passes timed separately)
    Gq = T // 4
    gyw = torch.empty(B, C, F, 6 * Gq, device=dev)
    xw = torch.empty(B, C, F, 6 * Gq, device=dev)
    Pw = torch.empty(B * S * C * C * 30, device=dev)
    gp = L.WinoGyParams(L.view4(gy), L.view4(gyw), B, C, F, T)
    sp = L.ScaleActParams(L.view4(x), L.view4(xw), None, 0, B, C, F, T, 0, 1)
    wp = L.WgradParams(L.view4(gyw), L.view4(xw), Pw.data_ptr(), B, C, C, F, T, 5, 3, dil, S, 1.0, 1)
    out = []
    for name, q in (("aid_wino_gy", gp), ("aid_scale_act", sp), ("aid_conv2d_wgrad", wp)):
        for _ in range(2):
            L.call(name, q)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            L.call(name, q)
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / n)
    print(f"   | F(4,3): {out[2] * 1e3:8.1f} us ({fl / out[2] / 1e9:6.1f} algorithmic TFLOP/s) + gy transform {out[0] * 1e3:6.1f} us + x transform {out[1] * 1e3:6.1f} us")
