#!/usr/bin/env python
"""EXPERIMENT: the row-shared F(4,3) kernel on the main layer shapes in ONE process (Winograd-domain input, gate + residual epilogue);
the kernel variant is chosen by AID_W4R_MODE / AID_W4R_RAW / AID_W4R_IL (read once per process).  usage: w4r_ab.py [reps]"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_inpainting_diffusion_amd import _lib as L

SHAPES = [(8, 256, 448, 32, 4), (8, 256, 448, 32, 32), (8, 256, 384, 64, 4), (8, 128, 320, 128, 4), (8, 128, 256, 256, 4), (8, 96, 192, 512, 4),
          (8, 96, 128, 1024, 2), (8, 64, 64, 2048, 1), (8, 64, 128, 1024, 2), (3, 256, 448, 32, 4), (3, 128, 320, 128, 4), (1, 256, 448, 32, 4), (1, 128, 320, 128, 4)]
if os.environ.get("W4R_SHAPES"):
    SHAPES = [tuple(int(v) for v in s.split(",")) for s in os.environ["W4R_SHAPES"].split(";")]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
tag = "mode%s raw%s il%s" % (os.environ.get("AID_W4R_MODE", "0"), os.environ.get("AID_W4R_RAW", "1"), os.environ.get("AID_W4R_IL", "1"))
tot = 0.0
for (B, C, F, T, dil) in SHAPES:
    x = torch.randn(B, C, F, T, device="cuda")
    y = torch.empty(B, C, F, T, device="cuda")
    w = torch.randn(C, C, 5, 3, device="cuda") / math.sqrt(C * 15)
    wp, wpw = L.pack_conv_weight(w), L.pack_conv_weight_wino(w)
    osc = torch.randn(B, C, device="cuda")
    xv = torch.empty(B, C, F, 6 * (T // 4), device="cuda")
    L.call("aid_scale_act", L.ScaleActParams(L.view4(x), L.view4(xv), None, 0, B, C, F, T, 0, 1))
    p = L.Conv2dParams()
    p.x, p.y, p.res, p.aux = L.view4(xv), L.view4(y), L.view4(x), L.view4(None)
    p.wp, p.wp_wino, p.wino_taps, p.x_wino = wp.data_ptr(), wpw.data_ptr(), 30, 1
    p.out_scale, p.out_scale_ld = osc.data_ptr(), osc.stride(0)
    p.B, p.Cin, p.Cout, p.F, p.T = B, C, C, F, T
    p.Cin_pad, p.Cout_pad = wp.shape[1], wp.shape[2]
    p.KH, p.KW, p.dilF, p.act, p.epi = 5, 3, dil, 0, 0
    p.alpha, p.res_scale = 1 / math.sqrt(2), 1.0
    for _ in range(3):
        L.call("aid_conv2d", p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.call("aid_conv2d", p)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * B * F * T * C * C * 15
    tot += ms
    print(f"{tag}: B{B} C{C} F{F} T{T} d{dil}: {ms:.4f} ms  {fl / ms / 1e9:6.1f} alg TF/s  exec frac {fl / 2 / ms / 1e9 / 157.3:.3f}")
print(f"{tag}: sum {tot:.3f} ms")
