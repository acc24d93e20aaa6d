#!/usr/bin/env python
"""Achieved HBM bandwidth of the element-wise / reduction kernels on the layer shapes of the 22 kHz network (B=8)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_inpainting_diffusion_amd import _lib as L

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

B = 8
for (C, F, T) in [(64, 64, 2048), (96, 128, 1024), (96, 192, 512), (128, 256, 256), (128, 320, 128), (256, 384, 64), (256, 448, 32)]:
    n = B * C * F * T
    x, gd, gy, out = (torch.randn(B, C, F, T, device="cuda") for _ in range(4))
    sc = torch.rand(B, C, device="cuda") + 0.5
    stats = torch.rand(B, 8, 2, device="cuda")
    ws = torch.zeros(B * 8 * (L.AID_STATS_SPLIT * 2 + 2), device="cuda", dtype=torch.float64)
    xv = torch.empty(B, C, F, 6 * (T // 4), device="cuda")
    scale = torch.empty(B, C, device="cuda"); gamma = torch.ones(C, device="cuda")
    res = []
    p = L.NormBwdParams(L.view4(gd), L.view4(x), L.view4(gy), L.view4(out), B, C, F, T, 8, stats.data_ptr(), ws.data_ptr(), 1e-7, 0.7, 0, 0)
    ms = timeit(lambda: L.call("aid_norm_bwd", p)); res.append("norm_bwd %.0f us %.2f TB/s" % (ms * 1e3, 16 * n / ms / 1e9))
    p2 = L.ScaleActParams(L.view4(x), L.view4(xv), sc.data_ptr(), sc.stride(0), B, C, F, T, 1, 1)
    ms = timeit(lambda: L.call("aid_scale_act", p2)); res.append("scale_act_wino %.0f us %.2f TB/s" % (ms * 1e3, 10 * n / ms / 1e9))
    p3 = L.ScaleActParams(L.view4(x), L.view4(out), sc.data_ptr(), sc.stride(0), B, C, F, T, 1, 0)
    ms = timeit(lambda: L.call("aid_scale_act", p3)); res.append("scale_act %.0f us %.2f TB/s" % (ms * 1e3, 8 * n / ms / 1e9))
    p4 = L.GroupStatsParams(L.view4(x), B, C, F, T, 8, gamma.data_ptr(), None, 0, 1e-7, scale.data_ptr(), stats.data_ptr(), ws.data_ptr())
    ms = timeit(lambda: L.call("aid_group_stats", p4)); res.append("group_stats %.0f us %.2f TB/s" % (ms * 1e3, 4 * n / ms / 1e9))
    p5 = L.Add2Params(L.view4(x), L.view4(gd), L.view4(out), B, C, F, T, 0.7, 0.7)
    ms = timeit(lambda: L.call("aid_add2", p5)); res.append("add2 %.0f us %.2f TB/s" % (ms * 1e3, 12 * n / ms / 1e9))
    print("C%d F%d T%d (%.0f MB): " % (C, F, T, 4 * n / 1e6) + " | ".join(res))
