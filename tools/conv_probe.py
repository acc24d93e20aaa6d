#!/usr/bin/env python
"""Micro-benchmark of aid_conv2d on one shape (used with rocprofv3 --pmc for MFMA / LDS / HBM counters).
   python tools/conv_probe.py B Cin Cout F T KH KW dil [reps] [act]"""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_inpainting_diffusion_amd import _lib as L


def main():
    a = [int(v) for v in sys.argv[1:9]]
    B, Cin, Cout, F, T, KH, KW, dil = a
    reps = int(sys.argv[9]) if len(sys.argv) > 9 else 10
    act = int(sys.argv[10]) if len(sys.argv) > 10 else 1
    dev = "cuda"
    x = torch.randn(B, Cin, F, T, device=dev)
    y = torch.empty(B, Cout, F, T, device=dev)
    w = torch.randn(Cout, Cin, KH, KW, device=dev) / math.sqrt(Cin * KH * KW)
    wp = L.pack_conv_weight(w)
    isc = torch.rand(B, Cin, device=dev) + 0.5
    osc = torch.randn(B, Cout, device=dev)
    res = x if Cin == Cout else (y if os.environ.get('PROBE_RMW') else None)
    p = L.Conv2dParams()
    p.x, p.y, p.res, p.aux = L.view4(x), L.view4(y), L.view4(res), L.view4(None)
    p.wp = wp.data_ptr()
    if act >= 0:
        p.in_scale, p.in_scale_ld = isc.data_ptr(), isc.stride(0)
    else:
        p.in_scale, p.in_scale_ld, act = None, 0, 0       # act < 0: no prologue at all (direct-to-LDS kernel eligible)
    p.out_scale, p.out_scale_ld = osc.data_ptr(), osc.stride(0)
    p.B, p.Cin, p.Cout, p.F, p.T = B, Cin, Cout, F, T
    p.Cin_pad, p.Cout_pad = wp.shape[1], wp.shape[2]
    p.KH, p.KW, p.dilF, p.act, p.epi = KH, KW, dil, act, 0
    p.alpha, p.res_scale = 1 / math.sqrt(2), 1.0
    if os.environ.get('PROBE_WINO', '0') != '0' and KH == 5:
        wpw = L.pack_conv_weight_wino(w)
        p.wp_wino, p.wino_taps = wpw.data_ptr(), wpw.shape[0]
    if os.environ.get('PROBE_V', '0') == '2':         # F(8,3): Winograd-domain input written by aid_scale_act(wino=2), 50-tap pack
        assert L.lib().aid_conv2d_wino8_supported(Cin, Cout, F, T, dil)
        wpw = L.pack_conv_weight_wino8(w)
        p.wp_wino, p.wino_taps = wpw.data_ptr(), wpw.shape[0]
        xv = torch.empty(B, Cin, F, 10 * (T // 8), device=dev)
        L.call("aid_scale_act", L.ScaleActParams(L.view4(x), L.view4(xv), None, 0, B, Cin, F, T, 0, 2))
        p.x, p.x_wino = L.view4(xv), 2
    elif os.environ.get('PROBE_V', '0') != '0':       # Winograd-domain input written by aid_scale_act(wino=1)
        assert L.lib().aid_conv2d_wino_input_supported(Cin, Cout, T)
        xv = torch.empty(B, Cin, F, 6 * (T // 4), device=dev)
        sp = L.ScaleActParams(L.view4(x), L.view4(xv), None, 0, B, Cin, F, T, 0, 1)
        L.call("aid_scale_act", sp)
        p.x, p.x_wino = L.view4(xv), 1
    for _ in range(2):
        L.call("aid_conv2d", p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.call("aid_conv2d", p)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * B * F * T * Cin * Cout * KH * KW
    print(f"conv {KH}x{KW} d{dil} B{B} Cin{Cin} Cout{Cout} F{F} T{T}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
