#!/usr/bin/env python
"""How much of an HBM-bound pass hides under an MFMA-bound conv of ANOTHER stream?  One F(8,3) (or F(4,3)) conv launch sequence on stream A, one
pre-pass sequence (aid_norm_bwd wform=2 / aid_scale_act wino=2) on stream B: time of A alone, B alone, both together.
   python tools/overlap_kernels_probe.py [B] [C F T]"""
import math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_inpainting_diffusion_amd import _lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
C, F, T = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (256, 384, 64)
dev = "cuda"
n = B * C * F * T


def conv_params(form):
    x = torch.randn(B, C, F, T, device=dev); y = torch.empty_like(x)
    w = torch.randn(C, C, 5, 3, device=dev) / math.sqrt(C * 15)
    wp = L.pack_conv_weight(w)
    wpw = L.pack_conv_weight_wino8(w) if form == 8 else L.pack_conv_weight_wino(w)
    xv = torch.empty(B, C, F, 10 * (T // 8) if form == 8 else 6 * (T // 4), device=dev)
    L.call("aid_scale_act", L.ScaleActParams(L.view4(x), L.view4(xv), None, 0, B, C, F, T, 0, 2 if form == 8 else 1))
    osc = torch.randn(B, C, device=dev)
    p = L.Conv2dParams()
    p.x, p.y, p.res, p.aux = L.view4(xv), L.view4(y), L.view4(x), L.view4(None)
    p.wp, p.wp_wino, p.wino_taps, p.x_wino = wp.data_ptr(), wpw.data_ptr(), wpw.shape[0], 2 if form == 8 else 1
    p.out_scale, p.out_scale_ld = osc.data_ptr(), osc.stride(0)
    p.B, p.Cin, p.Cout, p.F, p.T = B, C, C, F, T
    p.Cin_pad, p.Cout_pad = wp.shape[1], wp.shape[2]
    p.KH, p.KW, p.dilF, p.act, p.epi = 5, 3, 2, 0, 0
    p.alpha, p.res_scale = 0.7, 1.0
    return p, (x, y, w, wp, wpw, xv, osc)


def pass_params():
    gd, x, gy, out = (torch.randn(B, C, F, T, device=dev) for _ in range(4))
    sc = torch.rand(B, C, device=dev) + 0.5
    stats = torch.rand(B, 8, 2, device=dev)
    ws = torch.zeros(B * 8 * (L.AID_STATS_SPLIT * 2 + 2), device=dev, dtype=torch.float64)
    xv8 = torch.empty(B, C, F, 10 * (T // 8), device=dev)
    p = L.NormBwdParams(L.view4(gd), L.view4(x), L.view4(gy), L.view4(out), B, C, F, T, 8, stats.data_ptr(), ws.data_ptr(), 1e-7, 0.7, 0, 0)
    p.wout, p.wscale, p.wscale_ld, p.wform = L.view4(xv8), sc.data_ptr(), sc.stride(0), 2
    return p, (gd, x, gy, out, sc, stats, ws, xv8)


def main():
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    pp, keep2 = pass_params()
    for form in (8, 4):
        cp, keep = conv_params(form)
        nc, npass = 10, 40

        def run(do_conv, do_pass):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if do_conv:
                with torch.cuda.stream(sa):
                    for _ in range(nc):
                        L.call("aid_conv2d", cp)
            if do_pass:
                with torch.cuda.stream(sb):
                    for _ in range(npass):
                        L.call("aid_norm_bwd", pp)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e3
        for _ in range(2):
            run(True, True)
        ta, tb, tab = run(True, False), run(False, True), run(True, True)
        print(f"B{B} C{C} F{F} T{T}: F({form},3) conv x{nc} alone {ta:.2f} ms | norm_bwd(wform=2) x{npass} alone {tb:.2f} ms ({21 * n * npass / tb / 1e9:.2f} TB/s) | "
              f"together {tab:.2f} ms = {100 * (ta + tb - tab) / min(ta, tb):.0f} % of the shorter one hidden")


if __name__ == "__main__":
    main()
