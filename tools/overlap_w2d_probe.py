#!/usr/bin/env python
"""How much of an HBM-bound pass hides under the 2-D Winograd form's GEMM of ANOTHER stream?  (The same question tools/overlap_kernels_probe.py asks of the
fused F(8,3) kernel.)  Stream A: aid_wino2d_gemm x nc; stream B: the 2-D input pass / output pass / plain aid_norm_bwd x np; alone, then together.
   python tools/overlap_w2d_probe.py [B] [C F T dil]"""
import math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_inpainting_diffusion_amd import _lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
C, F, T, dil = (int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (256, 384, 64, 2)
dev = "cuda"


def main():
    lib = L.lib()
    N = int(lib.aid_conv2d_wino2d_positions(B, F, T, dil))
    x, res, y = (torch.randn(B, C, F, T, device=dev) for _ in range(3))
    w = torch.randn(C, C, 5, 3, device=dev) / math.sqrt(C * 15)
    wp, w2 = L.pack_conv_weight(w), L.pack_conv_weight_wino2d(w)
    isc = torch.rand(B, C, device=dev) + 0.5
    V, M = torch.empty(48 * C * N, device=dev), torch.empty(48 * C * N, device=dev)
    V2, M2, y2 = torch.empty_like(V), torch.empty_like(M), torch.empty_like(y)                # the pass stream works on its own buffers (another sub-batch)
    sp = L.ScaleActParams(L.view4(x), L.View(V2.data_ptr(), 0, 0, 0), isc.data_ptr(), isc.stride(0), B, C, F, T, 1, 3, dil)
    L.call("aid_scale_act", L.ScaleActParams(L.view4(x), L.View(V.data_ptr(), 0, 0, 0), isc.data_ptr(), isc.stride(0), B, C, F, T, 1, 3, dil))
    gp = L.Wino2dGemmParams(w2.data_ptr(), V.data_ptr(), M.data_ptr(), 48, C, C, wp.shape[1], wp.shape[2], N, 0)
    L.call("aid_wino2d_gemm", L.Wino2dGemmParams(w2.data_ptr(), V.data_ptr(), M2.data_ptr(), 48, C, C, wp.shape[1], wp.shape[2], N, 0))
    p = L.Conv2dParams()
    p.x, p.y, p.res, p.aux = L.View(V2.data_ptr(), 0, 0, 0), L.view4(y2), L.view4(res), L.view4(None)
    p.wp, p.wp_wino, p.wino_taps, p.x_wino = wp.data_ptr(), w2.data_ptr(), 48, 3
    p.out_scale, p.out_scale_ld = isc.data_ptr(), isc.stride(0)
    p.B, p.Cin, p.Cout, p.F, p.T = B, C, C, F, T
    p.Cin_pad, p.Cout_pad = wp.shape[1], wp.shape[2]
    p.KH, p.KW, p.dilF, p.act, p.epi = 5, 3, dil, 0, 0
    p.alpha, p.res_scale = 0.7, 1.0
    p.ws, p.ws_bytes = M2.data_ptr(), M2.numel() * 4
    gd, gy, out = (torch.randn(B, C, F, T, device=dev) for _ in range(3))
    stats = torch.rand(B, 8, 2, device=dev)
    ws = torch.zeros(B * 8 * (L.AID_STATS_SPLIT * 2 + 2), device=dev, dtype=torch.float64)
    nb = L.NormBwdParams(L.view4(gd), L.view4(x), L.view4(gy), L.view4(out), B, C, F, T, 8, stats.data_ptr(), ws.data_ptr(), 1e-7, 0.7, 0, 0)
    el = B * C * F * T * 4
    passes = [("2-D input pass", lambda: L.call("aid_scale_act", sp), el + 48 * C * N * 4), ("2-D output pass", lambda: L.call("aid_conv2d_wino2d_output", p), 48 * C * N * 4 + 2 * el),
              ("norm_bwd (plain)", lambda: L.call("aid_norm_bwd", nb), 4 * el)]
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    nc = 10
    for name, fn, nbytes in passes:
        def run(do_a, do_b, npass):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if do_a:
                with torch.cuda.stream(sa):
                    for _ in range(nc):
                        L.call("aid_wino2d_gemm", gp)
            if do_b:
                with torch.cuda.stream(sb):
                    for _ in range(npass):
                        fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e3
        run(True, True, 10)
        ta = run(True, False, 0)
        t1 = run(False, True, 10)
        npass = max(4, int(round(10 * ta / t1 * 0.8)))                   # the pass sequence about 0.8 of the GEMM sequence
        tb, tab = run(False, True, npass), run(True, True, npass)
        print(f"B{B} C{C} F{F} T{T} d{dil}: GEMM x{nc} alone {ta:.2f} ms | {name} x{npass} alone {tb:.2f} ms ({nbytes * npass / tb / 1e9:.2f} TB/s) | together {tab:.2f} ms: "
              f"{100 * (ta + tb - tab) / min(ta, tb):.0f} % of the shorter sequence hidden")


if __name__ == "__main__":
    main()
