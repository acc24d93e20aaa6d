#!/usr/bin/env python
"""The 2-D Winograd form with F(4,3) against F(8,3) along T, per layer (round 6): input pass | GEMM | output pass of aid_conv2d x_wino = 3 (48 planes over groups
of four samples) and x_wino = 4 (80 planes over groups of eight: 2.5 instead of 3.0 products per output, V / M 2.5 x instead of 3 x the activation), same inputs;
also the rel-L2 of the F(8,3) result against the F(4,3) one and of both against the fp64 direct convolution on the CPU (first dilation of every shape).
    python tools/w2d_tf_probe.py [B ...]"""
import math, os, sys
import torch
import torch.nn.functional as Fn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_inpainting_diffusion_amd import _lib
from w2d_probe import LAYER_SHAPES, time_call, PEAK


def main(batches):
    dev = torch.device("cuda:0")
    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(0)
    print("layer, dil, B | F(4,3) along T: input / gemm (frac of fp32 peak) / output = total us | F(8,3) along T: ... | ratio | rel-L2 (8 vs 4) [4 vs fp64, 8 vs fp64]")
    for B in batches:
        for label, C, Fd, T, dils in LAYER_SHAPES:
            x = torch.randn(B, C, Fd, T, generator=g)
            res = torch.randn(B, C, Fd, T, generator=g)
            w = torch.randn(C, C, 5, 3, generator=g) / (C * 15) ** 0.5
            xd, resd, wd = x.to(dev), res.to(dev), w.to(dev)
            isc, osc = torch.ones(B, C, device=dev), torch.ones(B, C, device=dev)
            wp = _lib.pack_conv_weight(wd)
            packs = {3: _lib.pack_conv_weight_wino2d(wd), 4: _lib.pack_conv_weight_wino2d8(wd)}
            el = B * C * Fd * T
            for k, dil in enumerate(dils):
                if not lib.aid_conv2d_wino2d_supported(C, C, Fd, T, dil) or T % 32:
                    continue
                N4 = int(lib.aid_conv2d_wino2d_positions(B, Fd, T, dil))
                out, ys = {}, {}
                for xw, nxi, N in ((3, 48, N4), (4, 80, N4 // 2)):
                    V, ws, y = torch.empty(nxi * C * N, device=dev), torch.empty(nxi * C * N, device=dev), torch.empty(B, C, Fd, T, device=dev)
                    sp = _lib.ScaleActParams(_lib.view4(xd), _lib.View(V.data_ptr(), 0, 0, 0), isc.data_ptr(), isc.stride(0), B, C, Fd, T, 1, xw, dil)
                    p = _lib.Conv2dParams()
                    p.x, p.y, p.res, p.aux = _lib.View(V.data_ptr(), 0, 0, 0), _lib.view4(y), _lib.view4(resd), _lib.view4(None)
                    p.wp, p.wp_wino, p.wino_taps, p.x_wino = wp.data_ptr(), packs[xw].data_ptr(), nxi, xw
                    p.out_scale, p.out_scale_ld = osc.data_ptr(), osc.stride(0)
                    p.B, p.Cin, p.Cout, p.F, p.T = B, C, C, Fd, T
                    p.Cin_pad, p.Cout_pad = wp.shape[1], wp.shape[2]
                    p.KH, p.KW, p.dilF, p.act, p.epi = 5, 3, dil, 0, 0
                    p.alpha, p.res_scale = 0.7071, 1.0
                    p.ws, p.ws_bytes = ws.data_ptr(), ws.numel() * 4
                    t_in = time_call(lambda: _lib.call("aid_scale_act", sp))
                    t_g = time_call(lambda: _lib.call("aid_conv2d_wino2d_gemm", p))
                    t_o = time_call(lambda: _lib.call("aid_conv2d_wino2d_output", p))
                    torch.cuda.synchronize()
                    out[xw] = (t_in, t_g, t_o, 2.0 * nxi * C * C * N / t_g / PEAK)
                    ys[xw] = y.cpu()
                    del V, ws, y
                e84 = float((ys[4] - ys[3]).norm() / ys[3].norm())
                extra = ""
                if k == 0 and B == batches[0]:
                    ref = 0.7071 * (Fn.conv2d(Fn.gelu(x.double()), w.double(), padding="same", dilation=(dil, 1)) + res.double())
                    extra = f" [{float((ys[3].double() - ref).norm() / ref.norm()):.1e}, {float((ys[4].double() - ref).norm() / ref.norm()):.1e}]"
                a, b = out[3], out[4]
                print(f"{label} d{dil:<2d} B{B} | {a[0] * 1e6:6.1f} / {a[1] * 1e6:6.1f} ({a[3]:5.3f}) / {a[2] * 1e6:6.1f} = {sum(a[:3]) * 1e6:7.1f} | "
                      f"{b[0] * 1e6:6.1f} / {b[1] * 1e6:6.1f} ({b[3]:5.3f}) / {b[2] * 1e6:6.1f} = {sum(b[:3]) * 1e6:7.1f} | {sum(b[:3]) / sum(a[:3]):5.3f} | {e84:.1e}{extra}", flush=True)


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [4])
