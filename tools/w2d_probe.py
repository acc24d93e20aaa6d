#!/usr/bin/env python
"""Probe of the 2-D Winograd path (csrc/aid_wino2d.hip): the batched GEMM alone (tile-shape variants), and -- once built -- the three passes of a
layer against the 1-D kernels.  GPU only.
    python tools/w2d_probe.py gemm [variant ...]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from audio_inpainting_diffusion_amd import _lib  # noqa: E402

PEAK = 157.3e12

# (label, Cin, Cout, N): N = B * ceil(F / (4 dil)) * dil * T / 4 at the shipped level shapes
GEMM_SHAPES = [
    ("L5 C256 F384 T64 B4", 256, 256, 4 * 96 * 16),
    ("L5 C256 F384 T64 B8", 256, 256, 8 * 96 * 16),
    ("L6 C256 F448 T32 B4", 256, 256, 4 * 112 * 8),
    ("L6 C256 F448 T32 B8", 256, 256, 8 * 112 * 8),
    ("L4 C128 F320 T128 B4", 128, 128, 4 * 80 * 32),
    ("L3 C128 F256 T256 B4", 128, 128, 4 * 64 * 64),
    ("L5 C256 B1", 256, 256, 96 * 16),
    ("L6 C256 B1", 256, 256, 112 * 8),
    ("L6 C256 B2", 256, 256, 2 * 112 * 8),
    ("L5 C256 B2", 256, 256, 2 * 96 * 16),
    ("L4 C128 B1", 128, 128, 80 * 32),
    ("L4 C128 B2", 128, 128, 2 * 80 * 32),
    ("L3 C128 B1", 128, 128, 64 * 64),
    ("L5d C128 B1", 128, 128, 96 * 16),
]


def time_call(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def gemm(variants):
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    print(f"{'shape':24s} " + " ".join(f"v{v:<12d}" for v in variants) + "   (us | fraction of the fp32 MFMA peak 157.3 TF)")
    for label, cin, cout, N in GEMM_SHAPES:
        cip, cop = _lib.pack_dims(cin, cout)
        U = (torch.randn(48, cip, cop, generator=g) / cin ** 0.5).to(dev)
        V = torch.randn(48, cin, N, generator=g).to(dev)
        M = torch.empty(48, cout, N, device=dev)
        ref = torch.bmm(U[:, :cin, :cout].transpose(1, 2).double(), V.double())
        row = []
        for v in variants:
            p = _lib.Wino2dGemmParams(U.data_ptr(), V.data_ptr(), M.data_ptr(), 48, cin, cout, cip, cop, N, v)
            M.zero_()
            try:
                _lib.call("aid_wino2d_gemm", p)
            except _lib.AidError as e:
                row.append(f"n/a ({str(e)[-30:]})")
                continue
            torch.cuda.synchronize()
            err = float((M.double() - ref).norm() / ref.norm())
            t = time_call(lambda: _lib.call("aid_wino2d_gemm", p))
            fl = 2.0 * 48 * cin * cout * N
            row.append(f"{t * 1e6:7.1f} {fl / t / PEAK:5.3f}" + ("" if err < 2e-6 else f" ERR {err:.1e}"))
        print(f"{label:24s} " + " ".join(f"{r:13s}" for r in row), flush=True)
        del U, V, M, ref


# (label, C, F, T, dilations) -- the C >= 128 levels of the 22.05 kHz network (level geometry)
LAYER_SHAPES = [
    ("L1 C96 F128 T1024", 96, 128, 1024, (1, 4)),
    ("L2 C96 F192 T512", 96, 192, 512, (1, 2, 4, 8)),
    ("L3d C96 F256 T256", 96, 256, 256, (1, 4)),
    ("L3 C128 F256 T256", 128, 256, 256, (1, 2, 4, 8, 16)),
    ("L4 C128 F320 T128", 128, 320, 128, (1, 2, 4, 8, 16, 32)),
    ("L5 C256 F384 T64", 256, 384, 64, (1, 2, 4, 8, 16, 32, 64)),
    ("L5d C128 F384 T64", 128, 384, 64, (1, 8, 64)),
    ("L6 C256 F448 T32", 256, 448, 32, (1, 2, 4, 8, 16, 32, 64)),
]


GEMM_VARIANTS = []


def layer(batches):
    """per layer: the 2-D path (input pass | GEMM | output pass) against the 1-D path the library picks today (pre-pass | fused conv)"""
    dev = torch.device("cuda:0")
    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(0)
    print("layer, dil, B | 2-D: input us (TB/s)  gemm us (frac)  output us (TB/s)  total | 1-D form: pre us  conv us  total | ratio")
    only = os.environ.get("PROBE_ONLY", "")                  # e.g. "L5 C256": restrict to the layer shapes whose label starts with it (PMC runs)
    for B in batches:
        for label, C, Fd, T, dils in LAYER_SHAPES:
            if only and not label.startswith(only):
                continue
            x = torch.randn(B, C, Fd, T, generator=g).to(dev)
            res = torch.randn(B, C, Fd, T, generator=g).to(dev)
            y = torch.empty_like(x)
            w = (torch.randn(C, C, 5, 3, generator=g) / (C * 15) ** 0.5).to(dev)
            isc, osc = torch.ones(B, C, device=dev), torch.ones(B, C, device=dev)
            TF = int(os.environ.get("PROBE_TF", "4"))            # 8: the F(4,5) x F(8,3) form (x_wino = 4, 80 planes)
            NXI, XW = (80, 4) if TF == 8 else (48, 3)
            wp, w4, w8 = _lib.pack_conv_weight(w), _lib.pack_conv_weight_wino(w), _lib.pack_conv_weight_wino8(w)
            w2 = _lib.pack_conv_weight_wino2d8(w) if TF == 8 else _lib.pack_conv_weight_wino2d(w)
            el = B * C * Fd * T
            for dil in dils:
                if not lib.aid_conv2d_wino2d_supported(C, C, Fd, T, dil) or (TF == 8 and T % 32):
                    continue
                N = int(lib.aid_conv2d_wino2d_positions(B, Fd, T, dil)) * 4 // TF
                V = torch.empty(NXI * C * N, device=dev)
                ws = torch.empty(NXI * C * N, device=dev)
                sp = _lib.ScaleActParams(_lib.view4(x), _lib.View(V.data_ptr(), 0, 0, 0), isc.data_ptr(), isc.stride(0), B, C, Fd, T, 1, XW, dil)
                gp = _lib.Wino2dGemmParams(w2.data_ptr(), V.data_ptr(), ws.data_ptr(), NXI, C, C, wp.shape[1], wp.shape[2], N, 0)

                def cpar(xin, xw, wpw, taps):
                    p = _lib.Conv2dParams()
                    p.x, p.y, p.res, p.aux = xin, _lib.view4(y), _lib.view4(res), _lib.view4(None)
                    p.wp, p.wp_wino, p.wino_taps, p.x_wino = wp.data_ptr(), wpw.data_ptr(), taps, xw
                    p.out_scale, p.out_scale_ld = osc.data_ptr(), osc.stride(0)
                    p.B, p.Cin, p.Cout, p.F, p.T = B, C, C, Fd, T
                    p.Cin_pad, p.Cout_pad = wp.shape[1], wp.shape[2]
                    p.KH, p.KW, p.dilF, p.act, p.epi = 5, 3, dil, 0, 0
                    p.alpha, p.res_scale = 0.7071, 1.0
                    return p
                p3 = cpar(_lib.View(V.data_ptr(), 0, 0, 0), XW, w2, NXI)
                p3.ws, p3.ws_bytes = ws.data_ptr(), ws.numel() * 4
                t_in = time_call(lambda: _lib.call("aid_scale_act", sp))
                t_g = time_call(lambda: _lib.call("aid_wino2d_gemm", gp))
                tv = []
                for v in GEMM_VARIANTS:                              # (tile-shape experiments on the layer's own V)
                    gp.variant = v
                    try:
                        tv.append(f"v{v}:{time_call(lambda: _lib.call('aid_wino2d_gemm', gp)) * 1e6:.0f}")
                    except _lib.AidError:
                        tv.append(f"v{v}:n/a")
                gp.variant = 0
                t_c = time_call(lambda: _lib.call("aid_conv2d", p3))
                t_out = t_c - t_g
                nfl = 2.0 * NXI * C * C * N
                # the 1-D path as the library picks it
                form = int(lib.aid_conv2d_wino_form(B, C, C, Fd, T, dil))
                cols = {4: 6 * (T // 4), 8: 10 * (T // 8)}.get(form)
                if cols is None:
                    print(f"{label} d{dil} B{B}: 1-D form {form}?")
                    continue
                xv = torch.empty(B, C, Fd, cols, device=dev)
                sp1 = _lib.ScaleActParams(_lib.view4(x), _lib.view4(xv), isc.data_ptr(), isc.stride(0), B, C, Fd, T, 1, {4: 1, 8: 2}[form])
                p1 = cpar(_lib.view4(xv), {4: 1, 8: 2}[form], {4: w4, 8: w8}[form], {4: 30, 8: 50}[form])
                t_p1 = time_call(lambda: _lib.call("aid_scale_act", sp1))
                t_c1 = time_call(lambda: _lib.call("aid_conv2d", p1))
                pad = N * 4 * TF / (B * Fd * T)
                print(f"{label} d{dil:<2d} B{B} | {t_in * 1e6:6.1f} ({(el * 4 + NXI * C * N * 4) / t_in / 1e12:4.2f})  {t_g * 1e6:6.1f} ({nfl / t_g / PEAK:5.3f})  "
                      f"{t_out * 1e6:6.1f} ({(NXI * C * N * 4 + 3 * el * 4) / t_out / 1e12:4.2f})  {(t_in + t_c) * 1e6:7.1f} | F({form},3) {t_p1 * 1e6:6.1f} {t_c1 * 1e6:7.1f} "
                      f"{(t_p1 + t_c1) * 1e6:7.1f} | {(t_in + t_c) / (t_p1 + t_c1):5.3f}  pad {pad:4.2f}  " + " ".join(tv), flush=True)
                del V, ws, xv
            del x, res, y


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "gemm"
    if what == "gemm":
        gemm([int(a) for a in sys.argv[2:]] or list(range(7)))
    elif what == "layer":                                        # layer B [B ...] [--variants v,v,...]
        args = sys.argv[2:]
        if "--variants" in args:
            i = args.index("--variants")
            GEMM_VARIANTS[:] = [int(v) for v in args[i + 1].split(",")]
            del args[i:i + 2]
        layer([int(a) for a in args] or [4])
