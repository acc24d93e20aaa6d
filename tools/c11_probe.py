"""Time the 1x1 layers of the 22.05 kHz network (batch 8 / 1) on the kernel the dispatcher picks; run once with AID_C11_RS=0 and once with 1.
usage: python tools/c11_probe.py [batch]"""
import math, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from audio_inpainting_diffusion_amd import _lib as L
DEV = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
# Cin, Cout, F, T, act, epi
SHAPES = [(64, 64, 64, 2048, 1, 0), (64, 64, 64, 2048, 0, 1), (64, 96, 128, 1024, 0, 0), (96, 96, 192, 512, 1, 0), (96, 96, 192, 512, 0, 1), (96, 96, 256, 256, 0, 1),
          (96, 128, 256, 256, 0, 0), (128, 128, 320, 128, 1, 0), (128, 128, 320, 128, 0, 1), (128, 64, 64, 2048, 0, 0), (192, 64, 128, 1024, 0, 0), (192, 96, 192, 512, 0, 0),
          (256, 96, 256, 256, 0, 0), (256, 128, 320, 128, 0, 0), (128, 256, 384, 64, 0, 0), (256, 256, 448, 32, 1, 0), (256, 256, 448, 32, 0, 1), (256, 256, 320, 128, 0, 0),
          (128, 192, 128, 1024, 0, 0), (192, 192, 192, 512, 0, 0), (192, 256, 256, 256, 0, 0)]
tot = 0.0
for Cin, Cout, Fd, T, act, epi in SHAPES:
    x = torch.randn(B, Cin, Fd, T, device=DEV)
    w = torch.randn(Cout, Cin, 1, 1, device=DEV) / math.sqrt(Cin)
    wp = L.pack_conv_weight(w)
    y = torch.empty(B, Cout, Fd, T, device=DEV)
    aux = torch.randn(B, Cout, Fd, T, device=DEV) if epi else None
    sc = torch.ones(B, max(Cin, Cout), device=DEV)
    p = L.Conv2dParams()
    p.x, p.y, p.res, p.aux = L.view4(x), L.view4(y), L.view4(None), L.view4(aux)
    p.wp = wp.data_ptr()
    if act:
        p.in_scale, p.in_scale_ld = sc.data_ptr(), sc.stride(0)
    if epi:
        p.aux_scale, p.aux_scale_ld = sc.data_ptr(), sc.stride(0)
        p.out_scale, p.out_scale_ld = sc.data_ptr(), sc.stride(0)
    p.B, p.Cin, p.Cout, p.F, p.T = B, Cin, Cout, Fd, T
    p.Cin_pad, p.Cout_pad = wp.shape[1], wp.shape[2]
    p.KH, p.KW, p.dilF, p.act, p.epi = 1, 1, 1, act, epi
    p.alpha, p.res_scale = 1.0, 1.0
    for _ in range(3):
        L.call("aid_conv2d", p)
    n = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        L.call("aid_conv2d", p)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    nbytes = 4 * B * Fd * T * (Cin + Cout * (1 + epi))
    fl = 2 * B * Fd * T * Cin * Cout
    tot += us
    print(f"B{B} Cin{Cin:4d} Cout{Cout:4d} F{Fd:4d} T{T:5d} act{act} epi{epi}: {us:7.1f} us  {nbytes / us / 1e6:5.2f} TB/s  {fl / us / 1e6:6.1f} TFLOP/s  {L.lib().aid_last_kernel().decode()}", flush=True)
    del x, y, aux
print(f"total {tot:.1f} us")
