#!/usr/bin/env python
"""Per-wave phase timestamps of the conv kernel's K loop (instrumented build tools/libaid_dbg.so; experiments only)."""
import ctypes as C, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_inpainting_diffusion_amd import _lib as L
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libaid_dbg.so"))
B, Cin, Cout, F, T, KH, KW, dil = [int(v) for v in sys.argv[1:9]]
dev = "cuda"
x = torch.randn(B, Cin, F, T, device=dev); y = torch.empty(B, Cout, F, T, device=dev)
w = torch.randn(Cout, Cin, KH, KW, device=dev) / math.sqrt(Cin * KH * KW)
wp = L.pack_conv_weight(w)
dbg = torch.zeros(8 * 16 * 4, dtype=torch.int64, device=dev)
lib.aid_set_dbg(C.c_void_p(dbg.data_ptr()))
p = L.Conv2dParams()
p.x, p.y, p.res, p.aux = L.view4(x), L.view4(y), L.view4(x if Cin == Cout else None), L.view4(None)
p.wp = wp.data_ptr(); p.B, p.Cin, p.Cout, p.F, p.T = B, Cin, Cout, F, T
p.Cin_pad, p.Cout_pad = wp.shape[1], wp.shape[2]
p.KH, p.KW, p.dilF, p.act, p.epi = KH, KW, dil, 0, 0
p.alpha, p.res_scale = 1.0, 1.0
lib.aid_conv2d.argtypes = [C.c_void_p, C.c_void_p]
for _ in range(3):
    rc = lib.aid_conv2d(C.addressof(p), torch.cuda.current_stream().cuda_stream); assert rc == 0
torch.cuda.synchronize()
d = dbg.cpu().reshape(8, 16, 4)
print("chunk wave : slot0  slots1-4  barrier-wait  total   (shader cycles; clock counter units)")
for ch in range(8):
    for wv in range(16):
        t0, t1, t2, t3 = [int(v) for v in d[ch, wv]]
        if t0 == 0: continue
        print(ch + 2, wv, ":", t1 - t0, t2 - t1, t3 - t2, t3 - t0)
