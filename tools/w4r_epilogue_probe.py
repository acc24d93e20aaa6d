import sys, os, math, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from audio_inpainting_diffusion_amd import _lib as L
DEV = "cuda"
def run(B, Cin, Cout, Fd, T, dil, epi, n=20):
    G = T // 4
    xv = torch.randn(B, Cin, Fd, 6 * G, device=DEV)
    w = torch.randn(Cout, Cin, 5, 3, device=DEV) / math.sqrt(Cin * 15)
    wp, wpw = L.pack_conv_weight(w), L.pack_conv_weight_wino(w)
    y = torch.empty(B, Cout, Fd, T, device=DEV)
    aux = torch.randn(B, Cout, Fd, T, device=DEV) if epi else None
    sc = torch.ones(B, Cout, device=DEV)
    p = L.Conv2dParams()
    p.x, p.y, p.res, p.aux = L.view4(xv), L.view4(y), L.view4(None), L.view4(aux)
    if epi:
        p.aux_scale, p.aux_scale_ld = sc.data_ptr(), sc.stride(0)
        p.out_scale, p.out_scale_ld = sc.data_ptr(), sc.stride(0)
    p.wp, p.wp_wino, p.wino_taps, p.x_wino = wp.data_ptr(), wpw.data_ptr(), 30, 1
    p.B, p.Cin, p.Cout, p.F, p.T = B, Cin, Cout, Fd, T
    p.Cin_pad, p.Cout_pad = wp.shape[1], wp.shape[2]
    p.KH, p.KW, p.dilF, p.act, p.epi = 5, 3, dil, 0, epi
    p.alpha, p.res_scale = 1.0, 1.0
    for _ in range(3): L.call("aid_conv2d", p)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): L.call("aid_conv2d", p)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B in (8, 3):
    for s in [(64,64,64,2048,1),(64,64,128,1024,2),(96,96,192,512,1),(96,96,256,256,4),(128,128,320,128,1),(128,128,256,256,2),(256,256,384,64,1),(256,256,448,32,1),(256,256,448,32,64)]:
        for epi in (0, 1):
            print(f"B{B} C{s[0]}->{s[1]} F{s[2]} T{s[3]} d{s[4]} epi{epi}: {run(B, s[0], s[1], s[2], s[3], s[4], epi):8.1f} us", flush=True)
