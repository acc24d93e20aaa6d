# quick A/B timing of the row-shared kernel on the B=1 shapes: split on/off (ws given or not)
import sys, math, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from audio_inpainting_diffusion_amd import _lib as L
DEV = "cuda"
def run(B, Cin, Cout, Fd, T, dil, use_ws, n=30):
    G = T // 4
    xv = torch.randn(B, Cin, Fd, 6 * G, device=DEV)
    w = torch.randn(Cout, Cin, 5, 3, device=DEV) / math.sqrt(Cin * 15)
    wp, wpw = L.pack_conv_weight(w), L.pack_conv_weight_wino(w)
    y = torch.empty(B, Cout, Fd, T, device=DEV)
    p = L.Conv2dParams()
    p.x, p.y, p.res, p.aux = L.view4(xv), L.view4(y), L.view4(None), L.view4(None)
    p.wp, p.wp_wino, p.wino_taps, p.x_wino = wp.data_ptr(), wpw.data_ptr(), 30, 1
    p.B, p.Cin, p.Cout, p.F, p.T = B, Cin, Cout, Fd, T
    p.Cin_pad, p.Cout_pad = wp.shape[1], wp.shape[2]
    p.KH, p.KW, p.dilF, p.act, p.epi = 5, 3, dil, 0, 0
    p.alpha, p.res_scale = 1.0, 1.0
    need = int(L.lib().aid_conv2d_wino_split_ws_bytes(B, Cin, Cout, Fd, T, dil))
    ws = None
    if use_ws and need:
        ws = torch.zeros(need // 4, device=DEV)
        p.ws, p.ws_bytes = ws.data_ptr(), need
    for _ in range(5): L.call("aid_conv2d", p)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): L.call("aid_conv2d", p)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, L.lib().aid_last_kernel().decode(), need
shapes = [(1,256,256,448,32,1),(1,256,256,448,32,16),(1,256,256,448,32,64),(1,128,256,448,32,1),(1,256,128,384,64,1),(1,256,256,384,64,1),(1,128,128,320,128,1),(1,128,128,320,128,32),(1,128,128,256,256,1),
          (1,96,96,256,256,1),(1,96,96,256,256,16),(1,96,96,192,512,1),(1,96,96,128,1024,1),(1,64,64,64,2048,1),(8,96,96,192,512,1),(8,96,96,256,256,4),(3,96,96,192,512,2)]
for s in shapes:
    a = run(*s, False); b = run(*s, True)
    print(f"B{s[0]} Cin{s[1]} Cout{s[2]} F{s[3]} T{s[4]} d{s[5]}: no-ws {a[0]:7.1f} us {a[1]:34s} | ws {b[0]:7.1f} us {b[1]} need={b[2]>>20} MB", flush=True)
