#!/usr/bin/env python
"""Per-launch timeline of one guided evaluation (forward body + input-VJP) of the full-size network on ONE stream: every C-ABI
launch bracketed by HIP events, printed in execution order with its tensor shape, then summed by category.
usage: plan_trace.py [batch] [workload] [--list]"""
import collections, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_inpainting_diffusion_amd import _lib
from audio_inpainting_diffusion_amd.config import make_args
from audio_inpainting_diffusion_amd.init import seeded_init_
from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
from audio_inpainting_diffusion_amd.plan import CONV_OPS

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(argv[0]) if argv else 8
wl = argv[1] if len(argv) > 1 else "maestro22k"
dev = torch.device("cuda")
args = make_args(wl)
net = seeded_init_(Unet_CQT_oct_with_attention(args, dev), 0)
net.split_streams = 1
net.use_graphs = False
L = args.exp.audio_len
x = torch.randn(B, L, device=dev) * 0.5
y = torch.randn(B, L, device=dev) * 0.063
mask = torch.ones(1, L, device=dev); mask[:, L // 2 - 3307: L // 2 + 3308] = 0
v = lambda a: torch.full((B,), a, device=dev)
run = lambda: net.denoise_guided(x, v(-0.2), v(1.5), v(0.1), v(0.3), True, y, mask)
for _ in range(2):
    run()
st = net._state(B)
plans = {"fwd": st["plan_body"], "bwd": st["plan_bwd"]}
structs = {}
for nm, pl in plans.items():
    for k in pl.keep:
        if isinstance(k, C.Structure):
            structs[C.addressof(k)] = k
tr = {nm: [] for nm in plans}
for nm, pl in plans.items():
    pl.trace = tr[nm]
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record()
torch.cuda.synchronize()
for pl in plans.values():
    pl.trace = None
total = e0.elapsed_time(e1)

def shape_of(k):
    f = {n for n, _ in k._fields_}
    if {"B", "C", "F", "T"} <= f:
        return "B%d C%-4d F%-3d T%-4d" % (k.B, k.C, k.F, k.T)
    if {"B", "H", "F", "T"} <= f:
        return "B%d H%d F%d T%d" % (k.B, k.H, k.F, k.T)
    return ""

cat = collections.OrderedDict()
rows = []
lane_us = {}
for nm in ("fwd", "bwd"):
    for (a0, a1, name, descr, addr, kn, lane) in tr[nm]:
        us = 1e3 * a0.elapsed_time(a1)
        k = structs.get(addr)
        sh = shape_of(k) if (k is not None and name not in CONV_OPS) else ""
        extra = ""
        if k is not None:
            t = type(k).__name__
            if t == "NormBwdParams": extra = ("wino " if k.wout.p else "") + ("acc" if k.accumulate else "")
            if t == "ScaleActParams": extra = ("act " if k.act else "") + ("wino2d" if k.wino == 3 else ("wino" if k.wino else ""))
            if t == "Add2Params": extra = "2in" if k.v.p else "1in"
            if t == "ResampleParams": extra = "up%d adj%d acc%d" % (k.up, k.adjoint, k.accumulate)
            if t == "GroupStatsParams": extra = "ws_n%d" % k.ws_n
        key = name.replace("aid_", "")
        if name in CONV_OPS:
            q = k
            if kn.startswith("conv53_wino"): key = "conv5x3 winograd (fused 1-D kernels)"
            elif kn.startswith("w2d_gemm"): key = "conv5x3 2-D winograd: batched GEMM"
            elif kn.startswith("w2d_output"): key = "conv5x3 2-D winograd: output pass"
            elif q.KH == 5: key = "conv5x3 few-channel (%s)" % kn.split("(")[0]
            elif q.F == 1: key = "conv1x1 qk GEMM"
            elif min(q.Cin, q.Cout) <= 8: key = "conv1x1 few-channel (2/8 <-> C)"
            elif q.act or q.epi: key = "conv1x1 CxC step (act/epi)"
            else: key = "conv1x1 proj/res (C<->C')"
        elif extra and name in ("aid_norm_bwd", "aid_scale_act"):
            key += " " + ("2-D winograd input pass" if "wino2d" in extra else ("wino" if "wino" in extra else "plain"))
        r = cat.setdefault(key, [0, 0.0]); r[0] += 1; r[1] += us
        rows.append((nm + str(lane), name, descr if name in CONV_OPS else sh + " " + extra, kn, us))
        lane_us[lane] = lane_us.get(lane, 0.0) + us
if "--list" in sys.argv:
    for nm, name, d, kn, us in rows:
        print("%s %-22s %-62s %-34s %8.1f us" % (nm, name.replace("aid_", ""), d, kn, us))
s = sum(v[1] for v in cat.values())
print("batch %d %s: guided evaluation %.2f ms wall (single stream, events around every launch), sum of launches %.2f ms, %d launches" % (B, wl, total, s / 1e3, len(rows)))
print("  time by lane: " + ", ".join("lane %d %.2f ms" % (l, u / 1e3) for l, u in sorted(lane_us.items())))
for k, (n, us) in sorted(cat.items(), key=lambda kv: -kv[1][1]):
    print("  %-44s n=%4d  %9.2f ms  %5.1f%%" % (k, n, us / 1e3, 100 * us / s))
