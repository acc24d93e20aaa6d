#!/usr/bin/env python
"""Histogram of the launch plans of one guided evaluation (forward body + input-VJP) of the full-size network: ops by name, and for the
element-wise passes the tensor shapes they run on.  usage: plan_histogram.py [batch]"""
import collections, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_inpainting_diffusion_amd import _lib
from audio_inpainting_diffusion_amd.config import make_args
from audio_inpainting_diffusion_amd.init import seeded_init_
from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda")
net = seeded_init_(Unet_CQT_oct_with_attention(make_args("maestro22k"), dev), 0)
net.split_streams = 1
st = net._state(B)
net._bwd_plan(st)
for nm in ("plan_body", "plan_bwd"):
    pl = st.get(nm)
    if pl is None:
        continue
    cnt = collections.Counter(d.split()[0] if d.startswith("conv") else d for d in pl.descr)
    print(nm, dict(cnt))
    structs = [k for k in pl.keep if isinstance(k, C.Structure)]
    shapes = collections.Counter()
    for k in structs:
        t = type(k).__name__
        if t == "Add2Params":
            shapes[("add2", k.B, k.C, k.F, k.T, "v" if k.v.p else "-")] += 1
        elif t == "NormBwdParams":
            shapes[("norm_bwd", k.B, k.C, k.F, k.T, "acc" if k.accumulate else "-", "wino" if k.wout.p else "plain")] += 1
        elif t == "ScaleActParams":
            shapes[("scale_act", k.B, k.C, k.F, k.T, "act" if k.act else "-", "wino" if k.wino else "plain")] += 1
    for k, v in sorted(shapes.items()):
        print("   ", k, v)
