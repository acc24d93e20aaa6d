#!/usr/bin/env python
"""Experiment: one B=8 guided evaluation on one stream vs two B=4 halves on two HIP streams (kernel-level overlap of the
HBM-bound passes and conv tails of one half with the MFMA-bound convs of the other).  usage: two_stream_probe.py [B] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_inpainting_diffusion_amd.config import make_args
from audio_inpainting_diffusion_amd.init import seeded_init_, seeded_normal
from audio_inpainting_diffusion_amd.masks import mask_from_args
from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda")
args = make_args("maestro22k")
net = seeded_init_(Unet_CQT_oct_with_attention(args, dev), 0)
nets = [net]
L = args.exp.audio_len
mask = mask_from_args(args).to(dev)
x = torch.stack([torch.from_numpy(seeded_normal(3, b, L)) for b in range(B)]).to(dev) * 0.3
y = (torch.stack([torch.from_numpy(seeded_normal(4, b, L)) for b in range(B)]).to(dev) * 0.063 * mask).contiguous()
v = lambda B_, val: torch.full((B_,), val, device=dev)
def coeffs(B_): return v(B_, -0.2), v(B_, 1.9), v(B_, 0.02), v(B_, 0.06)

def run_single():
    return net.denoise_guided(x, *coeffs(B), True, y, mask)

nsplit = int(sys.argv[3]) if len(sys.argv) > 3 else 2
nets = [net]
for _ in range(nsplit - 1):
    n2 = Unet_CQT_oct_with_attention(args, dev)
    n2.load_state_dict(net.state_dict())
    nets.append(n2)
streams = [torch.cuda.Stream() for _ in range(nsplit)]
bounds = [(i * B) // nsplit for i in range(nsplit + 1)]
def run_split():
    cur = torch.cuda.current_stream()
    outs = []
    for i in range(nsplit):
        lo, hi = bounds[i], bounds[i + 1]
        streams[i].wait_stream(cur)
        with torch.cuda.stream(streams[i]):
            nets[i].split_streams = 1
            outs.append(nets[i].denoise_guided(x[lo:hi].contiguous(), *coeffs(hi - lo), True, y[lo:hi].contiguous(), mask))
    for st in streams:
        cur.wait_stream(st)
    return outs

for name, fn in (("single stream B=%d" % B, run_single), ("%d streams, split %s" % (nsplit, [bounds[i + 1] - bounds[i] for i in range(nsplit)]), run_split)):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name}: {dt * 1e3:.1f} ms per guided evaluation of {B} segments = {B / dt:.2f} evals/s", flush=True)
# free-running variant: every sub-batch runs all its evaluations on its own stream, one join at the very end
def run_free(k):
    cur = torch.cuda.current_stream()
    for i in range(nsplit):
        streams[i].wait_stream(cur)
    for _ in range(k):
        for i in range(nsplit):
            lo, hi = bounds[i], bounds[i + 1]
            with torch.cuda.stream(streams[i]):
                nets[i].split_streams = 1
                nets[i].denoise_guided(x[lo:hi].contiguous(), *coeffs(hi - lo), True, y[lo:hi].contiguous(), mask)
    for st in streams:
        cur.wait_stream(st)
run_free(1); torch.cuda.synchronize()
t0 = time.perf_counter(); run_free(reps); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"{nsplit} free-running streams (no join between evaluations): {dt * 1e3:.1f} ms per guided evaluation of {B} segments = {B / dt:.2f} evals/s", flush=True)
ra = run_single(); rb = run_split(); torch.cuda.synchronize()
print("max |x_hat diff| split vs single:", float((torch.cat([o[0] for o in rb]) - ra[0]).abs().max()))
