#!/usr/bin/env python
"""Why do CU-masked sub-batch streams lose end to end (profiles/r06_cu_split_ab.txt: 16 + 16 CUs per XCD -> half the throughput) when a masked stream launches
as fast as a plain one (r06_cu_mask_launch_cost.txt) and masked streams DO overlap in the kernel-pair probe (r06_cu_mask_probe.txt)?  One guided evaluation
of a sub-batch of four (the product's sub-batch), full-size network, timed (a) on a plain stream, (b) on a masked stream with ALL 256 bits set, (c) on a stream
masked to half of every XCD; then TWO such evaluations concurrently on (d) two plain streams, (e) two all-bits masked streams, (f) the two complementary halves.
   python tools/cu_mask_e2e_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_inpainting_diffusion_amd.config import make_args
from audio_inpainting_diffusion_amd.init import seeded_init_
from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
from audio_inpainting_diffusion_amd.streams import cu_masked_stream, xcd_range_mask

dev = torch.device("cuda")
args = make_args("maestro22k")
net = seeded_init_(Unet_CQT_oct_with_attention(args, dev), 0)
net.split_streams = 1
B, L = 4, args.exp.audio_len
x = torch.randn(2, B, L, device=dev) * 0.5
y = torch.randn(2, B, L, device=dev) * 0.063
mask = torch.ones(1, L, device=dev); mask[:, L // 2 - 3307: L // 2 + 3308] = 0
v = lambda a: torch.full((B,), a, device=dev)
co = (v(-0.2), v(1.5), v(0.1), v(0.3))
sts = [net._state(B, i, (2 * B, 2)) for i in range(2)]          # two independent launch-plan states, as the two sub-batches of a batch of eight have


def one(i):
    return net._denoise_guided_one(x[i], *co, True, y[i], mask, None, sts[i])


def run(streams, reps=3):
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                one(i)
    for s in streams:
        s.synchronize()
    return (time.perf_counter() - t0) * 1e3 / reps


for i in range(2):
    one(i); one(i)
torch.cuda.synchronize()
full = [0xFFFFFFFF] * 8
cases = [("one evaluation, plain stream", [torch.cuda.Stream()]),
         ("one evaluation, masked stream with all 256 bits set", [cu_masked_stream(full)]),
         ("one evaluation, masked to CUs 0-15 of every XCD", [cu_masked_stream(xcd_range_mask(0, 16))]),
         ("two evaluations, two plain streams", [torch.cuda.Stream(), torch.cuda.Stream()]),
         ("two evaluations, two masked streams with all bits set", [cu_masked_stream(full), cu_masked_stream(full)]),
         ("two evaluations, CUs 0-15 | 16-31 of every XCD", [cu_masked_stream(xcd_range_mask(0, 16)), cu_masked_stream(xcd_range_mask(16, 32))])]
for label, ss in cases:
    run(ss, 1)
    print(f"{label:58s} {run(ss):8.2f} ms per round of {len(ss)} evaluation(s) of {B} segments", flush=True)
