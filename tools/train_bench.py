#!/usr/bin/env python
"""Iterations per second of the training step on the full-size network (the reference trains at batch 4, conf/exp/maestro22k_8s.yaml:31).
usage: train_bench.py [batch] [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_inpainting_diffusion_amd.config import make_args
from audio_inpainting_diffusion_amd.edm import EDM
from audio_inpainting_diffusion_amd.init import seeded_init_, seeded_normal
from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
from audio_inpainting_diffusion_amd.training import Trainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda")
args = make_args("maestro22k")
net = seeded_init_(Unet_CQT_oct_with_attention(args, dev), 0)
tr = Trainer(net, EDM(args), batch=B)
L = args.exp.audio_len
audio = torch.stack([torch.from_numpy(seeded_normal(5, b, L)) for b in range(B)]).to(dev) * 0.063
torch.manual_seed(0)
for _ in range(2):
    loss = tr.train_step(audio)
torch.cuda.synchronize()
st = net.train_state(B)
t0 = time.perf_counter()
for _ in range(n):
    loss = tr.train_step(audio)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
fl = st["plan_body"].flops + st["plan_bwd"].flops
print(f"training step, full-size 22.05 kHz network, batch {B}: {dt * 1e3:.1f} ms / iteration = {B / dt:.2f} segments/s; loss {float(loss):.5f}; "
      f"algorithmic conv/GEMM FLOPs per iteration {fl / 1e12:.2f} T -> {fl / dt / 1e12:.1f} TFLOP/s; state {st['nbytes'] / 1e9:.1f} GB")
# time split: forward + input-VJP only (what a guided sampler evaluation costs) for comparison
