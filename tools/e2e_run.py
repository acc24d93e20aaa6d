#!/usr/bin/env python
"""Complete sampling runs (all T steps of the tester's schedule) on the full-size random-init networks: wall time,
denoiser evaluations per second over the WHOLE run, finiteness and data consistency of the result.
   python tools/e2e_run.py [maestro22k|librispeech16k|musicnet44k] [batch]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_inpainting_diffusion_amd.config import make_args
from audio_inpainting_diffusion_amd.edm import EDM
from audio_inpainting_diffusion_amd.init import seeded_init_, seeded_normal
from audio_inpainting_diffusion_amd.masks import mask_from_args
from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
from audio_inpainting_diffusion_amd.sampler import Sampler

name = sys.argv[1] if len(sys.argv) > 1 else "maestro22k"
T, gap_ms, B_def = {"maestro22k": (36, 300.0, 8), "librispeech16k": (70, 50.0, 16), "musicnet44k": (128, 1500.0, 4)}[name]
B = int(sys.argv[2]) if len(sys.argv) > 2 else B_def
args = make_args(name, T=T, gap_ms=gap_ms, xi=0.25)
dev = torch.device("cuda")
net = seeded_init_(Unet_CQT_oct_with_attention(args, dev), 0)
L = args.exp.audio_len
y = torch.stack([torch.from_numpy(seeded_normal(7, b, L)) for b in range(B)]) * 0.063
mask = mask_from_args(args, generator=torch.Generator().manual_seed(99))
smp = Sampler(model=net, diff_params=EDM(args), args=args)
smp.seeds = list(range(100, 100 + B))
net.prepare()
torch.cuda.synchronize()
t0 = time.perf_counter()
out = smp.predict_inpainting((y * mask).to(dev), mask.to(dev))
torch.cuda.synchronize()
wall = time.perf_counter() - t0
evals = B * (2 * T - 1)
keep = mask[0].bool()
err = float((out.cpu()[:, keep] - y[:, keep]).abs().max())
print(f"{name}: B={B} T={T} guided (xi=0.25): {wall:.2f} s wall for the whole run (first call includes plan building), {evals} denoiser evaluations "
      f"-> {evals / wall:.2f} evals/s; finite={bool(torch.isfinite(out).all())}; max |out - y| on the known samples (outside the Hann ramps incl.) = {err:.2e}; "
      f"out rms in the gap = {float(out.cpu()[:, ~keep].pow(2).mean().sqrt()):.4f}")
t0 = time.perf_counter()
out2 = smp.predict_inpainting((y * mask).to(dev), mask.to(dev))
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"{name}: second run (plans cached): {wall:.2f} s -> {evals / wall:.2f} evals/s; identical to the first run: {bool(torch.equal(out, out2))}")
