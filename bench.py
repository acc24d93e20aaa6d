#!/usr/bin/env python
"""Benchmark of the EDM inpainting sampling hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--xi 0|0.25] [--batch 8]
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1], SURVEY.md section 8d "Config 2"): batch 8 x 22.05 kHz MAESTRO-shape
segments (L=184184) per GPU, centred 300 ms gap, tester parameters of conf/tester/inpainting_tester.yaml with
T=36, random-init (seeded) full-size network (186 M parameters), synthetic waveforms 0.063*N(0,1).
One "step" = one iteration of the sampling loop = stochastic churn + TWO denoiser evaluations (Heun) +
data-consistency projection for all 8 segments.  W untimed warm-up steps, then exactly K steps timed between
barrier+synchronize pairs; MAX over ranks.  value = denoiser evaluations per second over the whole job
(= n_gpus * 8 segments * 2 evaluations * K / wall); ms_per_step = wall / K.  Weak scaling: every GPU owns 8
segments; the only collective is the start-up weight broadcast (not timed).

Default branch: xi=0.25 = reconstruction guidance, the reference tester's shipped setting: every evaluation is a
forward pass PLUS the input-VJP through the whole denoiser (--xi 0 times the forward-only replacement branch).

roofline: the dominant kernel family is aid_conv2d (fp32 v_mfma_f32_32x32x2_f32 implicit GEMM, 99 % of the FLOPs;
the 5x3 layers run conv53_wino4_kernel = Winograd F(4,3) along T, the rest conv_mfma_kernel).  Every conv launch
inside the timed region (forward and VJP plans) is bracketed by HIP events on the launch stream;
achieved = sum of ALGORITHMIC (direct-convolution) conv FLOPs / sum of conv kernel time, peak = 157.3 TFLOP/s
(fp32 MFMA dense).  Winograd executes half the MFMAs of the direct form for the 5x3 layers, so `achieved` can
approach / exceed the direct-form peak; `executed_mfma_tflops` is the matrix-pipe rate actually issued.
traffic = HBM bytes per conv launch from the committed PMC pass of this command (profiles/), corrected per
MI355X_MICROARCH.md (FETCH_SIZE x2).
cpu_baseline: the CPU oracle (torch-CPU restatement of the reference path, oracle/) timed on this host's cores
for the same network and branch at B=1 (rank 0, N=1 only), bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def cpu_baseline(net, args, guided: bool, spectral: bool = False):
    """The CPU oracle (oracle/: torch-CPU restatement of the reference path) timed on this host for the same
    full-size network at B=1: one warm-up forward evaluation, then a bounded timed sample of the same branch the
    GPU number is quoted on (guided: 1 evaluation = forward with autograd graph + input gradient; xi=0: 2 forward
    evaluations)."""
    from oracle.nsgt_cqt import OracleCQT
    from oracle.unet import OracleUnet
    from oracle.edm import OracleEDM
    from audio_inpainting_diffusion_amd.init import seeded_normal
    n, bpo = args.network.cqt.num_octs, args.network.cqt.bins_per_oct
    L = args.exp.audio_len
    cores = torch.get_num_threads()
    orc = OracleUnet(n, bpo, OracleCQT(n, bpo, "oct", ("kaiser", 1), args.exp.sample_rate, L)).load_state_dict(net.state_dict())
    edm = OracleEDM()
    x = torch.from_numpy(seeded_normal(1, 0, L)).reshape(1, L) * 0.5
    y = torch.from_numpy(seeded_normal(2, 0, L)).reshape(1, L) * 0.063
    mask = torch.ones(1, L)
    mask[:, L // 2 - 3307: L // 2 + 3308] = 0
    s = torch.full((1, 1), 0.5)
    with torch.no_grad():
        edm.denoiser(x, orc, s)                      # warm-up
    t0 = time.time()
    if guided:
        n_timed = 1
        xr = x.clone().requires_grad_()
        xh = orc.CQTransform.apply_hpf_DC(edm.denoiser(xr, orc, s))
        if spectral:
            from oracle.sampler import spectral_mask_apply
            from audio_inpainting_diffusion_amd.masks import spectral_mask_from_args
            sm = spectral_mask_from_args(args)
            norm = torch.linalg.norm(spectral_mask_apply(y, sm) - spectral_mask_apply(xh, sm), dim=1, ord=2)
        else:
            norm = torch.linalg.norm(y * mask - mask * xh, dim=1, ord=2)
        torch.autograd.grad(norm.sum(), xr)
    else:
        n_timed = 2
        with torch.no_grad():
            for _ in range(n_timed):
                edm.denoiser(x, orc, s)
    dt = (time.time() - t0) / n_timed
    what = "guided (xi=0.25: forward with graph + input-VJP by torch.autograd)" if guided else "forward-only (xi=0)"
    return {"value": round(1.0 / dt, 4), "unit": "denoiser evaluations per second", "cores": cores, "kind": "port",
            "sample": f"B=1 full-size {args.exp.exp_name} network, {what}: 1 warm-up forward + {n_timed} timed evaluation(s), "
                      f"{dt:.2f} s each, torch {torch.__version__} CPU fp32, {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=0, help="segments per GPU (default: 8 / 16 / 4 by workload)")
    ap.add_argument("--xi", type=float, default=0.25, help="0.25 = reconstruction guidance (the reference tester's default, conf/tester/inpainting_tester.yaml:32); "
                                                          "0 = replacement branch (forward only)")
    ap.add_argument("--task", choices=["inpainting", "spectrogram"], default="inpainting",
                    help="inpainting = time-domain gap (the headline metric); spectrogram = STFT-domain mask "
                         "(predict_spectrogram_inpainting, conf/tester/inpainting_tester.yaml:78-87)")
    ap.add_argument("--workload", choices=["maestro22k", "librispeech16k", "musicnet44k"], default="maestro22k",
                    help="maestro22k = BASELINE.json configs[1] (the metric's configuration); librispeech16k = configs[3] "
                         "(16 kHz, 4 short gaps of 50 ms, T=70, batch 16); musicnet44k = configs[4] (44.1 kHz 8-octave "
                         "network, 1.5 s gap, T=128, batch 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--conv-table", action="store_true", help="print per-shape conv kernel times (stderr)")
    a = ap.parse_args()

    from audio_inpainting_diffusion_amd import dist as D
    from audio_inpainting_diffusion_amd.config import make_args
    from audio_inpainting_diffusion_amd.edm import EDM
    from audio_inpainting_diffusion_amd.init import seeded_init_, seeded_normal
    from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
    from audio_inpainting_diffusion_amd.sampler import Sampler

    rank, local, world = D.init_distributed()
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    local = local % max(1, torch.cuda.device_count())      # (several ranks may share a GPU in functional tests)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    T, gap_ms, B_def = {"maestro22k": (36, 300.0, 8), "librispeech16k": (70, 50.0, 16), "musicnet44k": (128, 1500.0, 4)}[a.workload]
    assert a.warmup + a.steps <= T - 1, "timed steps must be Heun steps (the last step of the schedule is Euler)"
    args = make_args(a.workload, audio_len=184184, T=T, gap_ms=gap_ms, xi=a.xi)
    L, B = args.exp.audio_len, (a.batch or B_def)

    net = Unet_CQT_oct_with_attention(args, dev)
    if rank == 0:
        seeded_init_(net, 0)                       # reference-scale gates (1e-7), like a fresh reference network
    t0 = time.time()
    nbytes = D.broadcast_parameters(net, src=0)    # RCCL broadcast of the flat fp32 weight buffer over xGMI
    torch.cuda.synchronize()
    t_bcast = time.time() - t0
    net.prepare()

    lo, hi = D.shard_range(world * B, rank, world)
    y = torch.stack([torch.from_numpy(seeded_normal(7, g, L)) for g in range(lo, hi)]) * 0.063
    from audio_inpainting_diffusion_amd.masks import mask_from_args, spectral_mask_from_args
    from audio_inpainting_diffusion_amd.sampler import prepare_smooth_mask
    smp = Sampler(model=net, diff_params=EDM(args), args=args)
    smp.seeds = D.item_seeds(1234, lo, hi)
    if a.task == "spectrogram":
        from audio_inpainting_diffusion_amd.stft import SpectralMask
        stc = args.tester.spectrogram_inpainting.stft
        smp.spectral = SpectralMask(spectral_mask_from_args(args), L, stc.n_fft, stc.hop_length, stc.win_length, stc.window, dev)
        smp.mask = smp.smask = None
        smp.y = smp.spectral.apply(y.to(dev).contiguous())
    else:
        mask = mask_from_args(args, generator=torch.Generator().manual_seed(99))     # tester_inpainting.py:231-254
        smp.mask = mask.to(dev)
        smp.y = (y * mask).to(dev).contiguous()
        smp.smask = prepare_smooth_mask(mask, args.tester.data_consistency.hann_size).to(dev).contiguous()

    state = smp.begin((B, L), dev)
    for i in range(a.warmup):
        smp.step(state, i)
    st = net._state(B)
    timing = []
    st["plan_body"].timing = timing
    if a.xi > 0:
        net._bwd_plan(st).timing = timing
    torch.cuda.synchronize()
    D.barrier()
    t0 = time.perf_counter()
    for i in range(a.warmup, a.warmup + a.steps):
        smp.step(state, i)
    torch.cuda.synchronize()
    D.barrier()
    wall = time.perf_counter() - t0
    st["plan_body"].timing = None
    if a.xi > 0:
        net._bwd_plan(st).timing = None
    wall = D.max_over_ranks(wall, dev)
    assert torch.isfinite(state["x"]).all()

    conv_ms = sum(t[0].elapsed_time(t[1]) for t in timing)
    conv_flops = sum(t[2] for t in timing)
    wino = getattr(net, "winograd_f4", None)
    # MFMA work actually issued: Winograd layers execute 6/12 (F(4,3)) or 4/6 (F(2,3)) of the direct products
    exec_flops = sum(t[2] * ((0.5 if wino else 2.0 / 3.0) if (t[3].startswith("conv 5x3") and "Cin2 " not in t[3] and "Cout2 " not in t[3]) else 1.0)
                     for t in timing)
    if a.conv_table and rank == 0:
        agg = {}
        for e0, e1, f, d, _nb in timing:
            r = agg.setdefault(d, [0, 0.0, 0])
            r[0] += 1; r[1] += e0.elapsed_time(e1); r[2] += f
        for d, (n, ms, f) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print("%-62s n=%3d  %8.3f ms/launch  %6.1f TF/s  %5.1f%% of conv time" % (d, n, ms / n, f / ms / 1e9, 100 * ms / conv_ms), file=sys.stderr)
    evals = world * B * 2 * a.steps
    if rank == 0:
        achieved = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        out = {
            "metric": "denoiser-steps/sec", "value": round(evals / wall, 3), "unit": "denoiser evaluations (one segment each) per second, whole job",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * wall / a.steps, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": {"maestro22k": "BASELINE.json configs[1]: batch %d x 22.05 kHz MAESTRO-shape segments (L=184184) per GPU, 300 ms gap, ",
                                    "librispeech16k": "BASELINE.json configs[3]: batch %d x 16 kHz LibriSpeech-shape segments (L=184184) per GPU, 4 gaps of 50 ms, ",
                                    "musicnet44k": "BASELINE.json configs[4]: batch %d x 44.1 kHz segments (L=184184) per GPU, 8-octave network, 1.5 s gap, "}[a.workload] % B
                                   + "T=%d EDM schedule, Heun steps %d..%d" % (T, a.warmup, a.warmup + a.steps - 1)
                                   + ("; STFT-domain mask (spectrogram inpainting)" if a.task == "spectrogram" else ""),
                       "branch": "xi=%g (%s)" % (a.xi, "reconstruction guidance: forward + input-VJP" if a.xi > 0 else "replacement / data-consistency: forward only"),
                       "segments_per_gpu": B, "evals_per_step": 2,
                       "network": "unet_cqt_oct_with_attention %s, %.1f M params, random-init (seeded)" % ("44.1 kHz 8-octave" if a.workload == "musicnet44k" else "7-octave", sum(p.numel() for p in net.parameters()) / 1e6),
                       "parallelism": "segments sharded %d-way, one process per GPU, weights broadcast once (%.0f MB in %.3f s), no collective in the loop" % (world, nbytes / 1e6, t_bcast)},
            "roofline": {"bound": "mfma", "kernel": "aid_conv2d: conv53_wino4v_kernel / conv53_wino4_kernel (Winograd F(4,3), 5x3 layers) + conv11_dma_kernel / conv_mfma_kernel (1x1, qk GEMMs), fp32 v_mfma_f32_32x32x2_f32",
                         "note": "achieved = ALGORITHMIC direct-form FLOPs / measured time; Winograd F(4,3) issues half of them as MFMAs (executed_mfma_tflops), so achieved can exceed the fp32 MFMA peak",
                         "achieved": round(achieved, 2), "peak": 157.3, "unit": "TFLOP/s", "frac": round(achieved / 157.3, 4),
                         "traffic": None, "launches": len(timing), "avg_launch_us": round(1e3 * conv_ms / max(1, len(timing)), 1),
                         "algorithmic_gflop_per_launch": round(conv_flops / max(1, len(timing)) / 1e9, 2),
                         "algorithmic_mb_per_launch": round(sum(t[4] for t in timing) / max(1, len(timing)) / 1e6, 1),
                         "executed_mfma_tflops": round(exec_flops / (conv_ms * 1e-3) / 1e12, 2) if conv_ms > 0 else 0.0,
                         "frac_executed_mfma": round(exec_flops / (conv_ms * 1e-3) / 1e12 / 157.3, 4) if conv_ms > 0 else 0.0,
                         "conv_time_fraction_of_wall": round(conv_ms * 1e-3 / wall, 3)},
        }
        tr = os.path.join(ROOT, "profiles", "r01_conv_traffic.json")     # PMC pass of this same command (see profiles/README.md)
        if os.path.exists(tr):
            try:
                out["roofline"]["traffic"] = json.load(open(tr)).get("hbm_bytes_per_conv_launch_corrected")
            except Exception:
                pass
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(net, args, guided=a.xi > 0, spectral=a.task == "spectrogram")
            out["cpu_baseline"]["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    D.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
