#!/usr/bin/env python
"""Benchmark of the EDM inpainting sampling hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--xi 0|0.25] [--batch 8] [--workload ...] [--gap-ms G]

N > 1: either the driver's form (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
127.0.0.1 ... bench.py --gpus N ...), or plain `python bench.py --gpus N`, which re-launches itself that way
(dist.launch_ranks: one rank per GPU over RCCL; when fewer than N GPUs are visible the ranks share them and the
collectives run on gloo -- a FUNCTIONAL run, labelled `functional_shared_gpu` in the output line).

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

roofline (fp32 MFMA, peak 157.3 TFLOP/s): every aid_conv2d launch of a single-stream pass (forward and VJP plans) is
bracketed by HIP events on the launch stream and attributed to the device kernel it dispatched to (aid_last_kernel).
The line reports the DOMINANT kernel BY ITS ROCPROF NAME (most GPU time: conv53_wino8r_kernel, the 5x3 layers in Winograd
F(8,3) form; conv53_wino4r_kernel = F(4,3) takes the launches whose tiles quantise better that way), over ALL its template instances / tile
kinds (`families` keeps the sub-family table):
    achieved = MFMA FLOPs that kernel ISSUES / its launch time, summed over every launch of >= 3 Heun steps
               (F(4,3) issues 6 products per 4 outputs x 3 taps = 1/2 of the direct-form FLOPs, F(8,3) 10 per 8 x 3 = 5/12; 1x1 / direct kernels all)
    frac     = achieved / 157.3
    algorithmic_tflops = direct-convolution FLOPs (2*B*F*T*Cin*Cout*KH*KW) / the same time -- may exceed the peak
    step_executed_frac = MFMA FLOPs issued by ALL conv / GEMM launches of a step / the step's wall time in the TIMED region / 157.3
`kernels` aggregates every conv kernel by name, `families` by name + tile kind, `all_conv` over all of them.  traffic = HBM bytes per launch
of the dominant kernel from the PMC passes of this same command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, --streams 1, FETCH
doubled for gfx950) committed under profiles/ -- counters cannot be read from inside this process; traffic_over_algorithmic relates it to this
run's algorithmic bytes per launch.
cpu_baseline: the CPU oracle (torch-CPU restatement of the reference path, oracle/) timed on this host's cores for the same
network and branch at B=1 (rank 0, N=1 only): per thread placement (physical cores of one socket / all physical cores) one warm-up and one
timed evaluation OF THE SAME BRANCH (a placement whose warm-up is already 1.5x slower is not timed again), then two more on the fastest placement; value = 1 / median of its three timings, `cores` = its thread count.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA = 157.3          # TFLOP/s, dense fp32 matrix (MI355X_MICROARCH.md)
# executed MFMA FLOPs as a fraction of the direct form: F(4,3) issues 6 products per 4 outputs x 3 taps = 1/2, F(8,3) 10 per 8 x 3 = 5/12
# the 2-D form F(4,5) x F(4,3) 48 per 16 x 15 = 1/5 (its GEMM is a plan node of its own: w2d_gemm_kernel; the output pass carries no FLOPs)
WINO_EXEC = {"conv53_wino4r_kernel": 0.5, "conv53_wino4v_kernel": 0.5, "conv53_wino4_kernel": 0.5, "conv53_wino8r_kernel": 10.0 / 24.0,
             "conv53_wino8r_ks_kernel": 10.0 / 24.0, "w2d_gemm_kernel": 48.0 / 240.0,
             "w2d_gemm_s6_kernel": 48.0 / 240.0}       # (variant: fp32-equivalent products; they run on the bf16 pipe, so its frac_of_fp32_mfma_peak may pass 1)


def ensure_built() -> None:
    """Build libaid_hip.so when it is missing or older than its sources (a clean checkout: the .so is git-ignored).  Every rank takes the same file
    lock, so one of them builds and the others find the library up to date; hipcc cross-compiles without a GPU."""
    import fcntl
    import build
    os.makedirs(os.path.join(ROOT, "audio_inpainting_diffusion_amd", "csrc", "build"), exist_ok=True)
    with open(os.path.join(ROOT, "audio_inpainting_diffusion_amd", "csrc", "build", ".lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            build.build(verbose=False)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)


def cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_thread_configs():
    """Thread placements the CPU baseline tries: the physical cores of ONE socket, then all physical cores (this process's affinity mask intersected
    with /proc/cpuinfo's physical id / core id); duplicates dropped.  [(label, [cpu ids])]"""
    try:
        avail = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return [("all", [])]
    cores = {}                                           # (socket, core) -> logical CPUs
    cur = {}
    try:
        for ln in open("/proc/cpuinfo").read().splitlines() + [""]:
            if not ln.strip():
                if "processor" in cur and int(cur["processor"]) in avail:
                    cores.setdefault((int(cur.get("physical id", 0)), int(cur.get("core id", cur["processor"]))), []).append(int(cur["processor"]))
                cur = {}
            elif ":" in ln:
                k, v = ln.split(":", 1)
                cur[k.strip()] = v.strip()
    except OSError:
        pass
    if not cores:
        return [("all logical", avail)]
    sockets = sorted({k[0] for k in cores})
    one = sorted(min(v) for k, v in cores.items() if k[0] == sockets[0])
    phys = sorted(min(v) for v in cores.values())
    out, seen = [], set()
    # (all logical CPUs = 2 SMT threads per core over both sockets is not offered when it oversubscribes the cores: measured once on the 2 x 64-core
    #  EPYC 9575F box, 879 s per evaluation against 18 s on one socket -- profiles/r04_bench_guided_first.json)
    for label, cpus in (("physical cores of one socket", one), ("all physical cores", phys)) + ((("all logical CPUs", avail),) if len(avail) == len(phys) else ()):
        if tuple(cpus) not in seen:
            seen.add(tuple(cpus))
            out.append((label, cpus))
    return out


def cpu_baseline(net, args, guided: bool, spectral: bool = False, n_timed: int = 3):
    """The CPU oracle (oracle/: torch-CPU restatement of the reference path) timed on this host for the same
    full-size network at B=1, on the branch the GPU number is quoted on (guided: forward with autograd graph + input gradient;
    xi=0: forward only).  Thread placement: cpu_thread_configs() are each given one warm-up and one timed evaluation; the FASTEST
    placement gets the remaining timed evaluations, and value = 1 / median of its `n_timed` timings (the thread count that is fastest,
    not the largest: all-cores fp32 convolution on a two-socket box is not the CPU's best)."""
    import torch
    from oracle.nsgt_cqt import OracleCQT
    from oracle.unet import OracleUnet
    from oracle.edm import OracleEDM
    from audio_inpainting_diffusion_amd.init import seeded_normal
    n, bpo = args.network.cqt.num_octs, args.network.cqt.bins_per_oct
    L = args.exp.audio_len
    affinity0 = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None
    threads0 = torch.get_num_threads()
    orc = OracleUnet(n, bpo, OracleCQT(n, bpo, "oct", ("kaiser", 1), args.exp.sample_rate, L)).load_state_dict(net.state_dict())
    edm = OracleEDM()
    x = torch.from_numpy(seeded_normal(1, 0, L)).reshape(1, L) * 0.5
    y = torch.from_numpy(seeded_normal(2, 0, L)).reshape(1, L) * 0.063
    mask = torch.ones(1, L)
    mask[:, L // 2 - 3307: L // 2 + 3308] = 0
    s = torch.full((1, 1), 0.5)
    def place(cpus):
        if cpus and affinity0 is not None:
            os.sched_setaffinity(0, cpus)
        torch.set_num_threads(len(cpus) if cpus else threads0)

    def one_eval():
        t0 = time.time()
        if guided:
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
            with torch.no_grad():
                edm.denoiser(x, orc, s)
        return time.time() - t0

    sweep = []
    try:
        for label, cpus in cpu_thread_configs():
            place(cpus)
            tw = one_eval()                              # warm-up of the same branch at this placement (autograd / oneDNN primitives) -- not counted ...
            if sweep and tw > 1.5 * min(r[0] for r in sweep):
                sweep.append((tw, label + " (warm-up evaluation only: already 1.5x slower)", cpus))      # ... unless it already rules the placement out
                continue
            sweep.append((one_eval(), label, cpus))
        best = min(sweep, key=lambda r: r[0])
        place(best[2])
        times = [best[0]] + [one_eval() for _ in range(max(0, n_timed - 1))]
    finally:
        if affinity0 is not None:
            os.sched_setaffinity(0, affinity0)
        torch.set_num_threads(threads0)
    cores = len(best[2]) if best[2] else threads0
    dt = sorted(times)[len(times) // 2]
    what = "guided (xi=0.25: forward with graph + input-VJP by torch.autograd)" if guided else "forward-only (xi=0)"
    return {"value": round(1.0 / dt, 4), "unit": "denoiser evaluations per second", "cores": cores, "kind": "port",
            "cpu_model": cpu_model(), "placement": best[1], "seconds_per_evaluation": [round(t, 2) for t in times],
            "thread_sweep": [{"placement": lb, "threads": len(c) if c else threads0, "seconds": round(t, 2)} for t, lb, c in sweep],
            "median_s": round(dt, 2), "min_s": round(min(times), 2),
            "sample": f"B=1 full-size {args.exp.exp_name} network, {what}: per thread placement 1 warm-up + 1 timed evaluation, then {max(0, n_timed - 1)} more "
                      f"on the fastest ({best[1]}, {cores} threads; median {dt:.2f} s, min {min(times):.2f} s; value = 1 / median), torch {torch.__version__} CPU fp32"}


def exec_fraction(kn: str) -> float:
    """MFMA FLOPs a conv launch ISSUES as a fraction of its direct-form FLOPs, from the kernel name aid_last_kernel reported: the 2-D form's GEMM over 80 planes
    (F(4,5) x F(8,3), names ending in +t8) issues 80 products per 4 x 8 outputs x 15 taps = 1/6, over 48 planes 1/5; the fused 1-D kernels 1/2 and 5/12"""
    if kn.startswith("w2d_gemm") and kn.endswith("+t8"):
        return 80.0 / 480.0
    return WINO_EXEC.get(kernel_base(kn), 1.0)


def kernel_base(kn: str) -> str:
    """device-kernel name as rocprofv3 lists it, without the tile kind / template instance aid_last_kernel appends"""
    return kn.split("(")[0].split("+")[0].split("<")[0]


def family_table(timing, by_kernel=False):
    """timing: (event0, event1, algorithmic_flops, description, algorithmic_bytes, kernel_name) per conv launch.
    by_kernel: aggregate by the device kernel's name (what rocprofv3 lists) instead of name + tile kind."""
    fam = {}
    for e0, e1, fl, _d, nb, kn in timing:
        base = kernel_base(kn)
        r = fam.setdefault(base if by_kernel else kn, dict(launches=0, ms=0.0, alg=0.0, exe=0.0, bytes=0.0))
        r["launches"] += 1
        r["ms"] += e0.elapsed_time(e1)
        r["alg"] += fl
        r["exe"] += fl * exec_fraction(kn)
        r["bytes"] += nb
    out = {}
    for kn, r in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]):
        sec = r["ms"] * 1e-3
        out[kn] = {"launches": r["launches"], "avg_launch_us": round(1e3 * r["ms"] / r["launches"], 1), "time_ms": round(r["ms"], 2),
                   "algorithmic_gflop_per_launch": round(r["alg"] / r["launches"] / 1e9, 2),
                   "executed_mfma_tflops": round(r["exe"] / sec / 1e12, 2), "frac_of_fp32_mfma_peak": round(r["exe"] / sec / 1e12 / PEAK_F32_MFMA, 4),
                   "algorithmic_tflops": round(r["alg"] / sec / 1e12, 2),
                   "algorithmic_mb_per_launch": round(r["bytes"] / r["launches"] / 1e6, 1), "algorithmic_tb_per_s": round(r["bytes"] / sec / 1e12, 3)}
    return out


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
                         "(16 kHz, 4 short gaps, T=70, batch 16); musicnet44k = configs[4] (44.1 kHz 8-octave "
                         "network, 1.5 s gap, T=128, batch 4)")
    ap.add_argument("--gap-ms", type=float, default=0.0, help="gap length in ms (default by workload: 300 / 50 / 1500; "
                                                              "configs[3] sweeps 25 / 50 / 100, conf/tester/inpainting_tester_shortgaps.yaml:74-75)")
    ap.add_argument("--streams", type=int, default=0, help="sub-batch HIP streams per evaluation (default: automatic, network._n_split; 1 = plain single-stream schedule)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--roof-steps", type=int, default=3, help="Heun steps of the single-stream roofline pass after the timed region (>= 1; 3 by default)")
    ap.add_argument("--no-graphs", action="store_true", help="(the default since round 5) eager launches at small batches")
    ap.add_argument("--graphs", action="store_true", help="A/B: HIP-graph replay of the evaluation at small batches (network.use_graphs; the default until round 4)")
    ap.add_argument("--no-pair-merge", action="store_true", help="A/B: separate input-gradient convs for proj_in and res_conv")
    ap.add_argument("--no-lanes", action="store_true", help="A/B: single-stream launch plans at small batches (plan.py lanes off)")
    ap.add_argument("--no-fused-norm-bwd", action="store_true", help="A/B: separate gate / Winograd-transform pre-pass before every dgrad conv")
    ap.add_argument("--no-fold-copies", action="store_true", help="A/B: launch every first-contribution scaled gradient copy of the reverse sweep (network.fold_grad_copies = False)")
    ap.add_argument("--no-fin", action="store_true", help="A/B: separate aid_group_stats / coefficient launches instead of the last tile of a sample folding the epilogue partials")
    ap.add_argument("--no-epilogue-stats", action="store_true", help="A/B: group statistics by their own read pass instead of the conv epilogue")
    ap.add_argument("--wino-forms", default="4,8,45,85", help="A/B: Winograd forms the 5x3 layers may use (default 4,8,45,85: the 2-D forms F(4,5) x F(8,3) / F(4,5) x F(4,3) and the fused F(8,3) where the library prefers them; "
                    "4,8,45: round 5's set; 4,8: the fused 1-D kernels only; 4: F(4,3) everywhere)")
    ap.add_argument("--mfma-split", type=int, default=0, choices=[0, 6], help="LABELLED VARIANT (never the default): 6 = the 2-D Winograd form's GEMMs on three bf16 pieces per fp32 operand, "
                    "six bf16 MFMA products, fp32 accumulation (aid_wino2d_set_split); the JSON's dtype says so")
    ap.add_argument("--w2d-min-channels", type=int, default=0, help="A/B: 256 keeps the 2-D Winograd form off the K = 128 levels (network.w2d_min_channels)")
    ap.add_argument("--w2d-force-max-t", type=int, default=0, help="A/B: the 2-D Winograd form on every SUPPORTED layer with T up to this (network.w2d_force_max_T)")
    ap.add_argument("--w2d-c96-max-t", type=int, default=-1, help="A/B: the 96-channel levels with T up to this take the 2-D Winograd form (network.w2d_c96_max_T)")
    ap.add_argument("--lanes-max-batch", type=int, default=-1, help="A/B: run the two lanes of the launch plans on two streams for (sub-)batches up to this size (network.lanes_max_batch; default 3)")
    ap.add_argument("--lanes-in-sub-batches", action="store_true", help="A/B: two-lane launch plans inside the sub-batch streams too (network.lanes_in_sub_batches)")
    ap.add_argument("--split", default="", help="A/B: explicit sub-batch sizes, e.g. 5,3 (network.split_sizes; implies --streams = their count)")
    ap.add_argument("--cu-partition", type=int, default=0, help="EXPERIMENT (A/B): spatial partition of the chip inside every sub-batch -- MFMA-bound kernels on 256 - N CUs, HBM-bound passes "
                    "on the last N / 8 CUs of every XCD, on CU-masked streams (network.cu_partition, streams.py); 0 = free-running sub-batch streams (the product schedule)")
    ap.add_argument("--cu-split", default="", help="EXPERIMENT (A/B): CUs per XCD for each sub-batch stream, e.g. 16,16 (disjoint halves) or 20,20 (overlapping) -- CU-masked sub-batch streams "
                    "without any cross-stream events (network.cu_split)")
    ap.add_argument("--conv-table", action="store_true", help="print per-shape conv kernel times (stderr)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as a plain process: become the launcher of N ranks of this same command
        from audio_inpainting_diffusion_amd.dist import launch_ranks
        sys.exit(launch_ranks(a.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))

    import torch
    ensure_built()
    from audio_inpainting_diffusion_amd import _lib
    from audio_inpainting_diffusion_amd import dist as D
    from audio_inpainting_diffusion_amd.config import make_args
    from audio_inpainting_diffusion_amd.edm import EDM
    from audio_inpainting_diffusion_amd.init import seeded_init_, seeded_normal
    from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
    from audio_inpainting_diffusion_amd.sampler import Sampler

    rank, local, world = D.init_distributed()
    if world > 1:                                      # N ranks share the host: each rank's CPU threads (noise generation, launch loop) go to its GPU's NUMA node
        if D.bind_rank_to_gpu_numa(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)), log=sys.stderr) is None:
            torch.set_num_threads(max(1, (os.cpu_count() or 8) // world))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    shared = int(os.environ.get("AID_SHARED_GPU", "0")) or int(world > torch.cuda.device_count())      # set by launch_ranks when ranks have to share GPUs; also true
                                                                                                        # under an external launcher with AID_DIST_BACKEND=gloo on fewer GPUs than ranks
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    T, gap_def, B_def = {"maestro22k": (36, 300.0, 8), "librispeech16k": (70, 50.0, 16), "musicnet44k": (128, 1500.0, 4)}[a.workload]
    gap_ms = a.gap_ms or gap_def
    assert a.warmup >= 0 and a.steps >= 1
    args = make_args(a.workload, audio_len=184184, T=T, gap_ms=gap_ms, xi=a.xi)
    L, B = args.exp.audio_len, (a.batch or B_def)

    net = Unet_CQT_oct_with_attention(args, dev)
    if a.mfma_split:
        assert _lib.lib().aid_wino2d_set_split(a.mfma_split) == 0
    if a.w2d_min_channels:
        net.w2d_min_channels = a.w2d_min_channels
    if a.w2d_force_max_t:
        net.w2d_force_max_T = a.w2d_force_max_t
    if a.w2d_c96_max_t >= 0:
        net.w2d_c96_max_T = a.w2d_c96_max_t
    if a.cu_partition:
        net.cu_partition = a.cu_partition
    if a.cu_split:
        net.cu_split = tuple(int(v) for v in a.cu_split.split(","))
    if a.no_epilogue_stats:
        net.epilogue_stats = False
    if a.no_fin:
        net.fuse_fin = False
    if a.no_fold_copies:
        net.fold_grad_copies = False
    if a.no_fused_norm_bwd:
        net.fuse_norm_bwd_wino = False
    if a.streams:
        net.split_streams = a.streams
    net.wino_forms = tuple(int(v) for v in a.wino_forms.split(","))
    if a.split:
        net.split_sizes = tuple(int(v) for v in a.split.split(","))
        a.streams = net.split_streams = len(net.split_sizes)
    if a.no_graphs:
        net.use_graphs = False
    if a.graphs:
        net.use_graphs = True
    if a.no_pair_merge:
        net.merge_pair_dgrad = False
    if a.no_lanes:
        net.lanes_max_batch = 0
    if a.lanes_max_batch >= 0:
        net.lanes_max_batch = a.lanes_max_batch
    if a.lanes_in_sub_batches:
        net.lanes_in_sub_batches = True
    if rank == 0:
        seeded_init_(net, 0)                       # reference-scale gates (1e-7), like a fresh reference network
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    nbytes = D.broadcast_parameters(net, src=0)    # ONE in-place broadcast of the flat fp32 weight buffer (RCCL over xGMI)
    torch.cuda.synchronize()
    t_bcast = time.time() - t0
    net.prepare()

    lo, hi = D.shard_range(world * B, rank, world)
    y = torch.stack([torch.from_numpy(seeded_normal(7, g, L)) for g in range(lo, hi)]) * 0.063
    from audio_inpainting_diffusion_amd.masks import mask_from_args, spectral_mask_from_args
    smp = Sampler(model=net, diff_params=EDM(args), args=args)
    smp.seeds = D.item_seeds(1234, lo, hi)
    if a.task == "spectrogram":
        smp.setup_spectrogram_inpainting(y.to(dev), spectral_mask_from_args(args), observed_is_clean=True)
    else:
        mask = mask_from_args(args, generator=torch.Generator().manual_seed(99))     # tester_inpainting.py:231-254
        smp.setup_inpainting((y * mask).to(dev), mask)

    # Every timed step is a Heun step (2 evaluations; the last step of a schedule is Euler): positions 0 .. T-4 of the schedule are walked,
    # and when --warmup + --steps exceed them the sampler starts the next batch's trajectory (smp.begin) inside the run, as a job over many
    # batches of segments does.  Two positions stay in reserve for the single-stream roofline pass.
    ROOF_STEPS = max(1, a.roof_steps)                  # Heun steps of the single-stream roofline pass (after one warm-up step)
    span = T - 2 - (ROOF_STEPS + 1)
    if a.cu_split or a.cu_partition:
        # hipExtStreamCreateWithCUMask makes BLOCKING streams: they synchronise implicitly with the NULL stream, which is torch's default stream -- every
        # element-wise launch of the sampler between two evaluations then drains and gates both masked streams through the runtime's legacy-stream path
        # (first A/B of the round: 16 + 16 CUs per XCD 29 evaluations/s; the same two evaluations without null-stream launches in between: level with
        # free-running streams, profiles/r06_cu_mask_e2e_probe.txt).  The experiment therefore runs its main stream on a non-blocking pool stream.
        torch.cuda.set_stream(torch.cuda.Stream())
    state = smp.begin((B, L), dev)

    def do_step(i):
        nonlocal state
        j = i % span
        if j == 0 and i > 0:
            state = smp.begin((B, L), dev)
        smp.step(state, j)
        return j

    for i in range(a.warmup):
        do_step(i)
    n_split = net._n_split(B)
    timing = []
    graphs = bool(net.use_graphs and B <= net.GRAPH_MAX_B and n_split == 1)     # small batches replay a captured HIP graph
    torch.cuda.synchronize()
    D.barrier()
    t0 = time.perf_counter()
    jlast = 0
    for i in range(a.warmup, a.warmup + a.steps):
        jlast = do_step(i)
    host_enqueue = time.perf_counter() - t0            # the host thread's share: all launches of the timed steps are enqueued (the GPU may still be running them)
    torch.cuda.synchronize()
    D.barrier()
    wall_rank = wall = time.perf_counter() - t0
    wall = D.max_over_ranks(wall, dev)
    assert torch.isfinite(state["x"]).all()
    # Kernel speeds are measured right after the timed region: the SAME sampler continues for one warm-up and ROOF_STEPS measured
    # Heun steps with the network forced onto one stream and eager launches (the product schedule overlaps kernels of different
    # sub-batches on concurrent HIP streams, or replays a captured HIP graph: a launch's start-to-end time there is not the kernel's speed).
    roofline_pass = ("separate single-stream pass of eager launches: 1 warm-up + %d measured Heun steps (steps %d..%d of the same run) right after "
                     "the timed region, %s" % (ROOF_STEPS, jlast + 2, jlast + 1 + ROOF_STEPS,
                                                "which replays a captured HIP graph per evaluation" if graphs else
                                                ("whose %d sub-batch streams overlap kernels" % n_split if n_split > 1 else "which runs the same single-stream schedule")))
    assert jlast + 1 + ROOF_STEPS <= T - 2
    net.split_streams, use_graphs = 1, net.use_graphs
    net.use_graphs = False
    smp.step(state, jlast + 1)
    for pl in net.timed_plans(B, a.xi > 0):
        pl.timing = timing
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for k in range(ROOF_STEPS):
        smp.step(state, jlast + 2 + k)
    torch.cuda.synchronize()
    wall_serial = (time.perf_counter() - t1) / ROOF_STEPS
    for pl in net.timed_plans(B, a.xi > 0):
        pl.timing = None
    net.split_streams, net.use_graphs = (a.streams or None), use_graphs

    if a.conv_table and rank == 0:
        agg = {}
        conv_ms = sum(t[0].elapsed_time(t[1]) for t in timing)
        for e0, e1, f, d, _nb, kn in timing:
            r = agg.setdefault(d + "  -> " + kn, [0, 0.0, 0])
            r[0] += 1; r[1] += e0.elapsed_time(e1); r[2] += f
        for d, (n, ms, f) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print("%-100s n=%3d  %8.3f ms/launch  %6.1f TF/s  %5.1f%% of conv time" % (d, n, ms / n, f / ms / 1e9, 100 * ms / conv_ms), file=sys.stderr)
    evals = world * B * 2 * a.steps
    # every rank's own line (where it ran, its own wall clock, broadcast time, single-stream step): a slow SCALE line can be read without a rerun
    ranks = D.gather_objects(D.rank_report(rank, local, wall_s=wall_rank, bcast_s=t_bcast, ms_per_step=1e3 * wall_rank / a.steps,
                                           single_stream_ms_per_step=1e3 * wall_serial, segments=[lo, hi]))
    if rank == 0:
        problems = D.placement_problems(ranks, shared=bool(shared), shared_requested=bool(int(os.environ.get("AID_SHARED_GPU", "0"))),
                                        visible_gpus=torch.cuda.device_count())
        for msg in problems["warnings"]:
            print("bench: WARNING: " + msg, file=sys.stderr)
        assert not problems["errors"], "; ".join(problems["errors"])
        fams = family_table(timing)
        kerns = family_table(timing, by_kernel=True)
        dom_name = next((k for k, v in kerns.items() if v["executed_mfma_tflops"] > 0), None)      # the MFMA kernel with the most GPU time (the 2-D form's output pass is timed as conv time but issues no MFMA)
        dom = kerns.get(dom_name, {})
        conv_ms = sum(v["time_ms"] for v in fams.values())
        alg = sum(t[2] for t in timing)
        exe = sum(t[2] * exec_fraction(t[5]) for t in timing)
        sec = max(conv_ms * 1e-3, 1e-12)
        out = {
            "metric": "denoiser-steps/sec", "value": round(evals / wall, 3), "unit": "denoiser evaluations (one segment each) per second, whole job",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * wall / a.steps, 2), "host_enqueue_ms_per_step": round(1e3 * host_enqueue / a.steps, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic",
            "dtype": "f32" if not a.mfma_split else "VARIANT: f32 storage and accumulation; the GEMMs of the 2-D Winograd layers multiply three bf16 pieces per f32 operand (exact split), six bf16-MFMA products each; all other kernels f32",
            "config": {"workload": {"maestro22k": "BASELINE.json configs[1]: batch %d x 22.05 kHz MAESTRO-shape segments (L=184184) per GPU, %g ms gap, ",
                                    "librispeech16k": "BASELINE.json configs[3]: batch %d x 16 kHz LibriSpeech-shape segments (L=184184) per GPU, 4 gaps of %g ms, ",
                                    "musicnet44k": "BASELINE.json configs[4]: batch %d x 44.1 kHz segments (L=184184) per GPU, 8-octave network, %g ms gap, "}[a.workload] % (B, gap_ms)
                                   + "T=%d EDM schedule, Heun steps %d..%d%s" % (T, a.warmup, a.warmup + a.steps - 1,
                                                                                  "" if a.warmup + a.steps <= T - 3 else " (positions wrap at %d: the next batch's trajectory starts inside the run)" % (T - 3))
                                   + ("; STFT-domain mask (spectrogram inpainting)" if a.task == "spectrogram" else ""),
                       "branch": "xi=%g (%s)" % (a.xi, "reconstruction guidance: forward + input-VJP" if a.xi > 0 else "replacement / data-consistency: forward only"),
                       "segments_per_gpu": B, "evals_per_step": 2,
                       "network": "unet_cqt_oct_with_attention %s, %.1f M params, random-init (seeded)" % ("44.1 kHz 8-octave" if a.workload == "musicnet44k" else "7-octave", sum(p.numel() for p in net.parameters()) / 1e6),
                       "parallelism": "segments sharded %d-way, one process per GPU, weights broadcast once in place (%.0f MB in %.3f s, %s), no collective in the loop"
                                      % (world, nbytes / 1e6, t_bcast, (torch.distributed.get_backend() if world > 1 else "single process")),
                       "sub_batch_streams": n_split, "cu_partition": a.cu_partition, "cu_split": a.cu_split or None, "hip_graph_replay": graphs, "functional_shared_gpu": bool(shared),
                       "backend": (torch.distributed.get_backend() if world > 1 else "single process"), "rccl_version": D.rccl_version(),
                       "visible_gpus": torch.cuda.device_count()},
            "ranks": ranks,
            "roofline": {"bound": "mfma", "kernel": dom_name, "measured_in": roofline_pass,
                         "definition": "achieved = MFMA FLOPs issued by ALL launches of the dominant kernel (the MFMA-issuing conv / GEMM kernel with the most GPU time, by device-kernel name, every template instance / tile kind) "
                                       "/ their summed duration (HIP events around each launch); Winograd F(4,3) issues 1/2, F(8,3) 5/12 and the 2-D forms (w2d_gemm_kernel: every 5x3 layer "
                                       "of the C >= 128 levels 3-6 and the bottleneck that the library gives it) F(4,5) x F(4,3) 1/5 and F(4,5) x F(8,3) (instances named +t8) 1/6 of the direct-form FLOPs; the 2-D form's output-transform + epilogue pass "
                                       "(w2d_output_kernel, HBM-bound, no FLOPs of its own) is timed as a conv launch too and counts in all_conv / conv_time_fraction_of_wall -- the fused 1-D kernels do that "
                                       "work inside the timed kernel -- its input-transform pass is the counterpart of the fused kernels' aid_scale_act pre-pass and, like it, is not; "
                                       "algorithmic_tflops = direct-form FLOPs / the same time; step_executed_frac = issued MFMA FLOPs of all conv / GEMM launches "
                                       "of one step / ms_per_step of the timed region / peak; non_winograd_conv_time_fraction_single_stream = 1 - (time of ALL 5x3 Winograd MFMA kernels: "
                                       "the fused 1-D kernels and the 2-D form's GEMM) / single-stream step time",
                         "step_executed_frac": round(exe / ROOF_STEPS / (wall / a.steps) / 1e12 / PEAK_F32_MFMA, 4),
                         "step_executed_tflops": round(exe / ROOF_STEPS / (wall / a.steps) / 1e12, 2),
                         "non_winograd_conv_time_fraction_single_stream": round(1.0 - sum(v["time_ms"] for k, v in kerns.items() if k in WINO_EXEC) / ROOF_STEPS / (1e3 * wall_serial), 3),
                         "non_dominant_kernel_time_fraction_single_stream": round(1.0 - dom.get("time_ms", 0.0) / ROOF_STEPS / (1e3 * wall_serial), 3),
                         "achieved": dom.get("executed_mfma_tflops"), "peak": PEAK_F32_MFMA, "unit": "TFLOP/s", "frac": dom.get("frac_of_fp32_mfma_peak"),
                         "algorithmic_tflops": dom.get("algorithmic_tflops"), "launches": dom.get("launches"), "avg_launch_us": dom.get("avg_launch_us"),
                         "algorithmic_gflop_per_launch": dom.get("algorithmic_gflop_per_launch"),
                         **_traffic_fields(dom_name, dom),
                         "share_of_conv_time": round(dom.get("time_ms", 0.0) / max(conv_ms, 1e-9), 3),
                         "all_conv": {"launches": len(timing), "executed_mfma_tflops": round(exe / sec / 1e12, 2), "frac_of_fp32_mfma_peak": round(exe / sec / 1e12 / PEAK_F32_MFMA, 4),
                                      "algorithmic_tflops": round(alg / sec / 1e12, 2), "conv_time_fraction_of_wall": round(sec / (ROOF_STEPS * wall_serial), 3),
                                      "single_stream_ms_per_step": round(1e3 * wall_serial, 2)},
                         "kernels": kerns, "families": fams},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(net, args, guided=a.xi > 0, spectral=a.task == "spectrogram")
            out["cpu_baseline"]["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    D.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def _kernel_sources_sha16():
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "audio_inpainting_diffusion_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "audio_inpainting_diffusion_amd", "csrc", "*.h"))
                    + [os.path.join(ROOT, "include", "aid_kernels.h")]):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _traffic_fields(dom_name, dom):
    """roofline.traffic = HBM bytes per launch of the dominant kernel from the PMC passes committed under profiles/ (FETCH_SIZE and WRITE_SIZE in
    separate rocprofv3 runs of this command with --streams 1, FETCH doubled as MI355X_MICROARCH.md prescribes for gfx950) -- counters cannot be read
    from inside this process.  Reported ONLY when that profile was taken on THIS build (it records the hash of the kernel sources, tools/conv_traffic.py);
    otherwise traffic is null and traffic_from_profile says which profile exists and why it was not used (no stale numbers)."""
    tp = _profile_traffic()
    same = bool(tp and tp.get("kernel_sources_sha16") and tp["kernel_sources_sha16"] == _kernel_sources_sha16())
    per = ((tp or {}).get("per_kernel_bytes") or {}) if same else {}
    t = per.get(dom_name) if dom_name else None
    alg = dom.get("algorithmic_mb_per_launch")
    if tp is not None:
        tp = dict(tp, used=same, note=("same kernel sources as this run" if same else "profile of ANOTHER build (kernel sources differ): not reported as this run's traffic"))
    return {"traffic": t, "traffic_unit": "HBM bytes per launch (PMC passes of this build, committed profile)", "traffic_source": (tp or {}).get("file") if same else None,
            "traffic_over_algorithmic": (round(t / (alg * 1e6), 3) if (t and alg) else None), "traffic_from_profile": tp}


def _profile_traffic():
    """HBM bytes per launch of the conv kernels from the newest committed PMC passes of this command (profiles/), or null."""
    for name in ("r06_conv_traffic.json", "r05_conv_traffic.json", "r04_conv_traffic.json", "r03_conv_traffic.json", "r02_conv_traffic.json", "r01_conv_traffic.json"):
        tr = os.path.join(ROOT, "profiles", name)
        if os.path.exists(tr):
            try:
                j = json.load(open(tr))
                return {"file": "profiles/" + name, "kernel_sources_sha16": j.get("kernel_sources_sha16"), "command": j.get("command"),
                        "hbm_bytes_per_conv_launch_corrected": j.get("hbm_bytes_per_conv_launch_corrected"),
                        "per_kernel_bytes": j.get("per_kernel_hbm_bytes_corrected")}
            except Exception:
                return None
    return None


if __name__ == "__main__":
    main()
