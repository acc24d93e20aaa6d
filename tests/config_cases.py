"""Recipes of the BASELINE.json configuration fixtures (tests/golden/config_evals.npz): the same seeded inputs are rebuilt by the generator
(tests/golden/make_config_golden.py: CPU oracle, build container) and by the GPU tests (tests/test_gpu_configs.py) -- no tensor travels,
only seeds.  A case = one workload + mask setting; it lists the denoiser evaluations of the first two Heun steps of its schedule
(t_hat_0, t_1, t_hat_1, t_2) and which batch items the fixture holds.  The input state of evaluation k is SYNTHETIC and reproducible:
x_k[b] = clean[b] + t_k * seeded_normal(900 + k, b) -- a noisy state at that noise level (the real trajectory's inputs depend on the previous
outputs at rounding level, so they could not be regenerated from seeds)."""
import numpy as np
import torch

CASES = {
    # name: (workload, T, gap_ms, batch, items in the fixture, network seed, base stream of the observations)
    "config1": ("maestro22k", 36, 300.0, 8, tuple(range(8)), 0, 51),
    "config3_gap25": ("librispeech16k", 70, 25.0, 16, (11,), 0, 32),
    "config3_gap50": ("librispeech16k", 70, 50.0, 16, (11,), 0, 32),
    "config3_gap100": ("librispeech16k", 70, 100.0, 16, (11,), 0, 32),
    "config4": ("musicnet44k", 128, 1500.0, 4, (0, 2), 1, 42),
}
N_PROJ, STRIDE = 8, 97


def build(name):
    """-> dict(args, B, items, net_seed, mask [1|B, L], y [B, L] (masked observations), evals = [(t_k, x_k [B, L])] * 4)"""
    from audio_inpainting_diffusion_amd.config import make_args
    from audio_inpainting_diffusion_amd.edm import EDM
    from audio_inpainting_diffusion_amd.init import seeded_normal
    from audio_inpainting_diffusion_amd.masks import mask_from_args
    wl, T, gap, B, items, net_seed, ystream = CASES[name]
    args = make_args(wl, T=T, gap_ms=gap, xi=0.25)
    L = args.exp.audio_len
    if args.tester.inpainting.mask_mode == "short":          # per-item gap positions (tester_inpainting.py:240-250: torch.randint per segment), kept
        hann = int(args.tester.data_consistency.hann_size)   # one cross-fade length away from the segment's ends: prepare_smooth_mask -- the reference's
        gap = int(gap * args.exp.sample_rate / 1000)         # (:302-325) and ours -- raises when a Hann ramp would not fit inside the segment
        mask = torch.ones(B, L)
        for b in range(B):
            starts = torch.randint(hann, L - gap - hann, (int(args.tester.inpainting.short.num_gaps),), generator=torch.Generator().manual_seed(40 + b))
            for st in starts.tolist():
                mask[b, st:st + gap] = 0
    else:
        mask = mask_from_args(args)
    clean = torch.stack([torch.from_numpy(seeded_normal(ystream, b, L)) for b in range(B)]) * 0.063
    y = clean * mask
    edm = EDM(args)
    tp = args.tester.diff_params                              # (what Sampler.update_diff_params installs)
    edm.sigma_min, edm.sigma_max, edm.ro, edm.sigma_data = tp.sigma_min, tp.sigma_max, tp.ro, tp.sigma_data
    edm.Schurn, edm.Stmin, edm.Stmax, edm.Snoise = tp.Schurn, tp.Stmin, tp.Stmax, tp.Snoise
    t = edm.create_schedule(T)
    g = edm.get_gamma(t)
    ts = [t[0] + g[0] * t[0], t[1], t[1] + g[1] * t[1], t[2]]
    evals = []
    for k, tk in enumerate(ts):
        noise = torch.stack([torch.from_numpy(seeded_normal(900 + k, b, L)) for b in range(B)])
        evals.append((tk.to(torch.float32), clean + float(tk) * noise))
    return dict(args=args, B=B, items=items, net_seed=net_seed, mask=mask, y=y, evals=evals, L=L)


def summarise(v, stream):
    """what the fixture stores of one [L] vector: N_PROJ seeded projections + squared norm (fp64) and every STRIDE-th sample"""
    from audio_inpainting_diffusion_amd.init import seeded_normal
    tv = v.detach().double().reshape(-1).numpy()
    probes = np.stack([seeded_normal(7100 + j, stream, tv.size) for j in range(N_PROJ)]).astype(np.float64)
    return np.concatenate([probes @ tv, [float(tv @ tv)]]), v.detach().reshape(-1)[::STRIDE].numpy().astype(np.float32)


def compare(v, proj, strided, stream):
    """(strided rel-L2, max projection difference / |reference|, relative squared-norm difference) of a [L] vector against its fixture entry"""
    got_p, got_s = summarise(v, stream)
    ref_n = np.sqrt(proj[-1])
    ds = float(np.linalg.norm(got_s.astype(np.float64) - strided) / max(np.linalg.norm(strided), 1e-30))
    return ds, float(np.abs(got_p[:N_PROJ] - proj[:N_PROJ]).max() / ref_n), float(abs(got_p[-1] - proj[-1]) / proj[-1])
