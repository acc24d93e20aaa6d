#!/usr/bin/env python
"""Generate tests/golden/config_evals.npz with the CPU ORACLE (oracle/, itself pinned to the imported reference by make_golden.py's fixtures
incl. the full-size guided evaluation) in the build container:

    python tests/golden/make_config_golden.py [case ...]        (default: every case of tests/config_cases.py; ~40 min on 8 cores)

For every case (BASELINE.json configs[1], configs[3] at 25 / 50 / 100 ms gaps, configs[4]) and every fixture item: the PROJECTED x_hat of the
four denoiser evaluations of the first two Heun steps -- reconstruction guidance (norm, autograd gradient, normalised step) and the
data-consistency projection included, i.e. exactly what OracleSampler.get_score returns -- as 8 seeded projections + squared norm + every
97th sample.  A full-size oracle item-evaluation is 25-45 s of CPU, so the GPU box never runs these: its tests rebuild the inputs from the
seeds (tests/config_cases.py) and compare against this file."""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from config_cases import CASES, build, summarise  # noqa: E402


def main():
    from audio_inpainting_diffusion_amd.init import seeded_state_dict
    from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
    from oracle.edm import OracleEDM
    from oracle.nsgt_cqt import OracleCQT
    from oracle.sampler import OracleSampler, smooth_mask_rows
    from oracle.unet import OracleUnet
    names = sys.argv[1:] or list(CASES)
    path = os.path.join(HERE, "config_evals.npz")
    d = dict(np.load(path)) if os.path.exists(path) else {}
    nets = {}
    for name in names:
        c = build(name)
        args, L = c["args"], c["L"]
        n, bpo, fs = args.network.cqt.num_octs, args.network.cqt.bins_per_oct, args.exp.sample_rate
        key = (n, fs, c["net_seed"])
        if key not in nets:
            shapes = [(k, tuple(v.shape)) for k, v in Unet_CQT_oct_with_attention(args, torch.device("meta")).state_dict().items()]
            sd = seeded_state_dict(shapes, c["net_seed"], gate_scale=10.0, affine_scale=10.0)
            nets.clear()                                    # one full-size oracle in memory at a time
            nets[key] = OracleUnet(n, bpo, OracleCQT(n, bpo, "oct", ("kaiser", 1), fs, L)).load_state_dict(sd)
        hann = int(args.tester.data_consistency.hann_size)
        osmp = OracleSampler(nets[key], OracleEDM(), T=int(args.tester.T), xi=0.25, hann_size=hann, audio_len=L)
        for b in c["items"]:
            mrow = c["mask"][b:b + 1] if c["mask"].shape[0] > 1 else c["mask"]
            osmp.mask, osmp.smask, osmp.y = mrow, smooth_mask_rows(mrow, hann), c["y"][b:b + 1]
            for k, (tk, xk) in enumerate(c["evals"]):
                t0 = time.time()
                osmp.trace = []
                osmp.get_score(xk[b:b + 1], tk)
                proj, strided = summarise(osmp.trace[0][0], 100 * k + b)
                d[f"{name}.b{b}.e{k}.proj"], d[f"{name}.b{b}.e{k}.s"] = proj, strided
                d[f"{name}.b{b}.e{k}.t"] = np.array(float(tk))
                print(f"{name} item {b} evaluation {k} (t = {float(tk):.4f}): |x_hat| = {np.sqrt(proj[-1]):.4f}   [{time.time() - t0:.0f} s]", flush=True)
                np.savez_compressed(path, **d)
    print("wrote", path, len(d), "arrays")


TRAJ_ITEM, TRAJ_STEPS, TRAJ_SEED0, TRAJ_FULL = 5, 8, 100, (3, 6)


def main_traj():
    """--traj: BASELINE.json configs[1], ONE item (index TRAJ_ITEM of the batch of 8, the second sub-batch stream), the first TRAJ_STEPS Heun
    steps (2 * TRAJ_STEPS guided evaluations) of the FREE-RUNNING oracle sampler from the real prior draw with the churn noise of the tester's
    schedule (per-item generator seeded TRAJ_SEED0 + item: prior, then one draw per churned step, edm_sampler_inpainting.py:201-251) ->
    tests/golden/config1_traj.npz: after every step the state x_{i+1} (8 projections + squared norm + every 97th sample), the projected x_hat of
    both evaluations of the step, and the FULL fp32 state entering steps TRAJ_FULL (restart points of the teacher-forced GPU test: states on a
    chaotic trajectory cannot be rebuilt from seeds).  ~12-20 min on 8 cores."""
    from audio_inpainting_diffusion_amd.init import seeded_state_dict
    from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
    from oracle.edm import OracleEDM
    from oracle.nsgt_cqt import OracleCQT
    from oracle.sampler import OracleSampler
    from oracle.unet import OracleUnet
    c = build("config1")
    args, L, b = c["args"], c["L"], TRAJ_ITEM
    n, bpo, fs = args.network.cqt.num_octs, args.network.cqt.bins_per_oct, args.exp.sample_rate
    shapes = [(k, tuple(v.shape)) for k, v in Unet_CQT_oct_with_attention(args, torch.device("meta")).state_dict().items()]
    sd = seeded_state_dict(shapes, c["net_seed"], gate_scale=10.0, affine_scale=10.0)
    net = OracleUnet(n, bpo, OracleCQT(n, bpo, "oct", ("kaiser", 1), fs, L)).load_state_dict(sd)
    hann = int(args.tester.data_consistency.hann_size)
    osmp = OracleSampler(net, OracleEDM(), T=int(args.tester.T), xi=0.25, hann_size=hann, audio_len=L)
    osmp.stop_after, osmp.states = TRAJ_STEPS, []
    t0 = time.time()
    orig = osmp.get_score

    def timed(x, t_i):
        r = orig(x, t_i)
        print(f"  evaluation {len(osmp.trace)} (t = {float(t_i):.4f}) done   [{time.time() - t0:.0f} s]", flush=True)
        return r
    osmp.get_score = timed
    mrow = c["mask"][b:b + 1] if c["mask"].shape[0] > 1 else c["mask"]
    osmp.predict_inpainting(c["y"][b:b + 1], mrow, seeds=[TRAJ_SEED0 + b], record=True)
    assert len(osmp.states) == TRAJ_STEPS and len(osmp.trace) == 2 * TRAJ_STEPS
    d = {"item": np.array(b), "steps": np.array(TRAJ_STEPS), "seed": np.array(TRAJ_SEED0 + b)}
    for i, x in enumerate(osmp.states):
        d[f"x{i + 1}.proj"], d[f"x{i + 1}.s"] = summarise(x[0], 300 + i)
        if i + 1 in TRAJ_FULL:
            d[f"x{i + 1}.full"] = x[0].numpy().astype(np.float32)
    for k, xh in enumerate(osmp.trace):
        d[f"xhat{k}.proj"], d[f"xhat{k}.s"] = summarise(xh[0], 400 + k)
    path = os.path.join(HERE, "config1_traj.npz")
    np.savez_compressed(path, **d)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    if "--traj" in sys.argv:
        main_traj()
    else:
        main()
