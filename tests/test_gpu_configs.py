"""GPU parity cases for the remaining BASELINE.json configurations:
configs[3] -- 16 kHz LibriSpeech-shape, batch 16, four short gaps, hann 100 (same network as cfg-A);
configs[4] -- 44.1 kHz 8-octave network, batch 4, 1.5 s gap.
The full-size B=1 guided evaluation of cfg-A is pinned to the oracle in test_gpu_vjp.py; here the large batches
are pinned to B=1 runs of the same items (batch independence at full size, 16 x the buffer offsets) and the
cfg-B guided evaluation (forward + input-VJP) to torch.autograd over the CPU oracle."""
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _coeffs(B, sigma=0.4):
    from oracle.edm import OracleEDM
    edm = OracleEDM()
    s = torch.full((B, 1), sigma)
    v = lambda t: t.reshape(-1).to(DEV).contiguous()
    return edm, s, (v(edm.cnoise(s)), v(edm.cin(s)), v(edm.cskip(s)), v(edm.cout(s)))


def test_config3_batch16_short_gaps_items_equal_their_b1_runs():
    from audio_inpainting_diffusion_amd.config import make_args
    from audio_inpainting_diffusion_amd.init import seeded_init_, seeded_normal
    from audio_inpainting_diffusion_amd.masks import mask_from_args
    from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
    args = make_args("librispeech16k", T=70, gap_ms=50.0)
    assert args.tester.data_consistency.hann_size == 100 and args.exp.sample_rate == 16000
    net = seeded_init_(Unet_CQT_oct_with_attention(args, torch.device(DEV)), 0, gate_scale=10.0, affine_scale=10.0)
    B, Ls = 16, args.exp.audio_len
    mask = torch.cat([mask_from_args(args, generator=torch.Generator().manual_seed(40 + b)) for b in range(B)])   # per-item gaps
    assert int((mask == 0).sum(1).max()) <= 4 * 800
    x = torch.stack([torch.from_numpy(seeded_normal(31, b, Ls)) for b in range(B)]) * 0.3
    y = torch.stack([torch.from_numpy(seeded_normal(32, b, Ls)) for b in range(B)]) * 0.063 * mask
    _, _, c16 = _coeffs(B)
    xh, g, nrm = net.denoise_guided(x.to(DEV), *c16, True, y.to(DEV), mask.to(DEV))
    xh, g, nrm = xh.cpu(), g.cpu(), nrm.cpu()
    assert torch.isfinite(xh).all() and torch.isfinite(g).all()
    _, _, c1 = _coeffs(1)
    for b in (0, 11, 15):
        xh1, g1, n1 = net.denoise_guided(x[b:b + 1].to(DEV), *c1, True, y[b:b + 1].to(DEV), mask[b:b + 1].to(DEV))
        e1, e2 = rel_l2(xh[b:b + 1], xh1.cpu()), rel_l2(g[b:b + 1], g1.cpu())
        print(f"config 3, item {b} of 16 vs its B=1 run: x_hat {e1:.2e}, rec_grads {e2:.2e}")
        assert e1 < 2e-6 and e2 < 5e-6 and abs(float(nrm[b]) - float(n1)) < 1e-5 * float(n1)


def test_config4_cfgB_batch4_guided_evaluation_vs_oracle_autograd():
    from audio_inpainting_diffusion_amd.config import make_args
    from audio_inpainting_diffusion_amd.init import seeded_init_, seeded_normal
    from audio_inpainting_diffusion_amd.masks import mask_from_args
    from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
    from oracle.nsgt_cqt import OracleCQT
    from oracle.unet import OracleUnet
    args = make_args("musicnet44k", T=128, gap_ms=1500.0)
    net = seeded_init_(Unet_CQT_oct_with_attention(args, torch.device(DEV)), 1, gate_scale=10.0, affine_scale=10.0)
    B, Ls = 4, args.exp.audio_len
    mask = mask_from_args(args)
    assert int((mask == 0).sum()) == 66150                       # 1.5 s at 44.1 kHz
    x = torch.stack([torch.from_numpy(seeded_normal(41, b, Ls)) for b in range(B)]) * 0.3
    y = torch.stack([torch.from_numpy(seeded_normal(42, b, Ls)) for b in range(B)]) * 0.063 * mask
    edm, s, c4 = _coeffs(B)
    xh, g, nrm = net.denoise_guided(x.to(DEV), *c4, True, y.to(DEV), mask.to(DEV))
    b = 2
    orc = OracleUnet(8, 64, OracleCQT(8, 64, "oct", ("kaiser", 1), 44100, Ls)).load_state_dict(net.state_dict())
    xr = x[b:b + 1].clone().requires_grad_()
    xh_ref = orc.CQTransform.apply_hpf_DC(edm.denoiser(xr, orc, s[:1]))
    norm = torch.linalg.norm(y[b:b + 1] - mask * xh_ref, dim=1, ord=2)
    g_ref = torch.autograd.grad(norm.sum(), xr)[0]
    e1, e2 = rel_l2(xh[b:b + 1].cpu(), xh_ref.detach()), rel_l2(g[b:b + 1].cpu(), g_ref)
    print(f"config 4 (cfg-B, B=4) item {b}: x_hat rel-L2 = {e1:.3e}, rec_grads rel-L2 = {e2:.3e}")
    assert e1 < 1e-4 and e2 < 1e-4 and abs(float(nrm[b].cpu()) - float(norm)) < 1e-4 * float(norm)


def test_config1_batch8_eight_free_running_heun_steps_vs_oracle_fixture():
    """SURVEY 8d parity gate (3) at the BENCHMARKED size: BASELINE.json configs[1] at full size and batch 8, guided branch, the
    first EIGHT Heun steps (16 guided evaluations) of the real loop -- prior draw, churn noise of the tester's schedule, Heun corrections
    (edm_sampler_inpainting.py:201-251) -- FREE-RUNNING on the GPU, item 5 (second sub-batch stream) against tests/golden/config1_traj.npz, which the
    CPU oracle sampler wrote in the build container for that item alone (make_config_golden.py --traj; per-item generator seeded 100 + item):
      * the state after every step and the projected x_hat of all 16 evaluations, free-running (errors chain through the trajectory): <= 1e-4;
      * teacher-forced: entering steps 3 and 6 the item's state is REPLACED by the oracle's own fp32 state (stored in full: a chaotic trajectory cannot
        be rebuilt from seeds), one Heun step is taken from it with the same churn draw, and both evaluations + the resulting state are compared.
    Supersedes round 5's two-step test that ran the oracle ON the GPU box (65 s of host time for two evaluations)."""
    import os
    import numpy as np
    from conftest import GOLDEN
    from config_cases import build, compare
    from audio_inpainting_diffusion_amd.edm import EDM
    from audio_inpainting_diffusion_amd.init import seeded_init_
    from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
    from audio_inpainting_diffusion_amd.sampler import Sampler
    z = np.load(os.path.join(GOLDEN, "config1_traj.npz"))
    item, steps, seed = int(z["item"]), int(z["steps"]), int(z["seed"])
    c = build("config1")
    args, B, Ls = c["args"], c["B"], c["L"]
    assert B == 8 and steps == 8 and seed == 100 + item
    net = seeded_init_(Unet_CQT_oct_with_attention(args, torch.device(DEV)), c["net_seed"], gate_scale=10.0, affine_scale=10.0)
    assert net._n_split(B) == 2                                  # the product schedule: two sub-batch streams
    smp = Sampler(model=net, diff_params=EDM(args), args=args)
    smp.seeds = [100 + b for b in range(B)]
    smp.setup_inpainting(c["y"].to(DEV), c["mask"])
    smp.trace = []
    state = smp.begin((B, Ls), torch.device(DEV))
    worst = {"free": 0.0, "forced": 0.0}

    def check(v, key, stream, bar, tag):
        ds, dp, dn = compare(v.cpu(), z[key + ".proj"], z[key + ".s"], stream)
        print(f"configs[1] B=8 item {item} {tag} {key}: strided rel-L2 {ds:.2e}, projections {dp:.2e}, squared norm {dn:.2e}")
        assert ds < bar and dp < 3 * bar and dn < 3 * bar, (key, tag, ds, dp, dn)
        return ds
    for i in range(steps):
        if f"x{i}.full" in z.files:                              # teacher-forced restart from the oracle's own state entering step i
            gst = [g.get_state() for g in smp._gens]
            forced = dict(state)
            forced["x"] = state["x"].clone()
            forced["x"][item] = torch.from_numpy(z[f"x{i}.full"]).to(DEV)
            ntr = len(smp.trace)
            smp.step(forced, i)
            worst["forced"] = max(worst["forced"], check(forced["x"][item], f"x{i + 1}", 300 + i, 2e-5, "teacher-forced"),
                                  check(smp.trace[ntr][item], f"xhat{2 * i}", 400 + 2 * i, 2e-5, "teacher-forced"),
                                  check(smp.trace[ntr + 1][item], f"xhat{2 * i + 1}", 400 + 2 * i + 1, 2e-5, "teacher-forced"))
            del smp.trace[ntr:]
            for g, st_ in zip(smp._gens, gst):                   # the free-running step below draws the SAME churn noise
                g.set_state(st_)
        smp.step(state, i)
        assert bool(torch.isfinite(state["x"]).all())
        worst["free"] = max(worst["free"], check(state["x"][item], f"x{i + 1}", 300 + i, 1e-4, "free-running"),
                            check(smp.trace[2 * i][item], f"xhat{2 * i}", 400 + 2 * i, 1e-4, "free-running"),
                            check(smp.trace[2 * i + 1][item], f"xhat{2 * i + 1}", 400 + 2 * i + 1, 1e-4, "free-running"))
    assert len(smp.trace) == 2 * steps
    print(f"configs[1] B=8, 8 Heun steps: worst free-running {worst['free']:.2e}, worst teacher-forced {worst['forced']:.2e}")


def test_cfgB_44k_8s_368368_forward_vs_oracle():
    """The 44.1 kHz network on its 8-second segments (conf/exp/musicnet44k_8s.yaml: L=368368, octave lengths 64..8192)."""
    from audio_inpainting_diffusion_amd.config import make_args
    from audio_inpainting_diffusion_amd.init import seeded_init_, seeded_normal
    from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
    from oracle.nsgt_cqt import OracleCQT
    from oracle.unet import OracleUnet
    Ls = 368368
    args = make_args("musicnet44k", audio_len=Ls, T=128, gap_ms=1500.0)
    net = seeded_init_(Unet_CQT_oct_with_attention(args, torch.device(DEV)), 1, gate_scale=10.0, affine_scale=10.0)
    x = torch.from_numpy(seeded_normal(12, 0, Ls)).reshape(1, Ls) * 0.5
    cn = torch.tensor([[-0.2]])
    with torch.no_grad():
        yv = net(x.to(DEV), cn.to(DEV)).cpu()
    orc = OracleUnet(8, 64, OracleCQT(8, 64, "oct", ("kaiser", 1), 44100, Ls)).load_state_dict(net.state_dict())
    with torch.no_grad():
        ref = orc(x, cn)
    e = rel_l2(yv, ref)
    print(f"cfg-B L=368368: rel-L2 vs oracle = {e:.3e}; GFLOP/eval = {net.flops_per_eval(1) / 1e9:.1f}")
    assert e < 1e-4
    assert abs(net.flops_per_eval(1) / 3.181e12 - 1) < 0.02      # 3.181 TFLOP


@pytest.mark.parametrize("case", ["config1", "config3_gap25", "config3_gap50", "config3_gap100", "config4"])
def test_config_evaluations_vs_oracle_fixture(case):
    """BASELINE.json configs[1] (all 8 items x the 4 evaluations of two Heun steps = 32 item-evaluations), configs[3] at 25 / 50 / 100 ms gaps
    (batch 16, hann 100, T = 70 schedule; item 11 against the ORACLE) and configs[4] (44.1 kHz 8-octave network, batch 4, T = 128; items 0 and 2),
    at full size and at the configuration's own batch: the projected x_hat of every evaluation -- fused guided evaluation, normalised guidance
    step, data-consistency projection, exactly the sampler's per-evaluation path -- against tests/golden/config_evals.npz, which the CPU oracle
    wrote in the build container (tests/golden/make_config_golden.py).  Inputs are rebuilt from seeds (tests/config_cases.py)."""
    import os
    import numpy as np
    from conftest import GOLDEN
    from config_cases import build, compare
    from audio_inpainting_diffusion_amd.edm import EDM
    from audio_inpainting_diffusion_amd.init import seeded_init_
    from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
    from audio_inpainting_diffusion_amd.sampler import Sampler
    z = np.load(os.path.join(GOLDEN, "config_evals.npz"))
    c = build(case)
    args, B = c["args"], c["B"]
    net = seeded_init_(Unet_CQT_oct_with_attention(args, torch.device(DEV)), c["net_seed"], gate_scale=10.0, affine_scale=10.0)
    smp = Sampler(model=net, diff_params=EDM(args), args=args)
    smp.setup_inpainting(c["y"].to(DEV), c["mask"])
    worst = [0.0, 0.0, 0.0]
    for k, (tk, xk) in enumerate(c["evals"]):
        smp.trace = []
        xd = xk.to(DEV)
        x_hat = smp._denoise(xd, tk)                       # fused guided evaluation + normalised guidance step
        smp._score_step(xd, x_hat, tk, 0.0, mode=0)        # projection stage (the step outputs are not used)
        xh = smp.trace[-1].cpu()
        assert bool(torch.isfinite(xh).all())
        for b in c["items"]:
            assert abs(float(z[f"{case}.b{b}.e{k}.t"]) - float(tk)) < 1e-6 * float(tk)
            e = compare(xh[b], z[f"{case}.b{b}.e{k}.proj"], z[f"{case}.b{b}.e{k}.s"], 100 * k + b)
            worst = [max(a, v) for a, v in zip(worst, e)]
            assert e[0] < 1e-4 and e[1] < 3e-4 and e[2] < 2e-4, (case, b, k, e)
    del smp, net
    print(f"{case}: {len(c['items']) * len(c['evals'])} item-evaluations at batch {B} vs the oracle fixture: worst strided rel-L2 {worst[0]:.2e}, "
          f"projections {worst[1]:.2e}, squared norm {worst[2]:.2e}")
