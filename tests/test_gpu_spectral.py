"""GPU parity of the spectrogram-inpainting path (reference
testing/edm_sampler_inpainting.py:271-290, :348-364): the STFT-mask operator kernels, their adjoint, the
projection, and the sampler driving the MI355X network, against the CPU oracle and the reference golden."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mask(L, n_fft, hop, seed=0, per_item=0):
    Lp = L + (n_fft - L % n_fft)
    shape = (n_fft // 2 + 1, 1 + Lp // hop) if not per_item else (per_item, n_fft // 2 + 1, 1 + Lp // hop)
    return (torch.rand(shape, generator=torch.Generator().manual_seed(seed)) > 0.3).float()


@pytest.mark.parametrize("case", [(2, 1000, 64, 16, 64), (1, 1024, 64, 16, 64), (3, 777, 128, 32, 128), (2, 1500, 64, 16, 32),
                                  (2, 184184, 1024, 256, 1024), (2, 65536, 2048, 512, 2048)])
def test_stft_mask_operator_adjoint_projection_vs_oracle(case):
    from audio_inpainting_diffusion_amd.stft import SpectralMask
    from oracle.sampler import spectral_mask_apply
    B, L, n_fft, hop, win = case
    mask = _mask(L, n_fft, hop)
    g0 = torch.Generator().manual_seed(1)
    x, g, y = (torch.randn(B, L, generator=g0) for _ in range(3))
    op = SpectralMask(mask, L, n_fft, hop, win, device=DEV)
    xt = x.double().requires_grad_()
    ref = spectral_mask_apply(xt, mask.double(), n_fft, hop, win)
    (ref * g.double()).sum().backward()
    got, adj = op.apply(x.to(DEV)).cpu(), op.adjoint(g.to(DEV)).cpu()
    prj = op.project(x.to(DEV), y.to(DEV)).cpu()
    e = (rel_l2(got, ref.detach()), rel_l2(adj, xt.grad), rel_l2(prj, y.double() + x.double() - ref.detach()))
    print(case, "apply %.2e adjoint %.2e project %.2e" % e)
    assert max(e) < 2e-6
    # <A x, g> == <x, A^T g> on the device results themselves
    lhs, rhs = float((got.double() * g.double()).sum()), float((x.double() * adj.double()).sum())
    assert abs(lhs - rhs) < 1e-5 * max(abs(lhs), 1.0)


def test_stft_mask_per_item_masks_and_reference_golden():
    from audio_inpainting_diffusion_amd.stft import SpectralMask
    from oracle.sampler import spectral_mask_apply
    z = np.load(os.path.join(GOLDEN, "sampler_spectral.npz"))
    L = int(z["L"])
    n_fft, hop, win = (int(v) for v in z["stft"])
    op = SpectralMask(torch.from_numpy(z["mask"]), L, n_fft, hop, win, device=DEV)
    for tag in ("sg_s0", "sg_s1"):            # the reference's own apply_spectral_mask output
        assert rel_l2(op.apply(torch.from_numpy(z[tag + ".y0"]).to(DEV)).cpu(), z[tag + ".y"]) < 2e-6
    m = _mask(L, n_fft, hop, seed=3, per_item=2)
    x = torch.randn(2, L, generator=torch.Generator().manual_seed(2))
    got = SpectralMask(m, L, n_fft, hop, win, device=DEV).apply(x.to(DEV)).cpu()
    for b in range(2):
        assert rel_l2(got[b:b + 1], spectral_mask_apply(x[b:b + 1], m[b], n_fft, hop, win)) < 2e-6


@pytest.mark.parametrize("xi", [0.25, 0.0])
def test_spectrogram_inpainting_sampler_vs_oracle(xi):
    """Sampler.predict_spectrogram_inpainting on the MI355X network, per evaluation, against the oracle sampler
    (teacher-forced comparison of every projected x_hat; B=2 with per-item seeds)."""
    import test_gpu_vjp as V
    from audio_inpainting_diffusion_amd.edm import EDM
    from audio_inpainting_diffusion_amd.sampler import Sampler
    from oracle.edm import OracleEDM
    from oracle.sampler import OracleSampler, spectral_mask_apply
    net, orc, z, kw, args = V._setup("a")
    Ls = kw["audio_len"]
    n_fft, hop = 256, 64
    args.tester.T, args.tester.posterior_sampling.xi = 3, xi
    st = args.tester.spectrogram_inpainting.stft
    st.n_fft, st.hop_length, st.win_length = n_fft, hop, n_fft
    Lp = Ls + (n_fft - Ls % n_fft)
    mask = torch.ones(n_fft // 2 + 1, 1 + Lp // hop)
    mask[10:60, 20:45] = 0
    y = spectral_mask_apply(torch.from_numpy(z["x"]) * 0.126, mask, n_fft, hop, n_fft)
    smp = Sampler(model=net, diff_params=EDM(args), args=args)
    smp.seeds, smp.trace = [5, 6], []
    out = smp.predict_spectrogram_inpainting(y.to(DEV), mask.to(DEV))
    osmp = OracleSampler(orc, OracleEDM(), T=3, xi=xi, audio_len=Ls)
    ref = osmp.predict_spectrogram_inpainting(y, mask, stft=(n_fft, hop, n_fft), seeds=[5, 6], record=True)
    errs = [rel_l2(a.cpu(), b) for a, b in zip(smp.trace, osmp.trace)]
    print("spectrogram inpainting xi=%g per-evaluation x_hat rel-L2:" % xi, ["%.2e" % e for e in errs],
          " final: %.2e" % rel_l2(out.cpu(), ref))
    assert len(errs) == 5 and errs[0] < 1e-4 and max(errs) < 5e-4


def test_spectrogram_guided_evaluation_on_sub_batch_streams_is_bit_identical():
    """A shared STFT mask lets the guided evaluation run as sub-batches on concurrent streams (own frame scratch per stream)."""
    import test_gpu_vjp as V
    from audio_inpainting_diffusion_amd.stft import SpectralMask
    from oracle.edm import OracleEDM
    net, orc, z, kw, args = V._setup("a")
    Ls = kw["audio_len"]
    n_fft, hop = 256, 64
    B = 6
    g0 = torch.Generator().manual_seed(4)
    x = (torch.randn(B, Ls, generator=g0) * 0.4).to(DEV)
    op = SpectralMask(_mask(Ls, n_fft, hop, seed=5), Ls, n_fft, hop, n_fft, device=DEV)
    y = op.apply((torch.randn(B, Ls, generator=g0) * 0.063).to(DEV))
    edm = OracleEDM()
    s = torch.rand(B, 1, generator=g0) * 0.8 + 0.05
    v = lambda t: t.reshape(-1).to(DEV).contiguous()
    co = (v(edm.cnoise(s)), v(edm.cin(s)), v(edm.cskip(s)), v(edm.cout(s)))
    res = {}
    for n in (1, 3):
        net.split_streams = n
        for _ in range(2):
            res[n] = net.denoise_guided(x, *co, True, y, None, op)
    torch.cuda.synchronize()
    for a_, b_ in zip(res[1], res[3]):
        assert torch.equal(a_, b_)
    net.split_streams = None
