"""EDM inpainting sampler on MI355X -- drop-in for ``testing.edm_sampler_inpainting.Sampler``.

Call surface kept from the reference (testing/edm_sampler_inpainting.py): ``Sampler(model, diff_params, args,
rid=False)`` (:10), public attributes ``xi / nb_steps / order`` (:19-32), ``predict_inpainting(y_masked, mask)``
(:327), ``predict_spectrogram_inpainting(y_masked, mask[F,T])`` (:348), ``predict_unconditional(shape, device)`` (:155),
``update_diff_params`` (:43).  The loop (:178-262) stays
in Python but is sync-free: the schedule, gamma and every per-step scalar live on the host (float32 torch
arithmetic identical to the reference), noise is drawn from the CPU generator in the reference's order and
copied asynchronously, and each step costs two fused element-wise launches besides the denoiser evaluations.

Batch semantics (new; the reference only ever runs B=1 and its guided branch raises for B>1, :75-:78): every
reduction the reference takes over the whole batch -- the guidance norm (:75), ``normguide`` (:83) and the mask
row used for smoothing (:307) -- is taken PER ITEM, so item b of a batch reproduces the reference's B=1 run.
``seeds=[...]`` gives every item its own CPU generator (prior draw, then one draw per churned step), making
results independent of batch composition and of how segments are sharded over GPUs.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import numpy as np
import torch

from . import _lib
from .stft import SpectralMask


def prepare_smooth_mask(mask: torch.Tensor, size: int) -> torch.Tensor:
    """Hann cross-fade of the mask edges, row by row (vectorised form of :302-325; the reference walks all L
    samples of row 0 in a Python loop and re-uses the result for every item)."""
    hann = torch.hann_window(size * 2)
    left, right = hann[0:size], hann[size:]
    m = mask.detach().to("cpu", torch.float32)
    out = m.clone()
    for r in range(m.shape[0]):
        row = m[r].numpy()
        prev = np.concatenate(([1.0], row[:-1]))
        for i in np.nonzero(row != prev)[0]:
            if row[i] == 0:
                out[r, i - size:i] = right
            if row[i] == 1:
                out[r, i:i + size] = left
    return out


class LambdaDegradation:
    """A degradation given as a torch callable (the ``degradation`` lambda of ``Sampler.predict_resample``, :164-173; the reference
    differentiates ``norm(y - degradation(x_hat))`` through it with torch.autograd, :65-81).  Here the callable and its VJP at x_hat are the
    only torch-eager pieces of a guided evaluation: ``apply`` evaluates it on a leaf copy of x_hat, ``adjoint`` pulls the analytic seed
    d norm / d degradation(x_hat) (aid_guidance_seed) back through it -- for a linear operator that is its adjoint, for any other one exactly
    the Jacobian-transpose product the reference's autograd forms -- and the network's hand-written input-VJP takes over from there.
    The reference defines no projection for it (``proj_convex_set`` only exists after predict_inpainting / predict_spectrogram_inpainting)."""
    shared_mask = False                              # (network.denoise_guided: one stream, no graph replay -- the callable is opaque)

    def __init__(self, fn):
        self.fn = fn
        self._leaf = self._den = None

    def apply(self, x_hat):
        with torch.enable_grad():
            self._leaf = x_hat.detach().requires_grad_(True)
            self._den = self.fn(self._leaf)
        if not torch.is_tensor(self._den) or self._den.shape[0] != x_hat.shape[0]:
            raise _lib.AidError("predict_resample: degradation(x_hat) must return a tensor with the batch axis first")
        return self._den.detach().float().contiguous()

    def adjoint(self, g):
        if self._den is None:
            raise _lib.AidError("LambdaDegradation.adjoint before apply")
        den, leaf, self._den, self._leaf = self._den, self._leaf, None, None
        if not den.requires_grad:                    # a degradation that does not depend on x_hat: the reference's autograd.grad raises as well
            raise RuntimeError("degradation(x_hat) does not depend on x_hat: reconstruction guidance has no gradient")
        (gx,) = torch.autograd.grad(den, leaf, g.reshape(den.shape).to(den.dtype))
        return gx.float().contiguous()

    def project(self, x, y):
        raise AttributeError("proj_convex_set is undefined for predict_resample (the reference defines it in predict_inpainting / "
                             "predict_spectrogram_inpainting only): set tester.data_consistency.use = False")


class Sampler:
    def __init__(self, model, diff_params, args, rid=False):
        self.model = model
        self.diff_params = diff_params
        self.args = args
        if not self.args.tester.diff_params.same_as_training:
            self.update_diff_params()
        self.order = self.args.tester.order
        self.xi = self.args.tester.posterior_sampling.xi
        dc = self.args.tester.data_consistency
        self.data_consistency = dc.use and dc.type == "always"
        self.data_consistency_end = dc.use and dc.type == "end"
        if self.data_consistency or self.data_consistency_end:
            self.smooth = bool(dc.smooth)
        self.nb_steps = self.args.tester.T
        self.rid = rid                              # True: predict_* return the reference's 8-tuple of per-step buffers (:185-191, :260)
        self.spectral = None
        self.y = self.mask = self.smask = self.degradation = None      # installed by predict_* / setup_*
        self.seeds: Optional[List[int]] = None      # per-item RNG seeds; None -> global torch CPU generator
        self.trace = None                           # set to [] to record every projected x_hat (tests)
        self.trace_in = None                        # set to [] to record (x, t) handed to every denoiser evaluation (teacher-forced tests)
        self.n_evals = 0

    def update_diff_params(self):
        dp, tp = self.diff_params, self.args.tester.diff_params
        dp.sigma_min, dp.sigma_max, dp.ro, dp.sigma_data = tp.sigma_min, tp.sigma_max, tp.ro, tp.sigma_data
        dp.Schurn, dp.Stmin, dp.Stmax, dp.Snoise = tp.Schurn, tp.Stmin, tp.Stmax, tp.Snoise

    # ---------------------------------------------------------------------------------------------------
    def _check_model(self):
        """The loop drives the model through the fused entry points ``denoise`` / ``denoise_guided`` of the MI355X network (network.py).
        A foreign torch model has to be wrapped explicitly: ``Sampler(model=generic.GenericModelAdapter(model, diff_params), ...)``."""
        if not (hasattr(self.model, "denoise") and hasattr(self.model, "denoise_guided")):
            raise _lib.AidError("Sampler needs a model with the fused entry points denoise / denoise_guided (the MI355X network); wrap any other "
                                "torch model in audio_inpainting_diffusion_amd.generic.GenericModelAdapter -- there is no implicit eager path")

    def _draw(self, shape, scale):
        """scale * randn(shape) from the CPU generator(s) in the reference's order (:212, edm.py:94), drawn straight into one of two
        pinned staging buffers (re-used alternately; an event guards the one still in flight) -> host tensor ready for an async copy."""
        ring = self.__dict__.setdefault("_pin_ring", {"shape": None})
        if ring["shape"] != tuple(shape):
            ring.update(shape=tuple(shape), bufs=[torch.empty(shape, dtype=torch.float32).pin_memory() for _ in range(2)], ev=[None, None], k=0)
        k = ring["k"] = ring["k"] ^ 1
        if ring["ev"][k] is not None:
            ring["ev"][k].synchronize()                   # the copy issued two draws ago (long finished; never blocks in steady state)
        buf = ring["bufs"][k]
        if self.seeds is None:
            torch.randn(shape, out=buf)
        else:
            for b, g in enumerate(self._gens):
                torch.randn([1, shape[1]], generator=g, out=buf[b:b + 1])
        if float(scale) != 1.0:
            buf.mul_(scale)
        return buf, k

    def _draw_to(self, shape, scale, device):
        buf, k = self._draw(shape, scale)
        out = buf.to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._pin_ring["ev"][k] = ev
        return out

    # ---- one denoiser evaluation -> projected x_hat --------------------------------------------------------
    def _scalars(self, t_i):
        """c_noise, c_in, c_skip, c_out of this evaluation as host floats, computed by diff_params in float32 torch arithmetic (edm.py:97-128)"""
        dp = self.diff_params
        s = t_i.reshape(1)
        return float(dp.cnoise(s)), float(dp.cin(s)), float(dp.cskip(s)), float(dp.cout(s))

    def _denoise(self, x, t_i):
        """x_hat = D(x; t_i) [+ guidance step] ; data-consistency projection is applied by the caller's fused
        step kernel.  t_i: 0-d float32 CPU tensor."""
        B = x.shape[0]
        self.n_evals += B
        if self.trace_in is not None:
            self.trace_in.append((x.clone(), float(t_i)))
        if self.y is not None and self.xi > 0:
            return self._denoise_guided(x, t_i)
        hpf = bool(self.y is None and self.args.tester.filter_out_cqt_DC_Nyq)   # (:122-123) unconditional only
        return self.model.denoise(x, *self._scalars(t_i), hpf)

    def _denoise_guided(self, x, t_i):
        """Reconstruction guidance (:57-105), per item: analytic seed of the chosen norm, the network's hand-written input-VJP,
        then the normalised step x_hat - t*xi/(||g||/sqrt(L) + 1e-6) * g in one launch (aid_guidance_step)."""
        B, L = x.shape
        ps = self.args.tester.posterior_sampling
        hpf = bool(self.args.tester.filter_out_cqt_DC_Nyq)
        x_hat, rec_grads, _ = self.model.denoise_guided(x, *self._scalars(t_i), hpf, self.y, self.mask, self.spectral,
                                                        norm_type=ps.norm, beta=float(getattr(ps, "smoothl1_beta", 1.0)))
        out = torch.empty_like(x_hat)
        upd = torch.empty_like(x_hat) if self.rid else None
        inv = float(np.float32(1.0) / np.float32(self.args.exp.audio_len ** 0.5))
        p = _lib.GuidanceStepParams(x_hat.data_ptr(), rec_grads.data_ptr(), out.data_ptr(), _lib.ptr(upd), None, B, L,
                                    float(t_i) * self.xi, inv, 1e-6)                # (:83, :87, :97)
        _lib.call("aid_guidance_step", p)
        if self.rid:                                                                 # (:92-99: denoised estimate, s*rec_grads, updated estimate)
            self._rid_last = (x_hat, upd, out)
        return out

    def _score_step(self, x, x_hat, t_i, h, mode, x0=None, d0=None):
        """fused: projection (:343) + d = -t*score (:105,:230) + Euler proposal (:240) or Heun combine (:247)."""
        B, L = x.shape
        xnext = torch.empty_like(x)
        dout = torch.empty_like(x) if mode == 0 else None
        # guided branch: project only with data_consistency.type == "always" (:100); replacement branch (xi == 0):
        # at EVERY evaluation whatever the type (:141-147) -- like the reference, which fails there when no projection
        # was ever defined (data_consistency.use False, :338)
        proj = self.y is not None and (self.data_consistency or self.xi == 0)
        if proj and not (self.data_consistency or self.data_consistency_end):
            raise AttributeError("proj_convex_set is undefined: the replacement branch (xi = 0) needs tester.data_consistency.use")
        if proj and self.spectral is not None:                     # y + x_hat - A(x_hat)   (:360)
            x_hat, proj = self.spectral.project(x_hat, self.y), False
        xh_out = torch.empty_like(x) if (self.trace is not None or self.rid) else None
        p = _lib.ScoreStepParams(x.data_ptr(), x_hat.data_ptr(), _lib.ptr(self.y) if proj else None,
                                 _lib.ptr(self.smask) if proj else None, (self.smask.stride(0) if self.smask.shape[0] > 1 else 0) if proj else 0,
                                 _lib.ptr(x0), _lib.ptr(d0), None, None,
                                 xnext.data_ptr(), _lib.ptr(dout), _lib.ptr(xh_out), B, L, mode, float(t_i), float(h))
        _lib.call("aid_score_step", p)
        if self.trace is not None:
            self.trace.append(xh_out)
        self._rid_pocs = xh_out
        return xnext, dout

    def _project(self, x):
        """smask * y + (1 - smask) * x (:343) by the projection stage of aid_score_step (its xh_out output; the step outputs go to scratch)."""
        B, L = x.shape
        out, scratch = torch.empty_like(x), torch.empty_like(x)
        p = _lib.ScoreStepParams(x.data_ptr(), x.data_ptr(), self.y.data_ptr(), self.smask.data_ptr(),
                                 self.smask.stride(0) if self.smask.shape[0] > 1 else 0, None, None, None, None,
                                 scratch.data_ptr(), None, out.data_ptr(), B, L, 0, 1.0, 0.0)
        _lib.call("aid_score_step", p)
        return out

    # ---------------------------------------------------------------------------------------------------
    def predict_unconditional(self, shape, device):
        self.y = self.degradation = None
        self.mask = self.smask = self.spectral = None
        return self.predict(shape, device)

    def begin(self, shape, device):
        """Draw the prior and set up the schedule; returns the loop state consumed by ``step``."""
        device = torch.device(device)
        if device.type != "cuda":
            raise _lib.AidError("the MI355X sampler runs on the GPU only (no CPU fallback)")
        self._check_model()
        dp = self.diff_params
        self._gens = None if self.seeds is None else [torch.Generator().manual_seed(int(s)) for s in self.seeds]
        t = dp.create_schedule(self.nb_steps)                 # host, float32
        gamma = dp.get_gamma(t)
        x = self._draw_to(shape, t[0], device)                # prior (edm.py:94)
        return dict(x=x, t=t, gamma=gamma, shape=tuple(shape), device=device)

    def step(self, state, i: int):
        """One iteration of the sampling loop (:201-251): churn, denoiser evaluation, Heun correction."""
        dp = self.diff_params
        t, gamma, x, shape, device = state["t"], state["gamma"], state["x"], state["shape"], state["device"]
        B, L = shape
        if gamma[i] == 0:
            t_hat = t[i]
        else:
            t_hat = t[i] + gamma[i] * t[i]
            eps = self._draw_to(shape, dp.Snoise, device)
            coef = (t_hat ** 2 - t[i] ** 2) ** (1 / 2)
            xn = torch.empty_like(x)
            p = _lib.AxpbyParams(x.data_ptr(), eps.data_ptr(), xn.data_ptr(), None, None, B, L, 1.0, float(coef))
            _lib.call("aid_axpby", p)                      # x + sqrt(t_hat^2 - t_i^2) * eps   (:214)
            x = xn
        rid = state.get("rid")
        if rid is not None:
            rid["xt"][i] = x.cpu()
        x_hat = self._denoise(x, t_hat)
        h = t[i + 1] - t_hat
        x_prime, d = self._score_step(x, x_hat, t_hat, h, mode=0)
        if rid is not None:
            rid["denoised"][i], rid["grads"][i], rid["grad_update"][i] = (v.cpu() for v in self._rid_last)
            rid["pocs"][i] = self._rid_pocs.cpu()
        if t[i + 1] != 0 and self.order == 2:
            x_hat2 = self._denoise(x_prime, t[i + 1])
            x, _ = self._score_step(x_prime, x_hat2, t[i + 1], h, mode=1, x0=x, d0=d)
        else:
            x = x_prime
        if rid is not None:
            rid["xt2"][i] = x.cpu()
        state["x"] = x
        return state

    def predict(self, shape, device):
        state = self.begin(shape, device)
        if self.rid:
            if not (self.y is not None and self.xi > 0):
                raise _lib.AidError("rid=True needs the reconstruction-guidance branch (the reference unpacks its 5-tuple only there, :217)")
            state["rid"] = {k: torch.zeros((self.nb_steps,) + tuple(shape)) for k in ("denoised", "grads", "grad_update", "pocs", "xt", "xt2")}
        for i in range(self.nb_steps):
            self.step(state, i)
        x = state["x"]
        if self.data_consistency_end and self.y is not None:                     # (:252) one last projection of the final state
            x = self._project(x) if self.spectral is None else self.spectral.project(x, self.y)
        if self.rid:                                                             # (:260)
            r = state["rid"]
            return x.detach(), r["denoised"], r["grads"], r["grad_update"], r["pocs"], r["xt"], r["xt2"], state["t"]
        return x.detach()

    def setup_inpainting(self, y_masked, mask):
        """Install observations y_masked[B,L] and mask[1|B,L] (any dtype / device / strides) for the time-domain
        degradation; ``predict`` / ``begin`` + ``step`` then run the loop."""
        if mask.dim() == 1:
            mask = mask.reshape(1, -1)
        dev = y_masked.device
        self.y = y_masked.contiguous().float()
        self.mask = mask.to(dev, torch.float32).contiguous()         # read as a raw float* by aid_guidance_seed
        if self.mask.shape[-1] != self.y.shape[-1] or self.mask.shape[0] not in (1, self.y.shape[0]):
            raise ValueError(f"mask {tuple(mask.shape)} does not match observations {tuple(y_masked.shape)}")
        self.spectral = None
        self.smask = None
        self.degradation = self.apply_mask                            # (:335)
        if self.data_consistency or self.data_consistency_end:
            sm = prepare_smooth_mask(mask, self.args.tester.data_consistency.hann_size) if self.smooth else mask
            self.smask = sm.to(dev, torch.float32).contiguous()

    def predict_inpainting(self, y_masked, mask):
        """y_masked[B,L], mask[1|B,L] -> inpainted [B,L] on y_masked.device   (:327-346)"""
        self.setup_inpainting(y_masked, mask)
        return self.predict(self.y.shape, self.y.device)

    def setup_spectrogram_inpainting(self, y, mask, observed_is_clean=False):
        """Install the STFT-domain degradation (mask[F,T] or [B,F,T]).  observed_is_clean: ``y`` is the clean signal
        and the observation is A(y) (what the tester computes before calling predict_spectrogram_inpainting)."""
        st = self.args.tester.spectrogram_inpainting.stft
        y = y.contiguous().float()
        self.mask = self.smask = None
        self.spectral = SpectralMask(mask, y.shape[-1], st.n_fft, st.hop_length, st.win_length, st.window, y.device)
        self.y = self.spectral.apply(y) if observed_is_clean else y
        self.degradation = self.apply_spectral_mask                   # (:359)

    def predict_resample(self, y, shape, degradation):
        """y[B, ...] = observations, shape = (B, L) of the signal, degradation = torch callable x[B, L] -> y-shaped tensor   (:164-173): reconstruction
        guidance through an arbitrary degradation (the reference's generic entry point; its inpainting methods are special cases of it)."""
        if y.dim() == 3 and self.args.tester.posterior_sampling.norm != "smoothl1" and type(self.model).__name__ != "GenericModelAdapter":
            # (:67-70: torch.linalg.norm(y - den, dim=(1, 2), ord=norm) over 3-D observations is the INDUCED matrix norm -- ord = 2 the largest singular
            #  value, ord = 1 the largest column sum -- not the element-wise norm aid_guidance_seed computes over the flattened item)
            raise _lib.AidError("predict_resample: norm = 2 / 1 over 3-D observations is the induced matrix norm (spectral / max column sum) in the reference "
                                "(edm_sampler_inpainting.py:67-75); the HIP guidance seed implements the element-wise norms only: use norm = 'smoothl1', "
                                "flatten the observations to [B, N], or wrap the model in generic.GenericModelAdapter (torch autograd, reference semantics)")
        self.y = y.contiguous().float()
        self.mask = self.smask = None
        self.spectral = LambdaDegradation(degradation)
        self.degradation = degradation
        return self.predict(tuple(shape), self.y.device)

    # ---- the remaining public methods of the reference class (its testers never call them; kept so that code written against
    # ---- testing.edm_sampler_inpainting.Sampler finds every name).  They run outside the sampling loop. ------------------------------
    def apply_mask(self, x, mask=None):
        """mask * x   (:264-269)"""
        m = self.mask if mask is None else mask
        return m.to(x.device, x.dtype) * x

    def apply_spectral_mask(self, x):
        """STFT -> mask -> inverse STFT of the installed spectrogram mask (:271-292), on the HIP STFT kernels (stft.SpectralMask)."""
        if not isinstance(self.spectral, SpectralMask):
            raise AttributeError("apply_spectral_mask: no spectrogram mask installed (predict_spectrogram_inpainting / setup_spectrogram_inpainting)")
        return self.spectral.apply(x.contiguous().float())

    def prepare_smooth_mask(self, mask, size=10):
        """(:302-325) as a method, like the reference's"""
        return prepare_smooth_mask(mask if mask.dim() > 1 else mask.reshape(1, -1), size)

    def proj_convex_set(self, x):
        """The data-consistency projection installed by predict_inpainting (:343) / predict_spectrogram_inpainting (:360); AttributeError when
        none was (the reference's attribute does not exist then)."""
        if self.y is None or not (self.data_consistency or self.data_consistency_end):
            raise AttributeError("proj_convex_set is undefined: needs observations and tester.data_consistency.use")
        x = x.detach().contiguous().float()
        return self._project(x) if self.spectral is None else self.spectral.project(x, self.y)

    def get_score(self, x, y, t_i, degradation=None):
        """ONE evaluation of the score (x_hat - x) / t_i^2 exactly as the loop forms it (:115-153): unconditional for y None (DC/Nyquist projector
        only), reconstruction guidance for xi > 0, replacement + projection for xi == 0.  ``y`` / ``degradation`` other than the installed ones are
        installed for this call (a callable degradation goes through LambdaDegradation).  With rid=True the guided branch returns the reference's
        5-tuple (score, denoised estimate, s * rec_grads, estimate after the guidance step, estimate after the projection; :106-108)."""
        t = torch.as_tensor(t_i, dtype=torch.float32).detach().reshape(()).cpu()
        x = x.detach().contiguous().float()
        B, L = x.shape
        saved = (self.y, self.mask, self.smask, self.spectral, self.degradation)
        try:
            if y is None:
                if degradation is not None:
                    raise AssertionError("unconditional sampling takes no degradation (:117)")
                self.y = None
            else:
                if y is not saved[0]:
                    self.y = y.contiguous().float()
                if callable(degradation) and degradation != saved[4] and degradation is not saved[3]:      # (bound methods compare by ==)
                    self.mask = self.smask = None
                    self.spectral, self.degradation = LambdaDegradation(degradation), degradation
            x_hat = self._denoise(x, t)
            keep, self.rid = self.rid, bool(self.rid and self.y is not None and self.xi > 0)
            try:
                _, d = self._score_step(x, x_hat, t, 0.0, mode=0)          # projection by the loop's rule (raises like the reference when undefined), d = -t * score
            finally:
                self.rid = keep
            score = torch.empty_like(x)
            _lib.call("aid_axpby", _lib.AxpbyParams(d.data_ptr(), d.data_ptr(), score.data_ptr(), None, None, B, L, -1.0 / float(t), 0.0))
            if self.rid and self.y is not None and self.xi > 0:
                return (score,) + tuple(self._rid_last) + (self._rid_pocs,)
            return score
        finally:
            self.y, self.mask, self.smask, self.spectral, self.degradation = saved

    def get_score_rec_guidance(self, x, y, t_i, degradation=None):
        """(:57-113) the guided evaluation on its own; needs xi > 0 like the branch that calls it in the reference (:129-131)."""
        if y is None or not self.xi > 0:
            raise _lib.AidError("get_score_rec_guidance needs observations and xi > 0")
        return self.get_score(x, y, t_i, degradation)

    def predict_spectrogram_inpainting(self, y_masked, mask):
        """y_masked[B,L], mask[F,T] (or [B,F,T]) over the STFT of tester.spectrogram_inpainting.stft -> [B,L]
        (:348-364): degradation = STFT-domain masking, projection y + x - A(x)."""
        self.setup_spectrogram_inpainting(y_masked, mask)
        return self.predict(self.y.shape, self.y.device)
