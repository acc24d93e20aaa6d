"""Training step on the MI355X kernels.

Mirrors what the reference does per iteration (training/trainer.py:253-304 ``train_step`` / ``update_ema`` with
diff_params/edm.py:150-193 ``prepare_train_preconditioning`` / ``loss_fn``):

    sigma ~ sample_ptrain_safe(B) ; noise = randn * sigma
    input = c_in (x + noise) ; target = (x - c_skip (x + noise)) / c_out ; error = net(input, c_noise) - target
    loss = mean(error**2) ; loss.backward() ; lr ramp-up ; clip_grad_norm_ ; Adam.step() ; EMA update

Host side (this file): the O(B) scalar work and the RNG, in the reference's float32 torch arithmetic.  Device side: the
network forward, the activation-gradient plan the guidance branch already uses, and the parameter-gradient / optimiser
kernels of csrc/aid_train.hip -- no torch.autograd, no torch.optim on the path.  Parameters, gradients, Adam moments and the
EMA copy are flat fp32 buffers (dist.flatten_parameters_), so optimiser, EMA and gradient clipping are three launches.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import _lib
from .dist import flatten_parameters_


def sample_ptrain_safe(edm, N: int, generator=None) -> torch.Tensor:
    """EDM.sample_ptrain_safe (edm.py:76-85): sigma on the sampling schedule's warp with exponent ro_train."""
    a = torch.rand(N, generator=generator)
    return (edm.sigma_max ** (1 / edm.ro_train) + a * (edm.sigma_min ** (1 / edm.ro_train) - edm.sigma_max ** (1 / edm.ro_train))) ** edm.ro_train


def prepare_train_preconditioning(edm, x: torch.Tensor, sigma: torch.Tensor, noise: Optional[torch.Tensor] = None):
    """edm.py:150-163: (c_in (x+n), target, c_noise); sigma [B,1] on x.device; noise defaults to randn * sigma (CPU RNG like the reference)."""
    if noise is None:
        noise = torch.randn(x.shape).to(x.device) * sigma
    cskip, cout, cin, cnoise = edm.cskip(sigma), edm.cout(sigma), edm.cin(sigma), edm.cnoise(sigma)
    target = (1 / cout) * (x - cskip * (x + noise))
    return cin * (x + noise), target, cnoise


def a_weighting_taps(fs: float, ntaps: int = 101):
    """The A-weighting FIR of the reference's perceptual error filter (utils/training_utils.py:94-120, after auraloss): analog IEC/CD 1672 filter ->
    bilinear transform -> 512-point magnitude response -> least-squares FIR fit; float32 taps."""
    import numpy as np
    import scipy.signal
    if ntaps % 2 == 0:
        raise ValueError(f"ntaps must be odd (ntaps={ntaps}).")
    f1, f2, f3, f4, A1000 = 20.598997, 107.65265, 737.86223, 12194.217, 1.9997
    nums = [(2 * np.pi * f4) ** 2 * (10 ** (A1000 / 20)), 0, 0, 0, 0]
    dens = np.polymul([1, 4 * np.pi * f4, (2 * np.pi * f4) ** 2], [1, 4 * np.pi * f1, (2 * np.pi * f1) ** 2])
    dens = np.polymul(np.polymul(dens, [1, 2 * np.pi * f3]), [1, 2 * np.pi * f2])
    b, a = scipy.signal.bilinear(nums, dens, fs=fs)
    w_iir, h_iir = scipy.signal.freqz(b, a, worN=512, fs=fs)
    return scipy.signal.firls(ntaps, w_iir, abs(h_iir), fs=fs).astype("float32")


class FirFilter:
    """y = conv1d(x, taps, padding=ntaps//2) on [B, L] GPU rows (FIRFilter.forward, utils/training_utils.py:122-137) and its adjoint, both through
    aid_resample_poly at ratio 1:1 (one phase of 2*width + 1 taps)."""

    def __init__(self, taps, device):
        t = torch.as_tensor(taps, dtype=torch.float32).reshape(-1)
        if t.numel() % 2 == 0:
            raise ValueError("odd number of taps expected")
        self.width = t.numel() // 2
        self.k = t.to(device).contiguous()
        self.kT = t.flip(0).to(device).contiguous()

    def _run(self, x, k):
        x = x.contiguous().float()
        B, L = x.shape
        y = torch.empty_like(x)
        p = _lib.ResamplePolyParams(x.data_ptr(), y.data_ptr(), k.data_ptr(), x.stride(0), y.stride(0), L, L, B, 1, 1, self.width, k.numel())
        _lib.call("aid_resample_poly", p)
        return y

    def apply(self, x):
        return self._run(x, self.k)

    def adjoint(self, g):
        return self._run(g, self.kT)


class Trainer:
    """The per-iteration part of the reference ``Trainer`` (optimizer conf/exp/*.yaml:12-19,62-69)."""

    def __init__(self, net, edm, lr=2e-4, beta1=0.9, beta2=0.999, eps=1e-8, lr_rampup_it=10000, use_grad_clip=True, max_grad_norm=1.0,
                 ema_rate=0.9999, ema_rampup=10000, batch=4, use_cqt_DC_correction=False, aweighting_taps=None):
        self.net, self.edm = net, edm
        self.lr, self.beta1, self.beta2, self.eps = float(lr), float(beta1), float(beta2), float(eps)
        self.lr_rampup_it, self.use_grad_clip, self.max_grad_norm = lr_rampup_it, bool(use_grad_clip), float(max_grad_norm)
        self.ema_rate, self.ema_rampup, self.batch = float(ema_rate), ema_rampup, int(batch)
        self.hpf_error = bool(use_cqt_DC_correction)
        # diff_params.aweighting.use_aweighting (edm.py:33-34, :189-190): pass a_weighting_taps(fs, ntaps); None = off (the shipped default)
        self.fir = None if aweighting_taps is None else FirFilter(aweighting_taps, next(net.parameters()).device)
        self.flat = flatten_parameters_(net)
        self.m, self.v = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        self.ema = self.flat.clone()                                   # EMA of every parameter, same flat layout (trainer.py:66-68 deepcopy)
        self.ws = torch.zeros(_lib.AID_SUMSQ_BLOCKS, device=self.flat.device, dtype=torch.float64)
        self.gstat = torch.zeros(2, device=self.flat.device, dtype=torch.float32)      # (gradient norm, clipping coefficient)
        self.it = 0                                                     # iterations done; the t of Adam's bias correction is it + 1
        self.steps = 0

    # -----------------------------------------------------------------------------------------------------------------------
    def loss_and_grads(self, audio: torch.Tensor, sigma: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None, accumulate: bool = False):
        """EDM.loss_fn (edm.py:166-193) + loss.backward(): returns (loss, error**2, sigma); gradients land in (accumulate: are added to) the
        state's flat buffer."""
        B = audio.shape[0]
        if sigma is None:
            sigma = sample_ptrain_safe(self.edm, B)
        sigma = sigma.reshape(B, 1).to(audio.device)
        inp, target, cnoise = prepare_train_preconditioning(self.edm, audio, sigma, noise)
        loss, err2 = self.net.loss_and_grads(inp, cnoise, target, hpf_error=self.hpf_error, fir=self.fir, accumulate=accumulate)
        return loss, err2, sigma

    def grads(self, B: int) -> torch.Tensor:
        return self.net.train_state(B)["gflat"]

    def _rehome(self):
        """The optimiser walks the flat parameter buffer.  If the parameters were re-homed since the constructor (``load_state_dict(assign=True)``,
        ``.to()``, a dtype change), the plan builder has made a NEW flat buffer and gradients follow that one: adopt it, so that Adam / EMA keep
        updating the live weights (moments and EMA keep their values when the layout is unchanged, otherwise the mismatch is an error)."""
        live = flatten_parameters_(self.net)
        if live.data_ptr() != self.flat.data_ptr():
            if live.numel() != self.flat.numel() or live.device != self.flat.device:
                raise _lib.AidError("Trainer: the network's flat parameter buffer changed size or device after the Trainer was built")
            self.flat = live

    def optimizer_step(self, B: int):
        """lr ramp-up (trainer.py:270-274), clip_grad_norm_ (:277-278), Adam.step (:281)."""
        self._rehome()
        g = self.grads(B)
        n = g.numel()
        if n != self.flat.numel():
            raise _lib.AidError("Trainer: gradient buffer and parameter buffer have different layouts")
        lr = self.lr * min(self.it / max(self.lr_rampup_it, 1e-8), 1) if self.it <= self.lr_rampup_it else self.lr
        _lib.call("aid_sumsq", _lib.SumsqParams(g.data_ptr(), self.ws.data_ptr(), self.gstat.data_ptr(), n,
                                                self.max_grad_norm if self.use_grad_clip else 0.0))
        self.steps += 1
        t = self.steps
        _lib.call("aid_adam", _lib.AdamParams(self.flat.data_ptr(), g.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                              self.gstat[1:].data_ptr(), n, lr, self.beta1, self.beta2, self.eps,
                                              1.0 - self.beta1 ** t, math.sqrt(1.0 - self.beta2 ** t)))
        self.net._packed_ver = None          # parameters changed behind torch's version counters: refresh the kernel-side packs

    def update_ema(self, batch: Optional[int] = None):
        """trainer.py:288-304 (t = it * args.exp.batch: the constructor's ``batch``; ``batch=`` overrides it for one call)"""
        t = self.it * (self.batch if batch is None else int(batch))
        rate = float(min(max(t / self.ema_rampup, 0.0), self.ema_rate)) if t < self.ema_rampup else self.ema_rate
        _lib.call("aid_ema", _lib.EmaParams(self.ema.data_ptr(), self.flat.data_ptr(), self.flat.numel(), rate))

    def train_step(self, audio, sigma=None, noise=None):
        """One iteration: returns the loss (device scalar) of the last accumulation round.  ``audio`` is one batch [B, L], or a list of batches =
        the reference's ``num_accumulation_rounds`` (trainer.py:259-266: one loss.backward() per round into the same gradients, unnormalised);
        ``sigma`` / ``noise`` then are lists too (or None)."""
        rounds = list(audio) if isinstance(audio, (list, tuple)) else [audio]
        sg = list(sigma) if isinstance(sigma, (list, tuple)) else [sigma] * len(rounds)
        nz = list(noise) if isinstance(noise, (list, tuple)) else [noise] * len(rounds)
        if len({a.shape[0] for a in rounds}) != 1:
            raise ValueError("accumulation rounds must share one batch size (they share one launch-plan state)")
        for r, a in enumerate(rounds):
            loss, _, _ = self.loss_and_grads(a, sg[r], nz[r], accumulate=r > 0)
        audio = rounds[0]
        self.optimizer_step(audio.shape[0])
        self.update_ema()                     # (training_loop: train_step, update_ema, then it += 1; trainer.py:366-368.  The EMA ramp counts
                                              #  it * args.exp.batch like the reference -- the CONFIGURED batch, whatever this call was fed;
                                              #  update_ema(batch=...) overrides it explicitly)
        self.it += 1
        return loss

    def _layout(self):
        """(name, offset, shape) of every fp32 parameter, then every fp32 buffer, in the flat buffers' order (dist.flatten_parameters_)"""
        items = [(k, v) for k, v in self.net.named_parameters() if v.dtype == torch.float32] + \
                [(k, v) for k, v in self.net.named_buffers() if v.dtype == torch.float32]
        out, off = [], 0
        for k, v in items:
            out.append((k, off, tuple(v.shape)))
            off += v.numel()
        assert off == self.flat.numel()
        return out

    def _views(self, flat):
        return {k: flat[off:off + int(torch.Size(shape).numel())].view(shape) for k, off, shape in self._layout()}

    def ema_state_dict(self):
        """state_dict-shaped view of the EMA buffer (what the reference checkpoints as 'ema'), in the module's own key order."""
        v = self._views(self.ema)
        return {k: v[k] for k in self.net.state_dict() if k in v}

    # ---- checkpoints in the reference's shape (training/trainer.py:186-199: {'it', 'network', 'optimizer', 'ema', 'args'}) -----------------
    def state_dict(self, args=None):
        """'optimizer' is what ``torch.optim.Adam(network.parameters(), ...).state_dict()`` of the reference (utils/setup.py:55-58) would hold:
        per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` indexed by position in ``network.parameters()`` (trainable parameters only), and
        one param group -- so a run on this Trainer resumes under the reference's trainer and vice versa."""
        m, v = self._views(self.m), self._views(self.v)
        params = list(self.net.named_parameters())
        state = {}
        if self.steps > 0:
            for i, (k, p) in enumerate(params):
                if p.requires_grad:
                    state[i] = {"step": torch.tensor(float(self.steps)), "exp_avg": m[k].clone(), "exp_avg_sq": v[k].clone()}
        group = {"lr": self.lr, "betas": (self.beta1, self.beta2), "eps": self.eps, "weight_decay": 0, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(params)))}
        return {"it": self.it, "network": {k: t.detach().clone() for k, t in self.net.state_dict().items()},
                "optimizer": {"state": state, "param_groups": [group]},
                "ema": {k: t.clone() for k, t in self.ema_state_dict().items()}, "args": args}

    @torch.no_grad()
    def load_state_dict(self, sd, strict: bool = True):
        """Restore a checkpoint of the reference's shape: weights and EMA in place (the flat views stay intact), Adam moments and step count
        from 'optimizer' (so bias correction and the lr ramp-up continue instead of restarting), ``it``."""
        self._rehome()
        self.net.load_state_dict(sd["network"], strict=strict)
        self.net._packed_ver = None
        ema = self._views(self.ema)
        for k, t in sd.get("ema", {}).items():
            if k in ema:
                ema[k].copy_(t.to(ema[k].device, torch.float32))
            elif strict:
                raise KeyError(f"unexpected key {k!r} in checkpoint['ema']")
        opt = sd.get("optimizer")
        if opt is not None:
            m, v = self._views(self.m), self._views(self.v)
            self.m.zero_()
            self.v.zero_()
            params = list(self.net.named_parameters())
            g = (opt.get("param_groups") or [{}])[0]
            if "params" in g and len(g["params"]) != len(params):
                raise _lib.AidError(f"checkpoint['optimizer'] covers {len(g['params'])} parameters, this network has {len(params)}: "
                                    "it was written for a different network configuration")
            steps = 0
            for i, st in opt.get("state", {}).items():
                if not 0 <= int(i) < len(params):
                    raise _lib.AidError(f"checkpoint['optimizer'] has state for parameter #{i}; this network has {len(params)} parameters")
                k = params[int(i)][0]
                for nm in ("exp_avg", "exp_avg_sq"):
                    if tuple(st[nm].shape) != tuple(m[k].shape):
                        raise _lib.AidError(f"checkpoint['optimizer'] {nm} of parameter #{i} ({k}) has shape {tuple(st[nm].shape)}, expected {tuple(m[k].shape)}")
                m[k].copy_(st["exp_avg"].to(m[k].device, torch.float32))
                v[k].copy_(st["exp_avg_sq"].to(v[k].device, torch.float32))
                steps = max(steps, int(round(float(st["step"]))))
            self.steps = steps
            # (the group's 'lr' is the RAMPED value of the last iteration -- trainer.py:270-274 overwrite it every step from args.exp.lr -- so the
            #  base learning rate stays the constructor's, as it stays args.exp.lr in the reference)
            if "betas" in g:
                self.beta1, self.beta2 = float(g["betas"][0]), float(g["betas"][1])
            self.eps = float(g.get("eps", self.eps))
        self.it = int(sd.get("it", self.it))
        return True

    def save_checkpoint(self, path: str, args=None):
        torch.save(self.state_dict(args), path)
