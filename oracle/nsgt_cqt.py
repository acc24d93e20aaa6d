"""CPU oracle: octave-mode non-stationary Gabor constant-Q transform (NSGT-CQT).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  **PARITY UNPINNED**: the reference calls the
external package ``cqt_nsgt_pytorch`` (``CQT_nsgt(numocts, binsoct, mode="oct", window=("kaiser",beta),
fs, audio_len, dtype, device)``, reference networks/unet_cqt_oct_with_projattention_adaLN_2.py:620;
``.fwd`` :743, ``.bwd`` :841, ``.apply_hpf_DC`` testing/edm_sampler_inpainting.py:63,123) whose source
is not under /root/reference and not installed.  What the call sites fix, and what this file honours:

* ``fwd(x[B,1,L])`` -> python list, index 0 = lowest octave, each ``[B,1,bins_per_oct,T_k]`` complex64,
  ``T_k`` powers of two halving per octave (unet...py:750,768-774,786,822,830);
* ``bwd(list)`` -> ``[B,1,L]`` real (cropped at unet...py:843);
* "oct" mode drops the DC and Nyquist bands; ``apply_hpf_DC`` is the matching projector, so
  ``bwd(fwd(x)) == apply_hpf_DC(x)``.

Algorithm (published NSGT, painless case; written band-by-band on purpose, the product builds
vectorised tables independently in ``audio_inpainting_diffusion_amd/cqt.py``):

  centre frequencies  f_k = fmin * 2**(k/bpo), fmin = (fs/2)/2**numocts, k = 0..K-1   (K = numocts*bpo)
  in DFT bins         b_k = f_k * L / fs ;  band list = [DC, b_0..b_{K-1}, Nyquist]
  window lengths      DC: round(2 b_0); band 0: round(b_0 * (2**(1/bpo) - 2**(-1/bpo)));
                      band k>=1: round(b_{k+1} - b_{k-1}) (b_K := L/2); Nyquist: round(2 (L/2 - b_{K-1}));
                      all clipped to >= 4
  windows             Kaiser(beta) sampled symmetrically about the rounded centre bin
  octave lengths      T_o = nextpow2(max window length in octave o); every band of octave o is
                      sampled with M_k = T_o time points
  analysis            c_k = IFFT_{M_k}( wrap_{M_k}( X[r_k + j] * g_k[j] ) ),  X = FFT_L(x)
  dual frame          gd_k = g_k / S,  S[v] = sum over ALL bands (incl. DC, Nyquist, mirrored negative
                      frequencies) of M_k * g_k[v - r_k]**2
  synthesis           Y[r_k + j] += FFT_{M_k}(c_k)[j mod M_k] * M_k * gd_k[j] ; y = irfft(Y[:L/2+1])
"""
from __future__ import annotations

import math
from typing import List, Sequence

import numpy as np
import torch


def _next_pow2(n: int) -> int:
    return 1 << (int(n) - 1).bit_length()


def _kaiser_centered(M: int, beta: float, half_sample_odd: bool = False) -> np.ndarray:
    """Kaiser window sampled at integer offsets j = -floor(M/2) .. ceil(M/2)-1 about the centre
    (half_sample_odd: odd lengths are sampled at j - 1/2, i.e. a periodic window rolled by M//2)."""
    j = np.arange(-(M // 2), M - M // 2, dtype=np.float64)
    if half_sample_odd and M % 2:
        j = j - 0.5
    r = 2.0 * j / M
    return np.i0(beta * np.sqrt(np.clip(1.0 - r * r, 0.0, None))) / np.i0(beta)


def _hann_centered(M: int, half_sample_odd: bool = False) -> np.ndarray:
    j = np.arange(-(M // 2), M - M // 2, dtype=np.float64)
    if half_sample_odd and M % 2:
        j = j - 0.5
    return 0.5 + 0.5 * np.cos(2.0 * np.pi * j / M)


# The design choices the call sites do not fix (band-0 / last-band / Nyquist-band length rule, window sampling, centre
# rounding).  Restated here as a plain dict, independently of the product's CQTRules dataclass
# (audio_inpainting_diffusion_amd/cqt.py documents every value); tests/test_cqt_conformance.py selects between them
# from a fixture dumped from the real package (tests/golden/make_cqt_golden.py) when one is available.
DEFAULT_RULES = dict(band0_len="constq", last_len="neighbours", nyq_len="gap",
                     window_sampling="integer", centre_rounding="nearest", last_centre="grid")


class OracleCQT:
    """Same constructor / method surface as the external ``CQT_nsgt`` in mode="oct"."""

    def __init__(self, numocts: int, binsoct: int, mode: str = "oct", window=("kaiser", 1.0),
                 fs: float = 44100, audio_len: int = 44100, dtype=torch.float32, device="cpu", rules=None):
        assert mode == "oct", "only the octave mode used by the reference U-Net is restated"
        L = int(audio_len)
        assert L % 2 == 0, "even signal length required"
        self.numocts, self.binsoct, self.fs, self.Ls = int(numocts), int(binsoct), float(fs), L
        self.dtype, self.device = dtype, torch.device(device)
        self.cdtype = torch.complex64 if dtype == torch.float32 else torch.complex128
        R = dict(DEFAULT_RULES)
        R.update(dict(rules or {}))
        self.rules = R
        K = self.numocts * self.binsoct
        step = 2.0 ** (1.0 / self.binsoct)
        fmin = (self.fs / 2.0) / 2.0 ** self.numocts
        frqs = fmin * step ** np.arange(K, dtype=np.float64)
        b = frqs * L / self.fs  # centre frequencies in DFT bins
        nyq = L / 2.0
        q = math.sqrt(step) / (step - 1.0) / 2.0

        # ---- band list: index 0 = DC, 1..K = constant-Q bands, K+1 = Nyquist -------------------
        centre = np.concatenate(([0.0], b, [nyq]))
        M = np.zeros(K + 2, dtype=np.int64)
        M[0] = int(np.round(2.0 * b[0]))
        if R["band0_len"] == "constq":
            M[1] = int(np.round(b[0] * (step - 1.0 / step)))
        elif R["band0_len"] == "f_over_q":
            M[1] = int(np.round(b[0] / q))
        else:
            assert R["band0_len"] == "to_dc"
            M[1] = int(np.round(b[1] - 0.0))
        for k in range(2, K + 1):
            M[k] = int(np.round(centre[k + 1] - centre[k - 1]))
        if R["last_len"] == "f_over_q":
            M[K] = int(np.round(b[K - 1] / q))
        else:
            assert R["last_len"] == "neighbours"
        if R["nyq_len"] == "f_over_q":
            M[K + 1] = int(np.round(b[K - 1] / q))
        else:
            assert R["nyq_len"] == "gap"
            M[K + 1] = int(np.round(2.0 * (nyq - b[K - 1])))
        M = np.maximum(M, 4)
        self.Lg = M.copy()  # window lengths (support in DFT bins)
        if R["last_centre"] == "midpoint":      # the highest constant-Q band moves half-way between its lower neighbour and Nyquist
            centre[K] = (centre[K - 1] + centre[K + 1]) / 2.0
        else:
            assert R["last_centre"] == "grid"
        if R["centre_rounding"] == "even":
            self.rc = (np.round(centre / 2.0) * 2).astype(np.int64)
        else:
            assert R["centre_rounding"] == "nearest"
            self.rc = np.round(centre).astype(np.int64)  # rounded centre bins

        half = R["window_sampling"] == "half_sample_odd"
        assert half or R["window_sampling"] == "integer"
        if isinstance(window, (tuple, list)):
            assert window[0] == "kaiser"
            mk = lambda m: _kaiser_centered(int(m), float(window[1]), half)
        elif window == "hann":
            mk = lambda m: _hann_centered(int(m), half)
        else:
            raise NotImplementedError(window)
        self.g = [mk(m) for m in self.Lg]

        # ---- octave sampling lengths --------------------------------------------------------
        self.size_per_oct: List[int] = []
        self.M = self.Lg.copy()
        for o in range(self.numocts):
            sl = slice(1 + o * self.binsoct, 1 + (o + 1) * self.binsoct)
            T = _next_pow2(int(self.Lg[sl].max()))
            self.size_per_oct.append(T)
            self.M[sl] = T
        # DC / Nyquist bands keep M = their own length (painless), they are never emitted.

        # ---- frame operator diagonal over the full circle (all bands + negative-frequency mirrors)
        S = np.zeros(L, dtype=np.float64)
        for k in range(K + 2):
            j = np.arange(-(self.Lg[k] // 2), self.Lg[k] - self.Lg[k] // 2)
            w = self.M[k] * self.g[k] ** 2
            np.add.at(S, (self.rc[k] + j) % L, w)
            if 1 <= k <= K:  # mirrored band at -r_k (window reflected)
                np.add.at(S, (-self.rc[k] - j) % L, w)
        self.S = S
        self.gd = []
        for k in range(K + 2):
            j = np.arange(-(self.Lg[k] // 2), self.Lg[k] - self.Lg[k] // 2)
            self.gd.append(self.g[k] / S[(self.rc[k] + j) % L])

        # ---- DC+Nyquist projector ---------------------------------------------------------------
        Hl = np.zeros(L, dtype=np.float64)
        for k in (0, K + 1):
            j = np.arange(-(self.Lg[k] // 2), self.Lg[k] - self.Lg[k] // 2)
            np.add.at(Hl, (self.rc[k] + j) % L, self.M[k] * self.g[k] * self.gd[k])
        self.Hhpf = torch.tensor(1.0 - Hl[: L // 2 + 1], dtype=dtype, device=self.device)

        self._g_t = [torch.tensor(g, dtype=dtype, device=self.device) for g in self.g]
        self._gd_t = [torch.tensor(g, dtype=dtype, device=self.device) for g in self.gd]

    # -----------------------------------------------------------------------------------------
    def fwd(self, x: torch.Tensor) -> List[torch.Tensor]:
        """x[B,1,L] real -> list (low octave first) of [B,1,binsoct,T_o] complex."""
        assert x.shape[-1] == self.Ls
        B = x.shape[0]
        X = torch.fft.fft(x.reshape(B, self.Ls).to(self.dtype), dim=-1)
        out = []
        for o in range(self.numocts):
            T = self.size_per_oct[o]
            C = torch.zeros(B, self.binsoct, T, dtype=self.cdtype, device=x.device)
            for i in range(self.binsoct):
                k = 1 + o * self.binsoct + i
                Lg = int(self.Lg[k])
                j = torch.arange(-(Lg // 2), Lg - Lg // 2, device=x.device)
                buf = torch.zeros(B, T, dtype=self.cdtype, device=x.device)
                buf[:, j % T] = X[:, (int(self.rc[k]) + j) % self.Ls] * self._g_t[k]
                C[:, i] = torch.fft.ifft(buf, dim=-1)
            out.append(C.unsqueeze(1))
        return out

    def bwd(self, c: Sequence[torch.Tensor]) -> torch.Tensor:
        """list of [B,1,binsoct,T_o] complex -> [B,1,L] real."""
        B = c[0].shape[0]
        Y = torch.zeros(B, self.Ls, dtype=self.cdtype, device=c[0].device)
        for o in range(self.numocts):
            T = self.size_per_oct[o]
            Co = c[o].reshape(B, self.binsoct, T).to(self.cdtype)
            Fo = torch.fft.fft(Co, dim=-1)
            for i in range(self.binsoct):
                k = 1 + o * self.binsoct + i
                Lg = int(self.Lg[k])
                j = torch.arange(-(Lg // 2), Lg - Lg // 2, device=Co.device)
                Y[:, (int(self.rc[k]) + j) % self.Ls] += Fo[:, i, j % T] * (float(self.M[k]) * self._gd_t[k])
        y = torch.fft.irfft(Y[:, : self.Ls // 2 + 1], n=self.Ls, dim=-1)
        return y.unsqueeze(1)

    def apply_hpf_DC(self, x: torch.Tensor) -> torch.Tensor:
        """x[B,L] -> x with the DC- and Nyquist-band frame components removed."""
        X = torch.fft.rfft(x.to(self.dtype), dim=-1)
        return torch.fft.irfft(X * self.Hhpf.to(x.device), n=self.Ls, dim=-1)
