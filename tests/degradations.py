"""The degradation callables of tests/golden/sampler_resample.npz, rebuilt from the stored FIR taps (same definitions as
tests/golden/make_golden.py::resample_degradations; test fixtures, not reference code)."""
import torch


def resample_degradations(k):
    kk = torch.as_tensor(k, dtype=torch.float32).view(1, 1, -1)
    return {0: lambda x: torch.nn.functional.conv1d(x.unsqueeze(1), kk.to(x.device), stride=2, padding=kk.shape[-1] // 2).squeeze(1),   # low-pass + decimate by 2
            1: lambda x: 0.05 * torch.tanh(x / 0.05)}                                                                                  # soft clipper (non-linear)
