import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def rel_l2(a, b):
    import torch
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + 1e-300))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def projected_rel_error(z, tag, tensors, small_tol=1e-3):
    """tests/golden/train_small.npz pins every trained tensor by four seeded random projections and its squared norm (make_golden.py::gen_training):
    returns the estimated rel-L2 error of ``tensors`` (name -> torch tensor) over all tensors: sqrt(sum mean_j dproj_j^2 / sum |ref|^2)."""
    import numpy as np
    from audio_inpainting_diffusion_amd.init import seeded_normal
    num = den = 0.0
    for i, k in enumerate([str(n) for n in z["names"]]):
        ref = z[f"proj.{tag}.{k}"]
        v = tensors[k].detach().double().reshape(-1).cpu().numpy()
        probes = np.stack([seeded_normal(9000 + j, i, v.size) for j in range(4)]).astype(np.float64)
        num += float(np.mean((probes @ v - ref[:4]) ** 2))
        den += float(ref[4])
        if f"{tag}.{k}" in z.files:
            # (zero-initialised biases are pure Adam steps lr * g / (|g| + eps) of near-eps gradients: fp32 noise in g shows up at 1e-4 there)
            assert rel_l2(tensors[k].detach().cpu(), z[f"{tag}.{k}"]) < small_tol, k
    return (num / den) ** 0.5


@pytest.fixture(autouse=True)
def _release_gpu_memory_between_tests(request):
    """Full-size launch plans own tens of GB (61 GB at batch 8, twice that at batch 16): drop what a test left behind before the next one builds its
    own network, so that consecutive full-size cases do not pile up on the 288 GB of one MI355X."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import gc
        import torch
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
