"""Two-lane launch plans (plan.py): the read / write sets of every op, the derived dependencies, the event schedule and its race check;
results on two streams (eager and inside a captured HIP graph) are bit-identical to the single-stream order."""
import ast
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _small(tag="a"):
    from audio_inpainting_diffusion_amd.config import small_args
    from audio_inpainting_diffusion_amd.init import seeded_init_
    from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
    z = np.load(os.path.join(GOLDEN, f"unet_small_{tag}.npz"))
    kw = ast.literal_eval(str(z["cfg"]))
    net = seeded_init_(Unet_CQT_oct_with_attention(small_args(**kw), torch.device(DEV)), int(z["seed"]), gate_scale=10.0, affine_scale=10.0)
    return net, z, kw


def test_view_disjointness_rules():
    from audio_inpainting_diffusion_amd.plan import _disjoint, _range
    X = torch.zeros(2, 8, 12, 16, device=DEV)
    r = _range
    assert _disjoint(r(X[:, :, :4]), r(X[:, :, 4:])) and _disjoint(r(X[:, :4]), r(X[:, 4:])) and _disjoint(r(X[:, :, :, :8]), r(X[:, :, :, 8:]))
    assert not _disjoint(r(X[:, :, :4]), r(X[:, :, 3:6])) and not _disjoint(r(X[:, :, :4]), r(X[:, :4])) and not _disjoint(r(X[:, :, :4]), r(X))
    assert not _disjoint(r(X[:, :, :4]), r(X.view(2, 8, 192)[:, :, :64]))      # different stride tuples: conservatively overlapping


@pytest.mark.parametrize("tag", ["a", "c"])
def test_plans_pass_the_race_check_and_have_two_lanes(tag):
    net, z, kw = _small(tag)
    st = net._state(2)
    net._bwd_plan(st)
    for nm in ("plan_body", "plan_bwd"):
        pl = st[nm]
        lanes = {o.lane for o in pl.ops}
        assert lanes == {0, 1}, lanes
        deps = pl.dependencies()
        assert sum(len(d) for d in deps) > len(pl.ops) // 2          # (a plan without dependencies would mean empty read / write sets)
        edges = pl.check()
        n1 = sum(o.lane for o in pl.ops)
        print(f"{tag} {nm}: {len(pl.ops)} ops, {n1} on lane 1, {edges} cross-lane event edges")
        assert 0 < edges < len(pl.ops)
    # every op but the plan's sources reads something that an earlier op or the caller wrote, and writes something
    assert all(o.writes for o in st["plan_body"].ops)


def test_race_check_catches_a_missing_edge():
    net, z, kw = _small("a")
    st = net._state(2)
    pl = st["plan_body"]
    waits, record, deps = pl.schedule()
    j = next(j for j, w in enumerate(waits) if w)
    dropped = waits[j].pop()
    try:
        with pytest.raises(AssertionError, match="may run before"):
            pl.check()
    finally:
        waits[j].append(dropped)
    pl.check()


@pytest.mark.parametrize("B", [1, 2])
def test_two_lanes_are_bit_identical_to_one(B):
    """guided evaluation (forward + input-VJP): one stream vs two lanes (eager) vs two lanes inside the captured HIP graph"""
    net, z, kw = _small("a")
    Ls = kw["audio_len"]
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(B, Ls, generator=g) * 0.5).to(DEV)
    y = (torch.randn(B, Ls, generator=g) * 0.063).to(DEV)
    mask = torch.ones(1, Ls, device=DEV)
    mask[:, 1800:2300] = 0
    co = (-0.2, 1.5, 0.1, 0.3)
    res = {}
    for mode in ("one", "two", "two+graph"):
        net._states.clear()
        net.lanes_max_batch = 0 if mode == "one" else 3
        net.use_graphs = mode == "two+graph"
        outs = [net.denoise_guided(x, *co, True, y, mask) + (net.denoise(x, *co, False),) for _ in range(3)]     # third call: graph replay
        res[mode] = outs[-1]
        for o in outs[:-1]:
            assert all(torch.equal(a, b) for a, b in zip(o, outs[-1]))
    st = net._state(B)
    assert st["plan_body"].lanes == 2 and "graphs" in st
    for mode in ("two", "two+graph"):
        for a, b in zip(res["one"], res[mode]):
            assert torch.equal(a, b), mode


def test_full_size_plans_pass_the_race_check():
    from audio_inpainting_diffusion_amd.config import make_args
    from audio_inpainting_diffusion_amd.init import seeded_init_
    from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
    net = seeded_init_(Unet_CQT_oct_with_attention(make_args("maestro22k"), torch.device(DEV)), 0)
    st = net._state(1)
    net._bwd_plan(st)
    for nm in ("plan_body", "plan_bwd"):
        pl = st[nm]
        edges = pl.check()
        print(f"full size {nm}: {len(pl.ops)} ops, {sum(o.lane for o in pl.ops)} on lane 1, {edges} cross-lane event edges")
