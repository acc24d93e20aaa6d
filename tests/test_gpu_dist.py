"""Multi-GPU path on ONE GPU: world-size independence of a sampler run, and the self-launching bench entry point.

RCCL refuses two ranks on one device, so on a 1-GPU box the two ranks share cuda:0 and the collectives run on the
gloo backend (staged through host memory by dist.py) -- the sharding, per-item seeds, weight broadcast and output
gather are the same code the 8-GPU run uses (BASELINE.json configs[2])."""
import ast
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(dev, seed_offset=0, T=2, xi=0.25):
    from audio_inpainting_diffusion_amd.config import small_args
    from audio_inpainting_diffusion_amd.init import seeded_init_
    from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
    z = np.load(os.path.join(GOLDEN, "unet_small_a.npz"))
    kw = ast.literal_eval(str(z["cfg"]))
    args = small_args(**kw)
    args.tester.T, args.tester.posterior_sampling.xi = T, xi
    args.tester.data_consistency.hann_size = 20
    net = Unet_CQT_oct_with_attention(args, torch.device(dev))
    seeded_init_(net, int(z["seed"]) + seed_offset, gate_scale=10.0, affine_scale=10.0)
    return net, args, kw


def _segments(n, L):
    from audio_inpainting_diffusion_amd.init import seeded_normal
    y = torch.stack([torch.from_numpy(seeded_normal(31, g, L)) for g in range(n)]) * 0.063
    mask = torch.ones(1, L)
    mask[:, 1800:2300] = 0
    return y, mask


def _sample(net, args, y, mask, seeds, dev):
    from audio_inpainting_diffusion_amd.edm import EDM
    from audio_inpainting_diffusion_amd.sampler import Sampler
    smp = Sampler(model=net, diff_params=EDM(args), args=args)
    smp.seeds = seeds
    return smp.predict_inpainting((y * mask).to(dev), mask.to(dev))


def _worker(rank, world, port, n_items, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      AID_DIST_BACKEND="gloo")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from audio_inpainting_diffusion_amd import dist as D
    r, _, w = D.init_distributed()
    torch.cuda.set_device(0)
    net, args, kw = _setup("cuda:0", seed_offset=rank)      # ranks start with DIFFERENT weights: the broadcast must fix that
    nbytes = D.broadcast_parameters(net, src=0)
    y, mask = _segments(n_items, kw["audio_len"])
    lo, hi = D.shard_range(n_items, r, w)
    out = _sample(net, args, y[lo:hi], mask, D.item_seeds(500, lo, hi), "cuda:0")
    allout = D.gather_outputs(out, n_items)
    D.barrier()
    q.put((rank, nbytes, (lo, hi), allout.cpu().numpy()))
    torch.distributed.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_one_process():
    """gather(2 ranks x B=2) == single-process B=4 sampler output (guided branch, T=2, small golden network)."""
    from audio_inpainting_diffusion_amd import dist as D
    n_items = 4
    net, args, kw = _setup(DEV)
    y, mask = _segments(n_items, kw["audio_len"])
    ref = _sample(net, args, y, mask, D.item_seeds(500, 0, n_items), DEV).cpu()
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [r[2] for r in res] == [(0, 2), (2, 4)]
    for rank, nbytes, _, allout in res:
        e = rel_l2(allout, ref)
        print(f"rank {rank}: gathered 2x2 segments vs single-process B=4: rel-L2 = {e:.2e} ({nbytes / 1e6:.2f} MB broadcast)")
        assert allout.shape == tuple(ref.shape) and e < 1e-6
    assert np.array_equal(res[0][3], res[1][3])


def test_flat_parameter_storage_survives_load_and_repack():
    """flatten_parameters_ re-homes the weights as views of one buffer; loading a state_dict afterwards must keep the
    views and refresh the kernel-side packs."""
    from audio_inpainting_diffusion_amd import dist as D
    net, args, kw = _setup(DEV)
    z = np.load(os.path.join(GOLDEN, "unet_small_a.npz"))
    x, cn = torch.from_numpy(z["x"]).to(DEV), torch.from_numpy(z["cnoise"]).to(DEV)
    with torch.no_grad():
        y0 = net(x, cn)
    flat = D.flatten_parameters_(net)
    assert all(p.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for p in net.parameters())
    assert D.flatten_parameters_(net) is flat                # idempotent
    with torch.no_grad():
        assert torch.equal(net(x, cn), y0)
    other, _, _ = _setup(DEV, seed_offset=7)
    with torch.no_grad():
        y_other = other(x, cn)
    net.load_state_dict(other.state_dict())
    assert all(p.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for p in net.parameters())
    with torch.no_grad():
        assert torch.equal(net(x, cn), y_other)


@pytest.mark.parametrize("xi,dctype", [(0.25, "end"), (0.0, "end"), (0.25, "always")])
def test_sampler_data_consistency_types_vs_oracle(xi, dctype):
    """data_consistency.type on the GPU sampler against the oracle (itself pinned by tests/golden/sampler_dc.npz)."""
    from audio_inpainting_diffusion_amd.edm import EDM
    from audio_inpainting_diffusion_amd.sampler import Sampler
    from oracle.edm import OracleEDM
    from oracle.nsgt_cqt import OracleCQT
    from oracle.sampler import OracleSampler
    from oracle.unet import OracleUnet
    net, args, kw = _setup(DEV, T=3, xi=xi)
    args.tester.data_consistency.type = dctype
    Ls = kw["audio_len"]
    y, mask = _segments(2, Ls)
    smp = Sampler(model=net, diff_params=EDM(args), args=args)
    smp.seeds = [5, 6]
    # the mask is handed over as float64 with a non-unit stride: the sampler must normalise it
    mask_odd = torch.stack([mask.double(), mask.double()], dim=-1)[..., 0]
    assert mask_odd.stride(-1) == 2
    out = smp.predict_inpainting((y * mask).to(DEV), mask_odd.to(DEV))
    cqt = OracleCQT(kw["num_octs"], kw["bins_per_oct"], "oct", ("kaiser", 1), kw["fs"], Ls)
    orc = OracleUnet(kw["num_octs"], kw["bins_per_oct"], cqt).load_state_dict(net.state_dict())
    osmp = OracleSampler(orc, OracleEDM(), T=3, xi=xi, hann_size=20, audio_len=Ls, dc_type=dctype)
    ref = osmp.predict_inpainting(y * mask, mask, seeds=[5, 6])
    e = rel_l2(out.cpu(), ref)
    print(f"xi={xi} type={dctype}: final rel-L2 vs oracle = {e:.2e}")
    assert e < 5e-4
    if dctype == "end" or xi == 0.0:
        assert float((out.cpu() - y)[:, :1700].abs().max()) < 1e-5        # projected after the loop / at every evaluation


def test_replacement_branch_without_projection_raises():
    from audio_inpainting_diffusion_amd.edm import EDM
    from audio_inpainting_diffusion_amd.sampler import Sampler
    net, args, kw = _setup(DEV, T=2, xi=0.0)
    args.tester.data_consistency.use = False
    y, mask = _segments(1, kw["audio_len"])
    smp = Sampler(model=net, diff_params=EDM(args), args=args)
    with pytest.raises(AttributeError):
        smp.predict_inpainting((y * mask).to(DEV), mask.to(DEV))


def test_sub_batch_streams_are_bit_identical_to_one_stream():
    """network.denoise / denoise_guided cut a batch into sub-batches on concurrent HIP streams: same bits as one stream WHEN every sub-batch launch
    takes the kernel instance the whole-batch launch takes -- true for this small network, whose launches all stay under one tile per CU.  (Every
    kernel's per-sample arithmetic is independent of the batch, but the Winograd form / tile instance of a 5x3 layer is a function of the LAUNCH shape,
    batch included: at full size a split can move a layer to another form.  test_full_size_split_vs_unsplit_agree_to_rounding states that tolerance.)"""
    from oracle.edm import OracleEDM
    net, args, kw = _setup(DEV)
    B, Ls = 7, kw["audio_len"]
    y, mask = _segments(B, Ls)
    g0 = torch.Generator().manual_seed(2)
    x = (torch.randn(B, Ls, generator=g0) * 0.4).to(DEV)
    edm = OracleEDM()
    s = torch.rand(B, 1, generator=g0) * 0.8 + 0.05
    v = lambda t: t.reshape(-1).to(DEV).contiguous()
    co = (v(edm.cnoise(s)), v(edm.cin(s)), v(edm.cskip(s)), v(edm.cout(s)))
    masks = torch.ones(B, Ls)
    for b in range(B):
        masks[b, 1000 + 100 * b: 1400 + 100 * b] = 0
    yd, md = (y * masks).to(DEV).contiguous(), masks.to(DEV)
    res = {}
    for n in (1, 2, 3, 7):                                   # 7: sub-batches of ONE segment (no split-K tiles there: those are for a whole batch of one)
        net.split_streams = n
        assert len(net.states_of(B)) == n
        for _ in range(2):                                   # second pass re-uses the launch plans
            res[n] = (net.denoise(x, *co, True), *net.denoise_guided(x, *co, True, yd, md))
    torch.cuda.synchronize()
    for n in (2, 3, 7):
        for a_, b_ in zip(res[1], res[n]):
            assert torch.equal(a_, b_)
    net.split_streams = None
    assert net._n_split(8) == 2 and net._n_split(4) == 2 and net._n_split(2) == 1

def test_cu_partitioned_sub_batch_streams_are_bit_identical_and_race_free():
    """EXPERIMENT mode network.cu_partition (round 6): inside every sub-batch the launch plans run on two CU-masked streams, lanes by kernel TYPE (MFMA-bound
    on 256 - N CUs, HBM-bound passes on N), with the cross-lane dependencies derived from the ops' read / write sets.  Same kernels, same launch shapes:
    the result must equal the free-running sub-batch streams to the bit, repeatedly (a missing edge shows as a race), and Plan.check() must cover every
    dependency of both re-tagged plans."""
    from oracle.edm import OracleEDM
    net, args, kw = _setup(DEV)
    B, Ls = 6, kw["audio_len"]
    y, _ = _segments(B, Ls)
    g0 = torch.Generator().manual_seed(5)
    x = (torch.randn(B, Ls, generator=g0) * 0.4).to(DEV)
    edm = OracleEDM()
    s = torch.rand(B, 1, generator=g0) * 0.8 + 0.05
    v = lambda t: t.reshape(-1).to(DEV).contiguous()
    co = (v(edm.cnoise(s)), v(edm.cin(s)), v(edm.cskip(s)), v(edm.cout(s)))
    masks = torch.ones(B, Ls)
    for b in range(B):
        masks[b, 900 + 100 * b: 1300 + 100 * b] = 0
    yd, md = (y * masks).to(DEV).contiguous(), masks.to(DEV)
    net.split_streams = 2
    ref = (net.denoise(x, *co, True), *net.denoise_guided(x, *co, True, yd, md))
    torch.cuda.synchronize()
    net._states.clear()
    net.cu_partition = 64
    net.cu_partition_ridge = 0.5                      # (reduced network: its convs are tiny -- make some of them count as MFMA-bound so both lanes are populated)
    try:
        for rep in range(4):
            got = (net.denoise(x, *co, True), *net.denoise_guided(x, *co, True, yd, md))
            torch.cuda.synchronize()
            for a_, b_ in zip(ref, got):
                assert torch.equal(a_, b_), f"repetition {rep}"
        edges = 0
        for st in net.states_of(B):
            for pl in (st["plan_body"], st["plan_bwd"]):
                assert pl.lanes == 2 and {o.lane for o in pl.ops} == {0, 1}
                edges += pl.check()
        assert edges > 0
    finally:
        net.cu_partition, net.split_streams = 0, None
        net._states.clear()


def test_full_size_split_vs_unsplit_agree_to_rounding():
    """Full-size network, batch 4, one guided evaluation: two sub-batches of two on concurrent streams against the unsplit batch.  The Winograd form and
    tile instance of a 5x3 layer are chosen from the launch shape (aid_conv2d_wino_form / aid_conv2d_wino2d_wanted: batch 2 and batch 4 launches of the
    same layer may differ -- 2-D form, F(8,3), F(4,3), K-group instances), so the two schedules agree to rounding, not to the bit: <= 5e-6."""
    from audio_inpainting_diffusion_amd.config import make_args
    from audio_inpainting_diffusion_amd.init import seeded_init_, seeded_normal
    from audio_inpainting_diffusion_amd.masks import long_gap_mask
    from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
    from oracle.edm import OracleEDM
    args = make_args("maestro22k")
    Ls, B = args.exp.audio_len, 4
    net = seeded_init_(Unet_CQT_oct_with_attention(args, torch.device(DEV)), 0, gate_scale=10.0, affine_scale=10.0)
    edm = OracleEDM()
    x = (torch.from_numpy(seeded_normal(31, 0, B * Ls)).reshape(B, Ls) * 0.3).to(DEV)
    y = torch.from_numpy(seeded_normal(32, 0, B * Ls)).reshape(B, Ls) * 0.063
    mask = long_gap_mask(Ls, 22050, 300)
    s = torch.tensor([[0.2], [0.6], [1.5], [4.0]])
    v = lambda t: t.reshape(-1).to(DEV).contiguous()
    co = (v(edm.cnoise(s)), v(edm.cin(s)), v(edm.cskip(s)), v(edm.cout(s)))
    res = {}
    for n in (1, 2):
        net.split_streams = n
        assert len(net.states_of(B)) == n
        res[n] = tuple(t.clone() for t in net.denoise_guided(x, *co, True, (y * mask).to(DEV), mask.to(DEV)))
    torch.cuda.synchronize()
    e = [rel_l2(a_.cpu(), b_.cpu()) for a_, b_ in zip(res[1][:2], res[2][:2])]
    print(f"full size, batch 4: 2 sub-batches of 2 vs unsplit: x_hat rel-L2 {e[0]:.2e}, rec_grads rel-L2 {e[1]:.2e}")
    assert e[0] < 5e-6 and e[1] < 5e-5


@pytest.mark.parametrize("B", [1, 2])
def test_hip_graph_replay_equals_eager_launches(B):
    """Small batches replay a captured HIP graph of the evaluation (network._graph_run): same bits as eager launches, for
    changing inputs, on the forward-only and the guided entry points and through a whole sampler run."""
    from audio_inpainting_diffusion_amd.edm import EDM
    from audio_inpainting_diffusion_amd.sampler import Sampler
    from oracle.edm import OracleEDM
    net, args, kw = _setup(DEV, T=3, xi=0.25)
    Ls = kw["audio_len"]
    y, mask = _segments(B, Ls)
    yd, md = (y * mask).to(DEV).contiguous(), mask.to(DEV)
    edm = OracleEDM()
    v = lambda t: t.reshape(-1).to(DEV).contiguous()
    g0 = torch.Generator().manual_seed(5)
    calls = []
    for _ in range(4):
        x = (torch.randn(B, Ls, generator=g0) * 0.4).to(DEV)
        s = torch.rand(B, 1, generator=g0) * 0.8 + 0.05
        calls.append((x, (v(edm.cnoise(s)), v(edm.cin(s)), v(edm.cskip(s)), v(edm.cout(s)))))
    out = {}
    for mode in (False, True):
        net.use_graphs = mode
        net._states.clear()
        out[mode] = [(net.denoise(x, *co, False), *net.denoise_guided(x, *co, True, yd, md)) for x, co in calls]
        if mode:
            st = net._state(B)
            assert sum("graph" in e for e in st["graphs"].values()) == 2        # captured on the second call, replayed after
    torch.cuda.synchronize()
    for ea, gr in zip(out[False], out[True]):
        for a_, b_ in zip(ea, gr):
            assert torch.equal(a_, b_)
    res = {}
    for mode in (False, True):
        net.use_graphs = mode
        net._states.clear()
        smp = Sampler(model=net, diff_params=EDM(args), args=args)
        smp.seeds = list(range(40, 40 + B))
        res[mode] = smp.predict_inpainting(yd, md)
    assert torch.equal(res[False], res[True])

@pytest.mark.parametrize("T,short_gaps", [(70, True), (128, False)])
def test_complete_T70_T128_schedules_vs_oracle(T, short_gaps):
    """The WHOLE sampling loops of BASELINE configs[3] (T = 70, four short gaps) and configs[4] (T = 128, one long gap) -- every Heun step, the churn of every
    step, the final Euler step onto t = 0 -- on the reduced-size network (only the first two steps of these schedules were in a test;
    the full-size complete runs are timed in profiles/r05_e2e_full_runs.txt): HIP sampler + HIP network against the oracle sampler + oracle network over
    139 / 255 chained guided evaluations per segment, two segments with their own seeds."""
    from oracle.edm import OracleEDM
    from oracle.nsgt_cqt import OracleCQT
    from oracle.sampler import OracleSampler
    from oracle.unet import OracleUnet
    net, args, kw = _setup(DEV, T=T, xi=0.25)
    Ls = kw["audio_len"]
    y, mask = _segments(2, Ls)
    if short_gaps:                                              # (inpainting_tester_shortgaps.yaml: several short gaps)
        mask = torch.ones(1, Ls)
        for g0 in (700, 1500, 2300, 3100):
            mask[:, g0:g0 + 90] = 0
    out = _sample(net, args, y, mask, [11, 12], DEV)
    cqt = OracleCQT(kw["num_octs"], kw["bins_per_oct"], "oct", ("kaiser", 1), kw["fs"], Ls)
    orc = OracleUnet(kw["num_octs"], kw["bins_per_oct"], cqt).load_state_dict(net.state_dict())
    osmp = OracleSampler(orc, OracleEDM(), T=T, xi=0.25, hann_size=20, audio_len=Ls)
    nthr = torch.get_num_threads()
    torch.set_num_threads(min(8, nthr))                         # (the reduced-size oracle is all small ops: 128 threads make it 6x slower than 8)
    try:
        ref = osmp.predict_inpainting(y * mask, mask, seeds=[11, 12])
    finally:
        torch.set_num_threads(nthr)
    e = rel_l2(out.cpu(), ref)
    keep = mask[0].bool()
    print(f"complete T = {T} schedule (2 x {2 * T - 1} chained guided evaluations, {'4 short gaps' if short_gaps else 'one long gap'}): final rel-L2 vs oracle = {e:.2e}")
    assert torch.isfinite(out).all() and e < 5e-5                              # (measured 5e-7 ... 7e-7)
    assert float((out.cpu()[:, keep] - y[:, keep]).abs().max()) < 0.2          # known samples stay near the observation (smooth-mask projection at every step)


def _full_step(net, args, y, mask, seeds, dev):
    """prior draw + ONE guided Heun step (churn, two evaluations with the input-VJP, projection, Heun combine) -> x after the step"""
    from audio_inpainting_diffusion_amd.edm import EDM
    from audio_inpainting_diffusion_amd.sampler import Sampler
    smp = Sampler(model=net, diff_params=EDM(args), args=args)
    smp.seeds = seeds
    smp.setup_inpainting((y * mask).to(dev), mask.to(dev))
    st = smp.begin(tuple(y.shape), torch.device(dev))
    smp.step(st, 0)
    return st["x"]


def _full_setup(dev, seed, n_items=4):
    from audio_inpainting_diffusion_amd.config import make_args
    from audio_inpainting_diffusion_amd.init import seeded_init_, seeded_normal
    from audio_inpainting_diffusion_amd.masks import mask_from_args
    from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
    args = make_args("maestro22k", audio_len=184184, T=36, gap_ms=300.0, xi=0.25)
    net = seeded_init_(Unet_CQT_oct_with_attention(args, torch.device(dev)), seed, gate_scale=10.0, affine_scale=10.0)
    L = args.exp.audio_len
    y = torch.stack([torch.from_numpy(seeded_normal(41, g, L)) for g in range(n_items)]) * 0.063
    return net, args, y, mask_from_args(args, generator=torch.Generator().manual_seed(99))


def _worker_full8(rank, world, port, q):
    """one of EIGHT ranks sharing cuda:0, one segment each (BASELINE configs[2]'s world size; collectives on gloo)"""
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      LOCAL_WORLD_SIZE=str(world), AID_DIST_BACKEND="gloo")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from audio_inpainting_diffusion_amd import dist as D
    r, local, w = D.init_distributed()
    torch.cuda.set_device(0)
    nthr = D.bind_rank_to_gpu_numa(local, w)                       # every rank gets its share of the host cores (of its GPU's NUMA node when sysfs says)
    net, args, y, mask = _full_setup("cuda:0", seed=rank % 2, n_items=world)     # odd ranks start from DIFFERENT weights: the one broadcast must fix that
    torch.cuda.synchronize()
    t0 = time.time()
    nbytes = D.broadcast_parameters(net, src=0)
    torch.cuda.synchronize()
    t_bcast = time.time() - t0
    per = int(os.environ.get("AID_TEST_SEGMENTS_PER_RANK", "1"))
    if per > 1:                                                    # configs[2]: 8 segments per rank -- the rank's own items only (64 full-length rows are rebuilt from seeds)
        from audio_inpainting_diffusion_amd.init import seeded_normal
        lo, hi = D.shard_range(world * per, r, w)
        y = torch.stack([torch.from_numpy(seeded_normal(41, g, args.exp.audio_len)) for g in range(lo, hi)]) * 0.063
        out = _full_step(net, args, y, mask, D.item_seeds(700, lo, hi), "cuda:0")
        peak = torch.cuda.max_memory_allocated() / 2 ** 30
        allout = D.gather_outputs(out, world * per)
        D.barrier()
        keep = np.concatenate([allout[:per].cpu().numpy(), allout[-per:].cpu().numpy()])         # (the first and the last rank's shards: what the test compares)
        q.put((rank, nbytes, (lo, hi), keep, t_bcast, nthr, torch.distributed.get_backend(), float(allout.double().pow(2).sum()), peak))
        torch.distributed.destroy_process_group()
        return
    lo, hi = D.shard_range(world, r, w)
    out = _full_step(net, args, y[lo:hi], mask, D.item_seeds(700, lo, hi), "cuda:0")
    allout = D.gather_outputs(out, world)
    D.barrier()
    q.put((rank, nbytes, (lo, hi), allout.cpu().numpy(), t_bcast, nthr, torch.distributed.get_backend()))
    torch.distributed.destroy_process_group()


def _worker_full(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      AID_DIST_BACKEND="gloo")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from audio_inpainting_diffusion_amd import dist as D
    r, _, w = D.init_distributed()
    torch.cuda.set_device(0)
    torch.set_num_threads(max(1, (os.cpu_count() or 8) // world))
    net, args, y, mask = _full_setup("cuda:0", seed=rank)         # rank 1 starts from DIFFERENT weights: the one broadcast must fix that
    nbytes = D.broadcast_parameters(net, src=0)
    lo, hi = D.shard_range(4, r, w)
    out = _full_step(net, args, y[lo:hi], mask, D.item_seeds(700, lo, hi), "cuda:0")
    allout = D.gather_outputs(out, 4)
    D.barrier()
    q.put((rank, nbytes, (lo, hi), allout.cpu().numpy()))
    torch.distributed.destroy_process_group()


def test_full_size_eight_ranks_on_one_gpu_equal_one_process():
    """Dress rehearsal of BASELINE configs[2]'s WORLD SIZE before the driver's first 8-GPU run: eight processes sharing cuda:0 (gloo; RCCL refuses
    two ranks on one device), the full-size 22.05 kHz network in each, ranks start from two different weight sets, ONE in-place 745 MB broadcast
    per rank, one segment per rank (global-index seeds), one guided Heun step, gather in global segment order == the single-process batch-8 run.
    Tolerance 5e-6, not 0: a whole batch of ONE runs the split-K instances of the 5x3 kernel (one extra association per tile, network.py)."""
    from audio_inpainting_diffusion_amd import dist as D
    world = 8
    net, args, y, mask = _full_setup(DEV, seed=0, n_items=world)
    ref = _full_step(net, args, y, mask, D.item_seeds(700, 0, world), DEV).cpu()
    del net
    torch.cuda.empty_cache()
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_full8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=2400) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    assert [r[2] for r in res] == [(i, i + 1) for i in range(world)]
    for rank, nbytes, _, allout, t_bcast, nthr, backend in res:
        e = rel_l2(allout, ref)
        print(f"full size, rank {rank}/8 ({backend}, {nthr} CPU threads, broadcast {nbytes / 1e6:.0f} MB in {t_bcast:.2f} s): gathered 8x1 segments vs "
              f"single-process B=8 after one guided Heun step: rel-L2 = {e:.2e}")
        assert allout.shape == tuple(ref.shape) and nbytes > 700e6 and e < 5e-6
        assert np.array_equal(allout, res[0][3])                   # every rank holds the same gathered result, in global segment order
    assert sum(r[5] or 0 for r in res) <= (os.cpu_count() or 8) or all(r[5] is None for r in res)     # the ranks' CPU shares do not overlap

def test_config2_job_shape_eight_ranks_several_segments_each_on_one_gpu():
    """BASELINE configs[2] at its FULL job shape on one GPU: 64 gap segments, 8 ranks x 8 segments each (eight processes sharing cuda:0 over gloo, the
    full-size network in each, one 745 MB broadcast per rank, global-index seeds, one guided Heun step = two evaluations with the input-VJP per
    segment, ONE all-gather of the 64 outputs).  A rank's batch of eight is the same launch sequence as a single process's batch of eight with the
    same seeds: the first and the last rank's shards of the gathered result equal those single-process runs to the BIT; every rank holds the same 64 rows.
    Eight batch-8 processes need 8 x the measured peak of one batch-8 step: where that does not fit the device the test falls back to 4 (then 2) segments
    per rank and says so (MI355X, 288 GB: a batch-8 guided step peaks at ~73 GiB -- its saved activations for the input-VJP -- so eight of them do not fit)."""
    from audio_inpainting_diffusion_amd import dist as D
    from audio_inpainting_diffusion_amd.init import seeded_normal
    world = 8
    total = torch.cuda.mem_get_info()[1] / 2 ** 30
    for per in (8, 4, 2):                               # 8 = configs[2]; fewer only when eight such processes do not fit this device's memory
        net, args, _, mask = _full_setup(DEV, seed=0, n_items=1)                                # (a fresh network per attempt: cached launch plans pin their buffers)
        Ls = args.exp.audio_len
        refs = {}
        torch.cuda.reset_peak_memory_stats()
        for r in (0, world - 1):
            lo, hi = D.shard_range(world * per, r, world)
            y = torch.stack([torch.from_numpy(seeded_normal(41, g, Ls)) for g in range(lo, hi)]) * 0.063
            refs[r] = _full_step(net, args, y, mask, D.item_seeds(700, lo, hi), DEV).cpu().numpy()
        peak = torch.cuda.max_memory_allocated() / 2 ** 30
        del net
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        free = torch.cuda.mem_get_info()[0] / 2 ** 30
        print(f"configs[2] rehearsal: one batch-{per} guided step peaks at {peak:.1f} GiB; eight ranks need {world * (peak + 2.0):.0f} of the {free:.0f} GiB free ({total:.0f} GiB device)")
        if free >= world * (peak + 2.0) + 8.0:
            break
    else:
        pytest.skip(f"8 ranks x {peak:.1f} GiB (one batch-2 step) do not fit the {free:.0f} GiB free on this device")
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    os.environ["AID_TEST_SEGMENTS_PER_RANK"] = str(per)
    try:
        procs = [ctx.Process(target=_worker_full8, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted((q.get(timeout=2400) for _ in range(world)), key=lambda t: t[0])
        for p in procs:
            p.join(timeout=600)
            assert p.exitcode == 0
    finally:
        os.environ.pop("AID_TEST_SEGMENTS_PER_RANK", None)
    assert [r[2] for r in res] == [(per * i, per * (i + 1)) for i in range(world)]
    for rank, nbytes, _, keep, t_bcast, nthr, backend, ssq, rpeak in res:
        assert keep.shape == (2 * per, Ls) and nbytes > 700e6 and np.isfinite(keep).all()
        assert np.array_equal(keep[:per], refs[0]) and np.array_equal(keep[per:], refs[world - 1])      # bit-identical to the single-process batch-8 runs of those shards
        assert ssq == res[0][7]                                                                          # every rank gathered the same 64 rows
        print(f"configs[2] shape, rank {rank}/8 ({backend}, {nthr} CPU threads, broadcast {t_bcast:.2f} s, peak {rpeak:.1f} GiB): {per} segments, shards of ranks 0 and 7 == single-process B = {per} runs")


def test_bench_self_launches_eight_ranks_rank0_only_json():
    """`python bench.py --gpus 8 --batch 1 --steps 1 --warmup 0` as the driver will start it (here: self-launched, 8 ranks on the visible GPUs):
    ONE JSON line, from rank 0, n_gpus 8, whole-job value; every rank logs its backend / device / CPU share on stderr."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0", "--batch", "1", "--roof-steps", "1", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=2400, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["value"] > 0 and j["scaling"] == "weak" and j["config"]["segments_per_gpu"] == 1
    assert j["config"]["functional_shared_gpu"] == (torch.cuda.device_count() < 8)
    import re
    ranks = re.findall(r"\[aid dist\] rank (\d+)/8 \(local \d+\): backend (\w+), device (cuda:\d+)", r.stderr)     # (the ranks' lines may interleave)
    assert sorted(int(k[0]) for k in ranks) == list(range(8)), r.stderr[-2000:]
    assert [x["rank"] for x in j["ranks"]] == list(range(8)) and [x["segments"] for x in j["ranks"]] == [[i, i + 1] for i in range(8)]
    assert all(x["wall_s"] > 0 and x["bcast_s"] >= 0 and x["cpu_threads"] >= 1 and x["pci"] for x in j["ranks"])      # the per-rank table a bad SCALE line is read from
    assert j["roofline"]["frac"] is not None and 0 < j["roofline"]["frac"] < 1.0
    assert j["config"]["backend"] in ("gloo", "nccl") and j["config"]["visible_gpus"] == torch.cuda.device_count()
    assert max(x["wall_s"] for x in j["ranks"]) * 1e3 <= j["ms_per_step"] * j["steps"] * 1.001 + 1e-6
    print("bench --gpus 8 (self-launched):", j["value"], "evals/s;", j["config"]["parallelism"])


_RCCL_SCRIPT = r"""
import json, os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from audio_inpainting_diffusion_amd import dist as D
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl" and D._active()
from test_gpu_dist import _setup
net, args, kw = _setup("cuda:0")
flat = D.flatten_parameters_(net)
before = flat.clone()
ptr = flat.data_ptr()
nbytes = D.broadcast_parameters(net, src=0)
torch.cuda.synchronize()
assert flat.data_ptr() == ptr and torch.equal(flat, before) and not D._host_staged(flat)       # in place, on device memory
big = torch.empty(745 * 1000 * 1000 // 4, dtype=torch.float32, device="cuda:0").normal_()   # the full-size network's flat buffer, by size
chk = float(big.double().sum())
dist.broadcast(big, src=0); torch.cuda.synchronize()
t0 = time.perf_counter(); dist.broadcast(big, src=0); torch.cuda.synchronize(); t_b = time.perf_counter() - t0
assert float(big.double().sum()) == chk
out = torch.arange(3 * 4096, dtype=torch.float32, device="cuda:0").reshape(3, 4096)
g = D.gather_outputs(out, 3)
assert g.is_cuda and torch.equal(g, out)
m = D.max_over_ranks(1.25, torch.device("cuda:0"))
reps = D.gather_objects(D.rank_report(0, 0, wall_s=0.5, segments=[0, 3]))        # bench.py's per-rank table over RCCL (object collective staged through cuda:0)
assert isinstance(reps, list) and len(reps) == 1 and reps[0]["rank"] == 0 and reps[0]["pci"] and reps[0]["segments"] == [0, 3] and D.rccl_version()
D.barrier(); torch.cuda.synchronize()
try:
    ver = ".".join(str(v) for v in torch.cuda.nccl.version())
except Exception:
    ver = "?"
print(json.dumps({"backend": dist.get_backend(), "bcast_bytes": nbytes, "t_bcast_745MB_s": t_b, "max": m, "rccl": ver}))
dist.destroy_process_group()
"""


def test_rccl_first_contact_world_size_one():
    """The only RCCL contact a 1-GPU box allows: a ONE-rank `nccl` (= RCCL) process group on cuda:0 running the SAME dist.py collectives the
    8-GPU job issues -- the in-place flat-buffer weight broadcast (device memory, not host-staged), a 745 MB broadcast by size, the output
    all-gather, the max-over-ranks all-reduce, the per-rank report table of the bench JSON (an object collective) and the barrier.  Proves RCCL initialises on the box's driver stack (dmabuf IPC mode) and accepts
    our tensors; says nothing about xGMI rates (BASELINE configs[2])."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "AID_DIST_BACKEND")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _RCCL_SCRIPT, ROOT], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["backend"] == "nccl" and j["bcast_bytes"] > 0 and j["max"] == 1.25
    print(f"RCCL {j['rccl']} one-rank group on cuda:0: flat weight broadcast {j['bcast_bytes'] / 1e6:.2f} MB in place, 745 MB broadcast "
          f"{j['t_bcast_745MB_s'] * 1e3:.1f} ms, all-gather / all-reduce(MAX) / barrier ok")
