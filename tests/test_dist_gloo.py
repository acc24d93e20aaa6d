"""World-size-2 `gloo` test (CPU) of the multi-GPU path: shard ranges, per-item seeds, flat weight broadcast,
output all-gather.  The sampling loop itself has no collective."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from audio_inpainting_diffusion_amd import dist as D
    from audio_inpainting_diffusion_amd.config import small_args
    from audio_inpainting_diffusion_amd.init import seeded_init_
    from audio_inpainting_diffusion_amd.network import Unet_CQT_oct_with_attention
    r, _, w = D.init_distributed("gloo")
    assert (r, w) == (rank, world)
    net = Unet_CQT_oct_with_attention(small_args(), torch.device("cpu"))
    seeded_init_(net, 100 + rank)                     # ranks start with DIFFERENT weights
    nbytes = D.broadcast_parameters(net, src=0)
    ref = seeded_init_(Unet_CQT_oct_with_attention(small_args(), torch.device("cpu")), 100)
    same = all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), ref.state_dict().values()))
    n_items, L = 5, 16
    lo, hi = D.shard_range(n_items, rank, world)
    seeds = D.item_seeds(1000, lo, hi)
    local = torch.stack([torch.full((L,), float(s)) for s in seeds]) if seeds else torch.zeros(0, L)
    allout = D.gather_outputs(local, n_items)
    t = D.max_over_ranks(1.0 + rank, torch.device("cpu"))
    reps = D.gather_objects(D.rank_report(rank, rank, wall_s=1.0 + rank, segments=[lo, hi]))      # bench.py's per-rank table: rank order on rank 0, None elsewhere
    assert (reps is None) if rank else ([x["rank"] for x in reps] == [0, 1] and [x["wall_s"] for x in reps] == [1.0, 2.0] and reps[1]["segments"] == [3, 5])
    D.barrier()
    q.put((rank, same, nbytes, (lo, hi), allout[:, 0].tolist(), t))
    dist.destroy_process_group()


def test_world_size_2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[3] for r in res] == [(0, 3), (3, 5)]
    for rank, same, nbytes, rng, col, t in res:
        assert same, "weights differ from rank 0 after the broadcast"
        assert nbytes > 0
        assert col == [1000.0, 1001.0, 1002.0, 1003.0, 1004.0]       # gathered in global segment order on every rank
        assert t == 2.0


def test_shard_ranges_cover_and_are_world_size_independent():
    from audio_inpainting_diffusion_amd.dist import item_seeds, shard_range
    for n in (1, 7, 8, 64):
        for w in (1, 2, 4, 8):
            rs = [shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n and all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            seeds = sum((item_seeds(5, lo, hi) for lo, hi in rs), [])
            assert seeds == list(range(5, 5 + n))


def _fake_sysfs(root, gpus, nodes):
    """gpus: [(bdf, numa_node)]; nodes: {node: cpulist text}"""
    for bdf, node in gpus:
        d = os.path.join(root, "bus/pci/devices", bdf)
        os.makedirs(d)
        with open(os.path.join(d, "numa_node"), "w") as f:
            f.write(f"{node}\n")
    for node, cpulist in nodes.items():
        d = os.path.join(root, f"devices/system/node/node{node}")
        os.makedirs(d)
        with open(os.path.join(d, "cpulist"), "w") as f:
            f.write(cpulist + "\n")


def test_rank_binding_on_a_faked_8_gpu_2_node_tree(tmp_path):
    """gpu_numa_node / plan_rank_binding against a faked sysfs tree shaped like an 8 x MI355X host (2 sockets x 64 cores with SMT, 4 GPUs per socket):
    every rank lands on its own GPU's node, the four ranks of a node split its CPUs evenly and disjointly; the fallbacks say why they were taken."""
    from audio_inpainting_diffusion_amd import dist as D
    bdfs = ["0000:%02x:00.0" % b for b in (0x05, 0x15, 0x65, 0x75, 0x85, 0x95, 0xe5, 0xf5)]
    _fake_sysfs(str(tmp_path), [(bdfs[i], 0 if i < 4 else 1) for i in range(8)], {0: "0-63,128-191", 1: "64-127,192-255"})
    bdf_of = lambda i: bdfs[i]
    assert [D.gpu_numa_node(i, str(tmp_path), bdf_of) for i in range(8)] == [0] * 4 + [1] * 4
    assert D._cpulist("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11]
    avail = list(range(256))
    plans = [D.plan_rank_binding(r, 8, 8, avail, str(tmp_path), bdf_of) for r in range(8)]
    node_cpus = {0: set(range(0, 64)) | set(range(128, 192)), 1: set(range(64, 128)) | set(range(192, 256))}
    for r, pl in enumerate(plans):
        assert pl["gpu"] == r and pl["pci"] == bdfs[r] and pl["numa_node"] == (0 if r < 4 else 1) and pl["fallback"] is None
        assert pl["peers"] == ([0, 1, 2, 3] if r < 4 else [4, 5, 6, 7])
        assert len(pl["cpus"]) == 32 and set(pl["cpus"]) <= node_cpus[pl["numa_node"]]
    for node in (0, 1):
        got = [c for pl in plans if pl["numa_node"] == node for c in pl["cpus"]]
        assert len(got) == len(set(got)) == 128 and set(got) == node_cpus[node]           # disjoint, and together the whole node
    # an affinity mask that excludes node 1's CPUs (a container pinned to socket 0): ranks 4-7 fall back to an even split and say so
    half = sorted(node_cpus[0])
    pl = D.plan_rank_binding(5, 8, 8, half, str(tmp_path), bdf_of)
    assert pl["numa_node"] == 1 and pl["fallback"] and len(pl["cpus"]) == 16 and set(pl["cpus"]) <= node_cpus[0]
    # numa_node = -1 (not reported) and an unknown device: even split of what is visible
    _fake_sysfs(str(tmp_path / "b"), [("0000:01:00.0", -1)], {})
    pl = D.plan_rank_binding(1, 2, 1, list(range(16)), str(tmp_path / "b"), lambda i: "0000:01:00.0")
    assert pl["numa_node"] is None and pl["fallback"] and pl["cpus"] == list(range(8, 16)) and pl["gpu"] == 0
    pl = D.plan_rank_binding(0, 2, 0, list(range(16)), str(tmp_path / "b"), bdf_of)
    assert pl["gpu"] is None and pl["fallback"] == "no GPU" and pl["cpus"] == list(range(8))
    # more ranks than GPUs (functional shared-GPU run): ranks wrap around the devices
    assert D.plan_rank_binding(9, 16, 8, avail, str(tmp_path), bdf_of)["gpu"] == 1
    rep = D.rank_report(0, 0, wall_s=1.23456789, bcast_s=0.5)
    assert rep["rank"] == 0 and rep["wall_s"] == 1.2346 and "binding_fallback" in rep and "cpu_threads" in rep
    assert D.gather_objects({"a": 1}) == [{"a": 1}]


def test_eight_gpus_visible_and_rccl_init_raises_is_loud_not_gloo(monkeypatch):
    """on a node that shows one GPU per rank, a failing RCCL bring-up must abort with a diagnostic -- never continue on gloo.
    Both failure points: init_process_group itself, and the first collective (RCCL creates its communicator lazily)."""
    import pytest
    from audio_inpainting_diffusion_amd import dist as D
    calls = []
    monkeypatch.setenv("WORLD_SIZE", "8"); monkeypatch.setenv("RANK", "3"); monkeypatch.setenv("LOCAL_RANK", "3"); monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    monkeypatch.delenv("AID_DIST_BACKEND", raising=False)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: calls.append(("set_device", d)))
    monkeypatch.setattr(dist, "is_initialized", lambda: False)

    def init_raises(backend=None, **kw):
        calls.append(("init", backend))
        raise dist.DistBackendError("ncclUnhandledCudaError: hipIpcGetMemHandle: invalid argument")
    monkeypatch.setattr(dist, "init_process_group", init_raises)
    with pytest.raises(RuntimeError, match="NOT falling back to gloo") as ei:
        D.init_distributed()
    assert "rank 3/8" in str(ei.value) and "8 GPUs visible" in str(ei.value) and "hipIpcGetMemHandle" in str(ei.value)
    assert calls == [("set_device", 3), ("init", "nccl")]            # one attempt, on RCCL, nothing after it

    calls.clear()
    monkeypatch.setattr(dist, "init_process_group", lambda backend=None, **kw: calls.append(("init", backend)))

    def contact_raises(local):
        calls.append(("first_contact", local))
        raise RuntimeError("NCCL error: unhandled system error (xGMI link down)")
    monkeypatch.setattr(D, "_rccl_first_contact", contact_raises)
    with pytest.raises(RuntimeError, match="xGMI link down"):
        D.init_distributed()
    assert calls == [("set_device", 3), ("init", "nccl"), ("first_contact", 3)]

    # fewer GPUs than ranks and no explicit backend: refused before any process group exists (launch_ranks sets AID_DIST_BACKEND=gloo itself)
    calls.clear()
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(RuntimeError, match="one GPU per rank"):
        D.init_distributed()
    assert calls == []


def test_placement_problems_keys_gpus_by_host_and_pci():
    from audio_inpainting_diffusion_amd.dist import placement_problems
    rep = lambda rank, host, gpu, pci: {"rank": rank, "host": host, "gpu": gpu, "pci": pci}
    eight = [rep(r, "n0", r, "0000:%02x:00.0" % (5 + 16 * r)) for r in range(8)]
    assert placement_problems(eight, False, False, 8) == {"errors": [], "warnings": []}
    # two nodes x four ranks on 8-GPU hosts: local indices repeat across hosts -- legitimate
    two_nodes = [rep(r, "n%d" % (r // 4), r % 4, "0000:%02x:00.0" % (5 + 16 * (r % 4))) for r in range(8)]
    assert placement_problems(two_nodes, False, False, 8) == {"errors": [], "warnings": []}
    # two ranks of ONE host on one GPU while the host has a GPU each: error
    bad = [dict(r) for r in eight]
    bad[5]["gpu"], bad[5]["pci"] = bad[4]["gpu"], bad[4]["pci"]
    assert placement_problems(bad, False, False, 8)["errors"]
    # a deliberately shared functional run on a multi-GPU box: warning only; the same with nobody having asked for it: error
    sh = [rep(r, "n0", 0, "0000:05:00.0") for r in range(2)]
    assert placement_problems(sh, True, True, 8) == {"errors": [], "warnings": [placement_problems(sh, True, True, 8)["warnings"][0]]}
    assert placement_problems(sh, True, False, 8)["errors"]
    # a 1-GPU box running 8 functional ranks: nothing to complain about
    assert placement_problems([rep(r, "n0", 0, "0000:05:00.0") for r in range(8)], True, True, 1) == {"errors": [], "warnings": []}
    assert placement_problems(eight[:1], False, False, 8) == {"errors": [], "warnings": []}
