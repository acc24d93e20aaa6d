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
