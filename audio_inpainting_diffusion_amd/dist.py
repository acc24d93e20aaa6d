"""Multi-GPU sharding of batched gap segments: one process per GPU, RCCL over xGMI.

The sampling path shards by independent units: segments never interact inside the
network (per-sample group statistics, per-sample attention) and our sampler reduces everything per item, so
rank r simply owns segments [lo_r, hi_r) with per-item RNG seeds derived from the GLOBAL segment index.
Collectives: ONE broadcast of the flat fp32 weight buffer from rank 0 at start-up (745 MB for the 22 kHz
network) and ONE all-gather of the outputs at the end -- nothing inside the sampling loop.  This replaces the
reference's dead NCCL scaffold (utils/torch_utils/distributed.py:14-31, never initialised).
Backend "nccl" is RCCL on ROCm; the same code runs under "gloo" on CPU for the unit tests.
"""
from __future__ import annotations

import os
from typing import List, Tuple

import torch
import torch.distributed as dist


def init_distributed(backend: str = None) -> Tuple[int, int, int]:
    """Initialise from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Returns
    (rank, local_rank, world_size); a no-op single-process world when WORLD_SIZE is unset or 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("AID_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition: rank r owns [lo, hi); sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def item_seeds(base_seed: int, lo: int, hi: int) -> List[int]:
    """Per-item RNG seeds from the global segment index, so results do not depend on world size."""
    return [base_seed + i for i in range(lo, hi)]


@torch.no_grad()
def broadcast_parameters(module: torch.nn.Module, src: int = 0) -> int:
    """Broadcast all parameters and buffers from ``src`` as ONE flat fp32 buffer.  Returns bytes sent."""
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
    if not tensors:
        return 0
    flat = torch.cat([t.reshape(-1).float() for t in tensors])
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat, src=src)
        off = 0
        for t in tensors:
            n = t.numel()
            t.copy_(flat[off:off + n].reshape(t.shape))
            off += n
    return flat.numel() * 4


@torch.no_grad()
def gather_outputs(local_out: torch.Tensor, n_items: int) -> torch.Tensor:
    """All-gather the per-rank outputs [n_local, L] into [n_items, L] (same on every rank)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return local_out
    world = dist.get_world_size()
    L = local_out.shape[-1]
    nmax = -(-n_items // world)
    pad = torch.zeros(nmax, L, dtype=local_out.dtype, device=local_out.device)
    pad[: local_out.shape[0]] = local_out
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_items, r, world)
        parts.append(bufs[r][: hi - lo])
    return torch.cat(parts, dim=0)


def max_over_ranks(value: float, device) -> float:
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
