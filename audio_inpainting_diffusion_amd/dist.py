"""Multi-GPU sharding of batched gap segments: one process per GPU, RCCL over xGMI.

The sampling path shards by independent units: segments never interact inside the
network (per-sample group statistics, per-sample attention) and our sampler reduces everything per item, so
rank r simply owns segments [lo_r, hi_r) with per-item RNG seeds derived from the GLOBAL segment index.
Collectives: ONE broadcast of the flat fp32 weight buffer from rank 0 at start-up (745 MB for the 22 kHz
network) and ONE all-gather of the outputs at the end -- nothing inside the sampling loop.  This replaces the
reference's dead NCCL scaffold (utils/torch_utils/distributed.py:14-31, never initialised).
Backend "nccl" is RCCL on ROCm; the same code runs under "gloo" on CPU for the unit tests, and under "gloo"
with device tensors when several ranks have to share one GPU (functional runs on a 1-GPU box: RCCL refuses
two ranks on one device) -- the collectives are then staged through host memory.

``launch_ranks`` is the self-launcher used by ``bench.py --gpus N`` when it was started as a plain ``python``
process: it re-executes the same command under ``torch.distributed.run`` (one rank per GPU, 127.0.0.1
rendezvous) and relays rank 0's output.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import List, Optional, Tuple

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
            if torch.cuda.device_count() < min(world, int(os.environ.get("LOCAL_WORLD_SIZE", world))):
                raise RuntimeError(f"RCCL needs one GPU per rank ({torch.cuda.device_count()} visible for "
                                   f"{world} ranks); set AID_DIST_BACKEND=gloo for a functional shared-GPU run")
            torch.cuda.set_device(local)
        try:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
            if backend == "nccl":
                _rccl_first_contact(local)
        except Exception as e:
            # never a silent switch to another backend: a node that has one GPU per rank and cannot bring RCCL up is a broken node, and a gloo
            # number from it would be read as an xGMI number
            raise RuntimeError(f"[aid dist] rank {rank}/{world} (local {local}): backend '{backend}' failed to initialise "
                               f"({type(e).__name__}: {e}); {torch.cuda.device_count() if torch.cuda.is_available() else 0} GPUs visible, "
                               f"HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}, rendezvous "
                               f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}.  NOT falling back to gloo: set AID_DIST_BACKEND=gloo "
                               f"explicitly for a functional (host-staged) run") from e
        dev = f"cuda:{local % max(1, torch.cuda.device_count())}" if torch.cuda.is_available() else "cpu"
        print(f"[aid dist] rank {rank}/{world} (local {local}): backend {dist.get_backend()}, device {dev}, "
              f"rendezvous {os.environ['MASTER_ADDR']}:{os.environ['MASTER_PORT']}", file=sys.stderr, flush=True)
    return rank, local, world


def _rccl_first_contact(local: int) -> None:
    """RCCL builds its communicator lazily: make the FIRST collective happen here (one 4-byte all-reduce on the rank's device + synchronize) so that a
    broken xGMI / IPC setup fails at start-up with this rank's name on it, not somewhere inside the weight broadcast."""
    t = torch.ones(1, device=torch.device("cuda", local))
    dist.all_reduce(t)
    torch.cuda.synchronize()
    if int(t.item()) != dist.get_world_size():
        raise RuntimeError(f"first RCCL all-reduce returned {float(t.item())} for world size {dist.get_world_size()}")


def placement_problems(ranks: List[dict], shared: bool, shared_requested: bool, visible_gpus: int) -> dict:
    """Sanity of a finished multi-rank run from its per-rank reports (rank_report): {"errors": [...], "warnings": [...]}.  Pure function.
    A physical GPU is identified by (host, PCI address) -- local device indices repeat across hosts -- and compared per host:
    two ranks of one host on one GPU is an error when that host had a GPU for each of them and nobody asked for sharing (AID_SHARED_GPU);
    a requested shared-GPU functional run on a box that could have given every rank its own GPU is a warning, not an abort."""
    out = {"errors": [], "warnings": []}
    if len(ranks) < 2:
        return out
    by_host: dict = {}
    for r in ranks:
        by_host.setdefault(r.get("host"), []).append(r)
    for host, rs in by_host.items():
        ids = [(r.get("pci") or r.get("gpu")) for r in rs]
        dup = len(set(ids)) < len(rs)
        enough = len(rs) <= visible_gpus
        if shared and enough:
            (out["warnings"] if shared_requested else out["errors"]).append(
                f"host {host}: ranks share GPUs although {visible_gpus} are visible for {len(rs)} ranks"
                + (" (AID_SHARED_GPU set: functional run)" if shared_requested else " (AID_DIST_BACKEND=gloo left set?)"))
        elif dup and enough and not shared:
            out["errors"].append(f"host {host}: two ranks report the same GPU: %r" % [(r["rank"], r.get("gpu"), r.get("pci")) for r in rs])
    return out


def _cpulist(text: str) -> List[int]:
    out: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def gpu_pci_bdf(device_index: int) -> Optional[str]:
    """PCI address "dddd:bb:dd.f" of a visible GPU (torch device properties), or None."""
    try:
        pr = torch.cuda.get_device_properties(device_index)
        return "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
    except Exception:
        return None


def gpu_numa_node(device_index: int, sysfs: str = "/sys", bdf_of=gpu_pci_bdf) -> Optional[int]:
    """NUMA node of a GPU from sysfs (<sysfs>/bus/pci/devices/<domain:bus:dev.fn>/numa_node), or None when it cannot be read / is -1.
    ``sysfs`` / ``bdf_of`` are injectable so that the lookup can be tested against a faked tree (tests/test_dist_gloo.py)."""
    try:
        bdf = bdf_of(device_index)
        node = int(open(os.path.join(sysfs, "bus/pci/devices", bdf, "numa_node")).read())
        return node if node >= 0 else None
    except Exception:
        return None


def plan_rank_binding(local: int, ranks_on_node: int, ndev: int, avail: List[int], sysfs: str = "/sys", bdf_of=gpu_pci_bdf) -> dict:
    """Which CPUs local rank ``local`` should run on: the cores of its GPU's NUMA node, shared evenly between the ranks whose GPUs sit on the same
    node; an even split of ``avail`` when sysfs does not say.  Pure function of its arguments (no affinity call): returns
    {"gpu", "pci", "numa_node", "cpus", "peers", "fallback"}; ``fallback`` names why the NUMA path was not taken (None when it was)."""
    avail = sorted(avail)
    gpu = local % ndev if ndev else None
    out = {"gpu": gpu, "pci": bdf_of(gpu) if ndev else None, "numa_node": None, "peers": list(range(ranks_on_node)), "fallback": None}
    mine = avail
    node = gpu_numa_node(gpu, sysfs, bdf_of) if ndev else None
    if node is None:
        out["fallback"] = "no GPU" if not ndev else "numa_node of the GPU not readable (or -1)"
    else:
        out["numa_node"] = node
        try:
            cpus = [c for c in _cpulist(open(os.path.join(sysfs, f"devices/system/node/node{node}/cpulist")).read()) if c in set(avail)]
            same = [r for r in range(ranks_on_node) if gpu_numa_node(r % ndev, sysfs, bdf_of) == node]
            if cpus and local in same:
                mine, out["peers"] = cpus, same
            else:
                out["fallback"] = "none of the node's CPUs is in this process's affinity mask"
        except OSError:
            out["fallback"] = "cpulist of the node not readable"
    peers = out["peers"]
    k, n = peers.index(local) if local in peers else 0, max(1, len(peers))
    out["cpus"] = mine[k * len(mine) // n:(k + 1) * len(mine) // n] or mine
    return out


_BINDING: dict = {}


def bind_rank_to_gpu_numa(local: int, ranks_on_node: int, log=None) -> Optional[int]:
    """Pin this rank's CPU threads (noise generation, launch loop) to the cores of its GPU's NUMA node (plan_rank_binding).  Returns the thread
    count set; the plan that was applied is kept for rank_report()."""
    try:
        avail = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return None
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    plan = plan_rank_binding(local, ranks_on_node, ndev, avail)
    share = plan["cpus"]
    try:
        os.sched_setaffinity(0, share)
    except OSError:
        return None
    torch.set_num_threads(max(1, len(share)))
    _BINDING.clear()
    _BINDING.update(plan)
    if log is not None:
        print(f"[aid dist] local rank {local}: GPU {plan['gpu']} ({plan['pci']}) NUMA node {plan['numa_node']}, {len(share)} CPU threads "
              f"({share[0]}..{share[-1]})" + (f" -- FALLBACK: {plan['fallback']}" if plan["fallback"] else ""), file=log, flush=True)
    return len(share)


def rccl_version() -> Optional[str]:
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception:
        return None


def rank_report(rank: int, local: int, **timings) -> dict:
    """One rank's line of the bench JSON's ``ranks`` table: where it ran (GPU, PCI address, NUMA node, CPU threads, binding fallback if any) and
    its own wall-clock numbers -- enough to tell a mis-bound or slow rank from a slow collective without a rerun."""
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    gpu = torch.cuda.current_device() if ndev else None
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count()
    r = {"rank": rank, "local_rank": local, "host": socket.gethostname(), "gpu": gpu, "pci": gpu_pci_bdf(gpu) if ndev else None,
         "gpu_name": torch.cuda.get_device_properties(gpu).name if ndev else None,
         "numa_node": _BINDING.get("numa_node", gpu_numa_node(gpu) if ndev else None), "cpu_threads": ncpu,
         "numa_peers": _BINDING.get("peers"), "binding_fallback": _BINDING.get("fallback", "bind_rank_to_gpu_numa not called")}
    r.update({k: (round(v, 4) if isinstance(v, float) else v) for k, v in timings.items()})
    return r


def gather_objects(obj, dst: int = 0):
    """Python objects of every rank on rank ``dst`` (list in rank order; None elsewhere); [obj] in a single-process world."""
    if not _active():
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)                     # (all_gather_object: the object collective every backend implements; RCCL stages through the current device)
    return out if dist.get_rank() == dst else None


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition: rank r owns [lo, hi); sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def item_seeds(base_seed: int, lo: int, hi: int) -> List[int]:
    """Per-item RNG seeds from the global segment index, so results do not depend on world size."""
    return [base_seed + i for i in range(lo, hi)]


def _active() -> bool:
    """Collectives run whenever a process group with more than one rank exists -- and also in a ONE-rank group on the RCCL backend, which nothing in
    this package creates (init_distributed is a no-op at world size 1): a caller that built one gets the same device-memory collectives the
    8-GPU run uses (tests/test_gpu_dist.py::test_rccl_first_contact_world_size_one, the only RCCL contact a 1-GPU box allows)."""
    return dist.is_initialized() and (dist.get_world_size() > 1 or dist.get_backend() == "nccl")


def _host_staged(t: torch.Tensor) -> bool:
    """gloo cannot run every collective on device memory: stage device tensors through the host for it."""
    return t.is_cuda and dist.get_backend() == "gloo"


@torch.no_grad()
def flatten_parameters_(module: torch.nn.Module) -> torch.Tensor:
    """Re-home every fp32 parameter and buffer of ``module`` as a view into ONE flat buffer (idempotent) and return
    that buffer.  The weight broadcast is then a single zero-copy collective on it -- no second 745 MB staging
    copy -- and in-place loads (``load_state_dict`` copies into ``.data``) keep the views intact."""
    tensors = [p for p in module.parameters()] + [b for b in module.buffers()]
    tensors = [t for t in tensors if t.dtype == torch.float32]
    flat = getattr(module, "_aid_flat", None)
    if flat is not None and len(tensors) and all(
            t.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for t in tensors):
        return flat
    total = sum(t.numel() for t in tensors)
    dev = tensors[0].device if tensors else torch.device("cpu")
    flat = torch.empty(total, dtype=torch.float32, device=dev)
    off = 0
    for t in tensors:
        n = t.numel()
        view = flat[off:off + n].view(t.shape)
        view.copy_(t.data)
        t.data = view                       # frees the old storage tensor by tensor: peak extra = one tensor
        off += n
    module._aid_flat = flat
    if hasattr(module, "_packed_ver"):      # kernel-side weight packs must be refreshed (network.prepare)
        module._packed_ver = None
    return flat


@torch.no_grad()
def broadcast_parameters(module: torch.nn.Module, src: int = 0) -> int:
    """Broadcast all fp32 parameters and buffers from ``src`` as ONE flat buffer, in place.  Returns bytes sent."""
    flat = flatten_parameters_(module)
    if flat.numel() == 0:
        return 0
    if _active():
        if _host_staged(flat):
            host = flat.cpu()
            dist.broadcast(host, src=src)
            flat.copy_(host)
        else:
            dist.broadcast(flat, src=src)
        if hasattr(module, "_packed_ver"):
            module._packed_ver = None
    return flat.numel() * 4


@torch.no_grad()
def gather_outputs(local_out: torch.Tensor, n_items: int) -> torch.Tensor:
    """All-gather the per-rank outputs [n_local, L] into [n_items, L] (same on every rank)."""
    if not _active():
        return local_out
    world = dist.get_world_size()
    L = local_out.shape[-1]
    nmax = -(-n_items // world)
    staged = _host_staged(local_out)
    work_dev = torch.device("cpu") if staged else local_out.device
    pad = torch.zeros(nmax, L, dtype=local_out.dtype, device=work_dev)
    pad[: local_out.shape[0]] = local_out.to(work_dev)
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_items, r, world)
        parts.append(bufs[r][: hi - lo])
    return torch.cat(parts, dim=0).to(local_out.device)


def max_over_ranks(value: float, device) -> float:
    if not _active():
        return value
    dev = torch.device("cpu") if dist.get_backend() == "gloo" else device
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if _active():
        dist.barrier()


# ---------------------------------------------------------------------------------------------------------
# self-launch (bench.py --gpus N started as a plain python process)
# ---------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n: int, argv: Optional[List[str]] = None, timeout: Optional[float] = None) -> int:
    """Re-execute ``argv`` (default: this process's command line) as ``n`` ranks under ``torch.distributed.run``
    on this node, 127.0.0.1 rendezvous, one rank per GPU.  When fewer than ``n`` GPUs are visible the ranks share
    them round-robin and the collectives run on gloo (functional mode, labelled as such by the caller).
    Returns the launcher's exit code; stdout/stderr of the ranks pass through."""
    argv = list(sys.argv if argv is None else argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev < n:
        env["AID_DIST_BACKEND"] = "gloo"
        env["AID_SHARED_GPU"] = str(max(1, ndev))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + argv
    return subprocess.run(cmd, env=env, timeout=timeout).returncode
