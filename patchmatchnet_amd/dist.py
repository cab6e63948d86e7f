"""One process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" for the CPU tests).

The hot path needs no collective: reference views are independent units; every (scan, light) group is cut into ``world``
CONTIGUOUS blocks of reference views (neighbouring reference views share most of their source views, so a
block keeps a rank's per-scan feature cache effective where a round-robin deal makes every rank encode nearly the whole scan).
The only exchange is the per-scan all-gather of the finished depth / confidence maps, needed because fusing reference view r
reads the maps of its source views, which other ranks produced (reference eval.py:236-237); after it every rank fuses its own
block of reference views.
"""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import torch
import torch.distributed as dist


def init_from_env(device_type: str = "cuda") -> Tuple[int, int, torch.device]:
    """Reads RANK / WORLD_SIZE / LOCAL_RANK (torchrun contract).  Returns (rank, world_size, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if device_type == "cuda":
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    else:
        device = torch.device("cpu")
    # launched by torch.distributed.run (WORLD_SIZE set, also for a single rank): the process group exists and every collective
    # below runs through it -- one rank over RCCL exercises the same library path as eight
    if "WORLD_SIZE" in os.environ and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            # a launcher sets the port; without one, several ranks cannot agree on a port by themselves (fail instead of colliding
            # with another job on a fixed default), and a single rank takes a free ephemeral one
            if world > 1:
                raise RuntimeError("WORLD_SIZE > 1 without MASTER_PORT: launch with torch.distributed.run (or set MASTER_PORT)")
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        # PMN_DIST_BACKEND=gloo lets several ranks share ONE GPU (tests/test_eval_gpu.py: RCCL needs a device per rank)
        backend = os.environ.get("PMN_DIST_BACKEND", "nccl" if device_type == "cuda" else "gloo")
        if backend == "nccl":
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, device


def bind_to_device_node(device: torch.device) -> str:
    """Pins this process (and every thread it starts afterwards) to the CPUs of the NUMA node its GPU hangs off.  One process per GPU
    on a two-socket host: the launch thread, the decode / writer pools and the pinned staging buffers (first touch) otherwise land on
    either socket by chance -- measured on the 2 x 64-core bench box as whole eval.py runs alternating between 270 and 380
    depth-maps/s (profiles/r04_eval_bench.md).  Returns a one-line description; never raises (no sysfs entry, no permission: no-op).
    PMN_NUMA_BIND=0 disables it."""
    if os.environ.get("PMN_NUMA_BIND", "1") == "0" or device.type != "cuda" or not hasattr(os, "sched_setaffinity"):
        return "numa binding off"
    try:
        p = torch.cuda.get_device_properties(device)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        base = "/sys/bus/pci/devices/" + bdf
        node = int(open(base + "/numa_node").read())
        cpus: List[int] = []
        for part in open(base + "/local_cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.extend(range(int(lo), int(hi or lo) + 1))
        allowed = sorted(set(cpus) & os.sched_getaffinity(0))
        if node < 0 or not allowed:
            return f"numa binding: {bdf} reports node {node}, nothing to bind to"
        os.sched_setaffinity(0, allowed)
        return f"bound to NUMA node {node} of {bdf}: {len(allowed)} cpus ({allowed[0]}..{allowed[-1]})"
    except (OSError, ValueError, AttributeError, RuntimeError) as e:
        return f"numa binding skipped ({type(e).__name__}: {e})"


def block_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, stop) of rank ``rank``'s block when n items are cut into ``world`` contiguous, near-equal blocks (the first
    n % world blocks are one longer)."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_views(view_ids: List[int], rank: int, world: int) -> List[int]:
    """Block ownership of one scan's reference views (in pair-file order): rank r owns view_ids[block_range(len, r, world)].
    MVSDataset.shard applies the same rule per (scan, light) group, so the two always agree."""
    a, b = block_range(len(view_ids), rank, world)
    return view_ids[a:b]


def gather_scan_buffer(local: Dict[int, torch.Tensor], view_ids: List[int], H: int, W: int, device: torch.device,
                       flat: bool = False) -> Tuple[torch.Tensor, Dict[int, int]]:
    """All-gather the [2,h,w] (depth, confidence) maps of one scan into ONE buffer.

    ``local`` holds this rank's maps keyed by view id; ``view_ids`` is the scan's full (ordered) list, owned in blocks as in
    ``shard_views``.  Every rank contributes ceil(len/world) slots (padding slots are zero); one ``all_gather_into_tensor``
    moves them -- ~15 MB per 1600x1200 view, i.e. 107.5 MB per rank for a 49-view DTU scan on 8 GPUs.  Returns
    (buffer [world * slots, 2, H, W], {view id: slot}) on every rank: what pmn_fuse_view consumes directly.
    ``flat`` (a scan whose views differ in size; H x W = the LARGEST view): slots are [2*H*W] floats and every view's maps are
    packed at the start of its slot at their own size -> buffer [world * slots, 2*H*W]."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    mine = shard_views(view_ids, rank, world)
    slots = (len(view_ids) + world - 1) // world
    shape = (2 * H * W,) if flat else (2, H, W)
    send = torch.zeros((slots,) + shape, dtype=torch.float32, device=device)
    for i, vid in enumerate(mine):
        m = local[vid].to(device=device, dtype=torch.float32)
        if flat:
            send[i, :m.numel()] = m.reshape(-1)
        else:
            send[i] = m
    if not dist.is_initialized():
        recv = send
    elif dist.get_backend() == "gloo" and send.is_cuda:  # gloo moves host memory: stage through it
        host = torch.empty((world * slots,) + shape, dtype=torch.float32)
        dist.all_gather_into_tensor(host, send.cpu())
        recv = host.to(device)
    else:
        recv = torch.empty((world * slots,) + shape, dtype=torch.float32, device=device)
        dist.all_gather_into_tensor(recv, send)
    slot_of = {}
    for r in range(world):
        for i, vid in enumerate(shard_views(view_ids, r, world)):
            slot_of[vid] = r * slots + i
    return recv, slot_of


def gather_scan_maps(local: Dict[int, torch.Tensor], view_ids: List[int], H: int, W: int, device: torch.device
                     ) -> Dict[int, torch.Tensor]:
    """``gather_scan_buffer`` as {view id: [2,H,W]} for ALL views of the scan on every rank."""
    buf, slot_of = gather_scan_buffer(local, view_ids, H, W, device)
    return {vid: buf[s] for vid, s in slot_of.items()}
