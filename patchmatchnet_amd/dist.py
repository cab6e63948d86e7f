"""One process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" for the CPU tests).

The hot path needs no collective: reference views are independent units and are sharded round-robin across ranks
(SURVEY.md 8(e)).  The only exchange is the per-scan all-gather of the finished depth / confidence maps, needed because
fusing reference view r reads the maps of its source views, which other ranks produced (reference eval.py:236-237).
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
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl" if device_type == "cuda" else "gloo", rank=rank, world_size=world)
    return rank, world, device


def shard_views(view_ids: List[int], rank: int, world: int) -> List[int]:
    """Round-robin ownership of reference views: rank r owns view_ids[r::world]."""
    return view_ids[rank::world]


def gather_scan_maps(local: Dict[int, torch.Tensor], view_ids: List[int], H: int, W: int, device: torch.device
                     ) -> Dict[int, torch.Tensor]:
    """All-gather the [2,H,W] (depth, confidence) maps of one scan.

    ``local`` holds this rank's maps keyed by view id; ``view_ids`` is the scan's full (ordered) list, owned round-robin
    as in ``shard_views``.  Every rank contributes ceil(len/world) slots (padding slots are zero); one
    ``all_gather_into_tensor`` moves them -- ~15 MB per 1600x1200 view, i.e. 107.5 MB per rank for a 49-view DTU scan
    on 8 GPUs.  Returns {view id: [2,H,W]} for ALL views of the scan on every rank."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    mine = shard_views(view_ids, rank, world)
    slots = (len(view_ids) + world - 1) // world
    send = torch.zeros((slots, 2, H, W), dtype=torch.float32, device=device)
    for i, vid in enumerate(mine):
        send[i] = local[vid].to(device=device, dtype=torch.float32)
    if world == 1:
        recv = send[None]
    else:
        recv = torch.empty((world, slots, 2, H, W), dtype=torch.float32, device=device)
        dist.all_gather_into_tensor(recv.view(world * slots, 2, H, W), send)
    out = {}
    for r in range(world):
        for i, vid in enumerate(shard_views(view_ids, r, world)):
            out[vid] = recv[r, i]
    return out
