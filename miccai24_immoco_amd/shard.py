"""Multi-GPU slice sharding.  Slices are independent, so the only
communication is ONE gather of the final images (RCCL over xGMI; `gloo` in the
CPU tests); there is no collective inside the optimisation loop.  One process
per GPU; contiguous blocks of slices per rank."""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [start, end) of rank `rank`; sizes differ by at most one."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def gather_images(local: torch.Tensor, n_total: int, group=None, dst: Optional[int] = None) -> Optional[torch.Tensor]:
    """local [n_local, H, W] complex64 of this rank's block -> [n_total, H, W] on every rank
    (dst=None, all_gather) or on rank `dst` only (gather).  Ragged blocks are padded to the
    largest block for the collective and trimmed afterwards."""
    if not dist.is_initialized():
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    H, W = local.shape[-2:]
    n_max = max(shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world))
    buf = torch.zeros((n_max, H, W, 2), device=local.device, dtype=torch.float32)
    if local.shape[0]:
        buf[: local.shape[0]] = torch.view_as_real(local.contiguous())
    if dst is None:
        out = torch.empty((world * n_max, H, W, 2), device=local.device, dtype=torch.float32)
        dist.all_gather_into_tensor(out, buf, group=group)
        out = out.view(world, n_max, H, W, 2)
    else:
        outs = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
        dist.gather(buf, outs, dst=dst, group=group)
        if rank != dst:
            return None
        out = torch.stack(outs)
    parts = []
    for r in range(world):
        a, b = shard_range(n_total, r, world)
        parts.append(out[r, : b - a])
    return torch.view_as_complex(torch.cat(parts).contiguous())


def solve_sharded(n_slices: int, solve_fn: Callable[[int], torch.Tensor], group=None,
                  dst: Optional[int] = None) -> Optional[torch.Tensor]:
    """Every rank solves its contiguous block with `solve_fn(global_slice_index) -> [H, W] complex64`
    and the final images are gathered once."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    # decided identically on EVERY rank before anything is solved: a rank with an empty block would not know
    # the image shape for the collective, and raising on that rank alone would leave the others waiting in
    # all_gather until the RCCL timeout
    if n_slices < world:
        raise ValueError(f"solve_sharded: {n_slices} slices for {world} ranks - every rank needs at least one slice")
    a, b = shard_range(n_slices, rank, world)
    imgs: List[torch.Tensor] = [solve_fn(i) for i in range(a, b)]
    local = torch.stack(imgs)
    return gather_images(local, n_slices, group=group, dst=dst)
