"""Multi-GPU layout of a slice batch: slices are independent frames, so rank r
of P simply owns a contiguous block of slices -- no collective on the data path
(SURVEY.md section 8e).  The only exchange is the table of frame sizes."""
import torch
import torch.distributed as dist


def shard_range(n_slices: int, rank: int, world: int):
    """Contiguous block partition: rank r owns [lo, hi)."""
    base, rem = divmod(n_slices, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_frame_sizes(local_sizes: torch.Tensor, n_slices: int, group=None) -> torch.Tensor:
    """All ranks learn every frame size (4 B per slice): all_gather of equal-size
    padded blocks, then trimmed back to each rank's true count."""
    world = dist.get_world_size(group)
    per = (n_slices + world - 1) // world
    pad = torch.zeros(per, dtype=local_sizes.dtype, device=local_sizes.device)
    pad[: local_sizes.numel()] = local_sizes
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    out = []
    for r in range(world):
        lo, hi = shard_range(n_slices, r, world)
        out.append(parts[r][: hi - lo])
    return torch.cat(out)


def global_offsets(all_sizes: torch.Tensor) -> torch.Tensor:
    """Exclusive prefix sum: where each frame would sit in one dense stream."""
    s = all_sizes.to(torch.int64)
    return torch.cumsum(s, 0) - s
