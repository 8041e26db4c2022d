"""Multi-GPU layout of a slice batch: slices are independent frames (one compression context per stream in the
reference: kompressor-zstd--nativelib/src/jvmCommonMain/kotlin/com/ensody/kompressor/zstd/ZstdCompressor.jvm.kt:14),
so rank r of P simply owns a contiguous block of slices -- no collective on the data path.

What does move, when the batch starts and ends on one rank, is buffers:
  * scatter_slices   root -> ranks, each rank's block of input slices (point-to-point sends: on xGMI all seven links of
                     the root carry one block each);
  * gather_frame_sizes  the u32 size of every frame (all_gather, 4 B per slice);
  * gather_frames    ranks -> root, each rank's densely packed frames at the offsets the size table gives (gather-v).
The same calls run over RCCL (device tensors, backend "nccl") and over gloo (host tensors; the CPU tests)."""
import torch
import torch.distributed as dist

_PIECE = 1 << 30          # bytes per point-to-point message


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


def _p2p(items, group=None):
    """items: ("send" | "recv", tensor, peer).  Over gloo device tensors go through host copies (the one-GPU rehearsal
    of the N > 1 control flow); over RCCL they are sent as they are."""
    if not items:
        return
    staged = dist.get_backend(group) == "gloo"
    ops, back = [], []
    for kind, t, peer in items:
        if staged and t.is_cuda:
            h = t.cpu() if kind == "send" else torch.empty(t.shape, dtype=t.dtype)
            if kind == "recv":
                back.append((t, h))
            t = h
        ops.append(dist.P2POp(dist.isend if kind == "send" else dist.irecv, t, peer, group))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    for t, h in back:
        t.copy_(h)


def _pieces(t: torch.Tensor):
    for o in range(0, t.numel(), _PIECE):
        yield t[o:o + _PIECE]


def scatter_slices(all_slices, local_out: torch.Tensor, n_slices: int, slice_bytes: int, root: int = 0, group=None) -> torch.Tensor:
    """Root holds the whole batch densely (n_slices x slice_bytes, uint8); every rank receives its shard_range block
    into local_out (uint8, at least its block's bytes).  all_slices is ignored on the other ranks.  Returns the view of
    local_out that holds the block."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_range(n_slices, rank, world)
    mine = local_out[: (hi - lo) * slice_bytes]
    ops = []
    if rank == root:
        for r in range(world):
            rlo, rhi = shard_range(n_slices, r, world)
            block = all_slices[rlo * slice_bytes: rhi * slice_bytes]
            if r == root:
                mine.copy_(block)
            else:
                ops += [("send", p, r) for p in _pieces(block)]
    else:
        ops = [("recv", p, root) for p in _pieces(mine)]
    _p2p(ops, group)
    return mine


def gather_slices(local_block: torch.Tensor, n_slices: int, slice_bytes: int, root: int = 0, group=None, out=None):
    """The inverse of scatter_slices: every rank's block of equal-size slices lands on the root in slice order
    (what a batch of decoded slices, or a batch generated rank-locally, looks like when one rank wants all of it).
    Returns the dense batch on the root, None elsewhere."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_range(n_slices, rank, world)
    mine = local_block[: (hi - lo) * slice_bytes]
    ops, everything = [], None
    if rank == root:
        total = n_slices * slice_bytes
        everything = out[:total] if (out is not None and out.numel() >= total) else torch.empty(total, dtype=torch.uint8, device=local_block.device)
        for r in range(world):
            rlo, rhi = shard_range(n_slices, r, world)
            block = everything[rlo * slice_bytes: rhi * slice_bytes]
            if r == root:
                block.copy_(mine)
            else:
                ops += [("recv", p, r) for p in _pieces(block)]
    else:
        ops = [("send", p, root) for p in _pieces(mine)]
    _p2p(ops, group)
    return everything


def gather_frames(local_dense: torch.Tensor, local_sizes: torch.Tensor, n_slices: int, root: int = 0, group=None, out=None):
    """Gather-v of the payload: local_dense holds this rank's frames back to back (sum(local_sizes) bytes).  Every rank
    gets (all_sizes, offsets); the root also gets the dense stream of all frames in slice order (frame i at
    stream[offsets[i] : offsets[i] + all_sizes[i]]), written into `out` when that is large enough.  Returns
    (stream or None, all_sizes, offsets)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    all_sizes = gather_frame_sizes(local_sizes, n_slices, group)
    offsets = global_offsets(all_sizes)
    # bytes per rank: P numbers, read back by every rank (the one host round trip of the exchange)
    csum = torch.cat([torch.zeros(1, dtype=torch.int64, device=all_sizes.device), torch.cumsum(all_sizes.to(torch.int64), 0)])
    cuts = [shard_range(n_slices, r, world)[0] for r in range(world)] + [n_slices]
    at = csum[torch.tensor(cuts, dtype=torch.int64, device=all_sizes.device)].tolist()
    starts, ends = at[:-1], at[1:]
    total = ends[-1]
    my_bytes = ends[rank] - starts[rank]
    ops = []
    stream = None
    if rank == root:
        stream = out if (out is not None and out.numel() >= total) else torch.empty(total, dtype=torch.uint8, device=local_dense.device)
        stream = stream[:total]
        for r in range(world):
            seg = stream[starts[r]: ends[r]]
            if r == root:
                seg.copy_(local_dense[:my_bytes])
            else:
                ops += [("recv", p, r) for p in _pieces(seg)]
    else:
        ops = [("send", p, root) for p in _pieces(local_dense[:my_bytes])]
    _p2p(ops, group)
    return stream, all_sizes, offsets
