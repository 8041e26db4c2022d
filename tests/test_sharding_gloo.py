"""N>1 path on CPU: two gloo ranks shard a slice batch, each produces its
frames' sizes (from the golden manifest, standing in for the device step) and
the size table is all-gathered; the result must equal the single-rank table."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers
from kompressor_amd import sharding


def _worker(rank, world, port, n, sizes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharding.shard_range(n, rank, world)
    local = torch.tensor(sizes[lo:hi], dtype=torch.int32)
    allsz = sharding.gather_frame_sizes(local, n)
    offs = sharding.global_offsets(allsz)
    total = torch.tensor([int(local.to(torch.int64).sum())])
    dist.all_reduce(total)
    q.put((rank, allsz.tolist(), offs[-1].item() + allsz[-1].item(), int(total.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_cover_everything():
    for n in [0, 1, 7, 8, 65536, 1048576, 1000003]:
        for w in [1, 2, 3, 4, 8]:
            prev = 0
            for r in range(w):
                lo, hi = sharding.shard_range(n, r, w)
                assert lo == prev and hi >= lo
                prev = hi
            assert prev == n
            sizes = [sharding.shard_range(n, r, w)[1] - sharding.shard_range(n, r, w)[0] for r in range(w)]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_size_exchange_gloo():
    rows = helpers.golden()["config1"][:1001]       # odd count: ragged shards
    sizes = [r[2] for r in rows]
    n = len(sizes)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, sizes, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, allsz, end, total in res:
        assert allsz == sizes
        assert end == sum(sizes) == total


# ---- payload movement: root scatter -> per-rank compress (the oracle stands in for the device step) -> gather-v ----
def _payload_worker(rank, world, port, n, S, q):
    import numpy as np
    from kompressor_amd import corpus
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharding.shard_range(n, rank, world)
    everything = torch.from_numpy(corpus.make(0, n, S, corpus.MIX_TEXT_BINARY, threads=1)) if rank == 0 else None
    local = torch.empty(max(1, (hi - lo) * S), dtype=torch.uint8)
    mine = sharding.scatter_slices(everything, local, n, S)
    # the block that arrived is the block this rank would have generated itself (configs[3]: rank-local generation)
    assert mine.numpy().tobytes() == corpus.make(lo, hi - lo, S, corpus.MIX_TEXT_BINARY, threads=1).tobytes()
    o = helpers.oracle()
    frames = [o.compress(mine[i * S:(i + 1) * S].numpy().tobytes()) for i in range(hi - lo)]
    dense = torch.frombuffer(bytearray(b"".join(frames) + b"\0"), dtype=torch.uint8)
    sizes = torch.tensor([len(f) for f in frames], dtype=torch.int32)
    stream, all_sizes, offs = sharding.gather_frames(dense, sizes, n)
    q.put((rank, None if stream is None else stream.numpy().tobytes(), all_sizes.tolist(), offs.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_scatter_compress_gather_payload_gloo():
    import numpy as np
    from kompressor_amd import corpus
    n, S = 13, 4096                                  # odd count: ragged shards (7 + 6)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_payload_worker, args=(r, 2, port, n, S, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r = q.get(timeout=180)
        res[r[0]] = r
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-rank answer: every slice through the oracle, frames back to back
    host = corpus.make(0, n, S, corpus.MIX_TEXT_BINARY, threads=1)
    o = helpers.oracle()
    frames = [o.compress(host[i * S:(i + 1) * S].tobytes()) for i in range(n)]
    assert res[0][1] == b"".join(frames)              # the root's dense stream
    assert res[1][1] is None
    for r in (0, 1):
        assert res[r][2] == [len(f) for f in frames]
        assert res[r][3] == list(np.cumsum([0] + [len(f) for f in frames[:-1]]))


def test_gather_frames_with_an_empty_shard_gloo():
    # more ranks than slices: rank 1 of 2 owns nothing when n = 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_payload_worker, args=(r, 2, port, 1, 2048, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r = q.get(timeout=180)
        res[r[0]] = r
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert len(res[0][2]) == 1 and len(res[0][1]) == res[0][2][0]
