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
