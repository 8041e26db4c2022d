"""The RCCL leg of the sharded path, as far as ONE GPU allows: `nccl` (= RCCL on ROCm) initialised
at world_size 1 in this process, the four exchange calls of kompressor_amd/sharding.py on DEVICE tensors against the
single-rank answer, and bench.py's --gpus 1 path forced through the distributed code (KMP_BENCH_FORCE_DIST=1) with its
scatter + compress + gather-v figure verified.  The two-rank payload test stays on gloo (tests/test_sharding_gloo.py);
the first N > 1 run on hardware is the driver's."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import helpers
from kompressor_amd import corpus, sharding

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.fixture(scope="module")
def rccl_world1():
    import torch.distributed as dist
    assert torch.cuda.is_available(), "these tests need the MI355X"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl"
    yield dist
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_calls_on_device_tensors_over_rccl(rccl_world1):
    """scatter_slices -> kmp_zstd_compress_batch + kmp_compact_batch -> gather_frame_sizes / gather_frames -> gather_slices, every
    buffer a device tensor, the collectives RCCL's: the root's stream is the single-rank stream (frames of the oracle, back to back)."""
    from kompressor_amd.batch import ZstdBatch
    dev = torch.device("cuda", 0)
    n, S = 257, 65536                                                     # an odd count: ragged shards wherever world > 1
    host = corpus.make(0, n, S, corpus.MIX_TEXT_BINARY)
    everything = torch.from_numpy(host).to(dev)
    local = torch.empty(n * S, dtype=torch.uint8, device=dev)
    mine = sharding.scatter_slices(everything, local, n, S)
    assert mine.is_cuda and torch.equal(mine, everything)
    b = ZstdBatch(max_slices=n, max_slice_bytes=S, device=0)
    in_off = torch.arange(n, dtype=torch.int64, device=dev) * S
    in_len = torch.full((n,), S, dtype=torch.int32, device=dev)
    dst, ooff, olen = b.compress(mine, in_off, in_len, check=True)
    dense, doff = b.compact(dst, ooff, olen)
    sizes = sharding.gather_frame_sizes(olen, n)
    assert sizes.is_cuda and torch.equal(sizes, olen)
    stream, all_sizes, offs = sharding.gather_frames(dense, olen, n)
    assert stream.is_cuda and torch.equal(all_sizes, olen) and torch.equal(offs, doff[:n])
    o = helpers.oracle()
    want = b"".join(o.compress(host[i * S:(i + 1) * S].tobytes()) for i in range(n))
    assert stream.cpu().numpy().tobytes() == want
    # gather-v into a caller's buffer, and the inverse of the scatter
    outbuf = torch.empty(len(want) + 4096, dtype=torch.uint8, device=dev)
    stream2, _, _ = sharding.gather_frames(dense, olen, n, out=outbuf)
    assert stream2.data_ptr() == outbuf.data_ptr() and stream2.cpu().numpy().tobytes() == want
    back = sharding.gather_slices(mine, n, S)
    assert back.is_cuda and torch.equal(back, everything)
    # an all_reduce of the kind bench.py uses for its max-over-ranks time
    t = torch.tensor([1.25], dtype=torch.float64, device=dev)
    rccl_world1.all_reduce(t, op=rccl_world1.ReduceOp.MAX)
    assert float(t.item()) == 1.25
    b.close()


@pytest.mark.timeout(600)
def test_bench_gpus_1_through_the_distributed_path():
    """bench.py --gpus 1 started the way the driver starts N > 1 (torch.distributed.run), with KMP_BENCH_FORCE_DIST=1 so that a single rank
    goes through init_process_group("nccl"), the size all_gather inside the step and the second figure (root scatter + step + gather-v):
    one JSON line, with_scatter_gather.verified == true.  A small batch: this is a test of the path, not a measurement."""
    env = dict(os.environ)
    env["KMP_BENCH_FORCE_DIST"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--slices", "4096", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-stream", "--no-pcie", "--no-extra"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=540, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["value"] > 0
    x = line["with_scatter_gather"]
    assert x["verified"] is True and "nccl" in x["what"] and x["value"] > 0
    print("[rccl world 1] " + json.dumps({"value": line["value"], "with_scatter_gather": x}))
