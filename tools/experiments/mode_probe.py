"""k_zstd_match's time over several contexts created one after the other in ONE process (is the fast / slow state of a run a
matter of where the workspace lands?)."""
import os, sys, time
sys.path.insert(0, '/root/repo')
os.environ.setdefault("KMP_ZSTD_AUTOTUNE", "0")
import numpy as np, torch
from kompressor_amd import corpus
from kompressor_amd.batch import ZstdBatch
n, S = 65536, 65536
dev = torch.device("cuda:0")
host = corpus.make(0, n, S)
src = torch.from_numpy(host).to(dev)
in_off = torch.arange(n, dtype=torch.int64, device=dev) * S
in_len = torch.full((n,), S, dtype=torch.int32, device=dev)
keep = []
for i in range(6):
    b = ZstdBatch(max_slices=n, max_slice_bytes=S, device=0)
    b.set_profiling(True)
    dst = torch.empty(n * b.out_stride + 64, dtype=torch.uint8, device=dev)
    out_off = torch.arange(n, dtype=torch.int64, device=dev) * b.out_stride
    out_len = torch.zeros(n, dtype=torch.int32, device=dev)
    ts = []
    for k in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        b.compress(src, in_off, in_len, dst, out_off, out_len)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("context", i, "step ms", [round(t, 1) for t in ts], flush=True)
    if i % 2 == 0:
        keep.append(torch.empty(3 << 30, dtype=torch.uint8, device=dev))     # shift where the next context's workspace lands
    b.close(); del dst
