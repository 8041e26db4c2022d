"""zstd level 3 / level 1 compression of batches of ONE content class (65 536 x 64 KiB): which content the parse kernels are
slowest on (a launch with every slice in flight lasts as long as its slowest slice)."""
import os, sys, time
sys.path.insert(0, '/root/repo')
os.environ.setdefault("KMP_ZSTD_AUTOTUNE", "0")
import numpy as np, torch
from kompressor_amd import corpus
from kompressor_amd.batch import ZstdBatch
n, S = 65536, 65536
dev = torch.device("cuda:0")
b = ZstdBatch(max_slices=n, max_slice_bytes=S, device=0)
in_off = torch.arange(n, dtype=torch.int64, device=dev) * S
in_len = torch.full((n,), S, dtype=torch.int32, device=dev)
dst = torch.empty(n * b.out_stride + 64, dtype=torch.uint8, device=dev)
out_off = torch.arange(n, dtype=torch.int64, device=dev) * b.out_stride
out_len = torch.zeros(n, dtype=torch.int32, device=dev)
for cls in "TXSBDIZRM":
    host = corpus.make(1 << 22, n, S, mix=ord(cls)) if cls != "M" else corpus.make(0, n, S)
    src = torch.from_numpy(host).to(dev)
    row = []
    for level in (3, 1):
        b.compress(src, in_off, in_len, dst, out_off, out_len, level=level)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(2):
            b.compress(src, in_off, in_len, dst, out_off, out_len, level=level)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
        row.append("level %d: %.1f ms (%.1f GB/s, ratio %.2f)" % (level, dt * 1e3, n * S / dt / 1e9, n * S / float(out_len.sum().item())))
    print(cls, " | ".join(row), flush=True)
    del src
b.close()
