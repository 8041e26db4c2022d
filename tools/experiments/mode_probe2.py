"""Several level-3 contexts ALIVE at once in one process (their workspaces at different places of the HBM), timed in turn:
does the parse kernel's fast / slow state go with the place?"""
import os, sys, time
sys.path.insert(0, '/root/repo')
os.environ.setdefault("KMP_ZSTD_AUTOTUNE", "0")
import numpy as np, torch
from kompressor_amd import corpus
from kompressor_amd.batch import ZstdBatch
n, S = 65536, 65536
dev = torch.device("cuda:0")
src = torch.from_numpy(corpus.make(0, n, S)).to(dev)
in_off = torch.arange(n, dtype=torch.int64, device=dev) * S
in_len = torch.full((n,), S, dtype=torch.int32, device=dev)
ctxs = []
for i in range(4):
    b = ZstdBatch(max_slices=n, max_slice_bytes=S, device=0)
    dst = torch.empty(n * b.out_stride + 64, dtype=torch.uint8, device=dev)
    out_off = torch.arange(n, dtype=torch.int64, device=dev) * b.out_stride
    out_len = torch.zeros(n, dtype=torch.int32, device=dev)
    ctxs.append((b, dst, out_off, out_len))
for rnd in range(3):
    row = []
    for b, dst, out_off, out_len in ctxs:
        b.compress(src, in_off, in_len, dst, out_off, out_len)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(2):
            b.compress(src, in_off, in_len, dst, out_off, out_len)
        torch.cuda.synchronize(); row.append(round((time.perf_counter() - t0) / 2 * 1e3, 1))
    print("round", rnd, "step ms per context:", row, flush=True)
for c in ctxs:
    c[0].close()
