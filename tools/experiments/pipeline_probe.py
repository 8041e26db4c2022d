"""Two contexts on two streams taking the steps alternately (what a caller that keeps two batches in flight gets): the
entropy launch of one batch runs beside the parse of the next.  Informational; bench.py measures one context on one stream."""
import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from kompressor_amd import corpus
from kompressor_amd.batch import ZstdBatch
n, S = 65536, 65536
dev = torch.device("cuda:0")
src = torch.from_numpy(corpus.make(0, n, S)).to(dev)
in_off = torch.arange(n, dtype=torch.int64, device=dev) * S
in_len = torch.full((n,), S, dtype=torch.int32, device=dev)
ctx = []
for i in range(2):
    b = ZstdBatch(max_slices=n, max_slice_bytes=S, device=0)
    st = torch.cuda.Stream()
    dst = torch.empty(n * b.out_stride + 64, dtype=torch.uint8, device=dev)
    out_off = torch.arange(n, dtype=torch.int64, device=dev) * b.out_stride
    out_len = torch.zeros(n, dtype=torch.int32, device=dev)
    ctx.append((b, st, dst, out_off, out_len))
torch.cuda.synchronize()

def run(steps, two):
    for k in range(steps):
        b, st, dst, out_off, out_len = ctx[k & 1 if two else 0]
        with torch.cuda.stream(st):
            b.compress(src, in_off, in_len, dst, out_off, out_len)

for two in (False, True, False, True):
    run(4, two)                                   # trial batches of the contexts + warmup
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(8, two)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
    print("two contexts / streams" if two else "one context", "%.1f ms per step, %.2f GB/s" % (dt * 1e3, n * S / dt / 1e9), "settings kept:", [c[0].last_chunks() for c in ctx], flush=True)
same = torch.equal(ctx[0][4], ctx[1][4])
print("frame sizes of the two contexts equal:", same)
for c in ctx:
    c[0].close()
