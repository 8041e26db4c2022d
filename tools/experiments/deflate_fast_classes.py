"""k_deflate_fast (raw DEFLATE level 1) on batches of ONE content class and of different sizes: how long a wave of only
incompressible slices takes when the device is full of them / nearly empty (is a sort of the lane slots worth it?)."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from kompressor_amd import corpus
from kompressor_amd.batch import ZstdBatch
S = 65536
dev = torch.device("cuda:0")
b = ZstdBatch(max_slices=65536, max_slice_bytes=S, device=0)
b.set_profiling(True)
for cls, n in [(c, 16384) for c in "TXSBDIZR"] + [("M", 65536), ("M", 16384)]:
    host = corpus.make(1 << 21, n, S, mix=ord(cls)) if cls != "M" else corpus.make(0, n, S)
    src = torch.from_numpy(host).to(dev)
    in_off = torch.arange(n, dtype=torch.int64, device=dev) * S
    in_len = torch.full((n,), S, dtype=torch.int32, device=dev)
    b.deflate(src, in_off, in_len, level=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dst, ooff, olen = b.deflate(src, in_off, in_len, level=1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(cls, n, "step %.1f ms" % (dt * 1e3), {k: round(v, 1) for k, v in b.deflate_kernel_ms().items()}, "ratio %.2f" % (n * S / float(olen.sum().item())), flush=True)
    del src, dst
b.close()
