"""Does it matter where the INPUT lies?  The same context compresses the same batch from a slow and from a fast region of the HBM."""
import ctypes, os, sys, time
sys.path.insert(0, '/root/repo')
os.environ.setdefault("KMP_ZSTD_AUTOTUNE", "0")
import numpy as np, torch
from kompressor_amd import corpus, _lib
from kompressor_amd.batch import ZstdBatch
lib = _lib.load()
n, S = 65536, 65536
dev = torch.device("cuda:0")
host = torch.from_numpy(corpus.make(0, n, S))
cands = []
for i in range(12):
    t = torch.empty(n * S + 64, dtype=torch.uint8, device=dev)
    ms = ctypes.c_float(0)
    assert lib.kmp_debug_probe_region(ctypes.c_void_p(t.data_ptr()), t.numel() & ~3, 4096, 128, ctypes.byref(ms), None) == 0
    cands.append((ms.value, t))
cands.sort(key=lambda x: x[0])
print("probe ms of the candidates:", [round(c[0], 2) for c in cands])
fast, slow = cands[0][1], cands[-1][1]
b = ZstdBatch(max_slices=n, max_slice_bytes=S, device=0)
in_off = torch.arange(n, dtype=torch.int64, device=dev) * S
in_len = torch.full((n,), S, dtype=torch.int32, device=dev)
dst = torch.empty(n * b.out_stride + 64, dtype=torch.uint8, device=dev)
out_off = torch.arange(n, dtype=torch.int64, device=dev) * b.out_stride
out_len = torch.zeros(n, dtype=torch.int32, device=dev)
for name, buf in (("fast", fast), ("slow", slow), ("fast", fast), ("slow", slow)):
    buf[: n * S].copy_(host)
    b.compress(buf, in_off, in_len, dst, out_off, out_len)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        b.compress(buf, in_off, in_len, dst, out_off, out_len)
    torch.cuda.synchronize()
    print("input in a", name, "region: %.1f ms per step" % ((time.perf_counter() - t0) / 3 * 1e3), flush=True)
b.close()
