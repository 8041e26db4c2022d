"""When do the slices of a batch finish inside one k_zstd_match launch?  (KMP_MATCH_FLAGS bit 8: the parser stamps the device's
100 MHz clock into each slice's record.)  Prints the share of slices still being parsed over the launch, by class."""
import os, sys, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["KMP_MATCH_FLAGS"] = str(6 | 256)
import numpy as np, torch
from kompressor_amd import corpus, _lib
from kompressor_amd.batch import ZstdBatch
n, S = 65536, 65536
src = torch.empty(n * S, dtype=torch.uint8, device="cuda")
for c in range(0, n, 4096): src[c * S:(c + 4096) * S] = torch.from_numpy(corpus.make(c, 4096, S)).cuda()
in_off = torch.arange(n, dtype=torch.int64, device="cuda") * S
in_len = torch.full((n,), S, dtype=torch.int32, device="cuda")
b = ZstdBatch(max_slices=n, max_slice_bytes=S)
b.set_profiling(True)
for _ in range(2): dst, ooff, olen = b.compress(src, in_off, in_len)
torch.cuda.synchronize()
ms = b.last_kernel_ms(0)
meta = np.zeros(n * 8, dtype=np.uint32)
assert _lib.load().kmp_debug_copy_meta(b._h, ctypes.c_void_p(meta.ctypes.data), n) == 0
meta = meta.reshape(n, 8)
t = (meta[:, 6].astype(np.uint64) | (meta[:, 7].astype(np.uint64) << np.uint64(32))).astype(np.float64) / 100e3      # ms
t -= t.min()
end = t.max()
print(f"k_zstd_match {ms:.1f} ms; first slice done at 0, last at {end:.1f} ms (the launch began ~{ms - end:.1f} ms before the first finished)")
cls = np.array([corpus.slice_class(i) for i in range(n)])
start = end - ms                      # launch start on this axis
edges = np.linspace(start, end, 21)
print("share of the batch's slices still in flight at 5 % steps of the launch:")
print(" ".join(f"{(t > e).mean() * 100:5.1f}" for e in edges))
for c_ in "TXSBDIZR":
    tt = t[cls == c_]
    print(f"class {c_}: finishes at {100 * (np.median(tt) - start) / ms:5.1f} % (median), {100 * (np.percentile(tt, 99) - start) / ms:5.1f} % (99th), share of batch {100 * len(tt) / n:.1f} %")
