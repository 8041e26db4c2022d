"""A slow and a fast 24 GiB region probed over growing prefixes: does the slow kind stay slow on a small footprint?"""
import ctypes, sys
sys.path.insert(0, '/root/repo')
import torch
from kompressor_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0"); G = 1 << 30
bufs = [torch.empty(24 * G, dtype=torch.uint8, device=dev) for _ in range(8)]
def rate(t, nbytes, blocks=4096, iters=384):
    ms = ctypes.c_float(0)
    assert lib.kmp_debug_probe_region(ctypes.c_void_p(t.data_ptr()), nbytes, blocks, iters, ctypes.byref(ms), None) == 0
    return blocks * 256 * iters * 2 / (ms.value * 1e-3) / 1e9
full = [rate(t, t.numel()) for t in bufs]
print("whole regions, G accesses/s:", [round(x, 1) for x in full])
slow = bufs[min(range(8), key=lambda i: full[i])]; fast = bufs[max(range(8), key=lambda i: full[i])]
for name, t in (("slow", slow), ("fast", fast)):
    print(name, {f"{g} GiB": round(rate(t, g * G), 1) for g in (1, 2, 4, 8, 16, 24)}, flush=True)
