"""Random-access rate over windows and growing spans of ONE 192 GiB allocation: how does the rate depend on where a window
lies and how wide it is?"""
import ctypes, sys
sys.path.insert(0, '/root/repo')
import torch
from kompressor_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0"); G = 1 << 30
big = torch.empty(192 * G, dtype=torch.uint8, device=dev)
base = big.data_ptr()
def rate(ptr, nbytes, blocks=4096, iters=384):
    ms = ctypes.c_float(0)
    assert lib.kmp_debug_probe_region(ctypes.c_void_p(ptr), nbytes, blocks, iters, ctypes.byref(ms), None) == 0
    return blocks * 256 * iters * 2 / (ms.value * 1e-3) / 1e9
print("24 GiB windows:", [round(rate(base + k * 24 * G, 24 * G), 1) for k in range(8)], flush=True)
print("8 GiB windows :", [round(rate(base + k * 8 * G, 8 * G), 1) for k in range(24)], flush=True)
for k in (1, 2, 3, 4, 6, 8):
    print(f"span of {24 * k} GiB from the start: {rate(base, 24 * k * G):.1f} G accesses/s", flush=True)
print("8192 blocks over the whole allocation: %.1f" % rate(base, 192 * G, blocks=8192, iters=192))
