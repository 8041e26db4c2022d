"""Random read-modify-write rate of consecutive 24 GiB regions of the HBM (k_region_probe through kmp_debug_probe_region):
is the level-3 parser's fast / slow state a property of WHERE its tables land?"""
import ctypes, sys
sys.path.insert(0, '/root/repo')
import torch
from kompressor_amd import _lib
lib = _lib.load()
lib.kmp_debug_probe_region.restype = ctypes.c_int
lib.kmp_debug_probe_region.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_float), ctypes.c_void_p]
dev = torch.device("cuda:0")
G = 1 << 30
bufs = []
for i in range(10):
    try:
        bufs.append(torch.empty(24 * G, dtype=torch.uint8, device=dev))
    except Exception as e:
        print("allocation", i, "failed:", str(e)[:60]); break
blocks, iters = 4096, 512
for rnd in range(2):
    row = []
    for t in bufs:
        ms = ctypes.c_float(0)
        rc = lib.kmp_debug_probe_region(ctypes.c_void_p(t.data_ptr()), t.numel(), blocks, iters, ctypes.byref(ms), None)
        assert rc == 0, _lib.last_error()
        row.append(round(blocks * 256 * iters * 2 / (ms.value * 1e-3) / 1e9, 1))
    print("round", rnd, "G accesses/s per region:", row, flush=True)
print("addresses (GiB):", [round(t.data_ptr() / G, 1) for t in bufs])
