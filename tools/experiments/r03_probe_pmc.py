"""What a random read + write pair costs at the memory side: k_region_probe (random 4-byte read at A, 4-byte write at B) over a
144 GiB span and over a 24 GiB span, run under rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum (tools/r03_probe.sh)."""
import ctypes, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from kompressor_amd import _lib
lib = _lib.load()
G = 1 << 30
big = torch.empty(144 * G, dtype=torch.uint8, device="cuda:0")
base = big.data_ptr()
def rate(ptr, nbytes, blocks=4096, iters=384):
    ms = ctypes.c_float(0)
    assert lib.kmp_debug_probe_region(ctypes.c_void_p(ptr), nbytes, blocks, iters, ctypes.byref(ms), None) == 0
    return blocks * 256 * iters / (ms.value * 1e-3) / 1e9, ms.value
for span in (144, 24):
    r, ms = rate(base, span * G)
    print(f"span {span} GiB: {r:.2f} G pairs/s ({2 * r:.1f} G accesses/s), {ms:.2f} ms for {4096 * 256 * 384} pairs (+ a warm-up launch of {4096 * 256 * 8})", flush=True)
