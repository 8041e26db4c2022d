"""kmp_zstd_compress_host_batch on a large batch held in (pageable) host memory: time and the frames against the device batch."""
import os, sys, time, ctypes, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from kompressor_amd import corpus, _lib
from kompressor_amd.batch import ZstdBatch, compress_bound
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
S = 65536
lib = _lib.load()
host = corpus.make(0, n, S)
lens = np.full(n, S, dtype=np.uint32); offs = np.arange(n, dtype=np.uint64) * S
cap = compress_bound(S); caps = np.full(n, cap, dtype=np.uint32); ooff = np.arange(n, dtype=np.uint64) * cap
dst = np.empty(n * cap + 64, dtype=np.uint8); olen = np.zeros(n, dtype=np.uint32)
p = lambda a: ctypes.c_void_p(a.ctypes.data)
for rep in range(3):
    t0 = time.perf_counter()
    rc = lib.kmp_zstd_compress_host_batch(0, 3, p(host), p(offs), p(lens), n, p(dst), p(ooff), p(caps), p(olen))
    dt = time.perf_counter() - t0
    assert rc == 0, (rc, _lib.last_error())
    print(f"pass {rep}: {n} x 64 KiB from host memory to frames in host memory: {dt * 1e3:.0f} ms = {n * S / dt / 1e9:.2f} GB/s (workers {os.environ.get('KMP_HOST_BULK_WORKERS', '2')}, pieces of {os.environ.get('KMP_HOST_BULK_SLICES', '16384')})", flush=True)
# against the device batch
m = min(n, 16384)
b = ZstdBatch(max_slices=m, max_slice_bytes=S)
src = torch.from_numpy(host[: m * S]).cuda()
d2, o2, l2 = b.compress(src, torch.arange(m, dtype=torch.int64, device="cuda") * S, torch.full((m,), S, dtype=torch.int32, device="cuda"), check=True)
torch.cuda.synchronize()
l2 = l2.cpu().numpy(); d2 = d2.cpu().numpy(); o2 = o2.cpu().numpy()
assert (l2 == olen[:m]).all()
for i in range(0, m, 97):
    assert dst[int(ooff[i]):int(ooff[i]) + int(olen[i])].tobytes() == d2[int(o2[i]):int(o2[i]) + int(l2[i])].tobytes(), i
print("frames equal the device batch's (every 97th of the first", m, "compared)")
