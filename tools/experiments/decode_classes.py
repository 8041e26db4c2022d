import sys, time, json
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from kompressor_amd import corpus
from kompressor_amd.batch import ZstdBatch
n, S = 16384, 65536
dev = torch.device("cuda:0")
b = ZstdBatch(max_slices=n, max_slice_bytes=S, device=0)
in_off = torch.arange(n, dtype=torch.int64, device=dev) * S
in_len = torch.full((n,), S, dtype=torch.int32, device=dev)
cap = torch.full((n,), S, dtype=torch.int32, device=dev)
for cls in "TXSBDIZR":
    host = corpus.make(1 << 20, n, S, mix=ord(cls))
    src = torch.from_numpy(host).to(dev)
    for level in (3, 1):
        dst, ooff, olen = b.compress(src, in_off, in_len, level=level)
        torch.cuda.synchronize()
        back = torch.empty(n * S + 64, dtype=torch.uint8, device=dev)
        b.decompress(dst, ooff, olen, cap, dst=back, out_off=in_off)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            _, _, l2, st = b.decompress(dst, ooff, olen, cap, dst=back, out_off=in_off)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        ok = int(st.abs().sum().item()) == 0 and torch.equal(back[: n * S], src)
        print(cls, "level", level, "ratio %.2f" % (n * S / float(olen.sum().item())), "decode %.1f GB/s" % (n * S / dt / 1e9), "ok" if ok else "MISMATCH", flush=True)
b.close()
