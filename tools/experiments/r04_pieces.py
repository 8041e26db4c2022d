"""Do the pieces of kmp_zstd_compress_batch_pieces run side by side?  Resident data, no copies: P = 1, 2, 4, 8 against the one-launch batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from kompressor_amd import corpus
from kompressor_amd.batch import ZstdBatch
n, S = 65536, 65536
dev = torch.device("cuda", 0)
src = torch.empty(n * S, dtype=torch.uint8, device=dev)
for lo in range(0, n, 16384):
    src[lo * S:(lo + 16384) * S] = torch.from_numpy(corpus.make(lo, 16384, S)).to(dev)
b = ZstdBatch(max_slices=n, max_slice_bytes=S, table_span_gib=100, table_retry=1)
in_off = torch.arange(n, dtype=torch.int64, device=dev) * S
in_len = torch.full((n,), S, dtype=torch.int32, device=dev)
dst = torch.empty(n * b.out_stride + 64, dtype=torch.uint8, device=dev)
out_off = torch.arange(n, dtype=torch.int64, device=dev) * b.out_stride
out_len = torch.zeros(n, dtype=torch.int32, device=dev)
def t(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
print("one launch: %.1f ms" % t(lambda: b.compress(src, in_off, in_len, dst, out_off, out_len)))
for P in (1, 2, 4, 8):
    streams = [torch.cuda.Stream(device=dev) for _ in range(P)]
    print("P = %d: %.1f ms" % (P, t(lambda: b.compress_pieces(src, in_off, in_len, dst, out_off, out_len, streams))))
    # staggered starts: piece p is released 20 ms after piece p - 1 (what a copy stream would do)
    if P > 1:
        def stag():
            for p in range(P):
                pass
            b.compress_pieces(src, in_off, in_len, dst, out_off, out_len, streams)
        # (no sleep primitive on the device here: the staggered case is what bench.py's end_to_end_pcie measures)
b.close()
