"""Development aid: rare paths of the split-phase parser (long backward growth, wide steps, long matches, tails) against zstd_match.h."""
import sys, os
sys.path.insert(0, os.path.dirname(__file__))
import numpy as np
from emu_match_cmp import compare
from kompressor_amd import corpus

rng = np.random.default_rng(11)
def rnd(n): return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
ds = []
for t in range(40):
    a = rnd(int(rng.integers(300, 5000))); b = rnd(int(rng.integers(200, 9000)))
    parts = [a, b, a, rnd(int(rng.integers(0, 40))), a[int(rng.integers(0, 100)):], b[:int(rng.integers(1, len(b)))], a]
    d = b"".join(parts)[:int(rng.integers(2000, 66000))]
    ds.append(d)
for t in range(24):       # text with far repeats and runs
    txt = corpus.make(4000 + t, 1, 20000, mix=ord("T")).tobytes()
    d = txt[:7000] + rnd(3000) + txt[100:9000] + b"\0" * int(rng.integers(1, 700)) + txt[50:6000] + rnd(int(rng.integers(0, 20)))
    ds.append(d[:int(rng.integers(1000, len(d)))])
for t in range(16):       # tails: sizes around the last 16 bytes
    base = corpus.make(5000 + t, 1, 4096, mix=ord("X")).tobytes()
    ds.append(base[:int(rng.integers(8, 64))]); ds.append(base[:1024 + t])
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
bad = compare(ds, G, 2, "rare")
print("BAD" if bad else "all equal", bad, len(ds))
# matches that run into the end of the slice: every distance of the match start from the end, several periods
ds = []
for per in (1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 33):
    pat = rnd(per)
    for tail in range(0, 44):
        pre = rnd(64 + per)
        body = (pat * 80)[:per + 9 + tail]
        ds.append(pre + body)
        ds.append(pre + body + rnd(1))
        ds.append(pre + body[:-1] + bytes([body[-1] ^ 1]))
bad = compare(ds, G, 4, "tails")
print("BAD" if bad else "all equal", bad, len(ds))
