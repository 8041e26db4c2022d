"""Which pre-splitter does libzstd 1.5.7 run for greedy / lazy / lazy2 (levels 5 .. 10)?  The first block's regenerated size in the live
library's frames (read by feeding ZSTD_decompressStream one block at a time) against a restated ZSTD_splitBlock_byChunks at
(sampling rate, hashLog) = (43, 8), (11, 9), (5, 10), (1, 10) and the from-borders splitter's result (level 1's)."""
import sys, ctypes
import numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
import helpers
from kompressor_amd import corpus
z = helpers.live_libzstd(); lib = z.lib
class Buf(ctypes.Structure): _fields_ = [("p", ctypes.c_void_p), ("size", ctypes.c_size_t), ("pos", ctypes.c_size_t)]
lib.ZSTD_createDStream.restype = ctypes.c_void_p
lib.ZSTD_decompressStream.argtypes = [ctypes.c_void_p, ctypes.POINTER(Buf), ctypes.POINTER(Buf)]; lib.ZSTD_decompressStream.restype = ctypes.c_size_t
lib.ZSTD_freeDStream.argtypes = [ctypes.c_void_p]
KNUTH = 0x9e3779b9
def events(chunk, rate, hashLog):
    a = np.frombuffer(chunk, dtype=np.uint8).astype(np.uint64)
    limit = len(a) - 2 + 1
    idx = np.arange(0, limit, rate)
    if hashLog == 8: h = a[idx]
    else: h = (((a[idx] | (a[idx + 1] << 8)) * KNUTH) & 0xFFFFFFFF) >> (32 - hashLog)
    return np.bincount(h.astype(np.int64), minlength=1 << hashLog).astype(object), limit // rate
def split(block, rate, hashLog):
    chunk = 8 << 10; penalty = 3
    past, pn = events(block[:chunk], rate, hashLog); pos = chunk
    while pos <= (128 << 10) - chunk:
        nw, nn = events(block[pos:pos + chunk], rate, hashLog)
        dev = int(sum(abs(int(x) * nn - int(y) * pn) for x, y in zip(past, nw)))
        if dev >= pn * nn * (14 + penalty) // 16: return pos
        past = past + nw; pn += nn
        if penalty > 0: penalty -= 1
        pos += chunk
    return 128 << 10
def block_sizes(frame, total):
    fhd = frame[4]; ss = (fhd >> 5) & 1; fcs = fhd >> 6; did = fhd & 3
    p = 5 + (0 if ss else 1) + (0, 1, 2, 4)[did] + ((1 if ss else 0) if fcs == 0 else (2, 4, 8)[fcs - 1])
    ds = lib.ZSTD_createDStream(); out = ctypes.create_string_buffer(total + 64); src = ctypes.create_string_buffer(frame, len(frame))
    ob = Buf(ctypes.cast(out, ctypes.c_void_p).value, total + 64, 0); sizes = []; fed = p
    ib = Buf(ctypes.cast(src, ctypes.c_void_p).value, fed, 0); lib.ZSTD_decompressStream(ds, ctypes.byref(ob), ctypes.byref(ib))
    while True:
        h = frame[fed] | (frame[fed + 1] << 8) | (frame[fed + 2] << 16)
        last, typ, sz = h & 1, (h >> 1) & 3, h >> 3
        fed += 3 + (1 if typ == 1 else sz)
        before = ob.pos
        ib = Buf(ctypes.cast(src, ctypes.c_void_p).value, fed, ib.pos)
        for _ in range(4): lib.ZSTD_decompressStream(ds, ctypes.byref(ob), ctypes.byref(ib))
        sizes.append(ob.pos - before)
        if last: break
    lib.ZSTD_freeDStream(ds)
    return sizes
rng = np.random.default_rng(3)
score = {}
for t in range(24):
    cut = int(rng.integers(1, 15)) * 8192 + int(rng.integers(-3000, 3000))
    a = corpus.make(100 + t, 1, cut, mix=ord("TXSB"[t % 4])).tobytes()
    b = corpus.make(200 + t, 1, 300000 - cut, mix=ord("BZTI"[(t // 4) % 4])).tobytes()
    d = corpus.make(300 + t, 1, 131072, mix=ord("T")).tobytes() + a + b          # (the first block is never split: savings start at 0)
    preds = {k: split(d[131072:262144], *k) for k in ((43, 8), (11, 9), (5, 10), (1, 10))}
    for lvl in (3, 5, 6, 7, 8, 9, 10, 13, 16):
        first = block_sizes(z.compress(d, lvl), len(d))[1]
        for k, pr in preds.items():
            score.setdefault(lvl, {}).setdefault(k, 0)
            score[lvl][k] += int(pr == first)
        score[lvl].setdefault("n", 0); score[lvl]["n"] += 1
        score[lvl].setdefault("distinct", 0); score[lvl]["distinct"] += int(len(set(preds.values())) > 1)
for lvl, sc in score.items(): print("level", lvl, sc)
