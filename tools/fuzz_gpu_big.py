"""Differential fuzz of the block-chain path (frames of several blocks, slices of 128 KiB + 1 .. 1.5 MiB): stress inputs
(tools/fuzzgen.c) and corpus classes; the frames the reference's one-shot driver gets (output slices of max(8192, n / 10) bytes:
libzstd stages the input in 128 KiB chunks), ZSTD_compress2's frames, levels 1, 2 and three negative levels (beyond their windows of 512 KiB / 1 MiB the fast extDict parse) -- every frame against
the binary libzstd 1.5.7 on the host cores, every frame decoded back on the GPU.  usage: python tools/fuzz_gpu_big.py [seed] [n]"""
import os, sys, ctypes, subprocess, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from kompressor_amd import corpus
from kompressor_amd.batch import ZstdBatch
from libzstd_ref import LibZstd

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
out_dir = os.path.join(ROOT, "gpurun_out"); os.makedirs(out_dir, exist_ok=True)
so = os.path.join(out_dir, "libfuzzgen.so")
subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "fuzzgen.c")], check=True)
FG = ctypes.CDLL(so); FG.fuzz_fill.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64]
rng = np.random.default_rng(seed)
MAXL = 1536 * 1024
lens = np.where(rng.random(N) < 0.5, rng.integers(131073, 524289, N), rng.integers(131073, MAXL + 1, N)).astype(np.int64)
lens = np.where(rng.random(N) < 0.1, rng.integers(1, 131073, N), lens).astype(np.int64)        # a tenth of them one-block slices in the same batches
lens = np.where(rng.random(N) < 0.06, rng.integers(1, 5, N) * 131072 + rng.integers(1, 80, N), lens).astype(np.int64)          # a short last block (the RLE / raw rules of tiny blocks)
lens[:6] = [131073, 131072 * 2, 131072 * 2 + 1, 262144 + 131072, MAXL, 524288]
offs = np.concatenate([[0], np.cumsum(lens[:-1])]).astype(np.int64)
total = int(lens.sum())
host = np.empty(total + 64, dtype=np.uint8)
for i in range(N):
    if i % 3 != 2: FG.fuzz_fill(host[offs[i]:].ctypes.data, int(lens[i]), seed * 7919 + i)
    else: host[offs[i]:offs[i] + lens[i]] = corpus.make(900000 + seed * N + i, 1, int(lens[i]), mix=ord("TXSBDIZR"[(i // 3) % 8]))
# where the last block is short (see lens above), half of the slices end in a run of one byte that starts before the block does
for i in range(N):
    t = int(lens[i]) % 131072
    if 0 < t < 80 and lens[i] > 131072 and i % 2 == 0:
        e = int(offs[i] + lens[i]); host[e - t - int(rng.integers(0, 40)):e] = host[e - t - 40]
print(f"seed {seed}: {N} slices, {total / 1e9:.2f} GB", flush=True)
z = LibZstd()
import threading
_tl = threading.local()
def one_shot(d, level):                          # ZSTD_compress2 on a context of the calling thread's own
    if not hasattr(_tl, "z"): _tl.z = LibZstd()
    return _tl.z.compress(d, level)
def ref_frames(fn, idx):
    chunks = [idx[k::16] for k in range(16)]
    with ThreadPoolExecutor(16) as ex: parts = list(ex.map(lambda ch: [fn(host[offs[i]:offs[i] + lens[i]].tobytes()) for i in ch], chunks))
    res = {}
    for ch, pa in zip(chunks, parts):
        for i, f in zip(ch, pa): res[int(i)] = f
    return res
src = torch.from_numpy(host).cuda()
d_off = torch.from_numpy(offs).cuda(); d_len = torch.from_numpy(lens.astype(np.int32)).cuda()
def gpu_frames(level, reference, idx, max_bytes, streaming=None):
    b = ZstdBatch(max_slices=len(idx), max_slice_bytes=max_bytes)
    sel = torch.from_numpy(np.asarray(idx, dtype=np.int64)).cuda()
    dst, ooff, olen = b.compress(src, d_off[sel], d_len[sel], level=level, reference=reference, streaming=streaming, check=True)
    torch.cuda.synchronize()
    d, oo, ol = dst.cpu().numpy(), ooff.cpu().numpy(), olen.cpu().numpy()
    frames = {int(idx[k]): d[int(oo[k]):int(oo[k]) + int(ol[k])].tobytes() for k in range(len(idx))}
    out, o2, l2, st = b.decompress(dst, ooff, olen, d_len[sel])
    torch.cuda.synchronize()
    assert int(st.abs().sum().item()) == 0 and bool((l2 == d_len[sel]).all()), "decode status"
    oh, o2h = out.cpu().numpy(), o2.cpu().numpy()
    for k in range(0, len(idx), 7):
        i = int(idx[k]); assert oh[int(o2h[k]):int(o2h[k]) + int(lens[i])].tobytes() == host[offs[i]:offs[i] + lens[i]].tobytes(), ("round trip", i)
    b.close()
    return frames
all_idx = np.arange(N); l1_idx = np.nonzero(lens <= 524288)[0]
l4_idx = np.nonzero(((lens > 16384) & (lens <= 131072)) | (lens > 262144))[0]          # the size classes level 4 runs as double-fast
bad_total = 0
cutr = np.random.default_rng(seed + 99)
def streamed(d, level, empty):                   # finish = false calls, then finish = true with / without data; 8 KiB output slices
    n = len(d)
    if empty or n < 2: return z.compress_streaming(d, [0, n, n], 8192, level)
    k = 1 + (hash(d[:64]) % (n - 1))
    return z.compress_streaming(d, [0, k, n], 8192, level)
for name, level, reference, idx, fn, streaming in (
        ("level 3, streamed, the closing call brings data", 3, False, all_idx, lambda d: streamed(d, 3, False), "data"),
        ("level 3, streamed, the closing call is empty", 3, False, all_idx, lambda d: streamed(d, 3, True), "empty"),
        ("level 1, streamed, the closing call brings data", 1, False, all_idx, lambda d: streamed(d, 1, False), "data"),
        ("level 3, the reference driver's frames (input staged in 128 KiB chunks)", 3, True, all_idx, lambda d: z.compress_streaming(d, [0, len(d)], max(8192, len(d) // 10), 3), None),
        ("level 3, ZSTD_compress2's frames", 3, False, all_idx, lambda d: one_shot(d, 3), None),
        ("level 1, the reference driver's frames", 1, True, all_idx, lambda d: z.compress_streaming(d, [0, len(d)], max(8192, len(d) // 10), 1), None),
        ("level 1, ZSTD_compress2's frames", 1, False, all_idx, lambda d: one_shot(d, 1), None),
        ("level -1, the reference driver's frames", -1, True, all_idx, lambda d: z.compress_streaming(d, [0, len(d)], max(8192, len(d) // 10), -1), None),
        ("level -5, ZSTD_compress2's frames", -5, False, all_idx, lambda d: one_shot(d, -5), None),
        ("level -3, streamed, the closing call brings data", -3, False, all_idx, lambda d: streamed(d, -3, False), "data"),
        ("level 2, streamed, the closing call brings data (beyond its 1 MiB window: the fast extDict parse)", 2, False, all_idx, lambda d: streamed(d, 2, False), "data"),
        ("level 2, the reference driver's frames", 2, True, all_idx, lambda d: z.compress_streaming(d, [0, len(d)], max(8192, len(d) // 10), 2), None),
        ("level 2, ZSTD_compress2's frames", 2, False, all_idx, lambda d: one_shot(d, 2), None),
        ("level 4 (its double-fast size classes), the reference driver's frames", 4, True, l4_idx, lambda d: z.compress_streaming(d, [0, len(d)], max(8192, len(d) // 10), 4), None),
        ("level 4 (its double-fast size classes), ZSTD_compress2's frames", 4, False, l4_idx, lambda d: one_shot(d, 4), None),
        ("level 4, streamed, the closing call brings data", 4, False, all_idx, lambda d: streamed(d, 4, False), "data"),
        ("level 4, streamed, the closing call is empty", 4, False, all_idx, lambda d: streamed(d, 4, True), "empty")):
    t0 = time.time()
    g = gpu_frames(level, reference, idx, MAXL, streaming)
    r = ref_frames(fn, idx)
    bad = [i for i in idx if g[int(i)] != r[int(i)]]
    bad_total += len(bad)
    print(f"{name}: {len(idx)} frames against libzstd 1.5.7, different: {len(bad)} {[(int(i), int(lens[i]), i % 3) for i in bad[:8]]}  ({time.time() - t0:.0f} s)", flush=True)
print("FUZZ OK" if bad_total == 0 else f"FUZZ FOUND {bad_total} DIFFERENCES")
sys.exit(0 if bad_total == 0 else 1)
