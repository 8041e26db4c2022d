"""Decoder fuzz on the GPU box: frames the binary libzstd 1.5.7 writes at levels -5 .. 19 (every strategy: fast, dfast, greedy, lazy,
btlazy2, btopt, btultra -- raw / RLE / compressed / treeless literals, predefined / RLE / FSE / repeated sequence tables, frames
of one and of many blocks) from stress inputs (tools/fuzzgen.c) and corpus classes -> kmp_zstd_decompress_batch -> every byte
compared with the source.  Two batches: ragged slices up to 128 KiB, and slices of 128 KiB .. 1 MiB.  usage: [seed] [n_small] [n_big]"""
import os, sys, ctypes, subprocess, time, threading
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from kompressor_amd import corpus
from kompressor_amd.batch import ZstdBatch
from libzstd_ref import LibZstd

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 12000
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 600
out_dir = os.path.join(ROOT, "gpurun_out"); os.makedirs(out_dir, exist_ok=True)
so = os.path.join(out_dir, "libfuzzgen.so")
subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "fuzzgen.c")], check=True)
FG = ctypes.CDLL(so); FG.fuzz_fill.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64]
_tl = threading.local()
def frame_of(d, level):
    if not hasattr(_tl, "z"): _tl.z = LibZstd()
    return _tl.z.compress(d, level)
LEVELS = [-5, 1, 2, 3, 4, 5, 6, 7, 9, 12, 15, 16, 19]
def run(tag, lens, max_bytes, salt):
    rng = np.random.default_rng(seed * 31 + salt)
    N = len(lens)
    offs = np.concatenate([[0], np.cumsum(lens[:-1])]).astype(np.int64); total = int(lens.sum())
    host = np.empty(total + 64, dtype=np.uint8)
    for i in range(N):
        if lens[i] == 0: continue
        if i % 3 != 2: FG.fuzz_fill(host[offs[i]:].ctypes.data, int(lens[i]), seed * 104729 + salt * 7 + i)
        else: host[offs[i]:offs[i] + lens[i]] = corpus.make(1200000 + seed * N + i, 1, int(lens[i]), mix=ord("TXSBDIZR"[(i // 3) % 8]))
    lv = np.array([LEVELS[k] for k in rng.integers(0, len(LEVELS), N)])
    heavy = (lv >= 15) & (lens > 65536); lv[heavy] = 9                         # (keep the CPU side to a minute)
    t0 = time.time()
    chunks = [np.arange(N)[k::16] for k in range(16)]
    with ThreadPoolExecutor(16) as ex: parts = list(ex.map(lambda ch: [frame_of(host[offs[i]:offs[i] + lens[i]].tobytes(), int(lv[i])) for i in ch], chunks))
    frames = [None] * N
    for ch, pa in zip(chunks, parts):
        for i, f in zip(ch, pa): frames[int(i)] = f
    t1 = time.time()
    flen = np.array([len(f) for f in frames], dtype=np.int64); foff = np.concatenate([[0], np.cumsum(flen[:-1])]).astype(np.int64)
    blob = np.frombuffer(b"".join(frames) + bytes(64), dtype=np.uint8).copy()
    b = ZstdBatch(max_slices=N, max_slice_bytes=max_bytes)
    cap = torch.from_numpy(np.maximum(lens, 1).astype(np.int32)).cuda()
    out, o2, l2, st = b.decompress(torch.from_numpy(blob).cuda(), torch.from_numpy(foff).cuda(), torch.from_numpy(flen.astype(np.int32)).cuda(), cap)
    torch.cuda.synchronize()
    sth, l2h, oh, o2h = st.cpu().numpy(), l2.cpu().numpy(), out.cpu().numpy(), o2.cpu().numpy()
    bad = [i for i in range(N) if sth[i] != 0 or l2h[i] != lens[i] or oh[int(o2h[i]):int(o2h[i]) + int(lens[i])].tobytes() != host[offs[i]:offs[i] + lens[i]].tobytes()]
    b.close()
    print(f"{tag}: {N} frames ({total / 1e9:.2f} GB, levels {sorted(set(int(x) for x in lv))}; libzstd {t1 - t0:.0f} s), wrong after decoding: {len(bad)} {[(i, int(lens[i]), int(lv[i]), int(sth[i])) for i in bad[:8]]}", flush=True)
    return len(bad)
rng0 = np.random.default_rng(seed)
small = np.where(rng0.random(NS) < 0.1, rng0.integers(0, 300, NS), rng0.integers(0, 131073, NS)).astype(np.int64); small[:6] = [0, 1, 7, 8, 131072, 65536]
big = rng0.integers(131073, (1 << 20) + 1, NB).astype(np.int64); big[:3] = [131073, 1 << 20, 262144]
n_bad = run("slices up to 128 KiB", small, 131072, 1) + run("slices of 128 KiB .. 1 MiB", big, 1 << 20, 2)
print("FUZZ OK" if n_bad == 0 else f"FUZZ FOUND {n_bad} WRONG FRAMES")
sys.exit(0 if n_bad == 0 else 1)
