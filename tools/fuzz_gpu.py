"""Differential fuzz on the GPU box: ragged slices of stress inputs (tools/fuzzgen.c: short-distance matches, repeated offsets,
small alphabets, periodic data, abrupt regime changes) and of the corpus classes through every level-3 path (team width 4, the
per-batch width 8, the split-phase parser, the fused kernel), levels 1, 2, 4, two of 5 .. 10 and three negative ones, and raw DEFLATE at levels 1, 6 and 9 -- EVERY frame compared with the binary
libzstd 1.5.7 (DEFLATE: with this machine's zlib) on the host cores (Pillow's copy: test infrastructure, looked up by oracle/libzstd_ref.py), and decoded back on
the GPU.  usage: python tools/fuzz_gpu.py [seed] [n_slices]"""
import os, sys, ctypes, subprocess, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from kompressor_amd import corpus
from kompressor_amd.batch import ZstdBatch
from libzstd_ref import find_libzstd_157

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 24000
out_dir = os.path.join(ROOT, "gpurun_out"); os.makedirs(out_dir, exist_ok=True)
so = os.path.join(out_dir, "libfuzzgen.so")
subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "fuzzgen.c")], check=True)
FG = ctypes.CDLL(so); FG.fuzz_fill.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64]
rng = np.random.default_rng(seed)
lens = np.where(rng.random(N) < 0.10, rng.integers(0, 300, N), np.where(rng.random(N) < 0.5, rng.integers(0, 131073, N), rng.integers(16385, 131073, N))).astype(np.int64)
lens[:10] = [0, 1, 7, 8, 9, 131072, 131071, 65536, 16384, 16385]
offs = np.concatenate([[0], np.cumsum(lens[:-1])]).astype(np.int64)
total = int(lens.sum())
host = np.empty(total + 64, dtype=np.uint8)
# three quarters stress inputs (one generator stream per slice), one quarter corpus classes cut raggedly
for i in range(N):
    if i % 4 != 3 and lens[i]:
        FG.fuzz_fill(host[offs[i]:].ctypes.data, int(lens[i]), seed * 1000003 + i)
for i in range(3, N, 4):
    if lens[i]:
        host[offs[i]:offs[i] + lens[i]] = corpus.make(700000 + seed * N + i, 1, int(lens[i]), mix=ord("TXSBDIZR"[(i // 4) % 8]))
print(f"seed {seed}: {N} slices, {total / 1e9:.2f} GB", flush=True)

lib = find_libzstd_157(); assert lib is not None, "no libzstd 1.5.7 on this machine"
lib.ZSTD_compress.restype = ctypes.c_size_t
lib.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
def ref_frames(level, idx):
    def work(chunk):
        out = []; buf = ctypes.create_string_buffer(140000)
        for i in chunk:
            r = lib.ZSTD_compress(buf, 140000, host[offs[i]:].ctypes.data, int(lens[i]), level)
            out.append(buf.raw[:r])
        return out
    chunks = [idx[k::16] for k in range(16)]
    with ThreadPoolExecutor(16) as ex: parts = list(ex.map(work, chunks))
    res = {}
    for ch, pa in zip(chunks, parts):
        for i, f in zip(ch, pa): res[int(i)] = f
    return res

src = torch.from_numpy(host).cuda()
d_off = torch.from_numpy(offs).cuda(); d_len = torch.from_numpy(lens.astype(np.int32)).cuda()
def gpu_frames(env, level, n_ctx, piece, idx):
    for k in ("KMP_MATCH_V2", "KMP_FUSE"): os.environ.pop(k, None)
    os.environ.update(env)
    b = ZstdBatch(max_slices=n_ctx, max_slice_bytes=131072, ablations=bool(env))      # (the split-phase parser and the fused kernel live in the ablation build)
    sel = torch.from_numpy(np.asarray(idx, dtype=np.int64)).cuda()
    o_all, l_all = d_off[sel], d_len[sel]
    frames = {}
    for lo in range(0, len(idx), piece):
        hi = min(len(idx), lo + piece)
        dst, ooff, olen = b.compress(src, o_all[lo:hi], l_all[lo:hi], level=level, check=True)
        torch.cuda.synchronize()
        d, oo, ol = dst.cpu().numpy(), ooff.cpu().numpy(), olen.cpu().numpy()
        for k in range(hi - lo): frames[int(idx[lo + k])] = d[int(oo[k]):int(oo[k]) + int(ol[k])].tobytes()
        cap = torch.clamp(l_all[lo:hi], min=1)
        out, o2, l2, st = b.decompress(dst, ooff, olen, cap)
        torch.cuda.synchronize()
        assert int(st.abs().sum().item()) == 0 and bool((l2 == l_all[lo:hi]).all()), "decode status"
        oh, o2h = out.cpu().numpy(), o2.cpu().numpy()
        for k in range(0, hi - lo, 97):                        # every 97th slice byte for byte (the lengths and statuses of all)
            i = int(idx[lo + k]); assert oh[int(o2h[k]):int(o2h[k]) + int(lens[i])].tobytes() == host[offs[i]:offs[i] + lens[i]].tobytes(), ("round trip", i)
    b.close()
    return frames

all_idx = np.arange(N)
big_idx = np.nonzero(lens > 16384)[0]
bad_total = 0
# levels 5 .. 10: greedy / lazy / lazy2 (zstd_lazy.h): two of them a seed (the CPU side is slow), on a third of the slices; levels 9 and 10 are
# served above 16 KiB (8 bytes .. 16 KiB: another strategy there) and below 8 bytes (a raw block at any level)
third = np.arange(seed % 3, N, 3)
lazy_legs = tuple((f"level {lvl}", {}, lvl, len(third), len(third), third if lvl < 9 else third[(lens[third] > 16384) | (lens[third] < 8)]) for lvl in ((5, 8), (6, 9), (7, 10))[seed % 3])
for name, env, level, n_ctx, piece, idx in lazy_legs + (
        ("level 4, every size (up to 16 KiB its greedy row)", {}, 4, N, N, all_idx),
        ("level 3, team width 4, one batch", {}, 3, N, N, all_idx),
        ("level 3, batches of 6 000 (team width 8)", {}, 3, 6000, 6000, all_idx),
        ("level 3, split-phase parser", {"KMP_MATCH_V2": "2"}, 3, N, N, all_idx),
        ("level 3, fused kernel", {"KMP_FUSE": "1"}, 3, N, N, all_idx),
        ("level 1", {}, 1, N, N, all_idx),
        ("level 2", {}, 2, N, N, all_idx),
        ("level -1", {}, -1, N, N, all_idx),
        ("level -7", {}, -7, N, N, all_idx),
        ("level -200", {}, -200, 8000, 8000, all_idx),
        ("level 4 (slices above 16 KiB)", {}, 4, N, N, big_idx),
        ("level 4, batches of 4 000 (team width 8)", {}, 4, 4000, 4000, big_idx)):
    t0 = time.time()
    g = gpu_frames(env, level, n_ctx, piece, idx)
    r = ref_frames(level, idx)
    bad = [i for i in idx if g[int(i)] != r[int(i)]]
    bad_total += len(bad)
    print(f"{name}: {len(idx)} frames against libzstd 1.5.7, different: {len(bad)} {[(int(i), int(lens[i]), i % 4) for i in bad[:8]]}  ({time.time() - t0:.0f} s)", flush=True)
# ZstdCompressor(3, dictionary): a raw-content dictionary shared by the batch (text + JSON of the corpus + stress bytes), two sizes
from libzstd_ref import LibZstd
import threading
_tl = threading.local()
def with_dict(d, dic):
    if not hasattr(_tl, "z"): _tl.z = LibZstd()
    return _tl.z.compress_with_dict(d, dic, 3)
for k in ("KMP_MATCH_V2", "KMP_FUSE"): os.environ.pop(k, None)
dict_idx = np.arange(0, N, 2)
def trained_dict(cap):
    """a dictionary in zstd's own format from the box's library: ZDICT_trainFromBuffer over slices of this very batch"""
    pick = [int(i) for i in range(1, N, 7) if 256 <= lens[i] <= 16384][:3000]
    buf = b"".join(host[offs[i]:offs[i] + lens[i]].tobytes() for i in pick); sizes = (ctypes.c_size_t * len(pick))(*[int(lens[i]) for i in pick])
    out = ctypes.create_string_buffer(cap)
    lib.ZDICT_trainFromBuffer.restype = ctypes.c_size_t
    lib.ZDICT_trainFromBuffer.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
    n_ = lib.ZDICT_trainFromBuffer(out, cap, buf, sizes, len(pick))
    assert not lib.ZSTD_isError(n_), "ZDICT_trainFromBuffer failed"
    return out.raw[:n_]
for dsize in (16384, 65536, -8192, -110000):              # (negative: a trained dictionary of at most that many bytes)
    t0 = time.time()
    if dsize < 0:
        dic = trained_dict(-dsize)
    else:
      dic = (corpus.make(4242 + seed, 1, dsize // 2, mix=ord("T")).tobytes() + corpus.make(4343 + seed, 1, dsize // 4, mix=ord("X")).tobytes()
           + host[offs[5]:offs[5] + dsize // 4].tobytes()).ljust(dsize, b"\0")[:dsize]
    bdic = ZstdBatch(max_slices=len(dict_idx), max_slice_bytes=131072)
    sel = torch.from_numpy(dict_idx.astype(np.int64)).cuda()
    dst, ooff, olen = bdic.compress(src, d_off[sel], d_len[sel], dictionary=dic, check=True)
    torch.cuda.synchronize()
    d, oo, ol = dst.cpu().numpy(), ooff.cpu().numpy(), olen.cpu().numpy()
    g = {int(dict_idx[k]): d[int(oo[k]):int(oo[k]) + int(ol[k])].tobytes() for k in range(len(dict_idx))}
    chunks = [dict_idx[k::16] for k in range(16)]
    with ThreadPoolExecutor(16) as ex: parts = list(ex.map(lambda ch: [with_dict(host[offs[i]:offs[i] + lens[i]].tobytes(), dic) for i in ch], chunks))
    r = {}
    for ch, pa in zip(chunks, parts):
        for i, f in zip(ch, pa): r[int(i)] = f
    bad = [i for i in dict_idx if g[int(i)] != r[int(i)]]
    bad_total += len(bad)
    cap = torch.clamp(d_len[sel], min=1)
    out, o2, l2, st = bdic.decompress(dst, ooff, olen, cap, dictionary=torch.from_numpy(np.frombuffer(dic, dtype=np.uint8).copy()).cuda())
    torch.cuda.synchronize()
    assert int(st.abs().sum().item()) == 0 and bool((l2 == d_len[sel]).all()), "decode with dictionary: status"
    bdic.close()
    print(f"level 3 with a {'trained (zstd-format)' if dsize < 0 else 'raw-content'} dictionary of {len(dic)} bytes: {len(dict_idx)} frames against libzstd 1.5.7, different: {len(bad)} {[(int(i), int(lens[i]), i % 4) for i in bad[:8]]}  ({time.time() - t0:.0f} s)", flush=True)

# raw DEFLATE (all nine levels over two seeds) against this machine's zlib, and inflate of what came out
import zlib
dfl_idx = np.arange(0, N, 3)                                     # a third of the slices (the level-9 search is slow on both sides)
def zlib_frames(level, idx, wb=15, ml=8):
    def work(chunk):
        out = []
        for i in chunk:
            c = zlib.compressobj(level, zlib.DEFLATED, -wb, ml, zlib.Z_DEFAULT_STRATEGY)
            out.append(c.compress(host[offs[i]:offs[i] + lens[i]].tobytes()) + c.flush())
        return out
    chunks = [idx[k::16] for k in range(16)]
    with ThreadPoolExecutor(16) as ex: parts = list(ex.map(work, chunks))
    res = {}
    for ch, pa in zip(chunks, parts):
        for i, f in zip(ch, pa): res[int(i)] = f
    return res
for k in ("KMP_MATCH_V2", "KMP_FUSE"): os.environ.pop(k, None)
bd = ZstdBatch(max_slices=len(dfl_idx), max_slice_bytes=131072)
sel = torch.from_numpy(dfl_idx.astype(np.int64)).cuda()
for level in ((1, 6, 9) if seed % 2 else (2, 3, 4, 5, 7, 8)):      # (odd seeds: the fast, default and slowest rows; even seeds: the others)
    t0 = time.time()
    dst, ooff, olen = bd.deflate(src, d_off[sel], d_len[sel], level=level, check=True)
    torch.cuda.synchronize()
    d, oo, ol = dst.cpu().numpy(), ooff.cpu().numpy(), olen.cpu().numpy()
    g = {int(dfl_idx[k]): d[int(oo[k]):int(oo[k]) + int(ol[k])].tobytes() for k in range(len(dfl_idx))}
    r = zlib_frames(level, dfl_idx)
    bad = [i for i in dfl_idx if g[int(i)] != r[int(i)]]
    bad_total += len(bad)
    cap = torch.clamp(d_len[sel], min=1)
    out, o2, l2, st = bd.inflate(dst, ooff, olen, cap)
    torch.cuda.synchronize()
    assert int(st.abs().sum().item()) == 0 and bool((l2 == d_len[sel]).all()), "inflate status"
    oh, o2h = out.cpu().numpy(), o2.cpu().numpy()
    for k in range(0, len(dfl_idx), 53):
        i = int(dfl_idx[k]); assert oh[int(o2h[k]):int(o2h[k]) + int(lens[i])].tobytes() == host[offs[i]:offs[i] + lens[i]].tobytes(), ("inflate round trip", i)
    print(f"raw DEFLATE level {level}: {len(dfl_idx)} streams against zlib {zlib.ZLIB_RUNTIME_VERSION}, different: {len(bad)} {[(int(i), int(lens[i]), i % 4) for i in bad[:8]]}  ({time.time() - t0:.0f} s)", flush=True)
# deflateInit2's windowBits / memLevel: four random settings per seed, alternately on a context for slices up to 64 KiB (the sort + wave-wide
# parse kernels) and on the one above (the same kernels, 64 KiB spans)
import random as _random
prng = _random.Random(seed * 7919 + 5)
small_idx = np.array([i for i in dfl_idx if lens[i] <= 65536], dtype=np.int64)
bs = ZstdBatch(max_slices=max(1, len(small_idx)), max_slice_bytes=65536)
for t in range(4):
    t0 = time.time()
    level, wb, ml = prng.randrange(1, 10), prng.randrange(9, 16), prng.randrange(1, 10)
    b_, idx_ = (bs, small_idx) if t % 2 == 0 else (bd, dfl_idx[::2])
    if len(idx_) == 0: continue
    sel_ = torch.from_numpy(idx_.astype(np.int64)).cuda()
    dst, ooff, olen = b_.deflate(src, d_off[sel_], d_len[sel_], level=level, check=True, window_bits=wb, mem_level=ml)
    torch.cuda.synchronize()
    d, oo, ol = dst.cpu().numpy(), ooff.cpu().numpy(), olen.cpu().numpy()
    g = {int(idx_[k]): d[int(oo[k]):int(oo[k]) + int(ol[k])].tobytes() for k in range(len(idx_))}
    r = zlib_frames(level, idx_, wb, ml)
    bad = [i for i in idx_ if g[int(i)] != r[int(i)]]
    bad_total += len(bad)
    print(f"raw DEFLATE level {level} windowBits {wb} memLevel {ml} ({'<= 64 KiB kernels' if t % 2 == 0 else 'a context for slices above 64 KiB: spans'}): {len(idx_)} streams against zlib {zlib.ZLIB_RUNTIME_VERSION}, different: {len(bad)} {[(int(i), int(lens[i]), i % 4) for i in bad[:8]]}  ({time.time() - t0:.0f} s)", flush=True)
bs.close()
bd.close()
print("FUZZ OK" if bad_total == 0 else f"FUZZ FOUND {bad_total} DIFFERENCES")
sys.exit(0 if bad_total == 0 else 1)
