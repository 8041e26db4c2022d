"""Differential fuzz of the DEFLATE kernels' settings on the GPU box: ragged slices of stress inputs (tools/fuzzgen.c) and corpus classes,
random (level, windowBits, memLevel) per round, against this machine's zlib on the host cores -- slices up to 64 KiB on a context that
runs the sort + wave-wide parse kernels, slices up to 400 KiB on one that takes them through the same kernels in 64 KiB spans (sizes near the
ends of zlib's window buffer among them: k * w_size + 2 * w_size - 262 .. + 262, where fill_window slides at the end of the input).
usage: python tools/fuzz_gpu_deflate.py [seed] [rounds] [slices per round]"""
import ctypes, os, subprocess, sys, time, zlib
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from kompressor_amd import corpus
from kompressor_amd.batch import ZstdBatch

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
N = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
out_dir = os.path.join(ROOT, "gpurun_out"); os.makedirs(out_dir, exist_ok=True)
so = os.path.join(out_dir, "libfuzzgen.so")
subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "fuzzgen.c")], check=True)
FG = ctypes.CDLL(so); FG.fuzz_fill.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64]
rng = np.random.default_rng(seed * 977 + 13)
small = ZstdBatch(max_slices=N, max_slice_bytes=65536)
big = ZstdBatch(max_slices=N // 8 + 1, max_slice_bytes=400000 + 64)
bad_total = 0
for r in range(rounds):
    t0 = time.time()
    level, wb, ml = int(rng.integers(1, 10)), int(rng.integers(9, 16)), int(rng.integers(1, 10))
    if r % 4 >= 2: wb, ml = 15, 8                  # half of the rounds at zlib's default window and memLevel (one of each size class)
    W = 1 << wb
    long_round = r % 2 == 1
    n = N // 8 if long_round else N
    cap = 400000 if long_round else 65536
    lens = np.where(rng.random(n) < 0.1, rng.integers(0, 600, n), rng.integers(0, cap + 1, n)).astype(np.int64)
    # a third of the slices end where zlib's window buffer is nearly full: around k * w_size + 2 * w_size - 262
    for i in range(0, n, 3):
        kmax = max(0, (cap - 2 * W) // W)
        lens[i] = min(cap, max(0, int(rng.integers(0, kmax + 1)) * W + 2 * W - 262 + int(rng.integers(-8, 270))))
    offs = np.concatenate([[0], np.cumsum(lens[:-1])]).astype(np.int64)
    host = np.zeros(int(lens.sum()) + 64, dtype=np.uint8)
    for i in range(n):
        if lens[i] and i % 4 != 3:
            FG.fuzz_fill(host[offs[i]:].ctypes.data, int(lens[i]), seed * 1000003 + r * 100003 + i)
        elif lens[i]:
            host[offs[i]:offs[i] + lens[i]] = corpus.make(800000 + seed * 7919 + r * N + i, 1, int(lens[i]), mix=ord("TXSBDIZR"[(i // 4) % 8]))
    b = big if long_round else small
    dst, ooff, olen = b.deflate(torch.from_numpy(host).cuda(), torch.from_numpy(offs).cuda(), torch.from_numpy(lens.astype(np.int32)).cuda(),
                                level=level, window_bits=wb, mem_level=ml, check=True)
    torch.cuda.synchronize()
    d, oo, ol = dst.cpu().numpy(), ooff.cpu().numpy(), olen.cpu().numpy()

    def work(chunk):
        out = []
        for i in chunk:
            c = zlib.compressobj(level, zlib.DEFLATED, -wb, ml, 0)
            out.append(c.compress(host[offs[i]:offs[i] + lens[i]].tobytes()) + c.flush())
        return out
    chunks = [range(k, n, 16) for k in range(16)]
    with ThreadPoolExecutor(16) as ex: parts = list(ex.map(work, chunks))
    bad = []
    for ch, pa in zip(chunks, parts):
        for i, f in zip(ch, pa):
            if d[int(oo[i]):int(oo[i]) + int(ol[i])].tobytes() != f: bad.append(i)
    bad_total += len(bad)
    print(f"round {r}: level {level} windowBits {wb} memLevel {ml} ({'slices up to 400 000 bytes: 64 KiB spans' if long_round else 'sort + parse kernels, <= 64 KiB'}): {n} streams against zlib {zlib.ZLIB_RUNTIME_VERSION}, "
          f"different: {len(bad)} {[(int(i), int(lens[i]), i % 4) for i in bad[:6]]}  ({time.time() - t0:.0f} s)", flush=True)
small.close(); big.close()
print("FUZZ OK" if bad_total == 0 else f"FUZZ FOUND {bad_total} DIFFERENCES")
sys.exit(0 if bad_total == 0 else 1)
