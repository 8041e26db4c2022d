"""zstd levels 5 .. 10 on the GPU: ragged slices against the oracle (test infrastructure), then the BASELINE mix timed at each level."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, random
import helpers
from kompressor_amd import corpus
from kompressor_amd.batch import ZstdBatch
o = helpers.oracle(); rng = random.Random(5); dev = torch.device("cuda", 0)
N = 600
datas = [corpus.make(rng.randrange(1 << 30), 1, rng.choice([rng.randrange(1, 16385), rng.randrange(16385, 131073), 65536, 131072, 16384, 16385]), mix=ord(rng.choice("TXSBDIZR"))).tobytes() for _ in range(N)]
stride = 131072 + 512
host = np.zeros(N * stride + 64, dtype=np.uint8)
for k, p in enumerate(datas): host[k * stride:k * stride + len(p)] = np.frombuffer(p, dtype=np.uint8)
src = torch.from_numpy(host).to(dev); offs = (torch.arange(N, dtype=torch.int64) * stride).to(dev); lens = torch.tensor([len(p) for p in datas], dtype=torch.int32).to(dev)
b = ZstdBatch(max_slices=N, max_slice_bytes=131072)
for lvl in (5, 6, 7, 8, 9, 10):
    dst, ooff, olen = b.compress(src, offs, lens, level=lvl)
    torch.cuda.synchronize()
    d, oo, ol = dst.cpu().numpy(), ooff.cpu().numpy(), olen.cpu().numpy()
    bad = 0
    for k, p in enumerate(datas):
        w = o.compress_lazy(p, lvl) or b""
        if d[int(oo[k]):int(oo[k]) + int(ol[k])].tobytes() != w: bad += 1
    try: st = b.status()
    except Exception as e: st = str(e)[:80]
    print(f"level {lvl}: {N} ragged slices against the oracle, different: {bad}; status: {st}", flush=True)
b.close()
n, S = 16384, 65536
src = torch.from_numpy(corpus.make(0, n, S)).to(dev)
b = ZstdBatch(max_slices=n, max_slice_bytes=S)
in_off = torch.arange(n, dtype=torch.int64, device=dev) * S; in_len = torch.full((n,), S, dtype=torch.int32, device=dev)
dst = torch.empty(n * b.out_stride + 64, dtype=torch.uint8, device=dev); out_off = torch.arange(n, dtype=torch.int64, device=dev) * b.out_stride; out_len = torch.zeros(n, dtype=torch.int32, device=dev)
for lvl in (3, 5, 6, 7, 9, 10):
    b.compress(src, in_off, in_len, dst, out_off, out_len, level=lvl); torch.cuda.synchronize()
    t0 = time.perf_counter(); b.compress(src, in_off, in_len, dst, out_off, out_len, level=lvl); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"level {lvl}: {n} x 64 KiB in {dt * 1e3:.1f} ms = {n * S / dt / 1e9:.2f} GB/s, ratio {n * S / float(out_len.sum().item()):.4f}", flush=True)
b.close()
