import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, random, helpers
from kompressor_amd import corpus
from kompressor_amd.batch import ZstdBatch
o = helpers.oracle(); rng = random.Random(6); dev = torch.device("cuda", 0)
N = 500
datas = [corpus.make(rng.randrange(1 << 30), 1, rng.choice([rng.randrange(1, 16385), rng.randrange(16385, 131073), 16384, 16385, 7, 8]), mix=ord(rng.choice("TXSBDIZR"))).tobytes() for _ in range(N)]
stride = 131072 + 512
host = np.zeros(N * stride + 64, dtype=np.uint8)
for k, p in enumerate(datas): host[k * stride:k * stride + len(p)] = np.frombuffer(p, dtype=np.uint8)
src = torch.from_numpy(host).to(dev); offs = (torch.arange(N, dtype=torch.int64) * stride).to(dev); lens = torch.tensor([len(p) for p in datas], dtype=torch.int32).to(dev)
b = ZstdBatch(max_slices=N, max_slice_bytes=131072)
dst, ooff, olen = b.compress(src, offs, lens, level=4); torch.cuda.synchronize()
d, oo, ol = dst.cpu().numpy(), ooff.cpu().numpy(), olen.cpu().numpy()
bad = 0
for k, p in enumerate(datas):
    w = o.compress_lazy(p, 4) if len(p) <= 16384 else o.compress_level(p, 4)
    if d[int(oo[k]):int(oo[k]) + int(ol[k])].tobytes() != w: bad += 1
print("level 4, all sizes up to 128 KiB:", N, "slices against the oracle, different:", bad, "status", b.status())
