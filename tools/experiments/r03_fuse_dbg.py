"""debug: k_zstd_l3_fused against the two-kernel pipeline, slice by slice"""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch

def run(fuse, n, S=65536):
    code = f"""
import os, sys, numpy as np, torch
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
from kompressor_amd import corpus
from kompressor_amd.batch import ZstdBatch
n, S = {n}, {S}
b = ZstdBatch(max_slices=n, max_slice_bytes=S)
src = torch.from_numpy(corpus.make(0, n, S)).cuda()
in_off = torch.arange(n, dtype=torch.int64, device="cuda") * S
in_len = torch.full((n,), S, dtype=torch.int32, device="cuda")
dst, ooff, olen = b.compress(src, in_off, in_len)
torch.cuda.synchronize()
np.save("/tmp/fz_{fuse}_len.npy", olen.cpu().numpy()); np.save("/tmp/fz_{fuse}_dst.npy", dst.cpu().numpy()); print(b.out_stride)
"""
    env = dict(os.environ, KMP_FUSE=str(fuse))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    if r.returncode: print(r.stderr[-2000:])
    return int(r.stdout.strip().splitlines()[-1])

for n in (16, 64, 1024):
    st = run(0, n); run(1, n)
    l0, l1 = np.load("/tmp/fz_0_len.npy"), np.load("/tmp/fz_1_len.npy")
    d0, d1 = np.load("/tmp/fz_0_dst.npy"), np.load("/tmp/fz_1_dst.npy")
    bad = [i for i in range(n) if l0[i] != l1[i] or not np.array_equal(d0[i*st:i*st+l0[i]], d1[i*st:i*st+l0[i]])]
    print("n", n, "differing slices", len(bad), bad[:24])
    for i in bad[:6]:
        a, b_ = d0[i*st:i*st+l0[i]], d1[i*st:i*st+l1[i]]
        m = min(len(a), len(b_)); k = int(np.argmax(a[:m] != b_[:m])) if (a[:m] != b_[:m]).any() else m
        print("  slice", i, "class", "TXSBTDTBIXTSZBTR"[i % 16], "len", l0[i], l1[i], "first diff at", k, a[:16].tobytes().hex(), b_[:16].tobytes().hex())
