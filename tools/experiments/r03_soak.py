"""Soak: ragged slices of every class and length through the level-3 path in its three forms of this round (team width 4, the
per-batch width 8, the split-phase parser) -- all frames must agree, a sample of them with the oracle, and they must decode."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
import helpers
from kompressor_amd import corpus
from kompressor_amd.batch import ZstdBatch
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = 24000
lens = np.where(rng.random(N) < 0.15, rng.integers(0, 300, N), rng.integers(0, 131073, N)).astype(np.int64)
lens[:8] = [0, 1, 7, 8, 9, 131072, 131071, 65536]
offs = np.concatenate([[0], np.cumsum(lens[:-1])]).astype(np.int64)
total = int(lens.sum())
host = np.empty(total + 64, dtype=np.uint8)
pos = 0
for i in range(0, N, 500):                       # 500 slices share a class piece (cheap to generate), cut raggedly
    m = min(500, N - i); need = int(lens[i:i + m].sum())
    blob = corpus.make(90000 + i, 1, max(need, 1), mix=ord("TXSBDIZR"[(i // 500) % 8]))
    host[pos:pos + need] = blob[:need]; pos += need
src = torch.from_numpy(host).cuda()
d_off = torch.from_numpy(offs).cuda(); d_len = torch.from_numpy(lens.astype(np.int32)).cuda()
def run(v2, n_ctx, pieces):
    os.environ["KMP_MATCH_V2"] = str(v2)
    b = ZstdBatch(max_slices=n_ctx, max_slice_bytes=131072)
    frames = []
    for lo in range(0, N, pieces):
        hi = min(N, lo + pieces)
        dst, ooff, olen = b.compress(src, d_off[lo:hi], d_len[lo:hi], check=True)
        torch.cuda.synchronize()
        d, oo, ol = dst.cpu().numpy(), ooff.cpu().numpy(), olen.cpu().numpy()
        frames += [d[int(oo[i]):int(oo[i]) + int(ol[i])].tobytes() for i in range(hi - lo)]
    # decode the last piece back
    cap = torch.from_numpy(np.maximum(lens[lo:hi], 1).astype(np.int32)).cuda()
    out, o2, l2, st = b.decompress(dst, ooff, olen, cap)
    torch.cuda.synchronize()
    assert int(st.abs().sum().item()) == 0 and (l2.cpu().numpy() == lens[lo:hi]).all()
    b.close()
    return frames
a = run(0, N, N)            # one batch of 24 000: team width 4
b8 = run(0, 6000, 6000)     # four batches of 6 000: the library picks width 8
c = run(2, N, N)            # the split-phase parser
bad = [i for i in range(N) if not (a[i] == b8[i] == c[i])]
print("slices", N, "bytes", total, "disagreements", len(bad), bad[:10])
o = helpers.oracle()
pick = rng.choice(N, 300, replace=False)
wrong = [int(i) for i in pick if a[i] != o.compress(host[offs[i]:offs[i] + lens[i]].tobytes())]
print("sample of 300 against the oracle: wrong", wrong[:10])
assert not bad and not wrong
print("SOAK OK")
