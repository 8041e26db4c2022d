"""Development aid: the split-phase parser (zstd_match2.h) against zstd_match.h on the CPU emulator, sequence for sequence."""
import ctypes, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import helpers
from kompressor_amd import corpus

def run(which, datas, G, nblocks):
    e = helpers.emu()
    n = len(datas)
    lens = np.array([len(d) for d in datas], dtype=np.uint32)
    offs = np.zeros(n, dtype=np.uint64); pos = 0
    for i, d in enumerate(datas): offs[i] = pos; pos += len(d)
    buf = np.zeros(pos + 64, dtype=np.uint8)
    for i, d in enumerate(datas): buf[int(offs[i]):int(offs[i]) + len(d)] = np.frombuffer(d, dtype=np.uint8)
    S = max([len(d) for d in datas] + [64])
    seq_cap = (S // 4 + 8 + 15) & ~15; lit_cap = S + 64
    seqs = np.zeros(n * seq_cap * 2, dtype=np.uint32); lits = np.zeros(n * lit_cap, dtype=np.uint8); meta = np.zeros(n * 8, dtype=np.uint32)
    r = getattr(e, which)(helpers._vp(buf), helpers._vp(offs), helpers._vp(lens), n, G, nblocks, helpers._vp(seqs), seq_cap, helpers._vp(lits), lit_cap, helpers._vp(meta), 7)
    assert r == 0, (which, r)
    return seqs.reshape(n, seq_cap, 2), meta.reshape(n, 8)

def compare(datas, G=4, nblocks=2, tag=""):
    s1, m1 = run("emu_zstd_match", datas, G, nblocks)
    s2, m2 = run("emu_zstd_match2", datas, G, nblocks)
    bad = 0
    for i in range(len(datas)):
        ns = int(m1[i, 0])
        if not (m1[i, :6] == m2[i, :6]).all() or not (s1[i, :ns] == s2[i, :ns]).all():
            bad += 1
            d = np.nonzero((s1[i, :max(ns, int(m2[i, 0]))] != s2[i, :max(ns, int(m2[i, 0]))]).any(axis=1))[0]
            first = int(d[0]) if len(d) else -1
            print(f"{tag} slice {i} len {len(datas[i])}: meta {m1[i, :6]} vs {m2[i, :6]}; first differing sequence {first}")
            if first >= 0:
                for j in range(max(0, first - 1), first + 2):
                    print("   ", j, [hex(x) for x in s1[i, j]], [hex(x) for x in s2[i, j]])
    return bad

if __name__ == "__main__":
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    bad = 0
    S = 65536
    buf = corpus.make(0, 32, S)
    bad += compare([buf[k * S:(k + 1) * S].tobytes() for k in range(32)], G, 2, "mix64k")
    sp = helpers.special_inputs()
    bad += compare(list(sp.values()), G, 2, "special")
    rng = np.random.default_rng(5)
    ds = []
    for t in range(48):
        nn = int(rng.integers(0, 70000)) if t % 3 else int(rng.integers(0, 300))
        cls = "TXSBDIZR"[t % 8]
        ds.append(corpus.make(9000 + t, 1, max(nn, 1), mix=ord(cls)).tobytes()[:nn])
    bad += compare(ds, G, 3, "ragged")
    big = corpus.make(500, 8, 131072)
    bad += compare([big[k * 131072:(k + 1) * 131072].tobytes() for k in range(8)], G, 1, "128k")
    print("BAD" if bad else "all equal", bad)
