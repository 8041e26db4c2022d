"""Counts what the level-3 parser's body does per wave on the CPU emulator (KX_STAT slots): development aid."""
import ctypes, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import helpers
from kompressor_amd import corpus

def run(first, nsl, S, G, which="emu_zstd_match"):
    e = helpers.emu()
    buf = corpus.make(first, nsl, S)
    offs = (np.arange(nsl, dtype=np.uint64) * S); lens = np.full(nsl, S, dtype=np.uint32)
    seq_cap = (S // 4 + 8 + 15) & ~15; lit_cap = S + 64
    seqs = np.zeros(nsl * seq_cap * 8, dtype=np.uint8); lits = np.zeros(nsl * lit_cap, dtype=np.uint8); meta = np.zeros(nsl * 8, dtype=np.uint32)
    st = (ctypes.c_ulonglong * 64)()
    e.emu_stats(st, 1)
    nblocks = nsl * G // 64
    r = getattr(e, which)(helpers._vp(buf), helpers._vp(offs), helpers._vp(lens), nsl, G, nblocks, helpers._vp(seqs), seq_cap, helpers._vp(lits), lit_cap, helpers._vp(meta), 7)
    assert r == 0, r
    e.emu_stats(st, 1)
    return list(st), meta.reshape(nsl, 8), nblocks

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "emu_zstd_match"
    nsl = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    st, meta, nb = run(0, nsl, 65536, 4, which)
    print("waves", nb, "slices", nsl, "nbSeq mean", meta[:, 0].mean(), "max", meta[:, 0].max())
    names = ["iters", "repcheck blk", "search blk", "match blk", "ext rounds", "team steps", "team seqs", "back rounds"]
    for i, nm in enumerate(names): print(f"{nm:14s} total {st[i]:9d}  per wave {st[i] / nb:10.1f}  per slice {st[i] / nsl:9.1f}")
    for i in range(8, 40):
        if st[i]: print(f"stat[{i}] total {st[i]:9d} per wave {st[i] / nb:10.1f} per slice {st[i] / nsl:9.1f}")
