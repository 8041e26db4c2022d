"""Development aid: configs[1] slices [first, first + count) through both parsers on the emulator."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(__file__))
import numpy as np
from emu_match_cmp import compare
import helpers
from kompressor_amd import corpus
first, count = int(sys.argv[1]), int(sys.argv[2]); G = int(sys.argv[3]) if len(sys.argv) > 3 else 4
S = 65536
buf = corpus.make(first, count, S)
st = (ctypes.c_ulonglong * 64)()
helpers.emu().emu_stats(st, 1)
bad = compare([buf[k * S:(k + 1) * S].tobytes() for k in range(count)], G, max(1, count * G // 64 // 2), f"cfg1[{first}..]")
helpers.emu().emu_stats(st, 1)
print("BAD" if bad else "all equal", bad, "stats", list(st)[:8])
