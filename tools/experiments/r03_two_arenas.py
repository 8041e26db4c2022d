"""Does a second arena, allocated while the first is held, land on better memory when the first is slow?  Six fresh processes."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = f"""
import sys; sys.path.insert(0, {ROOT!r})
import torch
from kompressor_amd.batch import ZstdBatch
torch.cuda.init()
b1 = ZstdBatch(max_slices=65536, max_slice_bytes=65536)
r1 = b1.table_rates()
b2 = ZstdBatch(max_slices=65536, max_slice_bytes=65536)
r2 = b2.table_rates()
b1.close()
b3 = ZstdBatch(max_slices=65536, max_slice_bytes=65536)
r3 = b3.table_rates()
print("first", r1, "second (first held)", r2, "third (first freed, second held)", r3)
"""
for i in range(6):
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, KMP_PLACE_VERBOSE="0"))
    print(r.stdout.strip() or r.stderr[-500:], flush=True)
