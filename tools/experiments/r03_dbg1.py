"""GPU debug: frames of the old and the new level-3 parser side by side, which slices differ, are the differences stable."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
from kompressor_amd import corpus
from kompressor_amd.batch import ZstdBatch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
S = 65536
src = torch.empty(n * S, dtype=torch.uint8, device="cuda")
for c in range(0, n, 4096): src[c * S:(c + 4096) * S] = torch.from_numpy(corpus.make(c, min(4096, n - c), S)).cuda()
in_off = torch.arange(n, dtype=torch.int64, device="cuda") * S
in_len = torch.full((n,), S, dtype=torch.int32, device="cuda")
def run(v2, extra={}):
    os.environ["KMP_MATCH_V2"] = str(v2); os.environ["KMP_ZSTD_AUTOTUNE"] = "0"
    for k_, v_ in extra.items(): os.environ[k_] = v_
    b = ZstdBatch(max_slices=n, max_slice_bytes=S)
    outs = []
    for rep in range(2):
        dst, ooff, olen = b.compress(src, in_off, in_len)
        torch.cuda.synchronize()
        outs.append((dst.clone(), olen.clone()))
    st = b.out_stride
    b.close()
    for k_ in extra: del os.environ[k_]
    return outs, st
ref, st = run(0)
for label, v2, extra in [("v2", 1, {}), ("v2 plain stores", 1, {"KMP_MATCH_FLAGS": "4"})]:
    new, _ = run(v2, extra)
    for rep in range(2):
        d_len = (new[rep][1] != ref[0][1]).nonzero().flatten().cpu().numpy()
        a = new[rep][0][: n * st].view(n, st); r = ref[0][0][: n * st].view(n, st)
        # compare only the frame bytes
        idx = torch.arange(st, device="cuda")[None, :] < ref[0][1][:, None]
        diff = ((a != r) & idx).any(dim=1).nonzero().flatten().cpu().numpy()
        print(label, "run", rep, "len differs:", len(d_len), "bytes differ:", len(diff), "first:", diff[:12], "classes:", "".join(corpus.slice_class(int(i)) for i in diff[:40]))
    same = torch.equal(new[0][0], new[1][0])
    print(label, "two runs identical:", same)
