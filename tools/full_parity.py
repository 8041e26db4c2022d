#!/usr/bin/env python3
"""Full-size parity run on the GPU box (TEST TOOL): every one of the 65 536 x 64 KiB slices of BASELINE configs[1] is
compressed on the GPU and by the binary libzstd 1.5.7 on the host threads (the reference's call: ZSTD_compress2 with
parameter 100 = level), and the frames are compared byte for byte; then the GPU decodes its frames back.  Levels 3, 1, 2.

    python tools/full_parity.py [--slices 65536] > gpurun_out/full_parity.txt
"""
import argparse
import ctypes
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slices", type=int, default=65536)
    args = ap.parse_args()
    import torch
    from kompressor_amd import corpus
    from kompressor_amd.batch import ZstdBatch
    from libzstd_ref import find_libzstd_157
    lib = find_libzstd_157()
    assert lib is not None and lib.ZSTD_versionNumber() == 10507
    lib.ZSTD_createCCtx.restype = ctypes.c_void_p
    lib.ZSTD_CCtx_setParameter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    lib.ZSTD_compress2.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    lib.ZSTD_compress2.restype = ctypes.c_size_t
    n, S = args.slices, 65536
    host = np.empty(n * S, dtype=np.uint8)
    corpus.fill(host, 0, n, S, corpus.MIX_CONFIG1)
    dev = torch.device("cuda", 0)
    src = torch.from_numpy(host).to(dev)
    in_off = torch.arange(n, dtype=torch.int64, device=dev) * S
    in_len = torch.full((n,), S, dtype=torch.int32, device=dev)
    b = ZstdBatch(max_slices=n, max_slice_bytes=S, device=0)
    stride = b.out_stride
    cores = min(os.cpu_count() or 1, 64)
    for level in (3, 1, 2):
        dst, ooff, olen = b.compress(src, in_off, in_len, level=level)
        torch.cuda.synchronize()
        g = dst.cpu().numpy(); gl = olen.cpu().numpy().astype(np.int64)
        cpu = np.zeros(n * stride, dtype=np.uint8); cl = np.zeros(n, dtype=np.int64)
        per = (n + cores - 1) // cores

        def run(t):
            cctx = lib.ZSTD_createCCtx()
            lib.ZSTD_CCtx_setParameter(cctx, 100, level)
            for i in range(t * per, min(n, (t + 1) * per)):
                cl[i] = lib.ZSTD_compress2(cctx, cpu.ctypes.data + i * stride, stride, host.ctypes.data + i * S, S)

        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(run, range(cores)))
        dt = time.perf_counter() - t0
        same_len = int((gl == cl).sum())
        bad = 0
        for i in range(n):
            if gl[i] != cl[i] or not np.array_equal(g[i * stride:i * stride + gl[i]], cpu[i * stride:i * stride + cl[i]]):
                bad += 1
        cap = torch.full((n,), S, dtype=torch.int32, device=dev)
        back, _, l2, st = b.decompress(dst, ooff, olen, cap, out_off=in_off)
        torch.cuda.synchronize()
        rt = bool(int(st.abs().sum().item()) == 0 and torch.equal(back[: n * S], src))
        print(f"level {level}: {n} slices x {S} B; frames identical to libzstd 1.5.7: {n - bad} of {n} (equal lengths: {same_len}); "
              f"total frame bytes GPU {int(gl.sum())} / libzstd {int(cl.sum())}; GPU decode of the GPU frames restores the input: {rt}; "
              f"libzstd on {cores} host threads: {n * S / dt / 1e9:.2f} GB/s", flush=True)
    # raw DEFLATE level 6 (configs[4]) against the host zlib, every slice
    import zlib
    dst, ooff, olen = b.deflate(src, in_off, in_len)
    torch.cuda.synchronize()
    g = dst.cpu().numpy(); go = ooff.cpu().numpy().astype(np.int64); gl = olen.cpu().numpy().astype(np.int64)
    bad = np.zeros(cores, dtype=np.int64); tot = np.zeros(cores, dtype=np.int64)
    per = (n + cores - 1) // cores

    def runz(t):
        for i in range(t * per, min(n, (t + 1) * per)):
            c = zlib.compressobj(6, zlib.DEFLATED, -15, 8, 0)
            ref = c.compress(host[i * S:(i + 1) * S].tobytes()) + c.flush()
            tot[t] += len(ref)
            if len(ref) != gl[i] or ref != g[go[i]:go[i] + gl[i]].tobytes():
                bad[t] += 1

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(runz, range(cores)))
    dt = time.perf_counter() - t0
    cap = torch.full((n,), S, dtype=torch.int32, device=dev)
    back, _, l2, st = b.inflate(dst, ooff, olen, cap, out_off=in_off)
    torch.cuda.synchronize()
    rt = bool(int(st.abs().sum().item()) == 0 and torch.equal(back[: n * S], src))
    print(f"raw DEFLATE level 6: {n} slices x {S} B; streams identical to zlib {zlib.ZLIB_RUNTIME_VERSION}: {n - int(bad.sum())} of {n}; "
          f"total stream bytes GPU {int(gl.sum())} / zlib {int(tot.sum())}; GPU inflate of the GPU streams restores the input: {rt}; "
          f"zlib on {cores} host threads (Python): {n * S / dt / 1e9:.2f} GB/s", flush=True)
    b.close()


if __name__ == "__main__":
    main()
