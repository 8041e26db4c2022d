#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its configs[1] workload:
zstd level-3 compression of a batch of 65 536 x 64 KiB slices (seeded synthetic
"Silesia-like" mix, kompressor_amd/csrc/corpus.c) per GPU, inputs resident in
HBM when the timed region starts.  One step = one pass of the whole hot path
(match kernel + entropy/frame kernel) over the batch.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config 1|3] [--slices M]

N > 1: one rank per GPU over RCCL.  Started by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the
environment) the process is a rank; started plainly with --gpus N > 1 it starts those N ranks itself as child
processes (before anything touches a GPU) and exits with their code.  Every rank owns its own block of slices
(weak scaling, no collective on the data path); a step compresses the block, packs the frames densely and
all-gathers the frame size table.  With N > 1 a second figure is measured for the same work with the batch starting
and ending on rank 0: root scatter of the slices + the step + gather-v of the frames inside the timed region
(key "with_scatter_gather").  --config 3 = BASELINE configs[3]'s per-GPU share: 131 072 slices per GPU, text and
binary classes alternating.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SLICE = 65536
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def host_cores():
    """What this process may use of the host: logical CPUs, the scheduler affinity, the cgroup's CPU quota (a GPU box hands
    one GPU's share of the host to a job), and the thread count the CPU legs use = the smallest of them."""
    nproc = os.cpu_count() or 1
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:
        aff = nproc
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except Exception:
            continue
    use = min(nproc, aff)
    if quota:
        use = max(1, min(use, int(quota + 0.5)))
    return {"nproc": nproc, "affinity": aff, "cgroup_quota": None if quota is None else round(quota, 2), "threads_used": use}


def build_cpu_bench():
    """oracle/cpu_bench.c -> oracle/_build/libcpubench.so (the GPU box has gcc; __graft_entry__.build() builds it too)."""
    src = os.path.join(ROOT, "oracle", "cpu_bench.c")
    out = os.path.join(ROOT, "oracle", "_build", "libcpubench.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-fvisibility=hidden", "-o", out, src, "-ldl", "-lpthread"], check=True)
    return out


def cpu_baseline(host, n_slices, frames_expected, sample=None, level=3):
    """Times the reference's arithmetic on the host cores (oracle/cpu_bench.c: plain pthreads, one context per thread): a
    binary libzstd 1.5.7 if this machine has one (kind "reference"), else the oracle's C restatement (kind "port").  All the
    cores this job may use: 1 warm-up + 5 passes over the first `sample` slices of the batch, the median pass; and one thread
    the same way on an eighth of the sample.  Bounded: a few tens of seconds."""
    import statistics
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    hc = host_cores()
    cores = hc["threads_used"]
    if sample is None:
        sample = min(n_slices, max(64, (1 << 30) // SLICE))          # 1 GiB: 16 384 slices of 64 KiB
    sample = min(sample, host.size // SLICE)
    lib = ctypes.CDLL(build_cpu_bench())
    lib.cpubench_set_zstd_level(int(level))
    lib.cpubench_zstd_l3.restype = ctypes.c_int
    lib.cpubench_zstd_l3.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_int,
                                     ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    zpath = None
    try:
        from libzstd_ref import find_libzstd_157
        z = find_libzstd_157()
        zpath = z._path if z is not None else None
    except Exception:
        zpath = None
    if zpath:
        kind = "reference"
        label = f"libzstd 1.5.7 ({os.path.basename(zpath)}) " + ("ZSTD_compress2" if SLICE <= 131072 else f"ZSTD_compressStream2(e_end), {max(8192, SLICE // 10)}-byte output slices") + f" level {level}"
        kpath = None
    else:
        import helpers
        kind = "port"
        label = "oracle/zstd_l3_ref.c (C restatement)"
        kpath = helpers.build_oracle()

    def run(threads, cnt, passes):
        secs = (ctypes.c_double * passes)()
        fb, err = ctypes.c_uint64(0), ctypes.c_uint64(0)
        rc = lib.cpubench_zstd_l3(zpath.encode() if zpath else None, kpath.encode() if kpath else None, host.ctypes.data, cnt, SLICE, threads, passes,
                                  secs, ctypes.byref(fb), ctypes.byref(err))
        if rc != 0:
            raise RuntimeError(f"cpubench_zstd_l3 failed ({rc})")
        return list(secs), fb.value, err.value

    passes = 6
    secs, fb, err = run(cores, sample, passes)
    med = statistics.median(secs[1:])
    one_n = max(16, sample // 8)
    secs1, _, err1 = run(1, one_n, passes)
    med1 = statistics.median(secs1[1:])
    ok = (frames_expected is None) or (fb == frames_expected)
    all_gbs, one_gbs = sample * SLICE / med / 1e9, one_n * SLICE / med1 / 1e9
    note = f" [{err + err1} CPU calls returned an error]" if (err + err1) else ""
    return {"value": round(all_gbs, 4), "unit": "GB/s", "cores": cores, "kind": kind,
            "single_thread": {"value": round(one_gbs, 4), "unit": "GB/s", "sample_slices": one_n},
            "per_thread_MBps_at_full_load": round(all_gbs * 1e3 / cores, 1),
            "host": hc,
            "passes_s": [round(x, 3) for x in secs],
            "sample": f"first {sample} slices of the same batch, {label}, {cores} pthreads (oracle/cpu_bench.c), one context per thread, "
                      f"1 warm-up + {passes - 1} passes, median {med:.2f} s; total frame bytes {'match' if ok else 'DIFFER from'} the GPU's{note}"}


def streaming_abi_figure(host, n_slices):
    """What a Kotlin caller can bind today: ZstdCompressor(3).transform(bytes) per slice through kmp_zstd_compress_stream (host
    memory in, host memory out, the reference's one-shot driver loop).  One context taking slices in turn, then 64 contexts on
    64 threads (AsyncSliceTransform.kt:56-65 runs transforms like that): the library coalesces closing calls that arrive
    together into one batch.  And the same slices through kmp_zstd_compress_host_batch, the call jni/zstd/BatchWrapper.cpp adds."""
    import threading
    from kompressor_amd import ZstdCompressor
    from kompressor_amd.batch import compress_host_batch
    slices = [host[i * SLICE:(i + 1) * SLICE].tobytes() for i in range(min(n_slices, 1024))]
    c = ZstdCompressor(3)
    c.transform_bytes(slices[0])                                   # (creates the engine)
    one_n = 64
    t0 = time.perf_counter()
    for i in range(one_n):
        c.transform_bytes(slices[i])
    one_dt = time.perf_counter() - t0
    T, per = 64, 16
    ctxs = [ZstdCompressor(3) for _ in range(T)]
    sizes = [0] * T

    def work(t):
        for k in range(per):
            sizes[t] += len(ctxs[t].transform_bytes(slices[(t * per + k) % len(slices)]))

    for warm in (True, False):
        th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        many_dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    frames = compress_host_batch(slices)
    hb_dt = time.perf_counter() - t0
    # the whole batch from (pageable) host memory to frames in host memory: the bulk engines (pieces on worker threads)
    bulk = None
    if n_slices >= 16384 and host.size >= n_slices * SLICE:
        import numpy as np
        from kompressor_amd import _lib
        from kompressor_amd.batch import compress_bound
        lib = _lib.load()
        cap = compress_bound(SLICE)
        lens = np.full(n_slices, SLICE, dtype=np.uint32); offs = np.arange(n_slices, dtype=np.uint64) * SLICE
        caps = np.full(n_slices, cap, dtype=np.uint32); ooff = np.arange(n_slices, dtype=np.uint64) * cap
        out = np.empty(n_slices * cap + 64, dtype=np.uint8); olen = np.zeros(n_slices, dtype=np.uint32)
        vp = lambda a: ctypes.c_void_p(a.ctypes.data)          # noqa: E731
        def passes(k):
            ts = []
            for _ in range(k):
                t0 = time.perf_counter()
                rc = lib.kmp_zstd_compress_host_batch(0, 3, vp(host), vp(offs), vp(lens), n_slices, vp(out), vp(ooff), vp(caps), vp(olen))
                ts.append(time.perf_counter() - t0)
                if rc != 0:
                    return None
            return ts
        times = passes(3)                                       # (the first pass makes the bulk compressor)
        if times:
            pageable = {"GBps": round(n_slices * SLICE / min(times[1:]) / 1e9, 3), "ms": round(min(times[1:]) * 1e3, 1),
                        "first_pass_ms_with_setup": round(times[0] * 1e3, 1)}
            fb = int(olen.astype(np.int64).sum())
            # the same buffers made page-stable (kmp_host_register: what a Kotlin caller does once for a direct ByteBuffer it reuses)
            from kompressor_amd.batch import host_register, host_unregister
            t0 = time.perf_counter()
            host_register(host); host_register(out)
            reg_ms = (time.perf_counter() - t0) * 1e3
            try:
                rt = passes(3)
            finally:
                host_unregister(host); host_unregister(out)
            if rt:
                bulk = {"GBps": round(n_slices * SLICE / min(rt[1:]) / 1e9, 3), "ms": round(min(rt[1:]) * 1e3, 1), "slices": n_slices,
                        "frame_bytes": int(olen.astype(np.int64).sum()), "frame_bytes_equal_pageable_run": bool(int(olen.astype(np.int64).sum()) == fb),
                        "memory": "registered (kmp_host_register: the device reads the slices and writes the frames itself)",
                        "register_ms_both_buffers_once": round(reg_ms, 1), "pageable": pageable}
            else:
                bulk = dict(pageable, slices=n_slices, frame_bytes=fb, memory="pageable (registered run failed)")
        del out
    one_us, many_us = one_dt / one_n * 1e6, many_dt / (T * per) * 1e6
    return {"one_context": {"us_per_slice": round(one_us, 1), "GBps": round(SLICE / one_us / 1e3, 4), "slices": one_n},
            "contexts_64": {"us_per_slice": round(many_us, 1), "GBps": round(SLICE / many_us / 1e3, 4), "slices": T * per, "threads": T,
                            "speedup_over_one_context": round(one_us / many_us, 1)},
            "host_batch_call": {"us_per_slice": round(hb_dt / len(slices) * 1e6, 1), "GBps": round(len(slices) * SLICE / hb_dt / 1e9, 3), "slices": len(slices)},
            "host_batch_bulk": bulk,
            "what": f"{SLICE // 1024} KiB slices in host memory, frames back in host memory, Python threads over ctypes (the C calls run without the GIL); "
                    "one_context / contexts_64: ZstdCompressor(3).transform_bytes per slice = kmp_zstd_compress_stream under the reference's driver loop "
                    "(closing calls of concurrent contexts are coalesced into one device batch); host_batch_call: kmp_zstd_compress_host_batch "
                    "(includes joining the slices into one buffer in Python); host_batch_bulk: the same call over the whole batch in pageable host memory "
                    "(pieces of 8 192 slices on four worker threads, each with its own pinned staging and device batch)"}


def cpu_decode_baseline(frames_host, offs, lens, n_slices, sample=32768):
    """libzstd 1.5.7 ZSTD_decompress over the first `sample` frames the GPU produced, on the host threads."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        from libzstd_ref import find_libzstd_157
        lib = find_libzstd_157()
    except Exception:
        lib = None
    if lib is None:
        return None
    cores = host_cores()["threads_used"]
    sample = min(n_slices, sample)
    lib.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    lib.ZSTD_decompress.restype = ctypes.c_size_t
    base = frames_host.ctypes.data
    per = (sample + cores - 1) // cores
    bad = [0] * cores

    def run(t):
        out = ctypes.create_string_buffer(SLICE)
        for i in range(t * per, min(sample, (t + 1) * per)):
            if lib.ZSTD_decompress(out, SLICE, base + int(offs[i]), int(lens[i])) != SLICE:
                bad[t] += 1

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(run, range(cores)))
    dt = time.perf_counter() - t0
    return {"value": round(sample * SLICE / dt / 1e9, 4), "unit": "GB/s", "cores": cores, "kind": "reference",
            "sample": f"first {sample} frames of the same batch, libzstd 1.5.7 ZSTD_decompress, {cores} threads, {dt:.2f} s wall"
                      + (f" [{sum(bad)} calls failed]" if sum(bad) else "")}


def _pmc_file():
    """profiles/pmc_latest.json (HBM bytes per launch from the round's rocprofv3 --pmc passes) and whether the kernels have
    changed since it was collected: the file records a hash of kompressor_amd/csrc at collection time (tools/summarize_profiles.py)."""
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
    except Exception:
        return None, None
    want = pj.get("csrc_sha256")
    if not want:
        return pj, "unknown (the file records no source hash)"
    return pj, (None if csrc_sha256() == want else "STALE: kompressor_amd/csrc changed since the counters were collected")


def csrc_sha256():
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "kompressor_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def cpu_codec_baseline(kind, base, offs, lens, n, cap, what, out_bytes_per_entry=None, level=6):
    """oracle/cpu_bench.c cpubench_codec on the host threads this job may use: kind 1 = libzstd 1.5.7 ZSTD_decompressStream driven
    like the reference's one-shot driver (+ the one-buffer ZSTD_decompressDCtx figure), 2 = zlib deflate(level, raw), 3 = zlib
    inflate(raw).  1 warm-up + 3 passes over the first n entries, the median; throughput in uncompressed bytes."""
    import statistics
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    hc = host_cores()
    cores = hc["threads_used"]
    lib = ctypes.CDLL(build_cpu_bench())
    lib.cpubench_codec.restype = ctypes.c_int
    lib.cpubench_codec.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32,
                                   ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.c_char_p, ctypes.c_int]
    zpath = None
    if kind == 1:
        try:
            from libzstd_ref import find_libzstd_157
            z = find_libzstd_157()
            zpath = z._path if z is not None else None
        except Exception:
            zpath = None
        if not zpath:
            return None
    offs = np.ascontiguousarray(offs[:n], dtype=np.uint64); lens = np.ascontiguousarray(lens[:n], dtype=np.uint32)
    passes = 4

    def run(stream_api):
        secs = (ctypes.c_double * passes)()
        ob, er = ctypes.c_uint64(0), ctypes.c_uint64(0)
        ver = ctypes.create_string_buffer(32)
        rc = lib.cpubench_codec(kind, zpath.encode() if zpath else None, level, stream_api, base.ctypes.data, offs.ctypes.data, lens.ctypes.data, n, cap,
                                cores, passes, secs, ctypes.byref(ob), ctypes.byref(er), ver, 32)
        if rc != 0:
            raise RuntimeError(f"cpubench_codec({kind}) failed ({rc})")
        return list(secs), ob.value, er.value, ver.value.decode()

    secs, ob, er, ver = run(1)
    med = statistics.median(secs[1:])
    unc = (ob if kind != 2 else int(lens.astype(np.int64).sum()))          # uncompressed bytes of one pass
    res = {"value": round(unc / med / 1e9, 4), "unit": "GB/s", "cores": cores, "kind": "reference", "passes_s": [round(x, 3) for x in secs],
           "sample": f"first {n} {what}, {cores} pthreads (oracle/cpu_bench.c), 1 warm-up + {passes - 1} passes, median {med:.3f} s"
                     + (f", zlib {ver}" if kind != 1 else f", {os.path.basename(zpath)}") + (f" [{er} calls failed]" if er else "")}
    if kind == 1:
        secs0, ob0, er0, _ = run(0)
        res["one_buffer_api"] = {"value": round(ob0 / statistics.median(secs0[1:]) / 1e9, 4), "unit": "GB/s", "what": "ZSTD_decompressDCtx into one 64 KiB buffer (no output slices)"}
        res["sample"] += "; ZSTD_decompressStream with the whole frame as input and output slices of max(8192, frame / 10) bytes, as SliceTransform.kt:33-45 drives Wrapper.cpp:178"
    if kind == 2:
        res["frame_bytes"] = ob
    return res


def extra_legs(b, torch, np, src, in_off, in_len, dst, out_off, out_len, host, n, dev, no_cpu, steps=3):
    """The other single-GPU BASELINE configs on the same batch, after the headline's timed region: configs[2] (ZstdDecompressor
    over the frames just written), configs[4] (raw DEFLATE level 6) and inflate of its streams.  Each: `steps` timed passes after
    one warm-up (inputs resident in HBM), the pipeline's HIP-event time for the roofline figure, a CPU figure from the same
    run (oracle/cpu_bench.c)."""
    S = SLICE
    in_bytes = n * S
    pj, stale = _pmc_file()
    pj = pj if (pj and pj.get("slices") == n and S == 65536) else None
    out = {}
    cap = torch.full((n,), S, dtype=torch.int32, device=dev)
    back = torch.empty(n * S + 64, dtype=torch.uint8, device=dev)

    def timed(fn):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for _ in range(steps):
            r = fn()
        e1.record(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps, e0.elapsed_time(e1) / steps, r

    # ---- configs[2]: decode --------------------------------------------------------------------------------------------
    lens = out_len.cpu().numpy().astype(np.int64)
    frame_bytes = int(lens.sum())
    dt, ms_ev, r = timed(lambda: b.decompress(dst, out_off, out_len, cap, dst=back, out_off=in_off))
    ok = bool(int(r[3].abs().sum().item()) == 0 and torch.equal(back[: n * S], src))
    ms_dec = b.last_kernel_ms(2)
    algo = in_bytes + frame_bytes + 16 * n
    traffic = None
    if pj:
        parts = [pj.get(k + "_hbm_bytes_per_launch") for k in ("k_zstd_decode", "k_zstd_seq_predecode", "k_zstd_lit_predecode")]
        traffic = sum(parts) if all(v is not None for v in parts) else None
    cpu = None
    if not no_cpu:
        ns = min(n, 16384)
        end = int(out_off[ns - 1].item()) + int(out_len[ns - 1].item())
        cpu = cpu_codec_baseline(1, dst[:end].cpu().numpy(), out_off[:ns].cpu().numpy(), out_len[:ns].cpu().numpy(), ns, S, "frames of the same batch")
    out["decode"] = {"metric": "zstd decompression throughput (decoded bytes per second), BASELINE configs[2]: ZstdDecompressor over the level-3 frames of configs[1]",
                     "value": round(in_bytes / dt / 1e9, 3), "unit": "GB/s", "ms_per_step": round(dt * 1e3, 3), "steps": steps, "roundtrip_ok": ok,
                     "roofline": {"bound": "hbm", "kernel": "decode pipeline (k_zstd_lit_predecode + k_zstd_seq_predecode side by side, then k_zstd_decode), one HIP-event bracket",
                                  "achieved": round(algo / (ms_dec * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(algo / (ms_dec * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": traffic, "avg_launch_ms": round(ms_dec, 3)},
                     "cpu_baseline": cpu}
    # ---- configs[4]: raw DEFLATE level 6 ---------------------------------------------------------------------------------
    d_dst = torch.empty(n * b.out_stride + 64, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(n, dtype=torch.int32, device=dev)
    dt, ms_ev, _ = timed(lambda: b.deflate(src, in_off, in_len, d_dst, out_off, d_len, level=6))
    kms = b.deflate_kernel_ms()
    dlens = d_len.cpu().numpy().astype(np.int64)
    import zlib as _z
    ok = True
    for i in (0, 1, 5, 777 % n, n - 1):
        f = d_dst[int(out_off[i]):int(out_off[i]) + int(dlens[i])].cpu().numpy().tobytes()
        ok = ok and _z.decompress(f, -15) == host[i * S:(i + 1) * S].tobytes()
    piece = min(n, 16384)
    algo_piece = (in_bytes + int(dlens.sum()) + 16 * n) * piece // n
    ms_lazy = float(kms.get("parse", 0.0)) or 1.0
    kms = {"k_deflate_sort (+ heaviest-first order)": kms["prepare"], "k_deflate_lazy": kms["parse"], "k_deflate_encode": kms["encode"]}
    cpu = None
    if not no_cpu:
        ns = min(n, 8192)
        cpu = cpu_codec_baseline(2, host, np.arange(ns, dtype=np.uint64) * S, np.full(ns, S, dtype=np.uint32), ns, S + S // 8, "slices of the same batch, deflateInit2(6, -15, 8, 0) + deflate(Z_FINISH)")
        if cpu is not None:
            cpu["stream_bytes_match_gpu"] = bool(cpu.pop("frame_bytes") == int(dlens[:ns].sum()))
    out["deflate6"] = {"metric": "raw DEFLATE level-6 compression throughput (uncompressed input bytes per second), BASELINE configs[4]",
                       "value": round(in_bytes / dt / 1e9, 3), "unit": "GB/s", "ms_per_step": round(dt * 1e3, 3), "steps": steps,
                       "ratio": round(in_bytes / float(dlens.sum()), 4), "zlib_inflate_spot_check_ok": ok,
                       "kernels_ms_first_piece": {k: round(v, 3) for k, v in kms.items()},
                       "roofline": {"bound": "hbm", "kernel": "k_deflate_lazy (a wave per slice walks zlib's lazy parse; bound by scalar issue and LDS round trips, not by HBM: DESIGN.md section 4.4), "
                                                              "one launch per piece of 16 384 slices, timed for the first piece while the next piece's k_deflate_sort runs beside it",
                                    "achieved": round(algo_piece / (ms_lazy * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(algo_piece / (ms_lazy * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                    "traffic": (pj.get("k_deflate_lazy_hbm_bytes_per_launch") if (pj and piece == 16384) else None),
                                    "slices_per_launch": piece, "avg_launch_ms": round(ms_lazy, 3)},
                       "cpu_baseline": cpu}
    # ---- inflate of those streams ------------------------------------------------------------------------------------------
    dt, ms_ev, r = timed(lambda: b.inflate(d_dst, out_off, d_len, cap, dst=back, out_off=in_off))
    ok = bool(int(r[3].abs().sum().item()) == 0 and torch.equal(back[: n * S], src))
    algo = in_bytes + int(dlens.sum()) + 16 * n
    cpu = None
    if not no_cpu:
        ns = min(n, 16384)
        end = int(out_off[ns - 1].item()) + int(d_len[ns - 1].item())
        cpu = cpu_codec_baseline(3, d_dst[:end].cpu().numpy(), out_off[:ns].cpu().numpy(), d_len[:ns].cpu().numpy(), ns, S, "streams of the same batch, inflateInit2(-15) + inflate(Z_FINISH)")
    out["inflate"] = {"metric": "raw DEFLATE decompression throughput (decoded bytes per second): ZlibDecompressor(ZlibFormat.Raw) over configs[4]'s streams",
                      "value": round(in_bytes / dt / 1e9, 3), "unit": "GB/s", "ms_per_step": round(dt * 1e3, 3), "steps": steps, "roundtrip_ok": ok,
                      "roofline": {"bound": "hbm", "kernel": "inflate pipeline (k_inflate_predecode, a lane per stream, then k_inflate_exec), one event bracket on the launch stream",
                                   "achieved": round(algo / (ms_ev * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(algo / (ms_ev * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                   "traffic": (pj.get("inflate_pipeline_hbm_bytes_per_step") if pj else None), "avg_launch_ms": round(ms_ev, 3)},
                      "cpu_baseline": cpu}
    if stale:
        for k in out:
            out[k]["roofline"]["traffic_note"] = stale
    del d_dst, back
    return out


def count_gpus_without_hip():
    """GPUs this process would see, counted without initialising the HIP runtime (the parent of the ranks must not hold one):
    the KFD topology's nodes with SIMDs, cut down to HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when set."""
    have = 0
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for d in os.listdir(base):
            try:
                props = dict(l.split()[:2] for l in open(os.path.join(base, d, "properties")).read().splitlines() if len(l.split()) >= 2)
                if int(props.get("simd_count", "0")) > 0:
                    have += 1
            except Exception:
                continue
    except Exception:
        have = -1                                            # unknown: let the ranks find out
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and have >= 0:
            have = min(have, len([x for x in v.split(",") if x.strip() != ""]))
    return have


def launch_ranks(n):
    """--gpus N > 1 without a launcher: start the N ranks as children of this process -- which never touches a GPU: the
    devices are counted from the KFD topology, not through HIP -- and hand back their exit code."""
    have = count_gpus_without_hip()
    if 0 <= have < n and not os.environ.get("KMP_BENCH_REHEARSAL"):
        print(f"bench.py: --gpus {n} but this machine shows {have} GPU(s)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, choices=[1, 3], default=1,
                    help="BASELINE.json configs index: 1 = 65 536 mixed-class slices per GPU (the headline), "
                         "3 = 131 072 text / binary slices per GPU (1 Mi slices on 8 GPUs)")
    ap.add_argument("--slices", type=int, default=0, help="slices per GPU (default: 65536 for --config 1, 131072 for --config 3)")
    ap.add_argument("--no-exchange", action="store_true", help="N > 1: skip the second figure (root scatter + gather-v inside the timed region)")
    ap.add_argument("--no-pcie", action="store_true", help="N = 1: skip the end-to-end figure that includes the host copies over PCIe")
    ap.add_argument("--team", type=int, default=0, help="lanes per slice in the match kernel (0 = library default)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--slice-class", default="", help="experiments: every slice of one corpus class (T X S B D I Z R) instead of the configuration's mix")
    ap.add_argument("--no-extra", action="store_true", help="N = 1: skip the legs for BASELINE configs[2] / [4] (decode, DEFLATE level 6, inflate) that follow the headline")
    ap.add_argument("--no-stream", action="store_true", help="N = 1: skip the figures of the streaming entry point (what a Kotlin caller binds)")
    ap.add_argument("--level", type=int, default=3, help="zstd level: 3 (BASELINE configs), 1 / 2 (strategy fast), 4 (its double-fast row: slices above 16 KiB up to 128 KiB), or a negative level (slices up to 128 KiB)")
    ap.add_argument("--dict-kib", type=int, default=0,
                    help="compress with a raw-content dictionary of this many KiB shared by all slices (ZstdCompressor(3, dictionary))")
    ap.add_argument("--dict-trained", action="store_true",
                    help="with --dict-kib: a dictionary in zstd's own format, trained by the box's libzstd 1.5.7 (ZDICT_trainFromBuffer) on other slices of the same corpus, instead of raw content")
    ap.add_argument("--slice-kib", type=int, default=64,
                    help="slice size in KiB (64 = BASELINE configs[1]; above 128 the frames have several blocks, up to 2048)")
    ap.add_argument("--deflate-window-bits", type=int, default=15, help="--mode deflate: deflateInit2's windowBits (9 .. 15; BASELINE configs[4] = 15)")
    ap.add_argument("--deflate-mem-level", type=int, default=8, help="--mode deflate: deflateInit2's memLevel (1 .. 9; BASELINE configs[4] = 8)")
    ap.add_argument("--deflate-level", type=int, default=None, help="zlib level of --mode deflate / inflate (1 .. 9, default 6 = BASELINE configs[4])")
    ap.add_argument("--mode", choices=["compress", "decompress", "deflate", "inflate"], default="compress",
                    help="compress = BASELINE configs[1] (the headline); decompress = configs[2] over the same frames; deflate = configs[4] (raw DEFLATE level 6); inflate = ZlibDecompressor over configs[4]'s streams")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    if args.slices <= 0:
        args.slices = 131072 if args.config == 3 else 65536

    # The ROCm runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a
    # queue run in order: the end-to-end leg below uses a stream per piece plus two copy streams, each of which wants a queue of its
    # own (measured: with 4 queues a piece's parse waited behind another piece's whole parse; INTEGRATION.md "Streams").  Read when the
    # runtime initialises, so set before anything touches HIP.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import numpy as np
    import torch
    from kompressor_amd import corpus, sharding
    from kompressor_amd.batch import ZstdBatch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    dist = None
    if world > 1 or os.environ.get("KMP_BENCH_FORCE_DIST"):      # the env switch lets a 1-rank launch exercise the RCCL path
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:          # KMP_BENCH_FORCE_DIST without a launcher: a process group of one
            os.environ.setdefault("MASTER_PORT", "29531"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
        if os.environ.get("KMP_BENCH_REHEARSAL"):
            # rehearsal of the N > 1 control flow on a box with ONE GPU: every rank on cuda:0, gloo instead of RCCL
            # (RCCL refuses two ranks on one device).  Never a measurement.
            local_rank = 0
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    global SLICE
    SLICE = args.slice_kib * 1024
    big = SLICE > 131072
    n = args.slices
    first = rank * n                                   # this rank's block of the global slice index space
    mix = corpus.MIX_TEXT_BINARY if args.config == 3 else corpus.MIX_CONFIG1
    if args.slice_class:
        mix = ord(args.slice_class[0])
    # this rank's HBM plan, checked against what the device has free before anything is allocated (configs[3]'s share is
    # 131 072 slices per GPU: the 8-GPU run must fail with a sentence, not with an allocator error half way through)
    if args.mode == "compress" and not big:
        GiB = float(1 << 30)
        stride = ((SLICE + (SLICE >> 8) + (((128 << 10) - SLICE) >> 11 if SLICE < (128 << 10) else 0)) + 8 + 63) & ~63
        slots = min(n, 65536)
        parts = {"slices": n * SLICE, "frames_strided": n * stride, "frames_dense": n * stride,
                 "context_parts": n * ((SLICE // 4 + 24) * 8 + 2 * (SLICE + 64) + 256 + 32) + slots * 393216,
                 "exchange_on_root": (n * world * SLICE + int(n * world * SLICE / 2.3)) if (world > 1 and not args.no_exchange and rank == 0) else 0,
                 "exchange_per_rank": n * SLICE if (world > 1 and not args.no_exchange) else 0}
        free_b, total_b = torch.cuda.mem_get_info(dev)
        # the arena's span: what is asked for, bounded by half of the free memory (the library's rule); on a rank whose other buffers
        # leave less than that, packed (the root of configs[3]'s exchange: 124 GiB of its own + the whole batch and its frames)
        span = min(int(os.environ.get("KMP_TABLE_SPAN_GIB", "100")) << 30, free_b // 2)
        others = sum(v for k, v in parts.items() if k != "context_parts")
        if others + span + (24 << 30) > free_b:
            span = 0
            os.environ["KMP_TABLE_SPAN_GIB"] = "0"
        parts["context_arena"] = max(parts.pop("context_parts"), span if slots * 393216 >= (4 << 30) else 0)
        need = sum(parts.values())
        plan = {"rank": rank, "need_GiB": round(need / GiB, 1), "free_GiB": round(free_b / GiB, 1), **{k: round(v / GiB, 1) for k, v in parts.items()}}
        print("bench.py HBM plan: " + json.dumps(plan), file=sys.stderr, flush=True)
        if need + (2 << 30) > free_b:
            print(f"bench.py: rank {rank} needs {need / GiB:.1f} GiB of HBM and {free_b / GiB:.1f} GiB are free: "
                  "fewer --slices, --no-exchange, or KMP_TABLE_SPAN_GIB=0 (a packed context arena)", file=sys.stderr)
            sys.exit(3)
    # rank-local generation from seeds, in pieces of 1 GiB; the host keeps the whole block only up to 4 GiB
    # (the CPU baseline and the spot checks read the first slices)
    src = torch.empty(n * SLICE, dtype=torch.uint8, device=dev)
    piece = max(1, (1 << 30) // SLICE)
    keep = n if n * SLICE <= (4 << 30) else piece
    host = np.empty(keep * SLICE, dtype=np.uint8)
    tmp = np.empty(piece * SLICE, dtype=np.uint8) if keep < n else None
    for lo in range(0, n, piece):
        cnt = min(piece, n - lo)
        buf = host[lo * SLICE:(lo + cnt) * SLICE] if lo + cnt <= keep else tmp[: cnt * SLICE]
        corpus.fill(buf, first + lo, cnt, SLICE, mix)
        src[lo * SLICE:(lo + cnt) * SLICE] = torch.from_numpy(buf).to(dev)
    del tmp
    in_off = torch.arange(n, dtype=torch.int64, device=dev) * SLICE
    in_len = torch.full((n,), SLICE, dtype=torch.int32, device=dev)
    # the bench asks for the arena's span and for the second-arena retry explicitly (kmp_batch_options; the library's default is no retry)
    span_gib = int(os.environ.get("KMP_TABLE_SPAN_GIB", "100"))
    retry = int(os.environ.get("KMP_TABLE_RETRY", "1"))
    b = ZstdBatch(max_slices=n, max_slice_bytes=SLICE, device=local_rank, team_lanes=args.team, table_span_gib=span_gib, table_retry=retry)
    print("bench.py context memory: " + json.dumps({k: round(v / 2**30, 2) for k, v in b.memory().items()}) + " GiB", file=sys.stderr, flush=True)
    dst = torch.empty(n * b.out_stride + 64, dtype=torch.uint8, device=dev)
    out_off = torch.arange(n, dtype=torch.int64, device=dev) * b.out_stride
    out_len = torch.zeros(n, dtype=torch.int32, device=dev)
    b.set_profiling(True)

    if args.mode in ("deflate", "inflate"):
        # configs[4]: ZlibCompressor(ZlibFormat.Raw, 6) over the same slices (--deflate-level 1 .. 9: zlib's other levels)
        dlevel = args.deflate_level if args.deflate_level is not None else (args.level if (1 <= args.level <= 9 and args.level != 3) else 6)   # (--level is zstd's, default 3: it only counts here when it is not 3)
        if not 1 <= dlevel <= 9:
            raise SystemExit("--deflate-level 1 .. 9")
        dwb, dml = args.deflate_window_bits, args.deflate_mem_level
        dparams = dict(window_bits=dwb, mem_level=dml)
        if (dwb, dml) != (15, 8):
            # other settings than zlib's defaults may expand a slice by an eighth (kmp_deflate_bound_params): wider strides
            dstride = (int(b.lib.kmp_deflate_bound_params(SLICE, dwb, dml)) + 63) & ~63
            dst = torch.empty(n * dstride + 64, dtype=torch.uint8, device=dev)
            out_off = torch.arange(n, dtype=torch.int64, device=dev) * dstride
        if args.mode == "inflate":
            # ZlibDecompressor(ZlibFormat.Raw) over the streams of configs[4] (made here, once): k_inflate_predecode + k_inflate_exec
            b.deflate(src, in_off, in_len, dst, out_off, out_len, level=dlevel, **dparams)
            torch.cuda.synchronize()
            lens = out_len.cpu().numpy().astype(np.int64)
            cap = torch.full((n,), SLICE, dtype=torch.int32, device=dev)
            back = torch.empty(n * SLICE + 64, dtype=torch.uint8, device=dev)
            for _ in range(max(args.warmup, 1)):
                b.inflate(dst, out_off, out_len, cap, dst=back, out_off=in_off)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter(); e0.record()
            for _ in range(args.steps):
                _, _, l2, st_ = b.inflate(dst, out_off, out_len, cap, dst=back, out_off=in_off)
            e1.record(); torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ok = bool(int(st_.abs().sum().item()) == 0 and torch.equal(back[: n * SLICE], src))
            ms_pipe = e0.elapsed_time(e1) / args.steps
            algo = n * SLICE + int(lens.sum())
            traffic = None
            try:
                traffic = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json"))).get("inflate_pipeline_hbm_bytes_per_step") if (n == 65536 and dlevel == 6) else None
            except Exception:
                traffic = None
            print(json.dumps({
                "metric": f"raw DEFLATE decompression throughput, level-{dlevel} streams of {args.slice_kib} KiB slices (decoded bytes per second)",
                "value": round(n * SLICE / (dt / args.steps) / 1e9, 3), "unit": "GB/s", "n_gpus": 1, "steps": args.steps, "warmup": max(args.warmup, 1),
                "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "u8", "data": "synthetic",
                "config": {"workload": f"ZlibDecompressor(ZlibFormat.Raw) over the {n} raw DEFLATE level-{dlevel} streams of BASELINE configs[4]'s slices", "roundtrip_ok": ok},
                "roofline": {"bound": "hbm", "kernel": "the inflate pipeline: k_inflate_predecode (a lane per stream; bound by instruction issue) then k_inflate_exec; one event bracket per step on the caller's stream",
                             "achieved": round(algo / (ms_pipe * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(algo / (ms_pipe * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": traffic},
                "kernels_ms": {"inflate_pipeline": round(ms_pipe, 3)}, "cpu_baseline": None}), flush=True)
            b.close()
            return
        for _ in range(args.warmup):
            b.deflate(src, in_off, in_len, dst, out_off, out_len, level=dlevel, **dparams)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            b.deflate(src, in_off, in_len, dst, out_off, out_len, level=dlevel, **dparams)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        kms = b.deflate_kernel_ms()
        lens = out_len.cpu().numpy().astype(np.int64)
        # spot check against the host's zlib
        import zlib as _z
        ok = True
        for i in (0, 1, 5, 777 % n, n - 1):
            f = dst[int(out_off[i]):int(out_off[i]) + int(lens[i])].cpu().numpy().tobytes()
            ok = ok and _z.decompress(f, -15) == host[i * SLICE:(i + 1) * SLICE].tobytes()
        # and back: k_inflate over the streams just written (ZlibDecompressor(ZlibFormat.Raw) batch)
        cap = torch.full((n,), SLICE, dtype=torch.int32, device=dev)
        back = torch.empty(n * SLICE + 64, dtype=torch.uint8, device=dev)
        b.inflate(dst, out_off, out_len, cap, dst=back, out_off=in_off)
        torch.cuda.synchronize()
        ti = time.perf_counter()
        _, _, l2, st = b.inflate(dst, out_off, out_len, cap, dst=back, out_off=in_off)
        torch.cuda.synchronize()
        inflate_s = time.perf_counter() - ti
        inflate_ok = bool(int(st.abs().sum().item()) == 0 and torch.equal(back[: n * SLICE], src))
        # CPU baseline: the host zlib through Python (its compress calls release the GIL) on the host threads
        from concurrent.futures import ThreadPoolExecutor
        cpu_cores = host_cores()["threads_used"]
        sample = min(n, 16384)
        per_t = (sample + cpu_cores - 1) // cpu_cores

        def _zrun(t):
            for i in range(t * per_t, min(sample, (t + 1) * per_t)):
                c = _z.compressobj(dlevel, _z.DEFLATED, -dwb, dml, 0)
                c.compress(host[i * SLICE:(i + 1) * SLICE].tobytes()); c.flush()

        t1 = time.perf_counter()
        if not args.no_cpu:
            with ThreadPoolExecutor(cpu_cores) as ex:
                list(ex.map(_zrun, range(cpu_cores)))
        cpu = sample * SLICE / max(time.perf_counter() - t1, 1e-9) / 1e9 if not args.no_cpu else 0.0
        # dominant kernel: k_deflate_lazy, one launch per piece of <= 16384 slices; algorithmic bytes per slice as in SURVEY 8d
        piece = min(n, 16384 if SLICE <= 65536 else max(1, (1 << 29) // (((SLICE + 63) // 64) * 64)))          # (a piece of the workspace: kmp_deflate.hip)
        algo_piece = (n * SLICE + int(lens.sum()) + 16 * n) * piece // n
        traffic_best = None
        try:
            traffic_best = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json"))).get("k_deflate_lazy_hbm_bytes_per_launch") if (piece == 16384 and (dlevel, dwb, dml, SLICE) == (6, 15, 8, 65536)) else None          # (the counters were collected at configs[4]'s settings)
        except Exception:
            traffic_best = None
        ms_best = float(kms.get("parse", 0.0)) or 1.0
        dfl_roofline = {"bound": "hbm", "kernel": "k_deflate_lazy (a wave per slice walks zlib's lazy parse; scalar-issue and LDS bound, HBM is not what limits it), first piece, beside the next piece's k_deflate_sort",
                        "achieved": round(algo_piece / (ms_best * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(algo_piece / (ms_best * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": traffic_best,
                        "slices_per_launch": piece, "avg_launch_ms": round(ms_best, 3)}
        if dlevel > 3 and SLICE > 65536:
            # slices above 64 KiB: the two kernels run once per 64 KiB span (kmp_deflate.hip); the events bracket all of a piece's launches
            segs = (SLICE - 65536 + 32767) // 32768 + 1
            dfl_roofline["kernel"] = f"k_deflate_sort_seg + k_deflate_lazy_seg, the {segs} + {segs} launches of the first piece (64 KiB spans; the other piece's run beside them)"
            kms = {f"k_deflate_sort_seg + k_deflate_lazy_seg x {segs}": kms["parse"], "k_deflate_encode": kms["encode"]}
        elif dlevel > 3:
            kms = {"k_deflate_sort (+ heaviest-first order)": kms["prepare"], "k_deflate_lazy": kms["parse"], "k_deflate_encode": kms["encode"]}
        if dlevel <= 3:
            # levels 1 .. 3: one k_deflate_fast launch over the whole batch (its time is reported in the parse slot of the events)
            piece = n if n <= 65536 else 65536
            algo_piece = (n * SLICE + int(lens.sum()) + 16 * n) * piece // n
            ms_fast = float(kms.get("parse", 0.0)) or 1.0
            kms = {"k_deflate_fast (with the clearing of its head tables)": ms_fast, "k_deflate_encode": float(kms.get("encode", 0.0))}
            dfl_roofline = {"bound": "hbm", "kernel": "k_deflate_fast (a lane per slice; chains of dependent HBM reads, DESIGN.md section 4.4)",
                            "achieved": round(algo_piece / (ms_fast * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(algo_piece / (ms_fast * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                            "traffic": 520_000_000_000 if (dlevel == 1 and n == 65536 and SLICE == 65536) else None,      # tools/pmc_dfl_fast.sh, DESIGN.md
                            "slices_per_launch": piece, "avg_launch_ms": round(ms_fast, 3)}
        print(json.dumps({
            "metric": f"raw DEFLATE level-{dlevel} compression throughput, {args.slice_kib} KiB-slice batch (uncompressed input bytes per second)",
            "value": round(n * SLICE / (dt / args.steps) / 1e9, 3), "unit": "GB/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[4]: {n} x 64 KiB slices, raw DEFLATE level 6 (windowBits 15, memLevel 8)" if (dlevel, dwb, dml, args.slice_kib) == (6, 15, 8, 64) else
                                    f"{n} x {args.slice_kib} KiB slices, raw DEFLATE level {dlevel} (windowBits {dwb}, memLevel {dml})"),
                       "ratio": round(n * SLICE / float(lens.sum()), 4), "inflate_spot_check_ok": ok,
                       "gpu_inflate_GBps": round(n * SLICE / inflate_s / 1e9, 3), "gpu_inflate_roundtrip_ok": inflate_ok},
            ("kernels_ms" if dlevel <= 3 else "kernels_ms_first_workspace_chunk"): {k: round(v, 3) for k, v in kms.items()},
            "roofline": dfl_roofline,
            "cpu_baseline": None if args.no_cpu else {"value": round(cpu, 4), "unit": "GB/s", "cores": cpu_cores, "kind": "reference",
                             "sample": f"first {sample} slices, zlib {_z.ZLIB_RUNTIME_VERSION} via Python, {cpu_cores} threads"}}), flush=True)
        b.close()
        return

    dictionary = None
    if args.dict_kib:
        dictionary = corpus.make(123456789, 1, args.dict_kib * 1024, mix=ord("T")).tobytes()
        if args.dict_trained:
            # what a user of small records does: zstd --train over a sample, ZstdCompressor(level, dictionary = the trained file)
            import ctypes
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            from libzstd_ref import find_libzstd_157
            zl = find_libzstd_157()
            if zl is None:
                print("bench.py --dict-trained: no libzstd 1.5.7 on this machine to train the dictionary with", file=sys.stderr); sys.exit(2)
            ns = 2000
            samples = corpus.make(777000, ns, SLICE if SLICE <= 16384 else 16384).tobytes()
            per = len(samples) // ns
            sizes = (ctypes.c_size_t * ns)(*([per] * ns))
            outb = ctypes.create_string_buffer(args.dict_kib * 1024)
            zl.ZDICT_trainFromBuffer.restype = ctypes.c_size_t
            zl.ZDICT_trainFromBuffer.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
            dn = zl.ZDICT_trainFromBuffer(outb, args.dict_kib * 1024, samples, sizes, ns)
            if zl.ZSTD_isError(dn):
                print("bench.py --dict-trained: ZDICT_trainFromBuffer failed", file=sys.stderr); sys.exit(2)
            dictionary = outb.raw[:dn]
        big = True                      # same reporting as the other one-launch-per-step paths (no per-kernel events)

    if args.level != 3:
        big = True

    # the frames leave a step densely packed ("all kernels incl. compaction")
    dense = torch.empty(n * b.out_stride + 64, dtype=torch.uint8, device=dev)
    dense_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)

    # above 128 KiB: the frames the reference's own one-shot driver gets (its output slices make libzstd stage the input in
    # 128 KiB chunks), not ZSTD_compress2's
    ref_pattern = SLICE > 131072 and dictionary is None

    def step(inp=src):
        b.compress(inp, in_off, in_len, dst, out_off, out_len, dictionary=dictionary, level=args.level, reference=ref_pattern)
        b.compact_into(dst, out_off, out_len, dense, dense_off)
        if dist is not None:
            return sharding.gather_frame_sizes(out_len, n * world)
        return out_len

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    k_match, k_entropy = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        if not big:
            # HIP-event durations of the two kernels of this step (events sit on the launch stream)
            k_match.append(b.last_kernel_ms(0))
            k_entropy.append(b.last_kernel_ms(1))
    fence()
    dt = time.perf_counter() - t0
    launches_timed = max(1, b.last_chunks())           # (of the timed steps: the legs further down run the batch in other shapes)
    if os.environ.get("KMP_BENCH_DEBUG") and k_match:
        print("per-step k_zstd_match ms:", [round(x, 1) for x in k_match], file=sys.stderr)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    lens = out_len.cpu().numpy().astype(np.int64)
    frame_bytes = int(lens.sum())
    in_bytes = n * SLICE

    # ---- second figure, N > 1: the batch starts and ends on rank 0 -------------------------------
    # untimed setup: the blocks every rank generated are collected on the root, as a caller's batch would sit there;
    # timed: root scatter of the slices (point-to-point sends over xGMI), the step above (compress + pack + size
    # all_gather is replaced by gather_frames, which contains it), gather-v of the dense frames to the root.
    exchange = None
    if dist is not None and args.mode == "compress" and not args.no_exchange:
        n_all = n * world
        everything = sharding.gather_slices(src, n_all, SLICE)
        src2 = torch.empty(n * SLICE, dtype=torch.uint8, device=dev)
        stream_buf = torch.empty(int(frame_bytes * world * 1.05) + (1 << 20), dtype=torch.uint8, device=dev) if rank == 0 else None

        def xstep():
            mine = sharding.scatter_slices(everything, src2, n_all, SLICE)
            b.compress(mine, in_off, in_len, dst, out_off, out_len, dictionary=dictionary, level=args.level, reference=ref_pattern)
            b.compact_into(dst, out_off, out_len, dense, dense_off)
            return sharding.gather_frames(dense, out_len, n_all, out=stream_buf)

        xsteps = max(1, min(args.steps, 3))
        xstep()
        fence()
        t0 = time.perf_counter()
        for _ in range(xsteps):
            stream, all_sizes, offs = xstep()
        fence()
        xdt = time.perf_counter() - t0
        t = torch.tensor([xdt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        xdt = float(t.item())
        # what arrived is what was generated locally, and the root's stream holds every rank's frames where the size table says
        ok = torch.tensor([int(torch.equal(src2, src))], dtype=torch.int64, device=dev)
        local_sum = dense[: int(dense_off[n].item())].sum(dtype=torch.int64).reshape(1)
        sums = [torch.zeros_like(local_sum) for _ in range(world)]
        dist.all_gather(sums, local_sum)
        if rank == 0:
            total = int(all_sizes.to(torch.int64).sum().item())
            ok &= int(stream.numel() == total and int(stream.sum(dtype=torch.int64).item()) == sum(int(x.item()) for x in sums))
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        exchange = {"value": round(n_all * SLICE / (xdt / xsteps) / 1e9, 3), "unit": "GB/s", "ms_per_step": round(xdt / xsteps * 1e3, 3),
                    "steps": xsteps, "verified": bool(ok.item()),
                    "what": "root scatter of the slices + compress + dense packing + all_gather of frame sizes + gather-v of the frames "
                            "to the root, all inside the timed region (point-to-point send/recv, " + dist.get_backend() + ")"}
        del everything, src2, stream_buf

    # ---- second line of SURVEY 8d, N = 1: the same step with the host copies over PCIe included (never `value`) --------
    # The batch goes through in PIECES that run side by side (kmp_zstd_compress_batch_pieces): piece p's slices come in on stream p,
    # its kernels follow on that stream while the later pieces are still arriving, its dense frames leave while the others are still
    # being coded -- H2D, kernels and D2H overlap and the device still has every slice in flight once all have arrived.
    pcie = None
    if dist is None and args.mode == "compress" and not args.no_pcie and not big:
        pin_in = torch.from_numpy(host[: n * SLICE]).pin_memory() if host.size >= n * SLICE else None
        if pin_in is not None:
            P = max(1, min(8, int(os.environ.get("KMP_BENCH_PCIE_PIECES", "8"))))
            pin_out = torch.empty(frame_bytes + 64, dtype=torch.uint8).pin_memory()
            pin_tot = torch.zeros(P, dtype=torch.int64).pin_memory()
            src2 = torch.empty(n * SLICE, dtype=torch.uint8, device=dev)
            streams = [torch.cuda.Stream(device=dev) for _ in range(P)]
            s_in, s_out = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)      # the copies: one after the other, in piece order
            ranges = [b.piece_range(n, P, p) for p in range(P)]
            poffs = [torch.zeros(cnt + 1, dtype=torch.int64, device=dev) for _, cnt in ranges]
            arrived = [torch.cuda.Event() for _ in range(P)]
            pdone = [torch.cuda.Event() for _ in range(P)]

            def pstep():
                for p, (first, cnt) in enumerate(ranges):
                    # (copies of different pieces queued on different streams would share the link and all arrive at the end)
                    with torch.cuda.stream(s_in):
                        src2[first * SLICE:(first + cnt) * SLICE].copy_(pin_in[first * SLICE:(first + cnt) * SLICE], non_blocking=True)
                        arrived[p].record(s_in)
                    streams[p].wait_event(arrived[p])
                b.compress_pieces(src2, in_off, in_len, dst, out_off, out_len, streams)
                for p, (first, cnt) in enumerate(ranges):
                    with torch.cuda.stream(streams[p]):
                        # the piece's frames, densely packed, into its own part of `dense` (worst-case placement: first * stride)
                        b.compact_piece(dst, out_off, out_len, first, cnt, dense[first * b.out_stride:], poffs[p], streams[p])
                        pin_tot[p:p + 1].copy_(poffs[p][cnt:cnt + 1], non_blocking=True)
                        pdone[p].record(streams[p])
                base = 0
                for p, (first, cnt) in enumerate(ranges):
                    pdone[p].synchronize()                        # (the host needs the piece's size to place it in the caller's buffer)
                    tot = int(pin_tot[p])
                    with torch.cuda.stream(s_out):
                        pin_out[base:base + tot].copy_(dense[first * b.out_stride:first * b.out_stride + tot], non_blocking=True)
                    base += tot
                s_out.synchronize()
                return base

            torch.cuda.synchronize()
            pstep()
            pt = []
            for _ in range(3):
                t0 = time.perf_counter()
                got = pstep()
                pt.append(time.perf_counter() - t0)
            pdt = min(pt)
            # what arrived is the frames of the timed steps, back to back in slice order
            same = bool(got == frame_bytes)
            if same:
                b.compact_into(dst, out_off, out_len, dense, dense_off)
                torch.cuda.synchronize()
                same = bool(torch.equal(pin_out[:frame_bytes], dense[:frame_bytes].cpu()))
            pcie = {"value": round(in_bytes / pdt / 1e9, 3), "unit": "GB/s", "ms_per_step": round(pdt * 1e3, 3), "pieces": P, "passes_ms": [round(x * 1e3, 1) for x in pt],
                    "frames_equal_the_one_launch_batch": same,
                    "what": f"pinned host slices -> HBM, the level-3 step, dense frames -> pinned host memory, in {P} pieces that run side by side "
                            "(kmp_zstd_compress_batch_pieces: the copies in on one stream in piece order, piece p's kernels and packing on stream p behind its copy, the copies out on one stream); best of 3 passes"}
            del src2, pin_in, pin_out, poffs

    if args.mode == "decompress":
        # configs[2]: ZstdDecompressor over the level-3 frames just produced (strided layout), decoded in place of a fresh buffer
        cap = torch.full((n,), SLICE, dtype=torch.int32, device=dev)
        back = torch.empty(n * SLICE + 64, dtype=torch.uint8, device=dev)
        for _ in range(args.warmup):
            b.decompress(dst, out_off, out_len, cap, dst=back, out_off=in_off)
        fence()
        kd = []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            _, _, l2, st = b.decompress(dst, out_off, out_len, cap, dst=back, out_off=in_off)
            kd.append(b.last_kernel_ms(2))
        fence()
        dt = time.perf_counter() - t0
        ok = bool(int(st.abs().sum().item()) == 0 and torch.equal(back[: n * SLICE], src))
        if rank == 0:
            ms_dec = float(np.mean(kd))
            algo = in_bytes + frame_bytes + 16 * n
            traffic_dec = None
            try:
                pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
                if pj.get("slices") == n:
                    parts = [pj.get(k + "_hbm_bytes_per_launch") for k in ("k_zstd_decode", "k_zstd_seq_predecode", "k_zstd_lit_predecode")]
                    traffic_dec = sum(parts) if all(v is not None for v in parts) else None
            except Exception:
                traffic_dec = None
            cpu_dec = None
            if not args.no_cpu and world == 1:
                ns = min(n, 32768)
                end = int(out_off[ns - 1].item()) + int(out_len[ns - 1].item())
                cpu_dec = cpu_decode_baseline(dst[:end].cpu().numpy(), out_off[:ns].cpu().numpy(), out_len[:ns].cpu().numpy(), n)
            print(json.dumps({
                "metric": f"zstd decompression throughput, level-3 frames of {SLICE // 1024} KiB slices (decoded bytes per second)",
                "value": round(world * in_bytes / (dt / args.steps) / 1e9, 3), "unit": "GB/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                "config": {"workload": (f"BASELINE configs[2]: ZstdDecompressor over the {n} level-3 frames of configs[1]" if (n == 65536 and SLICE == 65536) else
                                        f"ZstdDecompressor over the level-3 frames of {n} x {SLICE // 1024} KiB seeded slices"), "roundtrip_ok": ok},
                "roofline": {"bound": "hbm", "kernel": "the decode pipeline: k_zstd_lit_predecode + k_zstd_seq_predecode (side by side), then k_zstd_decode; one HIP-event bracket per step",
                             "achieved": round(algo / (ms_dec * 1e-3) / 1e9, 2),
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(algo / (ms_dec * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": traffic_dec},
                "kernels_ms": {"decode_pipeline": round(ms_dec, 3)},
                "cpu_baseline": cpu_dec}), flush=True)
        b.close()
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    if rank == 0 and big:
        # frames of several blocks: one k_zstd_big launch per step
        ms_step = dt / args.steps * 1e3
        algo_bytes = in_bytes + frame_bytes + 16 * n
        res = {"metric": "zstd level-3 compression throughput, multi-block frames (uncompressed input bytes per second)",
               "value": round(world * in_bytes / (dt / args.steps) / 1e9, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "u8", "data": "synthetic",
               "config": {"workload": (f"{n} x {args.slice_kib} KiB seeded mixed slices per GPU, ZstdCompressor(level=3, {'trained (zstd-format) ' if args.dict_trained else 'raw-content '}dictionary of {args.dict_kib} KiB), "
                                       "bit-identical to libzstd 1.5.7") if dictionary else
                                      (f"north_star slice-size sweep: {n} x {args.slice_kib} KiB seeded mixed slices per GPU, ZstdCompressor(level=3) "
                                       "one-shot frames of several blocks as the reference's driver gets them (libzstd 1.5.7 with output "
                                       "slices of n / 10 bytes: input staged in 128 KiB chunks), bit-identical"),
                          "slices_per_gpu": n, "slice_bytes": SLICE, "ratio": round(in_bytes / frame_bytes, 4)},
               "roofline": {"bound": "hbm", "kernel": "k_zstd_big (one launch per step: every wave walks the block chains of its slices)", "achieved": round(algo_bytes / (ms_step * 1e-3) / 1e9, 2),
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(algo_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": None}}
        # HBM bytes from the PMC passes of tools/profile_round.sh, when this is the configuration they were collected on
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
            tag, kern = None, None
            if dictionary and args.dict_kib == 16 and n == 65536 and SLICE == 65536: tag, kern = "dict16", "k_zstd_match_dict"
            elif args.level == 1 and n == 65536 and SLICE == 65536: tag, kern = "level1", "k_zstd_match_fast"
            elif args.level == 3 and not dictionary and SLICE == 262144 and n == 32768: tag, kern = "big256k", "k_zstd_big"
            elif args.level == 3 and not dictionary and SLICE == 1048576 and n == 8192: tag, kern = "big1m", "k_zstd_big"
            if tag:
                res["roofline"]["traffic"] = pj.get(f"{tag}:{kern}_hbm_bytes_per_launch")
        except Exception:
            pass
        if dictionary:
            res["roofline"]["kernel"] = "k_zstd_match_dict + k_zstd_entropy (one launch each per step)"
        if args.level != 3:
            res["roofline"]["kernel"] = ("k_zstd_big_fast (one launch per step: every wave walks the block chains of its slices)" if SLICE > 128 * 1024
                                         else "k_zstd_match + k_zstd_entropy with level 4's double-fast row (tables of 1 MiB per team)" if args.level == 4
                                         else "k_zstd_match_fast (a step of 1 - level) + k_zstd_entropy with the literals left raw" if args.level < 0
                                         else "k_zstd_lazy_sort + k_zstd_lazy (a wave per slice walks libzstd's greedy / lazy / lazy2 parse over the sorted positions: zstd_lazy.h) per piece of 16 384 slices, then k_zstd_entropy" if args.level >= 5
                                         else "k_zstd_match_fast + k_zstd_entropy (one launch each per step)")
            res["config"]["workload"] = f"{n} x {args.slice_kib} KiB seeded mixed slices per GPU, ZstdCompressor(level={args.level}), bit-identical to libzstd 1.5.7"
            res["metric"] = f"zstd level {args.level} compression throughput (uncompressed input bytes per second)"
        if not args.no_cpu and not dictionary and (args.level == 3 or SLICE <= 131072):
            # (levels above 5 are slow on the host: a smaller sample keeps the pass to seconds)
            sample = min(n, max(64, ((1 << 29) if args.level <= 4 else (1 << 27)) // SLICE))
            res["cpu_baseline"] = cpu_baseline(host, n, int(lens[:sample].sum()), sample, level=args.level)
        if exchange:
            res["with_scatter_gather"] = exchange
        print(json.dumps(res), flush=True)
    elif rank == 0:
        ms_step = dt / args.steps * 1e3
        value = world * in_bytes / (dt / args.steps) / 1e9
        ms_match = float(np.mean(k_match))
        ms_entropy = float(np.mean(k_entropy))
        # len_in + len_frame + 16 B metadata per slice; the batch goes through `launches` launches of each
        # kernel (chunks), ms_match is the mean launch duration
        launches = launches_timed
        algo_bytes = (in_bytes + frame_bytes + 16 * n) // launches
        achieved = algo_bytes / (ms_match * 1e-3) / 1e9
        traffic = None
        random_access = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                # counters were collected for the default configuration only
                if pj.get("slices") == n and pj.get("launches", 1) == launches and (args.team or 4) == pj.get("team", 4):
                    traffic = pj.get("zstd_match_hbm_bytes_per_launch")
                    rd, wr = pj.get("zstd_match_read_requests_per_launch"), pj.get("zstd_match_write_requests_per_launch")
                    reads_ps, pairs_ps = b.table_rates()
                    if rd and wr and reads_ps > 0 and pairs_ps > 0:
                        # The bound that does apply: every table insert brings a 64-byte line in and sends a
                        # 32-byte sector back -- a read + write PAIR -- and the remaining read requests bring lines in only.
                        # Both rates were measured on this context's own tables when it was created (k_table_probe: random
                        # load + store-into-the-same-word pairs, random loads); the request counts are the PMC pass's.
                        pairs, pure = wr, max(0, rd - wr)
                        model_ms = (pairs / pairs_ps + pure / reads_ps) * 1e3
                        random_access = {"read_requests_per_launch": rd, "write_requests_per_launch": wr,
                                         "pairs_per_s_on_these_tables": round(pairs_ps / 1e9, 2), "reads_per_s_on_these_tables": round(reads_ps / 1e9, 2),
                                         "unit_of_rates": "G/s", "floor_ms": round(model_ms, 1), "frac": round(min(1.0, model_ms / ms_match), 3),
                                         "measured_over_model": round(ms_match / model_ms, 3),
                                         "source": "TCC_EA0_RDREQ / WRREQ per launch from profiles/pmc_latest.json; rates measured at context creation (kmp_batch_table_rates)"}
            except Exception:
                traffic = None
        res = {
            "metric": f"zstd level-3 compression throughput, {args.slice_kib} KiB-slice batch (uncompressed input bytes per second)",
            "value": round(value, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[3], this job's share: {n} x {args.slice_kib} KiB slices per GPU x {world} GPU(s) "
                                    f"(1 048 576 slices on 8), text and binary classes alternating, " if args.config == 3 else
                                    (f"BASELINE configs[1]: {n} x 64 KiB" if args.slice_kib == 64 else f"configs[1]'s mix at another slice size: {n} x {args.slice_kib} KiB")
                                    + " seeded mixed slices per GPU (T/X/S/B/D/I/Z/R classes), ")
                                   + "ZstdCompressor(level=3) one-shot frames, bit-identical to libzstd 1.5.7"
                                   + (f" [EXPERIMENT: every slice of class {args.slice_class[0]}]" if args.slice_class else ""),
                       "slices_per_gpu": n, "slice_bytes": SLICE, "ratio": round(in_bytes / frame_bytes, 4),
                       "team_lanes": b.lib and (args.team or int(os.environ.get("KMP_TEAM_LANES", "8" if n <= 8192 else "4"))), "parallelism": f"slice-sharded x{world}",
                       "parser": "zstd_match.h",
                       "table_span_gib": round(b.memory()["arena"] / 2**30, 1) or "no arena", "table_retry": retry},
            "roofline": {"bound": "hbm", "kernel": "k_zstd_match", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": round(ms_match, 3), "launches_per_step": launches},
            "kernels_ms": {"k_zstd_match": round(ms_match, 3), "k_zstd_entropy": round(ms_entropy, 3)},
        }
        if launches > 1:
            res["kernels_ms_note"] = ("mean duration per launch; with several launches per step an entropy launch runs beside the next piece's parse and is "
                                      "stretched over it (its own cost is the last launch's, about a tenth of a parse launch): the step is the parse launches plus one entropy launch")
        if random_access:
            res["random_access_roofline"] = random_access
        if exchange:
            res["with_scatter_gather"] = exchange
        if pcie:
            res["end_to_end_pcie"] = pcie
        if traffic is not None:
            stale = _pmc_file()[1]
            if stale:
                res["roofline"]["traffic_note"] = stale
        if not args.no_extra and dist is None and args.config == 1 and SLICE == 65536 and not args.slice_class:
            # BASELINE configs[2] and [4] on the same batch, after (and outside) the headline's timed region
            res.update(extra_legs(b, torch, np, src, in_off, in_len, dst, out_off, out_len, host, n, dev, args.no_cpu))
        if not args.no_stream and world == 1 and SLICE <= 131072:
            res["streaming_abi"] = streaming_abi_figure(host, n)
        if not args.no_cpu and world == 1:          # the CPU baseline is a rank-0, N = 1 figure
            sample = min(n, 16384)
            res["cpu_baseline"] = cpu_baseline(host, n, int(lens[:sample].sum()), sample)
        print(json.dumps(res), flush=True)
    b.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
