#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its configs[1] workload:
zstd level-3 compression of a batch of 65 536 x 64 KiB slices (seeded synthetic
"Silesia-like" mix, kompressor_amd/csrc/corpus.c) per GPU, inputs resident in
HBM when the timed region starts.  One step = one pass of the whole hot path
(match kernel + entropy/frame kernel) over the batch.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--slices M]

N > 1: launched by torch.distributed.run, one rank per GPU; every rank owns its
own block of slices (weak scaling, no collective on the data path; the frame
size table is all-gathered over RCCL inside the timed region).
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SLICE = 65536
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(host, n_slices, frames_expected, sample=None):
    """Times the reference's arithmetic on the host cores: a binary libzstd 1.5.7
    if this machine has one (kind "reference"), else the oracle's C restatement
    (kind "port").  Bounded sample of the same workload."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    cores = min(os.cpu_count() or 1, 64)
    sample = min(n_slices, 8192) if sample is None else sample
    kind, label, workers = None, None, []
    try:
        from libzstd_ref import find_libzstd_157
        lib = find_libzstd_157()
    except Exception:
        lib = None
    if lib is not None:
        kind = "reference"
        label = f"libzstd 1.5.7 ({os.path.basename(lib._path)}) ZSTD_compress2 level 3"
        lib.ZSTD_createCCtx.restype = ctypes.c_void_p
        lib.ZSTD_CCtx_setParameter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        lib.ZSTD_compress2.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
        lib.ZSTD_compress2.restype = ctypes.c_size_t

        def make_worker():
            cctx = lib.ZSTD_createCCtx()
            lib.ZSTD_CCtx_setParameter(cctx, 100, 3)
            cap = SLICE + SLICE // 128 + 1024
            out = ctypes.create_string_buffer(cap)
            return lambda ptr: lib.ZSTD_compress2(cctx, out, cap, ptr, SLICE)
    else:
        import helpers
        k = helpers.oracle().lib
        kind = "port"
        label = "oracle/zstd_l3_ref.c (C restatement)"

        def make_worker():
            cap = SLICE + SLICE // 128 + 1024
            out = ctypes.create_string_buffer(cap)
            return lambda ptr: k.kref_zstd_l3_compress(out, cap, ctypes.c_char_p(ptr), SLICE)
    base = host.ctypes.data
    per = (sample + cores - 1) // cores
    totals = [0] * cores

    errors = [0] * cores

    def run(t):
        w = make_worker()
        s = 0
        for i in range(t * per, min(sample, (t + 1) * per)):
            r = w(base + i * SLICE)
            if r == 0 or r > (1 << 40):          # libzstd error codes are (size_t)-code
                errors[t] += 1
            else:
                s += r
        totals[t] = s

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(run, range(cores)))
    dt = time.perf_counter() - t0
    ok = (frames_expected is None) or (sum(totals) == frames_expected)
    if sum(errors):
        label += f" [{sum(errors)} of {sample} CPU calls returned an error]"
    return {"value": round(sample * SLICE / dt / 1e9, 4), "unit": "GB/s", "cores": cores, "kind": kind,
            "sample": f"first {sample} slices of the same batch, {label}, {cores} threads, one context per thread, "
                      f"{dt:.2f} s wall; total frame bytes {'match' if ok else 'DIFFER from'} the GPU's"}


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
    cores = min(os.cpu_count() or 1, 64)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--slices", type=int, default=65536, help="slices per GPU (BASELINE configs[1]: 65536)")
    ap.add_argument("--team", type=int, default=0, help="lanes per slice in the match kernel (0 = library default)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--level", type=int, default=3, help="zstd level: 3 (BASELINE configs), or 1 / 2 (strategy fast)")
    ap.add_argument("--dict-kib", type=int, default=0,
                    help="compress with a raw-content dictionary of this many KiB shared by all slices (ZstdCompressor(3, dictionary))")
    ap.add_argument("--slice-kib", type=int, default=64,
                    help="slice size in KiB (64 = BASELINE configs[1]; above 128 the frames have several blocks, up to 2048)")
    ap.add_argument("--mode", choices=["compress", "decompress", "deflate"], default="compress",
                    help="compress = BASELINE configs[1] (the headline); decompress = configs[2] over the same frames; deflate = configs[4] (raw DEFLATE level 6)")
    args = ap.parse_args()

    import numpy as np
    import torch
    from kompressor_amd import corpus, sharding
    from kompressor_amd.batch import ZstdBatch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    if world > 1 or os.environ.get("KMP_BENCH_FORCE_DIST"):      # the env switch lets a 1-rank launch exercise the RCCL path
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
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
    host = np.empty(n * SLICE, dtype=np.uint8)
    corpus.fill(host, first, n, SLICE, corpus.MIX_CONFIG1)
    src = torch.empty(n * SLICE, dtype=torch.uint8, device=dev)
    step_copy = 1 << 30
    for o in range(0, n * SLICE, step_copy):
        src[o:o + step_copy] = torch.from_numpy(host[o:o + step_copy]).to(dev)
    in_off = torch.arange(n, dtype=torch.int64, device=dev) * SLICE
    in_len = torch.full((n,), SLICE, dtype=torch.int32, device=dev)
    b = ZstdBatch(max_slices=n, max_slice_bytes=SLICE, device=local_rank, team_lanes=args.team)
    dst = torch.empty(n * b.out_stride + 64, dtype=torch.uint8, device=dev)
    out_off = torch.arange(n, dtype=torch.int64, device=dev) * b.out_stride
    out_len = torch.zeros(n, dtype=torch.int32, device=dev)
    b.set_profiling(True)

    if args.mode == "deflate":
        # configs[4]: ZlibCompressor(ZlibFormat.Raw, 6) over the same slices
        for _ in range(args.warmup):
            b.deflate(src, in_off, in_len, dst, out_off, out_len)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            b.deflate(src, in_off, in_len, dst, out_off, out_len)
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
        cpu_cores = min(os.cpu_count() or 1, 64)
        sample = min(n, 16384)
        per_t = (sample + cpu_cores - 1) // cpu_cores

        def _zrun(t):
            for i in range(t * per_t, min(sample, (t + 1) * per_t)):
                c = _z.compressobj(6, _z.DEFLATED, -15, 8, 0)
                c.compress(host[i * SLICE:(i + 1) * SLICE].tobytes()); c.flush()

        t1 = time.perf_counter()
        if not args.no_cpu:
            with ThreadPoolExecutor(cpu_cores) as ex:
                list(ex.map(_zrun, range(cpu_cores)))
        cpu = sample * SLICE / max(time.perf_counter() - t1, 1e-9) / 1e9 if not args.no_cpu else 0.0
        # dominant kernel: k_deflate_best, one launch per piece of <= 16384 slices; algorithmic bytes per slice as in SURVEY 8d
        piece = min(n, 16384)
        algo_piece = (n * SLICE + int(lens.sum()) + 16 * n) * piece // n
        traffic_best = None
        try:
            traffic_best = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json"))).get("k_deflate_best_hbm_bytes_per_launch") if piece == 16384 else None
        except Exception:
            traffic_best = None
        ms_best = float(kms.get("k_deflate_best", 0.0)) or 1.0
        dfl_roofline = {"bound": "hbm", "kernel": "k_deflate_best (LDS- and issue-bound chain walk; HBM is not what limits it)",
                        "achieved": round(algo_piece / (ms_best * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(algo_piece / (ms_best * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": traffic_best,
                        "slices_per_launch": piece, "avg_launch_ms": round(ms_best, 3)}
        print(json.dumps({
            "metric": "raw DEFLATE level-6 compression throughput, 64 KiB-slice batch (uncompressed input bytes per second)",
            "value": round(n * SLICE / (dt / args.steps) / 1e9, 3), "unit": "GB/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[4]: {n} x 64 KiB slices, raw DEFLATE level 6 (windowBits 15, memLevel 8)",
                       "ratio": round(n * SLICE / float(lens.sum()), 4), "inflate_spot_check_ok": ok,
                       "gpu_inflate_GBps": round(n * SLICE / inflate_s / 1e9, 3), "gpu_inflate_roundtrip_ok": inflate_ok},
            "kernels_ms_first_workspace_chunk": {k: round(v, 3) for k, v in kms.items()},
            "roofline": dfl_roofline,
            "cpu_baseline": None if args.no_cpu else {"value": round(cpu, 4), "unit": "GB/s", "cores": cpu_cores, "kind": "reference",
                             "sample": f"first {sample} slices, zlib {_z.ZLIB_RUNTIME_VERSION} via Python, {cpu_cores} threads"}}), flush=True)
        b.close()
        return

    dictionary = None
    if args.dict_kib:
        dictionary = corpus.make(123456789, 1, args.dict_kib * 1024, mix=ord("T")).tobytes()
        big = True                      # same reporting as the other one-launch-per-step paths (no per-kernel events)

    if args.level != 3:
        big = True

    def step():
        b.compress(src, in_off, in_len, dst, out_off, out_len, dictionary=dictionary, level=args.level)
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
    if os.environ.get("KMP_BENCH_DEBUG") and k_match:
        print("per-step k_zstd_match ms:", [round(x, 1) for x in k_match], file=sys.stderr)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    lens = out_len.cpu().numpy().astype(np.int64)
    frame_bytes = int(lens.sum())
    in_bytes = n * SLICE
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
                    traffic_dec = pj.get("k_zstd_decode_hbm_bytes_per_launch")
            except Exception:
                traffic_dec = None
            cpu_dec = None
            if not args.no_cpu and world == 1:
                ns = min(n, 32768)
                end = int(out_off[ns - 1].item()) + int(out_len[ns - 1].item())
                cpu_dec = cpu_decode_baseline(dst[:end].cpu().numpy(), out_off[:ns].cpu().numpy(), out_len[:ns].cpu().numpy(), n)
            print(json.dumps({
                "metric": "zstd decompression throughput, level-3 frames of 64 KiB slices (decoded bytes per second)",
                "value": round(world * in_bytes / (dt / args.steps) / 1e9, 3), "unit": "GB/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                "config": {"workload": f"BASELINE configs[2]: ZstdDecompressor over the {n} level-3 frames of configs[1]", "roundtrip_ok": ok},
                "roofline": {"bound": "hbm", "kernel": "k_zstd_decode", "achieved": round(algo / (ms_dec * 1e-3) / 1e9, 2),
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(algo / (ms_dec * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": traffic_dec},
                "kernels_ms": {"k_zstd_decode": round(ms_dec, 3)},
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
               "config": {"workload": (f"{n} x {args.slice_kib} KiB seeded mixed slices per GPU, ZstdCompressor(level=3, dictionary of {args.dict_kib} KiB), "
                                       "bit-identical to libzstd 1.5.7") if dictionary else
                                      (f"north_star slice-size sweep: {n} x {args.slice_kib} KiB seeded mixed slices per GPU, ZstdCompressor(level=3) "
                                       "one-shot frames of several blocks, bit-identical to libzstd 1.5.7"),
                          "slices_per_gpu": n, "slice_bytes": SLICE, "ratio": round(in_bytes / frame_bytes, 4)},
               "roofline": {"bound": "hbm", "kernel": "k_zstd_big (one launch per step: every wave walks the block chains of its slices)", "achieved": round(algo_bytes / (ms_step * 1e-3) / 1e9, 2),
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(algo_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": None}}
        if dictionary:
            res["roofline"]["kernel"] = "k_zstd_match_dict + k_zstd_entropy (one launch each per step)"
        if args.level != 3:
            res["roofline"]["kernel"] = ("k_zstd_big_fast (one launch per step: every wave walks the block chains of its slices)" if SLICE > 128 * 1024
                                         else "k_zstd_match_fast + k_zstd_entropy (one launch each per step)")
            res["config"]["workload"] = f"{n} x {args.slice_kib} KiB seeded mixed slices per GPU, ZstdCompressor(level={args.level}), bit-identical to libzstd 1.5.7"
            res["metric"] = f"zstd level-{args.level} compression throughput (uncompressed input bytes per second)"
        if not args.no_cpu and not dictionary and args.level == 3:
            sample = min(n, max(64, (1 << 29) // SLICE))
            res["cpu_baseline"] = cpu_baseline(host, n, int(lens[:sample].sum()), sample)
        print(json.dumps(res), flush=True)
    elif rank == 0:
        ms_step = dt / args.steps * 1e3
        value = world * in_bytes / (dt / args.steps) / 1e9
        ms_match = float(np.mean(k_match))
        ms_entropy = float(np.mean(k_entropy))
        # len_in + len_frame + 16 B metadata per slice; the batch goes through `launches` launches of each
        # kernel (chunks), ms_match is the mean launch duration
        launches = max(1, b.last_chunks())
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
                    if rd and wr:
                        # the bound that does apply: random 64-byte transactions, priced with tools/randgather.hip on this
                        # GPU (profiles/r01_random_access.txt): a probe + insert into one line 20 G/s, further reads 54 G/s
                        floor_ms = (wr / 20e9 + max(0, rd - wr) / 54e9) * 1e3
                        random_access = {"read_requests_per_launch": rd, "write_requests_per_launch": wr,
                                         "floor_ms": round(floor_ms, 1), "frac": round(floor_ms / ms_match, 3),
                                         "source": "TCC_EA0_RDREQ / WRREQ from profiles/pmc_latest.json; rates from profiles/r01_random_access.txt"}
            except Exception:
                traffic = None
        res = {
            "metric": "zstd level-3 compression throughput, 64 KiB-slice batch (uncompressed input bytes per second)",
            "value": round(value, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: {n} x 64 KiB seeded mixed slices per GPU (T/X/S/B/D/I/Z/R classes), "
                                   "ZstdCompressor(level=3) one-shot frames, bit-identical to libzstd 1.5.7",
                       "slices_per_gpu": n, "slice_bytes": SLICE, "ratio": round(in_bytes / frame_bytes, 4),
                       "team_lanes": b.lib and (args.team or int(os.environ.get("KMP_TEAM_LANES", "4"))), "parallelism": f"slice-sharded x{world}"},
            "roofline": {"bound": "hbm", "kernel": "k_zstd_match", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": round(ms_match, 3), "launches_per_step": launches},
            "kernels_ms": {"k_zstd_match": round(ms_match, 3), "k_zstd_entropy": round(ms_entropy, 3)},
        }
        if random_access:
            res["random_access_roofline"] = random_access
        if not args.no_cpu and world == 1:          # the CPU baseline is a rank-0, N = 1 figure
            sample = min(n, 8192)
            res["cpu_baseline"] = cpu_baseline(host, n, int(lens[:sample].sum()))
        print(json.dumps(res), flush=True)
    b.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
