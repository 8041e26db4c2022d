"""Batched device API over torch CUDA(ROCm) tensors.  torch is plumbing here:
device memory and streams.  All compute happens in libkompressor_hip.so."""
import ctypes

import torch

from . import _lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def compress_bound(n: int) -> int:
    return _lib.load().kmp_zstd_compress_bound(n)


def compress_host_batch(slices, level=3, device=0):
    """kmp_zstd_compress_host_batch: slices held in HOST memory (a list of bytes-like objects of at most 128 KiB) -> their
    frames, through pinned staging and the device batch (the call jni/zstd/BatchWrapper.cpp binds for the JVM)."""
    import numpy as np
    lib = _lib.load()
    n = len(slices)
    lens = np.array([len(s) for s in slices], dtype=np.uint32)
    offs = np.zeros(n, dtype=np.uint64)
    if n > 1:
        offs[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
    src = np.frombuffer(b"".join(bytes(s) for s in slices) + b"\0", dtype=np.uint8)
    caps = np.array([compress_bound(int(l)) for l in lens], dtype=np.uint32)
    ooff = np.zeros(n, dtype=np.uint64)
    if n > 1:
        ooff[1:] = np.cumsum(caps[:-1], dtype=np.uint64)
    dst = np.empty(int(caps.sum()) + 1, dtype=np.uint8)
    olen = np.zeros(n, dtype=np.uint32)
    p = lambda a: ctypes.c_void_p(a.ctypes.data)          # noqa: E731
    rc = lib.kmp_zstd_compress_host_batch(device, level, p(src), p(offs), p(lens), n, p(dst), p(ooff), p(caps), p(olen))
    if rc != 0:
        raise RuntimeError(f"kmp_zstd_compress_host_batch failed ({rc}): {_lib.last_error()}")
    return [dst[int(ooff[i]):int(ooff[i]) + int(olen[i])].tobytes() for i in range(n)]


def decompress_host_batch(frames, caps, device=0):
    """kmp_zstd_decompress_host_batch: frames in host memory -> (contents, statuses); caps[i] = room for entry i (<= 128 KiB)."""
    import numpy as np
    lib = _lib.load()
    n = len(frames)
    lens = np.array([len(f) for f in frames], dtype=np.uint32)
    offs = np.zeros(n, dtype=np.uint64)
    if n > 1:
        offs[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
    src = np.frombuffer(b"".join(bytes(f) for f in frames) + b"\0", dtype=np.uint8)
    caps = np.array(caps, dtype=np.uint32)
    ooff = np.zeros(n, dtype=np.uint64)
    if n > 1:
        ooff[1:] = np.cumsum(caps[:-1], dtype=np.uint64)
    dst = np.empty(int(caps.sum()) + 1, dtype=np.uint8)
    olen = np.zeros(n, dtype=np.uint32)
    st = np.zeros(n, dtype=np.uint32)
    p = lambda a: ctypes.c_void_p(a.ctypes.data)          # noqa: E731
    rc = lib.kmp_zstd_decompress_host_batch(device, p(src), p(offs), p(lens), n, p(dst), p(ooff), p(caps), p(olen), p(st))
    if rc != 0:
        raise RuntimeError(f"kmp_zstd_decompress_host_batch failed ({rc}): {_lib.last_error()}")
    return [dst[int(ooff[i]):int(ooff[i]) + int(olen[i])].tobytes() for i in range(n)], [int(x) for x in st]


def host_register(array):
    """kmp_host_register over a numpy array's memory (pins it and maps it for the device); unregister with host_unregister before
    the array goes away."""
    rc = _lib.load().kmp_host_register(ctypes.c_void_p(array.ctypes.data), array.nbytes)
    if rc != 0:
        raise RuntimeError(f"kmp_host_register failed ({rc}): {_lib.last_error()}")


def host_unregister(array):
    rc = _lib.load().kmp_host_unregister(ctypes.c_void_p(array.ctypes.data))
    if rc != 0:
        raise RuntimeError(f"kmp_host_unregister failed ({rc}): {_lib.last_error()}")


def host_engines_release(device=-1):
    """kmp_host_engines_release: the pinned staging, device buffers and contexts behind the host-batch calls and the coalescer."""
    rc = _lib.load().kmp_host_engines_release(device)
    if rc != 0:
        raise RuntimeError(f"kmp_host_engines_release failed ({rc}): {_lib.last_error()}")


class ZstdBatch:
    """Owns the device workspace for batches of up to `max_slices` slices of up to
    `max_slice_bytes` bytes (<= 128 KiB) on one GPU."""

    def __init__(self, max_slices, max_slice_bytes=65536, device=None, team_lanes=0, ablations=False, table_span_gib=None, table_retry=None):
        """table_span_gib / table_retry: kmp_batch_options (None = the environment's KMP_TABLE_SPAN_GIB / KMP_TABLE_RETRY, else the
        library's defaults: a span of 100 GiB bounded by half of the free memory, no second arena)."""
        # ablations=True: the library's ablation build (tests and A/B tools only: kompressor_amd/_lib.py load_ablations)
        self.lib = _lib.load_ablations() if ablations else _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("kompressor_amd needs a ROCm GPU (no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.max_slices = max_slices
        self.max_slice_bytes = max_slice_bytes
        h = ctypes.c_void_p()
        opts = _lib.BatchOptions(ctypes.sizeof(_lib.BatchOptions), team_lanes, -1 if table_span_gib is None else int(table_span_gib),
                                 -1 if table_retry is None else int(table_retry))
        rc = self.lib.kmp_batch_create_ex(ctypes.byref(h), self.device.index, max_slices, max_slice_bytes, ctypes.byref(opts))
        if rc != 0:
            raise RuntimeError(f"kmp_batch_create failed ({rc}): {self._err()}")
        self._h = h
        self.out_stride = (compress_bound(max_slice_bytes) + 8 + 63) & ~63

    def _err(self):
        return self.lib.kmp_last_error().decode()

    def close(self):
        if getattr(self, "_h", None):
            self.lib.kmp_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def status(self):
        """Waits for the batches queued so far and returns (rc, bits): KMP_STATUS_* bits raised on the device since the
        last call (1 = a slice longer than the context holds, 2 = a parser guard tripped, 4 = a slice of 16 KiB or less in a
        level-4 batch); such slices have out_len 0."""
        bits = ctypes.c_uint32()
        rc = self.lib.kmp_batch_status(self._h, ctypes.byref(bits), self._stream())
        return rc, bits.value

    def memory(self):
        """kmp_batch_memory: device bytes the context holds right now, by part (a dict)."""
        m = _lib.BatchMemoryInfo(); m.struct_bytes = ctypes.sizeof(m)
        if self.lib.kmp_batch_memory(self._h, ctypes.byref(m)) != 0:
            raise RuntimeError(self._err())
        return {k: int(getattr(m, k)) for k, _ in m._fields_[2:]}

    def table_rates(self):
        """(random loads per second, load + store pairs per second) over this context's level-3 team tables, measured at creation."""
        r, q = ctypes.c_float(), ctypes.c_float()
        if self.lib.kmp_batch_table_rates(self._h, ctypes.byref(r), ctypes.byref(q)) != 0:
            raise RuntimeError(self._err())
        return r.value, q.value

    def set_profiling(self, on=True):
        self.lib.kmp_batch_set_profiling(self._h, 1 if on else 0)

    def last_kernel_ms(self, which):
        ms = ctypes.c_float()
        rc = self.lib.kmp_batch_last_kernel_ms(self._h, which, ctypes.byref(ms))
        if rc != 0:
            raise RuntimeError(self._err())
        return ms.value

    def last_chunks(self):
        """Launches of each zstd compress kernel in the last batch."""
        return int(self.lib.kmp_batch_last_chunks(self._h))

    def _check(self, what):
        """check=True of compress / deflate: wait for the batch and raise on a status bit (a slice longer than the context
        holds, a parser guard) instead of handing back frames of length 0."""
        rc, bits = self.status()
        if rc != 0:
            raise RuntimeError(f"{what}: status bits {bits:#x} ({rc}): {self._err()}")

    def compress(self, src, in_off, in_len, dst=None, out_off=None, out_len=None, dictionary=None, level=3, streaming=None, reference=False, check=False):
        """src: uint8 device tensor; in_off int64, in_len int32 device tensors (n each).  dictionary: bytes of a
        dictionary shared by all slices (host memory; raw content, or zstd's own format -- magic EC30A437 -- with its tables,
        repeat offsets and ID; level 3, slices up to 128 KiB; its tables are built once per dictionary).
        level: 3 (default); 1 / 2 / a negative level at any size the context holds (without a dictionary); 4 up to 128 KiB and above
        256 KiB; 5 .. 10 (libzstd's greedy / lazy / lazy2 parsers) for slices up to 128 KiB -- at 9 and 10 a slice of 8 bytes .. 16 KiB is
        another strategy and comes back refused (out_len 0, status bit 4), as does a level-4 slice between 128 and 256 KiB.
        streaming: None = one-shot frames; "data" / "empty" = the frames of slices that arrived through finish = false
        calls, closed by a call with / without data (context created for slices above 128 KiB; levels 1 to 3).
        reference: the frames ZstdCompressor(level).transform(bytes) returns -- above 128 KiB the reference's output slices
        make libzstd stage the input in 128 KiB chunks, so they differ from ZSTD_compress2's (the default here) wherever
        the block pre-splitter cuts (no dictionary); up to 128 KiB both are the same.
        Returns (dst, out_off, out_len): frame i = dst[out_off[i] : out_off[i] + out_len[i]]."""
        n = in_len.numel()
        if dst is None:
            dst = torch.empty(n * self.out_stride + 64, dtype=torch.uint8, device=self.device)
        if out_off is None:
            out_off = torch.arange(n, dtype=torch.int64, device=self.device) * self.out_stride
        if out_len is None:
            out_len = torch.zeros(n, dtype=torch.int32, device=self.device)
        if reference and streaming is None:
            if dictionary is not None:
                raise ValueError("reference=True is served without a dictionary")
            rc = self.lib.kmp_zstd_compress_batch_reference(self._h, _ptr(src), _ptr(in_off), _ptr(in_len), n,
                                                            _ptr(dst), _ptr(out_off), _ptr(out_len), level, 0, self._stream())
        elif streaming is not None:
            rc = self.lib.kmp_zstd_compress_batch_stream_level(self._h, _ptr(src), _ptr(in_off), _ptr(in_len), n,
                                                               _ptr(dst), _ptr(out_off), _ptr(out_len), 1 if streaming == "empty" else 0, level, self._stream())
        elif level not in (0, 3):
            if dictionary is not None:
                raise ValueError("levels 1 and 2 are served without a dictionary")
            rc = self.lib.kmp_zstd_compress_batch_level(self._h, _ptr(src), _ptr(in_off), _ptr(in_len), n,
                                                        _ptr(dst), _ptr(out_off), _ptr(out_len), level, self._stream())
        elif dictionary is not None:
            rc = self.lib.kmp_zstd_compress_batch_dict(self._h, _ptr(src), _ptr(in_off), _ptr(in_len), n,
                                                       _ptr(dst), _ptr(out_off), _ptr(out_len), bytes(dictionary), len(dictionary), self._stream())
        else:
            rc = self.lib.kmp_zstd_compress_batch(self._h, _ptr(src), _ptr(in_off), _ptr(in_len), n,
                                                  _ptr(dst), _ptr(out_off), _ptr(out_len), self._stream())
        if rc != 0:
            raise RuntimeError(f"kmp_zstd_compress_batch failed ({rc}): {self._err()}")
        if check:
            self._check("kmp_zstd_compress_batch")
        return dst, out_off, out_len

    def piece_range(self, n, pieces, piece):
        """(first, count) of part `piece` of an n-slice batch cut into `pieces` (kmp_batch_piece_range)."""
        a, b = ctypes.c_uint32(), ctypes.c_uint32()
        self.lib.kmp_batch_piece_range(n, pieces, piece, ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    def compress_pieces(self, src, in_off, in_len, dst, out_off, out_len, streams):
        """kmp_zstd_compress_batch_pieces: the level-3 batch in len(streams) parts that run side by side, part p queued on the torch
        stream streams[p] behind whatever the caller queued there (the copy that brings the part's slices in)."""
        n = in_len.numel()
        arr = (ctypes.c_void_p * len(streams))(*[s.cuda_stream for s in streams])
        rc = self.lib.kmp_zstd_compress_batch_pieces(self._h, _ptr(src), _ptr(in_off), _ptr(in_len), n, _ptr(dst), _ptr(out_off), _ptr(out_len), len(streams), arr)
        if rc != 0:
            raise RuntimeError(f"kmp_zstd_compress_batch_pieces failed ({rc}): {self._err()}")
        return dst, out_off, out_len

    def compact_piece(self, src, in_off, lens, first, count, dst, offs, stream):
        """Dense packing of one part on `stream`: frames first .. first + count of the strided layout go to dst (a buffer of the part's
        own) back to back; offs (int64, count + 1 entries) receives the exclusive scan, offs[count] the part's bytes."""
        rc = self.lib.kmp_compact_batch(self._h, _ptr(src), ctypes.c_void_p(in_off.data_ptr() + 8 * first), ctypes.c_void_p(lens.data_ptr() + 4 * first), count,
                                        _ptr(dst), _ptr(offs), ctypes.c_void_p(stream.cuda_stream))
        if rc != 0:
            raise RuntimeError(f"kmp_compact_batch failed ({rc}): {self._err()}")

    def decompress(self, src, in_off, in_len, out_cap, dst=None, out_off=None, dictionary=None):
        """Frames -> slices.  out_cap: int32 device tensor of per-frame capacities.  dictionary: uint8 device tensor
        with a raw-content dictionary shared by all frames.  Returns (dst, out_off, out_len, status)."""
        n = in_len.numel()
        if out_off is None:
            out_off = torch.cumsum(out_cap.to(torch.int64), 0) - out_cap.to(torch.int64)
        if dst is None:
            total = int(out_cap.to(torch.int64).sum().item())
            dst = torch.empty(total + 64, dtype=torch.uint8, device=self.device)
        out_len = torch.zeros(n, dtype=torch.int32, device=self.device)
        status = torch.zeros(n, dtype=torch.int32, device=self.device)
        if dictionary is not None:
            rc = self.lib.kmp_zstd_decompress_batch_dict(self._h, _ptr(src), _ptr(in_off), _ptr(in_len), n,
                                                         _ptr(dst), _ptr(out_off), _ptr(out_cap), _ptr(out_len), _ptr(status),
                                                         _ptr(dictionary), dictionary.numel(), self._stream())
        else:
            rc = self.lib.kmp_zstd_decompress_batch(self._h, _ptr(src), _ptr(in_off), _ptr(in_len), n,
                                                    _ptr(dst), _ptr(out_off), _ptr(out_cap), _ptr(out_len), _ptr(status),
                                                    self._stream())
        if rc != 0:
            raise RuntimeError(f"kmp_zstd_decompress_batch failed ({rc}): {self._err()}")
        return dst, out_off, out_len, status

    _FORMATS = {"raw": 0, "zlib": 1, "gzip": 2, "auto": 3}

    def inflate(self, src, in_off, in_len, out_cap, zlib_wrapper=False, dst=None, out_off=None, format=None):   # noqa: A002
        """n DEFLATE streams -> slices; format "raw" / "zlib" / "gzip" / "auto" (zlib or gzip per stream).
        Returns (dst, out_off, out_len, status)."""
        fmt = self._FORMATS[format] if format is not None else (1 if zlib_wrapper else 0)
        n = in_len.numel()
        if out_off is None:
            out_off = torch.cumsum(out_cap.to(torch.int64), 0) - out_cap.to(torch.int64)
        if dst is None:
            dst = torch.empty(int(out_cap.to(torch.int64).sum().item()) + 64, dtype=torch.uint8, device=self.device)
        out_len = torch.zeros(n, dtype=torch.int32, device=self.device)
        status = torch.zeros(n, dtype=torch.int32, device=self.device)
        rc = self.lib.kmp_inflate_batch(self._h, _ptr(src), _ptr(in_off), _ptr(in_len), n, _ptr(dst), _ptr(out_off), _ptr(out_cap),
                                        _ptr(out_len), _ptr(status), fmt, self._stream())
        if rc != 0:
            raise RuntimeError(f"kmp_inflate_batch failed ({rc}): {self._err()}")
        return dst, out_off, out_len, status

    def deflate(self, src, in_off, in_len, dst=None, out_off=None, out_len=None, zlib_wrapper=False, format=None, level=6, check=False,   # noqa: A002
                window_bits=15, mem_level=8):
        """DEFLATE streams (zlib level 6 -- or any other level 1 .. 9: deflate_fast 1 .. 3, deflate_slow 4 .. 9 --; windowBits 9 .. 15 and
        memLevel 1 .. 9 as deflateInit2 takes them, 15 and 8 by default), format "raw" / "zlib" / "gzip", for slices up to the context's
        max_slice_bytes (64 KiB at least)."""
        fmt = self._FORMATS[format] if format is not None else (1 if zlib_wrapper else 0)
        if fmt == 3:
            raise ValueError("Compression can't be used with auto-detection")       # ZlibFormat.kt:28
        n = in_len.numel()
        params = (window_bits, mem_level) != (15, 8)
        stride = (self.lib.kmp_deflate_bound_params(max(self.max_slice_bytes, 65536), window_bits, mem_level) + 63) & ~63
        if dst is None:
            dst = torch.empty(n * stride + 64, dtype=torch.uint8, device=self.device)
        if out_off is None:
            out_off = torch.arange(n, dtype=torch.int64, device=self.device) * stride
        if out_len is None:
            out_len = torch.zeros(n, dtype=torch.int32, device=self.device)
        if params:
            rc = self.lib.kmp_deflate_compress_batch_params(self._h, _ptr(src), _ptr(in_off), _ptr(in_len), n, _ptr(dst), _ptr(out_off), _ptr(out_len),
                                                            fmt, level, window_bits, mem_level, self._stream())
        elif level in (-1, 6):
            fn = (self.lib.kmp_deflate_compress_batch, self.lib.kmp_zlib_compress_batch, self.lib.kmp_gzip_compress_batch)[fmt]
            rc = fn(self._h, _ptr(src), _ptr(in_off), _ptr(in_len), n, _ptr(dst), _ptr(out_off), _ptr(out_len), self._stream())
        else:
            rc = self.lib.kmp_deflate_compress_batch_level(self._h, _ptr(src), _ptr(in_off), _ptr(in_len), n, _ptr(dst), _ptr(out_off), _ptr(out_len),
                                                           fmt, level, self._stream())
        if rc != 0:
            raise RuntimeError(f"kmp_deflate_compress_batch failed ({rc}): {self._err()}")
        if check:
            self._check("kmp_deflate_compress_batch")
        return dst, out_off, out_len

    def deflate_kernel_ms(self):
        ms = (ctypes.c_float * 4)()
        if self.lib.kmp_deflate_last_kernel_ms(self._h, ms) != 0:
            raise RuntimeError(self._err())
        # the four stages of the first workspace piece; which kernels they are depends on the level and the slice size:
        #   prepare: k_deflate_sort + the heaviest-first order (slices <= 64 KiB, levels >= 4) | k_deflate_chains (longer slices)
        #   search : nothing                                                                  | k_deflate_best
        #   parse  : k_deflate_lazy                                                           | k_deflate_parse;  levels 1 .. 3: k_deflate_fast
        #   encode : k_deflate_encode
        return dict(zip(("prepare", "search", "parse", "encode"), (float(x) for x in ms)))

    def compact_into(self, src, in_off, lens, dst, offs):
        """Dense packing into caller-owned buffers, no host round trip: frame i goes to dst[offs[i] : offs[i] + lens[i]],
        offs (int64, n + 1 entries) = exclusive scan of lens, offs[n] = total bytes.  dst must hold sum(lens) bytes."""
        n = lens.numel()
        if offs.numel() < n + 1:
            raise ValueError("offs needs n + 1 entries")
        rc = self.lib.kmp_compact_batch(self._h, _ptr(src), _ptr(in_off), _ptr(lens), n, _ptr(dst), _ptr(offs), self._stream())
        if rc != 0:
            raise RuntimeError(f"kmp_compact_batch failed ({rc}): {self._err()}")
        return dst, offs

    def compact(self, src, in_off, lens):
        """Dense packing of n frames; returns (dst, offsets[n+1])."""
        n = lens.numel()
        offs = torch.empty(n + 1, dtype=torch.int64, device=self.device)
        total_cap = int(lens.to(torch.int64).sum().item())
        dst = torch.empty(total_cap + 64, dtype=torch.uint8, device=self.device)
        rc = self.lib.kmp_compact_batch(self._h, _ptr(src), _ptr(in_off), _ptr(lens), n, _ptr(dst), _ptr(offs), self._stream())
        if rc != 0:
            raise RuntimeError(f"kmp_compact_batch failed ({rc}): {self._err()}")
        return dst, offs
