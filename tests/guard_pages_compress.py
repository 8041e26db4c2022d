#!/usr/bin/env python3
"""Compressor kernels against guard pages (TEST INFRASTRUCTURE): every slice is placed so that it ends exactly at a PROT_NONE
page (and starts right after one), the output buffer likewise with kmp_zstd_compress_bound(len) + 1024 bytes, and the
match / entropy (zstd levels 3, 1, 2) and DEFLATE kernel bodies run on the CPU wave emulator: a read past the end of a slice
or a write past the output bound kills the process.  The frames are compared with the oracle / zlib on the way.

    python tests/guard_pages_compress.py zstd | l1 | l2 | neg | l4 | lazy5 .. lazy10 | deflate | deflatep [--quick]      (neg: level -3; l4: level 4, slices above 16 KiB;
    KXEMU_FUSE=1 / KXEMU_MATCH_V2=1 in the environment: the fused kernel / the split-phase parser for `zstd`)
"""
import ctypes
import os
import random
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np
import helpers, fuzz_decoders as F
from kompressor_amd import corpus
emu = helpers.emu()
o = helpers.oracle()
rng = random.Random(5)
which = sys.argv[1]
sizes = [1,2,3,5,7,8,9,15,16,17,31,33,63,64,65,100,255,256,1000,4095,4096,4097,9000,16384,16385,20000,65535,65536] + ([131071,131072] if not which.startswith('deflate') else [])
if which == 'deflatep':
    sizes += [65537, 98303, 98304, 98305, 140000]          # above 64 KiB: the kernels take the slice in 64 KiB spans
if '--quick' in sys.argv:
    sizes = [1, 7, 8, 9, 17, 64, 255, 4097, 20000] + ([65536] if which.startswith('deflate') else [131072]) + ([98305] if which == 'deflatep' else [])
if which == 'l4':
    os.environ['KXEMU_LEVEL'] = '4'; sizes = [S for S in sizes if S > 16384] + [16385, 40000]
for S in sizes:
    for mix in "TZRB":
        d = corpus.make(4000+S, 1, S, mix=ord(mix)).tobytes()
        g = F.Guarded(len(d), 0); g.write(d)
        in_off = np.array([g.off], dtype=np.uint64); in_len = np.array([len(d)], dtype=np.uint32)
        stride = helpers.compress_bound(len(d)) + 1024
        gout = F.Guarded(stride, 0)
        ooff = np.array([gout.off], dtype=np.uint64); olen = np.zeros(1, dtype=np.uint32)
        if which in ('zstd', 'l4'):
            fn = emu.emu_zstd_compress
            fn.argtypes = [ctypes.c_void_p]*3 + [ctypes.c_uint32, ctypes.c_int, ctypes.c_uint32] + [ctypes.c_void_p]*3 + [ctypes.c_uint32]
            r = fn(g.base, helpers._vp(in_off), helpers._vp(in_len), 1, 8, 1, gout.base, helpers._vp(ooff), helpers._vp(olen), 131072)
            assert r == 0
            f = gout.read(int(olen[0])); assert f == (o.compress_level(d, 4) if which == 'l4' else o.compress(d)), (S, mix)
        elif which in ('l1','l2','neg'):
            fn = emu.emu_zstd_compress_level
            fn.argtypes = [ctypes.c_void_p]*3 + [ctypes.c_uint32, ctypes.c_int, ctypes.c_uint32] + [ctypes.c_void_p]*3 + [ctypes.c_uint32, ctypes.c_int]
            lvl = 1 if which=='l1' else 2 if which=='l2' else -3
            r = fn(g.base, helpers._vp(in_off), helpers._vp(in_len), 1, 4, 1, gout.base, helpers._vp(ooff), helpers._vp(olen), 131072, lvl)
            assert r == 0
            f = gout.read(int(olen[0])); assert f == o.compress_level(d, lvl), (S, mix)
        elif which.startswith('lazy'):
            lvl = int(which[4:] or 7)              # lazy5 .. lazy10 (default 7): zstd_lazy.h (sort + parse + entropy bodies)
            w = o.compress_lazy(d, lvl)
            if w is None:
                g.close(); gout.close(); continue
            fn = emu.emu_zstd_compress_lazy
            fn.argtypes = [ctypes.c_void_p]*3 + [ctypes.c_uint32, ctypes.c_uint32] + [ctypes.c_void_p]*3 + [ctypes.c_uint32, ctypes.c_int]
            r = fn(g.base, helpers._vp(in_off), helpers._vp(in_len), 1, 1, gout.base, helpers._vp(ooff), helpers._vp(olen), max(len(d), 64), lvl)
            assert r == 0, r
            f = gout.read(int(olen[0])); assert f == w, (S, mix, lvl)
        elif which == 'deflate':
            import zlib
            fn = emu.emu_deflate
            fn.argtypes = [ctypes.c_void_p]*3 + [ctypes.c_uint32] + [ctypes.c_void_p]*5 + [ctypes.c_uint32]
            r = fn(g.base, helpers._vp(in_off), helpers._vp(in_len), 1, gout.base, helpers._vp(ooff), helpers._vp(olen), None, None, 0)
            assert r == 0
            f = gout.read(int(olen[0]))
            c = zlib.compressobj(6, zlib.DEFLATED, -15, 8); assert f == c.compress(d)+c.flush(), (S, mix)
        elif which == 'deflatep':
            # deflateInit2's windowBits / memLevel at random, every level; the output ends at a guard page right behind kmp_deflate_bound_params' room
            import zlib
            lvl, wb, ml = rng.randrange(1, 10), rng.randrange(9, 16), rng.randrange(1, 10)
            gout.close()
            room = len(d) + ((len(d) + 7) >> 3) + ((len(d) + 63) >> 6) + 5 + 18
            gout = F.Guarded(room, 0)
            ooff = np.array([gout.off], dtype=np.uint64)
            fn = emu.emu_deflate_params
            fn.argtypes = [ctypes.c_void_p]*3 + [ctypes.c_uint32] + [ctypes.c_void_p]*5 + [ctypes.c_uint32] + [ctypes.c_int]*4
            fmt = rng.randrange(3)
            r = fn(g.base, helpers._vp(in_off), helpers._vp(in_len), 1, gout.base, helpers._vp(ooff), helpers._vp(olen), None, None, fmt, lvl, wb, ml, 0)
            assert r == 0
            f = gout.read(int(olen[0]))
            c = zlib.compressobj(lvl, zlib.DEFLATED, (-wb, wb, wb + 16)[fmt], ml, 0); assert f == c.compress(d)+c.flush(), (S, mix, lvl, wb, ml, fmt)
        g.close(); gout.close()
print("GUARD OK", which)
