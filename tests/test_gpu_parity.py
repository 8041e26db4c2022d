"""Parity tests proper: the HIP path, called through the C ABI
(include/kompressor_hip.h), against the committed golden vectors, against the
oracle on the same seeded inputs, and -- at BASELINE.json's full batch size --
through size-independent properties (encode -> decode round trip, checksum of
checksums).  Byte work: everything is bit-exact, no tolerance."""
import base64
import hashlib

import numpy as np
import pytest

import helpers
from kompressor_amd import corpus

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def G():
    return helpers.golden()


@pytest.fixture(scope="module")
def batch():
    from kompressor_amd.batch import ZstdBatch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    b = ZstdBatch(max_slices=4096, max_slice_bytes=131072)
    yield b
    b.close()


def gpu_compress(batch, datas):
    n = len(datas)
    lens = np.array([len(d) for d in datas], dtype=np.int32)
    offs = np.zeros(n, dtype=np.int64)
    pos = 0
    for i, d in enumerate(datas):
        offs[i] = pos
        pos += len(d)
    host = np.zeros(pos + 64, dtype=np.uint8)
    for i, d in enumerate(datas):
        host[offs[i]:offs[i] + len(d)] = np.frombuffer(d, dtype=np.uint8)
    src = torch.from_numpy(host).cuda()
    dst, ooff, olen = batch.compress(src, torch.from_numpy(offs).cuda(), torch.from_numpy(lens).cuda())
    torch.cuda.synchronize()
    dst, ooff, olen = dst.cpu().numpy(), ooff.cpu().numpy(), olen.cpu().numpy()
    return [dst[ooff[i]:ooff[i] + olen[i]].tobytes() for i in range(n)]


gpu_compress_kw_last = [None]


def gpu_compress_kw(batch, datas, **kw):
    """gpu_compress with the keyword arguments of ZstdBatch.compress (level, reference, streaming ...)."""
    n = len(datas)
    lens = np.array([len(d) for d in datas], dtype=np.int32)
    offs = np.concatenate([[0], np.cumsum(lens[:-1].astype(np.int64))]).astype(np.int64)
    host = np.frombuffer(b"".join(datas) + bytes(64), dtype=np.uint8).copy()
    dst, ooff, olen = batch.compress(torch.from_numpy(host).cuda(), torch.from_numpy(offs).cuda(), torch.from_numpy(lens).cuda(), **kw)
    torch.cuda.synchronize()
    dd, oo, ol = dst.cpu().numpy(), ooff.cpu().numpy(), olen.cpu().numpy()
    frames = [dd[oo[i]:oo[i] + ol[i]].tobytes() for i in range(n)]
    gpu_compress_kw_last[0] = frames[-1] if frames else None
    return frames


def gpu_decompress(batch, frames, caps):
    n = len(frames)
    lens = np.array([len(f) for f in frames], dtype=np.int32)
    offs = np.zeros(n, dtype=np.int64)
    pos = 16
    for i, f in enumerate(frames):
        offs[i] = pos
        pos += (len(f) + 31) & ~15
    host = np.zeros(pos + 64, dtype=np.uint8)
    for i, f in enumerate(frames):
        host[offs[i]:offs[i] + len(f)] = np.frombuffer(f, dtype=np.uint8)
    cap = torch.tensor(caps, dtype=torch.int32).cuda()
    dst, ooff, olen, st = batch.decompress(torch.from_numpy(host).cuda(), torch.from_numpy(offs).cuda(),
                                           torch.from_numpy(lens).cuda(), cap)
    torch.cuda.synchronize()
    dst, ooff, olen, st = dst.cpu().numpy(), ooff.cpu().numpy(), olen.cpu().numpy(), st.cpu().numpy()
    return [dst[ooff[i]:ooff[i] + olen[i]].tobytes() for i in range(n)], [int(x) for x in st]


def test_golden_config1_2048_slices(G, batch):
    rows = G["config1"]
    S = 65536
    buf = corpus.make(0, len(rows), S)
    frames = gpu_compress(batch, [buf[i * S:(i + 1) * S].tobytes() for i in range(len(rows))])
    bad = [(i, cls) for (i, cls, flen, sha), f in zip(rows, frames) if len(f) != flen or helpers.sha256(f) != sha]
    assert not bad, f"{len(bad)} of {len(rows)} frames differ from libzstd 1.5.7, first: {bad[:5]}"


def test_golden_ladder_ragged_batch(G, batch):
    rows = G["ladder"]
    datas = []
    for r in rows:
        S, k = r["size"], r["index"] - 1000
        datas.append(corpus.make(1000, 8, S)[k * S:(k + 1) * S].tobytes() if S else b"")
    frames = gpu_compress(batch, datas)          # one ragged batch: sizes 0 .. 128 KiB mixed
    for r, f in zip(rows, frames):
        assert len(f) == r["len"] and helpers.sha256(f) == r["sha256"], r
        if "frame" in r:
            assert f == base64.b64decode(r["frame"])


def test_golden_specials_and_config0(G, batch):
    sp = helpers.special_inputs()
    rows = G["special"]
    frames = gpu_compress(batch, [sp[r["name"]] for r in rows])
    for r, f in zip(rows, frames):
        assert len(f) == r["len"] and helpers.sha256(f) == r["sha256"], r["name"]
    d = corpus.make(0, 1, 131072, mix=ord("R")).tobytes()
    f = gpu_compress(batch, [d])[0]
    assert len(f) == 131084 and helpers.sha256(f) == G["config0"]["sha256"]


@pytest.mark.parametrize("team", [4, 8, 16, 32, 64])
def test_every_team_width_against_oracle(team):
    from kompressor_amd.batch import ZstdBatch
    o = helpers.oracle()
    b = ZstdBatch(max_slices=512, max_slice_bytes=65536, team_lanes=team)
    S = 65536
    buf = corpus.make(20000, 256, S)
    datas = [buf[i * S:(i + 1) * S].tobytes() for i in range(256)]
    datas += [buf[i * S:i * S + 1 + (i * 977) % 60000].tobytes() for i in range(128)]      # ragged
    frames = gpu_compress(b, datas)
    b.close()
    for i, (d, f) in enumerate(zip(datas, frames)):
        assert f == o.compress(d), f"team {team} slice {i} len {len(d)}"


def test_repeated_calls_reuse_tables_without_clearing(batch):
    # hash tables are epoch-tagged, never cleared: a second batch must not see the first one's entries
    o = helpers.oracle()
    S = 65536
    for first in (30000, 30064, 30000):
        buf = corpus.make(first, 64, S)
        datas = [buf[i * S:(i + 1) * S].tobytes() for i in range(64)]
        frames = gpu_compress(batch, datas)
        for d, f in zip(datas, frames):
            assert f == o.compress(d)


def test_decoder_kat_and_golden_frames(G, batch):
    kat = base64.b64decode(G["reference_kats"]["zstd_sampleHello_frame_b64"])      # reference ZstdTest.kt:84-91
    outs, st = gpu_decompress(batch, [kat], [64])
    assert st == [0] and outs[0].decode() == "hello compression world"
    d = G["decode_only"]
    outs, st = gpu_decompress(batch, [base64.b64decode(r["frame"]) for r in d], [r["size"] for r in d])
    for r, o, s in zip(d, outs, st):
        assert s == 0 and helpers.sha256(o) == r["plain_sha256"], r


def test_decoder_error_codes(G, batch):
    good = base64.b64decode(next(r["frame"] for r in G["special"] if r["name"] == "ramp_64k"))
    flipped = bytearray(good)
    flipped[len(good) // 2] ^= 0x40
    outs, st = gpu_decompress(batch, [b"\x00" + good[1:], good[:-3], bytes(flipped), good, good], [65536, 65536, 65536, 100, 65536])
    assert st[0] == 10 and st[1] in (20, 72) and st[3] == 70 and st[4] == 0
    assert st[2] != 0 or outs[2] != bytes(range(256)) * 256
    assert outs[4] == bytes(range(256)) * 256


@pytest.mark.timeout(900)
def test_full_batch_roundtrip_and_checksum_of_checksums(monkeypatch):
    """BASELINE.json configs[1] size: 65 536 x 64 KiB.  decode(encode(x)) == x for every slice; the 2 048 slices the
    golden manifest covers hash to the manifest's values inside the big batch (batch position does not change a frame);
    and every one of the 65 536 frames -- from both level-3 parsers --, of configs[4]'s 65 536 DEFLATE streams, and of the
    131 072 slices of rank 0's block of configs[3] (as the 8-GPU run shards it) equals the reference library's, checked
    through one sha256 per 4 096-slice group.  The context's workspace is one arena: creating it takes what it holds."""
    from kompressor_amd.batch import ZstdBatch
    G = helpers.golden()
    n, S = 65536, 65536
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    b = ZstdBatch(max_slices=n, max_slice_bytes=S, table_retry=2)     # (2: always take the path that tries a second arena)
    torch.cuda.synchronize()
    taken = free0 - torch.cuda.mem_get_info()[0]
    # team tables 24 GiB + sequences 8.6 + literals 4.3 + staging words 4.3 GB + small change = 41 GiB of parts, laid out over
    # the arena's default span of 100 GiB (bounded by half of the free memory): what creation holds when it returns is the one
    # arena (the second one, tried on request, has been freed again -- or the first, when the second won); kmp_batch_memory says so
    assert 99 << 30 < taken < 103 << 30, taken / 2 ** 30
    mem = b.memory()
    assert (100 << 30) - (8 << 20) <= mem["arena"] <= 100 << 30 and 40 << 30 < mem["arena_used"] < 43 << 30 and mem["total"] < taken + (1 << 28), mem
    src = torch.empty(n * S, dtype=torch.uint8, device="cuda")
    chunk = 4096
    for c in range(0, n, chunk):
        src[c * S:(c + chunk) * S] = torch.from_numpy(corpus.make(c, chunk, S)).cuda()
    in_off = torch.arange(n, dtype=torch.int64, device="cuda") * S
    in_len = torch.full((n,), S, dtype=torch.int32, device="cuda")
    dst, ooff, olen = b.compress(src, in_off, in_len)
    torch.cuda.synchronize()
    lens = olen.cpu().numpy()
    assert lens.min() >= 13 and lens.max() <= 65546
    head = dst[: 2048 * b.out_stride].cpu().numpy()
    for i, cls, flen, sha in G["config1"]:
        f = head[i * b.out_stride:i * b.out_stride + lens[i]].tobytes()
        assert len(f) == flen and helpers.sha256(f) == sha, i
    cap = torch.full((n,), S, dtype=torch.int32, device="cuda")
    out, o2, l2, st = b.decompress(dst, ooff, olen, cap, out_off=in_off)
    torch.cuda.synchronize()
    assert int(st.abs().sum().item()) == 0
    assert int((l2 != S).sum().item()) == 0
    assert torch.equal(out[: n * S], src)
    ratio = n * S / float(lens.astype(np.int64).sum())
    assert 2.3 < ratio < 2.7, ratio
    # dense packing helper: offsets are the exclusive scan, bytes unchanged
    packed, offs = b.compact(dst, ooff, olen)
    torch.cuda.synchronize()
    offs = offs.cpu().numpy()
    assert offs[0] == 0 and offs[-1] == lens.astype(np.int64).sum()
    assert np.array_equal(np.diff(offs), lens.astype(np.int64))
    for i in (0, 1, 777, n - 1):
        a = packed[offs[i]:offs[i + 1]].cpu().numpy().tobytes()
        assert a == dst[int(ooff[i]):int(ooff[i]) + int(lens[i])].cpu().numpy().tobytes()
    # ALL 65 536 frames against libzstd 1.5.7: sha256 of the frames of every group of 4 096 slices, back to back
    # (tests/golden/fullsize_golden.json, generated by make_golden_fullsize.py from the binary library)
    F = helpers.fullsize_golden()
    assert F["slice_bytes"] == S
    host = packed.cpu().numpy()
    for g, total, sha in F["config1_zstd3"]:
        lo, hi = int(offs[g * F["group"]]), int(offs[(g + 1) * F["group"]])
        assert hi - lo == total and hashlib.sha256(host[lo:hi]).hexdigest() == sha, f"group {g} differs from libzstd 1.5.7"
    del host, packed
    # a second context beside the first gets a bounded arena by itself: half of what is free now, or packed -- never another 100 GiB
    free1 = torch.cuda.mem_get_info()[0]
    b2 = ZstdBatch(max_slices=n, max_slice_bytes=S)
    torch.cuda.synchronize()
    took2 = free1 - torch.cuda.mem_get_info()[0]
    assert took2 <= free1 // 2 + (1 << 30) and b2.memory()["arena"] <= free1 // 2, (took2, free1)
    dst2, ooff2, olen2 = b2.compress(src, in_off, in_len, check=True)
    torch.cuda.synchronize()
    assert torch.equal(olen2, olen)
    b2.close()
    del dst2
    # ... and all 65 536 raw DEFLATE level-6 streams of configs[4] against zlib, the same way
    ddst, doff, dlen = b.deflate(src, in_off, in_len)
    dpacked, doffs = b.compact(ddst, doff, dlen)
    torch.cuda.synchronize()
    assert b.status() == (0, 0)
    doffs = doffs.cpu().numpy()
    host = dpacked.cpu().numpy()
    for g, total, sha in F["config4_deflate6"]:
        lo, hi = int(doffs[g * F["group"]]), int(doffs[(g + 1) * F["group"]])
        assert hi - lo == total and hashlib.sha256(host[lo:hi]).hexdigest() == sha, f"group {g} differs from zlib {F['zlib']}"
    del host, dpacked, ddst
    # configs[3]: rank 0's block of the 8-GPU run (131 072 text / binary slices) in two batches of 65 536, reusing the context
    for half in range(2):
        for c in range(0, n, chunk):
            src[c * S:(c + chunk) * S] = torch.from_numpy(corpus.make(half * n + c, chunk, S, corpus.MIX_TEXT_BINARY)).cuda()
        dst, ooff, olen = b.compress(src, in_off, in_len, dst, ooff, olen, check=True)
        packed, offs = b.compact(dst, ooff, olen)
        torch.cuda.synchronize()
        offs = offs.cpu().numpy()
        host = packed.cpu().numpy()
        for g, total, sha in F["config3_zstd3_rank0"][half * 16:(half + 1) * 16]:
            gl = g - half * 16
            lo, hi = int(offs[gl * F["group"]]), int(offs[(gl + 1) * F["group"]])
            assert hi - lo == total and hashlib.sha256(host[lo:hi]).hexdigest() == sha, f"configs[3] group {g} differs from libzstd 1.5.7"
        del host, packed
    b.close()


@pytest.mark.timeout(900)
def test_ablation_build_gives_the_same_frames(monkeypatch):
    """The ablation build of the library (-DKMP_ABLATIONS: libkompressor_hip_abl.so, a second handle in this process) carries what
    the product build compiles out: the split-phase parser (zstd_match2.h, KMP_MATCH_V2), the fused kernel (KMP_FUSE), the trial of
    launch settings (KMP_ZSTD_AUTOTUNE).  All 65 536 configs[1] frames from each equal libzstd 1.5.7's (one sha256 per 4 096-slice
    group, tests/golden/fullsize_golden.json)."""
    from kompressor_amd.batch import ZstdBatch
    n, S = 65536, 65536
    F = helpers.fullsize_golden()
    src = torch.empty(n * S, dtype=torch.uint8, device="cuda")
    chunk = 4096
    for c in range(0, n, chunk):
        src[c * S:(c + chunk) * S] = torch.from_numpy(corpus.make(c, chunk, S)).cuda()
    in_off = torch.arange(n, dtype=torch.int64, device="cuda") * S
    in_len = torch.full((n,), S, dtype=torch.int32, device="cuda")

    def check(b, what, dst=None, ooff=None, olen=None):
        dst, ooff, olen = b.compress(src, in_off, in_len, dst, ooff, olen, check=True)
        packed, offs = b.compact(dst, ooff, olen)
        torch.cuda.synchronize()
        offs = offs.cpu().numpy(); host = packed.cpu().numpy()
        for g, total, sha in F["config1_zstd3"]:
            lo, hi = int(offs[g * F["group"]]), int(offs[(g + 1) * F["group"]])
            assert hi - lo == total and hashlib.sha256(host[lo:hi]).hexdigest() == sha, f"{what}: group {g} differs from libzstd 1.5.7"
        return dst, ooff, olen

    for what, env in (("split-phase parser", {"KMP_MATCH_V2": "2"}), ("fused kernel", {"KMP_FUSE": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        b = ZstdBatch(max_slices=n, max_slice_bytes=S, ablations=True, table_span_gib=0)      # (the packed arena)
        for k in env:
            monkeypatch.delenv(k)
        try:
            check(b, what)
        finally:
            b.close()
    # the trial of launch settings: "one launch of each kernel" on the first batch of this size, "two chunks" on the second, then the
    # faster -- all of them give the same frames
    monkeypatch.setenv("KMP_ZSTD_AUTOTUNE", "1")
    b = ZstdBatch(max_slices=n, max_slice_bytes=S, ablations=True, table_span_gib=0)
    monkeypatch.delenv("KMP_ZSTD_AUTOTUNE")
    try:
        seen = set()
        bufs = (None, None, None)
        for r in range(3):
            bufs = check(b, f"autotune batch {r}", *bufs)
            seen.add(b.last_chunks())
        assert seen == {1, 2}
    finally:
        b.close()


def test_streaming_abi_one_shot_like_the_reference(G):
    """ZstdCompressor(3).transform(bytes) / ZstdDecompressor().transform(frame) with the
    reference's driver loop (SliceTransform.kt:33-45) over the C ABI."""
    from kompressor_amd import ZstdCompressor, ZstdDecompressor
    sp = helpers.special_inputs()
    byname = {r["name"]: r for r in G["special"]}
    for name in ("empty", "one_byte", "hello", "abc_100", "ramp_64k", "zeros_128k", "two_symbols"):
        d = sp[name]
        f = ZstdCompressor(compression_level=3).transform_bytes(d)
        assert helpers.sha256(f) == byname[name]["sha256"], name
        assert ZstdDecompressor().transform_bytes(f) == d
    # configs[0]: 128 KiB random -> 131 084-byte frame (9 B header + raw block), 13 107-byte output chunks
    d = corpus.make(0, 1, 131072, mix=ord("R")).tobytes()
    f = ZstdCompressor(3).transform_bytes(d)
    assert len(f) == 131084 and helpers.sha256(f) == G["config0"]["sha256"]
    assert ZstdDecompressor().transform_bytes(f) == d
    # tinySample (ZstdTest.kt:19-25): 9000 random bytes round trip
    rnd = np.random.default_rng(1).integers(0, 256, 9000, dtype=np.uint8).tobytes()
    c = ZstdCompressor(3)
    assert ZstdDecompressor().transform_bytes(c.transform_bytes(rnd)) == rnd
    # a context is reusable for the next slice once a frame is flushed
    assert c.transform_bytes(sp["hello"]) == base64.b64decode(byname["hello"]["frame"])
    # reference decode KAT through the streaming ABI
    kat = base64.b64decode(G["reference_kats"]["zstd_sampleHello_frame_b64"])
    assert ZstdDecompressor().transform_bytes(kat) == b"hello compression world"
    # errors surface like the reference's IllegalStateException text
    with pytest.raises(RuntimeError, match="Bad zstd result code -40: Unsupported parameter"):
        ZstdCompressor(compression_level=19)                                          # (levels up to 10 are served where they are greedy / lazy / lazy2)
    # levels 1 and 2 beyond their windows (512 KiB, 1 MiB) were refused until round 4; now: the frame the reference's driver gets
    for lvl, nbytes in ((2, (1 << 20) + 1), (1, (512 << 10) + 1)):
        data = corpus.make(4242 + lvl, 1, nbytes, mix=ord("T")).tobytes()
        fz = ZstdCompressor(compression_level=lvl).transform_bytes(data)
        assert fz == helpers.oracle().compress_fast_buffered(data, lvl, stream=3), lvl
        assert ZstdDecompressor().transform_bytes(fz) == data
    with pytest.raises(RuntimeError, match="Unknown frame descriptor"):
        ZstdDecompressor().transform_bytes(b"\x00" * 32)
    # sampleRoundtrip (ZstdTest.kt:27-32): 1 MiB + 3 random bytes -> frame of several blocks, and back
    big = np.random.default_rng(2).integers(0, 256, (1 << 20) + 3, dtype=np.uint8).tobytes()
    c2 = ZstdCompressor(3)
    fb = c2.transform_bytes(big)
    # (above 128 KiB the driver's output slices are smaller than ZSTD_compressBound: libzstd stages the input in chunks)
    assert fb == helpers.oracle().compress_buffered(big) and ZstdDecompressor().transform_bytes(fb) == big
    txt = corpus.make(4242, 1, 700001, mix=ord("T")).tobytes()
    ft = c2.transform_bytes(txt)                                  # the context grows its staging once, then is reused
    assert ft == helpers.oracle().compress_buffered(txt) and ZstdDecompressor().transform_bytes(ft) == txt
    assert c2.transform_bytes(sp["hello"]) == base64.b64decode(byname["hello"]["frame"])
    # beyond the level-3 window (2 MiB) and libzstd's staging buffer (2 MiB + 128 KiB): the window slides as libzstd's does
    far = (corpus.make(4243, 1, 1500000, mix=ord("S")).tobytes() + bytes(50000)) * 2
    ff = ZstdCompressor(3).transform_bytes(far)
    assert ff == helpers.oracle().compress_buffered(far) and ZstdDecompressor().transform_bytes(ff) == far


def test_multiblock_frames_match_libzstd():
    """Slices above 128 KiB: frames of several blocks, bit-identical to libzstd 1.5.7 -- block
    pre-splitter, repcodes / Huffman table carried between blocks, raw, RLE and treeless-literals blocks --
    and back through the GPU decoder."""
    from kompressor_amd.batch import ZstdBatch
    rows = helpers.multiblock_golden()["rows"]
    inputs = helpers.multiblock_inputs()
    datas = [d for _, d in inputs]
    # small slices through the same context: the block path must produce the single-block frames too
    small = [corpus.make(77000 + k, 1, s).tobytes() for k, s in enumerate([0, 1, 7, 300, 65536, 131072])]
    o = helpers.oracle()
    b = ZstdBatch(max_slices=len(datas) + len(small), max_slice_bytes=2 << 20)
    try:
        frames = gpu_compress(b, datas + small)
        for (name, d), row, f in zip(inputs, rows, frames):
            assert len(f) == row["len"] and helpers.sha256(f) == row["sha256"], name
        for d, f in zip(small, frames[len(datas):]):
            assert f == o.compress(d), len(d)
        # several slices per wave, and the same steps as separate launches per round of blocks (experiment switches of the ablation build)
        import os
        # (the switches are read when a context is created)
        for key, val in (("KMP_BIG_SLICES_PER_WAVE", "4"), ("KMP_BIG_ROUNDS", "1")):
            os.environ[key] = val
            try:
                b2 = ZstdBatch(max_slices=len(datas) + len(small), max_slice_bytes=2 << 20, ablations=True)
            finally:
                del os.environ[key]
            try:
                again = gpu_compress(b2, datas + small)
                assert again == frames, key
                if key == "KMP_BIG_ROUNDS":
                    assert b2.lib.kmp_batch_last_rounds(b2._h) >= 16
            finally:
                b2.close()
        back, st = gpu_decompress(b, frames, [max(len(d), 1) for d in datas + small])
        assert st == [0] * len(frames)
        assert back == datas + small
    finally:
        b.close()


def test_fuzz_ragged_sizes_against_oracle(batch):
    """2 048 slices of arbitrary sizes 0 .. 128 KiB (every class, byte runs, short periods, class changes inside a
    slice) through the batched path, each compared with the oracle's frame; then a ragged batch of 96 slices of
    128 KiB+1 .. 1.5 MiB through the block-chain path."""
    import random
    from kompressor_amd.batch import ZstdBatch
    rng = random.Random(777)
    o = helpers.oracle()

    def piece(n):
        r = rng.random()
        if r < 0.08:
            return bytes([rng.randrange(256)]) * n
        if r < 0.16:
            unit = corpus.make(rng.randrange(1 << 30), 1, rng.choice([2, 3, 7, 40, 300]), mix=ord("R")).tobytes()
            return (unit * (n // len(unit) + 1))[:n]
        return corpus.make(rng.randrange(1 << 30), 1, n, mix=ord(rng.choice("TXSBDIZR"))).tobytes()

    def blob(total):
        parts, have = [], 0
        while have < total:
            n = min(total - have, rng.choice([1, 5, 64, 500, 4000, 20000, 70000, 200000]))
            parts.append(piece(n))
            have += n
        return b"".join(parts)

    sizes = [rng.choice([rng.randrange(0, 64), rng.randrange(0, 2000), rng.randrange(0, 20000), rng.randrange(0, 131073)]) for _ in range(2048)]
    datas = [blob(sz) for sz in sizes]
    for i, (d, f) in enumerate(zip(datas, gpu_compress(batch, datas))):
        assert f == o.compress(d), (i, len(d))
    big = [blob(rng.randrange(131073, 1536 * 1024)) for _ in range(96)]
    b = ZstdBatch(max_slices=len(big), max_slice_bytes=2 << 20)
    try:
        frames = gpu_compress(b, big)
        for i, (d, f) in enumerate(zip(big, frames)):
            assert f == o.compress(d), (i, len(d))
        back, st = gpu_decompress(b, frames, [len(d) for d in big])
        assert st == [0] * len(big) and back == big
    finally:
        b.close()


def test_decoder_with_raw_dictionary(batch):
    """ZstdDecompressor(dictionary) over frames libzstd 1.5.7 compressed with a raw-content dictionary: batch entry point
    (one dictionary for the batch) and the streaming entry point; without the dictionary the frames fail."""
    import json
    import os
    from kompressor_amd import ZstdDecompressor
    rows = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "zstd_dict_golden.json")))["rows"]
    for (name, d, plain), row in zip(helpers.dict_cases(), rows):
        frame = base64.b64decode(row["frame"])
        lens = torch.tensor([len(frame)] * 3, dtype=torch.int32).cuda()
        host = np.zeros(3 * 70000 + 64, dtype=np.uint8)
        for k in range(3):
            host[k * 70000:k * 70000 + len(frame)] = np.frombuffer(frame, dtype=np.uint8)
        offs = (torch.arange(3, dtype=torch.int64) * 70000).cuda()
        cap = torch.full((3,), max(len(plain), 1), dtype=torch.int32).cuda()
        dd = torch.from_numpy(np.frombuffer(d, dtype=np.uint8).copy()).cuda()
        out, ooff, olen, st = batch.decompress(torch.from_numpy(host).cuda(), offs, lens, cap, dictionary=dd)
        torch.cuda.synchronize()
        out, ooff, olen, st = out.cpu().numpy(), ooff.cpu().numpy(), olen.cpu().numpy(), st.cpu().numpy()
        for k in range(3):
            assert int(st[k]) == 0 and out[ooff[k]:ooff[k] + olen[k]].tobytes() == plain, (name, k)
        assert ZstdDecompressor(dictionary=d).transform_bytes(frame) == plain, name
        try:
            wrong = ZstdDecompressor().transform_bytes(frame)
        except RuntimeError:
            wrong = None
        assert wrong != plain, name
    # zstd's dictionary magic (EC30A437) with a damaged header behind it is refused on both sides, not taken as raw content
    # (well-formed dictionaries of that format: test_formatted_dictionaries below)
    fd = b"\x37\xa4\x30\xec" + bytes(range(200)) * 4
    dd = torch.from_numpy(np.frombuffer(fd, dtype=np.uint8).copy()).cuda()
    with pytest.raises(RuntimeError, match="header is damaged"):
        batch.decompress(torch.from_numpy(host).cuda(), offs, lens, cap, dictionary=dd)
    with pytest.raises(RuntimeError, match="header is damaged"):
        batch.compress(torch.from_numpy(host).cuda(), offs, torch.tensor([100] * 3, dtype=torch.int32).cuda(), dictionary=fd)


def test_formatted_dictionaries(batch):
    """ZstdCompressor(3, dictionary) / ZstdDecompressor(dictionary) with dictionaries in zstd's own format (magic EC30A437: Huffman and FSE
    tables, repeat offsets, an ID, then content) -- two trained by libzstd's ZDICT, eight built to leave symbols out of their tables: every
    committed frame of libzstd 1.5.7 (tests/golden/zstd_dict_formatted_golden.json) from the batch call, decoded back by the batch call
    together with libzstd's level-19 frames, the streaming entry points, a mismatching dictionary."""
    import hashlib
    from kompressor_amd import ZstdCompressor, ZstdDecompressor
    cases = helpers.formatted_dict_cases()
    nframes = 0
    for name, d, inputs, row in cases:
        n = len(inputs)
        lens = [len(p) for p in inputs]
        stride = 131072 + 512
        host = np.zeros(n * stride + 64, dtype=np.uint8)
        for k, p in enumerate(inputs):
            host[k * stride:k * stride + len(p)] = np.frombuffer(p, dtype=np.uint8)
        src = torch.from_numpy(host).cuda()
        offs = (torch.arange(n, dtype=torch.int64) * stride).cuda()
        dst, ooff, olen = batch.compress(src, offs, torch.tensor(lens, dtype=torch.int32).cuda(), dictionary=d)
        torch.cuda.synchronize()
        dsth, ooffh, olenh = dst.cpu().numpy(), ooff.cpu().numpy(), olen.cpu().numpy()
        frames = [dsth[int(ooffh[k]):int(ooffh[k]) + int(olenh[k])].tobytes() for k in range(n)]
        for p, f, (flen, fsha) in zip(inputs, frames, row["frames"]):
            assert len(f) == flen and hashlib.sha256(f).hexdigest() == fsha, (name, len(p))
            nframes += 1
        # and back, with libzstd's level-19 frames (its own choice of tables and "repeat" modes) among them
        frames += [base64.b64decode(x) for x in row["level19"]]
        plain = inputs + [inputs[i] for i in row["level19_inputs"]]
        m = len(frames)
        fh = np.zeros(m * stride + 64, dtype=np.uint8)
        for k, f in enumerate(frames):
            fh[k * stride:k * stride + len(f)] = np.frombuffer(f, dtype=np.uint8)
        foffs = (torch.arange(m, dtype=torch.int64) * stride).cuda()
        dd = torch.from_numpy(np.frombuffer(d, dtype=np.uint8).copy()).cuda()
        out, o2, l2, st = batch.decompress(torch.from_numpy(fh).cuda(), foffs, torch.tensor([len(f) for f in frames], dtype=torch.int32).cuda(),
                                           torch.tensor([max(len(p), 1) for p in plain], dtype=torch.int32).cuda(), dictionary=dd)
        torch.cuda.synchronize()
        out, o2, l2, st = out.cpu().numpy(), o2.cpu().numpy(), l2.cpu().numpy(), st.cpu().numpy()
        for k, p in enumerate(plain):
            assert int(st[k]) == 0 and out[int(o2[k]):int(o2[k]) + int(l2[k])].tobytes() == p, (name, k)
    assert nframes == 200
    # the streaming entry points (what the JNI binds): a context with the dictionary loaded, one slice per call
    name, d, inputs, row = cases[0]
    for i in (4, 9, 14, 17):
        f = ZstdCompressor(dictionary=d).transform_bytes(inputs[i])
        assert hashlib.sha256(f).hexdigest() == row["frames"][i][1], (name, i)
        assert ZstdDecompressor(dictionary=d).transform_bytes(f) == inputs[i]
    # a frame that names this dictionary's ID, decoded with another dictionary / a raw one / none: libzstd's "Dictionary mismatch"
    f = ZstdCompressor(dictionary=d).transform_bytes(inputs[12])
    for other in (cases[1][1], b"raw content, no magic " * 8, None):
        with pytest.raises(RuntimeError, match="Dictionary mismatch"):
            ZstdDecompressor(dictionary=other).transform_bytes(f)


def test_compress_with_raw_dictionary(batch):
    """ZstdCompressor(3, dictionary): every seeded (dictionary, input) pair against the frame libzstd 1.5.7 wrote (both
    parser variants), a 512-slice batch sharing one dictionary against the oracle, and the reference's dictionaryRoundtrip
    (ZstdTest.kt:49-65) through the streaming entry points."""
    import json
    import os
    from kompressor_amd import ZstdCompressor, ZstdDecompressor
    G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "zstd_dict_golden.json")))["compress"]
    n_checked = 0
    for (d, plain), (dsz, psz, tag, flen, sha) in zip(helpers.dict_compress_cases(), G):
        if dsz > 130560:
            continue
        src = torch.from_numpy(np.frombuffer(plain + bytes(64), dtype=np.uint8).copy()).cuda()
        off = torch.zeros(1, dtype=torch.int64, device="cuda")
        ln = torch.tensor([len(plain)], dtype=torch.int32, device="cuda")
        dst, ooff, olen = batch.compress(src, off, ln, dictionary=d)
        torch.cuda.synchronize()
        f = dst[: int(olen[0])].cpu().numpy().tobytes()
        assert len(f) == flen and helpers.sha256(f) == sha, (dsz, psz)
        n_checked += 1
    assert n_checked >= 140
    # one dictionary, many slices of both size classes
    o = helpers.oracle()
    d = corpus.make(5550, 1, 20000, mix=ord("T")).tobytes()
    datas = [corpus.make(5600 + i, 1, (65536, 12000, 131072, 300)[i % 4], mix=ord("TXS"[i % 3])).tobytes() for i in range(512)]
    lens = np.array([len(x) for x in datas], dtype=np.int32)
    offs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.int64)]).astype(np.int64)
    host = np.frombuffer(b"".join(datas) + bytes(64), dtype=np.uint8).copy()
    dst, ooff, olen = batch.compress(torch.from_numpy(host).cuda(), torch.from_numpy(offs).cuda(), torch.from_numpy(lens).cuda(), dictionary=d)
    torch.cuda.synchronize()
    hd, ho, hl = dst.cpu().numpy(), ooff.cpu().numpy(), olen.cpu().numpy()
    for i, x in enumerate(datas):
        assert hd[ho[i]:ho[i] + hl[i]].tobytes() == o.compress_dict(x, d)[0], i
    # dictionaryRoundtrip: smaller with the dictionary, decodes only with it
    sample = datas[1]
    with_d = ZstdCompressor(3, dictionary=d).transform_bytes(sample)
    assert with_d == o.compress_dict(sample, d)[0] and len(with_d) < len(ZstdCompressor(3).transform_bytes(sample))
    assert ZstdDecompressor(dictionary=d).transform_bytes(with_d) == sample
    try:
        wrong = ZstdDecompressor().transform_bytes(with_d)
    except RuntimeError:
        wrong = None
    assert wrong != sample


def test_levels_1_and_2(batch):
    """ZstdCompressor(level = 1 / 2) (the Ktor encoder's default is 1: ZstdContentEncoder.kt:11): the whole size ladder in
    one ragged batch and 256 slices of the 64 KiB mix per level against libzstd 1.5.7, decoded back on the GPU, and the
    streaming entry point."""
    from kompressor_amd import ZstdCompressor, ZstdDecompressor
    G = helpers.levels_golden()

    def run(datas, level):
        n = len(datas)
        lens = np.array([len(d) for d in datas], dtype=np.int32)
        offs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.int64)]).astype(np.int64) if n > 1 else np.zeros(1, dtype=np.int64)
        host = np.frombuffer(b"".join(datas) + bytes(64), dtype=np.uint8).copy()
        dst, ooff, olen = batch.compress(torch.from_numpy(host).cuda(), torch.from_numpy(offs).cuda(), torch.from_numpy(lens).cuda(), level=level)
        torch.cuda.synchronize()
        hd, ho, hl = dst.cpu().numpy(), ooff.cpu().numpy(), olen.cpu().numpy()
        return [hd[ho[i]:ho[i] + hl[i]].tobytes() for i in range(n)]

    ladder = [corpus.make(1000, 8, S)[k * S:(k + 1) * S].tobytes() if S else b"" for S, k, *_ in G["ladder"]]
    S = 65536
    buf = corpus.make(0, 256, S)
    mix = [buf[i * S:(i + 1) * S].tobytes() for i in range(256)]
    for lvl in (1, 2):
        frames = run(ladder, lvl)
        for row, f in zip(G["ladder"], frames):
            flen, sha = (row[2], row[3]) if lvl == 1 else (row[4], row[5])
            assert len(f) == flen and helpers.sha256(f) == sha, (row[0], row[1], lvl)
        back, st = gpu_decompress(batch, frames, [max(len(d), 1) for d in ladder])
        assert st == [0] * len(frames) and back == ladder
        frames = run(mix, lvl)
        for row, f in zip(G["config1"], frames):
            flen, sha = (row[1], row[2]) if lvl == 1 else (row[3], row[4])
            assert len(f) == flen and helpers.sha256(f) == sha, (row[0], lvl)
    o = helpers.oracle()
    f1 = ZstdCompressor(compression_level=1).transform_bytes(mix[3])
    assert f1 == o.compress_level(mix[3], 1) and ZstdDecompressor().transform_bytes(f1) == mix[3]


@pytest.mark.timeout(900)
def test_differential_fuzz_against_the_live_library():
    """tools/fuzz_gpu.py / fuzz_gpu_big.py at a small size (their long runs are in profiles/r03_fuzz.txt): ragged stress inputs and
    corpus slices through every level-3 path, levels 1, 2, 4 and three negative ones, two dictionary sizes and three DEFLATE levels,
    and frames of several blocks in seven forms -- every frame against the binary libzstd 1.5.7 / zlib of THIS machine.  The
    GPU boxes carry that library: where it is missing the test fails (helpers.require_live_libzstd), it does not skip."""
    import os
    import subprocess
    import sys
    helpers.require_live_libzstd()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for tool, args in (("fuzz_gpu.py", ["7", "6000"]), ("fuzz_gpu_big.py", ["7", "300"])):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", tool)] + args, capture_output=True, text=True, timeout=420)
        assert r.returncode == 0 and "FUZZ OK" in r.stdout, (tool, r.stdout[-1500:], r.stderr[-1500:])


def test_negative_levels(batch):
    """ZstdCompressor(level < 0) (libzstd's "fast" strategy with a step of 1 - level and literals left uncompressed): the whole size
    ladder in one ragged batch and 64 slices of the 64 KiB mix per level against libzstd 1.5.7, decoded back on the GPU; the host-batch
    call and the streaming entry point take the level."""
    from kompressor_amd import ZstdCompressor, ZstdDecompressor
    from kompressor_amd.batch import compress_host_batch
    G = helpers.neg_levels_golden()

    def run(datas, level):
        n = len(datas)
        lens = np.array([len(d) for d in datas], dtype=np.int32)
        offs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.int64)]).astype(np.int64) if n > 1 else np.zeros(1, dtype=np.int64)
        host = np.frombuffer(b"".join(datas) + bytes(64), dtype=np.uint8).copy()
        dst, ooff, olen = batch.compress(torch.from_numpy(host).cuda(), torch.from_numpy(offs).cuda(), torch.from_numpy(lens).cuda(), level=level, check=True)
        torch.cuda.synchronize()
        hd, ho, hl = dst.cpu().numpy(), ooff.cpu().numpy(), olen.cpu().numpy()
        return [hd[ho[i]:ho[i] + hl[i]].tobytes() for i in range(n)]

    ladder = [corpus.make(1000, 8, r[0])[r[1] * r[0]:(r[1] + 1) * r[0]].tobytes() if r[0] else b"" for r in G["ladder"]]
    S = 65536
    buf = corpus.make(0, 64, S)
    mix = [buf[i * S:(i + 1) * S].tobytes() for i in range(64)]
    for j, lvl in enumerate(G["levels"]):
        frames = run(ladder, lvl)
        for row, f in zip(G["ladder"], frames):
            assert len(f) == row[2 + 2 * j] and helpers.sha256(f)[:32] == row[3 + 2 * j], (row[0], row[1], lvl)
        back, st = gpu_decompress(batch, frames, [max(len(d), 1) for d in ladder])
        assert st == [0] * len(frames) and back == ladder
        frames = run(mix, lvl)
        for row, f in zip(G["config1"], frames):
            assert len(f) == row[1 + 2 * j] and helpers.sha256(f)[:32] == row[2 + 2 * j], (row[0], lvl)
    o = helpers.oracle()
    assert compress_host_batch(mix[:8], level=-5) == [o.compress_level(d, -5) for d in mix[:8]]
    f1 = ZstdCompressor(compression_level=-2).transform_bytes(mix[3])
    assert f1 == o.compress_level(mix[3], -2) and ZstdDecompressor().transform_bytes(f1) == mix[3]
    # above 128 KiB (up to the 512 KiB window of these levels): the frame the reference's driver gets, through the streaming entry point
    big = corpus.make(77, 1, 300000).tobytes()
    f2 = ZstdCompressor(compression_level=-3).transform_bytes(big)
    assert f2 == o.compress_level_big(big, -3, stream=3) and ZstdDecompressor().transform_bytes(f2) == big


def test_level_4_where_it_is_double_fast(batch, monkeypatch):
    """ZstdCompressor(level = 4) for slices above 16 KiB up to 128 KiB (libzstd runs that size class of level 4 as the double-fast
    parse with hash 17 / chain 17 / minimum match 4): the ragged sizes in one batch and 256 slices of the 64 KiB mix against
    libzstd 1.5.7, decoded back on the GPU; a slice of 16 KiB or less in such a batch (strategy "greedy" there: zstd_lazy.h) comes out
    right beside its neighbours; level 3 still works on the same context afterwards (its tables are another set); and the host-batch
    call takes the level."""
    from kompressor_amd.batch import compress_host_batch
    G = helpers.level4_golden()

    def run(datas, level, check=True):
        n = len(datas)
        lens = np.array([len(d) for d in datas], dtype=np.int32)
        offs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.int64)]).astype(np.int64) if n > 1 else np.zeros(1, dtype=np.int64)
        host = np.frombuffer(b"".join(datas) + bytes(64), dtype=np.uint8).copy()
        dst, ooff, olen = batch.compress(torch.from_numpy(host).cuda(), torch.from_numpy(offs).cuda(), torch.from_numpy(lens).cuda(), level=level, check=check)
        torch.cuda.synchronize()
        hd, ho, hl = dst.cpu().numpy(), ooff.cpu().numpy(), olen.cpu().numpy()
        return [hd[ho[i]:ho[i] + hl[i]].tobytes() for i in range(n)]

    ladder = [corpus.make(1000, 8, S)[k * S:(k + 1) * S].tobytes() for S, k, *_ in G["ladder"]]
    frames = run(ladder, 4)
    for (S, k, flen, sha), f in zip(G["ladder"], frames):
        assert len(f) == flen and helpers.sha256(f) == sha, (S, k)
    back, st = gpu_decompress(batch, frames, [len(d) for d in ladder])
    assert st == [0] * len(frames) and back == ladder
    S = 65536
    buf = corpus.make(0, 256, S)
    mix = [buf[i * S:(i + 1) * S].tobytes() for i in range(256)]
    frames4 = run(mix, 4)
    for (i, flen, sha), f in zip(G["config1"], frames4):
        assert len(f) == flen and helpers.sha256(f) == sha, i
    # a batch with slices of level 4's "greedy" size class (16 KiB or less: parsed by the kernels of levels 5 .. 10 since round 4) among the others
    batch.status()
    mixed = [mix[0], mix[1][:16384], mix[2], b"", mix[3][:20000], mix[4][:700], mix[5][:7]]
    frames = run(mixed, 4)
    o = helpers.oracle()
    assert frames[0] == frames4[0] and frames[2] == frames4[2] and frames[4] == o.compress_level(mixed[4], 4)
    for i in (1, 5, 6):
        assert frames[i] == o.compress_lazy(mixed[i], 4), i
    assert frames[3] == bytes.fromhex("28b52ffd2000010000")             # (the frame of an empty input)
    # level 3 on the same context afterwards
    G3 = helpers.golden()
    frames3 = run(mix[:16], 3)
    for (i, cls, flen, sha), f in zip(G3["config1"][:16], frames3):
        assert len(f) == flen and helpers.sha256(f) == sha, i
    assert compress_host_batch(mix[:8], level=4) == frames4[:8]
    # the streaming entry point (what ZstdCompressor(4).transform(bytes) binds): served for that size class, refused below it
    from kompressor_amd import ZstdCompressor
    assert ZstdCompressor(compression_level=4).transform_bytes(mix[5]) == frames4[5]
    assert ZstdCompressor(compression_level=4).transform_bytes(mix[5][:9000]) == o.compress_lazy(mix[5][:9000], 4)
    # frames of several blocks (above 256 KiB level 4 is double-fast again: window 21, chain 18, hash 18) and streams, through the
    # streaming entry point as the reference's driver calls it; 128 - 256 KiB is a "greedy" class: refused
    o = helpers.oracle()
    big = corpus.make(78, 1, 700000).tobytes()
    assert ZstdCompressor(compression_level=4).transform_bytes(big) == o.compress_buffered(big, True, level=4)
    with pytest.raises(Exception):
        ZstdCompressor(compression_level=4).transform_bytes(big[:200000])
    # many slices through few teams: every team parses four slices per batch on tables it never clears (epoch tags), three batches
    from kompressor_amd.batch import ZstdBatch
    monkeypatch.setenv("KMP_L4_TEAMS", "64")
    small = ZstdBatch(max_slices=4096, max_slice_bytes=131072)
    monkeypatch.delenv("KMP_L4_TEAMS")
    try:
        lens = torch.full((256,), S, dtype=torch.int32, device="cuda")
        offs = torch.arange(256, dtype=torch.int64, device="cuda") * S
        dev = torch.from_numpy(buf).cuda()
        for _ in range(3):
            dst, ooff, olen = small.compress(dev, offs, lens, level=4, check=True)
            torch.cuda.synchronize()
            hd, ho, hl = dst.cpu().numpy(), ooff.cpu().numpy(), olen.cpu().numpy()
            assert [hd[ho[i]:ho[i] + hl[i]].tobytes() for i in range(256)] == frames4
    finally:
        small.close()


def test_levels_5_to_10(batch):
    """ZstdCompressor(level = 5 .. 10): libzstd's strategies greedy / lazy / lazy2 with its row-based match finder (hash chains up to 16 KiB):
    the committed frames of libzstd 1.5.7 (tests/golden/zstd_lazy_levels_golden.json: every class, ragged sizes to 128 KiB, long runs,
    random bytes) from the batch call, decoded back; levels 9 and 10 refuse slices up to 16 KiB (another strategy there) beside serving
    the others; the host-batch call and the streaming entry point take the levels."""
    import hashlib
    from kompressor_amd import ZstdCompressor, ZstdDecompressor
    from kompressor_amd.batch import compress_host_batch
    G = helpers.lazy_levels_golden(); inputs = helpers.lazy_level_inputs()
    n = len(inputs); stride = 131072 + 512
    host = np.zeros(n * stride + 64, dtype=np.uint8)
    for k, p in enumerate(inputs):
        host[k * stride:k * stride + len(p)] = np.frombuffer(p, dtype=np.uint8)
    src = torch.from_numpy(host).cuda(); offs = (torch.arange(n, dtype=torch.int64) * stride).cuda()
    lens = torch.tensor([len(p) for p in inputs], dtype=torch.int32).cuda()
    checked = 0
    for lvl in range(5, 11):
        batch.status()
        dst, ooff, olen = batch.compress(src, offs, lens, level=lvl)
        torch.cuda.synchronize()
        d, oo, ol = dst.cpu().numpy(), ooff.cpu().numpy(), olen.cpu().numpy()
        frames = [d[int(oo[k]):int(oo[k]) + int(ol[k])].tobytes() for k in range(n)]
        refused = 0
        for p, f, (flen, fsha) in zip(inputs, frames, G["frames"][str(lvl)]):
            if lvl >= 9 and 8 <= len(p) <= 16384:
                assert f == b"", (lvl, len(p)); refused += 1
                continue
            assert len(f) == flen and hashlib.sha256(f).hexdigest() == fsha, (lvl, len(p))
            checked += 1
        bits = batch.status()[1]
        assert bool(bits & 4) == (refused > 0) and not (bits & 3), (lvl, bits)
        served = [(p, f) for p, f in zip(inputs, frames) if f]
        back, st = gpu_decompress(batch, [f for _, f in served], [max(len(p), 1) for p, _ in served])
        assert st == [0] * len(served) and back == [p for p, _ in served], lvl
    assert checked >= 560                                   # (672 frames less the slices levels 9 and 10 refuse)
    some = [inputs[i] for i in (5, 7, 8, 10, 30, 46)]
    o = helpers.oracle()
    assert compress_host_batch(some, level=7) == [o.compress_lazy(p, 7) for p in some]
    for lvl, i in ((5, 8), (6, 10), (10, 9), (8, 3)):
        f = ZstdCompressor(compression_level=lvl).transform_bytes(inputs[i])
        assert f == o.compress_lazy(inputs[i], lvl), (lvl, i)
        assert ZstdDecompressor().transform_bytes(f) == inputs[i]
    with pytest.raises(RuntimeError, match="Unsupported parameter"):
        ZstdCompressor(compression_level=9).transform_bytes(inputs[5])          # 5 000 bytes at level 9: libzstd's "btlazy2"
    with pytest.raises(RuntimeError, match="Unsupported parameter"):
        ZstdCompressor(compression_level=11)


def test_streaming_frames_finish_false_then_true():
    """The reference's streaming callers (SliceTransformRawSource.kt:32-55): input in several finish = false calls, then
    finish = true.  libzstd does not know the size then: the frames (no content size, window 2^21, 128 KiB input chunks)
    must equal what its streaming API writes -- through the batch entry point and through kmp_zstd_compress_stream."""
    import ctypes
    from kompressor_amd import _lib, ZstdDecompressor
    from kompressor_amd.batch import ZstdBatch
    rows = helpers.levels_golden()["stream"]
    cases = helpers.stream_cases()
    b = ZstdBatch(max_slices=8, max_slice_bytes=2 << 20)
    try:
        for (d, cuts), (size, fed, flen, sha) in zip(cases, rows):
            empty = cuts[-1] == cuts[-2]
            src = torch.from_numpy(np.frombuffer(d + bytes(64), dtype=np.uint8).copy()).cuda()
            dst, ooff, olen = b.compress(src, torch.zeros(1, dtype=torch.int64, device="cuda"),
                                         torch.tensor([len(d)], dtype=torch.int32, device="cuda"), streaming="empty" if empty else "data")
            torch.cuda.synchronize()
            f = dst[: int(olen[0])].cpu().numpy().tobytes()
            assert len(f) == flen and helpers.sha256(f) == sha, (size, cuts)
    finally:
        b.close()
    # the C ABI's streaming entry point with the reference's call pattern and 8 KiB output slices
    lib = _lib.load()
    for (d, cuts), (size, fed, flen, sha) in list(zip(cases, rows))[:12]:
        cctx = lib.kmp_zstd_create_cctx()
        out = bytearray()
        obuf = ctypes.create_string_buffer(8192)
        pieces = list(zip(cuts[:-1], cuts[1:]))
        for j, (a0, a1) in enumerate(pieces):
            end = j == len(pieces) - 1
            sp = ctypes.c_size_t(a0)
            while True:
                dp = ctypes.c_size_t(0)
                r = lib.kmp_zstd_compress_stream(cctx, obuf, 8192, ctypes.byref(dp), d, a1, ctypes.byref(sp), 2 if end else 0)
                assert not lib.kmp_zstd_is_error(r), lib.kmp_zstd_get_error_name(r)
                out += obuf.raw[:dp.value]
                if (end and r == 0) or (not end and sp.value == a1):
                    break
        lib.kmp_zstd_free_cctx(cctx)
        assert len(out) == flen and helpers.sha256(bytes(out)) == sha, (size, cuts)
        assert ZstdDecompressor().transform_bytes(bytes(out)) == d


def test_level1_multiblock_and_streaming_frames():
    """Level 1 above 128 KiB (up to its 512 KiB window): frames of several blocks one-shot, and the streaming frames the
    reference's Ktor ZstdContentEncoder (level 1, ZstdContentEncoder.kt:11) produces -- batch entry points and the
    kmp_zstd_compress_stream call pattern, against libzstd 1.5.7, decoded back on the GPU."""
    import ctypes
    from kompressor_amd import _lib, ZstdDecompressor
    from kompressor_amd.batch import ZstdBatch
    G = helpers.levels_golden()
    ins = dict(helpers.multiblock_inputs())
    rows = G["l1_multiblock"]
    datas = [ins[r[0]] for r in rows]
    lens = np.array([len(d) for d in datas], dtype=np.int32)
    offs = np.concatenate([[0], np.cumsum((lens[:-1] + 63) & ~63, dtype=np.int64)]).astype(np.int64)
    host = np.zeros(int(offs[-1]) + int(lens[-1]) + 64, dtype=np.uint8)
    for o_, d in zip(offs, datas):
        host[int(o_):int(o_) + len(d)] = np.frombuffer(d, dtype=np.uint8)
    b = ZstdBatch(max_slices=len(datas), max_slice_bytes=512 << 10)
    try:
        dst, ooff, olen = b.compress(torch.from_numpy(host).cuda(), torch.from_numpy(offs).cuda(), torch.from_numpy(lens).cuda(), level=1)
        torch.cuda.synchronize()
        hd, ho, hl = dst.cpu().numpy(), ooff.cpu().numpy(), olen.cpu().numpy()
        frames = [hd[ho[i]:ho[i] + hl[i]].tobytes() for i in range(len(datas))]
        for (name, n, flen, sha), f in zip(rows, frames):
            assert len(f) == flen and helpers.sha256(f) == sha, name
        back, st = gpu_decompress(b, frames, [len(d) for d in datas])
        assert st == [0] * len(frames) and back == datas
        cases = [(d, cuts) for d, cuts in helpers.stream_cases() if len(d) <= 512 * 1024]
        for (d, cuts), (size, fed, flen, sha) in zip(cases, G["l1_stream"]):
            empty = cuts[-1] == cuts[-2]
            src = torch.from_numpy(np.frombuffer(d + bytes(64), dtype=np.uint8).copy()).cuda()
            dst, ooff, olen = b.compress(src, torch.zeros(1, dtype=torch.int64, device="cuda"),
                                         torch.tensor([len(d)], dtype=torch.int32, device="cuda"), level=1, streaming="empty" if empty else "data")
            torch.cuda.synchronize()
            f = dst[: int(olen[0])].cpu().numpy().tobytes()
            assert len(f) == flen and helpers.sha256(f) == sha, (size, cuts)
    finally:
        b.close()
    # a context made for larger slices serves the same slices with the same frames (round 4: beyond the window it slides, it no longer refuses)
    b2 = ZstdBatch(max_slices=2, max_slice_bytes=2 << 20)
    try:
        o = helpers.oracle()
        for lvl in (1, 2):
            dst2, _, olen2 = b2.compress(src, torch.zeros(1, dtype=torch.int64, device="cuda"), torch.tensor([len(d)], dtype=torch.int32, device="cuda"), level=lvl, check=True)
            torch.cuda.synchronize()
            assert dst2[: int(olen2[0])].cpu().numpy().tobytes() == o.compress_level_big(d, lvl), lvl
    finally:
        b2.close()
    lib = _lib.load()
    for (d, cuts), (size, fed, flen, sha) in list(zip(cases, G["l1_stream"]))[:10]:
        cctx = lib.kmp_zstd_create_cctx()
        assert lib.kmp_zstd_cctx_set_parameter(cctx, 100, 1) == 0
        out = bytearray()
        obuf = ctypes.create_string_buffer(8192)
        pieces = list(zip(cuts[:-1], cuts[1:]))
        for j, (a0, a1) in enumerate(pieces):
            end = j == len(pieces) - 1
            sp = ctypes.c_size_t(a0)
            while True:
                dp = ctypes.c_size_t(0)
                r = lib.kmp_zstd_compress_stream(cctx, obuf, 8192, ctypes.byref(dp), d, a1, ctypes.byref(sp), 2 if end else 0)
                assert not lib.kmp_zstd_is_error(r), lib.kmp_zstd_get_error_name(r)
                out += obuf.raw[:dp.value]
                if (end and r == 0) or (not end and sp.value == a1):
                    break
        lib.kmp_zstd_free_cctx(cctx)
        assert len(out) == flen and helpers.sha256(bytes(out)) == sha, (size, cuts)
        assert ZstdDecompressor().transform_bytes(bytes(out)) == d
    # one-shot level 1 above 128 KiB through the streaming ABI (finish = true from the first call)
    from kompressor_amd import ZstdCompressor
    name, n, flen, sha = rows[3]
    f = ZstdCompressor(compression_level=1).transform_bytes(ins[name])
    assert len(f) == flen and helpers.sha256(f) == sha


def test_level2_multiblock_and_streaming_frames():
    """Level 2 above 128 KiB (up to its 1 MiB window).  Its row for 128 KiB < size <= 256 KiB is a double-fast one, the others
    are fast ones: one batch goes through both block-chain kernels.  ZSTD_compress2's frames, the frames the reference's
    one-shot driver gets (ZstdCompressor(2).transform(bytes)), streamed frames and the kmp_zstd_compress_stream call pattern,
    against libzstd 1.5.7 (tests/golden/zstd_level2_big_golden.json), decoded back on the GPU."""
    from kompressor_amd import ZstdCompressor
    from kompressor_amd.batch import ZstdBatch
    G = helpers.level2_big_golden()
    ins = dict(helpers.multiblock_inputs())
    rows = G["multiblock"]
    assert any(131072 < r[1] <= 262144 for r in rows) and any(r[1] > 262144 for r in rows)
    datas = [ins[r[0]] for r in rows]
    b = ZstdBatch(max_slices=len(datas), max_slice_bytes=1 << 20)
    try:
        f0 = gpu_compress_kw(b, datas, level=2)
        f3 = gpu_compress_kw(b, datas, level=2, reference=True)
        for (name, n, l0, s0, l3, s3), x, y in zip(rows, f0, f3):
            assert (len(x), helpers.sha256(x)) == (l0, s0) and (len(y), helpers.sha256(y)) == (l3, s3), name
        back, st = gpu_decompress(b, f3, [len(d) for d in datas])
        assert st == [0] * len(f3) and back == datas
        cases = [(d, cuts) for d, cuts in helpers.stream_cases() if len(d) <= 1024 * 1024]
        for (d, cuts), (size, fed, flen, sha) in zip(cases, G["stream"]):
            f = gpu_compress_kw(b, [d], level=2, streaming="empty" if cuts[-1] == cuts[-2] else "data")[0]
            assert len(f) == flen and helpers.sha256(f) == sha, (size, cuts)
    finally:
        b.close()
    # through the streaming ABI: one-shot (finish = true from the first call), a small and a large one of each row kind
    for name, n, l0, s0, l3, s3 in (rows[0], next(r for r in rows if 131072 < r[1] <= 262144), next(r for r in rows if r[1] > 300000)):
        f = ZstdCompressor(compression_level=2).transform_bytes(ins[name])
        assert (len(f), helpers.sha256(f)) == (l3, s3), name


def test_concatenated_and_skippable_frames(batch):
    """Several frames (and skippable frames) in one entry decode to the concatenated contents, as ZSTD_decompress and the
    reference's ZstdDecompressor (ZSTD_decompressStream, Wrapper.cpp:130-147) have it; garbage after a frame is an error."""
    import struct
    from kompressor_amd import ZstdCompressor, ZstdDecompressor
    a = corpus.make(31, 1, 5000, mix=ord("T")).tobytes()
    b = corpus.make(32, 1, 70000, mix=ord("X")).tobytes()
    c = b"hello compression world"
    comp = ZstdCompressor(3)
    fa, fb, fc = comp.transform_bytes(a), comp.transform_bytes(b), comp.transform_bytes(c)
    skip = struct.pack("<II", 0x184D2A53, 7) + b"ignored"
    cases = [(fa + fb, a + b, 0), (fa + skip + fc + fb, a + c + b, 0), (skip + fc, c, 0), (b"", b"", 0),
             (fa + b"garbage!!", None, 72), (b"garbage!!", None, 10), (fa + fb[:-5], None, 72)]
    outs, st = gpu_decompress(batch, [f for f, _, _ in cases], [len(a) + len(b) + len(c) + 8] * len(cases))
    for (f, want, code), out, s in zip(cases, outs, st):
        assert s == code, (len(f), s, code)
        if want is not None:
            assert out == want
    # the streaming entry point takes them frame by frame, like ZSTD_decompressStream
    assert ZstdDecompressor().transform_bytes(fa + skip + fc + fb) == a + c + b


@pytest.mark.timeout(300)
def test_mutated_frames_on_the_gpu(batch):
    """768 damaged frames (tests/fuzz_decoders.py's mutations: bit flips, truncation, splices, header bytes ...) decoded in
    one batch: every entry ends with a status, never with a hang or a fault; what the decoder accepts is what libzstd
    1.5.7 decodes, and what it rejects libzstd rejects too (or turns into other bytes than the original: its fast Huffman
    loop does not check that the literal streams end where they should, this decoder does)."""
    import random
    import fuzz_decoders as F
    z = helpers.require_live_libzstd()
    rng = random.Random(2025)
    srcs = F.sources(rng, 16)
    frames = [(z.compress(d, lvl), d) for d in srcs for lvl in ((1, 3, 19) if len(d) < 100000 else (3,))]
    only = [f for f, _ in frames]
    cases = []
    for it in range(768):
        f0, d = rng.choice(frames)
        f, what = (f0, "intact") if it % 32 == 0 else F.mutate(rng, f0, only)
        cap = len(d) + rng.choice((0, 0, 1, 64)) if rng.random() < 0.85 else rng.randrange(0, len(d) + 1)
        cases.append((f, d, max(cap, 1), what))
    outs, st = gpu_decompress(batch, [c[0] for c in cases], [c[2] for c in cases])
    accepted = 0
    for (f, d, cap, what), out, s in zip(cases, outs, st):
        try:
            ref = z.decompress(f, cap)
        except RuntimeError:
            ref = None
        if s == 0:
            accepted += 1
            assert ref is not None and out == ref, (what, len(f), cap)
        else:
            assert ref is None or s == 14 or (s == 20 and ref != d), (what, len(f), cap, s)
    assert accepted >= 24          # the intact ones at least


def test_fuzz_ragged_sizes_levels_1_and_2(batch):
    """1 024 slices of arbitrary sizes 0 .. 128 KiB per level (every class, byte runs, short periods, class changes inside a
    slice) at levels 1 and 2 against the oracle's frames (pinned to libzstd 1.5.7 by the golden vectors), and 48 slices of
    128 KiB+1 .. 512 KiB at level 1 through the block-chain path."""
    import random
    from kompressor_amd.batch import ZstdBatch
    rng = random.Random(31337)
    o = helpers.oracle()

    def piece(n):
        r = rng.random()
        if r < 0.08:
            return bytes([rng.randrange(256)]) * n
        if r < 0.16:
            unit = corpus.make(rng.randrange(1 << 30), 1, rng.choice([2, 3, 7, 40, 300]), mix=ord("R")).tobytes()
            return (unit * (n // len(unit) + 1))[:n]
        return corpus.make(rng.randrange(1 << 30), 1, n, mix=ord(rng.choice("TXSBDIZR"))).tobytes()

    def blob(total):
        parts, have = [], 0
        while have < total:
            n = min(total - have, rng.choice([1, 5, 64, 500, 4000, 20000, 70000, 200000]))
            parts.append(piece(n))
            have += n
        return b"".join(parts)

    def run(b, datas, level):
        n = len(datas)
        lens = np.array([len(d) for d in datas], dtype=np.int32)
        offs = np.concatenate([[0], np.cumsum(((lens[:-1] + 63) & ~63).astype(np.int64))]).astype(np.int64)
        host = np.zeros(int(offs[-1]) + int(lens[-1]) + 64, dtype=np.uint8)
        for o_, d in zip(offs, datas):
            host[int(o_):int(o_) + len(d)] = np.frombuffer(d, dtype=np.uint8)
        dst, ooff, olen = b.compress(torch.from_numpy(host).cuda(), torch.from_numpy(offs).cuda(), torch.from_numpy(lens).cuda(), level=level)
        torch.cuda.synchronize()
        hd, ho, hl = dst.cpu().numpy(), ooff.cpu().numpy(), olen.cpu().numpy()
        return [hd[ho[i]:ho[i] + hl[i]].tobytes() for i in range(n)]

    sizes = [rng.choice([rng.randrange(0, 64), rng.randrange(0, 2000), rng.randrange(0, 20000), rng.randrange(0, 131073)]) for _ in range(1024)]
    datas = [blob(sz) for sz in sizes]
    for lvl in (1, 2):
        for i, (d, f) in enumerate(zip(datas, run(batch, datas, lvl))):
            assert f == o.compress_level(d, lvl), (lvl, i, len(d))
    big = [blob(rng.randrange(131073, 512 * 1024 + 1)) for _ in range(48)]
    b = ZstdBatch(max_slices=len(big), max_slice_bytes=512 << 10)
    try:
        for i, (d, f) in enumerate(zip(big, run(b, big, 1))):
            assert f == o.compress_level_big(d, 1), (i, len(d))
    finally:
        b.close()


def test_epoch_wrap_on_the_gpu():
    """A context for 64 slices has 64 teams; 560 batches of 64 small slices take every team past its 9-bit epoch (tables
    cleared on the device, epoch restarts): the frames of the batches around the wrap and of the last one are the oracle's."""
    from kompressor_amd.batch import ZstdBatch
    o = helpers.oracle()
    b = ZstdBatch(max_slices=64, max_slice_bytes=65536)
    try:
        for r in range(560):
            datas = [corpus.make(20000 + 64 * (r % 5) + i, 1, 150 + ((r + i) % 9) * 31, mix=ord("TXSB"[i % 4])).tobytes() for i in range(64)]
            frames = gpu_compress(b, datas)
            if r in (0, 509, 510, 511, 512, 513, 559):
                for d, f in zip(datas, frames):
                    assert f == o.compress(d), r
    finally:
        b.close()


def test_slices_longer_than_the_context_are_refused_not_overrun():
    """A slice above the context's max_slice_bytes gets out_len 0 and raises the status word on every compress path
    (level 3, levels 1 / 2, dictionary, frames of several blocks); the other slices of the batch are untouched."""
    from kompressor_amd.batch import ZstdBatch
    o = helpers.oracle()
    b = ZstdBatch(max_slices=8, max_slice_bytes=16384)
    try:
        datas = [corpus.make(50 + i, 1, sz).tobytes() for i, sz in enumerate([16384, 16385, 100, 65536, 0])]

        def run(**kw):
            n = len(datas)
            lens = np.array([len(d) for d in datas], dtype=np.int32)
            offs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.int64)]).astype(np.int64)
            host = np.frombuffer(b"".join(datas) + bytes(64), dtype=np.uint8).copy()
            # strides sized for the largest slice so that a refused slice cannot be told from an overrun by accident
            stride = 70000
            dst = torch.zeros(n * stride + 64, dtype=torch.uint8, device="cuda")
            ooff = torch.arange(n, dtype=torch.int64, device="cuda") * stride
            _, _, olen = b.compress(torch.from_numpy(host).cuda(), torch.from_numpy(offs).cuda(), torch.from_numpy(lens).cuda(), dst=dst, out_off=ooff, **kw)
            torch.cuda.synchronize()
            d, ol = dst.cpu().numpy(), olen.cpu().numpy()
            return [d[i * stride:i * stride + ol[i]].tobytes() for i in range(n)]

        for kw, ref in (({}, o.compress), ({"level": 1}, lambda d: o.compress_level(d, 1)),
                        ({"dictionary": corpus.make(9, 1, 4096).tobytes()}, None)):
            frames = run(**kw)
            assert b.status() == (-3, 1), kw
            for d, f in zip(datas, frames):
                if len(d) > 16384:
                    assert f == b"", (kw, len(d))
                elif ref is not None:
                    assert f == ref(d), (kw, len(d))
                else:
                    assert len(f) > 0
        assert b.status() == (0, 0)
    finally:
        b.close()
    b = ZstdBatch(max_slices=4, max_slice_bytes=300000)
    try:
        datas = [corpus.make(60 + i, 1, sz).tobytes() for i, sz in enumerate([300000, 300001, 140000])]
        frames = gpu_compress(b, datas)
        assert b.status() == (-3, 1)
        assert frames[1] == b"" and frames[0] == o.compress(datas[0]) and frames[2] == o.compress(datas[2])
    finally:
        b.close()


def test_host_batch_and_the_coalescer_of_the_streaming_entry_point(G):
    """The batch a JVM can reach.  kmp_zstd_compress_host_batch / kmp_zstd_decompress_host_batch (what jni/zstd/BatchWrapper.cpp
    binds): host slices in, frames out, equal to the oracle's; more slices than the staging holds go through in pieces.  And
    kmp_zstd_compress_stream's coalescer: 48 contexts closing their slices at once from 48 threads get, each, the frame it would
    have got alone (AsyncSliceTransform.kt:56-65 runs transforms like this)."""
    import threading
    from kompressor_amd import ZstdCompressor, ZstdDecompressor
    from kompressor_amd.batch import compress_host_batch, decompress_host_batch
    o = helpers.oracle()
    rng = np.random.default_rng(3)
    datas = []
    for i in range(1100):                                   # > KMP_HOST_BATCH_SLICES (1024): two pieces
        n = int(rng.integers(0, 131073)) if i % 7 else (0, 1, 131072, 65536, 7, 8, 9)[(i // 7) % 7]
        datas.append(corpus.make(70000 + i, 1, max(n, 1), mix=ord("TXSBDIZR"[i % 8])).tobytes()[:n])
    frames = compress_host_batch(datas)
    for k in range(0, len(datas), 37):
        assert frames[k] == o.compress(datas[k]), (k, len(datas[k]))
    for lvl in (1, 2):
        fl = compress_host_batch(datas[:40], level=lvl)
        for d, f in zip(datas[:40], fl):
            assert f == o.compress_level(d, lvl), (lvl, len(d))
    outs, st = decompress_host_batch(frames, [max(len(d), 1) for d in datas])
    assert st == [0] * len(datas) and outs == datas
    # a large batch: more than twice the small engine's 1 024 slices -> the bulk engines, pieces on worker threads
    many = [datas[i % len(datas)] for i in range(5000)]
    fr2 = compress_host_batch(many)
    assert all(fr2[i] == frames[i % len(datas)] for i in range(0, 5000, 13))
    outs2, st2 = decompress_host_batch(fr2, [max(len(d), 1) for d in many])
    assert st2 == [0] * len(many) and all(outs2[i] == many[i] for i in range(0, 5000, 7))
    outs, st = decompress_host_batch([frames[3][:-2], b"junk" * 8, frames[5]], [131072, 131072, 4])
    assert st[0] != 0 and st[1] != 0 and (st[2] == 70 or len(datas[5]) <= 4)
    # the coalescer
    picks = [datas[i] for i in range(0, 48 * 3, 3)]
    results = [None] * len(picks)

    def work(t):
        c = ZstdCompressor(3)
        results[t] = (c.transform_bytes(picks[t]), c.transform_bytes(picks[(t + 1) % len(picks)]))      # (a context is reused)

    th = [threading.Thread(target=work, args=(t,)) for t in range(len(picks))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for t, (a, b2) in enumerate(results):
        assert a == o.compress(picks[t]) and b2 == o.compress(picks[(t + 1) % len(picks)]), t
    assert ZstdDecompressor().transform_bytes(results[0][0]) == picks[0]
    # ... and the decoder's: 48 ZstdDecompressor contexts at once, one of them fed a damaged frame
    back = [None] * len(picks)

    def unwork(t):
        d = ZstdDecompressor()
        if t == 7:
            bad = bytearray(results[t][0]); bad[len(bad) // 2] ^= 0x55
            try:
                back[t] = ("ok", d.transform_bytes(bytes(bad)))
            except RuntimeError as e_:
                back[t] = ("err", str(e_))
        else:
            back[t] = ("ok", d.transform_bytes(results[t][0]) + d.transform_bytes(results[t][1]))

    th = [threading.Thread(target=unwork, args=(t,)) for t in range(len(picks))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for t, (kind, val) in enumerate(back):
        if t == 7:
            assert kind == "err" or val != picks[t]
        else:
            assert kind == "ok" and val == picks[t] + picks[(t + 1) % len(picks)], t


def test_python_wrapper_check_raises_on_a_refused_slice():
    """ZstdBatch.compress(check=True) reads the status word: a slice longer than the context holds is an exception, not an
    empty frame that compact() would drop silently."""
    from kompressor_amd.batch import ZstdBatch
    b = ZstdBatch(max_slices=4, max_slice_bytes=16384)
    try:
        host = corpus.make(5, 1, 40000)
        src = torch.from_numpy(host).cuda()
        off = torch.tensor([0, 16384], dtype=torch.int64, device="cuda")
        ln = torch.tensor([16384, 20000], dtype=torch.int32, device="cuda")
        with pytest.raises(RuntimeError, match="status bits 0x1"):
            b.compress(src, off, ln, check=True)
        ln[1] = 16000
        _, _, olen = b.compress(src, off, ln, check=True)
        assert int(olen.min()) > 0
    finally:
        b.close()


def test_streaming_decoder_refuses_frames_beyond_its_staging():
    """kmp_zstd_decompress_stream stages frames of up to 1 GiB of content: a larger declared content size is refused when
    the header arrives; a frame without content size is decoded into a staging buffer that grows until it fits."""
    from kompressor_amd.zstd import ZstdDecompressor
    # 2 GiB declared in the header (fcs 4 bytes, single segment), nothing else needed to refuse it
    hdr = bytes([0x28, 0xB5, 0x2F, 0xFD, 0xA0]) + (2 << 30).to_bytes(4, "little") + bytes([0x01, 0x00, 0x00])
    with pytest.raises(RuntimeError, match="Unsupported frame parameter"):
        ZstdDecompressor().transform_bytes(hdr)
    # no content size: 20 raw blocks of 128 KiB after a window descriptor (2.5 MiB: the first staging guess is 10 MiB;
    # 100 RLE blocks: 12.5 MiB from a 400-byte frame: the staging grows twice)
    body = b"".join((((131072 << 3) | (1 if k == 19 else 0)).to_bytes(3, "little") + bytes([k]) * 131072) for k in range(20))
    out = ZstdDecompressor().transform_bytes(bytes([0x28, 0xB5, 0x2F, 0xFD, 0x00, 0x58]) + body)
    assert out == b"".join(bytes([k]) * 131072 for k in range(20))
    rle = b"".join((((131072 << 3) | 2 | (1 if k == 99 else 0)).to_bytes(3, "little") + bytes([k])) for k in range(100))
    out = ZstdDecompressor().transform_bytes(bytes([0x28, 0xB5, 0x2F, 0xFD, 0x00, 0x58]) + rle)
    assert out == b"".join(bytes([k]) * 131072 for k in range(100))


@pytest.mark.timeout(900)
def test_reference_driver_frames_above_128k_and_beyond_the_window():
    """What ZstdCompressor(3).transform(bytes) REALLY returns above 128 KiB: the reference's output slices
    (max(8192, n / 10) bytes, SliceTransform.kt:47-56) are smaller than ZSTD_compressBound, so libzstd stages the input in
    128 KiB chunks, and beyond 2 MiB + 128 KiB its staging buffer wraps (older lap = extDict segment, sliding window).
    57 inputs of 128 KiB + 1 .. 6.7 MiB against frames a binary libzstd 1.5.7 wrote under exactly those calls
    (tests/golden/make_golden_buffered.py): the one-shot driver and streamed, batch entry points; then the streaming C ABI
    with the reference's own driver loop, the in-place tail of a stream, and the decoder on all of it."""
    import ctypes
    from kompressor_amd import _lib, ZstdCompressor, ZstdDecompressor
    from kompressor_amd.batch import ZstdBatch
    B = helpers.buffered_golden()
    inputs = helpers.multiblock_inputs() + helpers.beyond_window_inputs()
    rows = {r["name"]: r for r in B["rows"]}
    small = [(nm, d) for nm, d in inputs if len(d) <= (2 << 20)]
    # contexts for <= 2 MiB use table entries with check bits, the 8 MiB one plain 32-bit indices
    for group, cap in ((small, 2 << 20), (inputs, 8 << 20)):
        datas = [d for _, d in group]
        b = ZstdBatch(max_slices=len(datas), max_slice_bytes=cap)
        try:
            for key, kw in (("oneshot", {"reference": True}), ("stream", {"streaming": "data"})):
                n = len(datas)
                lens = np.array([len(d) for d in datas], dtype=np.int32)
                offs = np.concatenate([[0], np.cumsum(lens[:-1].astype(np.int64))]).astype(np.int64)
                host = np.frombuffer(b"".join(datas) + bytes(64), dtype=np.uint8).copy()
                dst, ooff, olen = b.compress(torch.from_numpy(host).cuda(), torch.from_numpy(offs).cuda(), torch.from_numpy(lens).cuda(), **kw)
                torch.cuda.synchronize()
                assert b.status() == (0, 0)
                dd, oo, ol = dst.cpu().numpy(), ooff.cpu().numpy(), olen.cpu().numpy()
                frames = [dd[oo[i]:oo[i] + ol[i]].tobytes() for i in range(n)]
                for (nm, d), f in zip(group, frames):
                    r = rows[nm]
                    assert len(f) == r[key + "_len"] and helpers.sha256(f) == r[key + "_sha256"], (cap, key, nm)
                if key == "oneshot":
                    back, st = gpu_decompress(b, frames, [len(d) for d in datas])
                    assert st == [0] * n and back == datas
            # ZSTD_compress2's frames (the caller's array compressed in place) beyond the window
            for (nm, d), f in zip(group, gpu_compress(b, datas)):
                if "compress2_len" in rows[nm]:
                    assert len(f) == rows[nm]["compress2_len"] and helpers.sha256(f) == rows[nm]["compress2_sha256"], (cap, "compress2", nm)
        finally:
            b.close()
    # level 1 (the Ktor encoder's level) through the same driver, up to its 512 KiB window
    l1 = [(nm, d) for nm, d in inputs if len(d) <= (512 << 10)]
    b = ZstdBatch(max_slices=len(l1), max_slice_bytes=512 << 10)
    try:
        for (nm, d), f in zip(l1, gpu_compress_kw(b, [d for _, d in l1], level=1, reference=True)):
            assert len(f) == rows[nm]["l1_oneshot_len"] and helpers.sha256(f) == rows[nm]["l1_oneshot_sha256"], ("level 1", nm)
    finally:
        b.close()
    # the streaming C ABI driven by the reference's one-shot loop: output slices of max(8192, n / 10) bytes
    assert ZstdCompressor(1).transform_bytes(l1[-1][1]) == gpu_compress_kw_last[0]
    for nm in ("mixed_1_300000", "class_B_1m", "beyond_2228225", "beyond_3000000"):
        d = dict(inputs)[nm]
        f = ZstdCompressor(3).transform_bytes(d)
        assert len(f) == rows[nm]["oneshot_len"] and helpers.sha256(f) == rows[nm]["oneshot_sha256"], nm
        assert ZstdDecompressor().transform_bytes(f) == d
    # ... and with room for ZSTD_compressBound in the first output slice libzstd compresses in place: ZSTD_compress2's frame
    lib = _lib.load()
    d = dict(inputs)["class_B_1m"]
    cctx = lib.kmp_zstd_create_cctx()
    cap = lib.kmp_zstd_compress_bound(len(d))
    obuf = ctypes.create_string_buffer(cap)
    dp, sp = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert lib.kmp_zstd_compress_stream(cctx, obuf, cap, ctypes.byref(dp), d, len(d), ctypes.byref(sp), 2) == 0
    lib.kmp_zstd_free_cctx(cctx)
    g = next(r for r in helpers.multiblock_golden()["rows"] if r["name"] == "class_B_1m")
    assert dp.value == g["len"] and helpers.sha256(obuf.raw[:dp.value]) == g["sha256"]
    # a stream whose closing call brings a few bytes right after libzstd's staging buffer wrapped: compressed in place
    base = helpers.beyond_window_inputs()[-1][1]
    lap = 17 * 131072
    for row in B["tails"]:
        d = base[: row["laps"] * lap + row["tail"]]
        cctx = lib.kmp_zstd_create_cctx()
        out = bytearray()
        obuf = ctypes.create_string_buffer(8192)
        for a0, a1, end in ((0, row["laps"] * lap, False), (row["laps"] * lap, len(d), True)):
            sp = ctypes.c_size_t(a0)
            while True:
                dp = ctypes.c_size_t(0)
                r = lib.kmp_zstd_compress_stream(cctx, obuf, 8192, ctypes.byref(dp), d, a1, ctypes.byref(sp), 2 if end else 0)
                assert not lib.kmp_zstd_is_error(r), lib.kmp_zstd_get_error_name(r)
                out += obuf.raw[:dp.value]
                if (end and r == 0) or (not end and sp.value == a1):
                    break
        lib.kmp_zstd_free_cctx(cctx)
        assert len(out) == row["len"] and helpers.sha256(bytes(out)) == row["sha256"], row["name"]


@pytest.mark.timeout(900)
def test_large_stream_like_the_reference_largeSample():
    """ZstdTest.largeSample (ZstdTest.kt:67-82) pipes 256 MiB + 3 random bytes through ZstdCompressor(3) and
    ZstdDecompressor in 8 KiB pieces.  Same shape here at 24 MiB + 3 (random bytes: every block raw, eleven laps of
    libzstd's staging buffer), finish = false pieces then finish = true, and the frame must be libzstd's."""
    import ctypes
    from kompressor_amd import _lib, ZstdDecompressor
    lib = _lib.load()
    n = (24 << 20) + 3
    d = np.random.default_rng(42).integers(0, 256, n, dtype=np.uint8).tobytes()
    cctx = lib.kmp_zstd_create_cctx()
    out = bytearray()
    obuf = ctypes.create_string_buffer(8192)
    piece = 1 << 20                              # (8 KiB pieces in the reference; the staging is the same)
    pos = 0
    while pos < n:
        a1 = min(n, pos + piece)
        sp, dp = ctypes.c_size_t(pos), ctypes.c_size_t(0)
        r = lib.kmp_zstd_compress_stream(cctx, obuf, 8192, ctypes.byref(dp), d, a1, ctypes.byref(sp), 0)
        assert not lib.kmp_zstd_is_error(r) and sp.value == a1 and dp.value == 0
        pos = a1
    while True:
        sp, dp = ctypes.c_size_t(n), ctypes.c_size_t(0)
        r = lib.kmp_zstd_compress_stream(cctx, obuf, 8192, ctypes.byref(dp), d, n, ctypes.byref(sp), 2)
        assert not lib.kmp_zstd_is_error(r), lib.kmp_zstd_get_error_name(r)
        out += obuf.raw[:dp.value]
        if r == 0:
            break
    lib.kmp_zstd_free_cctx(cctx)
    f = bytes(out)
    # random bytes: a header, raw blocks of 128 KiB, the 3-byte tail
    nblocks = (n + 131071) // 131072
    assert len(f) == 6 + n + 3 * nblocks
    z = helpers.live_libzstd()
    if z is not None:
        assert f == z.compress_streaming(d, [0, n // 2, n], out_chunk=8192)
    assert hashlib.sha256(ZstdDecompressor().transform_bytes(f)).digest() == hashlib.sha256(d).digest()


def test_frames_of_other_encoder_settings_decode(batch):
    """ZstdDecompressor takes what any zstd encoder made: 104 frames libzstd 1.5.7 produced at levels 1 .. 22, with a content
    checksum, with a window larger than the content, of long runs and of short periods (tests/golden/foreign_frames.*:
    Huffman trees of every depth, single-stream literals, RLE / predefined / repeat sequence tables, offsets the
    pre-decoder takes in two containers).  Decoded in one batch, twice: with exact capacities and with room to spare."""
    rows = helpers.foreign_frames()
    frames = [f for _, f, _ in rows]
    for slack in (0, 777):
        outs, st = gpu_decompress(batch, frames, [len(p) + slack for _, _, p in rows])
        for (r, _, plain), o, s_ in zip(rows, outs, st):
            assert s_ == 0 and o == plain, (r["level"], r.get("cls"), r["size"], r["extra"])


def test_predecode_kernels_give_the_same_frames(monkeypatch):
    """The decoder is three kernels: the two pre-decoders (sequences: four lanes per frame; literals:
    one lane per Huffman stream) stage what k_zstd_decode then executes, and whatever they do not stage k_zstd_decode decodes
    itself.  Every combination (KMP_DECODE_PRE = 0: everything in k_zstd_decode, 1: sequences ahead, 2: literals ahead,
    default 3: both) must give the same bytes and the same status words: the golden frames, foreign-level and multi-block
    frames, damaged frames and a multi-block batch."""
    from kompressor_amd.batch import ZstdBatch
    G = helpers.golden()
    S = 65536
    rows = G["config1"][:512]
    buf = corpus.make(0, len(rows), S)
    datas = [buf[i * S:(i + 1) * S].tobytes() for i in range(len(rows))]
    big = [d for _, d in helpers.multiblock_inputs()[20:30]]
    ref = ZstdBatch(max_slices=800, max_slice_bytes=2 << 20)
    others = []
    for mode in ("0", "1", "2"):
        monkeypatch.setenv("KMP_DECODE_PRE", mode)           # (a switch of the ablation build)
        others.append(ZstdBatch(max_slices=800, max_slice_bytes=2 << 20, ablations=True))
    monkeypatch.delenv("KMP_DECODE_PRE")
    monkeypatch.setenv("KMP_DECODE_STAGE_SLICES", "256")         # staging for 256 entries: the batch goes through in pieces that reuse it
    others.append(ZstdBatch(max_slices=800, max_slice_bytes=2 << 20))
    monkeypatch.delenv("KMP_DECODE_STAGE_SLICES")
    try:
        frames = gpu_compress(ref, datas) + gpu_compress_kw(ref, big, reference=True)
        frames += [base64.b64decode(r["frame"]) for r in G["decode_only"]]
        rng = np.random.default_rng(5)
        for k in range(200):                      # damaged frames: same status, same bytes when accepted
            f = bytearray(frames[k % 512])
            for _ in range(int(rng.integers(1, 4))):
                f[int(rng.integers(0, len(f)))] ^= 1 << int(rng.integers(0, 8))
            frames.append(bytes(f[: int(rng.integers(len(f) // 2, len(f) + 1))]) if k % 3 == 0 else bytes(f))
        caps = [2 << 20] * len(frames)
        a, sa = gpu_decompress(ref, frames, caps)
        assert a[: len(datas)] == datas and a[len(datas): len(datas) + len(big)] == big
        for mode, ctx in zip(("KMP_DECODE_PRE=0", "KMP_DECODE_PRE=1", "KMP_DECODE_PRE=2", "KMP_DECODE_STAGE_SLICES=256"), others):
            b, sb = gpu_decompress(ctx, frames, caps)
            assert sa == sb and a == b, mode
    finally:
        ref.close()
        for ctx in others:
            ctx.close()


@pytest.mark.timeout(600)
def test_fuzz_long_slices_all_framings_against_oracle():
    """96 slices of arbitrary sizes 128 KiB + 1 .. 5.5 MiB (every class, byte runs, copies of earlier pieces at any distance)
    in the three framings libzstd has for them -- the reference's one-shot driver, a stream, ZSTD_compress2 in place --
    each compared with the oracle's frame (which is pinned on the binary library, tests/test_oracle_golden.py)."""
    import random
    from kompressor_amd.batch import ZstdBatch
    rng = random.Random(4711)
    o = helpers.oracle()
    lap = 17 * 131072

    def build(n):
        out = bytearray()
        while len(out) < n:
            r = rng.random()
            if r < 0.3 and len(out) > 1000:
                a = rng.randrange(0, len(out))
                out += out[a:a + rng.randrange(10, 300000)]
            elif r < 0.36:
                out += bytes([rng.randrange(256)]) * rng.randrange(1, 200000)
            else:
                out += corpus.make(rng.randrange(1 << 30), 1, rng.randrange(1000, 300000), mix=ord(rng.choice("TXSBDIZR"))).tobytes()
        return bytes(out[:n])

    sizes = [rng.choice([rng.randrange(131073, 600000), rng.randrange(600000, 2 << 20), (2 << 20) + rng.randrange(-5, 140000),
                         lap + rng.randrange(-3, 300000), 131072 * rng.randrange(2, 40) + rng.randrange(-2, 3),
                         rng.randrange(2 << 20, 5500000)]) for _ in range(96)]
    datas = [build(n) for n in sizes]
    for cap, sel in ((2 << 20, [d for d in datas if len(d) <= (2 << 20)]), (6 << 20, datas)):
        b = ZstdBatch(max_slices=len(sel), max_slice_bytes=cap)
        try:
            for kw, ref in (({"reference": True}, lambda d: o.compress_buffered(d, True)),
                            ({"streaming": "data"}, lambda d: o.compress_buffered(d, False)),
                            ({}, lambda d: o.compress_buffered(d, 2))):
                frames = gpu_compress_kw(b, sel, **kw)
                assert b.status() == (0, 0)
                for d, f in zip(sel, frames):
                    assert f == ref(d), (cap, kw, len(d))
        finally:
            b.close()


def test_batch_in_pieces_side_by_side(G):
    """kmp_zstd_compress_batch_pieces: the level-3 batch cut into 1 .. 8 parts, part p queued on a stream of its own behind the copy that
    brings its slices in, the parts sharing the context (disjoint team slots and workspace).  Frames are bit for bit the one-launch
    batch's (the 2 048 golden slices of configs[1] among them), for equal and for ragged slices, and a plain batch on the same context
    afterwards waits for the parts."""
    from kompressor_amd.batch import ZstdBatch
    n, S = 8192, 65536
    host = corpus.make(0, n, S)
    pin = torch.from_numpy(host).pin_memory()
    b = ZstdBatch(max_slices=n, max_slice_bytes=131072, team_lanes=4)
    try:
        src = torch.from_numpy(host).cuda()
        in_off = torch.arange(n, dtype=torch.int64, device="cuda") * S
        in_len = torch.full((n,), S, dtype=torch.int32, device="cuda")
        ref, ooff, rlen = b.compress(src, in_off, in_len, check=True)
        torch.cuda.synchronize()
        ref = ref.clone(); rlen = rlen.clone()
        lens = rlen.cpu().numpy()
        for i, cls, flen, sha in G["config1"]:
            assert lens[i] == flen
        for P in (1, 3, 4, 8):
            streams = [torch.cuda.Stream() for _ in range(P)]
            src2 = torch.zeros(n * S, dtype=torch.uint8, device="cuda")
            dst = torch.zeros_like(ref); olen = torch.zeros_like(rlen)
            torch.cuda.synchronize()
            covered = 0
            for p in range(P):
                first, cnt = b.piece_range(n, P, p)
                assert first == covered
                covered += cnt
                with torch.cuda.stream(streams[p]):
                    src2[first * S:(first + cnt) * S].copy_(pin[first * S:(first + cnt) * S], non_blocking=True)
            assert covered == n
            b.compress_pieces(src2, in_off, in_len, dst, ooff, olen, streams)
            # a plain batch right behind the pieces, on the default stream: it must wait for them (shared workspace)
            d3, o3, l3 = b.compress(src, in_off, in_len)
            torch.cuda.synchronize()
            assert b.status() == (0, 0)
            assert torch.equal(olen, rlen) and torch.equal(l3, rlen), P
            hd, hr, ho = dst.cpu().numpy(), ref.cpu().numpy(), ooff.cpu().numpy()
            for i in range(n):
                assert hd[ho[i]:ho[i] + lens[i]].tobytes() == hr[ho[i]:ho[i] + lens[i]].tobytes(), (P, i)
            for i, cls, flen, sha in G["config1"][::16]:
                assert helpers.sha256(hd[ho[i]:ho[i] + lens[i]].tobytes()) == sha, (P, i)
        # ragged slices (0 .. 128 KiB), five pieces, against the oracle
        rng = np.random.default_rng(11)
        sizes = [int(x) for x in rng.integers(0, 131073, 700)] + [0, 1, 7, 8, 131072]
        datas = [corpus.make(5000 + k, 1, max(s, 1), mix=ord("TXSBDIZR"[k % 8])).tobytes()[:s] for k, s in enumerate(sizes)]
        m = len(datas)
        rl = np.array(sizes, dtype=np.int32); ro = np.concatenate([[0], np.cumsum(rl[:-1].astype(np.int64))]).astype(np.int64)
        rh = torch.from_numpy(np.frombuffer(b"".join(datas) + bytes(64), dtype=np.uint8).copy()).cuda()
        streams = [torch.cuda.Stream() for _ in range(5)]
        dst = torch.zeros(m * b.out_stride + 64, dtype=torch.uint8, device="cuda")
        ooff2 = torch.arange(m, dtype=torch.int64, device="cuda") * b.out_stride
        olen2 = torch.zeros(m, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        b.compress_pieces(rh, torch.from_numpy(ro).cuda(), torch.from_numpy(rl).cuda(), dst, ooff2, olen2, streams)
        torch.cuda.synchronize()
        assert b.status() == (0, 0)
        o = helpers.oracle()
        hd, hl = dst.cpu().numpy(), olen2.cpu().numpy()
        for i, d in enumerate(datas):
            assert hd[i * b.out_stride:i * b.out_stride + hl[i]].tobytes() == o.compress(d), (i, len(d))
    finally:
        b.close()


@pytest.mark.timeout(600)
def test_host_batch_pipelined_registered_memory_and_release():
    """The pipelined bulk compressor behind kmp_zstd_compress_host_batch (kmp_coalesce.h): a large level-3 batch from pageable
    memory (pinned staging, worker threads), from memory registered with kmp_host_register (the device reads the slices and writes
    the frames itself: no staging copy), from registered memory whose slices do not lie one behind the other (the staged copy in,
    frames still written directly), with output regions that are too small (those slices: out_len 0, KMP_ERR_CAPACITY, the rest
    fine) -- every frame equal to the device batch's, a sample equal to the oracle's.  Then kmp_host_engines_release: the pinned
    staging and the device memory are given back, and the next call makes them again."""
    import ctypes
    from kompressor_amd import _lib
    from kompressor_amd.batch import ZstdBatch, compress_bound, host_engines_release, host_register, host_unregister
    lib = _lib.load()
    rng = np.random.default_rng(21)
    n = 6000
    sizes = rng.integers(1, 65537, n).astype(np.uint32); sizes[::97] = 65536; sizes[5] = 0
    offs = np.concatenate([[0], np.cumsum(sizes[:-1].astype(np.uint64))]).astype(np.uint64)
    total = int(sizes.astype(np.int64).sum())
    src = np.empty(total + 64, dtype=np.uint8)
    pos = 0
    for k in range(0, n, 500):
        blk = corpus.make(31000 + k, 500, 65536)
        for j in range(500):
            i = k + j
            src[pos:pos + sizes[i]] = blk[j * 65536:j * 65536 + sizes[i]]; pos += int(sizes[i])
    caps = np.array([compress_bound(int(s)) for s in sizes], dtype=np.uint32)
    ooff = np.concatenate([[0], np.cumsum(caps[:-1].astype(np.uint64))]).astype(np.uint64)
    vp = lambda a: ctypes.c_void_p(a.ctypes.data)          # noqa: E731

    def run(s, so, sl, cap=caps, oo=ooff):
        dst = np.zeros(int(cap.astype(np.int64).sum()) + 64, dtype=np.uint8)
        olen = np.full(n, 7, dtype=np.uint32)
        rc = lib.kmp_zstd_compress_host_batch(0, 3, vp(s), vp(so), vp(sl), n, vp(dst), vp(oo), vp(cap), vp(olen))
        return rc, dst, olen

    # the reference: the device batch over the same slices
    b = ZstdBatch(max_slices=n, max_slice_bytes=65536)
    d_dst, d_ooff, d_olen = b.compress(torch.from_numpy(src).cuda(), torch.from_numpy(offs.astype(np.int64)).cuda(), torch.from_numpy(sizes.astype(np.int32)).cuda(), check=True)
    torch.cuda.synchronize()
    hd, ho, hl = d_dst.cpu().numpy(), d_ooff.cpu().numpy(), d_olen.cpu().numpy()
    want = [hd[ho[i]:ho[i] + hl[i]].tobytes() for i in range(n)]
    b.close()
    o = helpers.oracle()
    for i in range(0, n, 211):
        assert want[i] == o.compress(src[int(offs[i]):int(offs[i]) + int(sizes[i])].tobytes()), i

    def same(dst, olen, oo=ooff):
        assert np.array_equal(olen, hl.astype(np.uint32))
        for i in range(n):
            assert dst[int(oo[i]):int(oo[i]) + int(olen[i])].tobytes() == want[i], i

    host_engines_release()                                     # (whatever earlier tests of this process left)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    rc, dst, olen = run(src, offs, sizes)                      # pageable in, pageable out
    assert rc == 0, lib.kmp_last_error()
    same(dst, olen)
    held = free0 - torch.cuda.mem_get_info()[0]
    assert held > (1 << 30)                                    # the bulk compressor is there now
    # registered memory both ways
    dst2 = np.zeros(int(caps.astype(np.int64).sum()) + 64, dtype=np.uint8)
    host_register(src); host_register(dst2)
    try:
        olen2 = np.full(n, 7, dtype=np.uint32)
        rc = lib.kmp_zstd_compress_host_batch(0, 3, vp(src), vp(offs), vp(sizes), n, vp(dst2), vp(ooff), vp(caps), vp(olen2))
        assert rc == 0, lib.kmp_last_error()
        same(dst2, olen2)
        # slices that do not lie one behind the other (every second pair swapped): the copy in falls back to staging
        perm = np.arange(n); perm[0:n - 1:2], perm[1:n:2] = np.arange(1, n, 2), np.arange(0, n - 1, 2)
        dst2[:] = 0; olen3 = np.zeros(n, dtype=np.uint32)
        po, pl = np.ascontiguousarray(offs[perm]), np.ascontiguousarray(sizes[perm])
        pc = np.ascontiguousarray(caps[perm]); poo = np.concatenate([[0], np.cumsum(pc[:-1].astype(np.uint64))]).astype(np.uint64)
        rc = lib.kmp_zstd_compress_host_batch(0, 3, vp(src), vp(po), vp(pl), n, vp(dst2), vp(poo), vp(pc), vp(olen3))
        assert rc == 0, lib.kmp_last_error()
        for i in range(n):
            assert dst2[int(poo[i]):int(poo[i]) + int(olen3[i])].tobytes() == want[int(perm[i])], i
        # output regions that are too small, registered (the scatter kernel refuses them) and pageable (the hand-out does)
        small = caps.copy(); small[10] = 8; small[4001] = max(int(hl[4001]) - 1, 1)
        for registered in (True, False):
            d4 = dst2 if registered else np.zeros_like(dst2)
            d4[:] = 0; olen4 = np.full(n, 7, dtype=np.uint32)
            rc = lib.kmp_zstd_compress_host_batch(0, 3, vp(src), vp(offs), vp(sizes), n, vp(d4), vp(ooff), vp(small), vp(olen4))
            assert rc == -3, (registered, rc)
            assert olen4[10] == 0 and olen4[4001] == 0
            ok = np.ones(n, dtype=bool); ok[[10, 4001]] = False
            assert np.array_equal(olen4[ok], hl.astype(np.uint32)[ok])
            for i in (9, 11, 4000, 4002, n - 1):
                assert d4[int(ooff[i]):int(ooff[i]) + int(olen4[i])].tobytes() == want[i], (registered, i)
            assert not d4[int(ooff[10]) + 8:int(ooff[11])].any()           # nothing written past a small region
    finally:
        host_unregister(src); host_unregister(dst2)
    # release: what the calls held comes back; the next call makes it again
    host_engines_release(0)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] >= free0 - (64 << 20)
    rc, dst, olen = run(src, offs, sizes)
    assert rc == 0, lib.kmp_last_error()
    same(dst, olen)
    host_engines_release()


def _stream_through_abi(lib, d, level, cuts, out_chunk=8192):
    """kmp_zstd_compress_stream the way the reference's streaming callers drive ZSTD_compressStream2 (SliceTransformRawSource.kt:32-55):
    d[cuts[i]:cuts[i+1]] fed with finish = false, the last piece with finish = true; output drained through out_chunk-byte slices."""
    import ctypes
    cctx = lib.kmp_zstd_create_cctx()
    assert lib.kmp_zstd_cctx_set_parameter(cctx, 100, level) == 0
    out = bytearray(); obuf = ctypes.create_string_buffer(out_chunk)
    pieces = list(zip(cuts[:-1], cuts[1:]))
    try:
        for j, (a0, a1) in enumerate(pieces):
            end = j == len(pieces) - 1
            sp = ctypes.c_size_t(a0)
            while True:
                dp = ctypes.c_size_t(0)
                r = lib.kmp_zstd_compress_stream(cctx, obuf, out_chunk, ctypes.byref(dp), d, a1, ctypes.byref(sp), 2 if end else 0)
                assert not lib.kmp_zstd_is_error(r), lib.kmp_zstd_get_error_name(r)
                out += obuf.raw[:dp.value]
                if (end and r == 0) or (not end and sp.value == a1 and dp.value < out_chunk):
                    break
    finally:
        lib.kmp_zstd_free_cctx(cctx)
    return bytes(out)


@pytest.mark.timeout(900)
def test_fast_levels_beyond_their_window():
    """Levels 1, 2 and the negative ones on slices and streams LONGER than the level's window (512 KiB / 1 MiB): libzstd's staging
    buffer wraps, the blocks behind the wrap go through ZSTD_compressBlock_fast_extDict (zstd_match_fast_ext_body) until the window
    has slid past the older segment.  The reference's Ktor encoder streams at level 1 (ZstdContentEncoder.kt:11 through
    BaseSliceTransformContentEncoder.kt:23-54), so every response above 640 KiB is such a frame.  All 24 committed inputs in the four
    framings against libzstd 1.5.7 (tests/golden/zstd_fast_window_golden.json) through the batch calls, a wide-index context, the
    streaming entry point on a 2 MiB and a 24 MiB level-1 stream fed in pieces (against the oracle and, where present, the live
    library), and everything decoded back on the GPU."""
    from kompressor_amd import _lib, ZstdCompressor, ZstdDecompressor
    from kompressor_amd.batch import ZstdBatch
    G = helpers.fast_window_golden()["rows"]
    ins = helpers.fast_window_inputs()
    o = helpers.oracle()
    for level in (1, -1, -5, 2):
        rows = [(r, d) for (nm, lv, d), r in zip(ins, G) if lv == level]
        datas = [d for _, d in rows]
        b = ZstdBatch(max_slices=len(datas), max_slice_bytes=max(len(d) for d in datas) + 1)
        try:
            for kw, key in ((dict(reference=True), "oneshot"), (dict(streaming="data"), "stream"), (dict(streaming="empty"), "stream_empty_end"), (dict(), "compress2")):
                frames = gpu_compress_kw(b, datas, level=level, check=True, **kw)
                for (r, d), f in zip(rows, frames):
                    assert (len(f), helpers.sha256(f)) == (r[key + "_len"], r[key + "_sha256"]), (r["name"], key)
            back, st = gpu_decompress(b, frames, [len(d) for d in datas])
            assert st == [0] * len(frames) and back == datas
        finally:
            b.close()
    # a context for slices of 4 MiB and more keeps plain 32-bit indices in its tables (no check bits): the same frames
    rows = [(r, d) for (nm, lv, d), r in zip(ins, G) if lv == 1][3:6]
    b = ZstdBatch(max_slices=len(rows), max_slice_bytes=5 << 20)
    try:
        frames = gpu_compress_kw(b, [d for _, d in rows], level=1, reference=True, check=True)
        for (r, d), f in zip(rows, frames):
            assert (len(f), helpers.sha256(f)) == (r["oneshot_len"], r["oneshot_sha256"]), r["name"]
    finally:
        b.close()
    # the streaming entry point: what a Ktor response of 2 MiB and of 24 MiB + 3 bytes becomes at level 1 (finish = false pieces, then finish = true)
    lib = _lib.load()
    z = helpers.live_libzstd()
    rng = np.random.default_rng(7)
    two = b"".join(d for _, _, d in ins[:3])[: 2 << 20]
    big = bytearray()
    while len(big) < (24 << 20) + 3:
        k = int(rng.integers(0, len(ins)))
        big += ins[k][2][: int(rng.integers(1000, 900000))]
    big = bytes(big[: (24 << 20) + 3])
    for d, cuts in ((two, [0, 300000, 1 << 20, len(two)]), (big, [0, 5 << 20, (17 << 20) + 11, len(big)])):
        f = _stream_through_abi(lib, d, 1, cuts)
        assert f == o.compress_fast_buffered(d, 1, stream=1), len(d)
        if z is not None:
            assert f == z.compress_streaming(d, cuts, out_chunk=8192, level=1), len(d)
        assert hashlib.sha256(ZstdDecompressor().transform_bytes(f)).digest() == hashlib.sha256(d).digest()
    # ... and one-shot through the reference's driver (finish = true from the first call), levels 1 and -3
    for level in (1, -3):
        f = ZstdCompressor(compression_level=level).transform_bytes(two)
        assert f == o.compress_fast_buffered(two, level, stream=3), level


def test_a_short_tail_of_one_byte_is_an_rle_block():
    """ZSTD_compressBlock_internal turns a block of one repeated byte into an RLE block when the entropy stage's result is below 25 bytes
    (0 = the block would go out raw), whatever the parser found: a tail of 10 zeros after 256 KiB is an RLE block at level 1 too, where the
    "fast" parser finds nothing in it.  Found by the differential fuzz (seed 102: one slice of 262 154 bytes, levels 1 / 2 / negative, every
    call pattern).  helpers.rle_tail_cases() against tests/golden/zstd_rle_tail_golden.json (libzstd 1.5.7) through the batch call at
    levels -5 .. 4, through the streaming entry points under the reference's driver against the live library, and back through the decoder."""
    from kompressor_amd.batch import ZstdBatch
    from kompressor_amd.zstd import ZstdCompressor
    G = helpers.rle_tail_golden()["rows"]
    cases = helpers.rle_tail_cases()
    datas = [d for _, d in cases]
    z = helpers.require_live_libzstd()
    b = ZstdBatch(max_slices=len(datas), max_slice_bytes=2 << 20)
    try:
        for lvl in (-5, -1, 1, 2, 3, 4):
            frames = gpu_compress_kw(b, datas, level=lvl)
            for (name, d), f in zip(cases, frames):
                if lvl == 4 and len(d) <= 262144:
                    continue                              # (level 4 is "greedy" there: not served, the frame is empty)
                assert [len(f), helpers.sha256(f)] == G[name][str(lvl)], (name, lvl)
            if lvl in (1, 3):
                back, st = gpu_decompress(b, frames, [len(d) for d in datas])
                assert st == [0] * len(frames) and back == datas
        # the reference's driver (staged 128 KiB chunks, output slices of max(8192, n / 10)): the same rule in the same function
        o = helpers.oracle()
        for name in ("run_262144_10_65", "run_131072_9_0", "run_171072_25_8"):
            d = dict(cases)[name]
            for lvl in (1, 3):
                got = ZstdCompressor(lvl).transform_bytes(d)
                want = o.compress_fast_buffered(d, lvl, stream=3) if lvl == 1 else o.compress_buffered(d, True)
                assert got == want, (name, lvl)
                assert z.decompress(got, len(d)) == d
    finally:
        b.close()
