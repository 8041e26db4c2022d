// zstd_common.h -- structs and small helpers shared by the zstd kernels and
// the host launcher.  Requires kx_wave.h (product) or the emulator's shadow.
#pragma once
// (the including translation unit provides kx_wave.h: csrc/ for gfx950, tests/emu/ for the CPU emulator)

// One LZ sequence as the match kernel hands it to the entropy kernel.
// Same meaning as libzstd's seqStore entry behind ZSTD_compressStream2
// (reference call site: kompressor-zstd--nativelib/src/jvmCommonMain/jni/Wrapper.cpp:112).
struct KSeq { u32 offBase; u16 litLength; u16 mlBase; };

struct KSliceMeta {
    u32 nbSeq;      // sequences stored
    u32 litSize;    // literal bytes stored by the match kernel (last literals excluded)
    u32 lastLL;     // trailing literals: src[n-lastLL, n)
    u32 longType;   // 0 none, 1 litLength > 65535 at longPos, 2 matchLength-3 > 65535 at longPos
    u32 longPos;
    u32 status;     // 0 ok, else kernel guard tripped
    u32 pad[2];
};

// Hash-table entry = epoch << 23 | check << 18 | index, index = position + 2 (0 = empty).  check: 5 bits derived from
// the bytes a candidate is compared on (the level-3 parser: 8 bytes for the long table, 4 for the short one); an entry
// whose check differs from the current position's cannot pass the compare, so its source line is never fetched.  The
// other parsers leave the field zero and ignore it.
#define KX_IDX_BITS 18
#define KX_IDX_MASK ((1u << KX_IDX_BITS) - 1)
#define KX_CHK_SHIFT KX_IDX_BITS
#define KX_CHK_BITS 5
#define KX_TAG_SHIFT (KX_IDX_BITS + KX_CHK_BITS)
#define KX_CHK_MASK (((1u << KX_CHK_BITS) - 1u) << KX_CHK_SHIFT)
#define KX_TAG_MASK (~((1u << KX_TAG_SHIFT) - 1u))
#define KX_EPOCH_MAX ((1u << (32 - KX_TAG_SHIFT)) - 1)
// block mode (slices up to 2 MiB, per-slice tables zeroed per batch, no epoch): entry = check << 22 | index
#define KX_BLK_IDX_BITS 22
#define KX_BLK_IDX_MASK ((1u << KX_BLK_IDX_BITS) - 1)
#define KX_BLK_CHK_MASK (((1u << KX_CHK_BITS) - 1u) << KX_BLK_IDX_BITS)
#define KX_TBL_LONG  (1u << 16)
#define KX_TBL_SHORT (1u << 15)    /* chainLog <= 15 for slices <= 128 KiB */
#define KX_TBL_ENTRIES (KX_TBL_LONG + KX_TBL_SHORT)
#define KX_MAX_SLICE (128u * 1024u)

// ---- frames of several blocks (slices above 128 KiB, up to KX_MAX_BIG_SLICE) -------------------------
// The slice is compressed block by block; what libzstd carries from one block of a frame to the next
// (ZSTD_compressedBlockState_t, window, hash tables) lives in HBM between the rounds.
#define KX_MAX_BIG_SLICE (1u << 30)          /* level 3; beyond the window (2 MiB) it slides as libzstd's does (KFrameState) */
#define KX_BLK_WIDE_FROM ((4u << 20) - 64u)  /* slices from this size on: table entries are plain 32-bit indices (no check bits) */
#define KX_BLOCK_MAX (128u * 1024u)
#define KX_BIG_TBL_LONG  (1u << 17)
#define KX_BIG_TBL_SHORT (1u << 16)
#define KX_BIG_TBL_ENTRIES (KX_BIG_TBL_LONG + KX_BIG_TBL_SHORT)     /* per slice, plain indices, zeroed per batch */
struct KFrameState {
    u32 ipos;          // input consumed = start of the next block
    u32 opos;          // frame bytes written
    u32 blockSize;     // size of the next block (0 = frame finished)
    u32 first;         // 1 until the first block is out
    u32 rep[3];        // repcodes confirmed by the last compressed block
    u32 hufValid;      // a Huffman table of an earlier block exists (HUF_repeat_check)
    u32 hufSel;        // which of the two table slots holds it
    int savings;       // input bytes - frame bytes of the blocks so far (ZSTD_compress_frameChunk)
    // libzstd's window as it stands for the next block (indices = stream position + 2; ZSTD_window_t after
    // ZSTD_window_update for the block's chunk and ZSTD_window_enforceMaxDist for the block) ...
    u32 lowLimit, dictLimit;
    // ... and where libzstd's staging buffer stands (the buffered frames: input taken in chunks of 128 KiB into a buffer of
    // window + 128 KiB bytes that wraps; the lap before the current one is the "extDict" segment)
    u32 bufPos;        // offset of the current chunk in the staging buffer
    u32 extBase;       // stream position that sat at offset 0 of the buffer during the previous lap
    u32 chunkEnd;      // end of the current chunk (stream position)
    u32 wflags;        // 1: an older segment exists; 2: the rest of the input is compressed in place (one chunk, new segment)
};

// How one block sees the window: ext = libzstd parses it with the extDict variant of the double-fast loop, the older
// segment being [dictStartIndex, prefixStartIndex); else the regular variant, whose lowest valid index follows from
// dictLimit and maxDist (ZSTD_getLowestPrefixIndex).
struct KBlockWin { u32 ext, dictStartIndex, prefixStartIndex, dictLimit, maxDist; };
KX_DEV KBlockWin kx_block_window(u32 lowLimit, u32 dictLimit, u32 ipos, u32 blockSize, u32 windowLog)
{
    KBlockWin w; w.ext = 0; w.dictStartIndex = 0; w.prefixStartIndex = 0; w.dictLimit = dictLimit; w.maxDist = 1u << windowLog;
    if (lowLimit < dictLimit) {                                 // ZSTD_window_hasExtDict
        u32 const endIdx = ipos + blockSize + 2u;
        u32 const low = (endIdx - lowLimit > w.maxDist) ? endIdx - w.maxDist : lowLimit;      // ZSTD_getLowestMatchIndex
        u32 const ps = dictLimit > low ? dictLimit : low;
        if (ps != low) { w.ext = 1; w.dictStartIndex = low; w.prefixStartIndex = ps; }
    }
    return w;
}
// max(dictLimit, curr - maxDist): ZSTD_getLowestPrefixIndex
KX_DEV u32 kx_lowest_prefix(u32 curr, u32 dictLimit, u32 maxDist) { return (curr - dictLimit > maxDist) ? curr - maxDist : dictLimit; }

// level-3 parameters of a one-shot slice of n bytes: what ZSTD_getCParams(3, n, 0) yields after size adjustment.
struct KParams { u32 windowLog, chainLog, hashLog, minMatch; };

KX_DEV KParams kx_params_l3(u32 n)
{
    KParams p;
    if (n <= 16384)       { p.windowLog = 14; p.chainLog = 14; p.hashLog = 15; p.minMatch = 4; }
    else if (n <= 131072) { p.windowLog = 17; p.chainLog = 15; p.hashLog = 16; p.minMatch = 5; }
    else if (n <= 262144) { p.windowLog = 18; p.chainLog = 16; p.hashLog = 16; p.minMatch = 4; }
    else                  { p.windowLog = 21; p.chainLog = 16; p.hashLog = 17; p.minMatch = 5; }
    u32 const srcLog = (n < 64) ? 6 : kx_hb32(n - 1) + 1;
    if (p.windowLog > srcLog) p.windowLog = srcLog;
    if (p.hashLog > p.windowLog + 1) p.hashLog = p.windowLog + 1;
    if (p.chainLog > p.windowLog) p.chainLog = p.windowLog;
    if (p.windowLog < 10) p.windowLog = 10;
    return p;
}

// Level 4 is double-fast in two of its four size classes (ZSTD_getCParams(4, n, 0): above 16 KiB up to 128 KiB {17,17,17, minMatch 4},
// above 256 KiB {21,18,18, minMatch 5}; the other two are strategy "greedy", another parser).  One-block slices of the first
// class are served: long table up to 2^17 entries, short table up to 2^17 (KX_TBL4_*: 1 MiB per team, from a table set of
// its own).  ok = false: not such a slice.
#define KX_TBL4_LONG (1u << 17)
#define KX_TBL4_ENTRIES (1u << 18)
// ... and, on the block-chain path, of the second (frames of several blocks: per-slice tables of 2^18 + 2^18 entries)
#define KX_BIG4_LONG (1u << 18)
#define KX_BIG4_ENTRIES (1u << 19)
KX_DEV bool kx_l4_served(u32 n) { return (n > 16384u && n <= 131072u) || n > 262144u; }
KX_DEV KParams kx_params_l4(u32 n, bool& ok)
{
    KParams p; p.windowLog = 17; p.chainLog = 17; p.hashLog = 17; p.minMatch = 4;
    if (n > 262144u) { p.windowLog = 21; p.chainLog = 18; p.hashLog = 18; p.minMatch = 5; }
    ok = kx_l4_served(n);
    u32 const srcLog = (n < 64) ? 6 : kx_hb32(n - 1) + 1;
    if (p.windowLog > srcLog) p.windowLog = srcLog;
    if (p.hashLog > p.windowLog + 1) p.hashLog = p.windowLog + 1;
    if (p.chainLog > p.windowLog) p.chainLog = p.windowLog;
    return p;
}

// Level 2 has one double-fast row: sizes above 128 KiB up to 256 KiB (window 18, chain 14, hash 14, minMatch 5); its other
// rows are "fast" ones (zstd_match_fast.h).  A batch at level 2 therefore goes through both block-chain kernels, each
// taking the slices of its class (KFrameArgs.cls: 0 every slice, 1 only that size class, 2 only the others).
KX_DEV KParams kx_params_l2_dfast() { KParams p; p.windowLog = 18; p.chainLog = 14; p.hashLog = 14; p.minMatch = 5; return p; }
KX_DEV bool kx_in_class(u32 cls, u32 n) { return cls == 0u || ((n > 131072u && n <= 262144u) == (cls == 1u)); }

// What a formatted dictionary gives a DEcoder besides its content (libzstd: ZSTD_loadDEntropy): the literals' Huffman table as weights, the
// three sequence tables as normalised counts -- the first block of a frame may refer to them as "the previous block's" (tree-less literals,
// "repeat" sequence tables) --, the repeat offsets a frame starts with, and the ID a frame's header may name.  Host: zstd_cdict_host.h.
struct KDictDPrior { u8 weights[256]; u32 nw, hufLog; short norm[3][64]; u32 log[3], max[3]; u32 rep[3]; u32 dictID; };       // [0] LL, [1] OF, [2] ML

KX_DEV u32 kx_frame_header_size(u32 n, u32 windowLog = 21)
{
    // magic(4) + FHD(1) + FCS; single segment while the window (2 MiB at level 3; 512 KiB / 1 MiB at the "fast" levels) covers the
    // content, else a window descriptor byte as well
    return 5 + ((n < 256) ? 1 : (n < 65536 + 256) ? 2 : 4) + (n > (1u << windowLog) ? 1u : 0u);
}
// frame header of a one-shot frame (content size known): returns its size
KX_DEV u32 kx_write_frame_header(u8* dst, u32 n, u32 windowLog = 21)
{
    u32 const fcsCode = (n >= 256) + (n >= 65536 + 256);
    bool const single = n <= (1u << windowLog);
    kx_st32(dst, 0xFD2FB528u);
    dst[4] = (u8)((single ? 1u << 5 : 0u) + (fcsCode << 6));
    u32 p = 5;
    if (!single) dst[p++] = (u8)((windowLog - 10) << 3);
    if (fcsCode == 0) dst[p++] = (u8)n;
    else if (fcsCode == 1) { kx_st16(dst + p, n - 256); p += 2; }
    else { kx_st32(dst + p, n); p += 4; }
    return p;
}

// 64-bit multiplicative hashes, top bits. Written with 32-bit pieces: only the
// high dword of the low 64 bits of the product is needed.
KX_DEV u32 kx_mulhi64_hi32(u32 w0, u32 w1, u32 p0, u32 p1)
{
    return kx_umulhi(w0, p0) + w0 * p1 + w1 * p0;
}
KX_DEV u32 kx_hash_long(u64 w, u32 hBits)
{
    u32 const hi = kx_mulhi64_hi32((u32)w, (u32)(w >> 32), 0xB7A56463u, 0xCF1BBCDCu);
    return hi >> (32 - hBits);
}
// check bits of a long-table entry: the 5 bits below the hash in the same product (all 8 bytes take part)
KX_DEV u32 kx_chk_long(u64 w, u32 hBits)
{
    u32 const hi = kx_mulhi64_hi32((u32)w, (u32)(w >> 32), 0xB7A56463u, 0xCF1BBCDCu);
    return (hi >> (32 - hBits - KX_CHK_BITS)) & ((1u << KX_CHK_BITS) - 1u);
}
// ... of a short-table entry: from the first 4 bytes only -- libzstd accepts a short candidate on 4 equal bytes even
// where the table hashes 5
KX_DEV u32 kx_chk_short(u64 w) { return ((u32)w * 2654435761u) >> (32 - KX_CHK_BITS); }
KX_DEV u32 kx_hash_short(u64 w, u32 hBits, u32 mls)
{
    if (mls == 4) return ((u32)w * 2654435761u) >> (32 - hBits);
    // mls == 5: ((w << 24) * 889523592379) >> (64 - hBits)
    u64 const x = w << 24;
    u32 const hi = kx_mulhi64_hi32((u32)x, (u32)(x >> 32), 0x1BBCDCBBu, 0xCFu);
    return hi >> (32 - hBits);
}

// minMatch 4..7 (levels 1 and 2 use 5, 6 and 7)
KX_DEV u32 kx_hash_short_any(u64 w, u32 hBits, u32 mls)
{
    if (mls <= 5) return kx_hash_short(w, hBits, mls);
    if (mls == 6) { u64 const x = w << 16; return kx_mulhi64_hi32((u32)x, (u32)(x >> 32), 0xBCDCBF9Bu, 0xCF1Bu /* 227718039650203 */) >> (32 - hBits); }
    { u64 const x = w << 8; return kx_mulhi64_hi32((u32)x, (u32)(x >> 32), 0xDCBFA563u, 0xCF1BBCu /* 58295818150454627 */) >> (32 - hBits); }
}

// 8 bytes at position p of a slice of n bytes without touching bytes >= n
// (p < n, n >= 8): bytes past the end read as zero.
KX_DEV u64 kx_ld64_clamped(const u8* src, int p, int n)
{
    if (p + 8 <= n) return kx_ld64(src + p);
    return kx_ld64(src + n - 8) >> (8 * (p + 8 - n));
}

// Workgroups go to the eight XCDs round robin by their index, so with "workgroup b takes slice b" a batch whose content
// has a period of 8 or 16 slices (the bench's class mix does: ...B at 7, R at 15 -> one XCD gets every incompressible slice)
// leaves one XCD with the expensive slices and seven waiting for it.  Here the workgroups of one XCD take a contiguous
// eighth of the batch instead: it (= workgroup index + k * grid size) is a virtual workgroup index out of n.
KX_DEV u32 kx_xcd_chunk(u32 it, u32 n)
{
    u32 const x = it & 7u, j = it >> 3, q = n >> 3, r = n & 7u;
    return x * q + (x < r ? x : r) + j;
}
