// zstd_decode.h -- zstd frame decoder, one wave per frame (RFC 8878): frame
// header (single-segment or window descriptor, optional content size and
// content checksum), raw / RLE / compressed blocks, literals (raw, RLE,
// Huffman 1 or 4 streams, tree reuse), sequences (predefined / RLE / FSE /
// repeat tables, 3 repeat offsets) and LZ execution.
//
// Replaces libzstd's ZSTD_decompressStream behind the reference's
// ZstdDecompressor (kompressor-zstd--nativelib/src/jvmCommonMain/jni/Wrapper.cpp:178,
// driven by .../zstd/ZstdDecompressor.jvm.kt:25-39) for frames whose content
// fits the caller's per-frame capacity.
//
// Execution shape: lane 0 parses headers and builds the (LDS-resident) Huffman
// and FSE decoding tables; lanes 0..3 walk the four Huffman streams; lane 0
// decodes sequences 64 at a time into LDS, then all 64 lanes execute their
// literal and match copies cooperatively.
#pragma once
#include "zstd_common.h"

struct KDecodeArgs {
    const u8* src; const u64* in_off; const u32* in_len; u32 n_slices;
    u8* dst; const u64* out_off; const u32* out_cap; u32* out_len; u32* status;
    u8* lits; u32 lit_cap;                 // per frame: decoded literals of one block
};

enum { KZE_GENERIC = 1, KZE_PREFIX = 10, KZE_FRAMEPARAM = 14, KZE_WINDOW = 16, KZE_CORRUPT = 20, KZE_CHECKSUM = 22,
       KZE_LITHDR = 24, KZE_DICT = 32, KZE_DSTSMALL = 70, KZE_SRCSIZE = 72 };

struct KDecodeLds {
    u16 huf[4096];          // Huffman decoding table: symbol | nbBits << 8
    u32 fse[3][512];        // LL, OF, ML: newStateBase | nbBits << 16 | symbol << 24
    short norm[64];
    u16 symnext[64];
    u8 tsym[512];
    u8 weights[256];
    u32 rank[16];
    u32 stage[194];         // 64 x (litLength, matchLength, offset) + count + error
    u32 bc[16];             // lane 0 -> wave broadcast slots
};

// ---- forward (LSB-first) bit reader over bytes, for table descriptions ----
struct KFwdBits { const u8* p; u32 size; u32 bit; };
KX_DEV u32 kfb_peek(const KFwdBits& b, u32 n)
{
    u32 const byte = b.bit >> 3; u64 w = 0;
    for (u32 i = 0; i < 5; i++) if (byte + i < b.size) w |= (u64)b.p[byte + i] << (8 * i);
    return (u32)((w >> (b.bit & 7)) & ((1ull << n) - 1ull));
}

// FSE table description -> norm[], returns bytes consumed (0 on error)
KX_DEV u32 kfse_read_ncount(short* norm, u32* maxSymbolValuePtr, u32* tableLogPtr, const u8* p, u32 size, u32 maxLog)
{
    KFwdBits b; b.p = p; b.size = size; b.bit = 0;
    if (size < 1) return 0;
    u32 const tableLog = kfb_peek(b, 4) + 5; b.bit += 4;
    if (tableLog > maxLog) return 0;
    *tableLogPtr = tableLog;
    int remaining = (1 << tableLog) + 1, threshold = 1 << tableLog; u32 nbBits = tableLog + 1;
    u32 sym = 0; u32 const maxSV = *maxSymbolValuePtr; bool previous0 = false;
    while (remaining > 1 && sym <= maxSV) {
        if (previous0) {
            for (;;) {
                u32 const r = kfb_peek(b, 2); b.bit += 2;
                for (u32 i = 0; i < r && sym <= maxSV; i++) norm[sym++] = 0;
                if (r != 3) break;
                if ((b.bit >> 3) > size) return 0;
            }
            if (sym > maxSV) break;      // remaining > 1 with no symbol left: caught below
        }
        int const max = (2 * threshold - 1) - remaining;
        int count;
        u32 const v = kfb_peek(b, nbBits);
        if ((int)(v & (u32)(threshold - 1)) < max) { count = (int)(v & (u32)(threshold - 1)); b.bit += nbBits - 1; }
        else { count = (int)(v & (u32)(2 * threshold - 1)); if (count >= threshold) count -= max; b.bit += nbBits; }
        count--;
        remaining -= count < 0 ? -count : count;
        norm[sym++] = (short)count;
        previous0 = (count == 0);
        while (remaining < threshold) { nbBits--; threshold >>= 1; }
        if ((b.bit >> 3) > size) return 0;
    }
    if (remaining != 1) return 0;
    u32 const used = (b.bit + 7) >> 3;
    if (used > size) return 0;
    for (u32 s = sym; s <= maxSV; s++) norm[s] = 0;
    *maxSymbolValuePtr = sym - 1;
    return used;
}

// FSE decoding table: entry = newStateBase | nbBits << 16 | symbol << 24
KX_DEV void kfse_build_dtable(u32* dt, const short* norm, u32 maxSymbolValue, u32 tableLog, u16* symnext, u8* tsym)
{
    u32 const tableSize = 1u << tableLog, mask = tableSize - 1;
    u32 const step = (tableSize >> 1) + (tableSize >> 3) + 3;
    u32 high = tableSize - 1;
    for (u32 s = 0; s <= maxSymbolValue; s++) {
        if (norm[s] == -1) { tsym[high--] = (u8)s; symnext[s] = 1; }
        else symnext[s] = (u16)norm[s];
    }
    u32 pos = 0;
    for (u32 s = 0; s <= maxSymbolValue; s++) {
        for (int i = 0; i < norm[s]; i++) {
            tsym[pos] = (u8)s;
            pos = (pos + step) & mask;
            while (pos > high) pos = (pos + step) & mask;
        }
    }
    for (u32 u = 0; u < tableSize; u++) {
        u32 const s = tsym[u]; u32 const next = symnext[s]++;
        u32 const nb = tableLog - kx_hb32(next);
        dt[u] = (((next << nb) - tableSize) & 0xFFFFu) | (nb << 16) | (s << 24);
    }
}

// ---- backward bit reader (streams written LSB-first, read from the end) ----
// `bits` = number of unread bits; reads never touch memory past the stream end.
struct KBackBits { const u8* base; int bits; };
KX_DEV bool kbb_init(KBackBits& b, const u8* p, u32 size)
{
    b.base = p; b.bits = 0;
    if (size == 0) return false;
    u32 const last = p[size - 1];
    if (last == 0) return false;
    b.bits = (int)(8 * (size - 1) + kx_hb32(last));
    return true;
}
// value of the next n (<= 32) bits without consuming; bits beyond the start read as 0
KX_DEV u32 kbb_peek(const KBackBits& b, u32 n)
{
    if (n == 0) return 0;
    int const top = b.bits;                       // exclusive
    int const end = (top + 7) >> 3;               // bytes [end-8, end)
    u64 const w = kx_ld64(b.base + end - 8);
    int const shift = 64 - (8 * end - top) - (int)n;
    u64 v = w >> shift;
    int const lo = top - (int)n;
    if (lo < 0) { v = (v >> (-lo)) << (-lo); }    // bits below the stream start are zero
    return (u32)(v & ((1ull << n) - 1ull));
}

// ---- Huffman table description -> LDS decoding table; returns bytes consumed, 0 on error
KX_DEV u32 khuf_read_dtable(KDecodeLds& lds, const u8* p, u32 size, u32* tableLogOut)
{
    if (size < 1) return 0;
    u32 const hb = p[0]; u32 nw = 0, used;
    if (hb >= 128) {
        nw = hb - 127; used = 1 + (nw + 1) / 2;
        if (used > size) return 0;
        for (u32 i = 0; i < nw; i += 2) { lds.weights[i] = p[1 + i / 2] >> 4; lds.weights[i + 1] = p[1 + i / 2] & 15; }
    } else {
        used = 1 + hb;
        if (hb == 0 || used > size) return 0;
        u32 maxSV = 12, tl = 0;
        u32 const h = kfse_read_ncount(lds.norm, &maxSV, &tl, p + 1, hb, 6);
        if (h == 0 || maxSV > 12) return 0;
        kfse_build_dtable(lds.fse[0], lds.norm, maxSV, tl, lds.symnext, lds.tsym);
        KBackBits b;
        if (!kbb_init(b, p + 1 + h, hb - h)) return 0;
        if (b.bits < (int)(2 * tl)) return 0;
        u32 s1 = kbb_peek(b, tl); b.bits -= (int)tl;
        u32 s2 = kbb_peek(b, tl); b.bits -= (int)tl;
        for (;;) {
            if (nw >= 255) return 0;
            u32 const e1 = lds.fse[0][s1];
            lds.weights[nw++] = (u8)(e1 >> 24);
            u32 const nb1 = (e1 >> 16) & 0xFF;
            if (b.bits < (int)nb1) { if (nw >= 255) return 0; lds.weights[nw++] = (u8)(lds.fse[0][s2] >> 24); break; }
            s1 = (e1 & 0xFFFFu) + kbb_peek(b, nb1); b.bits -= (int)nb1;
            if (nw >= 255) return 0;
            u32 const e2 = lds.fse[0][s2];
            lds.weights[nw++] = (u8)(e2 >> 24);
            u32 const nb2 = (e2 >> 16) & 0xFF;
            if (b.bits < (int)nb2) { if (nw >= 255) return 0; lds.weights[nw++] = (u8)(lds.fse[0][s1] >> 24); break; }
            s2 = (e2 & 0xFFFFu) + kbb_peek(b, nb2); b.bits -= (int)nb2;
        }
    }
    // last weight is implied: total must complete a power of two
    u32 total = 0;
    for (u32 i = 0; i < 16; i++) lds.rank[i] = 0;
    for (u32 i = 0; i < nw; i++) { u32 const w = lds.weights[i]; if (w > 12) return 0; lds.rank[w]++; if (w) total += 1u << (w - 1); }
    if (total == 0) return 0;
    u32 const tableLog = kx_hb32(total) + 1;
    if (tableLog > 12) return 0;
    u32 const rest = (1u << tableLog) - total;
    if (rest & (rest - 1)) return 0;              // must be a power of two
    u32 const lastW = kx_hb32(rest) + 1;
    lds.weights[nw] = (u8)lastW; lds.rank[lastW]++; nw++;
    if (lds.rank[1] < 2 || (lds.rank[1] & 1)) return 0;
    // starting index of every weight, then fill
    u32 next = 0;
    for (u32 w = 1; w <= tableLog; w++) { u32 const cur = next; next += lds.rank[w] << (w - 1); lds.rank[w] = cur; }
    for (u32 s = 0; s < nw; s++) {
        u32 const w = lds.weights[s];
        if (!w) continue;
        u32 const len = 1u << (w - 1); u32 const start = lds.rank[w]; u32 const nb = tableLog + 1 - w;
        for (u32 i = 0; i < len; i++) lds.huf[start + i] = (u16)(s | (nb << 8));
        lds.rank[w] += len;
    }
    *tableLogOut = tableLog;
    return used;
}

// one lane decodes one Huffman stream of `count` symbols; returns false on corruption
KX_DEV bool khuf_decode_stream(const KDecodeLds& lds, u32 tableLog, const u8* p, u32 size, u8* out, u32 count)
{
    KBackBits b;
    if (!kbb_init(b, p, size)) return false;
    u32 i = 0;
    while (i < count) {
        if (b.bits <= 0) return false;
        // 64-bit window whose top bit is the next unread bit of the stream
        int const end = (b.bits + 7) >> 3;
        int const slack = 8 * end - b.bits;                 // already-consumed bits of the top byte
        u64 w = kx_ld64(b.base + end - 8) << slack;
        int avail = 64 - slack; if (avail > b.bits) avail = b.bits;
        bool const tail = (avail == b.bits);                 // window reaches the stream start
        if (avail < 64) w &= ~0ull << (64 - avail);          // bits before the stream start read as zero
        int used = 0;
        while (i < count) {
            if (used + (int)tableLog > avail && !tail) break;                // refill
            u32 const e = lds.huf[(u32)(w >> (64 - tableLog))];
            int const nb = (int)(e >> 8);
            if (used + nb > avail) return false;
            out[i++] = (u8)e; w <<= nb; used += nb;
        }
        b.bits -= used;
    }
    return b.bits == 0;
}

KX_DEV u64 kxxh_round(u64 acc, u64 in)
{
    acc += in * 14029467366897019727ULL; acc = (acc << 31) | (acc >> 33); return acc * 11400714785074694791ULL;
}
KX_DEV u64 kxxh_merge(u64 acc, u64 v) { v = kxxh_round(0, v); acc ^= v; return acc * 11400714785074694791ULL + 9650029242287828579ULL; }
KX_DEV u64 kxxh64(const u8* p, u32 len)
{
    const u64 P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P3 = 1609587929392839161ULL,
              P4 = 9650029242287828579ULL, P5 = 2870177450012600261ULL;
    u64 h; u32 i = 0;
    if (len >= 32) {
        u64 v1 = P1 + P2, v2 = P2, v3 = 0, v4 = 0 - P1;
        for (; i + 32 <= len; i += 32) {
            v1 = kxxh_round(v1, kx_ld64(p + i)); v2 = kxxh_round(v2, kx_ld64(p + i + 8));
            v3 = kxxh_round(v3, kx_ld64(p + i + 16)); v4 = kxxh_round(v4, kx_ld64(p + i + 24));
        }
        h = ((v1 << 1) | (v1 >> 63)) + ((v2 << 7) | (v2 >> 57)) + ((v3 << 12) | (v3 >> 52)) + ((v4 << 18) | (v4 >> 46));
        h = kxxh_merge(h, v1); h = kxxh_merge(h, v2); h = kxxh_merge(h, v3); h = kxxh_merge(h, v4);
    } else h = P5;
    h += len;
    for (; i + 8 <= len; i += 8) { h ^= kxxh_round(0, kx_ld64(p + i)); h = ((h << 27) | (h >> 37)) * P1 + P4; }
    if (i + 4 <= len) { h ^= (u64)kx_ld32(p + i) * P1; h = ((h << 23) | (h >> 41)) * P2 + P3; i += 4; }
    for (; i < len; i++) { h ^= p[i] * P5; h = ((h << 11) | (h >> 53)) * P1; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

KX_DEV u32 kx_ll_base(u32 c)
{
    static const u32 LL_base[36] = { 0,1,2,3,4,5,6,7, 8,9,10,11,12,13,14,15, 16,18,20,22,24,28,32,40,
        48,64,0x80,0x100,0x200,0x400,0x800,0x1000, 0x2000,0x4000,0x8000,0x10000 };
    return LL_base[c];
}
KX_DEV u32 kx_ml_base(u32 c)
{
    static const u32 ML_base[53] = { 3,4,5,6,7,8,9,10, 11,12,13,14,15,16,17,18, 19,20,21,22,23,24,25,26, 27,28,29,30,31,32,33,34,
        35,37,39,41,43,47,51,59, 67,83,99,0x83,0x103,0x203,0x403,0x803, 0x1003,0x2003,0x4003,0x8003,0x10003 };
    return ML_base[c];
}
KX_DEV u32 kxd_ll_bits(u32 c)
{
    static const u8 LL_bits[36] = { 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0, 1,1,1,1,2,2,3,3, 4,6,7,8,9,10,11,12, 13,14,15,16 };
    return LL_bits[c];
}
KX_DEV u32 kxd_ml_bits(u32 c)
{
    static const u8 ML_bits[53] = { 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,
        1,1,1,1,2,2,3,3, 4,4,5,7,8,9,10,11, 12,13,14,15,16 };
    return ML_bits[c];
}

// lane 0: set up the decoding table of one symbol type. returns bytes consumed, KXD_FAIL on error
#define KXD_FAIL 0xFFFFFFFFu
KX_DEV u32 kxd_seq_table(KDecodeLds& lds, int t, u32 mode, const u8* p, u32 size, u32* tableLog, bool* valid)
{
    static const short LL_defaultNorm[36] = { 4,3,2,2,2,2,2,2, 2,2,2,2,2,1,1,1, 2,2,2,2,2,2,2,2, 2,3,2,1,1,1,1,1, -1,-1,-1,-1 };
    static const short ML_defaultNorm[53] = { 1,4,3,2,2,2,2,2, 2,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1,
                                              1,1,1,1,1,1,1,1, 1,1,1,1,1,1,-1,-1, -1,-1,-1,-1,-1 };
    static const short OF_defaultNorm[29] = { 1,1,1,1,1,1,2,2, 2,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1, -1,-1,-1,-1,-1 };
    u32 const maxSym = (t == 0) ? 35 : (t == 1) ? 31 : 52;
    u32 const maxLog = (t == 0) ? 9 : (t == 1) ? 8 : 9;
    if (mode == 0) {
        const short* dn = (t == 0) ? LL_defaultNorm : (t == 1) ? OF_defaultNorm : ML_defaultNorm;
        u32 const dmax = (t == 0) ? 35 : (t == 1) ? 28 : 52; u32 const dlog = (t == 1) ? 5 : 6;
        for (u32 s = 0; s <= dmax; s++) lds.norm[s] = dn[s];
        kfse_build_dtable(lds.fse[t], lds.norm, dmax, dlog, lds.symnext, lds.tsym);
        *tableLog = dlog; *valid = true;
        return 0;
    }
    if (mode == 1) {
        if (size < 1 || p[0] > maxSym) return KXD_FAIL;
        lds.fse[t][0] = (u32)p[0] << 24;          // nbBits 0, next state 0
        *tableLog = 0; *valid = true;
        return 1;
    }
    if (mode == 2) {
        u32 maxSV = maxSym, tl = 0;
        u32 const h = kfse_read_ncount(lds.norm, &maxSV, &tl, p, size, maxLog);
        if (h == 0) return KXD_FAIL;
        kfse_build_dtable(lds.fse[t], lds.norm, maxSV, tl, lds.symnext, lds.tsym);
        *tableLog = tl; *valid = true;
        return h;
    }
    return *valid ? 0 : KXD_FAIL;                // repeat
}

KX_DEV void kxd_wave_copy(u8* dst, const u8* src, u32 n, int lane)
{
    u32 i = (u32)lane * 8u;
    for (; i + 8 <= n; i += 512u) kx_st64(dst + i, kx_ld64(src + i));
    u32 const tail = n & ~7u;
    if (lane < (int)(n - tail)) dst[tail + lane] = src[tail + lane];
}

KX_DEV void zstd_decode_frame(const KDecodeArgs& a, KDecodeLds& lds, u32 f, int lane)
{
    const u8* const src = a.src + a.in_off[f];
    u32 const srcSize = a.in_len[f];
    u8* const dst = a.dst + a.out_off[f];
    u32 const cap = a.out_cap[f];
    u8* const lits = a.lits + (size_t)f * a.lit_cap;
    u32 err = 0;

    // ---- frame header (every lane computes the same thing) ---------------
    u32 pos = 0; u32 hasContent = 0, checksum = 0; u64 contentSize = 0; u64 windowSize = 0;
    if (srcSize < 5) err = KZE_SRCSIZE;
    else if (kx_ld32(src) != 0xFD2FB528u) err = KZE_PREFIX;
    if (!err) {
        u32 const fhd = src[4]; u32 const dictId = fhd & 3, single = (fhd >> 5) & 1, fcsId = fhd >> 6;
        checksum = (fhd >> 2) & 1;
        if (fhd & 0x08) err = KZE_FRAMEPARAM;
        pos = 5;
        if (!err && !single) {
            u32 const wd = src[pos++]; u32 const wlog = 10 + (wd >> 3);
            if (wlog > 31) err = KZE_WINDOW;
            else { windowSize = 1ull << wlog; windowSize += (windowSize >> 3) * (wd & 7); }
        }
        u32 const didSize = dictId == 3 ? 4 : dictId;
        u32 const fcsSize = fcsId == 0 ? single : (fcsId == 1 ? 2 : fcsId == 2 ? 4 : 8);
        if (!err && pos + didSize + fcsSize > srcSize) err = KZE_SRCSIZE;
        if (!err) {
            u32 did = 0;
            for (u32 i = 0; i < didSize; i++) did |= (u32)src[pos + i] << (8 * i);
            if (did) err = KZE_DICT;
            pos += didSize;
            if (fcsSize) {
                hasContent = 1;
                if (fcsSize == 1) contentSize = src[pos];
                else if (fcsSize == 2) contentSize = (u64)kx_ld16(src + pos) + 256;
                else if (fcsSize == 4) contentSize = kx_ld32(src + pos);
                else contentSize = kx_ld64(src + pos);
                pos += fcsSize;
                if (single) windowSize = contentSize;
                if (!err && contentSize > cap) err = KZE_DSTSMALL;
            }
        }
    }

    // ---- blocks -----------------------------------------------------------
    u32 op = 0;                       // bytes produced
    u32 rep1 = 1, rep2 = 4, rep3 = 8;
    u32 hufLog = 0; bool hufValid = false;
    u32 tlLL = 0, tlOF = 0, tlML = 0; bool vLL = false, vOF = false, vML = false;
    bool last = false;
    while (!err && !last) {
        if (pos + 3 > srcSize) { err = KZE_SRCSIZE; break; }
        u32 const bh = (u32)src[pos] | ((u32)src[pos + 1] << 8) | ((u32)src[pos + 2] << 16);
        last = bh & 1; u32 const btype = (bh >> 1) & 3; u32 const bsize = bh >> 3;
        pos += 3;
        if (btype == 3) { err = KZE_CORRUPT; break; }
        if (btype == 0) {
            if (pos + bsize > srcSize) { err = KZE_SRCSIZE; break; }
            if (op + bsize > cap) { err = KZE_DSTSMALL; break; }
            kxd_wave_copy(dst + op, src + pos, bsize, lane);
            op += bsize; pos += bsize;
            continue;
        }
        if (btype == 1) {
            if (pos + 1 > srcSize) { err = KZE_SRCSIZE; break; }
            if (op + bsize > cap) { err = KZE_DSTSMALL; break; }
            u8 const v = src[pos];
            for (u32 i = (u32)lane; i < bsize; i += 64) dst[op + i] = v;
            op += bsize; pos += 1;
            continue;
        }
        // ---- compressed block --------------------------------------------
        if (bsize > 128u * 1024u || pos + bsize > srcSize || bsize < 2) { err = KZE_CORRUPT; break; }
        const u8* const bp = src + pos; u32 const bend = bsize;
        // literals section header
        u32 const lh0 = bp[0]; u32 const ltype = lh0 & 3, sf = (lh0 >> 2) & 3;
        u32 lhSize, regen, comp = 0, nstreams = 1;
        if (ltype < 2) {
            if (sf == 0 || sf == 2) { lhSize = 1; regen = lh0 >> 3; }
            else if (sf == 1) { lhSize = 2; regen = kx_ld16(bp) >> 4; }
            else { lhSize = 3; regen = ((u32)bp[0] | ((u32)bp[1] << 8) | ((u32)bp[2] << 16)) >> 4; }
        } else {
            if (bend < 5) { err = KZE_CORRUPT; break; }
            u32 const w = kx_ld32(bp);
            if (sf < 2) { lhSize = 3; regen = (w >> 4) & 0x3FF; comp = (w >> 14) & 0x3FF; nstreams = sf ? 4 : 1; }
            else if (sf == 2) { lhSize = 4; regen = (w >> 4) & 0x3FFF; comp = w >> 18; nstreams = 4; }
            else { lhSize = 5; regen = (w >> 4) & 0x3FFFF; comp = (w >> 22) + ((u32)bp[4] << 10); nstreams = 4; }
        }
        if (regen > 128u * 1024u || regen > a.lit_cap) { err = KZE_CORRUPT; break; }
        const u8* litPtr = lits; u32 lpos = lhSize;
        if (ltype == 0) {
            if (lpos + regen > bend) { err = KZE_CORRUPT; break; }
            litPtr = bp + lpos; lpos += regen;
        } else if (ltype == 1) {
            if (lpos + 1 > bend) { err = KZE_CORRUPT; break; }
            u8 const v = bp[lpos];
            for (u32 i = (u32)lane; i < regen; i += 64) lits[i] = v;
            lpos += 1;
        } else {
            if (lpos + comp > bend || comp == 0) { err = KZE_CORRUPT; break; }
            u32 hused = 0;
            if (ltype == 2) {
                u32 r = 0;
                if (lane == 0) { u32 tl = 0; r = khuf_read_dtable(lds, bp + lpos, comp, &tl); lds.bc[0] = r; lds.bc[1] = tl; }
                kx_sync();
                r = lds.bc[0];
                if (r == 0) { err = KZE_CORRUPT; break; }
                hused = r; hufLog = lds.bc[1]; hufValid = true;
            } else if (!hufValid) { err = KZE_CORRUPT; break; }
            const u8* const sp = bp + lpos + hused; u32 const ssize = comp - hused;
            bool ok = true;
            if (nstreams == 1) {
                if (lane == 0) ok = khuf_decode_stream(lds, hufLog, sp, ssize, lits, regen);
            } else {
                if (ssize < 10) { err = KZE_CORRUPT; break; }
                u32 const c0 = kx_ld16(sp), c1 = kx_ld16(sp + 2), c2 = kx_ld16(sp + 4);
                if (6 + c0 + c1 + c2 > ssize) { err = KZE_CORRUPT; break; }
                u32 const c3 = ssize - 6 - c0 - c1 - c2;
                u32 const seg = (regen + 3) / 4;
                if (3 * seg > regen) { err = KZE_CORRUPT; break; }
                if (lane < 4) {
                    u32 const so = 6 + (lane > 0 ? c0 : 0) + (lane > 1 ? c1 : 0) + (lane > 2 ? c2 : 0);
                    u32 const sz = lane == 0 ? c0 : lane == 1 ? c1 : lane == 2 ? c2 : c3;
                    u32 const cnt = lane < 3 ? seg : regen - 3 * seg;
                    ok = khuf_decode_stream(lds, hufLog, sp + so, sz, lits + (u32)lane * seg, cnt);
                }
            }
            if (kx_any(!ok)) { err = KZE_CORRUPT; break; }
            lpos += comp;
            kx_sync();
        }
        kx_sync();
        // sequences header + tables (lane 0)
        if (lane == 0) {
            u32 e = 0, nbSeq = 0, p2 = lpos;
            if (p2 >= bend) e = KZE_CORRUPT;
            if (!e) {
                u32 const b0 = bp[p2++];
                if (b0 < 128) nbSeq = b0;
                else if (b0 < 255) { if (p2 >= bend) e = KZE_CORRUPT; else nbSeq = ((b0 - 128) << 8) + bp[p2++]; }
                else { if (p2 + 2 > bend) e = KZE_CORRUPT; else { nbSeq = kx_ld16(bp + p2) + 0x7F00; p2 += 2; } }
            }
            if (!e && nbSeq) {
                if (p2 >= bend) e = KZE_CORRUPT;
                else {
                    u32 const modes = bp[p2++];
                    if (modes & 3) e = KZE_CORRUPT;
                    u32 r;
                    if (!e) { r = kxd_seq_table(lds, 0, modes >> 6, bp + p2, bend - p2, &tlLL, &vLL); if (r == KXD_FAIL) e = KZE_CORRUPT; else p2 += r; }
                    if (!e) { r = kxd_seq_table(lds, 1, (modes >> 4) & 3, bp + p2, bend - p2, &tlOF, &vOF); if (r == KXD_FAIL) e = KZE_CORRUPT; else p2 += r; }
                    if (!e) { r = kxd_seq_table(lds, 2, (modes >> 2) & 3, bp + p2, bend - p2, &tlML, &vML); if (r == KXD_FAIL) e = KZE_CORRUPT; else p2 += r; }
                    if (!e && p2 >= bend) e = KZE_CORRUPT;
                }
            }
            lds.bc[0] = e; lds.bc[1] = nbSeq; lds.bc[2] = p2;
        }
        kx_sync();
        if (lds.bc[0]) { err = lds.bc[0]; break; }
        u32 const nbSeq = lds.bc[1]; u32 const spos = lds.bc[2];
        // lane 0 keeps the table logs / validity for later blocks; share them
        tlLL = kx_shfl(tlLL, 0); tlOF = kx_shfl(tlOF, 0); tlML = kx_shfl(tlML, 0);
        vLL = kx_shfl((u32)vLL, 0) != 0; vOF = kx_shfl((u32)vOF, 0) != 0; vML = kx_shfl((u32)vML, 0) != 0;
        u32 litUsed = 0;
        if (nbSeq) {
            KBackBits b; u32 sLL = 0, sOF = 0, sML = 0; bool bad = false;
            if (lane == 0) {
                if (!kbb_init(b, bp + spos, bend - spos)) bad = true;
                else if (b.bits < (int)(tlLL + tlOF + tlML)) bad = true;
                else {
                    sLL = kbb_peek(b, tlLL); b.bits -= (int)tlLL;
                    sOF = kbb_peek(b, tlOF); b.bits -= (int)tlOF;
                    sML = kbb_peek(b, tlML); b.bits -= (int)tlML;
                }
            }
            for (u32 done = 0; done < nbSeq && !err; ) {
                u32 const cnt = (nbSeq - done) < 64 ? (nbSeq - done) : 64;
                if (lane == 0) {
                    u32 i = 0;
                    for (; i < cnt && !bad; i++) {
                        u32 const eLL = lds.fse[0][sLL], eOF = lds.fse[1][sOF], eML = lds.fse[2][sML];
                        u32 const llc = eLL >> 24, ofc = eOF >> 24, mlc = eML >> 24;
                        if (ofc > 31 || llc > 35 || mlc > 52) { bad = true; break; }
                        u32 const llb = kxd_ll_bits(llc), mlb = kxd_ml_bits(mlc);
                        if (b.bits < (int)(ofc + mlb + llb)) { bad = true; break; }
                        u32 const ofv = (1u << ofc) + kbb_peek(b, ofc); b.bits -= (int)ofc;
                        u32 const ml = kx_ml_base(mlc) + kbb_peek(b, mlb); b.bits -= (int)mlb;
                        u32 const ll = kx_ll_base(llc) + kbb_peek(b, llb); b.bits -= (int)llb;
                        u32 off;
                        if (ofv > 3) { off = ofv - 3; rep3 = rep2; rep2 = rep1; rep1 = off; }
                        else {
                            u32 const idx = ofv - 1 + (ll == 0);
                            if (idx == 0) off = rep1;
                            else {
                                off = idx == 1 ? rep2 : idx == 2 ? rep3 : rep1 - 1;
                                if (off == 0) off = 1;
                                if (idx != 1) rep3 = rep2;
                                rep2 = rep1; rep1 = off;
                            }
                        }
                        lds.stage[3 * i] = ll; lds.stage[3 * i + 1] = ml; lds.stage[3 * i + 2] = off;
                        if (done + i + 1 < nbSeq) {
                            u32 const nLL = (eLL >> 16) & 0xFF, nML = (eML >> 16) & 0xFF, nOF = (eOF >> 16) & 0xFF;
                            if (b.bits < (int)(nLL + nML + nOF)) { bad = true; break; }
                            sLL = (eLL & 0xFFFFu) + kbb_peek(b, nLL); b.bits -= (int)nLL;
                            sML = (eML & 0xFFFFu) + kbb_peek(b, nML); b.bits -= (int)nML;
                            sOF = (eOF & 0xFFFFu) + kbb_peek(b, nOF); b.bits -= (int)nOF;
                        } else if (b.bits != 0) bad = true;
                    }
                    lds.stage[192] = bad ? 1u : 0u;
                }
                kx_sync();
                if (lds.stage[192]) { err = KZE_CORRUPT; break; }
                // execute the staged sequences, all lanes
                for (u32 i = 0; i < cnt; i++) {
                    u32 const ll = lds.stage[3 * i], ml = lds.stage[3 * i + 1], off = lds.stage[3 * i + 2];
                    if (litUsed + ll > regen) { err = KZE_CORRUPT; break; }
                    if ((u64)op + ll + ml > cap) { err = KZE_DSTSMALL; break; }
                    if (off > op + ll) { err = KZE_CORRUPT; break; }
                    kxd_wave_copy(dst + op, litPtr + litUsed, ll, lane);
                    op += ll; litUsed += ll;
                    kx_lockstep();
                    const u8* const ms = dst + op - off;
                    if (off >= 64) {
                        for (u32 base = 0; base < ml; base += 64) {
                            u32 const k = base + (u32)lane;
                            if (k < ml) dst[op + k] = ms[k];
                            kx_lockstep();
                        }
                    } else {
                        // overlapping copy: replicate the off-byte pattern, a multiple of off per step
                        u32 const chunk = (64 / off) * off;
                        u32 const m = (u32)lane % off;
                        for (u32 base = 0; base < ml; base += chunk) {
                            u32 const k = base + (u32)lane;
                            u8 v = 0;
                            if ((u32)lane < chunk && k < ml) v = ms[m];
                            if ((u32)lane < chunk && k < ml) dst[op + k] = v;
                        }
                        kx_lockstep();
                    }
                    op += ml;
                }
                kx_sync();
                done += cnt;
            }
            if (err) break;
        }
        // remaining literals
        if (litUsed > regen || (u64)op + (regen - litUsed) > cap) { err = (litUsed > regen) ? KZE_CORRUPT : KZE_DSTSMALL; break; }
        kxd_wave_copy(dst + op, litPtr + litUsed, regen - litUsed, lane);
        op += regen - litUsed;
        pos += bsize;
        kx_sync();
    }
    kx_sync();
    if (!err && hasContent && contentSize != op) err = KZE_CORRUPT;
    if (!err && checksum) {
        if (pos + 4 > srcSize) err = KZE_SRCSIZE;
        else {
            u32 bad = 0;
            if (lane == 0) bad = ((u32)kxxh64(dst, op) != kx_ld32(src + pos)) ? 1u : 0u;
            bad = kx_shfl(bad, 0);
            if (bad) err = KZE_CHECKSUM;
        }
    }
    if (lane == 0) { a.status[f] = err; a.out_len[f] = err ? 0u : op; }
}

KX_DEV void zstd_decode_body(const KDecodeArgs& a)
{
    KX_SHARED KDecodeLds lds;
    int const lane = kx_lane();
    for (u32 f = kx_block(); f < a.n_slices; f += kx_nblocks()) {
        zstd_decode_frame(a, lds, f, lane);
        kx_sync();
    }
}
