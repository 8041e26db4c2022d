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

// What k_zstd_seq_predecode (zstd_predecode.h: one lane per frame) leaves for a compressed block of an entry's first
// frame: where its decoded sequences (litLength, matchLength, offset) start in the entry's staging area, how many, whether
// they are there at all, and the repeat offsets after the block.
struct KPreBlk { u32 seq_off; u32 nbSeq; u32 ok; u32 rep[3]; u32 pad[2]; };
// ... and k_zstd_lit_predecode (one lane per Huffman stream): where the block's decoded literals start in the entry's
// literal staging area, how many, whether they are there
struct KPreLit { u32 off; u32 regen; u32 ok; u32 pad; };

struct KDecodeArgs {
    const u64* pre_stage; u32 pre_seq_cap; const KPreBlk* pre_blk; u32 pre_blk_cap; const u32* pre_nblk;   // null / 0: nothing pre-decoded
    const u8* pre_lits; u32 pre_lit_cap; const KPreLit* pre_lit; const u32* pre_nlit;                      // (records: pre_blk_cap per entry)
    const u8* src; const u64* in_off; const u32* in_len; u32 n_slices;
    u8* dst; const u64* out_off; const u32* out_cap; u32* out_len; u32* status;
    u8* lits; u32 lit_cap;                 // per frame: decoded literals of one block
    u32 flags;                             // timing-only ablations (results wrong): 1 skip Huffman walk, 2 skip sequences, 4 skip copies;
                                           //   window executor: 8 skip the flush, 16 short literals, 32 short matches from HBM, 64 all matches
    const u8* dict; u32 dict_size;         // dictionary content shared by the batch (history before every frame), or null / 0
    const KDictDPrior* dprior = nullptr;   // a formatted dictionary's tables and repeat offsets (dict is then its content part), or null
    u32 dict_id = 0;                       // the ID a frame's header may name (0: a raw-content dictionary, or none: any ID in a header is a mismatch)
};

enum { KZE_GENERIC = 1, KZE_PREFIX = 10, KZE_FRAMEPARAM = 14, KZE_WINDOW = 16, KZE_CORRUPT = 20, KZE_CHECKSUM = 22,
       KZE_LITHDR = 24, KZE_DICT = 32, KZE_WORKSPACE = 66, KZE_DSTSMALL = 70, KZE_SRCSIZE = 72 };

#define KXD_WIN 5120u
struct KDecodeLds {
    union {                     // phase-shared region: the phases of a block never overlap in time
        u16 huf[2048];          // literal phase: Huffman decoding table (depth <= 11, RFC 8878): symbol | nbBits << 8
        struct {                // Huffman table description: the weights' FSE table (tableLog <= 6) + spread scratch
            u16 wb[64]; u8 wc[64]; u8 tsym[64];
        } b;
        struct {                // sequence phase
            u16 fb[1280];       // FSE decoding tables LL [0,512) ML [512,1024) OF [1024,1280): newStateBase | nbBits << 12
            u8 fc[1280];        //   ... and the symbol (code) of each state
            u32 sbuf[258];      // staged words of the sequence bitstream (3 words below the first, zero at the stream start, + 254
                                //   + 1 spare); also the spread scratch
            u32 stage[194];     // 64 x (litLength, matchLength, offset) + error flag
        } q;
        struct {                // sequence phase of a block whose sequences were decoded ahead (no tables, no bitstream):
            u8 win[KXD_WIN];    // the newest output, [winBase, op + the chunk in the making): matches are resolved here
        } w;
    } u;
    u32 llx[36];                // per code: baseValue | extraBits << 24
    u32 mlx[53];
    short norm[64];
    u16 symnext[64];
    u8 weights[256];            // kept so a tree-less block can rebuild the Huffman table
    // what a later block's "repeat" mode needs to rebuild a sequence table (they share LDS with the Huffman table):
    short keepNorm[3][64]; u32 keepKind[3], keepLog[3], keepMax[3];      // kind 0 none, 1 RLE (symbol in keepMax), 2 FSE counts
    u32 rank[16];
    u32 bc[16];                 // lane 0 -> wave broadcast slots
};
KX_SHARED KDecodeLds g_kxd_lds;        // k_zstd_decode's workgroup (one wave) state; file scope so that a function called from the kernel can name it

enum { KXD_LL0 = 0, KXD_ML0 = 512, KXD_OF0 = 1024 };

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

// FSE decoding table: db[state] = newStateBase | nbBits << 12, dc[state] = symbol
KX_DEV void kfse_build_dtable(u16* db, u8* dc, const short* norm, u32 maxSymbolValue, u32 tableLog, u16* symnext, u8* tsym)
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
        db[u] = (u16)((((next << nb) - tableSize) & 0xFFFu) | (nb << 12)); dc[u] = (u8)s;
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
// weights[0..nw) -> decoding table (lane 0). false on an invalid weight set
template <class LDS>
KX_DEV bool khuf_fill_dtable(LDS& lds, u32 nw, u32 tableLog)
{
    for (u32 i = 0; i < 16; i++) lds.rank[i] = 0;
    for (u32 i = 0; i < nw; i++) lds.rank[lds.weights[i]]++;
    if (lds.rank[1] < 2 || (lds.rank[1] & 1)) return false;
    u32 next = 0;
    for (u32 w = 1; w <= tableLog; w++) { u32 const cur = next; next += lds.rank[w] << (w - 1); lds.rank[w] = cur; }
    for (u32 s = 0; s < nw; s++) {
        u32 const w = lds.weights[s];
        if (!w) continue;
        u32 const len = 1u << (w - 1); u32 const start = lds.rank[w]; u32 const nb = tableLog + 1 - w;
        for (u32 i = 0; i < len; i++) lds.u.huf[start + i] = (u16)(s | (nb << 8));
        lds.rank[w] += len;
    }
    return true;
}

// (FILL = false: the weights and the table log only; the caller has its own table layout)
template <class LDS, bool FILL = true>
KX_DEV u32 khuf_read_dtable(LDS& lds, const u8* p, u32 size, u32* tableLogOut, u32* nwOut)
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
        kfse_build_dtable(lds.u.b.wb, lds.u.b.wc, lds.norm, maxSV, tl, lds.symnext, lds.u.b.tsym);
        KBackBits b;
        if (!kbb_init(b, p + 1 + h, hb - h)) return 0;
        if (b.bits < (int)(2 * tl)) return 0;
        u32 s1 = kbb_peek(b, tl); b.bits -= (int)tl;
        u32 s2 = kbb_peek(b, tl); b.bits -= (int)tl;
        for (;;) {
            if (nw >= 255) return 0;
            u32 const e1 = lds.u.b.wb[s1];
            lds.weights[nw++] = lds.u.b.wc[s1];
            u32 const nb1 = e1 >> 12;
            if (b.bits < (int)nb1) { if (nw >= 255) return 0; lds.weights[nw++] = lds.u.b.wc[s2]; break; }
            s1 = (e1 & 0xFFFu) + kbb_peek(b, nb1); b.bits -= (int)nb1;
            if (nw >= 255) return 0;
            u32 const e2 = lds.u.b.wb[s2];
            lds.weights[nw++] = lds.u.b.wc[s2];
            u32 const nb2 = e2 >> 12;
            if (b.bits < (int)nb2) { if (nw >= 255) return 0; lds.weights[nw++] = lds.u.b.wc[s1]; break; }
            s2 = (e2 & 0xFFFu) + kbb_peek(b, nb2); b.bits -= (int)nb2;
        }
    }
    // last weight is implied: total must complete a power of two
    u32 total = 0;
    for (u32 i = 0; i < nw; i++) { u32 const w = lds.weights[i]; if (w > 12) return 0; if (w) total += 1u << (w - 1); }
    if (total == 0) return 0;
    u32 const tableLog = kx_hb32(total) + 1;
    if (tableLog > 11) return 0;
    u32 const rest = (1u << tableLog) - total;
    if (rest & (rest - 1)) return 0;              // must be a power of two
    lds.weights[nw] = (u8)(kx_hb32(rest) + 1); nw++;
    if constexpr (FILL) { if (!khuf_fill_dtable(lds, nw, tableLog)) return 0; }
    *tableLogOut = tableLog; *nwOut = nw;
    return used;
}

// One lane decodes one Huffman stream of `count` symbols; returns false on corruption.
// The unread bits sit at the top of a 64-bit container (hi : lo) held as two 32-bit halves -- a symbol is one shift of
// hi, one LDS look-up and a funnel shift -- and 32 more come in whenever 32 or fewer are left, from a word that was
// requested two refills earlier (the stream is cut into 32-bit words counted from its first byte; below it: zeros).
// Four symbols and two refill points per trip, the symbols leave as one word.  Reading past the stream's first bit
// is harmless (the trip count is the symbol count, table indices are tableLog bits) and shows in the final test:
// the stream must have been consumed to exactly its first bit.
template <class LDS>
KX_DEV bool khuf_decode_stream(const LDS& lds, u32 tableLog, const u8* p, u32 size, u8* out, u32 count)
{
    if (size == 0) return false;
    u32 const lastByte = p[size - 1];
    if (lastByte == 0) return false;
    u32 const totalBits = 8 * (size - 1) + kx_hb32(lastByte);
    int j = (int)((size - 1) >> 2);                                // the top word (1 to 4 bytes of it exist)
    u32 wt = 0;
    for (u32 k = 4u * (u32)j; k < size; k++) wt |= (u32)p[k] << (8u * (k - 4u * (u32)j));
    u32 const vb = totalBits - 32u * (u32)j;                       // its bits below the end mark: 0..31
    u32 hi = vb ? wt << (32u - vb) : 0u, lo = 0, avail = vb;
    u32 w1 = j >= 1 ? kx_ld32(p + 4 * (j - 1)) : 0u, w2 = j >= 2 ? kx_ld32(p + 4 * (j - 2)) : 0u;
    j -= 2;                                                        // the word w2 holds
    u32 consumed = 0; u32 const sh = 32u - tableLog;
#define KHUF_REFILL if (avail <= 32u) { u64 const c_ = (((u64)hi << 32) | lo) | ((u64)w1 << (32u - avail)); hi = (u32)(c_ >> 32); lo = (u32)c_; \
                                        avail += 32u; w1 = w2; j--; w2 = j >= 0 ? kx_ld32(p + 4 * j) : 0u; }
#define KHUF_SYM(k_) { u32 const e_ = lds.u.huf[hi >> sh]; u32 const nb_ = e_ >> 8; hi = kx_alignbit(hi, lo, 32u - nb_); lo <<= nb_; \
                       avail -= nb_; consumed += nb_; acc |= (e_ & 0xFFu) << (8 * (k_)); }
    u32 i = 0;
    for (; i + 4 <= count; i += 4) {
        u32 acc = 0;
        KHUF_REFILL KHUF_SYM(0) KHUF_SYM(1) KHUF_REFILL KHUF_SYM(2) KHUF_SYM(3)
        kx_st32(out + i, acc);
    }
    for (; i < count; i++) { u32 acc = 0; KHUF_REFILL KHUF_SYM(0) out[i] = (u8)acc; }
#undef KHUF_SYM
#undef KHUF_REFILL
    return consumed == totalBits;
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

KX_DEV int kxd_seq_base(int t) { return t == 0 ? KXD_LL0 : t == 1 ? KXD_OF0 : KXD_ML0; }

// lane 0: read one symbol type's table description and build its decoding table in lds.u.q (the region is shared
// with the Huffman table, so every block builds its tables; "repeat" rebuilds from what the last table was made of).
// Returns bytes consumed, KXD_FAIL on error.
#define KXD_FAIL 0xFFFFFFFFu
KX_DEV u32 kxd_seq_table(KDecodeLds& lds, int t, u32 mode, const u8* p, u32 size, u32* tableLog, bool build)
{
    static const short LL_defaultNorm[36] = { 4,3,2,2,2,2,2,2, 2,2,2,2,2,1,1,1, 2,2,2,2,2,2,2,2, 2,3,2,1,1,1,1,1, -1,-1,-1,-1 };
    static const short ML_defaultNorm[53] = { 1,4,3,2,2,2,2,2, 2,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1,
                                              1,1,1,1,1,1,1,1, 1,1,1,1,1,1,-1,-1, -1,-1,-1,-1,-1 };
    static const short OF_defaultNorm[29] = { 1,1,1,1,1,1,2,2, 2,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1, -1,-1,-1,-1,-1 };
    u32 const maxSym = (t == 0) ? 35 : (t == 1) ? 31 : 52;
    u32 const maxLog = (t == 0) ? 9 : (t == 1) ? 8 : 9;
    u16* const db = lds.u.q.fb + kxd_seq_base(t); u8* const dc = lds.u.q.fc + kxd_seq_base(t);
    u8* const spread = (u8*)lds.u.q.sbuf;               // free until the bitstream is staged
    u32 used = 0;
    if (mode == 0) {
        const short* dn = (t == 0) ? LL_defaultNorm : (t == 1) ? OF_defaultNorm : ML_defaultNorm;
        u32 const dmax = (t == 0) ? 35 : (t == 1) ? 28 : 52; u32 const dlog = (t == 1) ? 5 : 6;
        for (u32 s = 0; s <= dmax; s++) lds.keepNorm[t][s] = dn[s];
        lds.keepKind[t] = 2; lds.keepLog[t] = dlog; lds.keepMax[t] = dmax;
    } else if (mode == 1) {
        if (size < 1 || p[0] > maxSym) return KXD_FAIL;
        lds.keepKind[t] = 1; lds.keepLog[t] = 0; lds.keepMax[t] = p[0];
        used = 1;
    } else if (mode == 2) {
        u32 maxSV = maxSym, tl = 0;
        u32 const h = kfse_read_ncount(lds.keepNorm[t], &maxSV, &tl, p, size, maxLog);
        if (h == 0) { lds.keepKind[t] = 0; return KXD_FAIL; }
        lds.keepKind[t] = 2; lds.keepLog[t] = tl; lds.keepMax[t] = maxSV;
        used = h;
    } else if (lds.keepKind[t] == 0) return KXD_FAIL;    // repeat without a previous table
    if (!build) { }                                     // the block's sequences are decoded already: the description is kept for a later "repeat"
    else if (lds.keepKind[t] == 1) { db[0] = 0; dc[0] = (u8)lds.keepMax[t]; }          // nbBits 0, next state 0
    else kfse_build_dtable(db, dc, lds.keepNorm[t], lds.keepMax[t], lds.keepLog[t], lds.symnext, spread);
    *tableLog = lds.keepLog[t];
    return used;
}

KX_DEV void kxd_wave_copy(u8* dst, const u8* src, u32 n, int lane)
{
    u32 i = (u32)lane * 8u;
    for (; i + 8 <= n; i += 512u) kx_st64(dst + i, kx_ld64(src + i));
    u32 const tail = n & ~7u;
    if (lane < (int)(n - tail)) dst[tail + lane] = src[tail + lane];
}

// Exactly n (<= 32) bytes from s to d, the two ranges disjoint.  Every load is issued before the first store, so the
// copy costs one memory latency whatever n is (a byte loop costs one per step: the compiler must assume s and d alias).
// The pieces overlap instead of shrinking: [0,8) [8,16) [16,24) and the last eight bytes; two words below eight bytes;
// first, middle and last byte below four.
KX_DEV void kxd_copy32(u8* d, const u8* s, u32 n)
{
    if (n >= 8) {
        u32 const o1 = n > 16 ? 8u : 0u, o2 = n > 24 ? 16u : 0u;
        u64 const v0 = kx_ld64(s), v1 = kx_ld64(s + o1), v2 = kx_ld64(s + o2), vt = kx_ld64(s + n - 8);
        kx_st64(d, v0);
        if (n > 16) kx_st64(d + 8, v1);
        if (n > 24) kx_st64(d + 16, v2);
        kx_st64(d + n - 8, vt);
    } else if (n >= 4) {
        u32 const v0 = kx_ld32(s), vt = kx_ld32(s + n - 4);
        kx_st32(d, v0); kx_st32(d + n - 4, vt);
    } else if (n) {
        u8 const b0 = s[0], b1 = s[n >> 1], b2 = s[n - 1];
        d[0] = b0; d[n >> 1] = b1; d[n - 1] = b2;
    }
}

// kxd_copy32 in two halves, so that several copies' loads are in flight before the first store waits for one of them
struct KxdRun32 { u64 v0, v1, v2, vt; };
KX_DEV void kxd_load32(KxdRun32& r, const u8* s, u32 n)
{
    r.v0 = 0; r.v1 = 0; r.v2 = 0; r.vt = 0;
    if (n >= 8) {
        r.v0 = kx_ld64(s); r.vt = kx_ld64(s + n - 8);
        if (n > 16) { r.v1 = kx_ld64(s + 8); if (n > 24) r.v2 = kx_ld64(s + 16); }
    } else if (n >= 4) { r.v0 = kx_ld32(s); r.vt = kx_ld32(s + n - 4); }
    else if (n) { r.v0 = s[0]; r.v1 = s[n >> 1]; r.vt = s[n - 1]; }
}
KX_DEV void kxd_store32(u8* d, const KxdRun32& r, u32 n)
{
    if (n >= 8) {
        kx_st64(d, r.v0);
        if (n > 16) kx_st64(d + 8, r.v1);
        if (n > 24) kx_st64(d + 16, r.v2);
        kx_st64(d + n - 8, r.vt);
    } else if (n >= 4) { kx_st32(d, (u32)r.v0); kx_st32(d + n - 4, (u32)r.vt); }
    else if (n) { d[0] = (u8)r.v0; d[n >> 1] = (u8)r.v1; d[n - 1] = (u8)r.vt; }
}

// One byte of the frame's output (or of the dictionary before it) while a chunk is being put together: absolute
// position q below the write front.  What lies at or above winBase is in the LDS window, everything below the chunk's
// first byte has been flushed to dst; q < fbase is dictionary content.
KX_DEV u8 kxd_hist_byte(const u8* dict, u32 dict_size, const u8* win, u32 winBase, const u8* dst, u32 q, bool inDictRange, u32 dictBack)
{
    if (inDictRange) return dict[dict_size - dictBack];
    return q >= winBase ? win[q - winBase] : dst[q];
}

// ---- the sequences of one block, decoded ahead of this kernel, executed through an LDS window ----------------------
// Executing sequences against dst in HBM costs the texture addresser one slot per lane and instruction (every lane
// copies a few bytes somewhere else) and a trip to L2 per dependency round.  Here a chunk of up to 64 sequences is put
// together in LDS -- literals and far matches come in from HBM, near matches (source inside the window) are LDS to LDS
// copies, and the rounds that order dependent matches cost LDS latency -- and goes out to dst as one contiguous run.
// The window keeps the chunks already written as history while they fit; a chunk that does not fit behind them
// starts the window over, and a single sequence larger than the window is copied HBM to HBM by the whole wave.
// Same checks and error codes as the loop in zstd_decode_frame.
// Everything that is the same in every lane is taken with kx_bcast (v_readlane: the result lives in a scalar
// register), never kx_shfl: positions, sizes and with them every address stay scalar base + 32-bit lane offset.
struct KxdExecResult { u32 err, op, litUsed; };
#define KXD_EXEC_FAIL(e_) { KxdExecResult r_; r_.err = (e_); r_.op = op; r_.litUsed = litUsed; return r_; }
KX_DEV KxdExecResult kxd_exec_window(const u8* dict, u32 dict_size, u32 flags, const u64* preSeq, u32 nbSeq,
                                              u8* dst, u32 fbase, u32 cap, const u8* litPtr, u32 regen, u32 op)
{
    int const lane = kx_lane();
    u8* const win = g_kxd_lds.u.w.win;
    u32 litUsed = 0, winBase = op;
    u32 heldAt = 0xFFFFFFFFu, hLL = 0, hML = 0, hOF = 1;            // lane i holds sequence heldAt + i
    for (u32 done = 0; done < nbSeq; ) {
        u32 cnt = (nbSeq - done) < 64 ? (nbSeq - done) : 64;
        if (heldAt != done) {
            u32 const ix = done + (u32)lane;
            if (ix < nbSeq) { u64 const v = preSeq[ix]; hLL = (u32)v & 0xFFFFu; hML = (((u32)v >> 16) & 0xFFFFu) + 3u; hOF = (u32)(v >> 32); }
            heldAt = done;
        }
        u32 pLL = 0, pML = 0, pOF = 1;                                // the next chunk's, requested now
        { u32 const nx = done + 64u + (u32)lane; if (nx < nbSeq) { u64 const v = preSeq[nx]; pLL = (u32)v & 0xFFFFu; pML = (((u32)v >> 16) & 0xFFFFu) + 3u; pOF = (u32)(v >> 32); } }
        bool own = (u32)lane < cnt;
        u32 ll = own ? hLL : 0u, ml = own ? hML : 0u, off = own ? hOF : 1u;
        u32 sl = ll, st = ll + ml;                    // inclusive scans over the lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            u32 const a1 = kx_shfl(sl, lane - o), a2 = kx_shfl(st, lane - o);
            if (lane >= o) { sl += a1; st += a2; }
        }
        u32 const fit = kx_popc64(kx_ballot(own && st <= KXD_WIN));
        if (fit == 0) {
            // the chunk's first sequence alone is larger than the window: HBM to HBM, by the whole wave
            u32 const llE = kx_bcast(ll, 0), mlE = kx_bcast(ml, 0), offE = kx_bcast(off, 0);
            if (litUsed + llE > regen) KXD_EXEC_FAIL(KZE_CORRUPT)
            if ((u64)op + llE + mlE > cap) KXD_EXEC_FAIL(KZE_DSTSMALL)
            u32 const dE = op + llE;
            if (offE > dE - fbase + dict_size) KXD_EXEC_FAIL(KZE_CORRUPT)
            if (!(flags & 4u)) {
                kxd_wave_copy(dst + op, litPtr + litUsed, llE, lane);
                kx_lockstep();
                u32 const inDict = offE > dE - fbase ? offE - (dE - fbase) : 0u;
                const u8* const dsrc = dict + (dict_size - inDict);
                if (offE >= 64) {
                    for (u32 base = 0; base < mlE; base += 64) {
                        u32 const k = base + (u32)lane;
                        if (k < mlE) dst[dE + k] = (k < inDict) ? dsrc[k] : dst[dE + k - offE];
                        kx_lockstep();
                    }
                } else if (inDict) {
                    if (lane == 0) for (u32 k = 0; k < mlE; k++) dst[dE + k] = (k < inDict) ? dsrc[k] : dst[dE + k - offE];
                } else {
                    u32 const chunk = (64 / offE) * offE; u32 const m = (u32)lane % offE;
                    u8 v = 0;
                    if ((u32)lane < chunk) v = dst[dE - offE + m];
                    for (u32 base = 0; base < mlE; base += chunk) {
                        u32 const k = base + (u32)lane;
                        if ((u32)lane < chunk && k < mlE) dst[dE + k] = v;
                    }
                }
                kx_lockstep();
            }
            op += llE + mlE; litUsed += llE; winBase = op;
            done += 1;
            continue;
        }
        if (fit < cnt) { cnt = fit; own = (u32)lane < cnt; if (!own) { ll = 0; ml = 0; off = 1; } }
        u32 const totLit = kx_bcast(sl, (int)cnt - 1), totOut = kx_bcast(st, (int)cnt - 1);
        if (litUsed + totLit > regen) KXD_EXEC_FAIL(KZE_CORRUPT)
        if ((u64)op + totOut > cap) KXD_EXEC_FAIL(KZE_DSTSMALL)
        u32 const lp = litUsed + (sl - ll);           // my literals in the literal buffer
        u32 const dlit = op + (st - ll - ml);         // where they go
        u32 const dmat = dlit + ll;                   // where my match goes
        // a match may start inside the dictionary (the history before the frame's first byte)
        if (kx_any(own && off > dmat - fbase + dict_size)) KXD_EXEC_FAIL(KZE_CORRUPT)
        if (!(flags & 4u)) {
            if ((op - winBase) + totOut > KXD_WIN) winBase = op;      // no room behind the history: start over
            // ---- everything that depends on nothing in this chunk, with every HBM load in flight at once ----
            // literals (from the literal buffer) and "early" matches, whose whole source lies in earlier chunks' output
            // (below op): in the window if it reaches back that far, else in dst.
            u32 const srcPos = dmat - off;                                // meaningful when the match starts inside the frame
            bool const inDictL = off > dmat - fbase;
            bool const inWin = !inDictL && srcPos >= winBase;             // the source lies in the window
            bool const early = own && !inDictL && srcPos + ml <= op;
            u8* const dl_ = win + (dlit - winBase); u8* const dm_ = win + (dmat - winBase);
            u64 const longFar = kx_ballot(early && ml > 32 && !inWin);
            // (the first long match from dst: its first 512 bytes travel with the rest)
            int const lf = longFar ? (int)kx_ctz64(longFar) : 0;
            u32 const lfN = longFar ? kx_bcast(ml, lf) : 0u, lfS = kx_bcast(srcPos, lf), lfD = kx_bcast(dmat, lf) - winBase;
            u64 lfV = 0; u8 lfB = 0;
            if ((u32)lane * 8u + 8u <= lfN) lfV = kx_ld64(dst + lfS + (u32)lane * 8u);
            if ((lfN & ~7u) < 512u && (u32)lane < (lfN & 7u)) lfB = dst[lfS + (lfN & ~7u) + (u32)lane];
            KxdRun32 rl, rm;
            u32 const nl = (own && ll <= 32 && !(flags & 16u)) ? ll : 0u;
            u32 const nm = (early && ml <= 32 && !inWin && !(flags & 32u)) ? ml : 0u;
            kxd_load32(rl, litPtr + lp, nl);
            kxd_load32(rm, dst + srcPos, nm);
            if (early && ml <= 32 && inWin) kxd_copy32(dm_, win + (srcPos - winBase), ml);       // (LDS to LDS: disjoint, srcPos + ml <= op <= dmat)
            kxd_store32(dl_, rl, nl);
            kxd_store32(dm_, rm, nm);
            if ((u32)lane * 8u + 8u <= lfN) kx_st64(win + lfD + (u32)lane * 8u, lfV);
            if ((lfN & ~7u) < 512u && (u32)lane < (lfN & 7u)) win[lfD + (lfN & ~7u) + (u32)lane] = lfB;
            if (lfN > 512u) kxd_wave_copy(win + lfD + 512u, dst + lfS + 512u, lfN - 512u, lane);
            // the other long ones, one at a time: literals, further matches from dst, matches from the window
            for (u64 longs = kx_ballot(own && ll > 32); longs; longs &= longs - 1) {
                int const e = (int)kx_ctz64(longs);
                kxd_wave_copy(win + (kx_bcast(dlit, e) - winBase), litPtr + kx_bcast(lp, e), kx_bcast(ll, e), lane);
            }
            for (u64 m = longFar & (longFar - 1); m; m &= m - 1) {
                int const e = (int)kx_ctz64(m);
                kxd_wave_copy(win + (kx_bcast(dmat, e) - winBase), dst + kx_bcast(srcPos, e), kx_bcast(ml, e), lane);
            }
            for (u64 m = kx_ballot(early && ml > 32 && inWin); m; m &= m - 1) {
                int const e = (int)kx_ctz64(m);
                kxd_wave_copy(win + (kx_bcast(dmat, e) - winBase), win + (kx_bcast(srcPos, e) - winBase), kx_bcast(ml, e), lane);
            }
            kx_lockstep();
            // ---- the matches that read this chunk, in dependency rounds (as in zstd_decode_frame): LDS to LDS ----
            // a match copied by its own lane has its whole source in the window; anything else goes by the whole wave
            bool const byWave = ml > 32 || off < 8 || inDictL || !inWin;
            u64 const W = kx_ballot(own && byWave);
            for (u64 P = (flags & 64u) ? 0ull : kx_ballot(own && !early); P; ) {
                int const e = (int)kx_ctz64(P);
                u32 const dE = kx_bcast(dmat, e);
                if ((W >> e) & 1ull) {
                    u32 const mlE = kx_bcast(ml, e), offE = kx_bcast(off, e);
                    u32 const inDict = offE > dE - fbase ? offE - (dE - fbase) : 0u;      // leading bytes that come from the dictionary
                    u32 const sE = dE - offE;
                    u8* const dw = win + (dE - winBase);
                    if (!inDict && offE >= mlE && sE >= winBase) kxd_wave_copy(dw, win + (sE - winBase), mlE, lane);
                    else if (offE >= 64) {
                        for (u32 base = 0; base < mlE; base += 64) {
                            u32 const k = base + (u32)lane;
                            if (k < mlE) dw[k] = kxd_hist_byte(dict, dict_size, win, winBase, dst, sE + k, k < inDict, inDict - k);
                            kx_lockstep();
                        }
                    } else if (inDict) {
                        if (lane == 0) for (u32 k = 0; k < mlE; k++) dw[k] = kxd_hist_byte(dict, dict_size, win, winBase, dst, sE + k, k < inDict, inDict - k);
                    } else {
                        u32 const chunk = (64 / offE) * offE; u32 const m = (u32)lane % offE;
                        u8 v = 0;
                        if ((u32)lane < chunk) v = kxd_hist_byte(dict, dict_size, win, winBase, dst, sE + m, false, 0);
                        for (u32 base = 0; base < mlE; base += chunk) {
                            u32 const k = base + (u32)lane;
                            if ((u32)lane < chunk && k < mlE) dw[k] = v;
                        }
                    }
                    kx_lockstep();
                    P &= P - 1;
                    continue;
                }
                bool const safe = ((P >> lane) & 1ull) && !byWave && (lane == e || srcPos + ml <= dE);
                if (safe) {
                    // only the earliest pending match can overlap its own source (the others' sources end below its destination)
                    const u8* const s_ = win + (srcPos - winBase);
                    if (off >= ml) kxd_copy32(dm_, s_, ml);
                    else { u32 k = 0; for (; k + 8 <= ml; k += 8) kx_st64(dm_ + k, kx_ld64(s_ + k)); for (; k < ml; k++) dm_[k] = s_[k]; }
                }
                P &= ~kx_ballot(safe);
                kx_lockstep();
            }
            // ---- the chunk goes out ----
            if (!(flags & 8u)) {
                const u8* const w0 = win + (op - winBase); u8* const d0 = dst + op;
                u32 i = (u32)lane * 8u;
                for (; i + 8 <= totOut; i += 512u) kx_st64(d0 + i, kx_ld64(w0 + i));
                u32 const tail = totOut & ~7u;
                if ((u32)lane < totOut - tail) d0[tail + lane] = w0[tail + lane];
            }
            kx_lockstep();
        }
        op += totOut; litUsed += totLit;
        if (cnt == 64) { hLL = pLL; hML = pML; hOF = pOF; heldAt = done + 64; }
        done += cnt;
    }
    KXD_EXEC_FAIL(0u)
}

KX_DEV void zstd_decode_frame(const KDecodeArgs& a, KDecodeLds& lds, u32 f, int lane)
{
    const u8* const src = a.src + a.in_off[f];
    u32 const srcSize = a.in_len[f];
    u8* const dst = a.dst + a.out_off[f];
    u32 const cap = a.out_cap[f];
    u8* const lits = a.lits + (size_t)f * a.lit_cap;
    u32 err = 0;

    // One entry may hold several frames back to back (and skippable frames between them): their contents are
    // concatenated, as ZSTD_decompress / ZSTD_decompressStream do (ZSTD_decompressMultiFrame).
    u32 pos = 0;                      // bytes of the entry consumed
    u32 op = 0;                       // bytes produced
    u32 nframes = 0;
    u32 cblk = 0;                     // compressed blocks of the entry's first frame so far (index into the pre-decoded records)
    // what the pre-decode kernels left for the entry's first frame: records, and (top bit) whether they cover the frame to
    // its last block.  Then nothing a later block could need from an earlier one (a tree-less block's table, a "repeat"
    // mode's counts) has to be kept up here, and the descriptions of pre-decoded blocks are not even read.
    u32 const npreW = a.pre_nblk ? a.pre_nblk[f] : 0u, nlitW = a.pre_nlit ? a.pre_nlit[f] : 0u;
    u32 const npre = npreW & 0x7FFFFFFFu, nlitpre = nlitW & 0x7FFFFFFFu;
    bool const seqAll = (npreW >> 31) != 0, litAll = (nlitW >> 31) != 0;
    for (;;) {
    // ---- frame header (every lane computes the same thing) ---------------
    u32 hasContent = 0, checksum = 0; u64 contentSize = 0; u64 windowSize = 0;
    u32 const fstart = pos, fbase = op;       // this frame's first input byte / first output byte
    if (srcSize - pos < 5) { if (srcSize != pos) err = KZE_SRCSIZE; break; }      // an empty entry decodes to nothing, as ZSTD_decompress has it
    {
        u32 const magic = kx_ld32(src + pos);
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {                // skippable frame: magic, 4-byte size, payload
            if (srcSize - pos < 8) { err = KZE_SRCSIZE; break; }
            u32 const sz = kx_ld32(src + pos + 4);
            if (sz > srcSize - pos - 8) { err = KZE_SRCSIZE; break; }
            pos += 8 + sz; nframes++;
            continue;
        }
        if (magic != 0xFD2FB528u) { err = nframes ? KZE_SRCSIZE : KZE_PREFIX; break; }   // garbage after a complete frame: "Src size is incorrect"
    }
    {
        u32 const fhd = src[pos + 4]; u32 const dictId = fhd & 3, single = (fhd >> 5) & 1, fcsId = fhd >> 6;
        checksum = (fhd >> 2) & 1;
        if (fhd & 0x08) err = KZE_FRAMEPARAM;
        pos += 5;
        u32 const didSize = dictId == 3 ? 4 : dictId;
        u32 const fcsSize = fcsId == 0 ? single : (fcsId == 1 ? 2 : fcsId == 2 ? 4 : 8);
        if (!err && pos + (single ? 0u : 1u) + didSize + fcsSize > srcSize) err = KZE_SRCSIZE;
        if (!err && !single) {
            u32 const wd = src[pos++]; u32 const wlog = 10 + (wd >> 3);
            if (wlog > 31) err = KZE_WINDOW;
            else { windowSize = 1ull << wlog; windowSize += (windowSize >> 3) * (wd & 7); }
        }
        if (!err) {
            u32 did = 0;
            for (u32 i = 0; i < didSize; i++) did |= (u32)src[pos + i] << (8 * i);
            if (did && did != a.dict_id) err = KZE_DICT;           // "Dictionary mismatch" (a header without an ID takes whatever dictionary is loaded)
            pos += didSize;
            if (fcsSize) {
                hasContent = 1;
                if (fcsSize == 1) contentSize = src[pos];
                else if (fcsSize == 2) contentSize = (u64)kx_ld16(src + pos) + 256;
                else if (fcsSize == 4) contentSize = kx_ld32(src + pos);
                else contentSize = kx_ld64(src + pos);
                pos += fcsSize;
                if (single) windowSize = contentSize;
                if (!err && contentSize > cap - op) err = KZE_DSTSMALL;
            }
        }
    }
    (void)fstart; (void)windowSize;

    // ---- blocks -----------------------------------------------------------
    u32 rep1 = 1, rep2 = 4, rep3 = 8;
    u32 hufLog = 0, hufNw = 0; bool hufValid = false;
    u32 tlLL = 0, tlOF = 0, tlML = 0;
    if (lane < 3) lds.keepKind[lane] = 0;
    if (a.dprior) {
        // a formatted dictionary: every frame starts as if a block with these tables and repeat offsets had come before it (ZSTD_loadDEntropy)
        const KDictDPrior& dp = *a.dprior;
        rep1 = dp.rep[0]; rep2 = dp.rep[1]; rep3 = dp.rep[2];
        hufLog = dp.hufLog; hufNw = dp.nw; hufValid = true;
        for (u32 i = (u32)lane; i < 256u; i += 64u) lds.weights[i] = dp.weights[i];
        if (lane < 3) { lds.keepKind[lane] = 2; lds.keepLog[lane] = dp.log[lane]; lds.keepMax[lane] = dp.max[lane]; }
        for (int t = 0; t < 3; t++) lds.keepNorm[t][lane] = dp.norm[t][lane];
    }
    kx_sync();
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
        u32 const ord = (nframes == 0) ? cblk++ : 0xFFFFFFFFu;          // index into what the pre-decode kernels left for the entry's first frame
        if (pos + bsize > srcSize || bsize > 128u * 1024u) { err = KZE_SRCSIZE; break; }     // libzstd: "Src size is incorrect" for both
        if (bsize < 2) { err = KZE_CORRUPT; break; }
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
        if (regen > 128u * 1024u) { err = KZE_CORRUPT; break; }
        // the literal buffer belongs to a context created for smaller slices: not the frame's fault (raw literals are read in place)
        if (ltype != 0 && regen > a.lit_cap) { err = KZE_WORKSPACE; break; }
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
            // the block's literals may lie decoded in HBM already (k_zstd_lit_predecode)
            bool preLit = false;
            if (ord < nlitpre) {
                KPreLit const pl = a.pre_lit[(size_t)f * a.pre_blk_cap + ord];
                if (pl.ok && pl.regen == regen) { litPtr = a.pre_lits + (size_t)f * a.pre_lit_cap + pl.off; preLit = true; }
            }
            if (preLit && litAll) { /* neither the tree nor the streams are looked at */ } else
            if (ltype == 2) {
                u32 r = 0;
                if (lane == 0) { u32 tl = 0, nw = 0; r = khuf_read_dtable(lds, bp + lpos, comp, &tl, &nw); lds.bc[0] = r; lds.bc[1] = tl; lds.bc[2] = nw; }
                kx_sync();
                r = lds.bc[0];
                if (r == 0) { err = KZE_CORRUPT; break; }
                hused = r; hufLog = lds.bc[1]; hufNw = lds.bc[2]; hufValid = true;
            } else {
                if (!hufValid) { err = KZE_CORRUPT; break; }
                // tree-less block: the table's LDS was reused by the previous block's sequence phase
                if (lane == 0) khuf_fill_dtable(lds, hufNw, hufLog);
                kx_sync();
            }
            const u8* const sp = bp + lpos + hused; u32 const ssize = comp - hused;
            bool ok = true;
            // (literals decoded ahead in a frame not covered to its end: the table above is still kept up for a later
            // tree-less block that this kernel has to decode itself)
            if (preLit) { /* nothing to decode */ } else
            if (nstreams == 1) {
                if (lane == 0 && !(a.flags & 1u)) ok = khuf_decode_stream(lds, hufLog, sp, ssize, lits, regen);
            } else {
                if (ssize < 10) { err = KZE_CORRUPT; break; }
                u32 const c0 = kx_ld16(sp), c1 = kx_ld16(sp + 2), c2 = kx_ld16(sp + 4);
                if (6 + c0 + c1 + c2 > ssize) { err = KZE_CORRUPT; break; }
                u32 const c3 = ssize - 6 - c0 - c1 - c2;
                u32 const seg = (regen + 3) / 4;
                if (3 * seg > regen) { err = KZE_CORRUPT; break; }
                if (lane < 4 && !(a.flags & 1u)) {
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
        // sequences header (every lane computes the same thing), then the three tables (lane 0)
        u32 nbSeq = 0, spos = lpos, modes = 0;
        {
            u32 e = 0;
            if (spos >= bend) e = KZE_CORRUPT;
            if (!e) {
                u32 const b0 = bp[spos++];
                if (b0 < 128) nbSeq = b0;
                else if (b0 < 255) { if (spos >= bend) e = KZE_CORRUPT; else nbSeq = ((b0 - 128) << 8) + bp[spos++]; }
                else { if (spos + 2 > bend) e = KZE_CORRUPT; else { nbSeq = kx_ld16(bp + spos) + 0x7F00; spos += 2; } }
            }
            if (!e && nbSeq) {
                if (spos >= bend) e = KZE_CORRUPT;
                else { modes = bp[spos++]; if (modes & 3) e = KZE_CORRUPT; }
            }
            if (e) { err = e; break; }
        }
        // this block's sequences may lie decoded in HBM already (k_zstd_seq_predecode)
        const u64* preSeq = nullptr; u32 preRep1 = 0, preRep2 = 0, preRep3 = 0;
        if (ord < npre) {
            KPreBlk const pb = a.pre_blk[(size_t)f * a.pre_blk_cap + ord];
            if (pb.ok && pb.nbSeq == nbSeq && nbSeq) { preSeq = a.pre_stage + ((size_t)f * a.pre_seq_cap + pb.seq_off); preRep1 = pb.rep[0]; preRep2 = pb.rep[1]; preRep3 = pb.rep[2]; }
        }
        if (nbSeq && !(preSeq && seqAll)) {
            for (int t = 0; t < 3 && !err; t++) {
                u32 const mode = (modes >> (6 - 2 * t)) & 3u;
                if (lane == 0) {
                    u32 r;
                    if (t == 0) r = kxd_seq_table(lds, 0, mode, bp + spos, bend - spos, &tlLL, !preSeq);
                    else if (t == 1) r = kxd_seq_table(lds, 1, mode, bp + spos, bend - spos, &tlOF, !preSeq);
                    else r = kxd_seq_table(lds, 2, mode, bp + spos, bend - spos, &tlML, !preSeq);
                    lds.bc[4] = r; lds.bc[5] = (t == 0) ? tlLL : (t == 1) ? tlOF : tlML;
                }
                kx_sync();
                u32 const r = lds.bc[4];
                if (r == KXD_FAIL) { err = KZE_CORRUPT; break; }
                spos += r;
                kx_sync();
            }
            if (err) break;
            if (spos >= bend) { err = KZE_CORRUPT; break; }
        }
        // lane 0 keeps the table logs / validity for later blocks; share them
        tlLL = kx_bcast(tlLL, 0); tlOF = kx_bcast(tlOF, 0); tlML = kx_bcast(tlML, 0);
        u32 litUsed = 0;
        if (nbSeq && preSeq && !(a.flags & 2u)) {
            KxdExecResult const xr = kxd_exec_window(a.dict, a.dict_size, a.flags, preSeq, nbSeq, dst, fbase, cap, litPtr, regen, op);
            err = xr.err; op = xr.op; litUsed = xr.litUsed;
            if (err) break;
            rep1 = preRep1; rep2 = preRep2; rep3 = preRep3;      // the repeat offsets after the block, for a later block decoded here
            kx_sync();
        } else if (nbSeq && !(a.flags & 2u)) {
            const u8* const sq = bp + spos; u32 const ssz = bend - spos;
            u32 const lastByte = sq[ssz - 1];
            if (lastByte == 0) { err = KZE_CORRUPT; break; }
            int const totalWords = (int)((ssz + 3) >> 2);
            // lane 0's backward reader: bitPos = unread bits. Every sequence builds a 64-bit container
            // (top bit = next unread bit) from three LDS words that are requested together with the three
            // state-table entries; fields then cost two shifts each. sbuf keeps two zero words below
            // the stream start so no read is conditional.
            int bitPos = (int)(8 * (ssz - 1) + kx_hb32(lastByte));
            u32 st3 = 0; bool bad = false, primed = false;       // st3: this lane's FSE state (lane 0 OF, 1 ML, 2.. LL)
            int sbLo = -1;                       // first stream word held in lds.u.q.sbuf[3..] (uniform)
#define KXD_WORD(i) lds.u.q.sbuf[(i) - sbLo + 3]     /* i >= sbLo - 3: the container of an exhausted stream (bitPos 0) reaches word -3 */
#define KXD_CONTAINER(C_) u64 C_; { int const topw_ = (bitPos - 1) >> 5; \
                u32 const hi_ = KXD_WORD(topw_), mid_ = KXD_WORD(topw_ - 1), lo_ = KXD_WORD(topw_ - 2); \
                u32 const used_ = (u32)(32 * (topw_ + 1) - bitPos); \
                C_ = ((((u64)hi_ << 32) | mid_) << used_) | ((u64)lo_ >> (32u - used_)); }
#define KXD_GET(dst_, C_, n_) { u32 const n__ = (n_); dst_ = (u32)(((C_) >> 1) >> (63u - n__)); (C_) <<= n__; }
// n bits that start c bits below the container's top; independent of the other fields
#define KXD_AT(C_, c_, n_) ((u32)((((C_) << (c_)) >> 1) >> (63u - (n_))))
            for (u32 done = 0; done < nbSeq && !err; ) {
                u32 const cnt = (nbSeq - done) < 64 ? (nbSeq - done) : 64;
                bool const own = (u32)lane < cnt;
                u32 ll, ml, off;
                // keep >= 176 words (64 sequences x <= 88 bits) of stream below the read position in LDS (254 staged)
                int const curWord = (int)kx_bcast((u32)bitPos, 0) >> 5;
                if (sbLo < 0 || (sbLo > 0 && curWord - sbLo < 176)) {
                    int newLo = curWord + 2 - 254; if (newLo < 0) newLo = 0;
                    int hiW = curWord + 2; if (hiW > totalWords) hiW = totalWords;
                    kx_sync();
                    for (int i = newLo - 3 + lane; i < hiW; i += 64) {
                        int const o = 4 * i; u32 v = 0;
                        if (i >= 0) {
                            if (o + 4 <= (int)ssz) v = kx_ld32(sq + o);
                            else for (int k = 0; o + k < (int)ssz; k++) v |= (u32)sq[o + k] << (8 * k);
                        }
                        lds.u.q.sbuf[i - newLo + 3] = v;
                    }
                    sbLo = newLo;
                    kx_sync();
                }
                {
                    // Every lane runs the loop (the cost of an instruction does not depend on how many lanes are active);
                    // lanes 0, 1, 2 decode the OF, ML and LL field of a sequence at once: one table look-up, one extra-bits
                    // field and one state update per lane instead of three in a row on one lane.  Lanes >= 3 shadow lane 2.
                    // The three lanes' bit counts and values are exchanged with kx_bcast (v_readlane: uniform values, so
                    // the repeat-offset rules run on the scalar unit).
                    u32 const role = lane < 3 ? (u32)lane : 2u;
                    u32 const tb = role == 0 ? (u32)KXD_OF0 : role == 1 ? (u32)KXD_ML0 : (u32)KXD_LL0;
                    const u32* const xt = role == 1 ? lds.mlx : lds.llx;
                    if (!primed) {
                        u32 const need0 = tlLL + tlOF + tlML;
                        bad |= bitPos < (int)need0;
                        KXD_CONTAINER(C0)
                        // initial states in stream order LL, OF, ML
                        st3 = role == 2 ? KXD_AT(C0, 0u, tlLL) : role == 0 ? KXD_AT(C0, tlLL, tlOF) : KXD_AT(C0, tlLL + tlOF, tlML);
                        bitPos -= (int)need0; bitPos = bitPos < 0 ? 0 : bitPos;
                        primed = true;
                    }
                    bool const lastChunk = done + cnt == nbSeq;
                    u32 const full = lastChunk ? cnt - 1 : cnt;       // the block's final sequence updates no state
                    // a corrupt stream only sets `bad`: states stay inside their tables by construction, bitPos is clamped at 0
                    for (u32 i = 0; i < cnt; i++) {
                        KXD_CONTAINER(C)
                        u32 const e = lds.u.q.fb[tb + st3], code = lds.u.q.fc[tb + st3];
                        u32 const x = xt[code];                                     // baseValue | extraBits << 24 (LL, ML)
                        u32 const a = role == 0 ? code : x >> 24;
                        u32 const base = role == 0 ? (1u << code) : (x & 0xFFFFFFu);
                        u32 const n = (i < full) ? e >> 12 : 0u;
                        u32 const aO = kx_bcast(a, 0), aM = kx_bcast(a, 1), aL = kx_bcast(a, 2);
                        u32 const nO = kx_bcast(n, 0), nM = kx_bcast(n, 1), nL = kx_bcast(n, 2);
                        u32 const needA = aO + aM + aL, needB = nL + nM + nO;
                        bad |= bitPos < (int)(needA + needB);
                        // bit order inside a sequence: OF extra, ML extra, LL extra, then LL state, ML state, OF state
                        u32 const offA = role == 0 ? 0u : role == 1 ? aO : aO + aM;
                        u32 offB = role == 2 ? 0u : role == 1 ? nL : nL + nM;
                        u32 const xv = KXD_AT(C, offA, a);
                        u64 Cs = C;
                        if (needA + needB > 64) { bitPos -= (int)needA; KXD_CONTAINER(C2) bitPos += (int)needA; Cs = C2; }   // rare
                        else offB += needA;
                        u32 const yv = KXD_AT(Cs, offB, n);
                        bitPos -= (int)(needA + needB); bitPos = bitPos < 0 ? 0 : bitPos;
                        u32 const val = base + xv;
                        u32 const ofv = kx_bcast(val, 0), ml = kx_bcast(val, 1), ll = kx_bcast(val, 2);
                        // repeat-offset rules
                        bool const isRep = ofv <= 3;
                        u32 const idx = ofv - 1 + (ll == 0);                       // meaningful when isRep
                        u32 const rm1 = (rep1 - 1) ? rep1 - 1 : 1u;
                        u32 const roff = idx == 0 ? rep1 : idx == 1 ? rep2 : idx == 2 ? rep3 : rm1;
                        u32 const off = isRep ? roff : ofv - 3;
                        bool const sh2 = !isRep || idx >= 2, sh1 = !isRep || idx >= 1;
                        rep3 = sh2 ? rep2 : rep3; rep2 = sh1 ? rep1 : rep2; rep1 = off;
                        if (lane == 0) { lds.u.q.stage[3 * i] = ll; lds.u.q.stage[3 * i + 1] = ml; lds.u.q.stage[3 * i + 2] = off; }
                        st3 = (e & 0xFFFu) + yv;
                    }
                    if (lastChunk && !bad && bitPos != 0) bad = true;
                    if (lane == 0) lds.u.q.stage[192] = bad ? 1u : 0u;
                    kx_sync();
                    if (lds.u.q.stage[192]) { err = KZE_CORRUPT; break; }
                    ll = own ? lds.u.q.stage[3 * lane] : 0u; ml = own ? lds.u.q.stage[3 * lane + 1] : 0u; off = own ? lds.u.q.stage[3 * lane + 2] : 1u;
                }
                // ---- execute the chunk: lane i owns sequence i -----------------
                u32 sl = ll, st = ll + ml;                    // inclusive scans over the lanes
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    u32 const a1 = kx_shfl(sl, lane - o), a2 = kx_shfl(st, lane - o);
                    if (lane >= o) { sl += a1; st += a2; }
                }
                u32 const totLit = kx_bcast(sl, 63), totOut = kx_bcast(st, 63);
                if (litUsed + totLit > regen) { err = KZE_CORRUPT; break; }
                if ((u64)op + totOut > cap) { err = KZE_DSTSMALL; break; }
                u32 const lp = litUsed + (sl - ll);           // my literals in the literal buffer
                u32 const dlit = op + (st - ll - ml);         // where they go
                u32 const dmat = dlit + ll;                   // where my match goes
                // a match may start inside the dictionary (the history before the frame's first byte)
                if (kx_any(own && off > dmat - fbase + a.dict_size)) { err = KZE_CORRUPT; break; }
                // literals: short runs lane-serially (exact length), long runs by the whole wave
                if (a.flags & 4u) { op += totOut; litUsed += totLit; kx_sync(); done += cnt; continue; }
                if (own && ll <= 32) kxd_copy32(dst + dlit, litPtr + lp, ll);
                for (u64 longs = kx_ballot(own && ll > 32); longs; longs &= longs - 1) {
                    int const e = (int)kx_ctz64(longs);
                    kxd_wave_copy(dst + kx_bcast(dlit, e), litPtr + kx_bcast(lp, e), kx_bcast(ll, e), lane);      // e is uniform: v_readlane, no LDS permute
                }
                kx_lockstep();
                // matches in dependency rounds: a short match with offset >= 8 whose source lies before
                // the earliest pending match's destination is copied by its own lane; the earliest pending
                // one is always runnable; long or small-offset matches are copied by the whole wave in order
                for (u64 P = kx_ballot(own); P; ) {
                    int const e = (int)kx_ctz64(P);
                    u32 const mlE = kx_bcast(ml, e), offE = kx_bcast(off, e), dE = kx_bcast(dmat, e);
                    if (offE > dE - fbase) {
                        // starts in the dictionary: byte k comes from dict[dict_size - inDict + k] while that is inside
                        // the dictionary, then from the frame's own output
                        u32 const inDict = offE - (dE - fbase); const u8* const dsrc = a.dict + (a.dict_size - inDict);
                        if (offE >= 64) {
                            for (u32 base = 0; base < mlE; base += 64) {
                                u32 const k = base + (u32)lane;
                                if (k < mlE) dst[dE + k] = (k < inDict) ? dsrc[k] : dst[dE + k - offE];
                                kx_lockstep();
                            }
                        } else if (lane == 0) {
                            for (u32 k = 0; k < mlE; k++) dst[dE + k] = (k < inDict) ? dsrc[k] : dst[dE + k - offE];
                        }
                        kx_lockstep();
                        P &= P - 1;
                        continue;
                    }
                    if (mlE > 32 || offE < 8) {
                        const u8* const ms = dst + dE - offE;
                        if (offE >= 64) {
                            for (u32 base = 0; base < mlE; base += 64) {
                                u32 const k = base + (u32)lane;
                                if (k < mlE) dst[dE + k] = ms[k];
                                kx_lockstep();
                            }
                        } else {
                            u32 const chunk = (64 / offE) * offE; u32 const m = (u32)lane % offE;
                            u8 v = 0;
                            if ((u32)lane < chunk) v = ms[m];
                            for (u32 base = 0; base < mlE; base += chunk) {
                                u32 const k = base + (u32)lane;
                                if ((u32)lane < chunk && k < mlE) dst[dE + k] = v;
                            }
                            kx_lockstep();
                        }
                        P &= P - 1;
                        continue;
                    }
                    bool const safe = ((P >> lane) & 1ull) && ml <= 32 && off >= 8 && off <= dmat - fbase && (lane == e || dmat - off + ml <= dE);
                    if (safe) {
                        // only the earliest pending match can overlap its own source (the others' sources end below its destination)
                        const u8* const s_ = dst + dmat - off; u8* const d_ = dst + dmat;
                        if (off >= ml) kxd_copy32(d_, s_, ml);
                        else { u32 k = 0; for (; k + 8 <= ml; k += 8) kx_st64(d_ + k, kx_ld64(s_ + k)); for (; k < ml; k++) d_[k] = s_[k]; }
                    }
                    P &= ~kx_ballot(safe);
                    kx_lockstep();
                }
                op += totOut; litUsed += totLit;
                kx_sync();
                done += cnt;
            }
#undef KXD_AT
#undef KXD_GET
#undef KXD_CONTAINER
#undef KXD_WORD
            if (err) break;
        }
        // remaining literals
        if (a.flags & 2u) litUsed = 0;
        if (litUsed > regen || (u64)op + (regen - litUsed) > cap) { err = (litUsed > regen) ? KZE_CORRUPT : KZE_DSTSMALL; break; }
        kxd_wave_copy(dst + op, litPtr + litUsed, regen - litUsed, lane);
        op += regen - litUsed;
        pos += bsize;
        kx_sync();
    }
    kx_sync();
    if (!err && hasContent && contentSize != op - fbase) err = KZE_CORRUPT;
    if (!err && checksum) {
        if (pos + 4 > srcSize) err = KZE_SRCSIZE;
        else {
            u32 bad = 0;
            if (lane == 0) bad = ((u32)kxxh64(dst + fbase, op - fbase) != kx_ld32(src + pos)) ? 1u : 0u;
            bad = kx_shfl(bad, 0);
            if (bad) err = KZE_CHECKSUM;
            pos += 4;
        }
    }
    if (err) break;
    nframes++;
    }   // next frame of the entry
    if (lane == 0) { a.status[f] = err; a.out_len[f] = err ? 0u : op; }
}

KX_DEV void zstd_decode_body(const KDecodeArgs& a)
{
    KDecodeLds& lds = g_kxd_lds;
    int const lane = kx_lane();
    if (lane < 36) lds.llx[lane] = kx_ll_base((u32)lane) | (kxd_ll_bits((u32)lane) << 24);
    if (lane < 53) lds.mlx[lane] = kx_ml_base((u32)lane) | (kxd_ml_bits((u32)lane) << 24);
    kx_sync();
    for (u32 it = kx_block(); it < a.n_slices; it += kx_nblocks()) {
        u32 const f = kx_xcd_chunk(it, a.n_slices);
        zstd_decode_frame(a, lds, f, lane);
        kx_sync();
    }
}
