// zstd_cdict_host.h -- host-side construction of libzstd's CDict for a raw-content dictionary at level 3:
// parameters of ZSTD_getCParams(3, unknown source size, dictSize) in "create CDict" mode, then
// ZSTD_fillDoubleHashTableForCDict over the dictionary (tagged entries: index << 8 | tag, index = position + 2).
// Built once per dictionary on the host and uploaded (kmp_api.hip); the emulator harness uses the same code.
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>
static inline u32 hb32_host(u32 v) { return 31u - (u32)__builtin_clz(v); }
static inline void cdict_params(u32 dictSize, u32* W, u32* C, u32* H, u32* mml)
{
    u64 const rSize = (u64)dictSize + 500;
    if (rSize <= 16384)       { *W = 14; *C = 14; *H = 15; *mml = 4; }
    else if (rSize <= 131072) { *W = 17; *C = 15; *H = 16; *mml = 5; }
    else if (rSize <= 262144) { *W = 18; *C = 16; *H = 16; *mml = 4; }
    else                      { *W = 21; *C = 16; *H = 17; *mml = 5; }
    u32 const srcSize = 513, tSize = srcSize + dictSize;
    u32 const srcLog = (tSize < 64) ? 6 : hb32_host(tSize - 1) + 1;
    if (*W > srcLog) *W = srcLog;
    u32 dawl = *W;
    { u64 const windowSize = 1ull << *W; if (windowSize < (u64)dictSize + srcSize) dawl = hb32_host((u32)(dictSize + windowSize) - 1) + 1; }
    if (*H > dawl + 1) *H = dawl + 1;
    if (*C > dawl) *C = dawl;
    if (*W < 10) *W = 10;
}
static inline u64 rd64_host(const u8* p) { u64 v; memcpy(&v, p, 8); return v; }
static inline u32 hash_long_host(const u8* p, u32 hBits) { return (u32)((rd64_host(p) * 0xCF1BBCDCB7A56463ULL) >> (64 - hBits)); }
static inline u32 hash_short_host(const u8* p, u32 hBits, u32 mls)
{
    if (mls == 4) { u32 v; memcpy(&v, p, 4); return (v * 2654435761U) >> (32 - hBits); }
    return (u32)(((rd64_host(p) << 24) * 889523592379ULL) >> (64 - hBits));
}
static inline void cdict_fill(std::vector<u32>& tl, u32 hLog, std::vector<u32>& ts, u32 cLog, u32 mls, const u8* dict, size_t D)
{
    tl.assign((size_t)1 << hLog, 0u); ts.assign((size_t)1 << cLog, 0u);
    size_t const maxDict = (size_t)1 << ((hLog + 3 > cLog + 1) ? hLog + 3 : cLog + 1);     // larger dictionaries: only the suffix is indexed
    size_t const from = D > maxDict ? D - maxDict : 0;
    if (D - from <= 8) return;
    const u8* ip = dict + from; const u8* const iend = dict + D - 8;
    for (; ip + 2 <= iend; ip += 3) {
        u32 const curr = 2u + (u32)(ip - dict);
        for (u32 i = 0; i < 3; ++i) {
            u32 const sm = hash_short_host(ip + i, cLog + 8, mls), lg = hash_long_host(ip + i, hLog + 8);
            if (i == 0) ts[sm >> 8] = ((curr + i) << 8) | (sm & 0xFFu);
            if (i == 0 || tl[lg >> 8] == 0) tl[lg >> 8] = ((curr + i) << 8) | (lg & 0xFFu);
        }
    }
}


// ---- formatted dictionaries (magic EC30A437) -------------------------------------------------------------------------------
// Layout (zstd's dictionary format): magic, dictID, the Huffman table of the literals (a tree description as in a literals section),
// the offset / match-length / literal-length tables as FSE normalised counts, three repeat offsets, the content.  What libzstd does
// with one (ZSTD_loadCEntropy) is restated for KDictPrior (zstd_entropy.h, which must be included before this file): the tables become the
// first block's "previous" tables, valid as they stand when they leave no symbol out.  Returns 1 and fills *out / *contentOff, 0 when
// the bytes do not start with the magic (raw content), -1 when the header is damaged (libzstd: "Dictionary is corrupted").
struct cdict_fwdbits { const u8* p; size_t size; size_t bit; };
static inline u32 cdict_peek(const cdict_fwdbits& b, u32 n)
{
    u64 v = 0; size_t const byte = b.bit >> 3;
    for (u32 k = 0; k < 8 && byte + k < b.size; k++) v |= (u64)b.p[byte + k] << (8 * k);
    return (u32)((v >> (b.bit & 7)) & ((1ull << n) - 1));
}
static inline size_t cdict_read_ncount(short* norm, u32* maxSymbolValue, u32* tableLogOut, const u8* p, size_t size, u32 maxLog)
{
    cdict_fwdbits b{ p, size, 0 };
    if (size < 1) return 0;
    u32 const tableLog = cdict_peek(b, 4) + 5; b.bit += 4;
    if (tableLog > maxLog) return 0;
    *tableLogOut = tableLog;
    int remaining = (1 << tableLog) + 1, threshold = 1 << tableLog; u32 nbBits = tableLog + 1, sym = 0; u32 const maxSV = *maxSymbolValue; bool previous0 = false;
    while (remaining > 1 && sym <= maxSV) {
        if (previous0) {
            for (;;) {
                u32 const r = cdict_peek(b, 2); b.bit += 2;
                for (u32 i = 0; i < r && sym <= maxSV; i++) norm[sym++] = 0;
                if (r != 3) break;
                if ((b.bit >> 3) > size) return 0;
            }
            if (sym > maxSV) break;
        }
        int const max = (2 * threshold - 1) - remaining; int count;
        u32 const v = cdict_peek(b, nbBits);
        if ((int)(v & (u32)(threshold - 1)) < max) { count = (int)(v & (u32)(threshold - 1)); b.bit += nbBits - 1; }
        else { count = (int)(v & (u32)(2 * threshold - 1)); if (count >= threshold) count -= max; b.bit += nbBits; }
        count--;
        remaining -= count < 0 ? -count : count;
        norm[sym++] = (short)count;
        previous0 = (count == 0);
        while (remaining < threshold) { nbBits--; threshold >>= 1; }
        if ((b.bit >> 3) > size) return 0;
    }
    if (remaining != 1 || ((b.bit + 7) >> 3) > size) return 0;
    for (u32 s = sym; s <= maxSV; s++) norm[s] = 0;
    *maxSymbolValue = sym - 1;
    return (b.bit + 7) >> 3;
}
// the literals' tree description -> weights[0 .. *nw) (the implied last one included) and the table log; bytes read, 0 on error
static inline size_t cdict_read_huf_weights(u8* w, u32* nwOut, u32* tableLogOut, const u8* p, size_t size)
{
    if (size < 1) return 0;
    u32 const hb = p[0]; u32 nw = 0; size_t used;
    if (hb >= 128) {
        nw = hb - 127; used = 1 + (nw + 1) / 2;
        if (used > size) return 0;
        for (u32 i = 0; i < nw; i += 2) { w[i] = p[1 + i / 2] >> 4; w[i + 1] = p[1 + i / 2] & 15; }
    } else {
        used = 1 + hb;
        if (hb == 0 || used > size) return 0;
        short norm[16]; u32 maxSV = 12, tl = 0;
        size_t const h = cdict_read_ncount(norm, &maxSV, &tl, p + 1, hb, 6);
        if (h == 0) return 0;
        u16 db[64]; u8 dc[64], tsym[64]; u16 symnext[16];
        u32 const tableSize = 1u << tl, mask = tableSize - 1, step = (tableSize >> 1) + (tableSize >> 3) + 3; u32 high = tableSize - 1, pos = 0;
        for (u32 s = 0; s <= maxSV; s++) { if (norm[s] == -1) { tsym[high--] = (u8)s; symnext[s] = 1; } else symnext[s] = (u16)norm[s]; }
        for (u32 s = 0; s <= maxSV; s++) for (int k = 0; k < norm[s]; k++) { tsym[pos] = (u8)s; pos = (pos + step) & mask; while (pos > high) pos = (pos + step) & mask; }
        for (u32 u = 0; u < tableSize; u++) { u32 const s = tsym[u]; u32 const next = symnext[s]++; u32 const nb = tl - hb32_host(next); db[u] = (u16)((((next << nb) - tableSize) & 0xFFFu) | (nb << 12)); dc[u] = (u8)s; }
        const u8* const sp = p + 1 + h; size_t const ssz = hb - h;
        if (ssz == 0 || sp[ssz - 1] == 0) return 0;
        long bits = (long)(8 * (ssz - 1) + hb32_host(sp[ssz - 1]));
        auto take = [&](u32 n) { u32 r = 0; for (u32 k = 0; k < n; k++) { long const bi = bits - (long)n + (long)k; if (bi >= 0) r |= (u32)((sp[bi >> 3] >> (bi & 7)) & 1u) << k; } bits -= (long)n; return r; };
        if (bits < (long)(2 * tl)) return 0;
        u32 s1 = take(tl), s2 = take(tl);
        for (;;) {
            if (nw >= 255) return 0;
            u32 e = db[s1]; w[nw++] = dc[s1]; u32 nb = e >> 12;
            if (bits < (long)nb) { if (nw >= 255) return 0; w[nw++] = dc[s2]; break; }
            s1 = (e & 0xFFFu) + take(nb);
            if (nw >= 255) return 0;
            e = db[s2]; w[nw++] = dc[s2]; nb = e >> 12;
            if (bits < (long)nb) { if (nw >= 255) return 0; w[nw++] = dc[s1]; break; }
            s2 = (e & 0xFFFu) + take(nb);
        }
    }
    u32 total = 0;
    for (u32 i = 0; i < nw; i++) { if (w[i] > 12) return 0; if (w[i]) total += 1u << (w[i] - 1); }
    if (total == 0) return 0;
    u32 const tableLog = hb32_host(total) + 1;
    if (tableLog > 12) return 0;
    u32 const rest = (1u << tableLog) - total;
    if (rest & (rest - 1)) return 0;
    w[nw++] = (u8)(hb32_host(rest) + 1);
    *nwOut = nw; *tableLogOut = tableLog;
    return used;
}
// (dec: the same header for a decoder, optional.  avail: how many of the dictionary's dictSize bytes lie at `dict` -- a decoder's caller
// holds the dictionary in device memory and only its head is brought over; the header ends well inside 2 KiB.)
static inline int cdict_parse_formatted(const u8* dict, size_t dictSize, KDictPrior* out, size_t* contentOff, KDictDPrior* dec = nullptr, size_t avail = ~(size_t)0)
{
    if (avail > dictSize) avail = dictSize;
    if (dictSize < 8 || avail < 8 || memcmp(dict, "\x37\xA4\x30\xEC", 4) != 0) return 0;
    size_t const fullSize = dictSize; dictSize = avail;            // (reads stay inside what is here; the content's size comes from fullSize)
    KDictPrior scratchPrior; if (!out) out = &scratchPrior;
    if (dec) memset(dec, 0, sizeof(*dec));
    memset(out, 0, sizeof(*out));
    memcpy(&out->dictID, dict + 4, 4);
    size_t pos = 8;
    {
        u8 w[260]; u32 nw = 0, tableLog = 0;
        size_t const h = cdict_read_huf_weights(w, &nw, &tableLog, dict + pos, dictSize - pos);
        if (h == 0) return -1;
        pos += h;
        // canonical codes: within a length, in symbol order; the shorter codes take the higher values (HUF_readCTable)
        u32 nbPerRank[16] = { 0 }, valPerRank[16] = { 0 }; bool hasZero = false;
        for (u32 s = 0; s < nw; s++) { if (!w[s]) hasZero = true; nbPerRank[w[s] ? tableLog + 1 - w[s] : tableLog + 1]++; }
        { u32 min = 0; for (u32 n = tableLog; n > 0; n--) { valPerRank[n] = min; min += nbPerRank[n]; min >>= 1; } }
        for (u32 s = 0; s < nw; s++) { u32 const nb = w[s] ? tableLog + 1 - w[s] : 0u; u32 const val = valPerRank[w[s] ? nb : tableLog + 1]++; out->ct[s] = w[s] ? (val | (nb << 16)) : 0u; }
        out->hufMode = (!hasZero && nw == 256) ? 2u : 1u;
        if (dec) { if (tableLog > 11) return -1; memcpy(dec->weights, w, nw); dec->nw = nw; dec->hufLog = tableLog; }     // (the decoder's table holds depths to 11, the format's limit for literals)
    }
    u32 offMaxRead = 31;
    for (int k = 0; k < 3; k++) {                          // the format's order: offsets, match lengths, literal lengths
        int const t = k == 0 ? 1 : k == 1 ? 2 : 0;         // KDictPrior's: [0] LL, [1] OF, [2] ML
        u32 max = t == 0 ? 35u : t == 1 ? 31u : 52u, lg = 0;
        size_t const h = cdict_read_ncount(out->norm[t], &max, &lg, dict + pos, dictSize - pos, t == 1 ? 8u : 9u);
        if (h == 0) return -1;
        pos += h;
        out->log[t] = lg;
        if (dec) { memcpy(dec->norm[t], out->norm[t], sizeof(dec->norm[t])); dec->log[t] = lg; dec->max[t] = max; }
        if (t == 1) { out->maxSym[1] = 31; offMaxRead = max; }     // (the offset table is built over all 32 codes, the others over what was read)
        else {
            out->maxSym[t] = max;
            u32 const need = t == 0 ? 35u : 52u; bool ok = max >= need;
            for (u32 s = 0; ok && s <= need; s++) if (out->norm[t][s] == 0) ok = false;
            out->seqValid[t] = ok ? 1u : 0u;
        }
    }
    if (pos + 12 > dictSize) return -1;
    memcpy(out->rep, dict + pos, 12); pos += 12;
    if (dec) { memcpy(dec->rep, out->rep, 12); dec->dictID = out->dictID; }
    size_t const content = fullSize - pos;
    {
        u32 const offcodeMax = hb32_host((u32)content + (128u << 10)); u32 const need = offcodeMax < 31u ? offcodeMax : 31u;
        bool ok = offMaxRead >= need;
        for (u32 s = 0; ok && s <= need; s++) if (out->norm[1][s] == 0) ok = false;
        out->seqValid[1] = ok ? 1u : 0u;
    }
    for (int i = 0; i < 3; i++) if (out->rep[i] == 0 || out->rep[i] > content) return -1;
    *contentOff = pos;
    return 1;
}
