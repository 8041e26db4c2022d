/*
 * oracle/zstd_l3_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of what the reference's ZstdCompressor(level=3)
 * computes for a one-shot slice (reference call site:
 * kompressor-zstd--nativelib/src/jvmCommonMain/jni/Wrapper.cpp:112,
 * ZSTD_compressStream2(cctx,&out,&in,ZSTD_e_end), driven by
 * kompressor-core/.../SliceTransform.kt:33-45 with finish=true).
 *
 * The arithmetic lives in a third-party dependency that is ABSENT from
 * /root/reference: com.ensody.nativebuilds:zstd-libzstd:1.5.7.8
 * (gradle/libs.versions.toml:9,46) == upstream libzstd 1.5.7.  This file
 * restates its published algorithm for the level-3 path (strategy "dfast":
 * two-table greedy LZ, HUF literals, FSE sequences, zstd frame format
 * RFC 8878): single-block frames up to 128 KiB, and multi-block frames up to
 * 2 MiB (block pre-splitter, repcodes and Huffman table carried from block to
 * block, "treeless" literal sections, RLE blocks).
 *
 * Parity pin: byte-for-byte equality with a binary libzstd 1.5.7
 * (ZSTD_versionNumber()==10507) run in the build container through
 * oracle/libzstd_ref.py; the resulting vectors are committed under
 * tests/golden/ (generator: tests/golden/make_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this. The product path never links or calls it.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>

#define KREF_API __attribute__((visibility("default")))

typedef uint8_t  u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

/* ------------------------------------------------------------------ */
/* little helpers                                                      */
/* ------------------------------------------------------------------ */
static inline u32 rd32(const u8* p) { u32 v; memcpy(&v, p, 4); return v; }
static inline u64 rd64(const u8* p) { u64 v; memcpy(&v, p, 8); return v; }
static inline void wr16(u8* p, u32 v) { p[0] = (u8)v; p[1] = (u8)(v >> 8); }
static inline void wr24(u8* p, u32 v) { p[0] = (u8)v; p[1] = (u8)(v >> 8); p[2] = (u8)(v >> 16); }
static inline void wr32(u8* p, u32 v) { wr24(p, v); p[3] = (u8)(v >> 24); }
static inline u32 hb32(u32 v) { return 31u - (u32)__builtin_clz(v); }

/* ------------------------------------------------------------------ */
/* level-3 compression parameters as a function of the slice size      */
/* (libzstd: ZSTD_getCParams(3, srcSize, 0) after ZSTD_adjustCParams)  */
/* ------------------------------------------------------------------ */
typedef struct { u32 windowLog, chainLog, hashLog, minMatch; } kref_params;

KREF_API void kref_params_l3(size_t srcSize, u32* out4)
{
    /* base rows of the level-3 table per size class */
    u32 W, C, H, mml;
    if (srcSize <= 16384)        { W = 14; C = 14; H = 15; mml = 4; }
    else if (srcSize <= 131072)  { W = 17; C = 15; H = 16; mml = 5; }
    else if (srcSize <= 262144)  { W = 18; C = 16; H = 16; mml = 4; }
    else                         { W = 21; C = 16; H = 17; mml = 5; }
    {
        u32 const srcLog = (srcSize < 64) ? 6 : hb32((u32)(srcSize - 1)) + 1;
        if (W > srcLog) W = srcLog;
        if (H > W + 1) H = W + 1;
        if (C > W) C = W;
        if (W < 10) W = 10;
    }
    out4[0] = W; out4[1] = C; out4[2] = H; out4[3] = mml;
}

/* Level 4's double-fast rows (libzstd: ZSTD_getCParams(4, srcSize, 0)): above 16 KiB up to 128 KiB {17,17,17,mml 4},
 * above 256 KiB {21,18,18, mml 5}.  Its other rows are strategy "greedy" (up to 16 KiB and 128 - 256 KiB): returns 0 there. */
KREF_API int kref_params_l4(size_t srcSize, u32* out4)
{
    u32 W, C, H, mml;
    if (srcSize <= 16384) return 0;
    else if (srcSize <= 131072) { W = 17; C = 17; H = 17; mml = 4; }
    else if (srcSize <= 262144) return 0;
    else                        { W = 21; C = 18; H = 18; mml = 5; }
    {
        u32 const srcLog = hb32((u32)(srcSize - 1)) + 1;
        if (W > srcLog) W = srcLog;
        if (H > W + 1) H = W + 1;
        if (C > W) C = W;
    }
    out4[0] = W; out4[1] = C; out4[2] = H; out4[3] = mml;
    return 1;
}

/* ------------------------------------------------------------------ */
/* sequence store                                                      */
/* ------------------------------------------------------------------ */
typedef struct { u32 offBase; u16 litLength; u16 mlBase; } kref_seq;

typedef struct {
    kref_seq* seqs; size_t nbSeq;
    u8* lits; size_t litSize;
    int longLengthType;      /* 0 none, 1 literal length, 2 match length */
    size_t longLengthPos;
    int strategy;            /* 0 = dfast (levels 3), 1 = fast (levels 1, 2): only the encoding-type heuristic looks at it */
} seqstore;

static void store_seq(seqstore* ss, size_t litLength, const u8* lit, u32 offBase, size_t matchLength)
{
    memcpy(ss->lits + ss->litSize, lit, litLength);
    ss->litSize += litLength;
    if (litLength > 0xFFFF) { ss->longLengthType = 1; ss->longLengthPos = ss->nbSeq; }
    ss->seqs[ss->nbSeq].litLength = (u16)litLength;
    ss->seqs[ss->nbSeq].offBase = offBase;
    {
        size_t const mlBase = matchLength - 3;
        if (mlBase > 0xFFFF) { ss->longLengthType = 2; ss->longLengthPos = ss->nbSeq; }
        ss->seqs[ss->nbSeq].mlBase = (u16)mlBase;
    }
    ss->nbSeq++;
}

/* ------------------------------------------------------------------ */
/* double-fast match finder (noDict, first block of a frame)           */
/* ------------------------------------------------------------------ */
/* Optional event trace of the hash-table traffic (tools/table_traffic_model.c defines KREF_TRACE and the hook before
 * including this file): table 0 = long, 1 = short; kind 0 = probe, 1 = insert of the searched position, 2 = long-table
 * insert of ip+1, 3 = complementary insert after a match, 4 = insert at an immediate-repcode position. */
#ifdef KREF_TRACE
#define KREF_EV(table, bucket, kind) kref_trace_event((table), (u32)(bucket), (kind))
#else
#define KREF_EV(table, bucket, kind) ((void)0)
#endif
static inline size_t hash_long(const u8* p, u32 hBits)
{
    return (size_t)((rd64(p) * 0xCF1BBCDCB7A56463ULL) >> (64 - hBits));
}
static inline size_t hash_short(const u8* p, u32 hBits, u32 mls)
{
    switch (mls) {
    default:
    case 4: return (size_t)((rd32(p) * 2654435761U) >> (32 - hBits));
    case 5: return (size_t)(((rd64(p) << 24) * 889523592379ULL) >> (64 - hBits));
    case 6: return (size_t)(((rd64(p) << 16) * 227718039650203ULL) >> (64 - hBits));
    case 7: return (size_t)(((rd64(p) << 8) * 58295818150454627ULL) >> (64 - hBits));
    }
}
static size_t count_eq(const u8* ip, const u8* match, const u8* iend)
{
    const u8* const s = ip;
    while (ip < iend && *ip == *match) { ip++; match++; }
    return (size_t)(ip - s);
}

/* Table entries are "indices": index = position + 2 (the first window
 * index of a fresh libzstd match state is 2); 0 means empty. */
#define IDX0 2u

/* One block src[blockStart, blockStart + srcSize) of an input whose first byte has index 2.  The lowest valid index is
 * ZSTD_getLowestPrefixIndex(ms, curr, windowLog) = max(dictLimit, curr - maxDist): 2 while the input fits the window,
 * above it once the window has slid (the buffered frames further down) -- taken at the END of the block for the
 * candidates, at the first searched position for the repcodes. */
static size_t dfast_block_low(seqstore* ss, u32 rep[3], const u8* input, size_t blockStart, size_t srcSize,
                              u32* hashLong, u32 hBitsL, u32* hashSmall, u32 hBitsS, u32 mls, u32 dictLimit, u32 maxDist);
static size_t dfast_block(seqstore* ss, u32 rep[3], const u8* input, size_t blockStart, size_t srcSize,
                          u32* hashLong, u32 hBitsL, u32* hashSmall, u32 hBitsS, u32 mls)
{
    return dfast_block_low(ss, rep, input, blockStart, srcSize, hashLong, hBitsL, hashSmall, hBitsS, mls, IDX0, 0xFFFFFFFFu);
}
static size_t dfast_block_low(seqstore* ss, u32 rep[3], const u8* input, size_t blockStart, size_t srcSize,
                              u32* hashLong, u32 hBitsL, u32* hashSmall, u32 hBitsS, u32 mls, u32 dictLimit, u32 maxDist)
{
    u32 const endIndex = (u32)(blockStart + srcSize) + IDX0;
    u32 const prefixLowestIndex = (endIndex - dictLimit > maxDist) ? endIndex - maxDist : dictLimit;
    const u8* const base = input - IDX0;
    const u8* const src = input + blockStart;
    const u8* const istart = src;
    const u8* anchor = istart;
    const u8* const prefixLowest = base + prefixLowestIndex;
    const u8* const iend = istart + srcSize;
    const u8* const ilimit = iend - 8;
    u32 offset_1 = rep[0], offset_2 = rep[1];
    u32 offsetSaved1 = 0, offsetSaved2 = 0;
    size_t mLength; u32 offset; u32 curr = 0;
    size_t const kStepIncr = 1 << 8;
    const u8* nextStep; size_t step;
    size_t hl0, hl1; u32 idxl0, idxl1;
    const u8 *matchl0, *matchs0, *matchl1;
    const u8* ip = istart; const u8* ip1;

    if (srcSize < 8) return srcSize;  /* ilimit would precede istart */

    ip += ((ip - prefixLowest) == 0);
    {
        u32 const current = (u32)(ip - base);
        u32 const windowLow = (current - dictLimit > maxDist) ? current - maxDist : dictLimit;
        u32 const maxRep = current - windowLow;
        if (offset_2 > maxRep) { offsetSaved2 = offset_2; offset_2 = 0; }
        if (offset_1 > maxRep) { offsetSaved1 = offset_1; offset_1 = 0; }
    }

    for (;;) {
        step = 1;
        nextStep = ip + kStepIncr;
        ip1 = ip + step;
        if (ip1 > ilimit) goto _cleanup;

        hl0 = hash_long(ip, hBitsL);
        idxl0 = hashLong[hl0]; KREF_EV(0, hl0, 0);
        matchl0 = base + idxl0;

        do {
            size_t const hs0 = hash_short(ip, hBitsS, mls);
            u32 const idxs0 = hashSmall[hs0]; KREF_EV(1, hs0, 0);
            curr = (u32)(ip - base);
            matchs0 = base + idxs0;

            hashLong[hl0] = hashSmall[hs0] = curr; KREF_EV(0, hl0, 1); KREF_EV(1, hs0, 1);

            /* repcode at ip+1 */
            if ((offset_1 > 0) & (rd32(ip + 1 - offset_1) == rd32(ip + 1))) {
                mLength = count_eq(ip + 1 + 4, ip + 1 + 4 - offset_1, iend) + 4;
                ip++;
                store_seq(ss, (size_t)(ip - anchor), anchor, 1 /*REPCODE1*/, mLength);
                goto _match_stored;
            }

            hl1 = hash_long(ip1, hBitsL);

            /* long match at ip */
            if (idxl0 >= prefixLowestIndex && rd64(matchl0) == rd64(ip)) {
                mLength = count_eq(ip + 8, matchl0 + 8, iend) + 8;
                offset = (u32)(ip - matchl0);
                while (((ip > anchor) & (matchl0 > prefixLowest)) && (ip[-1] == matchl0[-1])) { ip--; matchl0--; mLength++; }
                goto _match_found;
            }

            idxl1 = hashLong[hl1]; KREF_EV(0, hl1, 0);
            matchl1 = base + idxl1;

            /* short match at ip */
            if (idxs0 >= prefixLowestIndex && rd32(matchs0) == rd32(ip)) goto _search_next_long;

            if (ip1 >= nextStep) { step++; nextStep += kStepIncr; }
            ip = ip1;
            ip1 += step;
            hl0 = hl1; idxl0 = idxl1; matchl0 = matchl1;
        } while (ip1 <= ilimit);

_cleanup:
        offsetSaved2 = ((offsetSaved1 != 0) && (offset_1 != 0)) ? offsetSaved1 : offsetSaved2;
        rep[0] = offset_1 ? offset_1 : offsetSaved1;
        rep[1] = offset_2 ? offset_2 : offsetSaved2;
        return (size_t)(iend - anchor);

_search_next_long:
        mLength = count_eq(ip + 4, matchs0 + 4, iend) + 4;
        offset = (u32)(ip - matchs0);
        if ((idxl1 > prefixLowestIndex) && (rd64(matchl1) == rd64(ip1))) {
            size_t const l1len = count_eq(ip1 + 8, matchl1 + 8, iend) + 8;
            if (l1len > mLength) {
                ip = ip1; mLength = l1len; offset = (u32)(ip - matchl1); matchs0 = matchl1;
            }
        }
        while (((ip > anchor) & (matchs0 > prefixLowest)) && (ip[-1] == matchs0[-1])) { ip--; matchs0--; mLength++; }

_match_found:
        offset_2 = offset_1;
        offset_1 = offset;
        if (step < 4) { hashLong[hl1] = (u32)(ip1 - base); KREF_EV(0, hl1, 2); }
        store_seq(ss, (size_t)(ip - anchor), anchor, offset + 3, mLength);

_match_stored:
        ip += mLength;
        anchor = ip;

        if (ip <= ilimit) {
            {
                u32 const indexToInsert = curr + 2;
                hashLong[hash_long(base + indexToInsert, hBitsL)] = indexToInsert;
                hashLong[hash_long(ip - 2, hBitsL)] = (u32)(ip - 2 - base);
                hashSmall[hash_short(base + indexToInsert, hBitsS, mls)] = indexToInsert;
                hashSmall[hash_short(ip - 1, hBitsS, mls)] = (u32)(ip - 1 - base);
                KREF_EV(0, hash_long(base + indexToInsert, hBitsL), 3); KREF_EV(0, hash_long(ip - 2, hBitsL), 3);
                KREF_EV(1, hash_short(base + indexToInsert, hBitsS, mls), 3); KREF_EV(1, hash_short(ip - 1, hBitsS, mls), 3);
            }
            while ((ip <= ilimit) && ((offset_2 > 0) & (rd32(ip) == rd32(ip - offset_2)))) {
                size_t const rLength = count_eq(ip + 4, ip + 4 - offset_2, iend) + 4;
                u32 const tmpOff = offset_2; offset_2 = offset_1; offset_1 = tmpOff;
                hashSmall[hash_short(ip, hBitsS, mls)] = (u32)(ip - base);
                hashLong[hash_long(ip, hBitsL)] = (u32)(ip - base);
                KREF_EV(1, hash_short(ip, hBitsS, mls), 4); KREF_EV(0, hash_long(ip, hBitsL), 4);
                store_seq(ss, 0, anchor, 1 /*REPCODE1*/, rLength);
                ip += rLength;
                anchor = ip;
            }
        }
    }
}

/* ------------------------------------------------------------------ */
/* forward bit writer (little-endian, LSB first) with end mark         */
/* ------------------------------------------------------------------ */
typedef struct { u64 acc; u32 nb; u8* start; u8* p; u8* end; int overflow; } bitw;

static void bw_init(bitw* b, u8* dst, size_t cap) { b->acc = 0; b->nb = 0; b->start = b->p = dst; b->end = dst + cap; b->overflow = 0; }
static inline void bw_flush(bitw* b)
{
    while (b->nb >= 8) {
        if (b->p < b->end) *b->p = (u8)b->acc; else b->overflow = 1;
        b->p++; b->acc >>= 8; b->nb -= 8;
    }
}
static inline void bw_add(bitw* b, u64 v, u32 n)
{
    if (n == 0) return;
    v &= (n >= 64) ? ~0ULL : ((1ULL << n) - 1);
    b->acc |= v << b->nb; b->nb += n;
    bw_flush(b);   /* n <= 32 here, acc has headroom: nb < 8 before add */
}
/* returns byte size, 0 on overflow */
static size_t bw_close(bitw* b)
{
    bw_add(b, 1, 1);
    bw_flush(b);
    if (b->nb) { if (b->p < b->end) *b->p = (u8)b->acc; else b->overflow = 1; b->p++; b->nb = 0; }
    if (b->overflow) return 0;
    return (size_t)(b->p - b->start);
}

/* ------------------------------------------------------------------ */
/* FSE: table-log choice, normalisation, header, encoding table        */
/* ------------------------------------------------------------------ */
#define FSE_MIN_TABLELOG 5
#define FSE_MAX_TABLELOG 12
#define KERR ((size_t)-1)

static u32 fse_min_tablelog(size_t srcSize, u32 maxSymbolValue)
{
    u32 const minBitsSrc = hb32((u32)srcSize) + 1;
    u32 const minBitsSymbols = hb32(maxSymbolValue) + 2;
    return minBitsSrc < minBitsSymbols ? minBitsSrc : minBitsSymbols;
}
static u32 fse_optimal_tablelog(u32 maxTableLog, size_t srcSize, u32 maxSymbolValue, u32 minus)
{
    u32 const maxBitsSrc = hb32((u32)(srcSize - 1)) - minus;
    u32 tableLog = maxTableLog;
    u32 const minBits = fse_min_tablelog(srcSize, maxSymbolValue);
    if (tableLog == 0) tableLog = 11;
    if (maxBitsSrc < tableLog) tableLog = maxBitsSrc;
    if (minBits > tableLog) tableLog = minBits;
    if (tableLog < FSE_MIN_TABLELOG) tableLog = FSE_MIN_TABLELOG;
    if (tableLog > FSE_MAX_TABLELOG) tableLog = FSE_MAX_TABLELOG;
    return tableLog;
}

static size_t fse_normalize_m2(short* norm, u32 tableLog, const u32* count, size_t total, u32 maxSymbolValue, short lowProbCount)
{
    short const NOT_YET_ASSIGNED = -2;
    u32 s, distributed = 0, ToDistribute;
    u32 const lowThreshold = (u32)(total >> tableLog);
    u32 lowOne = (u32)((total * 3) >> (tableLog + 1));

    for (s = 0; s <= maxSymbolValue; s++) {
        if (count[s] == 0) { norm[s] = 0; continue; }
        if (count[s] <= lowThreshold) { norm[s] = lowProbCount; distributed++; total -= count[s]; continue; }
        if (count[s] <= lowOne) { norm[s] = 1; distributed++; total -= count[s]; continue; }
        norm[s] = NOT_YET_ASSIGNED;
    }
    ToDistribute = (1u << tableLog) - distributed;
    if (ToDistribute == 0) return 0;

    if ((total / ToDistribute) > lowOne) {
        lowOne = (u32)((total * 3) / (ToDistribute * 2));
        for (s = 0; s <= maxSymbolValue; s++) {
            if ((norm[s] == NOT_YET_ASSIGNED) && (count[s] <= lowOne)) { norm[s] = 1; distributed++; total -= count[s]; continue; }
        }
        ToDistribute = (1u << tableLog) - distributed;
    }

    if (distributed == maxSymbolValue + 1) {
        u32 maxV = 0, maxC = 0;
        for (s = 0; s <= maxSymbolValue; s++) if (count[s] > maxC) { maxV = s; maxC = count[s]; }
        norm[maxV] += (short)ToDistribute;
        return 0;
    }
    if (total == 0) {
        for (s = 0; ToDistribute > 0; s = (s + 1) % (maxSymbolValue + 1)) if (norm[s] > 0) { ToDistribute--; norm[s]++; }
        return 0;
    }
    {
        u64 const vStepLog = 62 - tableLog;
        u64 const mid = (1ULL << (vStepLog - 1)) - 1;
        u64 const rStep = ((((u64)1 << vStepLog) * ToDistribute) + mid) / (u32)total;
        u64 tmpTotal = mid;
        for (s = 0; s <= maxSymbolValue; s++) {
            if (norm[s] == NOT_YET_ASSIGNED) {
                u64 const end = tmpTotal + (count[s] * rStep);
                u32 const sStart = (u32)(tmpTotal >> vStepLog);
                u32 const sEnd = (u32)(end >> vStepLog);
                u32 const weight = sEnd - sStart;
                if (weight < 1) return KERR;
                norm[s] = (short)weight;
                tmpTotal = end;
            }
        }
    }
    return 0;
}

/* returns tableLog, 0 for rle, KERR on error */
static size_t fse_normalize(short* norm, u32 tableLog, const u32* count, size_t total, u32 maxSymbolValue, u32 useLowProbCount)
{
    static u32 const rtbTable[] = { 0, 473195, 504333, 520860, 550000, 700000, 750000, 830000 };
    if (tableLog == 0) tableLog = 11;
    if (tableLog < FSE_MIN_TABLELOG) return KERR;
    if (tableLog > FSE_MAX_TABLELOG) return KERR;
    if (tableLog < fse_min_tablelog(total, maxSymbolValue)) return KERR;
    {
        short const lowProbCount = useLowProbCount ? -1 : 1;
        u64 const scale = 62 - tableLog;
        u64 const step = ((u64)1 << 62) / (u32)total;
        u64 const vStep = 1ULL << (scale - 20);
        int stillToDistribute = 1 << tableLog;
        u32 s, largest = 0; short largestP = 0;
        u32 const lowThreshold = (u32)(total >> tableLog);

        for (s = 0; s <= maxSymbolValue; s++) {
            if (count[s] == total) return 0;
            if (count[s] == 0) { norm[s] = 0; continue; }
            if (count[s] <= lowThreshold) { norm[s] = lowProbCount; stillToDistribute--; }
            else {
                short proba = (short)((count[s] * step) >> scale);
                if (proba < 8) {
                    u64 const restToBeat = vStep * rtbTable[proba];
                    proba += (count[s] * step) - ((u64)proba << scale) > restToBeat;
                }
                if (proba > largestP) { largestP = proba; largest = s; }
                norm[s] = proba;
                stillToDistribute -= proba;
            }
        }
        if (-stillToDistribute >= (norm[largest] >> 1)) {
            size_t const e = fse_normalize_m2(norm, tableLog, count, total, maxSymbolValue, lowProbCount);
            if (e == KERR) return KERR;
        } else norm[largest] += (short)stillToDistribute;
    }
    return tableLog;
}

/* writes the normalised-count header; dst must have room (caller gives >= 512) */
static size_t fse_write_ncount(u8* dst, size_t cap, const short* norm, u32 maxSymbolValue, u32 tableLog)
{
    u8* out = dst; u8* const oend = dst + cap;
    int nbBits; int const tableSize = 1 << tableLog;
    int remaining, threshold; u32 bitStream = 0; int bitCount = 0;
    u32 symbol = 0; u32 const alphabetSize = maxSymbolValue + 1; int previousIs0 = 0;

    bitStream += (tableLog - FSE_MIN_TABLELOG) << bitCount; bitCount += 4;
    remaining = tableSize + 1; threshold = tableSize; nbBits = (int)tableLog + 1;

    while ((symbol < alphabetSize) && (remaining > 1)) {
        if (previousIs0) {
            u32 start = symbol;
            while ((symbol < alphabetSize) && !norm[symbol]) symbol++;
            if (symbol == alphabetSize) break;
            while (symbol >= start + 24) {
                start += 24;
                bitStream += 0xFFFFU << bitCount;
                if (out > oend - 2) return KERR;
                out[0] = (u8)bitStream; out[1] = (u8)(bitStream >> 8); out += 2; bitStream >>= 16;
            }
            while (symbol >= start + 3) { start += 3; bitStream += 3U << bitCount; bitCount += 2; }
            bitStream += (symbol - start) << bitCount; bitCount += 2;
            if (bitCount > 16) {
                if (out > oend - 2) return KERR;
                out[0] = (u8)bitStream; out[1] = (u8)(bitStream >> 8); out += 2; bitStream >>= 16; bitCount -= 16;
            }
        }
        {
            int count = norm[symbol++];
            int const max = (2 * threshold - 1) - remaining;
            remaining -= count < 0 ? -count : count;
            count++;
            if (count >= threshold) count += max;
            bitStream += (u32)count << bitCount;
            bitCount += nbBits;
            bitCount -= (count < max);
            previousIs0 = (count == 1);
            if (remaining < 1) return KERR;
            while (remaining < threshold) { nbBits--; threshold >>= 1; }
        }
        if (bitCount > 16) {
            if (out > oend - 2) return KERR;
            out[0] = (u8)bitStream; out[1] = (u8)(bitStream >> 8); out += 2; bitStream >>= 16; bitCount -= 16;
        }
    }
    if (remaining != 1) return KERR;
    if (out > oend - 2) return KERR;
    out[0] = (u8)bitStream; out[1] = (u8)(bitStream >> 8);
    out += (bitCount + 7) / 8;
    return (size_t)(out - dst);
}

/* encoding table: per-symbol (deltaNbBits, deltaFindState) + next-state table */
typedef struct {
    u32 tableLog;
    u16 stateTable[1 << FSE_MAX_TABLELOG];
    u32 deltaNbBits[256];
    int deltaFindState[256];
} fse_ctable;

static void fse_build_ctable(fse_ctable* ct, const short* norm, u32 maxSymbolValue, u32 tableLog)
{
    u32 const tableSize = 1u << tableLog, tableMask = tableSize - 1;
    u32 const step = (tableSize >> 1) + (tableSize >> 3) + 3;
    u32 const maxSV1 = maxSymbolValue + 1;
    u16 cumul[258]; u8 tableSymbol[1 << FSE_MAX_TABLELOG];
    u32 highThreshold = tableSize - 1, u;

    ct->tableLog = tableLog;
    cumul[0] = 0;
    for (u = 1; u <= maxSV1; u++) {
        if (norm[u - 1] == -1) { cumul[u] = cumul[u - 1] + 1; tableSymbol[highThreshold--] = (u8)(u - 1); }
        else cumul[u] = cumul[u - 1] + (u16)norm[u - 1];
    }
    cumul[maxSV1] = (u16)(tableSize + 1);
    {
        u32 position = 0, symbol;
        for (symbol = 0; symbol < maxSV1; symbol++) {
            int n; int const freq = norm[symbol];
            for (n = 0; n < freq; n++) {
                tableSymbol[position] = (u8)symbol;
                position = (position + step) & tableMask;
                while (position > highThreshold) position = (position + step) & tableMask;
            }
        }
    }
    for (u = 0; u < tableSize; u++) { u8 const s = tableSymbol[u]; ct->stateTable[cumul[s]++] = (u16)(tableSize + u); }
    {
        u32 total = 0, s;
        for (s = 0; s <= maxSymbolValue; s++) {
            switch (norm[s]) {
            case 0: ct->deltaNbBits[s] = ((tableLog + 1) << 16) - (1u << tableLog); ct->deltaFindState[s] = 0; break;
            case -1: case 1:
                ct->deltaNbBits[s] = (tableLog << 16) - (1u << tableLog);
                ct->deltaFindState[s] = (int)(total - 1); total++; break;
            default: {
                u32 const maxBitsOut = tableLog - hb32((u32)norm[s] - 1);
                u32 const minStatePlus = (u32)norm[s] << maxBitsOut;
                ct->deltaNbBits[s] = (maxBitsOut << 16) - minStatePlus;
                ct->deltaFindState[s] = (int)(total - (u32)norm[s]);
                total += (u32)norm[s];
            } }
        }
    }
}
static void fse_build_ctable_rle(fse_ctable* ct, u8 symbol)
{
    ct->tableLog = 0; ct->stateTable[0] = 0; ct->stateTable[1] = 0;
    ct->deltaNbBits[symbol] = 0; ct->deltaFindState[symbol] = 0;
}
static inline u32 fse_init_state(const fse_ctable* ct, u32 symbol)
{
    u32 const dnb = ct->deltaNbBits[symbol];
    u32 const nbBitsOut = (dnb + (1u << 15)) >> 16;
    u32 const value = (nbBitsOut << 16) - dnb;
    return ct->stateTable[(value >> nbBitsOut) + ct->deltaFindState[symbol]];
}
static inline void fse_encode(bitw* b, const fse_ctable* ct, u32* state, u32 symbol)
{
    u32 const nbBitsOut = (*state + ct->deltaNbBits[symbol]) >> 16;
    bw_add(b, *state, nbBitsOut);
    *state = ct->stateTable[(*state >> nbBitsOut) + ct->deltaFindState[symbol]];
}

/* ------------------------------------------------------------------ */
/* Huffman literals                                                    */
/* ------------------------------------------------------------------ */
#define HUF_TABLELOG_MAX 12
#define LIT_HUF_LOG 11

typedef struct { u32 count; u16 parent; u8 byte; u8 nbBits; } hnode;
typedef struct { u16 curr, base; } rankpos;

#define RP_TABLE_SIZE 192
#define RP_LOG_BUCKETS_BEGIN 158
#define RP_DISTINCT_CUTOFF 165   /* 158 + highbit32(158)=7 */

static u32 huf_get_index(u32 count) { return (count < RP_DISTINCT_CUTOFF) ? count : hb32(count) + RP_LOG_BUCKETS_BEGIN; }
static void huf_swap(hnode* a, hnode* b) { hnode t = *a; *a = *b; *b = t; }
static void huf_insertion_sort(hnode* n, int low, int high)
{
    int i, size = high - low + 1; n += low;
    for (i = 1; i < size; ++i) {
        hnode const key = n[i]; int j = i - 1;
        while (j >= 0 && n[j].count < key.count) { n[j + 1] = n[j]; j--; }
        n[j + 1] = key;
    }
}
static int huf_partition(hnode* arr, int low, int high)
{
    u32 const pivot = arr[high].count; int i = low - 1, j = low;
    for (; j < high; j++) if (arr[j].count > pivot) { i++; huf_swap(&arr[i], &arr[j]); }
    huf_swap(&arr[i + 1], &arr[high]);
    return i + 1;
}
static void huf_quicksort(hnode* arr, int low, int high)
{
    if (high - low < 8) { huf_insertion_sort(arr, low, high); return; }
    while (low < high) {
        int const idx = huf_partition(arr, low, high);
        if (idx - low < high - idx) { huf_quicksort(arr, low, idx - 1); low = idx + 1; }
        else { huf_quicksort(arr, idx + 1, high); high = idx - 1; }
    }
}
static void huf_sort(hnode* huffNode, const u32* count, u32 maxSymbolValue)
{
    rankpos rp[RP_TABLE_SIZE]; u32 n; u32 const maxSV1 = maxSymbolValue + 1;
    memset(rp, 0, sizeof(rp));
    for (n = 0; n < maxSV1; ++n) rp[huf_get_index(count[n])].base++;
    for (n = RP_TABLE_SIZE - 1; n > 0; --n) { rp[n - 1].base += rp[n].base; rp[n - 1].curr = rp[n - 1].base; }
    for (n = 0; n < maxSV1; ++n) {
        u32 const c = count[n]; u32 const r = huf_get_index(c) + 1; u32 const pos = rp[r].curr++;
        huffNode[pos].count = c; huffNode[pos].byte = (u8)n;
    }
    for (n = RP_DISTINCT_CUTOFF; n < RP_TABLE_SIZE - 1; ++n) {
        int const bucketSize = rp[n].curr - rp[n].base; u32 const start = rp[n].base;
        if (bucketSize > 1) huf_quicksort(huffNode + start, 0, bucketSize - 1);
    }
}

static u32 huf_set_max_height(hnode* huffNode, u32 lastNonNull, u32 targetNbBits)
{
    u32 const largestBits = huffNode[lastNonNull].nbBits;
    if (largestBits <= targetNbBits) return largestBits;
    {
        int totalCost = 0; u32 const baseCost = 1u << (largestBits - targetNbBits);
        int n = (int)lastNonNull;
        while (huffNode[n].nbBits > targetNbBits) {
            totalCost += (int)(baseCost - (1u << (largestBits - huffNode[n].nbBits)));
            huffNode[n].nbBits = (u8)targetNbBits; n--;
        }
        while (huffNode[n].nbBits == targetNbBits) --n;
        totalCost >>= (largestBits - targetNbBits);
        {
            u32 const noSymbol = 0xF0F0F0F0; u32 rankLast[HUF_TABLELOG_MAX + 2];
            memset(rankLast, 0xF0, sizeof(rankLast));
            {
                u32 currentNbBits = targetNbBits; int pos;
                for (pos = n; pos >= 0; pos--) {
                    if (huffNode[pos].nbBits >= currentNbBits) continue;
                    currentNbBits = huffNode[pos].nbBits;
                    rankLast[targetNbBits - currentNbBits] = (u32)pos;
                }
            }
            while (totalCost > 0) {
                u32 nBitsToDecrease = hb32((u32)totalCost) + 1;
                for (; nBitsToDecrease > 1; nBitsToDecrease--) {
                    u32 const highPos = rankLast[nBitsToDecrease];
                    u32 const lowPos = rankLast[nBitsToDecrease - 1];
                    if (highPos == noSymbol) continue;
                    if (lowPos == noSymbol) break;
                    { u32 const highTotal = huffNode[highPos].count; u32 const lowTotal = 2 * huffNode[lowPos].count;
                      if (highTotal <= lowTotal) break; }
                }
                while ((nBitsToDecrease <= HUF_TABLELOG_MAX) && (rankLast[nBitsToDecrease] == noSymbol)) nBitsToDecrease++;
                totalCost -= 1 << (nBitsToDecrease - 1);
                huffNode[rankLast[nBitsToDecrease]].nbBits++;
                if (rankLast[nBitsToDecrease - 1] == noSymbol) rankLast[nBitsToDecrease - 1] = rankLast[nBitsToDecrease];
                if (rankLast[nBitsToDecrease] == 0) rankLast[nBitsToDecrease] = noSymbol;
                else {
                    rankLast[nBitsToDecrease]--;
                    if (huffNode[rankLast[nBitsToDecrease]].nbBits != targetNbBits - nBitsToDecrease) rankLast[nBitsToDecrease] = noSymbol;
                }
            }
            while (totalCost < 0) {
                if (rankLast[1] == noSymbol) {
                    while (huffNode[n].nbBits == targetNbBits) n--;
                    huffNode[n + 1].nbBits--;
                    rankLast[1] = (u32)(n + 1);
                    totalCost++;
                    continue;
                }
                huffNode[rankLast[1] + 1].nbBits--;
                rankLast[1]++;
                totalCost++;
            }
        }
    }
    return targetNbBits;
}

typedef struct { u16 val[256]; u8 nbBits[256]; } huf_ctable;

/* returns maxNbBits */
static u32 huf_build_ctable(huf_ctable* ct, const u32* count, u32 maxSymbolValue, u32 maxNbBits)
{
    hnode nodeTable[512 + 2]; hnode* const huffNode0 = nodeTable; hnode* const huffNode = huffNode0 + 1;
    int nonNullRank, lowS, lowN, nodeNb = 256, n, nodeRoot;
    memset(nodeTable, 0, sizeof(nodeTable));
    huf_sort(huffNode, count, maxSymbolValue);

    nonNullRank = (int)maxSymbolValue;
    while (huffNode[nonNullRank].count == 0) nonNullRank--;
    lowS = nonNullRank; nodeRoot = nodeNb + lowS - 1; lowN = nodeNb;
    huffNode[nodeNb].count = huffNode[lowS].count + huffNode[lowS - 1].count;
    huffNode[lowS].parent = huffNode[lowS - 1].parent = (u16)nodeNb;
    nodeNb++; lowS -= 2;
    for (n = nodeNb; n <= nodeRoot; n++) huffNode[n].count = (u32)(1U << 30);
    huffNode0[0].count = (u32)(1U << 31);
    while (nodeNb <= nodeRoot) {
        int const n1 = (huffNode[lowS].count < huffNode[lowN].count) ? lowS-- : lowN++;
        int const n2 = (huffNode[lowS].count < huffNode[lowN].count) ? lowS-- : lowN++;
        huffNode[nodeNb].count = huffNode[n1].count + huffNode[n2].count;
        huffNode[n1].parent = huffNode[n2].parent = (u16)nodeNb;
        nodeNb++;
    }
    huffNode[nodeRoot].nbBits = 0;
    for (n = nodeRoot - 1; n >= 256; n--) huffNode[n].nbBits = huffNode[huffNode[n].parent].nbBits + 1;
    for (n = 0; n <= nonNullRank; n++) huffNode[n].nbBits = huffNode[huffNode[n].parent].nbBits + 1;

    maxNbBits = huf_set_max_height(huffNode, (u32)nonNullRank, maxNbBits);
    {
        u16 nbPerRank[HUF_TABLELOG_MAX + 1] = { 0 }; u16 valPerRank[HUF_TABLELOG_MAX + 1] = { 0 };
        int const alphabetSize = (int)(maxSymbolValue + 1);
        for (n = 0; n <= nonNullRank; n++) nbPerRank[huffNode[n].nbBits]++;
        { u16 min = 0; for (n = (int)maxNbBits; n > 0; n--) { valPerRank[n] = min; min += nbPerRank[n]; min >>= 1; } }
        memset(ct, 0, sizeof(*ct));
        for (n = 0; n < alphabetSize; n++) ct->nbBits[huffNode[n].byte] = huffNode[n].nbBits;
        for (n = 0; n < alphabetSize; n++) ct->val[n] = valPerRank[ct->nbBits[n]]++;
    }
    return maxNbBits;
}

/* FSE-compress the Huffman weights; returns 0 if not compressible, 1 if single symbol */
static size_t huf_compress_weights(u8* dst, size_t dstSize, const u8* weightTable, size_t wtSize)
{
    u8* op = dst; u8* const oend = dst + dstSize;
    u32 maxSymbolValue = HUF_TABLELOG_MAX; u32 tableLog = 6;
    u32 count[HUF_TABLELOG_MAX + 1]; short norm[HUF_TABLELOG_MAX + 1]; fse_ctable ct;
    if (wtSize <= 1) return 0;
    {
        u32 maxCount = 0; size_t i; u32 s;
        memset(count, 0, sizeof(count));
        for (i = 0; i < wtSize; i++) count[weightTable[i]]++;
        while (!count[maxSymbolValue]) maxSymbolValue--;
        for (s = 0; s <= maxSymbolValue; s++) if (count[s] > maxCount) maxCount = count[s];
        if (maxCount == wtSize) return 1;
        if (maxCount == 1) return 0;
    }
    tableLog = fse_optimal_tablelog(tableLog, wtSize, maxSymbolValue, 2);
    { size_t const e = fse_normalize(norm, tableLog, count, wtSize, maxSymbolValue, 0); if (e == KERR) return KERR; }
    { size_t const h = fse_write_ncount(op, (size_t)(oend - op), norm, maxSymbolValue, tableLog); if (h == KERR) return KERR; op += h; }
    fse_build_ctable(&ct, norm, maxSymbolValue, tableLog);
    {
        /* FSE_compress_usingCTable: two interleaved states, symbols walked last -> first */
        bitw b; const u8* ip = weightTable + wtSize; u32 s1, s2; size_t n = wtSize; size_t cSize;
        if (n <= 2) return 0;
        bw_init(&b, op, (size_t)(oend - op));
        if (n & 1) { s1 = fse_init_state(&ct, *--ip); s2 = fse_init_state(&ct, *--ip); fse_encode(&b, &ct, &s1, *--ip); }
        else { s2 = fse_init_state(&ct, *--ip); s1 = fse_init_state(&ct, *--ip); }
        while (ip > weightTable) { fse_encode(&b, &ct, &s2, *--ip); fse_encode(&b, &ct, &s1, *--ip); }
        bw_add(&b, s2, tableLog); bw_add(&b, s1, tableLog);
        cSize = bw_close(&b);
        if (cSize == 0) return 0;
        op += cSize;
    }
    return (size_t)(op - dst);
}

static size_t huf_write_ctable(u8* dst, size_t maxDstSize, const huf_ctable* ct, u32 maxSymbolValue, u32 huffLog)
{
    u8 bitsToWeight[HUF_TABLELOG_MAX + 1]; u8 huffWeight[256]; u32 n;
    bitsToWeight[0] = 0;
    for (n = 1; n < huffLog + 1; n++) bitsToWeight[n] = (u8)(huffLog + 1 - n);
    for (n = 0; n < maxSymbolValue; n++) huffWeight[n] = bitsToWeight[ct->nbBits[n]];
    if (maxDstSize < 1) return KERR;
    {
        size_t const hSize = huf_compress_weights(dst + 1, maxDstSize - 1, huffWeight, maxSymbolValue);
        if (hSize == KERR) return KERR;
        if ((hSize > 1) & (hSize < maxSymbolValue / 2)) { dst[0] = (u8)hSize; return hSize + 1; }
    }
    if (maxSymbolValue > (256 - 128)) return KERR;
    if (((maxSymbolValue + 1) / 2) + 1 > maxDstSize) return KERR;
    dst[0] = (u8)(128 + (maxSymbolValue - 1));
    huffWeight[maxSymbolValue] = 0;
    for (n = 0; n < maxSymbolValue; n += 2) dst[(n / 2) + 1] = (u8)((huffWeight[n] << 4) + huffWeight[n + 1]);
    return ((maxSymbolValue + 1) / 2) + 1;
}

static size_t huf_encode_1x(u8* dst, size_t dstSize, const u8* src, size_t srcSize, const huf_ctable* ct)
{
    bitw b; size_t n;
    if (dstSize < 8) return 0;
    bw_init(&b, dst, dstSize);
    for (n = srcSize; n > 0; n--) bw_add(&b, ct->val[src[n - 1]], ct->nbBits[src[n - 1]]);
    return bw_close(&b);
}
static size_t huf_encode_4x(u8* dst, size_t dstSize, const u8* src, size_t srcSize, const huf_ctable* ct)
{
    size_t const segmentSize = (srcSize + 3) / 4;
    const u8* ip = src; const u8* const iend = src + srcSize;
    u8* op = dst; u8* const oend = dst + dstSize; int i;
    if (dstSize < 6 + 1 + 1 + 1 + 8) return 0;
    if (srcSize < 12) return 0;
    op += 6;
    for (i = 0; i < 3; i++) {
        size_t const c = huf_encode_1x(op, (size_t)(oend - op), ip, segmentSize, ct);
        if (c == 0 || c > 65535) return 0;
        wr16(dst + 2 * i, (u32)c); op += c; ip += segmentSize;
    }
    {
        size_t const c = huf_encode_1x(op, (size_t)(oend - op), ip, (size_t)(iend - ip), ct);
        if (c == 0 || c > 65535) return 0;
        op += c;
    }
    return (size_t)(op - dst);
}

/* HUF_compress{1X,4X}_repeat.  `old` = table of the previous compressed literals section of the frame
 * (NULL/oldValid 0 for the first one: HUF_repeat_none; afterwards libzstd holds HUF_repeat_check, never
 * _valid without a dictionary).  *usedOld is set when the old table was kept ("treeless" section);
 * when a new table is used it is copied to *old.  returns 0 = not compressible,
 * 1 = single symbol (rle), KERR = error, else compressed size (table + streams). */
static size_t huf_encode_with(u8* ostart, u8* op, u8* oend, const u8* src, size_t srcSize, int singleStream, const huf_ctable* ct)
{
    size_t const cSize = singleStream ? huf_encode_1x(op, (size_t)(oend - op), src, srcSize, ct)
                                      : huf_encode_4x(op, (size_t)(oend - op), src, srcSize, ct);
    if (cSize == 0) return 0;
    op += cSize;
    if ((size_t)(op - ostart) >= srcSize - 1) return 0;
    return (size_t)(op - ostart);
}
static size_t huf_compress(u8* dst, size_t dstSize, const u8* src, size_t srcSize, int singleStream, int suspectUncompressible,
                           huf_ctable* old, int oldValid, int preferRepeat, int* usedOld)
{
    u32 count[256]; u32 maxSymbolValue = 255; u32 huffLog = LIT_HUF_LOG; huf_ctable ct;
    u8* op = dst; u8* const oend = dst + dstSize; size_t i;
    int repeat = oldValid;          /* 0 HUF_repeat_none, 1 HUF_repeat_check (an earlier block's table), 2 HUF_repeat_valid (a dictionary's complete table) */
    *usedOld = 0;
    if (!srcSize) return 0;
    if (!dstSize) return 0;
    if (srcSize > 128 * 1024) return KERR;
    /* "If old table is valid, use it for small inputs": before anything is counted */
    if (preferRepeat && repeat == 2) { *usedOld = 1; return huf_encode_with(dst, op, oend, src, srcSize, singleStream, old); }

    if (suspectUncompressible && srcSize >= (4096 * 10)) {
        size_t largestTotal = 0; u32 c2[256]; u32 s, m;
        memset(c2, 0, sizeof(c2)); for (i = 0; i < 4096; i++) c2[src[i]]++;
        m = 0; for (s = 0; s < 256; s++) if (c2[s] > m) m = c2[s]; largestTotal += m;
        memset(c2, 0, sizeof(c2)); for (i = 0; i < 4096; i++) c2[src[srcSize - 4096 + i]]++;
        m = 0; for (s = 0; s < 256; s++) if (c2[s] > m) m = c2[s]; largestTotal += m;
        if (largestTotal <= ((2 * 4096) >> 7) + 4) return 0;
    }
    {
        u32 largest = 0, s;
        memset(count, 0, sizeof(count));
        for (i = 0; i < srcSize; i++) count[src[i]]++;
        while (!count[maxSymbolValue]) maxSymbolValue--;
        for (s = 0; s <= maxSymbolValue; s++) if (count[s] > largest) largest = count[s];
        if (largest == srcSize) { *dst = src[0]; return 1; }
        if (largest <= (srcSize >> 7) + 4) return 0;
    }
    /* HUF_validateCTable: every symbol present must have a code in the old table */
    if (repeat == 1) { u32 s; for (s = 0; s <= maxSymbolValue; s++) if (count[s] != 0 && old->nbBits[s] == 0) { repeat = 0; break; } }
    /* small inputs: keep the old table */
    if (preferRepeat && repeat) { *usedOld = 1; return huf_encode_with(dst, op, oend, src, srcSize, singleStream, old); }

    huffLog = fse_optimal_tablelog(huffLog, srcSize, maxSymbolValue, 1);
    huffLog = huf_build_ctable(&ct, count, maxSymbolValue, huffLog);
    {
        size_t const hSize = huf_write_ctable(op, dstSize, &ct, maxSymbolValue, huffLog);
        if (hSize == KERR) return KERR;
        if (repeat) {
            /* HUF_estimateCompressedSize of both tables */
            size_t oldBits = 0, newBits = 0; u32 s;
            for (s = 0; s <= maxSymbolValue; s++) { oldBits += (size_t)old->nbBits[s] * count[s]; newBits += (size_t)ct.nbBits[s] * count[s]; }
            if ((oldBits >> 3) <= hSize + (newBits >> 3) || hSize + 12 >= srcSize) {
                *usedOld = 1;
                return huf_encode_with(dst, op, oend, src, srcSize, singleStream, old);
            }
        }
        if (hSize + 12ul >= srcSize) return 0;
        op += hSize;
        if (old) *old = ct;          /* "Save new table" -- kept by the frame only if the block ends up compressed */
    }
    return huf_encode_with(dst, op, oend, src, srcSize, singleStream, &ct);
}

static size_t min_gain(size_t srcSize) { return (srcSize >> 6) + 2; }   /* strategy < btultra */

static size_t lit_raw(u8* dst, size_t cap, const u8* src, size_t srcSize)
{
    u32 const flSize = 1 + (srcSize > 31) + (srcSize > 4095);
    if (srcSize + flSize > cap) return KERR;
    switch (flSize) {
    case 1: dst[0] = (u8)(0 + (srcSize << 3)); break;
    case 2: wr16(dst, (u32)(0 + (1 << 2) + (srcSize << 4))); break;
    default: wr24(dst, (u32)(0 + (3 << 2) + (srcSize << 4))); break;
    }
    memcpy(dst + flSize, src, srcSize);
    return srcSize + flSize;
}
static size_t lit_rle(u8* dst, size_t cap, const u8* src, size_t srcSize)
{
    u32 const flSize = 1 + (srcSize > 31) + (srcSize > 4095);
    if (cap < 4) return KERR;
    switch (flSize) {
    case 1: dst[0] = (u8)(1 + (srcSize << 3)); break;
    case 2: wr16(dst, (u32)(1 + (1 << 2) + (srcSize << 4))); break;
    default: wr24(dst, (u32)(1 + (3 << 2) + (srcSize << 4))); break;
    }
    dst[flSize] = src[0];
    return flSize + 1;
}

/* ZSTD_compressLiterals.  prev/next: Huffman state before / after this block (next takes effect only if
 * the block is emitted compressed).  hufValid = a table exists (repeatMode HUF_repeat_check). */
typedef struct { huf_ctable ct; int valid; } kref_hufstate;

static __thread int g_lit_strategy = 2;          /* (ZSTD_compressLiterals: preferRepeat = strategy < ZSTD_lazy ? srcSize <= 1024 : 0) */
static size_t compress_literals(u8* dst, size_t cap, const u8* src, size_t srcSize, int suspectUncompressible,
                                const kref_hufstate* prev, kref_hufstate* next)
{
    size_t const lhSize = 3 + (srcSize >= 1024) + (srcSize >= 16384);
    int const singleStream = srcSize < 256 || (prev->valid == 2 && lhSize == 3);    /* (a valid table and a 3-byte header: one stream) */
    size_t cLitSize; int usedOld = 0; u32 hType = 2;
    *next = *prev;
    if (srcSize < (prev->valid == 2 ? 6u : 64u)) return lit_raw(dst, cap, src, srcSize);   /* ZSTD_minLiteralsToCompress(fast / dfast, repeatMode) */
    if (cap < lhSize + 1) return KERR;
    cLitSize = huf_compress(dst + lhSize, cap - lhSize, src, srcSize, singleStream, suspectUncompressible,
                            &next->ct, prev->valid, g_lit_strategy < 4 ? srcSize <= 1024 : 0, &usedOld);
    if (usedOld) hType = 3;       /* set_repeat */
    {
        size_t const minGain = min_gain(srcSize);
        if ((cLitSize == 0) || (cLitSize >= srcSize - minGain) || cLitSize == KERR) { *next = *prev; return lit_raw(dst, cap, src, srcSize); }
    }
    if (cLitSize == 1) {      /* one symbol -- or, below 8 bytes, possibly a one-byte stream: then the bytes decide */
        size_t k; int same = 1; for (k = 1; k < srcSize; k++) if (src[k] != src[0]) { same = 0; break; }
        if (srcSize >= 8 || same) { *next = *prev; return lit_rle(dst, cap, src, srcSize); }
    }
    if (hType == 2) next->valid = 1;            /* a newly built table: HUF_repeat_check for the next block */
    switch (lhSize) {
    case 3: wr24(dst, (u32)(hType + ((u32)(!singleStream) << 2) + ((u32)srcSize << 4) + ((u32)cLitSize << 14))); break;
    case 4: wr32(dst, (u32)(hType + (2 << 2) + ((u32)srcSize << 4) + ((u32)cLitSize << 18))); break;
    default: wr32(dst, (u32)(hType + (3 << 2) + ((u32)srcSize << 4) + ((u32)cLitSize << 22))); dst[4] = (u8)(cLitSize >> 10); break;
    }
    return lhSize + cLitSize;
}

/* ------------------------------------------------------------------ */
/* sequences section                                                   */
/* ------------------------------------------------------------------ */
static const u8 LL_bits[36] = { 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0, 1,1,1,1,2,2,3,3, 4,6,7,8,9,10,11,12, 13,14,15,16 };
static const u8 ML_bits[53] = { 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,
                                1,1,1,1,2,2,3,3, 4,4,5,7,8,9,10,11, 12,13,14,15,16 };
static const short LL_defaultNorm[36] = { 4,3,2,2,2,2,2,2, 2,2,2,2,2,1,1,1, 2,2,2,2,2,2,2,2, 2,3,2,1,1,1,1,1, -1,-1,-1,-1 };
static const short ML_defaultNorm[53] = { 1,4,3,2,2,2,2,2, 2,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1,
                                          1,1,1,1,1,1,1,1, 1,1,1,1,1,1,-1,-1, -1,-1,-1,-1,-1 };
static const short OF_defaultNorm[29] = { 1,1,1,1,1,1,2,2, 2,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1, -1,-1,-1,-1,-1 };

static u32 ll_code(u32 litLength)
{
    static const u8 LL_Code[64] = { 0,1,2,3,4,5,6,7, 8,9,10,11,12,13,14,15, 16,16,17,17,18,18,19,19,
        20,20,20,20,21,21,21,21, 22,22,22,22,22,22,22,22, 23,23,23,23,23,23,23,23,
        24,24,24,24,24,24,24,24, 24,24,24,24,24,24,24,24 };
    return (litLength > 63) ? hb32(litLength) + 19 : LL_Code[litLength];
}
static u32 ml_code(u32 mlBase)
{
    static const u8 ML_Code[128] = { 0,1,2,3,4,5,6,7, 8,9,10,11,12,13,14,15, 16,17,18,19,20,21,22,23, 24,25,26,27,28,29,30,31,
        32,32,33,33,34,34,35,35, 36,36,36,36,37,37,37,37, 38,38,38,38,38,38,38,38, 39,39,39,39,39,39,39,39,
        40,40,40,40,40,40,40,40, 40,40,40,40,40,40,40,40, 41,41,41,41,41,41,41,41, 41,41,41,41,41,41,41,41,
        42,42,42,42,42,42,42,42, 42,42,42,42,42,42,42,42, 42,42,42,42,42,42,42,42, 42,42,42,42,42,42,42,42 };
    return (mlBase > 127) ? hb32(mlBase) + 36 : ML_Code[mlBase];
}

enum { set_basic = 0, set_rle = 1, set_compressed = 2, set_repeat = 3 };

/* The sequence tables a formatted dictionary brings (ZSTD_loadCEntropy): normalised counts, and whether libzstd may use the table
 * without looking (FSE_repeat_valid: every symbol it could meet has a count) -- the only case the strategies below "lazy" reuse one. */
typedef struct { short norm[3][64]; u32 maxSym[3], log[3]; int valid[3]; } kref_seqprior;       /* [0] LL, [1] OF, [2] ML */
static __thread const kref_seqprior* g_seq_prior = NULL;

static u32 select_encoding_prior(const u32* count, u32 max, size_t mostFrequent, size_t nbSeq, u32 defaultNormLog, int isDefaultAllowed, int strategy, int priorValid);
static u32 select_encoding(const u32* count, u32 max, size_t mostFrequent, size_t nbSeq, u32 defaultNormLog, int isDefaultAllowed, int strategy)
{
    return select_encoding_prior(count, max, mostFrequent, nbSeq, defaultNormLog, isDefaultAllowed, strategy, 0);
}
static u32 select_encoding_prior(const u32* count, u32 max, size_t mostFrequent, size_t nbSeq, u32 defaultNormLog, int isDefaultAllowed, int strategy, int priorValid)
{
    (void)count; (void)max;
    if (mostFrequent == nbSeq) {
        if (isDefaultAllowed && nbSeq <= 2) return set_basic;
        return set_rle;
    }
    if (isDefaultAllowed) {   /* strategy fast(1) / dfast(2) < lazy */
        size_t const mult = 10 - (size_t)(strategy ? strategy : 2);
        size_t const dynamicFse_nbSeq_min = (((size_t)1 << defaultNormLog) * mult) >> 3;
        if (priorValid && nbSeq < 1000) return set_repeat;          /* staticFse_nbSeq_max */
        if ((nbSeq < dynamicFse_nbSeq_min) || (mostFrequent < (nbSeq >> (defaultNormLog - 1)))) return set_basic;
    }
    return set_compressed;
}

/* ZSTD_selectEncodingType from strategy "lazy" on: the three candidates are priced -- the default table (cross entropy), a table of
 * the block's own (its description + the entropy of the counts) -- and the cheaper one is taken (no previous table in a first block). */
static u32 inv_prob_log256(u32 x)        /* kInverseProbabilityLog256: (unsigned)(-log2(x / 256) * 256) */
{
    static u32 tab[256]; static int ready = 0;
    if (!ready) {
        /* -log2(x/256)*256 = 256 * (8 - log2 x), by integer arithmetic on a fixed-point log2 (no libm in the oracle's build) */
        u32 i;
        tab[0] = 0;
        for (i = 1; i < 256; i++) {
            /* log2(i) in 40-bit fixed point by repeated squaring */
            u64 m = (u64)i << 30; u32 e = 0; u64 frac = 0; int b;              /* m / 2^30 in [1, 2) */
            while (m >= ((u64)2 << 30)) { m >>= 1; e++; }
            for (b = 0; b < 40; b++) { m = (m * m) >> 30; frac <<= 1; if (m >= ((u64)2 << 30)) { m >>= 1; frac |= 1; } }
            {   /* value = 256 * (8 - e - frac / 2^40), floored */
                u64 const scaled = ((u64)(8 - e) << 40) - frac;          /* (8 - log2 i) in 40-bit fixed point */
                tab[i] = (u32)((scaled * 256) >> 40);
            }
        }
        ready = 1;
    }
    return tab[x];
}
static size_t entropy_cost(const u32* count, u32 max, size_t total)
{
    size_t cost = 0; u32 s;
    for (s = 0; s <= max; s++) { u32 norm = (u32)((256 * (u64)count[s]) / total); if (count[s] != 0 && norm == 0) norm = 1; cost += count[s] * inv_prob_log256(norm); }
    return cost >> 8;
}
static size_t cross_entropy_cost(const short* norm, u32 accuracyLog, const u32* count, u32 max)
{
    u32 const shift = 8 - accuracyLog; size_t cost = 0; u32 s;
    for (s = 0; s <= max; s++) { u32 const normAcc = (norm[s] != -1) ? (u32)norm[s] : 1; cost += count[s] * inv_prob_log256(normAcc << shift); }
    return cost >> 8;
}
/* the tables of the previous compressed block of a frame (ZSTD_fseCTables_t + repeat modes): from strategy "lazy" on, re-using one
 * is priced against the other two choices (ZSTD_fseBitCost).  mode: 0 none, 1 check (a table built for an earlier block). */
typedef struct { fse_ctable ct[3]; u32 maxSym[3]; int mode[3]; } kref_seqprev;
static __thread kref_seqprev* g_seq_prev = NULL;          /* set by the multi-block lazy frame loop; the block's choices are written to g_seq_next */
static __thread kref_seqprev* g_seq_next = NULL;
static size_t fse_bit_cost_table(const fse_ctable* ct, u32 ctMax, const u32* count, u32 max)
{
    size_t cost = 0; u32 s; u32 const tableLog = ct->tableLog;
    if (ctMax < max) return (size_t)-1;
    for (s = 0; s <= max; s++) {
        u32 const badCost = (tableLog + 1) << 8;
        u32 const minNbBits = ct->deltaNbBits[s] >> 16, threshold = (minNbBits + 1) << 16, tableSize = 1u << tableLog;
        u32 const deltaFromThreshold = threshold - (ct->deltaNbBits[s] + tableSize);
        u32 const bitCost = ((minNbBits + 1) << 8) - ((deltaFromThreshold << 8) >> tableLog);
        if (count[s] == 0) continue;
        if (bitCost >= badCost) return (size_t)-1;
        cost += (size_t)count[s] * bitCost;
    }
    return cost >> 8;
}
static u32 select_encoding_cost_prev(const u32* count, u32 max, size_t mostFrequent, size_t nbSeq, u32 FSELog, const short* defaultNorm, u32 defaultNormLog, int isDefaultAllowed, int which);
static u32 select_encoding_cost(const u32* count, u32 max, size_t mostFrequent, size_t nbSeq, u32 FSELog, const short* defaultNorm, u32 defaultNormLog, int isDefaultAllowed)
{
    if (mostFrequent == nbSeq) return (isDefaultAllowed && nbSeq <= 2) ? set_basic : set_rle;
    {
        size_t const basicCost = isDefaultAllowed ? cross_entropy_cost(defaultNorm, defaultNormLog, count, max) : (size_t)-1;
        short norm[64]; u8 wksp[512]; u32 cc[64]; u32 s; size_t ncount;
        u32 const tableLog = fse_optimal_tablelog(FSELog, nbSeq, max, 2);
        for (s = 0; s <= max; s++) cc[s] = count[s];
        if (fse_normalize(norm, tableLog, cc, nbSeq, max, nbSeq >= 2048) == KERR) return set_compressed;
        ncount = fse_write_ncount(wksp, sizeof(wksp), norm, max, tableLog);
        {
            size_t const compressedCost = (ncount << 3) + entropy_cost(count, max, nbSeq);
            if (basicCost <= compressedCost) return set_basic;
        }
    }
    return set_compressed;
}

/* ... with a previous block's table as the third candidate (which: 0 LL, 1 OF, 2 ML) */
static u32 select_encoding_cost_prev(const u32* count, u32 max, size_t mostFrequent, size_t nbSeq, u32 FSELog, const short* defaultNorm, u32 defaultNormLog, int isDefaultAllowed, int which)
{
    if (!g_seq_prev || !g_seq_prev->mode[which]) return select_encoding_cost(count, max, mostFrequent, nbSeq, FSELog, defaultNorm, defaultNormLog, isDefaultAllowed);
    if (mostFrequent == nbSeq) return (isDefaultAllowed && nbSeq <= 2) ? set_basic : set_rle;
    {
        size_t const basicCost = isDefaultAllowed ? cross_entropy_cost(defaultNorm, defaultNormLog, count, max) : (size_t)-1;
        size_t const repeatCost = fse_bit_cost_table(&g_seq_prev->ct[which], g_seq_prev->maxSym[which], count, max);
        short norm[64]; u8 wksp[512]; u32 cc[64]; u32 s; size_t ncount, compressedCost;
        u32 const tableLog = fse_optimal_tablelog(FSELog, nbSeq, max, 2);
        for (s = 0; s <= max; s++) cc[s] = count[s];
        if (fse_normalize(norm, tableLog, cc, nbSeq, max, nbSeq >= 2048) == KERR) return set_compressed;
        ncount = fse_write_ncount(wksp, sizeof(wksp), norm, max, tableLog);
        compressedCost = (ncount << 3) + entropy_cost(count, max, nbSeq);
        if (basicCost <= repeatCost && basicCost <= compressedCost) return set_basic;
        if (repeatCost <= compressedCost) return set_repeat;
    }
    return set_compressed;
}

/* returns header bytes written (KERR on error) and fills ct */
static size_t build_seq_ctable(u8* dst, size_t cap, fse_ctable* ct, u32 FSELog, u32 type, u32* count, u32 max,
                               const u8* codeTable, size_t nbSeq, const short* defaultNorm, u32 defaultNormLog, u32 defaultMax)
{
    switch (type) {
    case set_repeat:           /* (the caller has put the dictionary's table into ct) */
        return 0;
    case set_rle:
        fse_build_ctable_rle(ct, (u8)max);
        if (cap == 0) return KERR;
        *dst = codeTable[0];
        return 1;
    case set_basic:
        fse_build_ctable(ct, defaultNorm, defaultMax, defaultNormLog);
        return 0;
    default: {
        short norm[64]; size_t nbSeq_1 = nbSeq;
        u32 const tableLog = fse_optimal_tablelog(FSELog, nbSeq, max, 2);
        if (count[codeTable[nbSeq - 1]] > 1) { count[codeTable[nbSeq - 1]]--; nbSeq_1--; }
        if (fse_normalize(norm, tableLog, count, nbSeq_1, max, nbSeq_1 >= 2048) == KERR) return KERR;
        {
            size_t const NCountSize = fse_write_ncount(dst, cap, norm, max, tableLog);
            if (NCountSize == KERR) return KERR;
            fse_build_ctable(ct, norm, max, tableLog);
            return NCountSize;
        }
    } }
}

static size_t hist_codes(u32* count, u32* maxp, const u8* codes, size_t n)
{
    u32 max = *maxp, s; size_t i, largest = 0;
    memset(count, 0, (max + 1) * sizeof(u32));
    for (i = 0; i < n; i++) count[codes[i]]++;
    while (max > 0 && !count[max]) max--;
    *maxp = max;
    for (s = 0; s <= max; s++) if (count[s] > largest) largest = count[s];
    return largest;
}

/* returns section size; 0 means "give up, emit raw block"; KERR = error (dst too small) */
static size_t compress_sequences(u8* dst, size_t cap, const seqstore* ss)
{
    size_t const nbSeq = ss->nbSeq;
    u8* op = dst; u8* const oend = dst + cap;
    u8 *llCode, *ofCode, *mlCode; size_t i; size_t lastCountSize = 0;
    static __thread fse_ctable ctLL, ctOF, ctML;   /* (per thread: oracle/cpu_bench.c runs the restatement from several pthreads) */
    u32 count[64];

    if ((size_t)(oend - op) < 3 + 1) return KERR;
    if (nbSeq < 128) *op++ = (u8)nbSeq;
    else if (nbSeq < 0x7F00) { op[0] = (u8)((nbSeq >> 8) + 0x80); op[1] = (u8)nbSeq; op += 2; }
    else { op[0] = 0xFF; wr16(op + 1, (u32)(nbSeq - 0x7F00)); op += 3; }
    if (nbSeq == 0) return (size_t)(op - dst);

    llCode = (u8*)malloc(3 * nbSeq); ofCode = llCode + nbSeq; mlCode = ofCode + nbSeq;
    for (i = 0; i < nbSeq; i++) {
        llCode[i] = (u8)ll_code(ss->seqs[i].litLength);
        ofCode[i] = (u8)hb32(ss->seqs[i].offBase);
        mlCode[i] = (u8)ml_code(ss->seqs[i].mlBase);
    }
    if (ss->longLengthType == 1) llCode[ss->longLengthPos] = 35;
    if (ss->longLengthType == 2) mlCode[ss->longLengthPos] = 52;
    {
        u8* const seqHead = op++;
        u32 LLtype, Offtype, MLtype; size_t sz;
        { u32 max = 35; size_t const mf = hist_codes(count, &max, llCode, nbSeq);
          LLtype = ss->strategy >= 4 ? select_encoding_cost_prev(count, max, mf, nbSeq, 9, LL_defaultNorm, 6, 1, 0) : select_encoding_prior(count, max, mf, nbSeq, 6, 1, ss->strategy, g_seq_prior && g_seq_prior->valid[0]);
          if (LLtype == set_repeat) { if (ss->strategy >= 4 && g_seq_prev) ctLL = g_seq_prev->ct[0]; else fse_build_ctable(&ctLL, g_seq_prior->norm[0], g_seq_prior->maxSym[0], g_seq_prior->log[0]); }
          sz = build_seq_ctable(op, (size_t)(oend - op), &ctLL, 9, LLtype, count, max, llCode, nbSeq, LL_defaultNorm, 6, 35);
          if (sz == KERR) { free(llCode); return KERR; }
          if (LLtype == set_compressed) lastCountSize = sz;
          if (g_seq_next) { g_seq_next->ct[0] = ctLL; g_seq_next->mode[0] = (LLtype == set_compressed) ? 1 : (LLtype == set_repeat && g_seq_prev) ? g_seq_prev->mode[0] : 0; g_seq_next->maxSym[0] = (LLtype == set_repeat && g_seq_prev) ? g_seq_prev->maxSym[0] : max; }
          op += sz; }
        { u32 max = 31; size_t const mf = hist_codes(count, &max, ofCode, nbSeq);
          int const defaultAllowed = (max <= 28);
          Offtype = ss->strategy >= 4 ? select_encoding_cost_prev(count, max, mf, nbSeq, 8, OF_defaultNorm, 5, defaultAllowed, 1) : select_encoding_prior(count, max, mf, nbSeq, 5, defaultAllowed, ss->strategy, g_seq_prior && g_seq_prior->valid[1]);
          if (Offtype == set_repeat) { if (ss->strategy >= 4 && g_seq_prev) ctOF = g_seq_prev->ct[1]; else fse_build_ctable(&ctOF, g_seq_prior->norm[1], g_seq_prior->maxSym[1], g_seq_prior->log[1]); }
          sz = build_seq_ctable(op, (size_t)(oend - op), &ctOF, 8, Offtype, count, max, ofCode, nbSeq, OF_defaultNorm, 5, 28);
          if (sz == KERR) { free(llCode); return KERR; }
          if (Offtype == set_compressed) lastCountSize = sz;
          if (g_seq_next) { g_seq_next->ct[1] = ctOF; g_seq_next->mode[1] = (Offtype == set_compressed) ? 1 : (Offtype == set_repeat && g_seq_prev) ? g_seq_prev->mode[1] : 0; g_seq_next->maxSym[1] = (Offtype == set_repeat && g_seq_prev) ? g_seq_prev->maxSym[1] : max; }
          op += sz; }
        { u32 max = 52; size_t const mf = hist_codes(count, &max, mlCode, nbSeq);
          MLtype = ss->strategy >= 4 ? select_encoding_cost_prev(count, max, mf, nbSeq, 9, ML_defaultNorm, 6, 1, 2) : select_encoding_prior(count, max, mf, nbSeq, 6, 1, ss->strategy, g_seq_prior && g_seq_prior->valid[2]);
          if (MLtype == set_repeat) { if (ss->strategy >= 4 && g_seq_prev) ctML = g_seq_prev->ct[2]; else fse_build_ctable(&ctML, g_seq_prior->norm[2], g_seq_prior->maxSym[2], g_seq_prior->log[2]); }
          sz = build_seq_ctable(op, (size_t)(oend - op), &ctML, 9, MLtype, count, max, mlCode, nbSeq, ML_defaultNorm, 6, 52);
          if (sz == KERR) { free(llCode); return KERR; }
          if (MLtype == set_compressed) lastCountSize = sz;
          if (g_seq_next) { g_seq_next->ct[2] = ctML; g_seq_next->mode[2] = (MLtype == set_compressed) ? 1 : (MLtype == set_repeat && g_seq_prev) ? g_seq_prev->mode[2] : 0; g_seq_next->maxSym[2] = (MLtype == set_repeat && g_seq_prev) ? g_seq_prev->maxSym[2] : max; }
          op += sz; }
        *seqHead = (u8)((LLtype << 6) + (Offtype << 4) + (MLtype << 2));
    }
    {
        bitw b; u32 stML, stOF, stLL; size_t n; size_t streamSize;
        bw_init(&b, op, (size_t)(oend - op));
        stML = fse_init_state(&ctML, mlCode[nbSeq - 1]);
        stOF = fse_init_state(&ctOF, ofCode[nbSeq - 1]);
        stLL = fse_init_state(&ctLL, llCode[nbSeq - 1]);
        bw_add(&b, ss->seqs[nbSeq - 1].litLength, LL_bits[llCode[nbSeq - 1]]);
        bw_add(&b, ss->seqs[nbSeq - 1].mlBase, ML_bits[mlCode[nbSeq - 1]]);
        bw_add(&b, ss->seqs[nbSeq - 1].offBase, ofCode[nbSeq - 1]);
        for (n = nbSeq - 2; n < nbSeq; n--) {
            u8 const llc = llCode[n], ofc = ofCode[n], mlc = mlCode[n];
            fse_encode(&b, &ctOF, &stOF, ofc);
            fse_encode(&b, &ctML, &stML, mlc);
            fse_encode(&b, &ctLL, &stLL, llc);
            bw_add(&b, ss->seqs[n].litLength, LL_bits[llc]);
            bw_add(&b, ss->seqs[n].mlBase, ML_bits[mlc]);
            bw_add(&b, ss->seqs[n].offBase, ofc);
        }
        bw_add(&b, stML, ctML.tableLog);
        bw_add(&b, stOF, ctOF.tableLog);
        bw_add(&b, stLL, ctLL.tableLog);
        streamSize = bw_close(&b);
        free(llCode);
        if (streamSize == 0) return KERR;
        op += streamSize;
        if (lastCountSize && (lastCountSize + streamSize) < 4) return 0;
    }
    return (size_t)(op - dst);
}

/* ------------------------------------------------------------------ */
/* block pre-splitter (libzstd 1.5.7 zstd_preSplit.c, ZSTD_splitBlock   */
/* level 1 == "byChunks" with sampling rate 43 and the byte value as    */
/* the event: what ZSTD_optimalBlockSize picks for strategy dfast)      */
/* ------------------------------------------------------------------ */
typedef struct { u32 events[256]; size_t nbEvents; } kref_fp;

static void fp_record(kref_fp* fp, const u8* p, size_t srcSize)
{
    size_t const limit = srcSize - 2 + 1; size_t n;       /* HASHLENGTH 2 */
    memset(fp, 0, sizeof(*fp));
    for (n = 0; n < limit; n += 43) fp->events[p[n]]++;
    fp->nbEvents = limit / 43;
}
static int fp_differ(const kref_fp* ref, const kref_fp* nw, int penalty)
{
    u64 const p50 = (u64)ref->nbEvents * (u64)nw->nbEvents;
    u64 deviation = 0; u64 threshold; int n;
    for (n = 0; n < 256; n++) {
        int64_t const d = (int64_t)ref->events[n] * (int64_t)nw->nbEvents - (int64_t)nw->events[n] * (int64_t)ref->nbEvents;
        deviation += (u64)(d < 0 ? -d : d);
    }
    threshold = p50 * (u64)(14 + penalty) / 16;            /* THRESHOLD_BASE 14, THRESHOLD_PENALTY_RATE 16 */
    return deviation >= threshold;
}
static size_t split_block_by_chunks(const u8* p)            /* block of exactly 128 KiB */
{
    kref_fp past, nw; int penalty = 3; size_t pos; int n;
    size_t const blockSize = 128 << 10, chunk = 8 << 10;
    fp_record(&past, p, chunk);
    for (pos = chunk; pos <= blockSize - chunk; pos += chunk) {
        fp_record(&nw, p + pos, chunk);
        if (fp_differ(&past, &nw, penalty)) return pos;
        for (n = 0; n < 256; n++) past.events[n] += nw.events[n];
        past.nbEvents += nw.nbEvents;
        if (penalty > 0) penalty--;
    }
    return blockSize;
}
/* ZSTD_optimalBlockSize for level 3 */
KREF_API size_t kref_optimal_block_size(const u8* src, size_t remaining, int64_t savings)
{
    size_t const blockSizeMax = 128 << 10;
    if (remaining < blockSizeMax) return remaining;
    if (savings < 3) return blockSizeMax;
    return split_block_by_chunks(src);
}

/* ------------------------------------------------------------------ */
/* block + frame                                                       */
/* ------------------------------------------------------------------ */
typedef struct {
    u32* hashLong; u32* hashSmall; kref_seq* seqs; u8* lits;
} kref_wksp;

/* per-frame state carried from block to block (ZSTD_compressedBlockState_t): repcodes and the Huffman
 * table; the FSE tables never reach FSE_repeat_valid without a dictionary, so they are not state. */
typedef struct { u32 rep[3]; kref_hufstate huf; int isFirstBlock; } kref_frame_state;

/* ZSTD_compressBlock_internal for the block input[blockStart, +srcSize).
 * returns 0 => emit a raw block, 1 => RLE block, else the compressed-block body size. */
/* what the window looks like to one block (indices = stream position + 2): ext = 1 selects the extDict variant with the
 * older segment [dictStartIndex, prefixStartIndex), else the regular variant with its lowest valid index */
typedef struct { int ext; u32 dictStartIndex, prefixStartIndex, dictLimit, maxDist; } kref_blockwin;
static size_t compress_block_body_win(u8* dst, size_t cap, const u8* input, size_t blockStart, size_t srcSize, const u32* P,
                                      kref_wksp* w, kref_frame_state* fs, seqstore* ssOut, const kref_blockwin* win);
static size_t compress_block_body(u8* dst, size_t cap, const u8* input, size_t blockStart, size_t srcSize, const u32* P,
                                  kref_wksp* w, kref_frame_state* fs, seqstore* ssOut)
{
    return compress_block_body_win(dst, cap, input, blockStart, srcSize, P, w, fs, ssOut, NULL);
}
static size_t dfast_extdict_seg(seqstore* ss, u32 rep[3], const u8* istart, size_t srcSize, const u8* base, const u8* dictBase,
                                u32 dictStartIndex, u32 prefixStartIndex,
                                u32* hashLong, u32 hBitsL, u32* hashSmall, u32 hBitsS, u32 mls);
static size_t compress_block_body_win(u8* dst, size_t cap, const u8* input, size_t blockStart, size_t srcSize, const u32* P,
                                      kref_wksp* w, kref_frame_state* fs, seqstore* ssOut, const kref_blockwin* win)
{
    seqstore ss; u32 rep[3]; kref_hufstate nextHuf;
    const u8* const src = input + blockStart;
    size_t lastLL, litC, seqC, cSize;
    memset(&ss, 0, sizeof(ss)); ss.seqs = w->seqs; ss.lits = w->lits;
    if (srcSize < 2 + 3 + 1 + 1) { if (ssOut) *ssOut = ss; return 0; }   /* MIN_CBLOCK_SIZE + blockHeader + 1 + 1 */
    memcpy(rep, fs->rep, sizeof(rep));
    if (win && win->ext)          /* the whole stream is contiguous here, so both segments sit behind the same base */
        lastLL = dfast_extdict_seg(&ss, rep, src, srcSize, input - IDX0, input - IDX0, win->dictStartIndex, win->prefixStartIndex,
                                   w->hashLong, P[2], w->hashSmall, P[1], P[3]);
    else
        lastLL = dfast_block_low(&ss, rep, input, blockStart, srcSize, w->hashLong, P[2], w->hashSmall, P[1], P[3],
                                 win ? win->dictLimit : IDX0, win ? win->maxDist : 0xFFFFFFFFu);
    memcpy(ss.lits + ss.litSize, src + srcSize - lastLL, lastLL); ss.litSize += lastLL;
    if (ssOut) *ssOut = ss;
    cSize = 0;
    {
        int const suspect = (ss.nbSeq == 0) || (ss.litSize / ss.nbSeq >= 20);
        litC = compress_literals(dst, cap, ss.lits, ss.litSize, suspect, &fs->huf, &nextHuf);
        if (litC == KERR) { if (srcSize > cap) return KERR; goto _entropy_done; }
        seqC = compress_sequences(dst + litC, cap - litC, &ss);
        if (seqC == KERR) { if (srcSize > cap) return KERR; goto _entropy_done; }
        if (seqC == 0) goto _entropy_done;
        cSize = litC + seqC;
        { size_t const maxCSize = srcSize - min_gain(srcSize); if (cSize >= maxCSize) cSize = 0; }
    }
_entropy_done:
    /* a block that is one repeated byte becomes an RLE block, except the first block of a frame (ZSTD_compressBlock_internal: the test is on
     * the entropy stage's result, cSize < rleMaxLength = 25 -- 0 when the block would go out raw --, not on the sequence store:
     * ZSTD_maybeRLE belongs to the block splitter's path) */
    if (!fs->isFirstBlock && cSize < 25) {
        size_t i; int same = 1;
        for (i = 1; i < srcSize; i++) if (src[i] != src[0]) { same = 0; break; }
        if (same) { dst[0] = src[0]; cSize = 1; }
    }
    if (cSize > 1) { memcpy(fs->rep, rep, sizeof(rep)); fs->huf = nextHuf; }   /* confirmRepcodesAndEntropyTables */
    return cSize;
}

KREF_API size_t kref_compress_bound(size_t srcSize)
{
    return srcSize + (srcSize >> 8) + ((srcSize < (128 << 10)) ? (((128 << 10) - srcSize) >> 11) : 0);
}

static size_t write_frame_header_id(u8* dst, size_t srcSize, u32 windowLog, u32 dictID);
static size_t write_frame_header(u8* dst, size_t srcSize, u32 windowLog) { return write_frame_header_id(dst, srcSize, windowLog, 0); }
/* (ZSTD_writeFrameHeader: a formatted dictionary's ID follows the window descriptor in 1, 2 or 4 bytes) */
static size_t write_frame_header_id(u8* dst, size_t srcSize, u32 windowLog, u32 dictID)
{
    size_t pos = 0;
    u32 const windowSize = 1u << windowLog;
    u32 const singleSegment = (windowSize >= srcSize);
    u32 const fcsCode = (srcSize >= 256) + (srcSize >= 65536 + 256);
    u32 const idCode = (dictID > 0) + (dictID >= 256) + (dictID >= 65536);
    wr32(dst, 0xFD2FB528u); pos = 4;
    dst[pos++] = (u8)(idCode + (singleSegment << 5) + (fcsCode << 6));
    if (!singleSegment) dst[pos++] = (u8)((windowLog - 10) << 3);
    switch (idCode) {
    case 1: dst[pos++] = (u8)dictID; break;
    case 2: wr16(dst + pos, dictID); pos += 2; break;
    case 3: wr32(dst + pos, dictID); pos += 4; break;
    default: break;
    }
    switch (fcsCode) {
    case 0: if (singleSegment) dst[pos++] = (u8)srcSize; break;
    case 1: wr16(dst + pos, (u32)(srcSize - 256)); pos += 2; break;
    default: wr32(dst + pos, (u32)srcSize); pos += 4; break;
    }
    return pos;
}

#define KREF_MAX_SRC (2u << 20)       /* window never slides: srcSize <= 1 << windowLog */

static int wksp_alloc(kref_wksp* w, const u32* P)
{
    w->hashLong = (u32*)calloc((size_t)1 << P[2], sizeof(u32));
    w->hashSmall = (u32*)calloc((size_t)1 << P[1], sizeof(u32));
    w->seqs = (kref_seq*)malloc(sizeof(kref_seq) * ((128 << 10) / 3 + 8));
    w->lits = (u8*)malloc((128 << 10) + 32);
    return w->hashLong && w->hashSmall && w->seqs && w->lits;
}
static void wksp_free(kref_wksp* w) { free(w->hashLong); free(w->hashSmall); free(w->seqs); free(w->lits); }

/* One-shot level-3 frame (ZSTD_compress2 with the size known). Returns the frame size, or (size_t)-1 if
 * dst is too small or srcSize is outside this restatement's scope (> 2 MiB).  blockSizesOut (optional,
 * room for srcSize / 8192 + 2 entries) receives the sizes of the input blocks, *nbBlocksOut their number. */
static size_t compress_blocks_dfast(u8* dst, size_t cap, const u8* src, size_t srcSize, u32* blockSizesOut, u32* nbBlocksOut, const u32* Pin)
{
    u32 P[4]; kref_wksp w; kref_frame_state fs; size_t pos, ipos = 0; int64_t savings = 0; u32 nb = 0;
    if (nbBlocksOut) *nbBlocksOut = 0;
    if (srcSize > KREF_MAX_SRC) return KERR;
    if (cap < kref_compress_bound(srcSize)) return KERR;
    if (Pin) memcpy(P, Pin, sizeof(P)); else kref_params_l3(srcSize, P);
    pos = write_frame_header(dst, srcSize, P[0]);
    if (srcSize == 0) { wr24(dst + pos, 1); return pos + 3; }
    if (!wksp_alloc(&w, P)) { wksp_free(&w); return KERR; }
    fs.rep[0] = 1; fs.rep[1] = 4; fs.rep[2] = 8; fs.huf.valid = 0; memset(&fs.huf.ct, 0, sizeof(fs.huf.ct)); fs.isFirstBlock = 1;
    while (ipos < srcSize) {
        size_t const remaining = srcSize - ipos;
        size_t const blockSize = kref_optimal_block_size(src + ipos, remaining, savings);
        u32 const lastBlock = (blockSize == remaining);
        u8* const body = dst + pos + 3;
        size_t cSize = compress_block_body(body, cap - pos - 3, src, ipos, blockSize, P, &w, &fs, NULL);
        if (cSize == KERR) { wksp_free(&w); return KERR; }
        if (cSize == 0) {
            wr24(dst + pos, lastBlock + (0 << 1) + (u32)(blockSize << 3));
            memcpy(body, src + ipos, blockSize);
            cSize = 3 + blockSize;
        } else if (cSize == 1) {
            wr24(dst + pos, lastBlock + (1 << 1) + (u32)(blockSize << 3));
            cSize = 3 + 1;
        } else {
            wr24(dst + pos, lastBlock + (2 << 1) + (u32)(cSize << 3));
            cSize += 3;
        }
        savings += (int64_t)blockSize - (int64_t)cSize;
        if (blockSizesOut) blockSizesOut[nb] = (u32)blockSize;
        nb++;
        ipos += blockSize; pos += cSize; fs.isFirstBlock = 0;
    }
    wksp_free(&w);
    if (nbBlocksOut) *nbBlocksOut = nb;
    return pos;
}

KREF_API size_t kref_zstd_l3_compress_blocks(u8* dst, size_t cap, const u8* src, size_t srcSize, u32* blockSizesOut, u32* nbBlocksOut)
{
    return compress_blocks_dfast(dst, cap, src, srcSize, blockSizesOut, nbBlocksOut, NULL);
}
KREF_API size_t kref_zstd_l3_compress(u8* dst, size_t cap, const u8* src, size_t srcSize)
{
    return compress_blocks_dfast(dst, cap, src, srcSize, NULL, NULL, NULL);
}
/* Level 4 where it is double-fast (kref_params_l4): the same parse and entropy stage with that level's table sizes and
 * minimum match.  (size_t)-1 outside those size classes. */
KREF_API size_t kref_zstd_l4_compress(u8* dst, size_t cap, const u8* src, size_t srcSize)
{
    u32 P[4];
    if (srcSize == 0 || !kref_params_l4(srcSize, P)) return KERR;
    return compress_blocks_dfast(dst, cap, src, srcSize, NULL, NULL, P);
}

/* Stage taps for kernel-by-kernel diffing: the seqStore of the single block of a slice <= 128 KiB.
 * seqsOut: nbSeq x (offBase u32, litLength u16, mlBase u16); litsOut: literal bytes. */
KREF_API size_t kref_zstd_l3_seqstore(const u8* src, size_t srcSize, void* seqsOut, size_t* nbSeqOut,
                                      u8* litsOut, size_t* litSizeOut, int* longType, size_t* longPos)
{
    u32 P[4]; kref_wksp w; seqstore ss; kref_frame_state fs; u8* tmp; size_t cap = kref_compress_bound(srcSize) + 64; size_t r;
    if (srcSize > 131072) return KERR;
    kref_params_l3(srcSize, P);
    if (!wksp_alloc(&w, P)) { wksp_free(&w); return KERR; }
    fs.rep[0] = 1; fs.rep[1] = 4; fs.rep[2] = 8; fs.huf.valid = 0; memset(&fs.huf.ct, 0, sizeof(fs.huf.ct)); fs.isFirstBlock = 1;
    tmp = (u8*)malloc(cap);
    r = compress_block_body(tmp, cap, src, 0, srcSize, P, &w, &fs, &ss);
    memcpy(seqsOut, ss.seqs, ss.nbSeq * sizeof(kref_seq)); *nbSeqOut = ss.nbSeq;
    memcpy(litsOut, ss.lits, ss.litSize); *litSizeOut = ss.litSize;
    *longType = ss.longLengthType; *longPos = ss.longLengthPos;
    wksp_free(&w); free(tmp);
    return r;
}

/* ================================================================== */
/* Compressing with a raw-content dictionary                           */
/* (Kompressor: ZstdCompressor(level, dictionary) -> loadCompressor-    */
/* Dictionary, Wrapper.cpp:41-56 = ZSTD_CCtx_loadDictionary, then the   */
/* one-shot ZSTD_compressStream2(..., ZSTD_e_end)).                     */
/* libzstd 1.5.7 turns the dictionary into a CDict whose tables are     */
/* sized for the DICTIONARY (+513 bytes), then either copies those      */
/* tables into the context and parses with the "extDict" double-fast    */
/* variant (input above 16 KiB), or attaches the CDict and parses with  */
/* the "dictMatchState" variant (input up to 16 KiB).  One block        */
/* (input <= 128 KiB), dictionary of 8 bytes .. 128 KiB.                */
/* ================================================================== */
#define KREF_UNKNOWN ((u64)-1)
typedef struct { u32 W, C, H, mml; } kref_cpar;

static kref_cpar cpar_row(u64 rSize)
{
    kref_cpar p;
    if (rSize <= 16384)       { p.W = 14; p.C = 14; p.H = 15; p.mml = 4; }
    else if (rSize <= 131072) { p.W = 17; p.C = 15; p.H = 16; p.mml = 5; }
    else if (rSize <= 262144) { p.W = 18; p.C = 16; p.H = 16; p.mml = 4; }
    else                      { p.W = 21; p.C = 16; p.H = 17; p.mml = 5; }
    return p;
}
static u32 dict_and_window_log(u32 windowLog, u64 srcSize, u64 dictSize)
{
    if (dictSize == 0) return windowLog;
    {
        u64 const windowSize = 1ULL << windowLog;
        u64 const dictAndWindowSize = dictSize + windowSize;
        if (windowSize >= dictSize + srcSize) return windowLog;
        if (dictAndWindowSize >= (1ULL << 31)) return 31;
        return hb32((u32)dictAndWindowSize - 1) + 1;
    }
}
enum { CPM_NOATTACH = 0, CPM_ATTACH = 1, CPM_CREATECDICT = 2 };
/* ZSTD_adjustCParams_internal */
static kref_cpar adjust_cpar(kref_cpar p, u64 srcSize, u64 dictSize, int mode)
{
    if (mode == CPM_CREATECDICT && dictSize && srcSize == KREF_UNKNOWN) srcSize = 513;
    if (mode == CPM_ATTACH) dictSize = 0;
    if (srcSize <= (1ULL << 30) && dictSize <= (1ULL << 30)) {
        u32 const tSize = (u32)(srcSize + dictSize);
        u32 const srcLog = (tSize < 64) ? 6 : hb32(tSize - 1) + 1;
        if (p.W > srcLog) p.W = srcLog;
    }
    if (srcSize != KREF_UNKNOWN) {
        u32 const dawl = dict_and_window_log(p.W, srcSize, dictSize);
        if (p.H > dawl + 1) p.H = dawl + 1;
        if (p.C > dawl) p.C -= (p.C - dawl);
    }
    if (p.W < 10) p.W = 10;
    return p;
}
/* ZSTD_getCParams_internal(3, srcSizeHint, dictSize, mode) */
static kref_cpar get_cpar(u64 srcSizeHint, u64 dictSize, int mode)
{
    u64 rSize;
    if (mode == CPM_ATTACH) dictSize = 0;
    {
        int const unknown = srcSizeHint == KREF_UNKNOWN;
        u64 const added = (unknown && dictSize > 0) ? 500 : 0;
        rSize = (unknown && dictSize == 0) ? KREF_UNKNOWN : srcSizeHint + dictSize + added;
    }
    return adjust_cpar(cpar_row(rSize), srcSizeHint, dictSize, mode);
}

static size_t count_2segments(const u8* ip, const u8* match, const u8* iEnd, const u8* mEnd, const u8* iStart)
{
    const u8* const vEnd = (ip + (mEnd - match) < iEnd) ? ip + (mEnd - match) : iEnd;
    size_t const matchLength = count_eq(ip, match, vEnd);
    if (match + matchLength != mEnd) return matchLength;
    return matchLength + count_eq(ip + matchLength, iStart, iEnd);
}
static int index_overlap_check(u32 prefixLowestIndex, u32 repIndex) { return ((u32)((prefixLowestIndex - 1) - repIndex) >= 3); }

/* ZSTD_compressBlock_doubleFast_extDict_generic over the block [istart, istart + srcSize): indices below
 * prefixStartIndex live behind dictBase (the older segment, valid from dictStartIndex on), the others behind base. */
static size_t dfast_extdict_seg(seqstore* ss, u32 rep[3], const u8* istart, size_t srcSize, const u8* base, const u8* dictBase,
                                u32 dictStartIndex, u32 prefixStartIndex,
                                u32* hashLong, u32 hBitsL, u32* hashSmall, u32 hBitsS, u32 mls);
/* ... with a dictionary: dictionary = indices [2, 2 + D) behind dictBase, input from index 2 + D */
static size_t dfast_extdict(seqstore* ss, u32 rep[3], const u8* src, size_t srcSize, const u8* dict, size_t D,
                            u32* hashLong, u32 hBitsL, u32* hashSmall, u32 hBitsS, u32 mls)
{
    u32 const prefixStartIndex = (u32)(IDX0 + D);
    return dfast_extdict_seg(ss, rep, src, srcSize, src - prefixStartIndex, dict - IDX0, IDX0, prefixStartIndex, hashLong, hBitsL, hashSmall, hBitsS, mls);
}
static size_t dfast_extdict_seg(seqstore* ss, u32 rep[3], const u8* istart, size_t srcSize, const u8* base, const u8* dictBase,
                                u32 dictStartIndex, u32 prefixStartIndex,
                                u32* hashLong, u32 hBitsL, u32* hashSmall, u32 hBitsS, u32 mls)
{
    const u8* ip = istart; const u8* anchor = istart;
    const u8* const iend = istart + srcSize; const u8* const ilimit = iend - 8;
    const u8* const prefixStart = base + prefixStartIndex;
    const u8* const dictStart = dictBase + dictStartIndex;
    const u8* const dictEnd = dictBase + prefixStartIndex;
    u32 offset_1 = rep[0], offset_2 = rep[1];

    if (srcSize >= 8) while (ip < ilimit) {
        size_t const hSmall = hash_short(ip, hBitsS, mls);
        u32 const matchIndex = hashSmall[hSmall];
        const u8* const matchBase = matchIndex < prefixStartIndex ? dictBase : base;
        const u8* match = matchBase + matchIndex;
        size_t const hLong = hash_long(ip, hBitsL);
        u32 const matchLongIndex = hashLong[hLong];
        const u8* const matchLongBase = matchLongIndex < prefixStartIndex ? dictBase : base;
        const u8* matchLong = matchLongBase + matchLongIndex;
        u32 const curr = (u32)(ip - base);
        u32 const repIndex = curr + 1 - offset_1;
        const u8* const repBase = repIndex < prefixStartIndex ? dictBase : base;
        const u8* const repMatch = repBase + repIndex;
        size_t mLength;
        hashSmall[hSmall] = hashLong[hLong] = curr;

        if ((index_overlap_check(prefixStartIndex, repIndex) & (offset_1 <= curr + 1 - dictStartIndex))
            && (rd32(repMatch) == rd32(ip + 1))) {
            const u8* repMatchEnd = repIndex < prefixStartIndex ? dictEnd : iend;
            mLength = count_2segments(ip + 1 + 4, repMatch + 4, iend, repMatchEnd, prefixStart) + 4;
            ip++;
            store_seq(ss, (size_t)(ip - anchor), anchor, 1, mLength);
        } else {
            if ((matchLongIndex > dictStartIndex) && (rd64(matchLong) == rd64(ip))) {
                const u8* const matchEnd = matchLongIndex < prefixStartIndex ? dictEnd : iend;
                const u8* const lowMatchPtr = matchLongIndex < prefixStartIndex ? dictStart : prefixStart;
                u32 offset;
                mLength = count_2segments(ip + 8, matchLong + 8, iend, matchEnd, prefixStart) + 8;
                offset = curr - matchLongIndex;
                while (((ip > anchor) & (matchLong > lowMatchPtr)) && (ip[-1] == matchLong[-1])) { ip--; matchLong--; mLength++; }
                offset_2 = offset_1; offset_1 = offset;
                store_seq(ss, (size_t)(ip - anchor), anchor, offset + 3, mLength);
            } else if ((matchIndex > dictStartIndex) && (rd32(match) == rd32(ip))) {
                size_t const h3 = hash_long(ip + 1, hBitsL);
                u32 const matchIndex3 = hashLong[h3];
                const u8* const match3Base = matchIndex3 < prefixStartIndex ? dictBase : base;
                const u8* match3 = match3Base + matchIndex3;
                u32 offset;
                hashLong[h3] = curr + 1;
                if ((matchIndex3 > dictStartIndex) && (rd64(match3) == rd64(ip + 1))) {
                    const u8* const matchEnd = matchIndex3 < prefixStartIndex ? dictEnd : iend;
                    const u8* const lowMatchPtr = matchIndex3 < prefixStartIndex ? dictStart : prefixStart;
                    mLength = count_2segments(ip + 9, match3 + 8, iend, matchEnd, prefixStart) + 8;
                    ip++;
                    offset = curr + 1 - matchIndex3;
                    while (((ip > anchor) & (match3 > lowMatchPtr)) && (ip[-1] == match3[-1])) { ip--; match3--; mLength++; }
                } else {
                    const u8* const matchEnd = matchIndex < prefixStartIndex ? dictEnd : iend;
                    const u8* const lowMatchPtr = matchIndex < prefixStartIndex ? dictStart : prefixStart;
                    mLength = count_2segments(ip + 4, match + 4, iend, matchEnd, prefixStart) + 4;
                    offset = curr - matchIndex;
                    while (((ip > anchor) & (match > lowMatchPtr)) && (ip[-1] == match[-1])) { ip--; match--; mLength++; }
                }
                offset_2 = offset_1; offset_1 = offset;
                store_seq(ss, (size_t)(ip - anchor), anchor, offset + 3, mLength);
            } else {
                ip += ((ip - anchor) >> 8) + 1;
                continue;
            }
        }
        ip += mLength;
        anchor = ip;
        if (ip <= ilimit) {
            u32 const indexToInsert = curr + 2;
            hashLong[hash_long(base + indexToInsert, hBitsL)] = indexToInsert;
            hashLong[hash_long(ip - 2, hBitsL)] = (u32)(ip - 2 - base);
            hashSmall[hash_short(base + indexToInsert, hBitsS, mls)] = indexToInsert;
            hashSmall[hash_short(ip - 1, hBitsS, mls)] = (u32)(ip - 1 - base);
            while (ip <= ilimit) {
                u32 const current2 = (u32)(ip - base);
                u32 const repIndex2 = current2 - offset_2;
                const u8* repMatch2 = repIndex2 < prefixStartIndex ? dictBase + repIndex2 : base + repIndex2;
                if ((index_overlap_check(prefixStartIndex, repIndex2) & (offset_2 <= current2 - dictStartIndex))
                    && (rd32(repMatch2) == rd32(ip))) {
                    const u8* const repEnd2 = repIndex2 < prefixStartIndex ? dictEnd : iend;
                    size_t const repLength2 = count_2segments(ip + 4, repMatch2 + 4, iend, repEnd2, prefixStart) + 4;
                    u32 const tmpOffset = offset_2; offset_2 = offset_1; offset_1 = tmpOffset;
                    store_seq(ss, 0, anchor, 1, repLength2);
                    hashSmall[hash_short(ip, hBitsS, mls)] = current2;
                    hashLong[hash_long(ip, hBitsL)] = current2;
                    ip += repLength2;
                    anchor = ip;
                    continue;
                }
                break;
            }
        }
    }
    rep[0] = offset_1; rep[1] = offset_2;
    return (size_t)(iend - anchor);
}

/* ZSTD_compressBlock_doubleFast_dictMatchState_generic: the CDict keeps its own (tagged) tables dl / ds; the working
 * tables start empty; index space as above (dictIndexDelta = 0) */
static size_t dfast_dms(seqstore* ss, u32 rep[3], const u8* src, size_t srcSize, const u8* dict, size_t D,
                        u32* hashLong, u32 hBitsL, u32* hashSmall, u32 hBitsS, u32 mls,
                        const u32* dictHashLong, u32 dictHBitsL, const u32* dictHashSmall, u32 dictHBitsS)
{
    const u8* const istart = src; const u8* ip = istart; const u8* anchor = istart;
    const u8* const iend = istart + srcSize; const u8* const ilimit = iend - 8;
    u32 const prefixLowestIndex = (u32)(IDX0 + D);
    const u8* const base = src - prefixLowestIndex;
    const u8* const prefixLowest = base + prefixLowestIndex;
    const u8* const dictBase = dict - IDX0;
    const u8* const dictStart = dictBase + IDX0;
    const u8* const dictEnd = dictBase + prefixLowestIndex;
    u32 const dictIndexDelta = 0;
    u32 const dictAndPrefixLength = (u32)((ip - prefixLowest) + (dictEnd - dictStart));
    u32 offset_1 = rep[0], offset_2 = rep[1];
    ip += (dictAndPrefixLength == 0);

    if (srcSize >= 8) while (ip < ilimit) {
        size_t mLength; u32 offset;
        size_t const h2 = hash_long(ip, hBitsL);
        size_t const h = hash_short(ip, hBitsS, mls);
        size_t const dictHashAndTagL = hash_long(ip, dictHBitsL);
        size_t const dictHashAndTagS = hash_short(ip, dictHBitsS, mls);
        u32 const dictMatchIndexAndTagL = dictHashLong[dictHashAndTagL >> 8];
        u32 const dictMatchIndexAndTagS = dictHashSmall[dictHashAndTagS >> 8];
        int const dictTagsMatchL = (dictMatchIndexAndTagL & 0xFF) == (dictHashAndTagL & 0xFF);
        int const dictTagsMatchS = (dictMatchIndexAndTagS & 0xFF) == (dictHashAndTagS & 0xFF);
        u32 const curr = (u32)(ip - base);
        u32 const matchIndexL = hashLong[h2];
        u32 matchIndexS = hashSmall[h];
        const u8* matchLong = base + matchIndexL;
        const u8* match = base + matchIndexS;
        u32 const repIndex = curr + 1 - offset_1;
        const u8* repMatch = (repIndex < prefixLowestIndex) ? dictBase + (repIndex - dictIndexDelta) : base + repIndex;
        hashLong[h2] = hashSmall[h] = curr;

        if (index_overlap_check(prefixLowestIndex, repIndex) && (rd32(repMatch) == rd32(ip + 1))) {
            const u8* repMatchEnd = repIndex < prefixLowestIndex ? dictEnd : iend;
            mLength = count_2segments(ip + 1 + 4, repMatch + 4, iend, repMatchEnd, prefixLowest) + 4;
            ip++;
            store_seq(ss, (size_t)(ip - anchor), anchor, 1, mLength);
            goto _match_stored;
        }
        if ((matchIndexL >= prefixLowestIndex) && (rd64(matchLong) == rd64(ip))) {
            mLength = count_eq(ip + 8, matchLong + 8, iend) + 8;
            offset = (u32)(ip - matchLong);
            while (((ip > anchor) & (matchLong > prefixLowest)) && (ip[-1] == matchLong[-1])) { ip--; matchLong--; mLength++; }
            goto _match_found;
        } else if (dictTagsMatchL) {
            u32 const dictMatchIndexL = dictMatchIndexAndTagL >> 8;
            const u8* dictMatchL = dictBase + dictMatchIndexL;
            if (dictMatchL > dictStart && rd64(dictMatchL) == rd64(ip)) {
                mLength = count_2segments(ip + 8, dictMatchL + 8, iend, dictEnd, prefixLowest) + 8;
                offset = (u32)(curr - dictMatchIndexL - dictIndexDelta);
                while (((ip > anchor) & (dictMatchL > dictStart)) && (ip[-1] == dictMatchL[-1])) { ip--; dictMatchL--; mLength++; }
                goto _match_found;
            }
        }
        if (matchIndexS > prefixLowestIndex) {
            if (rd32(match) == rd32(ip)) goto _search_next_long;
        } else if (dictTagsMatchS) {
            u32 const dictMatchIndexS = dictMatchIndexAndTagS >> 8;
            match = dictBase + dictMatchIndexS;
            matchIndexS = dictMatchIndexS + dictIndexDelta;
            if (match > dictStart && rd32(match) == rd32(ip)) goto _search_next_long;
        }
        ip += ((ip - anchor) >> 8) + 1;
        continue;

_search_next_long:
        {
            size_t const hl3 = hash_long(ip + 1, hBitsL);
            size_t const dictHashAndTagL3 = hash_long(ip + 1, dictHBitsL);
            u32 const matchIndexL3 = hashLong[hl3];
            u32 const dictMatchIndexAndTagL3 = dictHashLong[dictHashAndTagL3 >> 8];
            int const dictTagsMatchL3 = (dictMatchIndexAndTagL3 & 0xFF) == (dictHashAndTagL3 & 0xFF);
            const u8* matchL3 = base + matchIndexL3;
            hashLong[hl3] = curr + 1;
            if ((matchIndexL3 >= prefixLowestIndex) && (rd64(matchL3) == rd64(ip + 1))) {
                mLength = count_eq(ip + 9, matchL3 + 8, iend) + 8;
                ip++;
                offset = (u32)(ip - matchL3);
                while (((ip > anchor) & (matchL3 > prefixLowest)) && (ip[-1] == matchL3[-1])) { ip--; matchL3--; mLength++; }
                goto _match_found;
            } else if (dictTagsMatchL3) {
                u32 const dictMatchIndexL3 = dictMatchIndexAndTagL3 >> 8;
                const u8* dictMatchL3 = dictBase + dictMatchIndexL3;
                if (dictMatchL3 > dictStart && rd64(dictMatchL3) == rd64(ip + 1)) {
                    mLength = count_2segments(ip + 1 + 8, dictMatchL3 + 8, iend, dictEnd, prefixLowest) + 8;
                    ip++;
                    offset = (u32)(curr + 1 - dictMatchIndexL3 - dictIndexDelta);
                    while (((ip > anchor) & (dictMatchL3 > dictStart)) && (ip[-1] == dictMatchL3[-1])) { ip--; dictMatchL3--; mLength++; }
                    goto _match_found;
                }
            }
        }
        if (matchIndexS < prefixLowestIndex) {
            mLength = count_2segments(ip + 4, match + 4, iend, dictEnd, prefixLowest) + 4;
            offset = (u32)(curr - matchIndexS);
            while (((ip > anchor) & (match > dictStart)) && (ip[-1] == match[-1])) { ip--; match--; mLength++; }
        } else {
            mLength = count_eq(ip + 4, match + 4, iend) + 4;
            offset = (u32)(ip - match);
            while (((ip > anchor) & (match > prefixLowest)) && (ip[-1] == match[-1])) { ip--; match--; mLength++; }
        }

_match_found:
        offset_2 = offset_1; offset_1 = offset;
        store_seq(ss, (size_t)(ip - anchor), anchor, offset + 3, mLength);

_match_stored:
        ip += mLength;
        anchor = ip;
        if (ip <= ilimit) {
            u32 const indexToInsert = curr + 2;
            hashLong[hash_long(base + indexToInsert, hBitsL)] = indexToInsert;
            hashLong[hash_long(ip - 2, hBitsL)] = (u32)(ip - 2 - base);
            hashSmall[hash_short(base + indexToInsert, hBitsS, mls)] = indexToInsert;
            hashSmall[hash_short(ip - 1, hBitsS, mls)] = (u32)(ip - 1 - base);
            while (ip <= ilimit) {
                u32 const current2 = (u32)(ip - base);
                u32 const repIndex2 = current2 - offset_2;
                const u8* repMatch2 = repIndex2 < prefixLowestIndex ? dictBase + repIndex2 - dictIndexDelta : base + repIndex2;
                if (index_overlap_check(prefixLowestIndex, repIndex2) && (rd32(repMatch2) == rd32(ip))) {
                    const u8* const repEnd2 = repIndex2 < prefixLowestIndex ? dictEnd : iend;
                    size_t const repLength2 = count_2segments(ip + 4, repMatch2 + 4, iend, repEnd2, prefixLowest) + 4;
                    u32 const tmpOffset = offset_2; offset_2 = offset_1; offset_1 = tmpOffset;
                    store_seq(ss, 0, anchor, 1, repLength2);
                    hashSmall[hash_short(ip, hBitsS, mls)] = current2;
                    hashLong[hash_long(ip, hBitsL)] = current2;
                    ip += repLength2;
                    anchor = ip;
                    continue;
                }
                break;
            }
        }
    }
    rep[0] = offset_1; rep[1] = offset_2;
    return (size_t)(iend - anchor);
}

/* ZSTD_fillDoubleHashTableForCDict (dtlm_full): tagged entries (index << 8 | tag), tables of 1 << (log + 8) hash bits */
static void fill_cdict_tables(u32* hashLarge, u32 hLog, u32* hashSmall, u32 cLog, u32 mls, const u8* dict, size_t D, size_t from)
{
    const u8* const base = dict - IDX0;
    const u8* ip = dict + from;
    const u8* const iend = dict + D - 8;
    for (; ip + 3 - 1 <= iend; ip += 3) {
        u32 const curr = (u32)(ip - base); u32 i;
        for (i = 0; i < 3; ++i) {
            size_t const smHashAndTag = hash_short(ip + i, cLog + 8, mls);
            size_t const lgHashAndTag = hash_long(ip + i, hLog + 8);
            if (i == 0) hashSmall[smHashAndTag >> 8] = ((curr + i) << 8) | (u32)(smHashAndTag & 0xFF);
            if (i == 0 || hashLarge[lgHashAndTag >> 8] == 0) hashLarge[lgHashAndTag >> 8] = ((curr + i) << 8) | (u32)(lgHashAndTag & 0xFF);
        }
    }
}

/* One-shot level-3 frame with a raw-content dictionary; srcSize <= 128 KiB, 8 <= dictSize <= 128 KiB.
 * modeOut (optional): 1 = CDict attached (dictMatchState parse), 0 = CDict tables copied (extDict parse). */
/* ---- formatted dictionaries (magic EC30A437): ZSTD_loadCEntropy + the content behind it ----------------------------
 * Layout: magic, dictID, a Huffman table description (as in a literals section), the offset / match-length / literal-length
 * tables as FSE normalised counts, three repeat offsets, the content.  A CCtx that loads one starts its first block with
 * these tables as "previous block" (so small inputs use them without describing tables of their own) and with these
 * repeat offsets; matches are searched in the content as with a raw-content dictionary; the frame header names the ID. */
typedef struct { const u8* p; size_t size; size_t bit; } fwdbits;
static u32 fb_peek(const fwdbits* b, u32 n)
{
    u64 v = 0; size_t const byte = b->bit >> 3; u32 k;
    for (k = 0; k < 8 && byte + k < b->size; k++) v |= (u64)b->p[byte + k] << (8 * k);
    return (u32)((v >> (b->bit & 7)) & ((1ull << n) - 1));
}
/* FSE_readNCount; returns the bytes read, 0 on error */
static size_t fse_read_ncount(short* norm, u32* maxSymbolValuePtr, u32* tableLogPtr, const u8* p, size_t size, u32 maxLog)
{
    fwdbits b; u32 tableLog, nbBits, sym = 0, maxSV = *maxSymbolValuePtr, s; int remaining, threshold; int previous0 = 0;
    b.p = p; b.size = size; b.bit = 0;
    if (size < 1) return 0;
    tableLog = fb_peek(&b, 4) + 5; b.bit += 4;
    if (tableLog > maxLog) return 0;
    *tableLogPtr = tableLog;
    remaining = (1 << tableLog) + 1; threshold = 1 << tableLog; nbBits = tableLog + 1;
    while (remaining > 1 && sym <= maxSV) {
        if (previous0) {
            for (;;) {
                u32 const r = fb_peek(&b, 2); u32 i; b.bit += 2;
                for (i = 0; i < r && sym <= maxSV; i++) norm[sym++] = 0;
                if (r != 3) break;
                if ((b.bit >> 3) > size) return 0;
            }
            if (sym > maxSV) break;
        }
        {
            int const max = (2 * threshold - 1) - remaining; int count;
            u32 const v = fb_peek(&b, nbBits);
            if ((int)(v & (u32)(threshold - 1)) < max) { count = (int)(v & (u32)(threshold - 1)); b.bit += nbBits - 1; }
            else { count = (int)(v & (u32)(2 * threshold - 1)); if (count >= threshold) count -= max; b.bit += nbBits; }
            count--;
            remaining -= count < 0 ? -count : count;
            norm[sym++] = (short)count;
            previous0 = (count == 0);
            while (remaining < threshold) { nbBits--; threshold >>= 1; }
            if ((b.bit >> 3) > size) return 0;
        }
    }
    if (remaining != 1) return 0;
    if (((b.bit + 7) >> 3) > size) return 0;
    for (s = sym; s <= maxSV; s++) norm[s] = 0;
    *maxSymbolValuePtr = sym - 1;
    return (b.bit + 7) >> 3;
}
/* HUF_readStats + HUF_readCTable: the weights (direct nibbles, or FSE-coded with two interleaved states read backwards), the implied
 * last weight, then the canonical codes by rank.  Returns the bytes read (0 on error); *hasZero: a symbol below the last has no code. */
static size_t huf_read_ctable(huf_ctable* ct, u32* nbSymbolsOut, int* hasZero, const u8* p, size_t size)
{
    u8 w[256]; u32 nw = 0, total = 0, tableLog, rest, i; size_t used; u32 hb;
    memset(ct, 0, sizeof(*ct));
    if (size < 1) return 0;
    hb = p[0];
    if (hb >= 128) {
        nw = hb - 127; used = 1 + (nw + 1) / 2;
        if (used > size) return 0;
        for (i = 0; i < nw; i += 2) { w[i] = p[1 + i / 2] >> 4; if (i + 1 < 256) w[i + 1] = p[1 + i / 2] & 15; }
    } else {
        short norm[16]; u32 maxSV = 12, tl = 0; size_t h; u16 db[64]; u8 dc[64]; u16 symnext[16]; u8 tsym[64];
        used = 1 + hb;
        if (hb == 0 || used > size) return 0;
        h = fse_read_ncount(norm, &maxSV, &tl, p + 1, hb, 6);
        if (h == 0) return 0;
        {   /* decoding table: spread, then per state the new state's base and bit count */
            u32 const tableSize = 1u << tl, mask = tableSize - 1, step = (tableSize >> 1) + (tableSize >> 3) + 3; u32 high = tableSize - 1, pos = 0, sy, u;
            for (sy = 0; sy <= maxSV; sy++) { if (norm[sy] == -1) { tsym[high--] = (u8)sy; symnext[sy] = 1; } else symnext[sy] = (u16)norm[sy]; }
            for (sy = 0; sy <= maxSV; sy++) { int k; for (k = 0; k < norm[sy]; k++) { tsym[pos] = (u8)sy; pos = (pos + step) & mask; while (pos > high) pos = (pos + step) & mask; } }
            for (u = 0; u < tableSize; u++) { u32 const sy2 = tsym[u]; u32 const next = symnext[sy2]++; u32 const nb = tl - hb32(next); db[u] = (u16)((((next << nb) - tableSize) & 0xFFFu) | (nb << 12)); dc[u] = (u8)sy2; }
        }
        {   /* the stream, backwards from its last set bit */
            const u8* const sp = p + 1 + h; size_t const ssz = hb - h; long bits; u32 s1, s2;
            #define BB_PEEK(n_) ({ u32 r_ = 0; long k_; for (k_ = 0; k_ < (long)(n_); k_++) { long const bi_ = bits - (long)(n_) + k_; if (bi_ >= 0) r_ |= (u32)((sp[bi_ >> 3] >> (bi_ & 7)) & 1u) << k_; } r_; })
            if (ssz == 0 || sp[ssz - 1] == 0) return 0;
            bits = (long)(8 * (ssz - 1) + hb32(sp[ssz - 1]));
            if (bits < (long)(2 * tl)) return 0;
            s1 = BB_PEEK(tl); bits -= tl; s2 = BB_PEEK(tl); bits -= tl;
            for (;;) {
                u32 e, nb;
                if (nw >= 255) return 0;
                e = db[s1]; w[nw++] = dc[s1]; nb = e >> 12;
                if (bits < (long)nb) { if (nw >= 255) return 0; w[nw++] = dc[s2]; break; }
                s1 = (e & 0xFFFu) + BB_PEEK(nb); bits -= nb;
                if (nw >= 255) return 0;
                e = db[s2]; w[nw++] = dc[s2]; nb = e >> 12;
                if (bits < (long)nb) { if (nw >= 255) return 0; w[nw++] = dc[s1]; break; }
                s2 = (e & 0xFFFu) + BB_PEEK(nb); bits -= nb;
            }
            #undef BB_PEEK
        }
    }
    for (i = 0; i < nw; i++) { if (w[i] > 12) return 0; if (w[i]) total += 1u << (w[i] - 1); }
    if (total == 0) return 0;
    tableLog = hb32(total) + 1;
    if (tableLog > 12) return 0;                 /* HUF_TABLELOG_MAX */
    rest = (1u << tableLog) - total;
    if (rest & (rest - 1)) return 0;
    w[nw++] = (u8)(hb32(rest) + 1);
    {
        u32 nbPerRank[16], valPerRank[16]; u32 n; u16 min = 0;
        memset(nbPerRank, 0, sizeof(nbPerRank)); memset(valPerRank, 0, sizeof(valPerRank));
        *hasZero = 0;
        for (n = 0; n < nw; n++) { if (w[n] == 0) *hasZero = 1; ct->nbBits[n] = w[n] ? (u8)(tableLog + 1 - w[n]) : 0; nbPerRank[w[n] ? tableLog + 1 - w[n] : tableLog + 1]++; }
        for (n = tableLog; n > 0; n--) { valPerRank[n] = min; min = (u16)(min + nbPerRank[n]); min >>= 1; }
        for (n = 0; n < nw; n++) ct->val[n] = (u16)valPerRank[w[n] ? tableLog + 1 - w[n] : tableLog + 1]++;
    }
    *nbSymbolsOut = nw;
    return used;
}
typedef struct { u32 dictID; kref_hufstate huf; kref_seqprior seq; u32 rep[3]; size_t contentOff; } kref_dictinfo;
static int ncount_repeat_valid(const short* norm, u32 dictMax, u32 max) { u32 s; if (dictMax < max) return 0; for (s = 0; s <= max; s++) if (norm[s] == 0) return 0; return 1; }
/* 1 = formatted and loaded, 0 = raw content (no magic), -1 = formatted but corrupt (libzstd: dictionary_corrupted) */
static int load_formatted_dict(kref_dictinfo* di, const u8* dict, size_t dictSize)
{
    size_t pos = 8, h, content; u32 nsym = 0; int hasZero = 1; u32 offMax;
    memset(di, 0, sizeof(*di));
    if (dictSize < 8 || rd32(dict) != 0xEC30A437u) return 0;
    di->dictID = rd32(dict + 4);
    h = huf_read_ctable(&di->huf.ct, &nsym, &hasZero, dict + pos, dictSize - pos);
    if (h == 0) return -1;
    di->huf.valid = (!hasZero && nsym == 256) ? 2 : 1;
    pos += h;
    { u32 max = 31, lg = 0; h = fse_read_ncount(di->seq.norm[1], &max, &lg, dict + pos, dictSize - pos, 8); if (h == 0) return -1; di->seq.maxSym[1] = 31; di->seq.log[1] = lg; offMax = max; pos += h; }
    { u32 max = 52, lg = 0; h = fse_read_ncount(di->seq.norm[2], &max, &lg, dict + pos, dictSize - pos, 9); if (h == 0) return -1; di->seq.maxSym[2] = max; di->seq.log[2] = lg; di->seq.valid[2] = ncount_repeat_valid(di->seq.norm[2], max, 52); pos += h; }
    { u32 max = 35, lg = 0; h = fse_read_ncount(di->seq.norm[0], &max, &lg, dict + pos, dictSize - pos, 9); if (h == 0) return -1; di->seq.maxSym[0] = max; di->seq.log[0] = lg; di->seq.valid[0] = ncount_repeat_valid(di->seq.norm[0], max, 35); pos += h; }
    if (pos + 12 > dictSize) return -1;
    di->rep[0] = rd32(dict + pos); di->rep[1] = rd32(dict + pos + 4); di->rep[2] = rd32(dict + pos + 8); pos += 12;
    content = dictSize - pos;
    { u32 const offcodeMax = hb32((u32)content + (128u << 10)); di->seq.valid[1] = ncount_repeat_valid(di->seq.norm[1], offMax, offcodeMax < 31 ? offcodeMax : 31); }
    { int i; for (i = 0; i < 3; i++) if (di->rep[i] == 0 || di->rep[i] > content) return -1; }
    di->contentOff = pos;
    return 1;
}

/* Test infrastructure: a formatted dictionary put together from statistics the caller chooses -- a literal sample (its Huffman table;
 * byte values the sample lacks get no code: HUF_repeat_check), three code histograms (normalised as they are, zero counts kept:
 * FSE_repeat_check where a code is missing), repeat offsets and content.  ZDICT's own dictionaries give every symbol a count, so the
 * "check" paths need these.  Returns the dictionary's size, KERR when the tables do not fit the format. */
KREF_API size_t kref_build_dictionary(u8* dst, size_t cap, u32 dictID, const u8* litSample, size_t litSize,
                                      const u32* llCount, const u32* ofCount, const u32* mlCount, const u32* rep3, const u8* content, size_t contentSize)
{
    size_t pos = 8, h; u32 count[256]; u32 maxSV = 255, huffLog, t; size_t i;
    if (cap < 8 + 1024 + 12 + contentSize) return KERR;
    wr32(dst, 0xEC30A437u); wr32(dst + 4, dictID);
    memset(count, 0, sizeof(count));
    for (i = 0; i < litSize; i++) count[litSample[i]]++;
    while (maxSV > 0 && !count[maxSV]) maxSV--;
    { huf_ctable ct; huffLog = fse_optimal_tablelog(LIT_HUF_LOG, litSize, maxSV, 1); huffLog = huf_build_ctable(&ct, count, maxSV, huffLog);
      h = huf_write_ctable(dst + pos, cap - pos, &ct, maxSV, huffLog); if (h == KERR) return KERR; pos += h; }
    for (t = 0; t < 3; t++) {
        const u32* const c = t == 0 ? ofCount : t == 1 ? mlCount : llCount;       /* the format's order: offsets, match lengths, literal lengths */
        u32 const lim = t == 0 ? 31u : t == 1 ? 52u : 35u, maxLog = t == 0 ? 8u : 9u;
        u32 max = lim, lg; size_t total = 0; short norm[64]; u32 cc[64];
        for (i = 0; i <= lim; i++) { cc[i] = c[i]; total += c[i]; }
        while (max > 0 && !cc[max]) max--;
        if (total < 2) return KERR;
        lg = fse_optimal_tablelog(maxLog, total, max, 2);
        if (fse_normalize(norm, lg, cc, total, max, total >= 2048) == KERR) return KERR;
        h = fse_write_ncount(dst + pos, cap - pos, norm, max, lg); if (h == KERR) return KERR; pos += h;
    }
    wr32(dst + pos, rep3[0]); wr32(dst + pos + 4, rep3[1]); wr32(dst + pos + 8, rep3[2]); pos += 12;
    memcpy(dst + pos, content, contentSize);
    return pos + contentSize;
}

KREF_API size_t kref_zstd_l3_compress_dict(u8* dst, size_t cap, const u8* src, size_t srcSize, const u8* dict, size_t dictSize, int* modeOut)
{
    kref_cpar cd, fp; kref_wksp w; seqstore ss; u32 rep[3] = { 1, 4, 8 }; kref_hufstate h0, h1;
    u32 *dl, *ds; size_t pos, lastLL, litC, seqC, cSize = 0, loaded = dictSize, from = 0; u8* body; int attach;
    kref_dictinfo di; int formatted; size_t const fullSize = dictSize;
    if (srcSize > 131072 || dictSize < 8 || dictSize > 131072 || srcSize == 0) return KERR;
    if (cap < kref_compress_bound(srcSize)) return KERR;
    formatted = load_formatted_dict(&di, dict, dictSize);
    if (formatted < 0) return KERR;
    /* CDict: parameters for the dictionary alone (its whole size, entropy part included: what libzstd's parameter selection and
     * the frame's window see), the content loaded, tables filled */
    cd = get_cpar(KREF_UNKNOWN, fullSize, CPM_CREATECDICT);
    if (formatted) { dict += di.contentOff; dictSize -= di.contentOff; loaded = dictSize; rep[0] = di.rep[0]; rep[1] = di.rep[1]; rep[2] = di.rep[2]; }
    { size_t const maxDict = (size_t)1 << ((cd.H + 3 > cd.C + 1) ? cd.H + 3 : cd.C + 1); if (loaded > maxDict) { from = loaded - maxDict; } }
    dl = (u32*)calloc((size_t)1 << cd.H, sizeof(u32)); ds = (u32*)calloc((size_t)1 << cd.C, sizeof(u32));
    if (dictSize - from > 8) fill_cdict_tables(dl, cd.H, ds, cd.C, cd.mml, dict, dictSize, from);
    attach = srcSize <= 16 * 1024;                        /* attachDictSizeCutoffs[ZSTD_dfast] */
    if (modeOut) *modeOut = attach;
    fp = get_cpar(srcSize, fullSize, attach ? CPM_ATTACH : CPM_NOATTACH);      /* the frame's window */
    pos = write_frame_header_id(dst, srcSize, fp.W, formatted ? di.dictID : 0);
    body = dst + pos + 3;
    w.seqs = (kref_seq*)malloc(sizeof(kref_seq) * ((128 << 10) / 3 + 8)); w.lits = (u8*)malloc((128 << 10) + 32);
    memset(&ss, 0, sizeof(ss)); ss.seqs = w.seqs; ss.lits = w.lits;
    if (srcSize >= 7) {
        if (attach) {
            kref_cpar wp = adjust_cpar(cd, srcSize, fullSize, CPM_ATTACH);       /* working tables: resized for the input only */
            w.hashLong = (u32*)calloc((size_t)1 << wp.H, sizeof(u32)); w.hashSmall = (u32*)calloc((size_t)1 << wp.C, sizeof(u32));
            lastLL = dfast_dms(&ss, rep, src, srcSize, dict, dictSize, w.hashLong, wp.H, w.hashSmall, wp.C, wp.mml, dl, cd.H + 8, ds, cd.C + 8);
        } else {
            size_t i;
            w.hashLong = (u32*)calloc((size_t)1 << cd.H, sizeof(u32)); w.hashSmall = (u32*)calloc((size_t)1 << cd.C, sizeof(u32));
            for (i = 0; i < ((size_t)1 << cd.H); i++) w.hashLong[i] = dl[i] >> 8;      /* ZSTD_copyCDictTableIntoCCtx */
            for (i = 0; i < ((size_t)1 << cd.C); i++) w.hashSmall[i] = ds[i] >> 8;
            lastLL = dfast_extdict(&ss, rep, src, srcSize, dict, dictSize, w.hashLong, cd.H, w.hashSmall, cd.C, cd.mml);
        }
        memcpy(ss.lits + ss.litSize, src + srcSize - lastLL, lastLL); ss.litSize += lastLL;
        h0.valid = 0; memset(&h0.ct, 0, sizeof(h0.ct));
        if (formatted) h0 = di.huf;
        {
            int const suspect = (ss.nbSeq == 0) || (ss.litSize / ss.nbSeq >= 20);
            litC = compress_literals(body, cap - pos - 3, ss.lits, ss.litSize, suspect, &h0, &h1);
            if (litC != KERR) {
                g_seq_prior = formatted ? &di.seq : NULL;
                seqC = compress_sequences(body + litC, cap - pos - 3 - litC, &ss);
                g_seq_prior = NULL;
                if (seqC != KERR && seqC != 0) { cSize = litC + seqC; if (cSize >= srcSize - min_gain(srcSize)) cSize = 0; }
            }
        }
        free(w.hashLong); free(w.hashSmall);
    }
    free(w.seqs); free(w.lits); free(dl); free(ds);
    if (cSize == 0) { wr24(dst + pos, 1 + (0 << 1) + (u32)(srcSize << 3)); memcpy(body, src, srcSize); return pos + 3 + srcSize; }
    wr24(dst + pos, 1 + (2 << 1) + (u32)(cSize << 3));
    return pos + 3 + cSize;
}


/* ================================================================== */
/* Levels 1 and 2: strategy "fast" (one hash table), slices <= 128 KiB  */
/* (Ktor's ZstdContentEncoder default is level 1: kompressor-zstd-ktor  */
/* ZstdContentEncoder.kt:11).                         */
/* libzstd 1.5.7 ZSTD_compressBlock_fast_noDict_generic: positions are  */
/* searched in adjacent pairs, the repcode is tried two positions ahead */
/* of the first of a pair, the pair distance grows after 128 bytes      */
/* without a match.                                                     */
/* ================================================================== */
KREF_API void kref_params_fast(int level, size_t srcSize, u32* out4)   /* windowLog, (unused), hashLog, minMatch */
{
    u32 W, H, mml;
    if (level < 0) {                       /* row 0 of libzstd's tables ("fast" with targetLength = -level: kref_fast_step) */
        if (srcSize <= 16384)        { W = 14; H = 13; mml = 5; }
        else if (srcSize <= 131072)  { W = 17; H = 12; mml = 5; }
        else if (srcSize <= 262144)  { W = 18; H = 13; mml = 5; }
        else                         { W = 19; H = 13; mml = 6; }
    } else
    if (level == 1) {
        if (srcSize <= 16384)        { W = 14; H = 15; mml = 5; }
        else if (srcSize <= 131072)  { W = 17; H = 13; mml = 6; }
        else if (srcSize <= 262144)  { W = 18; H = 14; mml = 6; }
        else                         { W = 19; H = 14; mml = 7; }
    } else {
        if (srcSize <= 16384)        { W = 14; H = 15; mml = 4; }
        else if (srcSize <= 131072)  { W = 17; H = 15; mml = 5; }
        else if (srcSize <= 262144)  { W = 18; H = 14; mml = 5; }
        else                         { W = 20; H = 16; mml = 6; }
    }
    {
        u32 const srcLog = (srcSize < 64) ? 6 : hb32((u32)(srcSize - 1)) + 1;
        if (W > srcLog) W = srcLog;
        if (H > W + 1) H = W + 1;
        if (W < 10) W = 10;
    }
    out4[0] = W; out4[1] = 0; out4[2] = H; out4[3] = mml;
}

/* stepSize of ZSTD_compressBlock_fast: targetLength + !targetLength + 1, targetLength = -level for negative levels, else 0 */
static size_t kref_fast_step(int level) { return level < 0 ? (size_t)(-(long)level) + 1 : 2; }

static size_t fast_block_step(seqstore* ss, u32 rep[3], const u8* src, size_t srcSize, u32* hashTable, u32 hlog, u32 mls, size_t stepSize)
{
    const u8* const base = src - IDX0;
    const u8* const istart = src;
    u32 const prefixStartIndex = IDX0;
    const u8* const prefixStart = base + prefixStartIndex;
    const u8* const iend = istart + srcSize;
    const u8* const ilimit = iend - 8;
    const u8* anchor = istart; const u8* ip0 = istart; const u8 *ip1, *ip2, *ip3;
    u32 current0 = 0;
    u32 rep_offset1 = rep[0], rep_offset2 = rep[1], offsetSaved1 = 0, offsetSaved2 = 0;
    size_t hash0, hash1; u32 matchIdx; u32 offcode; const u8* match0; size_t mLength;
    size_t step; const u8* nextStep; size_t const kStepIncr = 1 << 7;

    if (srcSize < 8) return srcSize;
    ip0 += (ip0 == prefixStart);
    {
        u32 const curr = (u32)(ip0 - base); u32 const maxRep = curr - prefixStartIndex;
        if (rep_offset2 > maxRep) { offsetSaved2 = rep_offset2; rep_offset2 = 0; }
        if (rep_offset1 > maxRep) { offsetSaved1 = rep_offset1; rep_offset1 = 0; }
    }
_start:
    step = stepSize;
    nextStep = ip0 + kStepIncr;
    ip1 = ip0 + 1; ip2 = ip0 + step; ip3 = ip2 + 1;
    if (ip3 >= ilimit) goto _cleanup;
    hash0 = hash_short(ip0, hlog, mls);
    hash1 = hash_short(ip1, hlog, mls);
    matchIdx = hashTable[hash0];
    do {
        u32 const rval = rd32(ip2 - rep_offset1);
        current0 = (u32)(ip0 - base);
        hashTable[hash0] = current0;
        if ((rd32(ip2) == rval) & (rep_offset1 > 0)) {
            ip0 = ip2;
            match0 = ip0 - rep_offset1;
            mLength = ip0[-1] == match0[-1];
            ip0 -= mLength; match0 -= mLength;
            offcode = 1;
            mLength += 4;
            hashTable[hash1] = (u32)(ip1 - base);
            goto _match;
        }
        if (matchIdx >= prefixStartIndex && rd32(base + matchIdx) == rd32(ip0)) {
            hashTable[hash1] = (u32)(ip1 - base);
            goto _offset;
        }
        matchIdx = hashTable[hash1];
        hash0 = hash1;
        hash1 = hash_short(ip2, hlog, mls);
        ip0 = ip1; ip1 = ip2; ip2 = ip3;
        current0 = (u32)(ip0 - base);
        hashTable[hash0] = current0;
        if (matchIdx >= prefixStartIndex && rd32(base + matchIdx) == rd32(ip0)) {
            if (step <= 4) hashTable[hash1] = (u32)(ip1 - base);
            goto _offset;
        }
        matchIdx = hashTable[hash1];
        hash0 = hash1;
        hash1 = hash_short(ip2, hlog, mls);
        ip0 = ip1; ip1 = ip2; ip2 = ip0 + step; ip3 = ip1 + step;
        if (ip2 >= nextStep) { step++; nextStep += kStepIncr; }
    } while (ip3 < ilimit);

_cleanup:
    offsetSaved2 = ((offsetSaved1 != 0) && (rep_offset1 != 0)) ? offsetSaved1 : offsetSaved2;
    rep[0] = rep_offset1 ? rep_offset1 : offsetSaved1;
    rep[1] = rep_offset2 ? rep_offset2 : offsetSaved2;
    return (size_t)(iend - anchor);

_offset:
    match0 = base + matchIdx;
    rep_offset2 = rep_offset1;
    rep_offset1 = (u32)(ip0 - match0);
    offcode = rep_offset1 + 3;
    mLength = 4;
    while (((ip0 > anchor) & (match0 > prefixStart)) && (ip0[-1] == match0[-1])) { ip0--; match0--; mLength++; }

_match:
    mLength += count_eq(ip0 + mLength, match0 + mLength, iend);
    store_seq(ss, (size_t)(ip0 - anchor), anchor, offcode, mLength);
    ip0 += mLength;
    anchor = ip0;
    if (ip0 <= ilimit) {
        hashTable[hash_short(base + current0 + 2, hlog, mls)] = current0 + 2;
        hashTable[hash_short(ip0 - 2, hlog, mls)] = (u32)(ip0 - 2 - base);
        if (rep_offset2 > 0) {
            while ((ip0 <= ilimit) && (rd32(ip0) == rd32(ip0 - rep_offset2))) {
                size_t const rLength = count_eq(ip0 + 4, ip0 + 4 - rep_offset2, iend) + 4;
                { u32 const tmpOff = rep_offset2; rep_offset2 = rep_offset1; rep_offset1 = tmpOff; }
                hashTable[hash_short(ip0, hlog, mls)] = (u32)(ip0 - base);
                ip0 += rLength;
                store_seq(ss, 0, anchor, 1, rLength);
                anchor = ip0;
            }
        }
    }
    goto _start;
}

/* One-shot frame at level 1 or 2 (strategy fast) or at a negative level ("fast" with a larger step and literals left
 * uncompressed: ZSTD_literalsCompressionIsDisabled), srcSize <= 128 KiB (one block). */
KREF_API size_t kref_zstd_fast_compress(u8* dst, size_t cap, const u8* src, size_t srcSize, int level)
{
    u32 P[4]; kref_wksp w; seqstore ss; u32 rep[3] = { 1, 4, 8 }; kref_hufstate h0, h1;
    size_t pos, lastLL, litC, seqC, cSize = 0; u8* body;
    if (srcSize > 131072 || (level != 1 && level != 2 && level >= 0) || level < -131072) return KERR;
    if (cap < kref_compress_bound(srcSize)) return KERR;
    kref_params_fast(level, srcSize, P);
    pos = write_frame_header(dst, srcSize, P[0]);
    if (srcSize == 0) { wr24(dst + pos, 1); return pos + 3; }
    body = dst + pos + 3;
    w.hashLong = (u32*)calloc((size_t)1 << P[2], sizeof(u32));
    w.seqs = (kref_seq*)malloc(sizeof(kref_seq) * ((128 << 10) / 3 + 8)); w.lits = (u8*)malloc((128 << 10) + 32);
    memset(&ss, 0, sizeof(ss)); ss.seqs = w.seqs; ss.lits = w.lits; ss.strategy = 1;
    if (srcSize >= 7) {
        lastLL = fast_block_step(&ss, rep, src, srcSize, w.hashLong, P[2], P[3], kref_fast_step(level));
        memcpy(ss.lits + ss.litSize, src + srcSize - lastLL, lastLL); ss.litSize += lastLL;
        h0.valid = 0; memset(&h0.ct, 0, sizeof(h0.ct));
        {
            int const suspect = (ss.nbSeq == 0) || (ss.litSize / ss.nbSeq >= 20);
            litC = (level < 0) ? lit_raw(body, cap - pos - 3, ss.lits, ss.litSize) : compress_literals(body, cap - pos - 3, ss.lits, ss.litSize, suspect, &h0, &h1);
            if (litC != KERR) {
                seqC = compress_sequences(body + litC, cap - pos - 3 - litC, &ss);
                if (seqC != KERR && seqC != 0) { cSize = litC + seqC; if (cSize >= srcSize - min_gain(srcSize)) cSize = 0; }
            }
        }
    }
    free(w.hashLong); free(w.seqs); free(w.lits);
    if (cSize == 0) { wr24(dst + pos, 1 + (0 << 1) + (u32)(srcSize << 3)); memcpy(body, src, srcSize); return pos + 3 + srcSize; }
    wr24(dst + pos, 1 + (2 << 1) + (u32)(cSize << 3));
    return pos + 3 + cSize;
}


/* ================================================================== */
/* Streaming frames: input fed with ZSTD_e_continue, then ZSTD_e_end    */
/* (the reference's kotlinx-io / Ktor callers: SliceTransformRawSource  */
/* .kt:32-55, BaseSliceTransformContentEncoder.kt:23-54, through        */
/* Wrapper.cpp:112 with finish = false).  The size is unknown when the   */
/* frame starts: level-3 parameters of the "unknown size" row (window    */
/* 2^21, hash 2^17, chain 2^16, minMatch 5), a header with a window      */
/* descriptor and no content size; the library buffers the input and     */
/* compresses it in chunks of 128 KiB, so the pre-splitter only ever     */
/* sees one chunk.  If the stream stops exactly at a chunk boundary and  */
/* the final call brings no data, an empty last block closes the frame.  */
/* Total input <= 2 MiB (the window never slides).                       */
/* ================================================================== */
KREF_API size_t kref_zstd_l3_compress_stream(u8* dst, size_t cap, const u8* src, size_t srcSize, int emptyEnd)
{
    u32 const P[4] = { 21, 16, 17, 5 };
    kref_wksp w; kref_frame_state fs; size_t pos, ipos = 0; int64_t savings = 0;
    size_t const blockSizeMax = 128 << 10;
    if (srcSize > KREF_MAX_SRC) return KERR;
    if (cap < kref_compress_bound(srcSize) + 16) return KERR;
    wr32(dst, 0xFD2FB528u); dst[4] = 0; dst[5] = (u8)((P[0] - 10) << 3); pos = 6;
    if (!wksp_alloc(&w, P)) { wksp_free(&w); return KERR; }
    fs.rep[0] = 1; fs.rep[1] = 4; fs.rep[2] = 8; fs.huf.valid = 0; memset(&fs.huf.ct, 0, sizeof(fs.huf.ct)); fs.isFirstBlock = 1;
    if (srcSize % blockSizeMax != 0 || srcSize == 0) emptyEnd = (srcSize == 0);
    while (ipos < srcSize) {
        size_t const chunkEnd = (ipos + blockSizeMax < srcSize) ? ipos + blockSizeMax : srcSize;     /* chunks start at multiples of 128 KiB */
        int const lastChunk = (chunkEnd == srcSize) && !emptyEnd;
        if (ipos != 0 && ipos % blockSizeMax == 0 && ipos == blockSizeMax) savings -= 6;             /* the frame header counts as produced from the second chunk on */
        while (ipos < chunkEnd) {
            size_t const remaining = chunkEnd - ipos;
            size_t const blockSize = kref_optimal_block_size(src + ipos, remaining, savings);
            u32 const lastBlock = lastChunk && (blockSize == remaining);
            u8* const body = dst + pos + 3;
            size_t cSize = compress_block_body(body, cap - pos - 3, src, ipos, blockSize, P, &w, &fs, NULL);
            if (cSize == KERR) { wksp_free(&w); return KERR; }
            if (cSize == 0) { wr24(dst + pos, lastBlock + (0 << 1) + (u32)(blockSize << 3)); memcpy(body, src + ipos, blockSize); cSize = 3 + blockSize; }
            else if (cSize == 1) { wr24(dst + pos, lastBlock + (1 << 1) + (u32)(blockSize << 3)); cSize = 3 + 1; }
            else { wr24(dst + pos, lastBlock + (2 << 1) + (u32)(cSize << 3)); cSize += 3; }
            savings += (int64_t)blockSize - (int64_t)cSize;
            ipos += blockSize; pos += cSize; fs.isFirstBlock = 0;
        }
    }
    wksp_free(&w);
    if (emptyEnd) { wr24(dst + pos, 1); pos += 3; }
    return pos;
}


/* ================================================================== */
/* Level-3 frames as the reference really obtains them above 128 KiB:  */
/* SliceTransform.transform(ByteArray) hands ZSTD_compressStream2 output */
/* slices of max(8192, n / 10) bytes (SliceTransform.kt:33-56), less     */
/* than ZSTD_compressBound(n), so libzstd does NOT compress the caller's */
/* array in place: it stages the input in its own buffer of              */
/* windowSize + 128 KiB bytes and compresses it in chunks of 128 KiB     */
/* (ZSTD_compressStream_generic, buffered mode).  Consequences restated  */
/* here: the block pre-splitter only sees one chunk; the frame header    */
/* counts as produced from the second chunk on; once the stream is longer */
/* than the buffer the chunks wrap to its start, the previous lap becomes */
/* an "extDict" segment that the new chunks overwrite from the front      */
/* (ZSTD_window_update), blocks are then parsed by the extDict variant    */
/* of the double-fast loop, and the window's low limit follows            */
/* ZSTD_window_enforceMaxDist / ZSTD_getLowestMatchIndex.                 */
/* knownSize = 1: finish = true from the first call (size pledged: header */
/* with content size, parameters by size); 0: data arrived with           */
/* finish = false first (kref_zstd_l3_compress_stream's frames, any length). */
/* One more thing follows from the output slices: whenever libzstd's       */
/* staging buffer is empty (right after a wrap) and the room left in the   */
/* CURRENT output slice is at least ZSTD_compressBound(what is left of the */
/* input it can see), it compresses that rest straight from the caller's   */
/* memory (ZSTD_compressEnd): one chunk however long, a new segment that   */
/* does not overwrite the staged lap.  outChunk = size of the output slices */
/* of the one-shot driver (max(8192, n / 10); every call fills its slice,   */
/* so the room at a wrap is outChunk - produced % outChunk); tailDirect =   */
/* for a stream, the bytes the closing call brought when that applied to    */
/* it (they must start on a lap boundary), else 0.                          */
/* knownSize = 2: no staging at all -- ZSTD_compress2 into a buffer of      */
/* ZSTD_compressBound bytes compresses the caller's array in place, as one  */
/* chunk (kref_zstd_l3_compress's frames, here for any length: beyond 2 MiB */
/* only the window's low limit moves).                                      */
/* ================================================================== */
typedef struct { u32 lowLimit, dictLimit; } kref_window;

static size_t dfast_compress_buffered(u8* dst, size_t cap, const u8* src, size_t srcSize, int knownSize, int emptyEnd,
                                      size_t outChunk, size_t tailDirect, const u32* Pknown);
KREF_API size_t kref_zstd_l3_compress_buffered(u8* dst, size_t cap, const u8* src, size_t srcSize, int knownSize, int emptyEnd,
                                               size_t outChunk, size_t tailDirect)
{ return dfast_compress_buffered(dst, cap, src, srcSize, knownSize, emptyEnd, outChunk, tailDirect, NULL); }
/* The same at level 4, where its rows are double-fast ones (kref_params_l4; a stream of unknown size: window 21, chain 18, hash 18). */
KREF_API size_t kref_zstd_l4_compress_buffered(u8* dst, size_t cap, const u8* src, size_t srcSize, int knownSize, int emptyEnd,
                                               size_t outChunk, size_t tailDirect)
{
    u32 P[4] = { 21, 18, 18, 5 };
    if (knownSize && (srcSize == 0 || !kref_params_l4(srcSize, P))) return KERR;
    return dfast_compress_buffered(dst, cap, src, srcSize, knownSize, emptyEnd, outChunk, tailDirect, P);
}

/* Pknown: the parameters of a frame whose size is known, when they are not level 3's (level 2 has a double-fast row for
 * 128 KiB < size <= 256 KiB: kref_zstd_fast_compress_big) */
static size_t dfast_compress_buffered(u8* dst, size_t cap, const u8* src, size_t srcSize, int knownSize, int emptyEnd,
                                      size_t outChunk, size_t tailDirect, const u32* Pknown)
{
    int tail = 0;
    u32 P[4]; kref_wksp w; kref_frame_state fs; size_t pos, ipos = 0, hdr; int64_t savings = 0;
    size_t const blockSizeMax = 128 << 10;
    kref_window win; size_t windowSize, inBuffSize, bufPos = 0, extBase = 0; u32 maxDist; int haveExt = 0;
    if (srcSize >= 0xF0000000u) return KERR;
    if (cap < kref_compress_bound(srcSize) + 16) return KERR;
    if (knownSize) { if (Pknown) memcpy(P, Pknown, sizeof(P)); else kref_params_l3(srcSize, P); pos = write_frame_header(dst, srcSize, P[0]); emptyEnd = 0; }
    else { if (Pknown) memcpy(P, Pknown, sizeof(P)); else { P[0] = 21; P[1] = 16; P[2] = 17; P[3] = 5; } wr32(dst, 0xFD2FB528u); dst[4] = 0; dst[5] = (u8)((P[0] - 10) << 3); pos = 6; }
    hdr = pos;
    if (knownSize && srcSize == 0) { wr24(dst + pos, 1); return pos + 3; }
    if (!wksp_alloc(&w, P)) { wksp_free(&w); return KERR; }
    fs.rep[0] = 1; fs.rep[1] = 4; fs.rep[2] = 8; fs.huf.valid = 0; memset(&fs.huf.ct, 0, sizeof(fs.huf.ct)); fs.isFirstBlock = 1;
    if (srcSize % blockSizeMax != 0 || srcSize == 0) emptyEnd = (!knownSize && srcSize == 0);
    maxDist = 1u << P[0];
    windowSize = (knownSize && srcSize < ((size_t)1 << P[0])) ? (srcSize ? srcSize : 1) : ((size_t)1 << P[0]);
    inBuffSize = windowSize + (blockSizeMax < windowSize ? blockSizeMax : windowSize);
    win.lowLimit = IDX0; win.dictLimit = IDX0;
    while (ipos < srcSize) {
        size_t chunkEnd, chunkLen; int lastChunk;
        /* the staging buffer is empty and the rest fits the output slice: compressed in place, as one chunk */
        if (ipos != 0 && bufPos == 0 && !tail) {
            if (knownSize && outChunk) {
                size_t const room = outChunk - pos % outChunk, r = srcSize - ipos;
                if (room >= r + (r >> 8) + (r < blockSizeMax ? (blockSizeMax - r) >> 11 : 0)) tail = 1;
            } else if (!knownSize && tailDirect && ipos + tailDirect == srcSize) tail = 1;
        }
        chunkEnd = (!tail && knownSize != 2 && ipos + blockSizeMax < srcSize) ? ipos + blockSizeMax : srcSize;     /* chunks start at multiples of 128 KiB */
        chunkLen = chunkEnd - ipos;
        lastChunk = (chunkEnd == srcSize) && !emptyEnd;
        if (ipos == blockSizeMax && knownSize != 2) savings -= (int64_t)hdr;                /* the frame header counts as produced from the second chunk on */
        /* ZSTD_window_update for the chunk at inBuff + bufPos */
        if (ipos != 0 && bufPos == 0) {                                     /* the chunk is not contiguous with the previous one: new segment */
            win.lowLimit = win.dictLimit;
            win.dictLimit = (u32)ipos + IDX0;
            if (win.dictLimit - win.lowLimit < 8) win.lowLimit = win.dictLimit;
            haveExt = 1;                                                    /* the older segment: stream position extBase sat at inBuff + 0 */
        }
        if (haveExt && !tail) {
            /* input and older segment overlap in the buffer: the overwritten front of the segment is given up */
            size_t const extLoPhys = (size_t)(win.lowLimit - IDX0) - extBase, extHiPhys = (size_t)(win.dictLimit - IDX0) - extBase;
            if (bufPos + chunkLen > extLoPhys && bufPos < extHiPhys) {
                size_t const high = extBase + bufPos + chunkLen + IDX0;
                win.lowLimit = high > win.dictLimit ? win.dictLimit : (u32)high;
            }
        }
        while (ipos < chunkEnd) {
            size_t const remaining = chunkEnd - ipos;
            size_t const blockSize = kref_optimal_block_size(src + ipos, remaining, savings);
            u32 const lastBlock = lastChunk && (blockSize == remaining);
            u8* const body = dst + pos + 3;
            size_t cSize; kref_blockwin bw;
            u32 const startIdx = (u32)ipos + IDX0, endIdx = (u32)(ipos + blockSize) + IDX0;
            /* ZSTD_window_enforceMaxDist(window, blockStart, maxDist) */
            if (startIdx > maxDist) {
                u32 const newLow = startIdx - maxDist;
                if (win.lowLimit < newLow) win.lowLimit = newLow;
                if (win.dictLimit < win.lowLimit) win.dictLimit = win.lowLimit;
            }
            bw.ext = 0; bw.dictStartIndex = 0; bw.prefixStartIndex = 0;
            bw.dictLimit = win.dictLimit; bw.maxDist = maxDist;                /* the regular variant takes ZSTD_getLowestPrefixIndex itself */
            if (win.lowLimit < win.dictLimit) {                               /* ZSTD_window_hasExtDict: the extDict variant */
                u32 const low = (endIdx - win.lowLimit > maxDist) ? endIdx - maxDist : win.lowLimit;            /* ZSTD_getLowestMatchIndex */
                u32 const prefixStart = win.dictLimit > low ? win.dictLimit : low;
                if (prefixStart != low) { bw.ext = 1; bw.dictStartIndex = low; bw.prefixStartIndex = prefixStart; }
            }
            cSize = compress_block_body_win(body, cap - pos - 3, src, ipos, blockSize, P, &w, &fs, NULL, &bw);
            if (cSize == KERR) { wksp_free(&w); return KERR; }
            if (cSize == 0) { wr24(dst + pos, lastBlock + (0 << 1) + (u32)(blockSize << 3)); memcpy(body, src + ipos, blockSize); cSize = 3 + blockSize; }
            else if (cSize == 1) { wr24(dst + pos, lastBlock + (1 << 1) + (u32)(blockSize << 3)); cSize = 3 + 1; }
            else { wr24(dst + pos, lastBlock + (2 << 1) + (u32)(cSize << 3)); cSize += 3; }
            savings += (int64_t)blockSize - (int64_t)cSize;
            ipos += blockSize; pos += cSize; fs.isFirstBlock = 0;
        }
        /* prepare the next chunk's place: inBuffTarget = inBuffPos + 128 KiB must fit the buffer */
        bufPos += chunkLen;
        if (bufPos + blockSizeMax > inBuffSize) { extBase = ipos - bufPos; bufPos = 0; }
    }
    wksp_free(&w);
    if (emptyEnd) { wr24(dst + pos, 1); pos += 3; }
    return pos;
}


/* ================================================================== */
/* Levels 1 and 2 above 128 KiB: frames of several blocks              */
/* (one-shot, and the streaming frames of finish = false callers --     */
/* the Ktor encoder's case: level 1, size unknown).  The pre-splitter   */
/* for strategy "fast" is ZSTD_splitBlock level 0 ("fromBorders": byte  */
/* histograms of the first, last and middle 512 bytes of the 128 KiB).  */
/* Input no larger than the window (512 KiB at level 1 unknown size).   */
/* ================================================================== */
static void hist512(u32* h, const u8* p) { int i; memset(h, 0, 256 * sizeof(u32)); for (i = 0; i < 512; i++) h[p[i]]++; }
static u64 fp_dist512(const u32* a, const u32* b)
{
    u64 d = 0; int n;
    for (n = 0; n < 256; n++) { int64_t const x = (int64_t)a[n] * 512 - (int64_t)b[n] * 512; d += (u64)(x < 0 ? -x : x); }
    return d;
}
static size_t split_block_from_borders(const u8* p)          /* block of exactly 128 KiB */
{
    u32 first[256], last[256], mid[256];
    size_t const blockSize = 128 << 10;
    hist512(first, p); hist512(last, p + blockSize - 512);
    { u64 const p50 = 512ull * 512ull; if (!(fp_dist512(first, last) >= p50 * 14 / 16)) return blockSize; }
    hist512(mid, p + blockSize / 2 - 256);
    {
        u64 const dB = fp_dist512(first, mid), dE = fp_dist512(last, mid);
        u64 const minDistance = 512 * 512 / 3;
        int64_t const diff = (int64_t)dB - (int64_t)dE;
        if ((u64)(diff < 0 ? -diff : diff) < minDistance) return 64 << 10;
        return (dB > dE) ? (32 << 10) : (96 << 10);
    }
}

/* fast_block for a block inside a larger input (tables and repcodes carried) */
static size_t fast_block_at(seqstore* ss, u32 rep[3], const u8* input, size_t blockStart, size_t srcSize, u32* hashTable, u32 hlog, u32 mls, size_t stepSize);

KREF_API size_t kref_zstd_fast_compress_big(u8* dst, size_t cap, const u8* src, size_t srcSize, int level, int stream, int emptyEnd)
{
    /* stream: 0 = the caller's array compressed in place (ZSTD_compress2 into a bound-sized buffer); 1 = streaming frame
     * (size unknown); 3 = one-shot through the reference's driver (size known, but its output slices are smaller than
     * the bound, so the input is staged in chunks of 128 KiB like a stream's) */
    u32 P[4]; kref_wksp w; seqstore ss; kref_frame_state fs; kref_hufstate nextHuf;
    size_t pos, ipos = 0, hdr; int64_t savings = 0; size_t const blockSizeMax = 128 << 10;
    int const chunked = stream != 0, unknown = stream == 1 || stream == 2;
    if ((level != 1 && level != 2 && level >= 0) || level < -131072) return KERR;      /* negative levels: row 0 of the tables, a step of 1 - level, raw literals */
    if (level == 2 && !unknown && srcSize > 131072 && srcSize <= 262144) {
        /* level 2's row for this size class is a double-fast one: window 18, chain 14, hash 14, minMatch 5 */
        u32 const Pd[4] = { 18, 14, 14, 5 };
        if (stream == 3) return dfast_compress_buffered(dst, cap, src, srcSize, 1, 0, srcSize / 10 > 8192 ? srcSize / 10 : 8192, 0, Pd);
        return compress_blocks_dfast(dst, cap, src, srcSize, NULL, NULL, Pd);
    }
    if (unknown) { P[0] = (level == 2) ? 20 : 19; P[2] = (level == 1) ? 14 : (level == 2) ? 16 : 13; P[3] = (level == 1) ? 7 : 6; }
    else kref_params_fast(level, srcSize, P);
    if (srcSize > ((size_t)1 << P[0])) return KERR;                  /* the window would slide */
    if (cap < kref_compress_bound(srcSize) + 16) return KERR;
    if (unknown) { wr32(dst, 0xFD2FB528u); dst[4] = 0; dst[5] = (u8)((P[0] - 10) << 3); pos = 6; }
    else pos = write_frame_header(dst, srcSize, P[0]);
    hdr = pos;
    if (!unknown && srcSize == 0) { wr24(dst + pos, 1); return pos + 3; }
    w.hashLong = (u32*)calloc((size_t)1 << P[2], sizeof(u32)); w.hashSmall = NULL;
    w.seqs = (kref_seq*)malloc(sizeof(kref_seq) * ((128 << 10) / 3 + 8)); w.lits = (u8*)malloc((128 << 10) + 32);
    fs.rep[0] = 1; fs.rep[1] = 4; fs.rep[2] = 8; fs.huf.valid = 0; memset(&fs.huf.ct, 0, sizeof(fs.huf.ct)); fs.isFirstBlock = 1;
    if (!unknown) emptyEnd = 0; else if (srcSize % blockSizeMax != 0 || srcSize == 0) emptyEnd = (srcSize == 0);
    while (ipos < srcSize) {
        size_t const chunkEnd = (chunked && ipos + blockSizeMax < srcSize) ? (ipos / blockSizeMax + 1) * blockSizeMax : srcSize;
        int const lastChunk = (chunkEnd == srcSize) && !emptyEnd;
        if (chunked && ipos == blockSizeMax) savings -= (int64_t)hdr;
        while (ipos < chunkEnd) {
            size_t const remaining = chunkEnd - ipos;
            size_t const blockSize = (remaining < blockSizeMax) ? remaining : (savings < 3) ? blockSizeMax : split_block_from_borders(src + ipos);
            u32 const lastBlock = lastChunk && (blockSize == remaining);
            u8* const body = dst + pos + 3; const u8* const bsrc = src + ipos;
            size_t cSize = 0, lastLL, litC, seqC; u32 rep[3];
            memset(&ss, 0, sizeof(ss)); ss.seqs = w.seqs; ss.lits = w.lits; ss.strategy = 1;
            if (blockSize >= 7) {
                memcpy(rep, fs.rep, sizeof(rep));
                lastLL = fast_block_at(&ss, rep, src, ipos, blockSize, w.hashLong, P[2], P[3], kref_fast_step(level));
                memcpy(ss.lits + ss.litSize, bsrc + blockSize - lastLL, lastLL); ss.litSize += lastLL;
                {
                    int const suspect = (ss.nbSeq == 0) || (ss.litSize / ss.nbSeq >= 20);
                    if (level < 0) { nextHuf = fs.huf; litC = lit_raw(body, cap - pos - 3, ss.lits, ss.litSize); }
                    else litC = compress_literals(body, cap - pos - 3, ss.lits, ss.litSize, suspect, &fs.huf, &nextHuf);
                    if (litC != KERR) {
                        seqC = compress_sequences(body + litC, cap - pos - 3 - litC, &ss);
                        if (seqC != KERR && seqC != 0) { cSize = litC + seqC; if (cSize >= blockSize - min_gain(blockSize)) cSize = 0; }
                    }
                }
                if (!fs.isFirstBlock && cSize < 25) {
                    size_t i; int same = 1;
                    for (i = 1; i < blockSize; i++) if (bsrc[i] != bsrc[0]) { same = 0; break; }
                    if (same) { body[0] = bsrc[0]; cSize = 1; }
                }
                if (cSize > 1) { memcpy(fs.rep, rep, sizeof(rep)); fs.huf = nextHuf; }
            }
            if (cSize == 0) { wr24(dst + pos, lastBlock + (0 << 1) + (u32)(blockSize << 3)); memcpy(body, bsrc, blockSize); cSize = 3 + blockSize; }
            else if (cSize == 1) { wr24(dst + pos, lastBlock + (1 << 1) + (u32)(blockSize << 3)); cSize = 3 + 1; }
            else { wr24(dst + pos, lastBlock + (2 << 1) + (u32)(cSize << 3)); cSize += 3; }
            savings += (int64_t)blockSize - (int64_t)cSize;
            ipos += blockSize; pos += cSize; fs.isFirstBlock = 0;
        }
    }
    free(w.hashLong); free(w.seqs); free(w.lits);
    if (emptyEnd) { wr24(dst + pos, 1); pos += 3; }
    return pos;
}
static size_t fast_block_at(seqstore* ss, u32 rep[3], const u8* input, size_t blockStart, size_t srcSize, u32* hashTable, u32 hlog, u32 mls, size_t stepSize)
{
    const u8* const base = input - IDX0;
    const u8* const src = input + blockStart;
    const u8* const istart = src;
    u32 const prefixStartIndex = IDX0;
    const u8* const prefixStart = base + prefixStartIndex;
    const u8* const iend = istart + srcSize;
    const u8* const ilimit = iend - 8;
    const u8* anchor = istart; const u8* ip0 = istart; const u8 *ip1, *ip2, *ip3;
    u32 current0 = 0;
    u32 rep_offset1 = rep[0], rep_offset2 = rep[1], offsetSaved1 = 0, offsetSaved2 = 0;
    size_t hash0, hash1; u32 matchIdx; u32 offcode; const u8* match0; size_t mLength;
    size_t step; const u8* nextStep; size_t const kStepIncr = 1 << 7;

    if (srcSize < 8) return srcSize;   /* (libzstd runs into _cleanup at once: repcodes unchanged) */
    ip0 += (ip0 == prefixStart);
    {
        u32 const curr = (u32)(ip0 - base); u32 const maxRep = curr - prefixStartIndex;
        if (rep_offset2 > maxRep) { offsetSaved2 = rep_offset2; rep_offset2 = 0; }
        if (rep_offset1 > maxRep) { offsetSaved1 = rep_offset1; rep_offset1 = 0; }
    }
_start:
    step = stepSize;
    nextStep = ip0 + kStepIncr;
    ip1 = ip0 + 1; ip2 = ip0 + step; ip3 = ip2 + 1;
    if (ip3 >= ilimit) goto _cleanup;
    hash0 = hash_short(ip0, hlog, mls);
    hash1 = hash_short(ip1, hlog, mls);
    matchIdx = hashTable[hash0];
    do {
        u32 const rval = rd32(ip2 - rep_offset1);
        current0 = (u32)(ip0 - base);
        hashTable[hash0] = current0;
        if ((rd32(ip2) == rval) & (rep_offset1 > 0)) {
            ip0 = ip2;
            match0 = ip0 - rep_offset1;
            mLength = ip0[-1] == match0[-1];
            ip0 -= mLength; match0 -= mLength;
            offcode = 1;
            mLength += 4;
            hashTable[hash1] = (u32)(ip1 - base);
            goto _match;
        }
        if (matchIdx >= prefixStartIndex && rd32(base + matchIdx) == rd32(ip0)) {
            hashTable[hash1] = (u32)(ip1 - base);
            goto _offset;
        }
        matchIdx = hashTable[hash1];
        hash0 = hash1;
        hash1 = hash_short(ip2, hlog, mls);
        ip0 = ip1; ip1 = ip2; ip2 = ip3;
        current0 = (u32)(ip0 - base);
        hashTable[hash0] = current0;
        if (matchIdx >= prefixStartIndex && rd32(base + matchIdx) == rd32(ip0)) {
            if (step <= 4) hashTable[hash1] = (u32)(ip1 - base);
            goto _offset;
        }
        matchIdx = hashTable[hash1];
        hash0 = hash1;
        hash1 = hash_short(ip2, hlog, mls);
        ip0 = ip1; ip1 = ip2; ip2 = ip0 + step; ip3 = ip1 + step;
        if (ip2 >= nextStep) { step++; nextStep += kStepIncr; }
    } while (ip3 < ilimit);

_cleanup:
    offsetSaved2 = ((offsetSaved1 != 0) && (rep_offset1 != 0)) ? offsetSaved1 : offsetSaved2;
    rep[0] = rep_offset1 ? rep_offset1 : offsetSaved1;
    rep[1] = rep_offset2 ? rep_offset2 : offsetSaved2;
    return (size_t)(iend - anchor);

_offset:
    match0 = base + matchIdx;
    rep_offset2 = rep_offset1;
    rep_offset1 = (u32)(ip0 - match0);
    offcode = rep_offset1 + 3;
    mLength = 4;
    while (((ip0 > anchor) & (match0 > prefixStart)) && (ip0[-1] == match0[-1])) { ip0--; match0--; mLength++; }

_match:
    mLength += count_eq(ip0 + mLength, match0 + mLength, iend);
    store_seq(ss, (size_t)(ip0 - anchor), anchor, offcode, mLength);
    ip0 += mLength;
    anchor = ip0;
    if (ip0 <= ilimit) {
        hashTable[hash_short(base + current0 + 2, hlog, mls)] = current0 + 2;
        hashTable[hash_short(ip0 - 2, hlog, mls)] = (u32)(ip0 - 2 - base);
        if (rep_offset2 > 0) {
            while ((ip0 <= ilimit) && (rd32(ip0) == rd32(ip0 - rep_offset2))) {
                size_t const rLength = count_eq(ip0 + 4, ip0 + 4 - rep_offset2, iend) + 4;
                { u32 const tmpOff = rep_offset2; rep_offset2 = rep_offset1; rep_offset1 = tmpOff; }
                hashTable[hash_short(ip0, hlog, mls)] = (u32)(ip0 - base);
                ip0 += rLength;
                store_seq(ss, 0, anchor, 1, rLength);
                anchor = ip0;
            }
        }
    }
    goto _start;
}


/* ================================================================== */
/* Strategy "fast" (levels 1, 2, negative) beyond its window (round 4) */
/* The staging logic of dfast_compress_buffered is not about the       */
/* strategy: ZSTD_compressStream_generic stages the input in chunks of */
/* 128 KiB into a buffer of window + 128 KiB bytes whatever the level; */
/* once that buffer has wrapped the lap before is an older segment and */
/* libzstd 1.5.7 parses the blocks with                                 */
/* ZSTD_compressBlock_fast_extDict_generic, restated below, until      */
/* ZSTD_window_enforceMaxDist has pushed the segment out of the        */
/* window.  The Ktor encoder streams at level 1                         */
/* (kompressor-zstd-ktor ZstdContentEncoder.kt:11): window 2^19, so    */
/* every response above 640 KiB goes through here.                      */
/* ================================================================== */

/* ZSTD_compressBlock_fast_noDict_generic for a block inside a longer input whose window may have slid: matches are valid from
 * ZSTD_getLowestPrefixIndex at the block's END on, repcodes are checked against the one at the first searched position. */
static size_t fast_block_low(seqstore* ss, u32 rep[3], const u8* input, size_t blockStart, size_t srcSize, u32* hashTable, u32 hlog, u32 mls,
                             size_t stepSize, u32 dictLimit, u32 maxDist)
{
    const u8* const base = input - IDX0;
    const u8* const src = input + blockStart;
    const u8* const istart = src;
    u32 const endIndex = (u32)(blockStart + srcSize) + IDX0;
    u32 const prefixStartIndex = (endIndex - dictLimit > maxDist) ? endIndex - maxDist : dictLimit;
    const u8* const prefixStart = base + prefixStartIndex;
    const u8* const iend = istart + srcSize;
    const u8* const ilimit = iend - 8;
    const u8* anchor = istart; const u8* ip0 = istart; const u8 *ip1, *ip2, *ip3;
    u32 current0 = 0;
    u32 rep_offset1 = rep[0], rep_offset2 = rep[1], offsetSaved1 = 0, offsetSaved2 = 0;
    size_t hash0, hash1; u32 matchIdx; u32 offcode; const u8* match0; size_t mLength;
    size_t step; const u8* nextStep; size_t const kStepIncr = 1 << 7;

    if (srcSize < 8) return srcSize;
    ip0 += (ip0 == prefixStart);
    {
        u32 const curr = (u32)(ip0 - base);
        u32 const windowLow = (curr - dictLimit > maxDist) ? curr - maxDist : dictLimit;
        u32 const maxRep = curr - windowLow;
        if (rep_offset2 > maxRep) { offsetSaved2 = rep_offset2; rep_offset2 = 0; }
        if (rep_offset1 > maxRep) { offsetSaved1 = rep_offset1; rep_offset1 = 0; }
    }
_start:
    step = stepSize;
    nextStep = ip0 + kStepIncr;
    ip1 = ip0 + 1; ip2 = ip0 + step; ip3 = ip2 + 1;
    if (ip3 >= ilimit) goto _cleanup;
    hash0 = hash_short(ip0, hlog, mls);
    hash1 = hash_short(ip1, hlog, mls);
    matchIdx = hashTable[hash0];
    do {
        u32 const rval = rd32(ip2 - rep_offset1);
        current0 = (u32)(ip0 - base);
        hashTable[hash0] = current0;
        if ((rd32(ip2) == rval) & (rep_offset1 > 0)) {
            ip0 = ip2;
            match0 = ip0 - rep_offset1;
            mLength = ip0[-1] == match0[-1];
            ip0 -= mLength; match0 -= mLength;
            offcode = 1;
            mLength += 4;
            hashTable[hash1] = (u32)(ip1 - base);
            goto _match;
        }
        if (matchIdx >= prefixStartIndex && rd32(base + matchIdx) == rd32(ip0)) {
            hashTable[hash1] = (u32)(ip1 - base);
            goto _offset;
        }
        matchIdx = hashTable[hash1];
        hash0 = hash1;
        hash1 = hash_short(ip2, hlog, mls);
        ip0 = ip1; ip1 = ip2; ip2 = ip3;
        current0 = (u32)(ip0 - base);
        hashTable[hash0] = current0;
        if (matchIdx >= prefixStartIndex && rd32(base + matchIdx) == rd32(ip0)) {
            if (step <= 4) hashTable[hash1] = (u32)(ip1 - base);
            goto _offset;
        }
        matchIdx = hashTable[hash1];
        hash0 = hash1;
        hash1 = hash_short(ip2, hlog, mls);
        ip0 = ip1; ip1 = ip2; ip2 = ip0 + step; ip3 = ip1 + step;
        if (ip2 >= nextStep) { step++; nextStep += kStepIncr; }
    } while (ip3 < ilimit);

_cleanup:
    offsetSaved2 = ((offsetSaved1 != 0) && (rep_offset1 != 0)) ? offsetSaved1 : offsetSaved2;
    rep[0] = rep_offset1 ? rep_offset1 : offsetSaved1;
    rep[1] = rep_offset2 ? rep_offset2 : offsetSaved2;
    return (size_t)(iend - anchor);

_offset:
    match0 = base + matchIdx;
    rep_offset2 = rep_offset1;
    rep_offset1 = (u32)(ip0 - match0);
    offcode = rep_offset1 + 3;
    mLength = 4;
    while (((ip0 > anchor) & (match0 > prefixStart)) && (ip0[-1] == match0[-1])) { ip0--; match0--; mLength++; }

_match:
    mLength += count_eq(ip0 + mLength, match0 + mLength, iend);
    store_seq(ss, (size_t)(ip0 - anchor), anchor, offcode, mLength);
    ip0 += mLength;
    anchor = ip0;
    if (ip0 <= ilimit) {
        hashTable[hash_short(base + current0 + 2, hlog, mls)] = current0 + 2;
        hashTable[hash_short(ip0 - 2, hlog, mls)] = (u32)(ip0 - 2 - base);
        if (rep_offset2 > 0) {
            while ((ip0 <= ilimit) && (rd32(ip0) == rd32(ip0 - rep_offset2))) {
                size_t const rLength = count_eq(ip0 + 4, ip0 + 4 - rep_offset2, iend) + 4;
                { u32 const tmpOff = rep_offset2; rep_offset2 = rep_offset1; rep_offset1 = tmpOff; }
                hashTable[hash_short(ip0, hlog, mls)] = (u32)(ip0 - base);
                ip0 += rLength;
                store_seq(ss, 0, anchor, 1, rLength);
                anchor = ip0;
            }
        }
    }
    goto _start;
}

/* ZSTD_compressBlock_fast_extDict_generic over the block [istart, istart + srcSize): indices below prefixStartIndex are the older
 * segment (valid from dictStartIndex on).  The whole stream is contiguous here, so both segments sit behind the same base and
 * ZSTD_count_2segments is an ordinary count; what stays of the two segments are the index rules: a match that starts in the older
 * segment does not grow backwards past its start, a repcode that would straddle the boundary is refused. */
static size_t fast_extdict_block(seqstore* ss, u32 rep[3], const u8* input, size_t blockStart, size_t srcSize, u32* hashTable, u32 hlog, u32 mls,
                                 size_t stepSize, u32 dictStartIndex, u32 prefixStartIndex)
{
    const u8* const base = input - IDX0;
    const u8* const istart = input + blockStart;
    const u8* anchor = istart;
    const u8* const dictStart = base + dictStartIndex;
    const u8* const prefixStart = base + prefixStartIndex;
    const u8* const iend = istart + srcSize;
    const u8* const ilimit = iend - 8;
    u32 offset_1 = rep[0], offset_2 = rep[1], offsetSaved1 = 0, offsetSaved2 = 0;
    const u8* ip0 = istart; const u8 *ip1, *ip2, *ip3;
    u32 current0 = 0;
    size_t hash0, hash1; u32 idx; u32 offcode; const u8* match0; size_t mLength;
    size_t step; const u8* nextStep; size_t const kStepIncr = 1 << 7;

    if (srcSize < 8) return srcSize;
    {
        u32 const curr = (u32)(ip0 - base);
        u32 const maxRep = curr - dictStartIndex;
        if (offset_2 >= maxRep) { offsetSaved2 = offset_2; offset_2 = 0; }
        if (offset_1 >= maxRep) { offsetSaved1 = offset_1; offset_1 = 0; }
    }
_start:
    step = stepSize;
    nextStep = ip0 + kStepIncr;
    ip1 = ip0 + 1; ip2 = ip0 + step; ip3 = ip2 + 1;
    if (ip3 >= ilimit) goto _cleanup;
    hash0 = hash_short(ip0, hlog, mls);
    hash1 = hash_short(ip1, hlog, mls);
    idx = hashTable[hash0];
    do {
        {   /* repcode at ip2 */
            u32 const current2 = (u32)(ip2 - base);
            u32 const repIndex = current2 - offset_1;
            u32 rval;
            if (((u32)(prefixStartIndex - repIndex) >= 4) & (offset_1 > 0)) rval = rd32(base + repIndex);
            else rval = rd32(ip2) ^ 1;
            current0 = (u32)(ip0 - base);
            hashTable[hash0] = current0;
            if (rd32(ip2) == rval) {
                ip0 = ip2;
                match0 = base + repIndex;
                mLength = ip0[-1] == match0[-1];
                ip0 -= mLength; match0 -= mLength;
                offcode = 1;
                mLength += 4;
                goto _match;
            }
        }
        {
            u32 const mval = idx >= dictStartIndex ? rd32(base + idx) : rd32(ip0) ^ 1;
            if (rd32(ip0) == mval) goto _offset;
        }
        idx = hashTable[hash1];
        hash0 = hash1;
        hash1 = hash_short(ip2, hlog, mls);
        ip0 = ip1; ip1 = ip2; ip2 = ip3;
        current0 = (u32)(ip0 - base);
        hashTable[hash0] = current0;
        {
            u32 const mval = idx >= dictStartIndex ? rd32(base + idx) : rd32(ip0) ^ 1;
            if (rd32(ip0) == mval) goto _offset;
        }
        idx = hashTable[hash1];
        hash0 = hash1;
        hash1 = hash_short(ip2, hlog, mls);
        ip0 = ip1; ip1 = ip2; ip2 = ip0 + step; ip3 = ip1 + step;
        if (ip2 >= nextStep) { step++; nextStep += kStepIncr; }
    } while (ip3 < ilimit);

_cleanup:
    offsetSaved2 = ((offsetSaved1 != 0) && (offset_1 != 0)) ? offsetSaved1 : offsetSaved2;
    rep[0] = offset_1 ? offset_1 : offsetSaved1;
    rep[1] = offset_2 ? offset_2 : offsetSaved2;
    return (size_t)(iend - anchor);

_offset:
    {
        u32 const offset = current0 - idx;
        const u8* const lowMatchPtr = idx < prefixStartIndex ? dictStart : prefixStart;
        match0 = base + idx;
        offset_2 = offset_1;
        offset_1 = offset;
        offcode = offset + 3;
        mLength = 4;
        while (((ip0 > anchor) & (match0 > lowMatchPtr)) && (ip0[-1] == match0[-1])) { ip0--; match0--; mLength++; }
    }
_match:
    mLength += count_eq(ip0 + mLength, match0 + mLength, iend);
    store_seq(ss, (size_t)(ip0 - anchor), anchor, offcode, mLength);
    ip0 += mLength;
    anchor = ip0;
    if (ip1 < ip0) hashTable[hash1] = (u32)(ip1 - base);
    if (ip0 <= ilimit) {
        hashTable[hash_short(base + current0 + 2, hlog, mls)] = current0 + 2;
        hashTable[hash_short(ip0 - 2, hlog, mls)] = (u32)(ip0 - 2 - base);
        while (ip0 <= ilimit) {
            u32 const repIndex2 = (u32)(ip0 - base) - offset_2;
            if ((index_overlap_check(prefixStartIndex, repIndex2) & (offset_2 > 0)) && rd32(base + repIndex2) == rd32(ip0)) {
                size_t const repLength2 = count_eq(ip0 + 4, base + repIndex2 + 4, iend) + 4;
                { u32 const tmpOffset = offset_2; offset_2 = offset_1; offset_1 = tmpOffset; }
                store_seq(ss, 0, anchor, 1, repLength2);
                hashTable[hash_short(ip0, hlog, mls)] = (u32)(ip0 - base);
                ip0 += repLength2;
                anchor = ip0;
                continue;
            }
            break;
        }
    }
    goto _start;
}

/* how many blocks the last buffered "fast" frame of this thread parsed with the extDict variant (the tests make sure their inputs
 * reach it) */
static __thread unsigned kref_fast_ext_blocks_tl = 0;
KREF_API unsigned kref_fast_ext_blocks(void) { return kref_fast_ext_blocks_tl; }

/* The buffered frame at a "fast" level (1, 2, negative), any length below 4 GiB: dfast_compress_buffered's window bookkeeping with
 * this strategy's pre-splitter ("fromBorders"), parser pair and literal rule.  knownSize / emptyEnd / outChunk / tailDirect as there. */
static size_t fast_compress_buffered(u8* dst, size_t cap, const u8* src, size_t srcSize, int level, int knownSize, int emptyEnd, size_t outChunk, size_t tailDirect)
{
    int tail = 0;
    u32 P[4]; kref_wksp w; kref_frame_state fs; seqstore ss; kref_hufstate nextHuf; size_t pos, ipos = 0, hdr; int64_t savings = 0;
    size_t const blockSizeMax = 128 << 10;
    kref_window win; size_t windowSize, inBuffSize, bufPos = 0, extBase = 0; u32 maxDist; int haveExt = 0;
    kref_fast_ext_blocks_tl = 0;
    if ((level != 1 && level != 2 && level >= 0) || level < -131072) return KERR;
    if (srcSize >= 0xF0000000u) return KERR;
    if (cap < kref_compress_bound(srcSize) + 16) return KERR;
    if (knownSize) { kref_params_fast(level, srcSize, P); pos = write_frame_header(dst, srcSize, P[0]); emptyEnd = 0; }
    else { P[0] = (level == 2) ? 20 : 19; P[1] = 0; P[2] = (level == 1) ? 14 : (level == 2) ? 16 : 13; P[3] = (level == 1) ? 7 : 6; wr32(dst, 0xFD2FB528u); dst[4] = 0; dst[5] = (u8)((P[0] - 10) << 3); pos = 6; }
    hdr = pos;
    if (knownSize && srcSize == 0) { wr24(dst + pos, 1); return pos + 3; }
    w.hashLong = (u32*)calloc((size_t)1 << P[2], sizeof(u32)); w.hashSmall = NULL;
    w.seqs = (kref_seq*)malloc(sizeof(kref_seq) * ((128 << 10) / 3 + 8)); w.lits = (u8*)malloc((128 << 10) + 32);
    if (!w.hashLong || !w.seqs || !w.lits) { free(w.hashLong); free(w.seqs); free(w.lits); return KERR; }
    fs.rep[0] = 1; fs.rep[1] = 4; fs.rep[2] = 8; fs.huf.valid = 0; memset(&fs.huf.ct, 0, sizeof(fs.huf.ct)); fs.isFirstBlock = 1;
    if (srcSize % blockSizeMax != 0 || srcSize == 0) emptyEnd = (!knownSize && srcSize == 0);
    maxDist = 1u << P[0];
    windowSize = (knownSize && srcSize < ((size_t)1 << P[0])) ? (srcSize ? srcSize : 1) : ((size_t)1 << P[0]);
    inBuffSize = windowSize + (blockSizeMax < windowSize ? blockSizeMax : windowSize);
    win.lowLimit = IDX0; win.dictLimit = IDX0;
    while (ipos < srcSize) {
        size_t chunkEnd, chunkLen; int lastChunk;
        if (ipos != 0 && bufPos == 0 && !tail) {
            if (knownSize && outChunk) {
                size_t const room = outChunk - pos % outChunk, r = srcSize - ipos;
                if (room >= r + (r >> 8) + (r < blockSizeMax ? (blockSizeMax - r) >> 11 : 0)) tail = 1;
            } else if (!knownSize && tailDirect && ipos + tailDirect == srcSize) tail = 1;
        }
        chunkEnd = (!tail && knownSize != 2 && ipos + blockSizeMax < srcSize) ? ipos + blockSizeMax : srcSize;
        chunkLen = chunkEnd - ipos;
        lastChunk = (chunkEnd == srcSize) && !emptyEnd;
        if (ipos == blockSizeMax && knownSize != 2) savings -= (int64_t)hdr;
        if (ipos != 0 && bufPos == 0) {
            win.lowLimit = win.dictLimit;
            win.dictLimit = (u32)ipos + IDX0;
            if (win.dictLimit - win.lowLimit < 8) win.lowLimit = win.dictLimit;
            haveExt = 1;
        }
        if (haveExt && !tail) {
            size_t const extLoPhys = (size_t)(win.lowLimit - IDX0) - extBase, extHiPhys = (size_t)(win.dictLimit - IDX0) - extBase;
            if (bufPos + chunkLen > extLoPhys && bufPos < extHiPhys) {
                size_t const high = extBase + bufPos + chunkLen + IDX0;
                win.lowLimit = high > win.dictLimit ? win.dictLimit : (u32)high;
            }
        }
        while (ipos < chunkEnd) {
            size_t const remaining = chunkEnd - ipos;
            size_t const blockSize = (remaining < blockSizeMax) ? remaining : (savings < 3) ? blockSizeMax : split_block_from_borders(src + ipos);
            u32 const lastBlock = lastChunk && (blockSize == remaining);
            u8* const body = dst + pos + 3; const u8* const bsrc = src + ipos;
            size_t cSize = 0, lastLL, litC, seqC; u32 rep[3];
            u32 const startIdx = (u32)ipos + IDX0, endIdx = (u32)(ipos + blockSize) + IDX0;
            int ext = 0; u32 dsi = 0, psi = 0;
            if (startIdx > maxDist) {                                          /* ZSTD_window_enforceMaxDist */
                u32 const newLow = startIdx - maxDist;
                if (win.lowLimit < newLow) win.lowLimit = newLow;
                if (win.dictLimit < win.lowLimit) win.dictLimit = win.lowLimit;
            }
            if (win.lowLimit < win.dictLimit) {                               /* ZSTD_window_hasExtDict */
                u32 const low = (endIdx - win.lowLimit > maxDist) ? endIdx - maxDist : win.lowLimit;
                u32 const prefixStart = win.dictLimit > low ? win.dictLimit : low;
                if (prefixStart != low) { ext = 1; dsi = low; psi = prefixStart; }
            }
            memset(&ss, 0, sizeof(ss)); ss.seqs = w.seqs; ss.lits = w.lits; ss.strategy = 1;
            if (blockSize >= 7) {
                memcpy(rep, fs.rep, sizeof(rep));
                if (ext) kref_fast_ext_blocks_tl++;
                if (ext) lastLL = fast_extdict_block(&ss, rep, src, ipos, blockSize, w.hashLong, P[2], P[3], kref_fast_step(level), dsi, psi);
                else lastLL = fast_block_low(&ss, rep, src, ipos, blockSize, w.hashLong, P[2], P[3], kref_fast_step(level), win.dictLimit, maxDist);
                memcpy(ss.lits + ss.litSize, bsrc + blockSize - lastLL, lastLL); ss.litSize += lastLL;
                {
                    int const suspect = (ss.nbSeq == 0) || (ss.litSize / ss.nbSeq >= 20);
                    if (level < 0) { nextHuf = fs.huf; litC = lit_raw(body, cap - pos - 3, ss.lits, ss.litSize); }
                    else litC = compress_literals(body, cap - pos - 3, ss.lits, ss.litSize, suspect, &fs.huf, &nextHuf);
                    if (litC != KERR) {
                        seqC = compress_sequences(body + litC, cap - pos - 3 - litC, &ss);
                        if (seqC != KERR && seqC != 0) { cSize = litC + seqC; if (cSize >= blockSize - min_gain(blockSize)) cSize = 0; }
                    }
                }
                if (!fs.isFirstBlock && cSize < 25) {
                    size_t i; int same = 1;
                    for (i = 1; i < blockSize; i++) if (bsrc[i] != bsrc[0]) { same = 0; break; }
                    if (same) { body[0] = bsrc[0]; cSize = 1; }
                }
                if (cSize > 1) { memcpy(fs.rep, rep, sizeof(rep)); fs.huf = nextHuf; }
            }
            if (cSize == 0) { wr24(dst + pos, lastBlock + (0 << 1) + (u32)(blockSize << 3)); memcpy(body, bsrc, blockSize); cSize = 3 + blockSize; }
            else if (cSize == 1) { wr24(dst + pos, lastBlock + (1 << 1) + (u32)(blockSize << 3)); cSize = 3 + 1; }
            else { wr24(dst + pos, lastBlock + (2 << 1) + (u32)(cSize << 3)); cSize += 3; }
            savings += (int64_t)blockSize - (int64_t)cSize;
            ipos += blockSize; pos += cSize; fs.isFirstBlock = 0;
        }
        bufPos += chunkLen;
        if (bufPos + blockSizeMax > inBuffSize) { extBase = ipos - bufPos; bufPos = 0; }
    }
    free(w.hashLong); free(w.seqs); free(w.lits);
    if (emptyEnd) { wr24(dst + pos, 1); pos += 3; }
    return pos;
}
/* stream as in kref_zstd_fast_compress_big (0 in place, 1 / 2 streaming frame closed with / without data, 3 the reference's one-shot
 * driver); outChunk = 0: max(8192, n / 10) in mode 3; tailDirect: dfast_compress_buffered's.  Any length: beyond the level's window
 * (512 KiB at level 1 and the negative levels, 1 MiB at level 2) the window slides as libzstd's does. */
KREF_API size_t kref_zstd_fast_compress_buffered(u8* dst, size_t cap, const u8* src, size_t srcSize, int level, int stream, int emptyEnd, size_t outChunk, size_t tailDirect)
{
    int const unknown = stream == 1 || stream == 2;
    if (level == 2 && !unknown && srcSize > 131072 && srcSize <= 262144) return kref_zstd_fast_compress_big(dst, cap, src, srcSize, level, stream, emptyEnd);   /* (its double-fast row) */
    if (stream == 3 && !outChunk) outChunk = srcSize / 10 > 8192 ? srcSize / 10 : 8192;
    return fast_compress_buffered(dst, cap, src, srcSize, level, unknown ? 0 : stream == 0 ? 2 : 1, stream == 2 ? 1 : emptyEnd, stream == 3 ? outChunk : 0, unknown ? tailDirect : 0);
}

/* ================================================================== */
/* Levels 5 .. 10 (and 4 .. 8 up to 16 KiB): strategies "greedy",      */
/* "lazy", "lazy2" -- libzstd 1.5.7 zstd_lazy.c                        */
/* ZSTD_compressBlock_lazy_generic with depth 0 / 1 / 2 over            */
/*  * the row-based match finder (ZSTD_RowFindBestMatch: windowLog     */
/*    above 14 on a machine with 128-bit vectors, which is where the    */
/*    reference's JNI library runs), or                                 */
/*  * the hash-chain match finder (ZSTD_HcFindBestMatch: windowLog up   */
/*    to 14, i.e. inputs up to 16 KiB).                                 */
/* One block: inputs up to 128 KiB, no dictionary, first block of a     */
/* frame.  Both finders insert every position they pass (rows keep the  */
/* newest 15 / 31 / 63 per row, tagged with 8 more hash bits; chains    */
/* keep everything) -- except in "lazy skipping" (long stretches        */
/* without a match: only searched positions go in) and behind matches   */
/* longer than 384 (the row finder inserts the first 96 and the last    */
/* 32 positions only).                                                  */
/* ================================================================== */
typedef struct { u32 W, C, H, S, mml, strat; } kref_lpar;       /* strat: 3 greedy, 4 lazy, 5 lazy2 */
/* ZSTD_getCParams(level, n, 0) for these levels; 0 when the level is another strategy at this size */
static int lazy_params(int level, size_t n, kref_lpar* p)
{
    u32 srcLog, tW;
    if (n == 0 || n > 131072) return 0;
    if (n < 8 && level >= 4 && level <= 10) { p->W = 10; p->C = 10; p->H = 11; p->S = 3; p->mml = 4; p->strat = 3; return 1; }      /* (nothing to parse: a raw block whatever the strategy) */
    if (n <= 16384) {
        static const u32 S16[9] = { 0, 0, 0, 0, 4, 3, 4, 6, 8 }; static const u32 ST16[9] = { 0, 0, 0, 0, 3, 4, 5, 5, 5 };
        if (level < 4 || level > 8) return 0;
        tW = 14; p->C = 14; p->H = 14; p->S = S16[level]; p->mml = 4; p->strat = ST16[level];
    } else {
        static const u32 S128[11] = { 0, 0, 0, 0, 0, 3, 3, 3, 4, 5, 6 }; static const u32 ST128[11] = { 0, 0, 0, 0, 0, 3, 4, 5, 5, 5, 5 };
        if (level < 5 || level > 10) return 0;
        tW = 17; p->C = 16; p->H = 17; p->S = S128[level]; p->mml = 4; p->strat = ST128[level];
    }
    srcLog = (n < 64) ? 6 : hb32((u32)(n - 1)) + 1;
    p->W = tW < srcLog ? tW : srcLog;
    if (p->H > p->W + 1) p->H = p->W + 1;
    if (p->C > p->W) p->C = p->W;
    if (p->W < 10) p->W = 10;
    return 1;
}
KREF_API int kref_params_lazy(int level, size_t n, u32* out6)
{
    kref_lpar p; if (!lazy_params(level, n, &p)) return 0;
    out6[0] = p.W; out6[1] = p.C; out6[2] = p.H; out6[3] = p.S; out6[4] = p.mml; out6[5] = p.strat; return 1;
}

typedef struct {
    const u8* base;              /* index i <-> base[i]: the input's first byte has index 2 */
    const u8* iend;
    u32* hashTable; u32* chainTable; u8* tagTable;
    u32 nextToUpdate; int lazySkipping;
    u32 rowLog, rowHashLog, S, C, H, mls; int rows;
} kref_lazyms;

static u32 lz_hash(const u8* p, u32 hBits, u32 mls) { return (u32)hash_short(p, hBits, mls); }
static u32 row_next_index(u8* tagRow, u32 rowMask) { u32 next = ((u32)*tagRow - 1u) & rowMask; next += (next == 0) ? rowMask : 0; *tagRow = (u8)next; return next; }
static void row_insert(kref_lazyms* ms, u32 idx)
{
    u32 const hash = lz_hash(ms->base + idx, ms->rowHashLog + 8, ms->mls);
    u32 const relRow = (hash >> 8) << ms->rowLog;
    u32 const pos = row_next_index(ms->tagTable + relRow, (1u << ms->rowLog) - 1);
    ms->tagTable[relRow + pos] = (u8)hash; ms->hashTable[relRow + pos] = idx;
}
/* ZSTD_row_update_internal: everything from nextToUpdate up to (not including) target goes in; a gap above 384: its first 96 and last 32 */
static void row_update(kref_lazyms* ms, u32 target)
{
    u32 idx = ms->nextToUpdate;
    if (target - idx > 384) { u32 const bound = idx + 96; for (; idx < bound; idx++) row_insert(ms, idx); idx = target - 32; }
    for (; idx < target; idx++) row_insert(ms, idx);
    ms->nextToUpdate = target;
}
/* ZSTD_RowFindBestMatch: the row's entries with the position's tag, newest first, at most 1 << min(searchLog, rowLog) of them; the
 * position itself goes into the row before they are compared; the longest wins, the newer one among equals */
static size_t row_find(kref_lazyms* ms, const u8* ip, size_t* offBasePtr)
{
    u32 const curr = (u32)(ip - ms->base), rowEntries = 1u << ms->rowLog, rowMask = rowEntries - 1;
    u32 const capped = ms->S < ms->rowLog ? ms->S : ms->rowLog; u32 nbAttempts = 1u << capped;
    u32 const lowLimit = IDX0; size_t ml = 4 - 1; u32 hash, buf[64], nb = 0, k;
    if (!ms->lazySkipping) row_update(ms, curr); else ms->nextToUpdate = curr;
    hash = lz_hash(ip, ms->rowHashLog + 8, ms->mls);
    {
        u32 const relRow = (hash >> 8) << ms->rowLog; u8 const tag = (u8)hash;
        u32* const row = ms->hashTable + relRow; u8* const tagRow = ms->tagTable + relRow;
        u32 const head = (u32)tagRow[0] & rowMask;
        for (k = 0; k < rowEntries && nbAttempts > 0; k++) {
            u32 const matchPos = (head + k) & rowMask;
            if (tagRow[matchPos] != tag) continue;
            if (matchPos == 0) continue;
            if (row[matchPos] < lowLimit) break;
            buf[nb++] = row[matchPos]; nbAttempts--;
        }
        { u32 const pos = row_next_index(tagRow, rowMask); tagRow[pos] = tag; row[pos] = ms->nextToUpdate++; }
    }
    for (k = 0; k < nb; k++) {
        const u8* const match = ms->base + buf[k]; size_t cur = 0;
        if (rd32(match + ml - 3) == rd32(ip + ml - 3)) cur = count_eq(ip, match, ms->iend);
        if (cur > ml) { ml = cur; *offBasePtr = (size_t)(curr - buf[k]) + 3; if (ip + cur == ms->iend) break; }
    }
    return ml;
}
/* ZSTD_HcFindBestMatch: the chain of earlier positions with the same hash, newest first, 1 << searchLog of them at most */
static size_t hc_find(kref_lazyms* ms, const u8* ip, size_t* offBasePtr)
{
    u32 const curr = (u32)(ip - ms->base), chainSize = 1u << ms->C, chainMask = chainSize - 1;
    u32 const lowLimit = IDX0, minChain = curr > chainSize ? curr - chainSize : 0; u32 nbAttempts = 1u << ms->S; size_t ml = 4 - 1; u32 matchIndex;
    {   /* ZSTD_insertAndFindFirstIndex_internal */
        u32 idx = ms->nextToUpdate;
        while (idx < curr) {
            u32 const h = lz_hash(ms->base + idx, ms->H, ms->mls);
            ms->chainTable[idx & chainMask] = ms->hashTable[h]; ms->hashTable[h] = idx; idx++;
            if (ms->lazySkipping) break;
        }
        ms->nextToUpdate = curr;
        matchIndex = ms->hashTable[lz_hash(ip, ms->H, ms->mls)];
    }
    for (; (matchIndex >= lowLimit) & (nbAttempts > 0); nbAttempts--) {
        const u8* const match = ms->base + matchIndex; size_t cur = 0;
        if (rd32(match + ml - 3) == rd32(ip + ml - 3)) cur = count_eq(ip, match, ms->iend);
        if (cur > ml) { ml = cur; *offBasePtr = (size_t)(curr - matchIndex) + 3; if (ip + cur == ms->iend) break; }
        if (matchIndex <= minChain) break;
        matchIndex = ms->chainTable[matchIndex & chainMask];
    }
    return ml;
}
static size_t lazy_search(kref_lazyms* ms, const u8* ip, size_t* offBasePtr) { return ms->rows ? row_find(ms, ip, offBasePtr) : hc_find(ms, ip, offBasePtr); }

/* ZSTD_compressBlock_lazy_generic (noDict), depth = strat - 3.  src: the whole input = the frame's first block.  Returns the last literals. */
static size_t lazy_block_at(seqstore* ss, u32 rep[3], const u8* src, size_t srcSize, kref_lazyms* ms, u32 depth, const u8* frameStart);
static size_t lazy_block(seqstore* ss, u32 rep[3], const u8* src, size_t srcSize, kref_lazyms* ms, u32 depth) { return lazy_block_at(ss, rep, src, srcSize, ms, depth, src); }
/* ... a block of a frame that began at frameStart (the window has not moved: the frame fits it) */
static size_t lazy_block_at(seqstore* ss, u32 rep[3], const u8* src, size_t srcSize, kref_lazyms* ms, u32 depth, const u8* frameStart)
{
    const u8* const istart = src; const u8* ip = istart; const u8* anchor = istart; const u8* const iend = istart + srcSize;
    const u8* const ilimit = ms->rows ? iend - 8 - 8 : iend - 8;
    const u8* const prefixLowest = frameStart;
    u32 offset_1 = rep[0], offset_2 = rep[1], offsetSaved1 = 0, offsetSaved2 = 0;
    ip += (ip == frameStart);                          /* dictAndPrefixLength == 0 */
    { u32 const maxRep = (u32)(ip - frameStart); if (offset_2 > maxRep) { offsetSaved2 = offset_2; offset_2 = 0; } if (offset_1 > maxRep) { offsetSaved1 = offset_1; offset_1 = 0; } }
    ms->lazySkipping = 0;
    while (ip < ilimit) {
        size_t matchLength = 0, offBase = 1; const u8* start = ip + 1;
        if ((offset_1 > 0) && (rd32(ip + 1 - offset_1) == rd32(ip + 1))) {
            matchLength = count_eq(ip + 1 + 4, ip + 1 + 4 - offset_1, iend) + 4;
            if (depth == 0) goto _storeSequence;
        }
        {   size_t offbaseFound = 999999999;
            size_t const ml2 = lazy_search(ms, ip, &offbaseFound);
            if (ml2 > matchLength) { matchLength = ml2; start = ip; offBase = offbaseFound; }
        }
        if (matchLength < 4) {
            size_t const step = ((size_t)(ip - anchor) >> 8) + 1;          /* kSearchStrength */
            ip += step;
            ms->lazySkipping = step > 8;                                    /* kLazySkippingStep */
            continue;
        }
        if (depth >= 1)
        while (ip < ilimit) {
            ip++;
            if ((offBase) && ((offset_1 > 0) && (rd32(ip) == rd32(ip - offset_1)))) {
                size_t const mlRep = count_eq(ip + 4, ip + 4 - offset_1, iend) + 4;
                int const gain2 = (int)(mlRep * 3);
                int const gain1 = (int)(matchLength * 3 - hb32((u32)offBase) + 1);
                if ((mlRep >= 4) && (gain2 > gain1)) { matchLength = mlRep; offBase = 1; start = ip; }
            }
            {   size_t ofbCandidate = 999999999;
                size_t const ml2 = lazy_search(ms, ip, &ofbCandidate);
                int const gain2 = (int)(ml2 * 4 - hb32((u32)ofbCandidate));
                int const gain1 = (int)(matchLength * 4 - hb32((u32)offBase) + 4);
                if ((ml2 >= 4) && (gain2 > gain1)) { matchLength = ml2; offBase = ofbCandidate; start = ip; continue; }
            }
            if ((depth == 2) && (ip < ilimit)) {
                ip++;
                if ((offBase) && ((offset_1 > 0) && (rd32(ip) == rd32(ip - offset_1)))) {
                    size_t const mlRep = count_eq(ip + 4, ip + 4 - offset_1, iend) + 4;
                    int const gain2 = (int)(mlRep * 4);
                    int const gain1 = (int)(matchLength * 4 - hb32((u32)offBase) + 1);
                    if ((mlRep >= 4) && (gain2 > gain1)) { matchLength = mlRep; offBase = 1; start = ip; }
                }
                {   size_t ofbCandidate = 999999999;
                    size_t const ml2 = lazy_search(ms, ip, &ofbCandidate);
                    int const gain2 = (int)(ml2 * 4 - hb32((u32)ofbCandidate));
                    int const gain1 = (int)(matchLength * 4 - hb32((u32)offBase) + 7);
                    if ((ml2 >= 4) && (gain2 > gain1)) { matchLength = ml2; offBase = ofbCandidate; start = ip; continue; }
                }
            }
            break;
        }
        if (offBase > 3) {              /* a real offset: catch up, then it becomes the newest repeat offset */
            size_t const off = offBase - 3;
            while (((start > anchor) & (start - off > prefixLowest)) && (start[-1] == (start - off)[-1])) { start--; matchLength++; }
            offset_2 = offset_1; offset_1 = (u32)off;
        }
_storeSequence:
        {   size_t const litLength = (size_t)(start - anchor);
            store_seq(ss, litLength, anchor, (u32)offBase, matchLength);
            anchor = ip = start + matchLength;
        }
        if (ms->lazySkipping) ms->lazySkipping = 0;
        while (((ip <= ilimit) & (offset_2 > 0)) && (rd32(ip) == rd32(ip - offset_2))) {
            matchLength = count_eq(ip + 4, ip + 4 - offset_2, iend) + 4;
            { u32 const t = offset_2; offset_2 = offset_1; offset_1 = t; }
            store_seq(ss, 0, anchor, 1, matchLength);
            ip += matchLength; anchor = ip;
        }
    }
    offsetSaved2 = ((offsetSaved1 != 0) && (offset_1 != 0)) ? offsetSaved1 : offsetSaved2;
    rep[0] = offset_1 ? offset_1 : offsetSaved1;
    rep[1] = offset_2 ? offset_2 : offsetSaved2;
    return (size_t)(iend - anchor);
}

KREF_API size_t kref_zstd_lazy_compress(u8* dst, size_t cap, const u8* src, size_t srcSize, int level)
{
    kref_lpar P; kref_lazyms ms; seqstore ss; u32 rep[3] = { 1, 4, 8 }; kref_hufstate h0, h1; kref_seq* seqs; u8* lits; u8* padded;
    size_t pos, lastLL, litC, seqC, cSize = 0; u8* body;
    if (!lazy_params(level, srcSize, &P)) return KERR;
    if (cap < kref_compress_bound(srcSize)) return KERR;
    pos = write_frame_header(dst, srcSize, P.W);
    body = dst + pos + 3;
    seqs = (kref_seq*)malloc(sizeof(kref_seq) * ((128 << 10) / 3 + 8)); lits = (u8*)malloc((128 << 10) + 32);
    padded = (u8*)malloc(srcSize + 2 + 32); memset(padded, 0, srcSize + 2 + 32); memcpy(padded + 2, src, srcSize);      /* index 2 = first byte */
    memset(&ss, 0, sizeof(ss)); ss.seqs = seqs; ss.lits = lits; ss.strategy = (int)P.strat;
    memset(&ms, 0, sizeof(ms));
    ms.base = padded; ms.iend = padded + 2 + srcSize; ms.nextToUpdate = IDX0; ms.S = P.S; ms.C = P.C; ms.H = P.H; ms.mls = P.mml < 4 ? 4 : P.mml > 6 ? 6 : P.mml;
    ms.rows = P.W > 14;
    ms.rowLog = P.S < 4 ? 4 : P.S > 6 ? 6 : P.S; ms.rowHashLog = P.H - ms.rowLog;
    ms.hashTable = (u32*)calloc((size_t)1 << P.H, sizeof(u32)); ms.chainTable = (u32*)calloc((size_t)1 << P.C, sizeof(u32)); ms.tagTable = (u8*)calloc((size_t)1 << P.H, 1);
    if (srcSize >= 7) {
        lastLL = lazy_block(&ss, rep, padded + 2, srcSize, &ms, P.strat - 3);
        memcpy(ss.lits + ss.litSize, src + srcSize - lastLL, lastLL); ss.litSize += lastLL;
        h0.valid = 0; memset(&h0.ct, 0, sizeof(h0.ct));
        {
            int const suspect = (ss.nbSeq == 0) || (ss.litSize / ss.nbSeq >= 20);
            litC = compress_literals(body, cap - pos - 3, ss.lits, ss.litSize, suspect, &h0, &h1);
            if (litC != KERR) {
                seqC = compress_sequences(body + litC, cap - pos - 3 - litC, &ss);
                if (seqC != KERR && seqC != 0) { cSize = litC + seqC; if (cSize >= srcSize - min_gain(srcSize)) cSize = 0; }
            }
        }
    }
    free(ms.hashTable); free(ms.chainTable); free(ms.tagTable); free(seqs); free(lits); free(padded);
    if (cSize == 0) { wr24(dst + pos, 1 + (0 << 1) + (u32)(srcSize << 3)); memcpy(body, src, srcSize); return pos + 3 + srcSize; }
    wr24(dst + pos, 1 + (2 << 1) + (u32)(cSize << 3));
    return pos + 3 + cSize;
}


/* ------------------------------------------------------------------ */
/* levels 4 .. 10 above 128 KiB: frames of several blocks               */
/* (ZSTD_compress_frameChunk with the lazy parsers; no product path yet */
/* -- groundwork for SURVEY 8f rank 4's remainder; the window does not  */
/* move: srcSize <= 1 << windowLog)                                     */
/* ------------------------------------------------------------------ */
/* ZSTD_splitBlock_byChunks at the levels the lazy strategies take (zstd_preSplit.c): events are a hash of two bytes sampled every
 * `rate` bytes -- greedy / lazy: rate 11, 9 bits; lazy2: rate 5, 10 bits (checked on the live library: tools/experiments/r04_presplit_levels.py) */
static size_t split_block_by_chunks_gen(const u8* p, u32 rate, u32 hashLog)
{
    static __thread u32 past[1024], nw[1024]; size_t pastN, nwN; int penalty = 3; size_t pos; u32 n; u32 const size = 1u << hashLog;
    size_t const blockSize = 128 << 10, chunk = 8 << 10, limit = chunk - 2 + 1;
#define FP_REC(tab, cnt, q) { size_t i_; memset(tab, 0, size * sizeof(u32)); for (i_ = 0; i_ < limit; i_ += rate) tab[(u32)(((u32)(q)[i_] | ((u32)(q)[i_ + 1] << 8)) * 0x9E3779B9u) >> (32 - hashLog)]++; cnt = limit / rate; }
    FP_REC(past, pastN, p)
    for (pos = chunk; pos <= blockSize - chunk; pos += chunk) {
        u64 deviation = 0, threshold;
        FP_REC(nw, nwN, p + pos)
        for (n = 0; n < size; n++) { int64_t const d = (int64_t)past[n] * (int64_t)nwN - (int64_t)nw[n] * (int64_t)pastN; deviation += (u64)(d < 0 ? -d : d); }
        threshold = (u64)pastN * (u64)nwN * (u64)(14 + penalty) / 16;
        if (deviation >= threshold) return pos;
        for (n = 0; n < size; n++) past[n] += nw[n];
        pastN += nwN;
        if (penalty > 0) penalty--;
    }
#undef FP_REC
    return blockSize;
}
/* P6: windowLog, chainLog, hashLog, searchLog, minMatch, strategy (3 greedy, 4 lazy, 5 lazy2) as ZSTD_getCParams + ZSTD_adjustCParams give them */
KREF_API size_t kref_zstd_lazy_compress_blocks(u8* dst, size_t cap, const u8* src, size_t srcSize, const u32* P6, u32* blockSizesOut, u32* nbBlocksOut)
{
    kref_lazyms ms; seqstore ss; kref_frame_state fs; kref_seqprev sp[2]; int cur = 0; kref_seq* seqs; u8* lits; u8* padded;
    size_t pos, ipos = 0; int64_t savings = 0; u32 nb = 0;
    u32 const W = P6[0], C = P6[1], H = P6[2], S = P6[3], mml = P6[4], strat = P6[5];
    if (nbBlocksOut) *nbBlocksOut = 0;
    if (srcSize > ((size_t)1 << W) || W <= 14 || cap < kref_compress_bound(srcSize)) return KERR;          /* (row-based finder only; the window never moves) */
    pos = write_frame_header(dst, srcSize, W);
    if (srcSize == 0) { wr24(dst + pos, 1); return pos + 3; }
    seqs = (kref_seq*)malloc(sizeof(kref_seq) * ((128 << 10) / 3 + 8)); lits = (u8*)malloc((128 << 10) + 32);
    padded = (u8*)malloc(srcSize + 2 + 32); memset(padded, 0, srcSize + 2 + 32); memcpy(padded + 2, src, srcSize);
    memset(&ms, 0, sizeof(ms));
    ms.base = padded; ms.nextToUpdate = IDX0; ms.S = S; ms.C = C; ms.H = H; ms.mls = mml < 4 ? 4 : mml > 6 ? 6 : mml; ms.rows = 1;
    ms.rowLog = S < 4 ? 4 : S > 6 ? 6 : S; ms.rowHashLog = H - ms.rowLog;
    ms.hashTable = (u32*)calloc((size_t)1 << H, sizeof(u32)); ms.chainTable = NULL; ms.tagTable = (u8*)calloc((size_t)1 << H, 1);
    fs.rep[0] = 1; fs.rep[1] = 4; fs.rep[2] = 8; fs.huf.valid = 0; memset(&fs.huf.ct, 0, sizeof(fs.huf.ct)); fs.isFirstBlock = 1;
    memset(sp, 0, sizeof(sp));
    g_lit_strategy = (int)strat;
    while (ipos < srcSize) {
        size_t const remaining = srcSize - ipos, blockSizeMax = 128 << 10;
        size_t const blockSize = (remaining < blockSizeMax) ? remaining : (savings < 3) ? blockSizeMax
                               : (strat >= 5 ? split_block_by_chunks_gen(src + ipos, 5, 10) : split_block_by_chunks_gen(src + ipos, 11, 9));
        u32 const lastBlock = (blockSize == remaining);
        u8* const body = dst + pos + 3; size_t const bcap = cap - pos - 3;
        const u8* const bsrc = padded + 2 + ipos;
        size_t cSize = 0; u32 rep[3]; kref_hufstate nextHuf; size_t lastLL, litC, seqC;
        memset(&ss, 0, sizeof(ss)); ss.seqs = seqs; ss.lits = lits; ss.strategy = (int)strat;
        if (blockSize >= 2 + 3 + 1 + 1) {
            u32 const curr = (u32)(bsrc - padded);
            if (curr > ms.nextToUpdate + 384) { u32 const gap = curr - ms.nextToUpdate - 384; ms.nextToUpdate = curr - (gap < 192 ? gap : 192); }      /* limited update after a very long match */
            memcpy(rep, fs.rep, sizeof(rep));
            ms.iend = bsrc + blockSize;
            lastLL = lazy_block_at(&ss, rep, bsrc, blockSize, &ms, strat - 3, padded + 2);
            memcpy(ss.lits + ss.litSize, src + ipos + blockSize - lastLL, lastLL); ss.litSize += lastLL;
            {
                int const suspect = (ss.nbSeq == 0) || (ss.litSize / ss.nbSeq >= 20);
                g_seq_prev = &sp[cur]; g_seq_next = &sp[cur ^ 1]; sp[cur ^ 1] = sp[cur];
                litC = compress_literals(body, bcap, ss.lits, ss.litSize, suspect, &fs.huf, &nextHuf);
                if (litC != KERR) {
                    seqC = compress_sequences(body + litC, bcap - litC, &ss);
                    if (seqC != KERR && seqC != 0) { cSize = litC + seqC; if (cSize >= blockSize - min_gain(blockSize)) cSize = 0; }
                }
                g_seq_prev = NULL; g_seq_next = NULL;
            }
            if (!fs.isFirstBlock && cSize < 25) {
                size_t i; int same = 1;
                for (i = 1; i < blockSize; i++) if (src[ipos + i] != src[ipos]) { same = 0; break; }
                if (same) { body[0] = src[ipos]; cSize = 1; }
            }
            if (cSize > 1) { memcpy(fs.rep, rep, sizeof(rep)); fs.huf = nextHuf; cur ^= 1; }          /* confirmRepcodesAndEntropyTables */
        }
        if (cSize == 0) { wr24(dst + pos, lastBlock + (0 << 1) + (u32)(blockSize << 3)); memcpy(body, src + ipos, blockSize); cSize = 3 + blockSize; }
        else if (cSize == 1) { wr24(dst + pos, lastBlock + (1 << 1) + (u32)(blockSize << 3)); cSize = 3 + 1; }
        else { wr24(dst + pos, lastBlock + (2 << 1) + (u32)(cSize << 3)); cSize += 3; }
        savings += (int64_t)blockSize - (int64_t)cSize;
        if (blockSizesOut) blockSizesOut[nb] = (u32)blockSize;
        nb++; ipos += blockSize; pos += cSize; fs.isFirstBlock = 0;
    }
    g_lit_strategy = 2;
    free(ms.hashTable); free(ms.tagTable); free(seqs); free(lits); free(padded);
    if (nbBlocksOut) *nbBlocksOut = nb;
    return pos;
}

/* ZSTD_getCParams(level, n, 0) + ZSTD_adjustCParams for 128 KiB < n: the "<= 256 KB" table (levels 4 .. 10) and the default one
 * (levels 5 .. 10; level 4 is double-fast there).  Read off the live library by search (which (hashLog, searchLog, minMatch, strategy)
 * reproduce its frames: all of these were unique) and equal to the tables of zstd 1.5.7's clevels.h as remembered.  0: not a lazy level here. */
static int lazy_params_big(int level, size_t n, u32* P6)
{
    u32 const srcLog = hb32((u32)(n - 1)) + 1; u32 W, H, S, mml, strat;
    if (n <= 131072 || n > ((size_t)2 << 20)) return 0;
    if (n <= 262144) {
        static const u32 Ht[11] = { 0,0,0,0, 17, 18, 19, 19, 19, 19, 19 }, St[11] = { 0,0,0,0, 3, 5, 3, 4, 4, 5, 6 }, Mt[11] = { 0,0,0,0, 5, 5, 5, 4, 4, 4, 4 }, Tt[11] = { 0,0,0,0, 3, 3, 4, 4, 5, 5, 5 };
        if (level < 4 || level > 10) return 0;
        W = 18; H = Ht[level]; S = St[level]; mml = Mt[level]; strat = Tt[level];
    } else {
        static const u32 Wt[11] = { 0,0,0,0,0, 21, 21, 21, 21, 22, 22 }, Ht[11] = { 0,0,0,0,0, 19, 19, 20, 20, 21, 22 }, St[11] = { 0,0,0,0,0, 3, 3, 4, 4, 4, 5 }, Tt[11] = { 0,0,0,0,0, 3, 4, 4, 5, 5, 5 };
        if (level < 5 || level > 10) return 0;
        W = Wt[level]; H = Ht[level]; S = St[level]; mml = 5; strat = Tt[level];
    }
    if (W > srcLog) W = srcLog;
    if (H > W + 1) H = W + 1;
    P6[0] = W; P6[1] = 16; P6[2] = H; P6[3] = S; P6[4] = mml; P6[5] = strat;
    return 1;
}
KREF_API size_t kref_zstd_lazy_compress_big(u8* dst, size_t cap, const u8* src, size_t srcSize, int level, u32* blockSizesOut, u32* nbBlocksOut)
{
    u32 P6[6];
    if (!lazy_params_big(level, srcSize, P6)) return KERR;
    return kref_zstd_lazy_compress_blocks(dst, cap, src, srcSize, P6, blockSizesOut, nbBlocksOut);
}
