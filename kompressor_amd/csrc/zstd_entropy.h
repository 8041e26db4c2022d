// zstd_entropy.h -- entropy stage + frame writer of zstd level 3, one wave per
// slice: literals section (raw / RLE / Huffman 1 or 4 streams, weights FSE- or
// 4-bit-coded), sequences section (predefined / RLE / FSE tables, interleaved
// tANS bitstream), compressed-vs-raw block decision, block + frame headers.
//
// Replaces the second half of what libzstd 1.5.7 does behind the reference's
// ZSTD_compressStream2(..., ZSTD_e_end) call
// (kompressor-zstd--nativelib/src/jvmCommonMain/jni/Wrapper.cpp:112); the bytes
// must equal that library's frame exactly, so every heuristic (table-log
// choice, count normalisation, Huffman depth limiting and tie order, encoding
// type selection, minimum-gain tests) follows the published algorithm.
//
// Execution shape: histograms, literal bit-packing and copies are wave-parallel
// (64 lanes, LDS atomics, shuffle scans, atomic-OR bit placement); the tiny
// table constructions (<= 256 symbols) and the tANS state chain are run by
// lane 0 while the other 8191 resident waves of the chip hide its latency.
#pragma once
#include "zstd_common.h"
#include "zstd_match.h"
#include "zstd_match_ext.h"
#include "zstd_match_fast.h"

struct KDictPrior;
struct KEntropyArgs {
    const u8* src; const u64* in_off; const u32* in_len; u32 n_slices;
    const KSeq* seqs; u32 seq_cap; u8* lits; u32 lit_cap; const KSliceMeta* meta;
    u32* scratch; u32 scratch_words;         // per slice: Huffman stream staging (u32 aligned)
    u8* dst; const u64* out_off; u32* out_len;
    const KDictPrior* prior = nullptr;       // k_zstd_entropy_prior only: the formatted dictionary's tables
    u32 flags;                               // timing experiments only (results become wrong): 1 no literal coding, 2 no sequence coding (timing experiments, results become wrong);
                                             // 8: the match kernel copied no literals, gather them here; 32: strategy "fast" (levels 1, 2); bits 8 .. 10: the strategy's number when it is another (3 greedy, 4 lazy, 5 lazy2);
                                             // 64: literals are left uncompressed (negative levels: ZSTD_literalsCompressionIsDisabled)
};

#define KXE_ERR 0xFFFFFFFFu

// What a formatted dictionary (magic EC30A437) brings besides its content (libzstd: ZSTD_loadCEntropy): a Huffman table for the literals,
// the three sequence tables as normalised counts, repeat offsets, an ID for the frame header.  Parsed on the host (zstd_cdict_host.h),
// one copy in device memory per loaded dictionary.  The first -- here: only -- block of a slice is coded with these as its "previous
// block": literals may be coded with the table unseen, sequence tables the dictionary marks complete are reused below 1 000 sequences.
struct KDictPrior {
    u32 ct[256];              // Huffman code table: val | nbBits << 16 (KEntropyLds.ct's form)
    u32 hufMode;              // 1: an entry may be missing (HUF_repeat_check); 2: every byte value has a code (HUF_repeat_valid)
    u32 seqValid[3];          // [0] LL, [1] OF, [2] ML: FSE_repeat_valid
    u32 maxSym[3], log[3];
    short norm[3][64];
    u32 rep[3]; u32 dictID;
};

struct KHNode { u32 count; u16 parent; u8 byte; u8 nbBits; };
struct alignas(8) KSeqDelta { u32 nb; int fs; };

// ---- LDS of one entropy wave ------------------------------------------
struct KEntropyLds {
    u32 hist[256];            // literal histogram / sequence-code histograms (3 x 64)
    u32 ct[256];              // Huffman code table: val | nbBits << 16
    union {
        struct {              // literal phase: Huffman tree construction
            KHNode node[516]; // ([0] is the sentinel before huffNode[0])
            u32 rank[192];    // bucket sort positions: curr | base << 16
            u32 qstack[40];   // explicit quicksort stack
            u8 weight[256];   // Huffman weights (behind everything their FSE coding uses of the sequence phase's arrays below: the
                              // state table, dnb[0], dfs[0] and the spread scratch end at byte 4 608 of the union, this starts at 5 056)
        } huf;
        struct {              // sequence phase (stateLL/dnb[0]/dfs[0] also serve the Huffman-weight FSE)
            u16 stateLL[512];    // FSE next-state tables
            u16 stateML[512];
            u16 stateOF[256];
            u32 dnb[3][64];      // FSE deltaNbBits
            int dfs[3][64];      // FSE deltaFindState
            u32 stage[64];       // codes of 64 staged sequences: ll | of << 8 | ml << 16
            u16 sbits[3][64];    // per staged sequence and stream: the chain's state before the sequence
            u32 cbuf[192];       // bit assembly buffer of one 64-sequence chunk
            KSeqDelta pp[3][65]; // per staged sequence and stream: deltaNbBits / deltaFindState (times two: a byte offset) of its code, looked up by
                                 //   all lanes at once so that the state chains only wait for the state table; one 8-byte read a step
        } seq;
    } u;
    short norm[3][64];
    u16 cumul[3][66];
    u8 ncbuf[3][80];          // table descriptions of the three symbol types, before concatenation
    u32 cnt[16];
};      // 10 196 bytes: sixteen waves per CU (160 KiB of LDS)

KX_DEV u16* kxe_state(KEntropyLds& lds, int t) { return t == 0 ? lds.u.seq.stateLL : t == 1 ? lds.u.seq.stateOF : lds.u.seq.stateML; }
// FSE spread scratch (LL / weights 512 B, ML 512 B, OF 256 B): the staging area of the bitstream loop (stage, sbits,
// cbuf: 1408 contiguous bytes), which is idle while tables are built
KX_DEV u8* kxe_tsym(KEntropyLds& lds, int t) { return (u8*)lds.u.seq.stage + (t == 0 ? 0 : t == 1 ? 1024 : 512); }

// ======================= lane-0 serial helpers ==========================
struct KBitW { u64 acc; u32 nb; u8* p; };
KX_DEV void kbw_init(KBitW& b, u8* dst) { b.acc = 0; b.nb = 0; b.p = dst; }
KX_DEV void kbw_add(KBitW& b, u32 v, u32 n)
{
    if (n == 0) return;
    u64 const m = (n >= 32) ? 0xFFFFFFFFull : ((1ull << n) - 1ull);
    b.acc |= ((u64)v & m) << b.nb; b.nb += n;
    if (b.nb >= 32) { kx_st32(b.p, (u32)b.acc); b.p += 4; b.acc >>= 32; b.nb -= 32; }
}
KX_DEV u32 kbw_close(KBitW& b, u8* start)
{
    kbw_add(b, 1, 1);
    while (b.nb > 0) { *b.p++ = (u8)b.acc; b.acc >>= 8; b.nb = (b.nb >= 8) ? b.nb - 8 : 0; }
    return (u32)(b.p - start);
}

KX_DEV u32 kfse_min_tablelog(u32 srcSize, u32 maxSymbolValue)
{
    u32 const a = kx_hb32(srcSize) + 1, b = kx_hb32(maxSymbolValue) + 2;
    return a < b ? a : b;
}
KX_DEV u32 kfse_optimal_tablelog(u32 maxTableLog, u32 srcSize, u32 maxSymbolValue, u32 minus)
{
    u32 const maxBitsSrc = kx_hb32(srcSize - 1) - minus;
    u32 tableLog = maxTableLog;
    u32 const minBits = kfse_min_tablelog(srcSize, maxSymbolValue);
    if (maxBitsSrc < tableLog) tableLog = maxBitsSrc;
    if (minBits > tableLog) tableLog = minBits;
    if (tableLog < 5) tableLog = 5;
    if (tableLog > 12) tableLog = 12;
    return tableLog;
}

KX_DEV u32 kfse_normalize_m2(short* norm, u32 tableLog, const u32* count, u32 total, u32 maxSymbolValue, short lowProbCount)
{
    short const NOT_YET_ASSIGNED = -2;
    u32 s, distributed = 0, ToDistribute;
    u32 const lowThreshold = total >> tableLog;
    u32 lowOne = (u32)(((u64)total * 3) >> (tableLog + 1));
    for (s = 0; s <= maxSymbolValue; s++) {
        if (count[s] == 0) { norm[s] = 0; continue; }
        if (count[s] <= lowThreshold) { norm[s] = lowProbCount; distributed++; total -= count[s]; continue; }
        if (count[s] <= lowOne) { norm[s] = 1; distributed++; total -= count[s]; continue; }
        norm[s] = NOT_YET_ASSIGNED;
    }
    ToDistribute = (1u << tableLog) - distributed;
    if (ToDistribute == 0) return 0;
    if ((total / ToDistribute) > lowOne) {
        lowOne = (u32)(((u64)total * 3) / (ToDistribute * 2));
        for (s = 0; s <= maxSymbolValue; s++) {
            if ((norm[s] == NOT_YET_ASSIGNED) && (count[s] <= lowOne)) { norm[s] = 1; distributed++; total -= count[s]; }
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
        u64 const mid = (1ull << (vStepLog - 1)) - 1;
        u64 const rStep = (((1ull << vStepLog) * ToDistribute) + mid) / total;
        u64 tmpTotal = mid;
        for (s = 0; s <= maxSymbolValue; s++) {
            if (norm[s] == NOT_YET_ASSIGNED) {
                u64 const end = tmpTotal + (count[s] * rStep);
                u32 const weight = (u32)(end >> vStepLog) - (u32)(tmpTotal >> vStepLog);
                if (weight < 1) return KXE_ERR;
                norm[s] = (short)weight;
                tmpTotal = end;
            }
        }
    }
    return 0;
}

// returns tableLog, 0 for "one symbol only", KXE_ERR on error
KX_DEV u32 kfse_normalize(short* norm, u32 tableLog, const u32* count, u32 total, u32 maxSymbolValue, u32 useLowProbCount)
{
    if (tableLog < 5 || tableLog > 12) return KXE_ERR;
    if (tableLog < kfse_min_tablelog(total, maxSymbolValue)) return KXE_ERR;
    short const lowProbCount = useLowProbCount ? -1 : 1;
    u64 const scale = 62 - tableLog;
    u64 const step = (1ull << 62) / total;
    u64 const vStep = 1ull << (scale - 20);
    int stillToDistribute = 1 << tableLog;
    u32 s, largest = 0; short largestP = 0;
    u32 const lowThreshold = total >> tableLog;
    for (s = 0; s <= maxSymbolValue; s++) {
        if (count[s] == total) return 0;
        if (count[s] == 0) { norm[s] = 0; continue; }
        if (count[s] <= lowThreshold) { norm[s] = lowProbCount; stillToDistribute--; }
        else {
            short proba = (short)((count[s] * step) >> scale);
            if (proba < 8) {
                u32 const rtb = proba == 0 ? 0u : proba == 1 ? 473195u : proba == 2 ? 504333u : proba == 3 ? 520860u
                              : proba == 4 ? 550000u : proba == 5 ? 700000u : proba == 6 ? 750000u : 830000u;
                u64 const restToBeat = vStep * rtb;
                proba += (count[s] * step) - ((u64)proba << scale) > restToBeat;
            }
            if (proba > largestP) { largestP = proba; largest = s; }
            norm[s] = proba;
            stillToDistribute -= proba;
        }
    }
    if (-stillToDistribute >= (norm[largest] >> 1)) {
        if (kfse_normalize_m2(norm, tableLog, count, total, maxSymbolValue, lowProbCount) == KXE_ERR) return KXE_ERR;
    } else norm[largest] += (short)stillToDistribute;
    return tableLog;
}

// writes the normalised counts header; returns its size (KXE_ERR on a malformed distribution)
KX_DEV u32 kfse_write_ncount(u8* dst, const short* norm, u32 maxSymbolValue, u32 tableLog)
{
    u8* out = dst;
    int nbBits; int const tableSize = 1 << tableLog;
    int remaining, threshold; u32 bitStream = 0; int bitCount = 0;
    u32 symbol = 0; u32 const alphabetSize = maxSymbolValue + 1; int previousIs0 = 0;
    bitStream += (tableLog - 5) << bitCount; bitCount += 4;
    remaining = tableSize + 1; threshold = tableSize; nbBits = (int)tableLog + 1;
    while ((symbol < alphabetSize) && (remaining > 1)) {
        if (previousIs0) {
            u32 start = symbol;
            while ((symbol < alphabetSize) && !norm[symbol]) symbol++;
            if (symbol == alphabetSize) break;
            while (symbol >= start + 24) {
                start += 24;
                bitStream += 0xFFFFu << bitCount;
                out[0] = (u8)bitStream; out[1] = (u8)(bitStream >> 8); out += 2; bitStream >>= 16;
            }
            while (symbol >= start + 3) { start += 3; bitStream += 3u << bitCount; bitCount += 2; }
            bitStream += (symbol - start) << bitCount; bitCount += 2;
            if (bitCount > 16) { out[0] = (u8)bitStream; out[1] = (u8)(bitStream >> 8); out += 2; bitStream >>= 16; bitCount -= 16; }
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
            if (remaining < 1) return KXE_ERR;
            while (remaining < threshold) { nbBits--; threshold >>= 1; }
        }
        if (bitCount > 16) { out[0] = (u8)bitStream; out[1] = (u8)(bitStream >> 8); out += 2; bitStream >>= 16; bitCount -= 16; }
    }
    if (remaining != 1) return KXE_ERR;
    out[0] = (u8)bitStream; out[1] = (u8)(bitStream >> 8);
    out += (bitCount + 7) / 8;
    return (u32)(out - dst);
}

struct KFseCT { u16* state; u32* dnb; int* dfs; u32 tableLog; };

KX_DEV void kfse_build_ctable(KFseCT& ct, const short* norm, u32 maxSymbolValue, u32 tableLog, u16* cumul, u8* tableSymbol)
{
    u32 const tableSize = 1u << tableLog, tableMask = tableSize - 1;
    u32 const step = (tableSize >> 1) + (tableSize >> 3) + 3;
    u32 const maxSV1 = maxSymbolValue + 1;
    u32 highThreshold = tableSize - 1, u;
    ct.tableLog = tableLog;
    cumul[0] = 0;
    for (u = 1; u <= maxSV1; u++) {
        if (norm[u - 1] == -1) { cumul[u] = cumul[u - 1] + 1; tableSymbol[highThreshold--] = (u8)(u - 1); }
        else cumul[u] = cumul[u - 1] + (u16)norm[u - 1];
    }
    cumul[maxSV1] = (u16)(tableSize + 1);
    {
        u32 position = 0, symbol;
        for (symbol = 0; symbol < maxSV1; symbol++) {
            int const freq = norm[symbol];
            for (int i = 0; i < freq; i++) {
                tableSymbol[position] = (u8)symbol;
                position = (position + step) & tableMask;
                while (position > highThreshold) position = (position + step) & tableMask;
            }
        }
    }
    for (u = 0; u < tableSize; u++) { u8 const s = tableSymbol[u]; ct.state[cumul[s]++] = (u16)(tableSize + u); }
    {
        u32 total = 0, s;
        for (s = 0; s <= maxSymbolValue; s++) {
            int const nc = norm[s];
            if (nc == 0) { ct.dnb[s] = ((tableLog + 1) << 16) - (1u << tableLog); ct.dfs[s] = 0; }
            else if (nc == -1 || nc == 1) { ct.dnb[s] = (tableLog << 16) - (1u << tableLog); ct.dfs[s] = (int)(total - 1); total++; }
            else {
                u32 const maxBitsOut = tableLog - kx_hb32((u32)nc - 1);
                u32 const minStatePlus = (u32)nc << maxBitsOut;
                ct.dnb[s] = (maxBitsOut << 16) - minStatePlus;
                ct.dfs[s] = (int)(total - (u32)nc);
                total += (u32)nc;
            }
        }
    }
}
KX_DEV void kfse_build_ctable_rle(KFseCT& ct, u32 symbol)
{
    ct.tableLog = 0; ct.state[0] = 0; ct.state[1] = 0; ct.dnb[symbol] = 0; ct.dfs[symbol] = 0;
}
KX_DEV u32 kfse_init_state(const KFseCT& ct, u32 symbol)
{
    u32 const dnb = ct.dnb[symbol];
    u32 const nbBitsOut = (dnb + (1u << 15)) >> 16;
    u32 const value = (nbBitsOut << 16) - dnb;
    return ct.state[(value >> nbBitsOut) + ct.dfs[symbol]];
}
KX_DEV void kfse_encode(KBitW& b, const KFseCT& ct, u32& state, u32 symbol)
{
    u32 const nbBitsOut = (state + ct.dnb[symbol]) >> 16;
    kbw_add(b, state, nbBitsOut);
    state = ct.state[(state >> nbBitsOut) + ct.dfs[symbol]];
}

// ---- Huffman table construction (lane 0) -----------------------------
KX_DEV u32 khuf_get_index(u32 count) { return (count < 165u) ? count : kx_hb32(count) + 158u; }

KX_DEV void khuf_insertion_sort(KHNode* n, int low, int high)
{
    int const size = high - low + 1; n += low;
    for (int i = 1; i < size; ++i) {
        KHNode const key = n[i]; int j = i - 1;
        while (j >= 0 && n[j].count < key.count) { n[j + 1] = n[j]; j--; }
        n[j + 1] = key;
    }
}
KX_DEV int khuf_partition(KHNode* arr, int low, int high)
{
    u32 const pivot = arr[high].count; int i = low - 1;
    for (int j = low; j < high; j++) if (arr[j].count > pivot) { i++; KHNode t = arr[i]; arr[i] = arr[j]; arr[j] = t; }
    { KHNode t = arr[i + 1]; arr[i + 1] = arr[high]; arr[high] = t; }
    return i + 1;
}
// the published sort: quicksort, last element as pivot, insertion sort below 8
// elements on entry; the explicit stack replays the recursion order
KX_DEV void khuf_quicksort(KHNode* arr, int lo0, int hi0, u32* stack)
{
    int sp = 0;
    stack[sp++] = (u32)lo0 | ((u32)hi0 << 10) | (1u << 20);
    while (sp > 0) {
        u32 const f = stack[--sp];
        int low = (int)(f & 1023u), high = (int)((f >> 10) & 1023u) ; bool const fresh = (f >> 20) & 1u;
        if ((f >> 21) & 1u) high = -1;                 // encoded "high = low - 1" underflow
        if (fresh && high - low < 8) { khuf_insertion_sort(arr, low, high); continue; }
        if (!(low < high)) continue;
        int const idx = khuf_partition(arr, low, high);
        u32 a, b;   // b is run first
        if (idx - low < high - idx) {
            a = (u32)(idx + 1) | ((u32)high << 10);                                   // continue the loop
            b = (idx - 1 < 0) ? ((u32)low | (1u << 20) | (1u << 21)) : ((u32)low | ((u32)(idx - 1) << 10) | (1u << 20));
        } else {
            a = (idx - 1 < 0) ? ((u32)low | (1u << 21)) : ((u32)low | ((u32)(idx - 1) << 10));
            b = (u32)(idx + 1) | ((u32)high << 10) | (1u << 20);
        }
        stack[sp++] = a; stack[sp++] = b;
    }
}

KX_DEV void khuf_sort(KHNode* huffNode, const u32* count, u32 maxSymbolValue, u32* rank, u32* qstack)
{
    u32 n; u32 const maxSV1 = maxSymbolValue + 1;
    // rank[i] = curr | base << 16
    for (n = 0; n < 192; n++) rank[n] = 0;
    for (n = 0; n < maxSV1; ++n) rank[khuf_get_index(count[n])] += 1u << 16;
    for (n = 191; n > 0; --n) {
        u32 const b = (rank[n - 1] >> 16) + (rank[n] >> 16);
        rank[n - 1] = b | (b << 16);
    }
    for (n = 0; n < maxSV1; ++n) {
        u32 const c = count[n]; u32 const r = khuf_get_index(c) + 1;
        u32 const pos = rank[r] & 0xFFFFu; rank[r] += 1;
        huffNode[pos].count = c; huffNode[pos].byte = (u8)n;
    }
    for (n = 165; n < 191; ++n) {
        int const bucketSize = (int)(rank[n] & 0xFFFFu) - (int)(rank[n] >> 16);
        u32 const start = rank[n] >> 16;
        if (bucketSize > 1) khuf_quicksort(huffNode + start, 0, bucketSize - 1, qstack);
    }
}

KX_DEV u32 khuf_set_max_height(KHNode* huffNode, u32 lastNonNull, u32 targetNbBits)
{
    u32 const largestBits = huffNode[lastNonNull].nbBits;
    if (largestBits <= targetNbBits) return largestBits;
    int totalCost = 0; u32 const baseCost = 1u << (largestBits - targetNbBits);
    int n = (int)lastNonNull;
    while (huffNode[n].nbBits > targetNbBits) {
        totalCost += (int)(baseCost - (1u << (largestBits - huffNode[n].nbBits)));
        huffNode[n].nbBits = (u8)targetNbBits; n--;
    }
    while (huffNode[n].nbBits == targetNbBits) --n;
    totalCost >>= (largestBits - targetNbBits);
    u32 const noSymbol = 0xF0F0F0F0u; u32 rankLast[14];
    for (int i = 0; i < 14; i++) rankLast[i] = noSymbol;
    {
        u32 currentNbBits = targetNbBits;
        for (int pos = n; pos >= 0; pos--) {
            if (huffNode[pos].nbBits >= currentNbBits) continue;
            currentNbBits = huffNode[pos].nbBits;
            rankLast[targetNbBits - currentNbBits] = (u32)pos;
        }
    }
    while (totalCost > 0) {
        u32 nBitsToDecrease = kx_hb32((u32)totalCost) + 1;
        for (; nBitsToDecrease > 1; nBitsToDecrease--) {
            u32 const highPos = rankLast[nBitsToDecrease];
            u32 const lowPos = rankLast[nBitsToDecrease - 1];
            if (highPos == noSymbol) continue;
            if (lowPos == noSymbol) break;
            if (huffNode[highPos].count <= 2 * huffNode[lowPos].count) break;
        }
        while ((nBitsToDecrease <= 12) && (rankLast[nBitsToDecrease] == noSymbol)) nBitsToDecrease++;
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
    return targetNbBits;
}

// builds lds.ct from lds.hist; returns the table's depth
KX_DEV u32 khuf_build_ctable(KEntropyLds& lds, u32 maxSymbolValue, u32 maxNbBits)
{
    KHNode* const huffNode0 = lds.u.huf.node; KHNode* const huffNode = huffNode0 + 1;
    for (int i = 0; i < 516; i++) { KHNode z; z.count = 0; z.parent = 0; z.byte = 0; z.nbBits = 0; lds.u.huf.node[i] = z; }
    khuf_sort(huffNode, lds.hist, maxSymbolValue, lds.u.huf.rank, lds.u.huf.qstack);
    int nonNullRank = (int)maxSymbolValue;
    while (huffNode[nonNullRank].count == 0) nonNullRank--;
    int lowS = nonNullRank, nodeNb = 256; int const nodeRoot = nodeNb + lowS - 1; int lowN = nodeNb, n;
    huffNode[nodeNb].count = huffNode[lowS].count + huffNode[lowS - 1].count;
    huffNode[lowS].parent = huffNode[lowS - 1].parent = (u16)nodeNb;
    nodeNb++; lowS -= 2;
    for (n = nodeNb; n <= nodeRoot; n++) huffNode[n].count = 1u << 30;
    huffNode0[0].count = 1u << 31;
    while (nodeNb <= nodeRoot) {
        int const n1 = (huffNode[lowS].count < huffNode[lowN].count) ? lowS-- : lowN++;
        int const n2 = (huffNode[lowS].count < huffNode[lowN].count) ? lowS-- : lowN++;
        huffNode[nodeNb].count = huffNode[n1].count + huffNode[n2].count;
        huffNode[n1].parent = huffNode[n2].parent = (u16)nodeNb;
        nodeNb++;
    }
    huffNode[nodeRoot].nbBits = 0;
    for (n = nodeRoot - 1; n >= 256; n--) huffNode[n].nbBits = (u8)(huffNode[huffNode[n].parent].nbBits + 1);
    for (n = 0; n <= nonNullRank; n++) huffNode[n].nbBits = (u8)(huffNode[huffNode[n].parent].nbBits + 1);
    maxNbBits = khuf_set_max_height(huffNode, (u32)nonNullRank, maxNbBits);
    {
        u32 nbPerRank[13], valPerRank[13];
        for (n = 0; n < 13; n++) { nbPerRank[n] = 0; valPerRank[n] = 0; }
        int const alphabetSize = (int)(maxSymbolValue + 1);
        for (n = 0; n <= nonNullRank; n++) nbPerRank[huffNode[n].nbBits]++;
        { u32 min = 0; for (n = (int)maxNbBits; n > 0; n--) { valPerRank[n] = min; min += nbPerRank[n]; min >>= 1; } }
        for (n = 0; n < 256; n++) lds.ct[n] = 0;
        for (n = 0; n < alphabetSize; n++) lds.ct[huffNode[n].byte] = (u32)huffNode[n].nbBits << 16;
        for (n = 0; n < alphabetSize; n++) { u32 const nb = lds.ct[n] >> 16; lds.ct[n] = (nb << 16) | (valPerRank[nb]++ & 0xFFFFu); }
    }
    return maxNbBits;
}

// FSE-compress the Huffman weights. 0 = not compressible, 1 = single symbol, KXE_ERR = error
KX_DEV u32 khuf_compress_weights(u8* dst, KEntropyLds& lds, u32 wtSize)
{
    u8* op = dst; u32 maxSymbolValue = 12; u32 tableLog = 6;
    const u8* const weightTable = lds.u.huf.weight;
    if (wtSize <= 1) return 0;
    {
        u32 maxCount = 0;
        for (u32 s = 0; s <= 12; s++) lds.cnt[s] = 0;
        for (u32 i = 0; i < wtSize; i++) lds.cnt[weightTable[i]]++;
        while (!lds.cnt[maxSymbolValue]) maxSymbolValue--;
        for (u32 s = 0; s <= maxSymbolValue; s++) if (lds.cnt[s] > maxCount) maxCount = lds.cnt[s];
        if (maxCount == wtSize) return 1;
        if (maxCount == 1) return 0;
    }
    tableLog = kfse_optimal_tablelog(tableLog, wtSize, maxSymbolValue, 2);
    if (kfse_normalize(lds.norm[0], tableLog, lds.cnt, wtSize, maxSymbolValue, 0) == KXE_ERR) return KXE_ERR;
    { u32 const h = kfse_write_ncount(op, lds.norm[0], maxSymbolValue, tableLog); if (h == KXE_ERR) return KXE_ERR; op += h; }
    KFseCT ct; ct.state = lds.u.seq.stateLL; ct.dnb = lds.u.seq.dnb[0]; ct.dfs = lds.u.seq.dfs[0];
    kfse_build_ctable(ct, lds.norm[0], maxSymbolValue, tableLog, lds.cumul[0], kxe_tsym(lds, 0));
    {
        KBitW b; const u8* ip = weightTable + wtSize; u32 s1, s2;
        if (wtSize <= 2) return 0;
        kbw_init(b, op);
        if (wtSize & 1) { s1 = kfse_init_state(ct, *--ip); s2 = kfse_init_state(ct, *--ip); kfse_encode(b, ct, s1, *--ip); }
        else { s2 = kfse_init_state(ct, *--ip); s1 = kfse_init_state(ct, *--ip); }
        while (ip > weightTable) { kfse_encode(b, ct, s2, *--ip); kfse_encode(b, ct, s1, *--ip); }
        kbw_add(b, s2, tableLog); kbw_add(b, s1, tableLog);
        op += kbw_close(b, op);
    }
    return (u32)(op - dst);
}

// Huffman tree description; returns its size or KXE_ERR
KX_DEV u32 khuf_write_ctable(u8* dst, KEntropyLds& lds, u32 maxSymbolValue, u32 huffLog)
{
    for (u32 n = 0; n < maxSymbolValue; n++) { u32 const nb = lds.ct[n] >> 16; lds.u.huf.weight[n] = (u8)(nb ? huffLog + 1 - nb : 0); }
    {
        u32 const hSize = khuf_compress_weights(dst + 1, lds, maxSymbolValue);
        if (hSize == KXE_ERR) return KXE_ERR;
        if ((hSize > 1) & (hSize < maxSymbolValue / 2)) { dst[0] = (u8)hSize; return hSize + 1; }
    }
    if (maxSymbolValue > (256 - 128)) return KXE_ERR;
    dst[0] = (u8)(128 + (maxSymbolValue - 1));
    lds.u.huf.weight[maxSymbolValue] = 0;
    for (u32 n = 0; n < maxSymbolValue; n += 2) dst[(n / 2) + 1] = (u8)((lds.u.huf.weight[n] << 4) + lds.u.huf.weight[n + 1]);
    return ((maxSymbolValue + 1) / 2) + 1;
}

// ======================= wave-parallel pieces ===========================
KX_DEV u32 kx_wave_max(u32 v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { u32 const t = kx_shfl(v, lane ^ o); v = t > v ? t : v; }
    return v;
}
KX_DEV u32 kx_wave_sum(u32 v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v += kx_shfl(v, lane ^ o);
    return v;
}

// byte histogram of p[0..n) into lds.hist (zeroed here); returns the largest count
KX_DEV u32 kx_wave_hist(KEntropyLds& lds, const u8* p, u32 n, int lane)
{
    for (int i = lane; i < 256; i += 64) lds.hist[i] = 0;
    kx_sync();
    for (u32 i = (u32)lane * 8u; i + 8 <= n; i += 512u) {
        u64 const w = kx_ld64(p + i);
#pragma unroll
        for (int b = 0; b < 8; b++) kx_lds_inc(&lds.hist[(w >> (8 * b)) & 0xFF]);
    }
    if (lane < (int)(n & 7u)) kx_lds_inc(&lds.hist[p[(n & ~7u) + lane]]);
    kx_sync();
    u32 m = 0;
    for (int s = lane; s < 256; s += 64) { u32 const c = lds.hist[s]; m = c > m ? c : m; }
    return kx_wave_max(m, lane);
}

KX_DEV void kx_wave_copy(u8* dst, const u8* src, u32 n, int lane)
{
    u32 i = (u32)lane * 8u;
    for (; i + 8 <= n; i += 512u) kx_st64(dst + i, kx_ld64(src + i));
    u32 const tail = n & ~7u;
    if (lane < (int)(n - tail)) dst[tail + lane] = src[tail + lane];
}

KX_DEV u32 kx_min_gain(u32 srcSize) { return (srcSize >> 6) + 2; }

KX_DEV u32 klit_header_raw_rle(u8* dst, u32 type, u32 litSize)
{
    u32 const flSize = 1 + (litSize > 31) + (litSize > 4095);
    if (flSize == 1) dst[0] = (u8)(type + (litSize << 3));
    else if (flSize == 2) kx_st16(dst, type + (1u << 2) + (litSize << 4));
    else { u32 const h = type + (3u << 2) + (litSize << 4); dst[0] = (u8)h; dst[1] = (u8)(h >> 8); dst[2] = (u8)(h >> 16); }
    return flSize;
}

// Huffman-code `lits[0..litSize)` into `op` (after the tree description of hSize
// bytes that lane 0 already wrote at op - hSize).  Returns the stream bytes
// (incl. jump table), 0 if a stream does not fit its 16-bit size field.
KX_DEV u32 khuf_encode_streams(KEntropyLds& lds, u8* op, const u8* lits, u32 litSize, bool single,
                               u32* scratch, int lane)
{
    int const LPS = single ? 64 : 16;
    int const s = lane / LPS, j = lane % LPS;
    u32 const seg = single ? litSize : (litSize + 3) / 4;
    u32 const sBeg = (u32)s * seg;
    u32 const sEnd = single ? litSize : ((s == 3) ? litSize : sBeg + seg);
    u32 const ns = sEnd - sBeg;
    u32 const chunk = (ns + (u32)LPS - 1) / (u32)LPS;
    u32 b0 = sBeg + (u32)j * chunk; if (b0 > sEnd) b0 = sEnd;
    u32 b1 = b0 + chunk; if (b1 > sEnd) b1 = sEnd;
    // pass 1: bits of my chunk
    u32 bits = 0;
    {
        u32 i = b0;
        for (; i + 8 <= b1; i += 8) {
            u64 const w = kx_ld64(lits + i);
#pragma unroll
            for (int b = 0; b < 8; b++) bits += lds.ct[(w >> (8 * b)) & 0xFF] >> 16;
        }
        for (; i < b1; i++) bits += lds.ct[lits[i]] >> 16;
    }
    // suffix sums inside the stream's lane group
    u32 v = bits;
    for (int o = 1; o < LPS; o <<= 1) { u32 const t = kx_shfl(v, lane + o); if (j + o < LPS) v += t; }
    u32 const after = v - bits;
    u32 const c0 = (kx_shfl(v, 0) + 8) >> 3;
    u32 const c1 = (kx_shfl(v, 16) + 8) >> 3;
    u32 const c2 = (kx_shfl(v, 32) + 8) >> 3;
    u32 const c3 = (kx_shfl(v, 48) + 8) >> 3;
    u32 total, soff;
    if (single) { total = c0; soff = 0; }
    else {
        if (c0 > 65535 || c1 > 65535 || c2 > 65535 || c3 > 65535) return 0;
        total = c0 + c1 + c2 + c3;
        soff = (s > 0 ? c0 : 0) + (s > 1 ? c1 : 0) + (s > 2 ? c2 : 0);
    }
    u32 const words = (total + 3) >> 2;
    for (u32 w = (u32)lane; w < words; w += 64) scratch[w] = 0;
    kx_sync();
    // pass 2: place my chunk's codes, last symbol first, LSB first
    {
        u32 const absBit = 8u * soff + after;
        u32 wi = absBit >> 5; u32 fill = absBit & 31u; u64 acc = 0; bool first = true;
#define KX_PUT_SYM(sym) { u32 const ce = lds.ct[(sym)]; acc |= (u64)(ce & 0xFFFFu) << fill; fill += ce >> 16; \
            if (fill >= 32) { if (first) kx_atomic_or(&scratch[wi], (u32)acc); else scratch[wi] = (u32)acc; \
                              first = false; wi++; acc >>= 32; fill -= 32; } }
        u32 i = b1;
        for (; i >= b0 + 8; i -= 8) {
            u64 const w = kx_ld64(lits + i - 8);
#pragma unroll
            for (int b = 7; b >= 0; b--) KX_PUT_SYM((w >> (8 * b)) & 0xFF)
        }
        for (; i > b0; i--) KX_PUT_SYM(lits[i - 1])
        if (j == 0) { acc |= 1ull << fill; fill++; }        // end mark of the stream
        if (fill >= 32) { if (first) kx_atomic_or(&scratch[wi], (u32)acc); else scratch[wi] = (u32)acc; first = false; wi++; acc >>= 32; fill -= 32; }
#undef KX_PUT_SYM
        if (fill > 0 && (u32)acc != 0) kx_atomic_or(&scratch[wi], (u32)acc);
    }
    kx_sync();
    // move to the frame
    u8* const sdst = single ? op : op + 6;
    if (!single && lane == 0) { kx_st16(op, c0); kx_st16(op + 2, c1); kx_st16(op + 4, c2); }
    for (u32 w = (u32)lane; w < (total >> 2); w += 64) kx_st32(sdst + 4 * w, scratch[w]);
    if (lane < (int)(total & 3u)) sdst[(total & ~3u) + lane] = (u8)(scratch[total >> 2] >> (8 * lane));
    return single ? total : total + 6;
}

// Huffman table of an earlier block of the same frame (block mode only; libzstd: prevCBlock->entropy.huf with
// repeatMode HUF_repeat_check).  `newCt` receives the table built for this block when it is the one used.
struct KHufPrev { const u32* ct; bool valid; u32* newCt; u32 outcome; bool complete = false; };    // outcome: 0 raw/rle, 2 new table, 3 old table kept
// (complete: the table codes every byte value -- a formatted dictionary's, libzstd's HUF_repeat_valid: used without a look at the
// histogram for small inputs, never validated, and literals from 6 bytes on are worth coding; newCt may be null)

// literals section at `dst`; returns its size (uniform across the wave)
KX_DEV u32 kzstd_literals(KEntropyLds& lds, u8* dst, const u8* lits, u32 litSize, bool suspect, u32* scratch, int lane,
                          KHufPrev* prev = nullptr, bool disabled = false)
{
    u32 const lhSize = 3 + (litSize >= 1024) + (litSize >= 16384);
    bool const complete = prev && prev->valid && prev->complete;
    bool const single = litSize < 256 || (complete && lhSize == 3);
    u32 hType = 2;
    u32 cLit = 0;      // 0 => raw, 1 => rle
    bool oneByte = false;       // cLit == 1 is a real one-byte stream, not the run-length mark
    if (complete && litSize >= 6 && litSize <= 1024 && !disabled) {
        // HUF_compress_internal's first heuristic: a valid old table and a small input (HUF_flags_preferRepeat): coded with it, unseen
        for (int sy = lane; sy < 256; sy += 64) lds.ct[sy] = prev->ct[sy];
        kx_sync();
        u32 const sz = khuf_encode_streams(lds, dst + lhSize, lits, litSize, single, scratch, lane);
        hType = 3;
        if (sz != 0 && sz < litSize - 1) cLit = sz;
        if (cLit == 0 || cLit >= litSize - kx_min_gain(litSize)) cLit = 0;
        else if (cLit == 1) {
            // ZSTD_compressLiterals: a size of 1 usually means "one symbol"; below 8 bytes it may be a real stream, and then the bytes decide
            bool const same = kx_all((u32)lane >= litSize || lits[lane] == lits[0]);
            if (!(litSize >= 8 || same)) oneByte = true;
        }
    } else
    if (litSize >= (complete ? 6u : 64u) && !disabled) {
        bool go = true;
        if (suspect && litSize >= 40960) {
            u32 const lb = kx_wave_hist(lds, lits, 4096, lane);
            u32 const le = kx_wave_hist(lds, lits + litSize - 4096, 4096, lane);
            if (lb + le <= ((2 * 4096) >> 7) + 4) go = false;
        }
        if (go) {
            u32 const largest = kx_wave_hist(lds, lits, litSize, lane);
            if (largest == litSize) cLit = 1;
            else if (largest > (litSize >> 7) + 4) {
                u32 hSize = 0, useOld = 0;
                if (lane == 0) {
                    u32 maxSymbolValue = 255;
                    while (!lds.hist[maxSymbolValue]) maxSymbolValue--;
                    // HUF_validateCTable: the old table must code every symbol present
                    bool repeat = prev && prev->valid;
                    if (repeat && !prev->complete) for (u32 sy = 0; sy <= maxSymbolValue; sy++) if (lds.hist[sy] != 0 && (prev->ct[sy] >> 16) == 0) { repeat = false; break; }
                    if (repeat && litSize <= 1024) useOld = 1;          // HUF_flags_preferRepeat (strategy < lazy)
                    else {
                        u32 huffLog = kfse_optimal_tablelog(11, litSize, maxSymbolValue, 1);
                        huffLog = khuf_build_ctable(lds, maxSymbolValue, huffLog);
                        hSize = khuf_write_ctable(dst + lhSize, lds, maxSymbolValue, huffLog);
                        if (repeat && hSize != KXE_ERR) {
                            // HUF_estimateCompressedSize of both tables
                            u32 oldBits = 0, newBits = 0;
                            for (u32 sy = 0; sy <= maxSymbolValue; sy++) { oldBits += (prev->ct[sy] >> 16) * lds.hist[sy]; newBits += (lds.ct[sy] >> 16) * lds.hist[sy]; }
                            if ((oldBits >> 3) <= hSize + (newBits >> 3) || hSize + 12 >= litSize) useOld = 1;
                        }
                        if (!useOld && prev && prev->newCt && hSize != KXE_ERR && hSize + 12 < litSize)
                            for (int sy = 0; sy < 256; sy++) prev->newCt[sy] = lds.ct[sy];       // "save new table"
                    }
                    if (useOld) { for (int sy = 0; sy < 256; sy++) lds.ct[sy] = prev->ct[sy]; hSize = 0; }
                }
                hSize = kx_shfl(hSize, 0); useOld = kx_shfl(useOld, 0);
                if (useOld) hType = 3;
                kx_sync();
                if (hSize != KXE_ERR && (useOld || hSize + 12 < litSize)) {
                    u32 const sz = khuf_encode_streams(lds, dst + lhSize + hSize, lits, litSize, single, scratch, lane);
                    if (sz != 0 && hSize + sz < litSize - 1) cLit = hSize + sz;
                }
            }
        }
        if (cLit != 1 && (cLit == 0 || cLit >= litSize - kx_min_gain(litSize))) cLit = 0;
    }
    if (prev) prev->outcome = (cLit > 1 || oneByte) ? hType : 0u;
    if (cLit == 0) {
        u32 fl = 0;
        if (lane == 0) fl = klit_header_raw_rle(dst, 0, litSize);
        fl = kx_shfl(fl, 0);
        kx_sync();
        kx_wave_copy(dst + fl, lits, litSize, lane);
        return fl + litSize;
    }
    if (cLit == 1 && !oneByte) {
        u32 fl = 0;
        if (lane == 0) { fl = klit_header_raw_rle(dst, 1, litSize); dst[fl] = lits[0]; }
        fl = kx_shfl(fl, 0);
        return fl + 1;
    }
    if (lane == 0) {
        if (lhSize == 3) { u32 const h = hType + ((u32)(!single) << 2) + (litSize << 4) + (cLit << 14); dst[0] = (u8)h; dst[1] = (u8)(h >> 8); dst[2] = (u8)(h >> 16); }
        else if (lhSize == 4) kx_st32(dst, hType + (2u << 2) + (litSize << 4) + (cLit << 18));
        else { kx_st32(dst, hType + (3u << 2) + (litSize << 4) + (cLit << 22)); dst[4] = (u8)(cLit >> 10); }
    }
    return lhSize + cLit;
}

// ---- sequences -----------------------------------------------------------
KX_DEV u32 kx_ll_code(u32 litLength)
{
    static const u8 LL_Code[64] = { 0,1,2,3,4,5,6,7, 8,9,10,11,12,13,14,15, 16,16,17,17,18,18,19,19,
        20,20,20,20,21,21,21,21, 22,22,22,22,22,22,22,22, 23,23,23,23,23,23,23,23,
        24,24,24,24,24,24,24,24, 24,24,24,24,24,24,24,24 };
    return (litLength > 63) ? kx_hb32(litLength) + 19 : LL_Code[litLength];
}
KX_DEV u32 kx_ml_code(u32 mlBase)
{
    static const u8 ML_Code[128] = { 0,1,2,3,4,5,6,7, 8,9,10,11,12,13,14,15, 16,17,18,19,20,21,22,23, 24,25,26,27,28,29,30,31,
        32,32,33,33,34,34,35,35, 36,36,36,36,37,37,37,37, 38,38,38,38,38,38,38,38, 39,39,39,39,39,39,39,39,
        40,40,40,40,40,40,40,40, 40,40,40,40,40,40,40,40, 41,41,41,41,41,41,41,41, 41,41,41,41,41,41,41,41,
        42,42,42,42,42,42,42,42, 42,42,42,42,42,42,42,42, 42,42,42,42,42,42,42,42, 42,42,42,42,42,42,42,42 };
    return (mlBase > 127) ? kx_hb32(mlBase) + 36 : ML_Code[mlBase];
}
KX_DEV u32 kx_ll_bits(u32 c)
{
    static const u8 LL_bits[36] = { 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0, 1,1,1,1,2,2,3,3, 4,6,7,8,9,10,11,12, 13,14,15,16 };
    return LL_bits[c];
}
KX_DEV u32 kx_ml_bits(u32 c)
{
    static const u8 ML_bits[53] = { 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,
        1,1,1,1,2,2,3,3, 4,4,5,7,8,9,10,11, 12,13,14,15,16 };
    return ML_bits[c];
}

struct KSeqCodes { u32 ll, of, ml; };
KX_DEV KSeqCodes kx_seq_codes(const KSeq& q, u32 idx, u32 longType, u32 longPos)
{
    KSeqCodes c;
    c.ll = kx_ll_code(q.litLength); c.of = kx_hb32(q.offBase); c.ml = kx_ml_code(q.mlBase);
    if (longType == 1 && idx == longPos) c.ll = 35;
    if (longType == 2 && idx == longPos) c.ml = 52;
    return c;
}

enum { KSET_BASIC = 0, KSET_RLE = 1, KSET_COMPRESSED = 2 };

KX_DEV u32 kx_select_encoding(u32 mostFrequent, u32 nbSeq, u32 defaultNormLog, bool isDefaultAllowed, u32 mult)
{
    if (mostFrequent == nbSeq) return (isDefaultAllowed && nbSeq <= 2) ? KSET_BASIC : KSET_RLE;
    if (isDefaultAllowed) {
        u32 const dynamicFse_nbSeq_min = ((1u << defaultNormLog) * mult) >> 3;   // mult = 10 - strategy: 8 for dfast (level 3), 9 for fast (levels 1, 2)
        if ((nbSeq < dynamicFse_nbSeq_min) || (mostFrequent < (nbSeq >> (defaultNormLog - 1)))) return KSET_BASIC;
    }
    return KSET_COMPRESSED;
}

// one lane per symbol type (t = 0 LL, 1 OF, 2 ML): choose the mode, write the table
// description into lds.ncbuf[t], build the encoding table. Returns description
// bytes (KXE_ERR on error).
// ---- ZSTD_selectEncodingType from strategy "lazy" on: the candidates are priced, the cheapest is taken ---------------------
// kInverseProbabilityLog256[x] = floor(-log2(x / 256) * 256): the cost, in 1/256 bits, of a symbol of probability x / 256
KX_DEV u32 kx_inv_prob_log256(u32 x)
{
    static const u16 tab[256] = {
        0, 2048, 1792, 1642, 1536, 1453, 1386, 1329, 1280, 1236, 1197, 1162, 1130, 1100, 1073, 1047,
        1024, 1001, 980, 960, 941, 923, 906, 889, 874, 859, 844, 830, 817, 804, 791, 779,
        768, 756, 745, 734, 724, 714, 704, 694, 685, 676, 667, 658, 650, 642, 633, 626,
        618, 610, 603, 595, 588, 581, 574, 567, 561, 554, 548, 542, 535, 529, 523, 517,
        512, 506, 500, 495, 489, 484, 478, 473, 468, 463, 458, 453, 448, 443, 438, 434,
        429, 424, 420, 415, 411, 407, 402, 398, 394, 390, 386, 382, 377, 373, 370, 366,
        362, 358, 354, 350, 347, 343, 339, 336, 332, 329, 325, 322, 318, 315, 311, 308,
        305, 302, 298, 295, 292, 289, 286, 282, 279, 276, 273, 270, 267, 264, 261, 258,
        256, 253, 250, 247, 244, 241, 239, 236, 233, 230, 228, 225, 222, 220, 217, 215,
        212, 209, 207, 204, 202, 199, 197, 194, 192, 190, 187, 185, 182, 180, 178, 175,
        173, 171, 168, 166, 164, 162, 159, 157, 155, 153, 151, 149, 146, 144, 142, 140,
        138, 136, 134, 132, 130, 128, 126, 123, 121, 119, 117, 115, 114, 112, 110, 108,
        106, 104, 102, 100, 98, 96, 94, 93, 91, 89, 87, 85, 83, 82, 80, 78,
        76, 74, 73, 71, 69, 67, 66, 64, 62, 61, 59, 57, 55, 54, 52, 50,
        49, 47, 46, 44, 42, 41, 39, 37, 36, 34, 33, 31, 30, 28, 26, 25,
        23, 22, 20, 19, 17, 16, 14, 13, 11, 10, 8, 7, 5, 4, 2, 1 };
    return tab[x];
}
// ZSTD_entropyCost: the block's own distribution, probabilities in 1/256 (a symbol that occurs counts at least 1 / 256)
KX_DEV u32 kx_entropy_cost(const u32* count, u32 max, u32 total)
{
    u32 cost = 0;
    for (u32 s = 0; s <= max; s++) { u32 norm = (256u * count[s]) / total; if (count[s] != 0 && norm == 0) norm = 1; cost += count[s] * kx_inv_prob_log256(norm); }
    return cost >> 8;
}
// ZSTD_crossEntropyCost: coded with the default table
KX_DEV u32 kx_cross_entropy_cost(const short* dnorm, u32 accuracyLog, const u32* count, u32 max)
{
    u32 const shift = 8 - accuracyLog; u32 cost = 0;
    for (u32 s = 0; s <= max; s++) { u32 const normAcc = (dnorm[s] != -1) ? (u32)dnorm[s] : 1u; cost += count[s] * kx_inv_prob_log256(normAcc << shift); }
    return cost >> 8;
}

#define KSET_REPEAT 3u
KX_DEV u32 kx_build_seq_table(KEntropyLds& lds, int t, u32* count, u32 nbSeq, u32 lastCode, u32 firstCode,
                              u32& typeOut, KFseCT& ct, u32 strat, const KDictPrior* prior = nullptr)
{
    u32 const mult = 10u - strat;             // (ZSTD_fast 1, ZSTD_dfast 2, ZSTD_greedy 3; from ZSTD_lazy = 4 on the choice is by price)
    static const short LL_defaultNorm[36] = { 4,3,2,2,2,2,2,2, 2,2,2,2,2,1,1,1, 2,2,2,2,2,2,2,2, 2,3,2,1,1,1,1,1, -1,-1,-1,-1 };
    static const short ML_defaultNorm[53] = { 1,4,3,2,2,2,2,2, 2,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1,
                                              1,1,1,1,1,1,1,1, 1,1,1,1,1,1,-1,-1, -1,-1,-1,-1,-1 };
    static const short OF_defaultNorm[29] = { 1,1,1,1,1,1,2,2, 2,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1, -1,-1,-1,-1,-1 };
    u32 const maxCode = (t == 0) ? 35u : (t == 1) ? 31u : 52u;
    u32 const FSELog = (t == 1) ? 8u : 9u;
    u32 const defaultNormLog = (t == 1) ? 5u : 6u;
    u32 const defaultMax = (t == 0) ? 35u : (t == 1) ? 28u : 52u;
    u8* const op = lds.ncbuf[t]; short* const norm = lds.norm[t];
    u32 max = maxCode, mostFrequent = 0;
    while (max > 0 && !count[max]) max--;
    for (u32 s = 0; s <= max; s++) if (count[s] > mostFrequent) mostFrequent = count[s];
    bool const defaultAllowed = (t != 1) || (max <= 28);
    const short* const dnorm = (t == 0) ? LL_defaultNorm : (t == 1) ? OF_defaultNorm : ML_defaultNorm;
    u32 type = kx_select_encoding(mostFrequent, nbSeq, defaultNormLog, defaultAllowed, mult);
    if (strat >= 4u && mostFrequent != nbSeq) {
        // the default table against a table of the block's own: its description (ZSTD_NCountCost: normalised over all nbSeq codes, written
        // out to be measured) plus the entropy of the counts; no previous table in a first block
        u32 const basicCost = defaultAllowed ? kx_cross_entropy_cost(dnorm, defaultNormLog, count, max) : 0xFFFFFFFFu;
        u32 const tl = kfse_optimal_tablelog(FSELog, nbSeq, max, 2);
        type = KSET_COMPRESSED;
        if (kfse_normalize(norm, tl, count, nbSeq, max, nbSeq >= 2048) != KXE_ERR) {
            u32 const nc = kfse_write_ncount(op, norm, max, tl);
            if (nc != KXE_ERR && basicCost <= (nc << 3) + kx_entropy_cost(count, max, nbSeq)) type = KSET_BASIC;
        }
    }
    // ZSTD_selectEncodingType below strategy "lazy": a table that is valid as it stands (only a dictionary's can be) is reused for fewer than
    // 1 000 sequences, unless one code makes up the whole block or the default table is not allowed
    if (prior && prior->seqValid[t] && mostFrequent != nbSeq && defaultAllowed && nbSeq < 1000u) type = KSET_REPEAT;
    typeOut = type;
    ct.state = kxe_state(lds, t); ct.dnb = lds.u.seq.dnb[t]; ct.dfs = lds.u.seq.dfs[t]; ct.tableLog = 0;
    if (type == KSET_REPEAT) {
        u32 const pm = prior->maxSym[t];
        for (u32 s = 0; s <= pm; s++) norm[s] = prior->norm[t][s];
        kfse_build_ctable(ct, norm, pm, prior->log[t], lds.cumul[t], kxe_tsym(lds, t));
        return 0;
    }
    if (type == KSET_RLE) { kfse_build_ctable_rle(ct, max); *op = (u8)firstCode; return 1; }
    if (type == KSET_BASIC) {
        for (u32 s = 0; s <= defaultMax; s++) norm[s] = dnorm[s];
        kfse_build_ctable(ct, norm, defaultMax, defaultNormLog, lds.cumul[t], kxe_tsym(lds, t));
        return 0;
    }
    {
        u32 nbSeq_1 = nbSeq;
        u32 const tableLog = kfse_optimal_tablelog(FSELog, nbSeq, max, 2);
        if (count[lastCode] > 1) { count[lastCode]--; nbSeq_1--; }
        if (kfse_normalize(norm, tableLog, count, nbSeq_1, max, nbSeq_1 >= 2048) == KXE_ERR) return KXE_ERR;
        u32 const NCountSize = kfse_write_ncount(op, norm, max, tableLog);
        if (NCountSize == KXE_ERR) return KXE_ERR;
        kfse_build_ctable(ct, norm, max, tableLog, lds.cumul[t], kxe_tsym(lds, t));
        return NCountSize;
    }
}

// OR `n` (<= 32) bits of v into the LDS bit buffer at bit position pos
KX_DEV void kx_cbuf_put(u32* cbuf, u32 pos, u32 v, u32 n)
{
    if (n == 0) return;
    u64 const x = ((u64)v & ((n >= 32) ? 0xFFFFFFFFull : ((1ull << n) - 1ull))) << (pos & 31u);
    u32 const w = pos >> 5;
    kx_lds_or(&cbuf[w], (u32)x);
    if ((u32)(x >> 32)) kx_lds_or(&cbuf[w + 1], (u32)(x >> 32));
}

// sequences section at dst; returns size, 0 => "emit a raw block instead"
// `cap`: bytes the section may take before the block is certain to be emitted raw (block size minus the literals
// section): the writer stops there, so a pathological block can never run past the slice's output room.
KX_DEV u32 kzstd_sequences(KEntropyLds& lds, u8* dst, const KSeq* seqs, u32 nbSeq, u32 longType, u32 longPos, int lane, u32 cap, u32 xflags = 0, const KDictPrior* prior = nullptr)
{
    u32 hdr = 0;
    if (lane == 0) {
        if (nbSeq < 128) { dst[0] = (u8)nbSeq; hdr = 1; }
        else if (nbSeq < 0x7F00) { dst[0] = (u8)((nbSeq >> 8) + 0x80); dst[1] = (u8)nbSeq; hdr = 2; }
        else { dst[0] = 0xFF; kx_st16(dst + 1, nbSeq - 0x7F00); hdr = 3; }
    }
    hdr = kx_shfl(hdr, 0);
    if (nbSeq == 0) return hdr;
    // code histograms: LL at [0,64), OF at [64,128), ML at [128,192)
    for (int i = lane; i < 192; i += 64) lds.hist[i] = 0;
    kx_sync();
    for (u32 i = (u32)lane; i < nbSeq; i += 64) {
        KSeqCodes const c = kx_seq_codes(seqs[i], i, longType, longPos);
        kx_lds_inc(&lds.hist[c.ll]); kx_lds_inc(&lds.hist[64 + c.of]); kx_lds_inc(&lds.hist[128 + c.ml]);
    }
    kx_sync();
    // the three tables, one lane each
    KFseCT ct; ct.state = lds.u.seq.stateLL; ct.dnb = lds.u.seq.dnb[0]; ct.dfs = lds.u.seq.dfs[0]; ct.tableLog = 0;
    u32 mySz = 0, myType = 0;
    {
        KSeqCodes const cl = kx_seq_codes(seqs[nbSeq - 1], nbSeq - 1, longType, longPos);
        KSeqCodes const cf = kx_seq_codes(seqs[0], 0, longType, longPos);
        if (lane < 3) {
            u32 const lastCode = lane == 0 ? cl.ll : lane == 1 ? cl.of : cl.ml;
            u32 const firstCode = lane == 0 ? cf.ll : lane == 1 ? cf.of : cf.ml;
            mySz = kx_build_seq_table(lds, lane, lds.hist + 64 * lane, nbSeq, lastCode, firstCode, myType, ct, (xflags >> 8) ? (xflags >> 8) & 7u : ((xflags & 32u) ? 1u : 2u), prior);
        }
    }
    kx_sync();
    u32 const sz0 = kx_shfl(mySz, 0), sz1 = kx_shfl(mySz, 1), sz2 = kx_shfl(mySz, 2);
    u32 const ty0 = kx_shfl(myType, 0), ty1 = kx_shfl(myType, 1), ty2 = kx_shfl(myType, 2);
    u32 const tl0 = kx_shfl(ct.tableLog, 0), tl1 = kx_shfl(ct.tableLog, 1), tl2 = kx_shfl(ct.tableLog, 2);
    if (sz0 == KXE_ERR || sz1 == KXE_ERR || sz2 == KXE_ERR) return 0;
    u8* op = dst + hdr;
    if (lane == 0) *op = (u8)((ty0 << 6) + (ty1 << 4) + (ty2 << 2));
    op++;
    if ((u32)lane < sz0) op[lane] = lds.ncbuf[0][lane];
    if ((u32)lane < sz1) op[sz0 + lane] = lds.ncbuf[1][lane];
    if ((u32)lane < sz2) op[sz0 + sz1 + lane] = lds.ncbuf[2][lane];
    if ((u32)lane + 64 < sz0) op[lane + 64] = lds.ncbuf[0][lane + 64];
    if ((u32)lane + 64 < sz1) op[sz0 + lane + 64] = lds.ncbuf[1][lane + 64];
    if ((u32)lane + 64 < sz2) op[sz0 + sz1 + lane + 64] = lds.ncbuf[2][lane + 64];
    op += sz0 + sz1 + sz2;
    u32 const lastCountSize = (ty2 == KSET_COMPRESSED) ? sz2 : (ty1 == KSET_COMPRESSED) ? sz1 : (ty0 == KSET_COMPRESSED) ? sz0 : 0;

    // tANS bitstream. Per 64-sequence chunk (walked last -> first): every lane stages one
    // sequence's codes; lanes 0..2 run the LL / OF / ML state chains; then every lane packs
    // its sequence's bits into an LDS buffer at a scanned bit offset; whole words go out.
    if (xflags & 16u) return 0;                       // timing experiment: tables only
    u8* const streamStart = op;
    u32* const cbuf = lds.u.seq.cbuf;
    for (int i = lane; i < 192; i += 64) cbuf[i] = 0;
    u32 state = 0; u32 bitpos = 0;
    for (u32 hi = nbSeq; hi > 0; ) {
        u32 const cnt = hi > 64 ? 64u : hi;
        bool const valid = (u32)lane < cnt;
        KSeq q; q.offBase = 1; q.litLength = 0; q.mlBase = 0; KSeqCodes c; c.ll = 0; c.of = 0; c.ml = 0;
        u32 dLL = 0, dOF = 0, dML = 0;
        bool const first = (hi == nbSeq);
        if (valid) {
            u32 const idx = hi - 1 - (u32)lane;            // lane order == stream order
            q = seqs[idx]; c = kx_seq_codes(q, idx, longType, longPos);
            lds.u.seq.stage[lane] = c.ll | (c.of << 8) | (c.ml << 16);
            dLL = lds.u.seq.dnb[0][c.ll]; dOF = lds.u.seq.dnb[1][c.of]; dML = lds.u.seq.dnb[2][c.ml];
            KSeqDelta d0, d1, d2;                                                                    // (fs: byte offsets into the state table)
            d0.nb = dLL; d0.fs = 2 * lds.u.seq.dfs[0][c.ll]; d1.nb = dOF; d1.fs = 2 * lds.u.seq.dfs[1][c.of]; d2.nb = dML; d2.fs = 2 * lds.u.seq.dfs[2][c.ml];
            lds.u.seq.pp[0][lane] = d0; lds.u.seq.pp[1][lane] = d1; lds.u.seq.pp[2][lane] = d2;
        }
        kx_sync();
        // The chains are what the kernel's time goes into once everything else is wave-parallel (three lanes, one LDS read
        // after the other, and the other waves of the SIMD doing the same): a step is kept to add, shift, shift-add and the
        // state table's load -- the state itself is stored, and the lane that owns the sequence derives the bit count and the
        // bits from it afterwards -- and a whole chunk runs without loop control.
        if (lane < 3) {
            const u8* const stb = (const u8*)ct.state;
            if (!first && cnt == 64u) {
                // (LDS instructions with three live lanes hold the pipe like full ones, and with every wave of the CU in its chains the
                // pipe is what they queue for: one 8-byte read for the next step's deltas, the states stored two at a time)
                KSeqDelta d = lds.u.seq.pp[lane][0];
                u32 even = 0;
#pragma unroll
                for (u32 s = 0; s < 64u; s++) {
                    KSeqDelta const d1 = lds.u.seq.pp[lane][s + 1];
                    if (s & 1u) *(u32*)&lds.u.seq.sbits[lane][s - 1] = even | (state << 16); else even = state;
                    u32 const nb = (state + d.nb) >> 16;
                    state = *(const u16*)(stb + d.fs + (int)((state >> nb) << 1));
                    d = d1;
                }
            } else {
                u32 s = 0;
                if (first) {
                    state = kfse_init_state(ct, (lds.u.seq.stage[0] >> (8u * (u32)lane)) & 0xFFu);
                    lds.u.seq.sbits[lane][0] = 0; s = 1;
                }
                KSeqDelta d = lds.u.seq.pp[lane][s];
                for (; s < cnt; s++) {
                    KSeqDelta const d1 = lds.u.seq.pp[lane][s + 1];
                    lds.u.seq.sbits[lane][s] = (u16)state;
                    u32 const nb = (state + d.nb) >> 16;
                    state = *(const u16*)(stb + d.fs + (int)((state >> nb) << 1));
                    d = d1;
                }
            }
        }
        kx_sync();
        u32 const rLL = valid ? lds.u.seq.sbits[0][lane] : 0u, rOF = valid ? lds.u.seq.sbits[1][lane] : 0u, rML = valid ? lds.u.seq.sbits[2][lane] : 0u;
        bool const coded = valid && !(first && lane == 0);              // the first sequence only sets the states
        u32 const nLL = coded ? (rLL + dLL) >> 16 : 0u, nOF = coded ? (rOF + dOF) >> 16 : 0u, nML = coded ? (rML + dML) >> 16 : 0u;
        u32 const sLL = rLL & ((1u << nLL) - 1u), sOF = rOF & ((1u << nOF) - 1u), sML = rML & ((1u << nML) - 1u);
        u32 const llb = valid ? kx_ll_bits(c.ll) : 0u, mlb = valid ? kx_ml_bits(c.ml) : 0u, ofb = valid ? c.of : 0u;
        u32 const mybits = nLL + nOF + nML + llb + mlb + ofb;
        u32 v = mybits;                                           // inclusive prefix sum over lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { u32 const t = kx_shfl(v, lane - o); if (lane >= o) v += t; }
        u32 const total = kx_bcast(v, 63);
        if ((u32)(streamStart - dst) + ((bitpos + total) >> 3) >= cap) return 0;            // larger than the block: raw block for certain
        u32 pos = (bitpos & 31u) + (v - mybits);
        if (valid) {
            // order inside a sequence: OF state, ML state, LL state, LL extra, ML extra, OF extra
            u32 const a = (sOF & 0xFFFu) | ((sML & 0xFFFu) << nOF) | ((sLL & 0xFFFu) << (nOF + nML));
            kx_cbuf_put(cbuf, pos, a, nOF + nML + nLL); pos += nOF + nML + nLL;
            kx_cbuf_put(cbuf, pos, q.litLength, llb); pos += llb;
            kx_cbuf_put(cbuf, pos, q.mlBase, mlb); pos += mlb;
            kx_cbuf_put(cbuf, pos, q.offBase, ofb);
        }
        kx_sync();
        u32 const nbits = (bitpos & 31u) + total; u32 const nfull = nbits >> 5;
        u8* const wbase = streamStart + 4u * (bitpos >> 5);
        for (u32 w = (u32)lane; w < nfull; w += 64) kx_st32(wbase + 4u * w, cbuf[w]);
        u32 const carry = cbuf[nfull];
        kx_sync();
        for (u32 w = (u32)lane; w <= nfull; w += 64) cbuf[w] = (w == 0) ? carry : 0u;
        kx_sync();
        bitpos += total;
        hi -= cnt;
    }
    // final states (ML, OF, LL) then the end mark
    {
        u32 const base = bitpos & 31u;
        if (lane == 2) kx_cbuf_put(cbuf, base, state, tl2);
        if (lane == 1) kx_cbuf_put(cbuf, base + tl2, state, tl1);
        if (lane == 0) { kx_cbuf_put(cbuf, base + tl2 + tl1, state, tl0); kx_cbuf_put(cbuf, base + tl2 + tl1 + tl0, 1u, 1u); }
        kx_sync();
        u32 const endbits = bitpos + tl2 + tl1 + tl0 + 1;
        u32 const streamSize = (endbits + 7) >> 3;
        u8* const wbase = streamStart + 4u * (bitpos >> 5);
        u32 const tailBytes = streamSize - 4u * (bitpos >> 5);
        if ((u32)lane < tailBytes) wbase[lane] = (u8)(cbuf[lane >> 2] >> (8 * (lane & 3)));
        kx_sync();
        if (lastCountSize && (lastCountSize + streamSize) < 4) return 0;
        return (u32)(streamStart + streamSize - dst);
    }
}

// ---- literals: gathered from the source ------------------------------------
// The match kernel stores no literals.  Sequence i's literals are
// src[P_i, P_i + ll_i) with P_i = sum over j < i of (ll_j + ml_j); 64 sequences
// per round, two shuffle scans give every lane its source and literal offsets.
KX_DEV void kx_gather_literals(u8* lits, const u8* src, u32 n, const KSeq* seqs, u32 nbSeq, u32 longType, u32 longPos, int lane)
{
    u32 sp = 0, lp = 0;
    for (u32 base = 0; base < nbSeq; base += 64u) {
        u32 const i = base + (u32)lane;
        u32 ll = 0, adv = 0;
        if (i < nbSeq) {
            KSeq const q = seqs[i];
            ll = q.litLength; adv = (u32)q.mlBase + 3u;
            if (i == longPos) { if (longType == 1) ll += 0x10000u; if (longType == 2) adv += 0x10000u; }
            adv += ll;
        }
        u32 sl = ll, sa = adv;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            u32 const tl = kx_shfl(sl, lane - o), ta = kx_shfl(sa, lane - o);
            if (lane >= o) { sl += tl; sa += ta; }
        }
        u32 const myL = lp + sl - ll, myS = sp + sa - adv;
        // long runs: the whole wave copies
        u64 big = kx_ballot(ll > 32u);
        while (big) {
            int const j = (int)kx_ctz64(big); big &= big - 1ull;
            kx_wave_copy(lits + kx_bcast(myL, j), src + kx_bcast(myS, j), kx_bcast(ll, j), lane);
        }
        if (ll > 0 && ll <= 32u) {
            u8* const d = lits + myL;
            u32 c = 0;
            for (; c + 8u <= ll; c += 8u) kx_st64(d + c, kx_ld64(src + myS + c));
            if (c < ll) {
                u64 w = kx_ld64_clamped(src, (int)(myS + c), (int)n);
                u32 const rem = ll - c;
                if (rem & 4u) { kx_st32(d + c, (u32)w); w >>= 32; c += 4u; }
                if (rem & 2u) { kx_st16(d + c, (u32)w); w >>= 16; c += 2u; }
                if (rem & 1u) d[c] = (u8)w;
            }
        }
        lp += kx_bcast(sl, 63); sp += kx_bcast(sa, 63);
    }
}

// ---- one slice -> one frame ---------------------------------------------
// PRIOR: the batch was parsed against a formatted dictionary (a.prior): its ID goes into the frame header, its tables stand in for a
// previous block's.  A template so that the kernel of every other batch is the code it was.
template <bool PRIOR = false>
KX_DEV void zstd_entropy_slice(const KEntropyArgs& a, KEntropyLds& lds, u32 slice, int lane)
{
    const u8* const src = a.src + a.in_off[slice];
    u32 const n = a.in_len[slice];
    u8* const dst = a.dst + a.out_off[slice];
    u32 const dictID = PRIOR ? a.prior->dictID : 0u;
    u32 const idCode = PRIOR ? (dictID > 0u) + (dictID >= 256u) + (dictID >= 65536u) : 0u, idBytes = idCode == 3u ? 4u : idCode;
    u32 const fh = kx_frame_header_size(n) + idBytes;
    if (lane == 0) {
        u32 const fcsCode = (n >= 256) + (n >= 65536 + 256);
        kx_st32(dst, 0xFD2FB528u);
        dst[4] = (u8)(idCode + (1u << 5) + (fcsCode << 6));
        u8* p = dst + 5;
        if (idCode == 1) p[0] = (u8)dictID; else if (idCode == 2) kx_st16(p, dictID); else if (idCode == 3) kx_st32(p, dictID);
        p += idBytes;
        if (fcsCode == 0) p[0] = (u8)n;
        else if (fcsCode == 1) kx_st16(p, n - 256);
        else kx_st32(p, n);
    }
    u8* const bh = dst + fh; u8* const body = bh + 3;
    if (n == 0) {
        if (lane == 0) { bh[0] = 1; bh[1] = 0; bh[2] = 0; a.out_len[slice] = fh + 3; }
        return;
    }
    u32 cSize = 0;
    if (n >= 7) {
        KSliceMeta const mm = a.meta[slice];
        const KSeq* const seqs = a.seqs + (size_t)slice * a.seq_cap;
        u8* const lits = a.lits + (size_t)slice * a.lit_cap;
        u32 const litSize = mm.litSize + mm.lastLL;
        if (a.flags & 8u) kx_gather_literals(lits, src, n, seqs, mm.nbSeq, mm.longType, mm.longPos, lane);
        // complete the literal buffer with the trailing literals
        kx_wave_copy(lits + mm.litSize, src + (n - mm.lastLL), mm.lastLL, lane);
        kx_sync();
        bool const suspect = (mm.nbSeq == 0) || (litSize / mm.nbSeq >= 20);
        KHufPrev hp; hp.ct = nullptr; hp.valid = false; hp.newCt = nullptr; hp.outcome = 0;
        if (PRIOR) { hp.ct = a.prior->ct; hp.valid = a.prior->hufMode != 0; hp.complete = a.prior->hufMode == 2; }
        u32 const litSec = (a.flags & 1u) ? 3u : kzstd_literals(lds, body, lits, litSize, suspect, a.scratch + (size_t)slice * a.scratch_words, lane, PRIOR ? &hp : nullptr, (a.flags & 64u) != 0);
        kx_sync();
        // (levels 5 .. 10 pass the level in bits 12 ..: which of the strategies greedy / lazy / lazy2 a slice was parsed with follows from its size,
        // zstd_lazy.h kx_lazy_params; the sequence coder takes the strategy's number in bits 8 .. 10)
        u32 xf = a.flags & 0xFFFu;
        if (a.flags >> 12) { u32 const lvl = a.flags >> 12; xf |= (n <= 16384u ? (lvl == 4u ? 3u : lvl == 5u ? 4u : 5u) : (lvl == 4u ? 2u : lvl == 5u ? 3u : lvl == 6u ? 4u : 5u)) << 8; }
        u32 const seqSec = (a.flags & 2u) ? 0u : kzstd_sequences(lds, body + litSec, seqs, mm.nbSeq, mm.longType, mm.longPos, lane, litSec < n ? n - litSec : 0u, xf, PRIOR ? a.prior : nullptr);
        if (seqSec != 0) {
            cSize = litSec + seqSec;
            if (cSize >= n - kx_min_gain(n)) cSize = 0;
        }
    }
    kx_sync();
    if (cSize == 0) {
        kx_wave_copy(body, src, n, lane);
        if (lane == 0) { u32 const h = 1u + (0u << 1) + (n << 3); bh[0] = (u8)h; bh[1] = (u8)(h >> 8); bh[2] = (u8)(h >> 16); a.out_len[slice] = fh + 3 + n; }
    } else if (lane == 0) {
        u32 const h = 1u + (2u << 1) + (cSize << 3); bh[0] = (u8)h; bh[1] = (u8)(h >> 8); bh[2] = (u8)(h >> 16);
        a.out_len[slice] = fh + 3 + cSize;
    }
}

// ---- frames of several blocks: one block of one slice ---------------------------------
// libzstd 1.5.7 ZSTD_compress_frameChunk / ZSTD_compressBlock_internal for level 3.  The match kernel (block
// mode) has parsed the block [ipos, ipos + blockSize); this writes the block (compressed / raw / RLE) behind the
// frame bytes so far, carries repcodes and the Huffman table forward when the block came out compressed, and
// chooses the size of the next block (the pre-splitter, zstd_preSplit.c "byChunks" at sampling rate 43).
struct KFrameArgs {
    const u8* src; const u64* in_off; const u32* in_len; u32 n_slices;
    const KSeq* seqs; u32 seq_cap; u8* lits; u32 lit_cap; const KSliceMeta* meta;
    u32* scratch; u32 scratch_words;
    u8* dst; const u64* out_off; u32* out_len;
    KFrameState* fstate; u32* hufct;         // per slice: state, two Huffman table slots of 256 words
    u32* remaining;                          // frames not finished yet (decremented here)
    u32 strategy;                            // 0: double-fast (level 3; level 2's 128 .. 256 KiB row); 1: fast (levels 1, 2): other pre-splitter, other encoding-type constant
    u32 level2;                              // 1: level 2's parameters (window 2^20 / 2^18 instead of 2^19 / level 3's)
    u32 fast_step0 = 0;                      // strategy 1 only: 0 = level 1 / 2; else a negative level: row 0 of libzstd's tables, a step of 1 - level, literals left raw
    u32 cls;                                 // which slices this launch takes (kx_in_class): a level-2 batch goes through both kernels
    u32 stream;                              // 0: one-shot frames as ZSTD_compress2 writes them into a bound-sized buffer (size known, the
                                             // caller's array compressed in place); 1: streaming frames (finish = false ... finish = true:
                                             // no content size, window 2^21, input taken in chunks of 128 KiB); 2: same, and the
                                             // closing call brought no data (an empty last block closes a frame that ends on a chunk boundary);
                                             // 3: one-shot frames as the reference's driver gets them (SliceTransform.kt:33-56: output slices of
                                             // max(8192, n / 10) bytes, below ZSTD_compressBound: size known, input staged in chunks of 128 KiB)
    u32 tail_direct;                         // streams: bytes the closing call brought onto an empty staging buffer with room for
                                             // their bound in its output slice (compressed in place as one chunk); else 0
    u32 out_chunk;                           // stream == 3: size of the driver's output slices; 0 = the reference's max(8192, n / 10)
    u32* status_word = nullptr;              // the context's status word: bit 1 (KMP_STATUS_KERNEL_GUARD) when a block's parser tripped its loop guard
};

// The block before fs.ipos is out: libzstd's staging buffer and window move on to the block that starts there
// (ZSTD_compressStream_generic's buffered path, ZSTD_window_update per chunk, ZSTD_window_enforceMaxDist per block;
// the tests hold a CPU restatement of the same rules, pinned on the binary library).  Returns what is left of the chunk: the
// input the pre-splitter may look at.  fs.ipos < n.
KX_DEV u32 kx_frame_window_step(KFrameState& fs, u32 n, u32 mode, u32 windowLog, u32 tailDirect, u32 outChunkArg)
{
    u32 const maxDist = 1u << windowLog;
    if (mode != 0 && fs.ipos == fs.chunkEnd) {
        // a chunk of 128 KiB is done; the next one goes behind it in the staging buffer (window + 128 KiB bytes, the window
        // being the content size when that is known and smaller) or, when it would not fit, to the buffer's start
        u32 const windowSize = (mode == 3 && n < maxDist) ? n : maxDist;
        u32 const inBuffSize = windowSize + (KX_BLOCK_MAX < windowSize ? KX_BLOCK_MAX : windowSize);
        fs.bufPos += KX_BLOCK_MAX;
        if (fs.bufPos + KX_BLOCK_MAX > inBuffSize) { fs.extBase = fs.ipos - fs.bufPos; fs.bufPos = 0; }
        bool tail = false;
        if (fs.bufPos == 0) {
            // empty staging buffer: if the current output slice has room for the bound of everything still to come,
            // libzstd compresses that rest where it lies (ZSTD_compressEnd), as one chunk
            if (mode == 3) {
                u32 const outChunk = outChunkArg ? outChunkArg : (n / 10u > 8192u ? n / 10u : 8192u);     // SliceTransform.kt:47-56
                u32 const room = outChunk - fs.opos % outChunk, r = n - fs.ipos;       // every earlier call filled its slice
                tail = room >= r + (r >> 8) + (r < KX_BLOCK_MAX ? (KX_BLOCK_MAX - r) >> 11 : 0u);
            } else tail = tailDirect != 0 && fs.ipos + tailDirect == n;
        }
        fs.chunkEnd = (!tail && fs.ipos + KX_BLOCK_MAX < n) ? fs.ipos + KX_BLOCK_MAX : n;
        if (tail) fs.wflags |= 2u;
        // ZSTD_window_update: a chunk that does not follow its predecessor in memory starts a new segment ...
        if (fs.bufPos == 0) {
            fs.lowLimit = fs.dictLimit; fs.dictLimit = fs.ipos + 2u;
            if (fs.dictLimit - fs.lowLimit < 8u) fs.lowLimit = fs.dictLimit;
            fs.wflags |= 1u;
        }
        // ... and the front of the older segment that the chunk overwrites in the buffer is given up
        if ((fs.wflags & 1u) && !tail) {
            u32 const chunkLen = fs.chunkEnd - fs.ipos;
            u32 const extLo = (fs.lowLimit - 2u) - fs.extBase, extHi = (fs.dictLimit - 2u) - fs.extBase;
            if (fs.bufPos + chunkLen > extLo && fs.bufPos < extHi) {
                u32 const high = fs.extBase + fs.bufPos + chunkLen + 2u;
                fs.lowLimit = high > fs.dictLimit ? fs.dictLimit : high;
            }
        }
    }
    // ZSTD_window_enforceMaxDist(window, block start, maxDist)
    u32 const startIdx = fs.ipos + 2u;
    if (startIdx > maxDist) {
        u32 const newLow = startIdx - maxDist;
        if (fs.lowLimit < newLow) fs.lowLimit = newLow;
        if (fs.dictLimit < fs.lowLimit) fs.dictLimit = fs.lowLimit;
    }
    return fs.chunkEnd - fs.ipos;
}

// ZSTD_splitBlock_byChunks(level 0) on the 128 KiB at p: where the byte statistics change, in steps of 8 KiB.
// Uses lds.hist (new chunk) and lds.ct (chunks so far).
KX_DEV u32 kx_split_block(KEntropyLds& lds, const u8* p, int lane)
{
    u32 nbPast = 8191u / 43u; int penalty = 3; u32 result = KX_BLOCK_MAX;
    for (int i = lane; i < 256; i += 64) lds.ct[i] = 0;
    kx_sync();
    for (u32 j = (u32)lane; j * 43u < 8191u; j += 64) kx_lds_inc(&lds.ct[p[j * 43u]]);
    kx_sync();
    for (u32 pos = 8192u; pos <= KX_BLOCK_MAX - 8192u; pos += 8192u) {
        for (int i = lane; i < 256; i += 64) lds.hist[i] = 0;
        kx_sync();
        for (u32 j = (u32)lane; j * 43u < 8191u; j += 64) kx_lds_inc(&lds.hist[p[pos + j * 43u]]);
        kx_sync();
        u32 const nbNew = 8191u / 43u;
        u32 dev = 0;
        for (int i = lane; i < 256; i += 64) {
            int const d = (int)(lds.ct[i] * nbNew) - (int)(lds.hist[i] * nbPast);
            dev += (u32)(d < 0 ? -d : d);
        }
        dev = kx_wave_sum(dev, lane);
        u32 const threshold = nbPast * nbNew * (u32)(14 + penalty) / 16u;
        if (dev >= threshold) { result = pos; break; }
        for (int i = lane; i < 256; i += 64) lds.ct[i] += lds.hist[i];
        nbPast += nbNew;
        if (penalty > 0) penalty--;
        kx_sync();
    }
    kx_sync();
    return result;
}

// ZSTD_splitBlock level 0 ("fromBorders", what strategy fast gets): byte histograms of the first, the last and the middle
// 512 bytes of the 128 KiB at p.  Uses lds.hist (first), lds.ct (last) and lds.u.seq.cbuf.. (middle: 256 words of the union).
KX_DEV u32 kx_split_block_borders(KEntropyLds& lds, const u8* p, int lane)
{
    u32* const hF = lds.hist; u32* const hL = lds.ct; u32* const hM = (u32*)lds.u.huf.node;      // the union is idle here
    for (int i = lane; i < 256; i += 64) { hF[i] = 0; hL[i] = 0; hM[i] = 0; }
    kx_sync();
    for (int i = lane; i < 512; i += 64) {
        kx_lds_inc(&hF[p[i]]); kx_lds_inc(&hL[p[KX_BLOCK_MAX - 512 + i]]); kx_lds_inc(&hM[p[KX_BLOCK_MAX / 2 - 256 + i]]);
    }
    kx_sync();
    u32 dFL = 0, dFM = 0, dLM = 0;
    for (int i = lane; i < 256; i += 64) {
        int const f = (int)hF[i], l = (int)hL[i], m = (int)hM[i];
        dFL += (u32)((f > l ? f - l : l - f) * 512); dFM += (u32)((f > m ? f - m : m - f) * 512); dLM += (u32)((l > m ? l - m : m - l) * 512);
    }
    dFL = kx_wave_sum(dFL, lane); dFM = kx_wave_sum(dFM, lane); dLM = kx_wave_sum(dLM, lane);
    kx_sync();
    if (!(dFL >= 512u * 512u * 14u / 16u)) return KX_BLOCK_MAX;
    u32 const diff = dFM > dLM ? dFM - dLM : dLM - dFM;
    if (diff < 512u * 512u / 3u) return 64u << 10;
    return (dFM > dLM) ? (32u << 10) : (96u << 10);
}

KX_DEV void zstd_frame_block(const KFrameArgs& a, KEntropyLds& lds, u32 slice, int lane)
{
    KFrameState fs = a.fstate[slice];
    if (fs.blockSize == 0) return;                       // frame finished in an earlier round
    const u8* const src = a.src + a.in_off[slice];
    u32 const n = a.in_len[slice];
    u8* const dst = a.dst + a.out_off[slice];
    u32* const hufct = a.hufct + (size_t)slice * 512u;
    bool const streaming = a.stream == 1 || a.stream == 2;       // size unknown when the frame starts
    bool const emptyEnd = a.stream == 2 && (n % KX_BLOCK_MAX) == 0;
    // the window a frame of known size is written with: level 3's 2^21 at most, or the "fast" level's own (2^19 / 2^20), past which the
    // header carries a window descriptor instead of the single-segment flag
    u32 const fastLevel = a.fast_step0 ? 0u : a.level2 ? 2u : 1u;
    u32 const wlogKnown = a.strategy ? kx_window_log_fast(fastLevel, n, false) : 21u;
    if (fs.ipos == 0 && streaming) {
        // streaming frame header: no content size, window descriptor for 2^21
        if (lane == 0) { kx_st32(dst, 0xFD2FB528u); dst[4] = 0; dst[5] = (u8)(((a.strategy ? (a.level2 ? 20 : 19) : 21) - 10) << 3); }
        fs.opos = 6;
    } else if (fs.ipos == 0) {
        // frame header: content size; single segment while the window covers the slice
        if (lane == 0) kx_write_frame_header(dst, n, wlogKnown);
        fs.opos = kx_frame_header_size(n, wlogKnown);
    }
    u32 const bs = fs.blockSize;
    const u8* const bsrc = src + fs.ipos;
    bool const lastBlock = fs.ipos + bs == n && !emptyEnd;
    u8* const bh = dst + fs.opos; u8* const body = bh + 3;
    u32 cSize = 0;
    KSliceMeta mm; mm.nbSeq = 0; mm.litSize = 0; mm.lastLL = bs; mm.longType = 0; mm.longPos = 0; mm.status = 0; mm.pad[0] = fs.rep[0]; mm.pad[1] = fs.rep[1];
    KHufPrev hp; hp.ct = hufct + 256u * fs.hufSel; hp.valid = fs.hufValid != 0; hp.newCt = hufct + 256u * (fs.hufSel ^ 1u); hp.outcome = 0;
    if (bs >= 7) {                                       // MIN_CBLOCK_SIZE + block header + 1 + 1
        mm = a.meta[slice];
        if (mm.status != 0 && lane == 0 && a.status_word) kx_atomic_or(a.status_word, 2u);          // (never expected; the frame cannot be trusted)
        const KSeq* const seqs = a.seqs + (size_t)slice * a.seq_cap;
        u8* const lits = a.lits + (size_t)slice * a.lit_cap;
        u32 const litSize = mm.litSize + mm.lastLL;
        if (a.strategy) kx_gather_literals(lits, bsrc, bs, seqs, mm.nbSeq, mm.longType, mm.longPos, lane);   // the level-1 parse stores no literals
        kx_wave_copy(lits + mm.litSize, bsrc + (bs - mm.lastLL), mm.lastLL, lane);
        kx_sync();
        bool const suspect = (mm.nbSeq == 0) || (litSize / mm.nbSeq >= 20);
        u32 const litSec = kzstd_literals(lds, body, lits, litSize, suspect, a.scratch + (size_t)slice * a.scratch_words, lane, &hp, a.fast_step0 != 0);
        kx_sync();
        u32 const seqSec = kzstd_sequences(lds, body + litSec, seqs, mm.nbSeq, mm.longType, mm.longPos, lane, litSec < bs ? bs - litSec : 0u, a.strategy ? 32u : 0u);
        if (seqSec != 0) {
            cSize = litSec + seqSec;
            if (cSize >= bs - kx_min_gain(bs)) cSize = 0;
        }
        // a block of one repeated byte becomes an RLE block, except the first block of a frame (ZSTD_compressBlock_internal tests the entropy
        // stage's result: cSize < rleMaxLength = 25, 0 when the block would go out raw -- a short tail of one byte the "fast" parser finds
        // nothing in is an RLE block too; round 4's fuzz found it)
        if (!fs.first && cSize < 25u) {
            u32 const b0 = bsrc[0]; bool diff = false;
            for (u32 i = (u32)lane; i < bs; i += 64) diff |= bsrc[i] != b0;
            if (!kx_any(diff)) cSize = 1;
        }
    }
    kx_sync();
    u32 outSize;
    if (cSize == 0) {
        kx_wave_copy(body, bsrc, bs, lane);
        if (lane == 0) { u32 const h = (u32)lastBlock + (0u << 1) + (bs << 3); bh[0] = (u8)h; bh[1] = (u8)(h >> 8); bh[2] = (u8)(h >> 16); }
        outSize = 3 + bs;
    } else if (cSize == 1) {
        if (lane == 0) { u32 const h = (u32)lastBlock + (1u << 1) + (bs << 3); bh[0] = (u8)h; bh[1] = (u8)(h >> 8); bh[2] = (u8)(h >> 16); body[0] = bsrc[0]; }
        outSize = 3 + 1;
    } else {
        if (lane == 0) { u32 const h = (u32)lastBlock + (2u << 1) + (cSize << 3); bh[0] = (u8)h; bh[1] = (u8)(h >> 8); bh[2] = (u8)(h >> 16); }
        outSize = 3 + cSize;
        // ZSTD_blockState_confirmRepcodesAndEntropyTables
        fs.rep[0] = mm.pad[0]; fs.rep[1] = mm.pad[1];
        if (hp.outcome == 2) { fs.hufSel ^= 1u; fs.hufValid = 1; }
    }
    fs.savings += (int)bs - (int)outSize;
    fs.ipos += bs; fs.opos += outSize; fs.first = 0;
    // ZSTD_optimalBlockSize for the next block (staged input is compressed in chunks of 128 KiB; the frame header counts
    // as produced from the second chunk on)
    u32 next = 0;
    if (a.stream && fs.ipos == KX_BLOCK_MAX) fs.savings -= streaming ? 6 : (int)kx_frame_header_size(n, wlogKnown);
    if (fs.ipos < n) {
        // the window of the level: beyond it libzstd's staging buffer wraps and the window slides (the "fast" levels too, since round 4)
        u32 const windowLog = a.strategy ? kx_window_log_fast(fastLevel, n, streaming) : (streaming ? 21u : a.level2 ? 18u : kx_params_l3(n).windowLog);
        u32 const remaining = kx_frame_window_step(fs, n, a.stream, windowLog, a.tail_direct, a.out_chunk);
        if (remaining < KX_BLOCK_MAX) next = remaining;
        else if (fs.savings < 3) next = KX_BLOCK_MAX;
        else next = a.strategy ? kx_split_block_borders(lds, src + fs.ipos, lane) : kx_split_block(lds, src + fs.ipos, lane);
    }
    fs.blockSize = next;
    if (lane == 0) {
        if (next == 0 && emptyEnd) { u8* const e = dst + fs.opos; e[0] = 1; e[1] = 0; e[2] = 0; fs.opos += 3; }
        a.fstate[slice] = fs;
        if (next == 0) { a.out_len[slice] = fs.opos; kx_atomic_add(a.remaining, 0xFFFFFFFFu); }
    }
}

KX_DEV void zstd_frame_body(const KFrameArgs& a)
{
    KX_SHARED KEntropyLds lds;
    int const lane = kx_lane();
    for (u32 it = kx_block(); it < a.n_slices; it += kx_nblocks()) {
        u32 const slice = kx_xcd_chunk(it, a.n_slices);
        zstd_frame_block(a, lds, slice, lane);
        kx_sync();
    }
}

// ---- a whole frame of several blocks by one wave ------------------------------------------------
// Every block's size depends on the bytes the blocks before it produced, so a slice is a chain
// split -> parse -> entropy-code -> split ...; slices are independent of each other.  One wave owns a slice
// and walks the chain; the parse is the block-mode match body run by the wave's first team.
// (kx_sync = wait for the wave's stores + barrier: what one step wrote to HBM is what the next one reads.)
struct KBigArgs { KMatchArgs m; KFrameArgs e; u32* counters; u32 spw; };   // counters: one work-queue head per workgroup
                                                                           // spw: slices per wave, 1 .. 64 / G

template <int G, bool FAST = false>      // FAST: the level-1 parse (a.e.strategy == 1)
KX_DEV void zstd_big_body(const KBigArgs& a)
{
    KX_SHARED KEntropyLds lds;
    int const lane = kx_lane();
    u32 const spw = a.spw;
    u32 const ngroups = (a.e.n_slices + spw - 1) / spw;
    for (u32 it = kx_block(); it < ngroups; it += kx_nblocks()) {
        u32 const grp = kx_xcd_chunk(it, ngroups);
        // this wave's slices: their blocks are parsed side by side (one team each), then coded one after the other
        u32 const base = grp * spw;
        u32 const cnt = (a.e.n_slices - base < spw) ? a.e.n_slices - base : spw;
        KMatchArgs m = a.m;
        m.in_off += base; m.in_len += base; m.n_slices = cnt;
        m.seqs += (size_t)base * m.seq_cap; m.lits += (size_t)base * m.lit_cap; m.meta += base; m.fstate += base;
        m.big_tables += (size_t)base * (FAST ? (size_t)KX_BIG_TBL_ENTRIES : (size_t)m.big_stride);
        m.counter = a.counters + kx_block();
        for (u32 guard = 0; guard < KX_MAX_BIG_SLICE / 64u; guard++) {         // (a block is at least 8 KiB unless it ends a chunk)
            bool open = false;
            for (u32 t = 0; t < cnt; t++) open |= a.e.fstate[base + t].blockSize != 0 && kx_in_class(a.e.cls, a.e.in_len[base + t]);
            if (!open) break;
            if (lane == 0) *m.counter = 0;
            kx_sync();
            // blocks behind a wrap of libzstd's staging buffer go through the extDict variant of the parse (each body skips the other's blocks)
            bool ext = false;
            for (u32 t = 0; t < cnt; t++) {
                KFrameState const& f = a.e.fstate[base + t];
                if (f.blockSize != 0 && f.lowLimit < f.dictLimit) ext = true;
            }
            if (FAST) {
                KFastArgs fa; fa.m = m; fa.level = a.e.fast_step0 ? 0u : a.e.level2 ? 2u : 1u; if (a.e.fast_step0) fa.step0 = a.e.fast_step0;
                zstd_match_fast_body<G, true>(fa);
                if (ext) {
                    kx_sync();
                    if (lane == 0) *m.counter = 0;
                    kx_sync();
                    zstd_match_fast_ext_body<G>(fa);
                }
            } else {
                zstd_match_body<G, true>(m);
                if (ext) {
                    kx_sync();
                    if (lane == 0) *m.counter = 0;
                    kx_sync();
                    zstd_match_ext_body<G>(m);
                }
            }
            kx_sync();
            for (u32 t = 0; t < cnt; t++) { if (kx_in_class(a.e.cls, a.e.in_len[base + t])) zstd_frame_block(a.e, lds, base + t, lane); kx_sync(); }
        }
    }
}

// ---- parse and entropy stage in one launch -------------------------------------------------
// The parse kernel's waves wait for their table loads three quarters of the time; a wave entropy-codes each slice the
// moment one of its teams has parsed it (zstd_match_body's DONE hook), in the issue slots the SIMD's other waves leave idle.
// (a real call: the entropy stage's registers are its own, the parse loop keeps its allocation and only what lives across the call
// is saved around it)
KX_DEV_NOINLINE void zstd_entropy_call(const KEntropyArgs* e, u32 slice)
{
    KX_SHARED KEntropyLds lds;
    zstd_entropy_slice(*e, lds, slice, kx_lane());
    kx_sync();
}
struct KFuseDone {
    static constexpr bool on = true;
    const KEntropyArgs* e;
    KX_MEMBER void operator()(u32 slice) const { zstd_entropy_call(e, slice); }
};
template <int G>
KX_DEV void zstd_l3_fused_body(const KMatchArgs& a, const KEntropyArgs& e)
{
    KFuseDone done; done.e = &e;
    zstd_match_body<G, false, KFuseDone>(a, done);
}

template <bool PRIOR = false>
KX_DEV void zstd_entropy_body(const KEntropyArgs& a)
{
    KX_SHARED KEntropyLds lds;
    int const lane = kx_lane();
    for (u32 it = kx_block(); it < a.n_slices; it += kx_nblocks()) {
        u32 const slice = kx_xcd_chunk(it, a.n_slices);
        zstd_entropy_slice<PRIOR>(a, lds, slice, lane);
        kx_sync();
    }
}
