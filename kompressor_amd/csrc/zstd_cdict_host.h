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

