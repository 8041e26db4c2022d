/* fuzzgen.c -- TEST TOOL (tools/r03_fuzz.py): seeded inputs built to stress a speculative LZ parser -- literal runs from small
 * alphabets, matches at very short distances (inside one search step), repeats of the last three offsets, long runs, periodic
 * data with sparse noise, and abrupt changes between these regimes.  No relation to any corpus; deterministic from the seed.
 *   gcc -O2 -shared -fPIC -o gpurun_out/libfuzzgen.so tools/fuzzgen.c */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
static uint64_t nx(uint64_t* s) { uint64_t z = (*s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static uint32_t below(uint64_t* s, uint32_t n) { return (uint32_t)((nx(s) >> 32) * (uint64_t)n >> 32); }
__attribute__((visibility("default")))
void fuzz_fill(uint8_t* dst, size_t n, uint64_t seed)
{
    uint64_t s = seed * 0xD1342543DE82EF95ull + 1; size_t p = 0; uint32_t rep[3] = { 1, 4, 8 };
    while (p < n) {
        uint32_t const regime = below(&s, 8); size_t const end = p + 64 + below(&s, 1u << (6 + below(&s, 9))); size_t const e = end < n ? end : n;
        uint32_t const alpha = 1u + below(&s, regime < 2 ? 4 : regime < 5 ? 32 : 256); uint8_t const base = (uint8_t)nx(&s);
        if (regime == 7) {                         /* periodic with sparse noise */
            uint32_t const per = 1 + below(&s, 1 + below(&s, 24)); size_t const st = p;
            for (; p < e; p++) dst[p] = (p - st < per) ? (uint8_t)(base + below(&s, alpha)) : dst[p - per];
            { uint32_t k = below(&s, 6); while (k--) { size_t const q = st + below(&s, (uint32_t)(e - st)); dst[q] = (uint8_t)nx(&s); } }
            continue;
        }
        while (p < e) {
            uint32_t const ll = below(&s, 4) ? below(&s, 1 + below(&s, regime == 6 ? 200 : 12)) : 0;
            for (uint32_t i = 0; i < ll && p < e; i++) dst[p++] = (uint8_t)(base + below(&s, alpha));
            if (p == 0 || p >= e) continue;
            {
                uint32_t const kind = below(&s, 10); uint32_t off;
                if (kind < 3) off = rep[below(&s, 3)];
                else if (kind < 6) off = 1 + below(&s, 16);                      /* inside one search step of a team */
                else if (kind < 8) off = 1 + below(&s, 1024);
                else off = 1 + below(&s, (uint32_t)(p < 131072 ? p : 131072));
                if (off > p) off = (uint32_t)p;
                if (off == 0) off = 1;
                uint32_t ml = 3 + below(&s, 1 + below(&s, below(&s, 8) ? 24 : 600));
                rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = off;
                for (; ml && p < e; ml--, p++) dst[p] = dst[p - off];
            }
        }
    }
}
