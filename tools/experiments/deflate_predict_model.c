/* deflate_predict_model.c -- would a PREDICTED parse find the positions zlib's lazy parse asks longest_match for?
 *
 * k_deflate_best walks the hash chain of EVERY position (128 steps at level 6); zlib's parse asks for about one position in
 * five.  The scheme DESIGN.md section 8.4 costs: a cheap pass gives every position a depth-limited match (chain limit D), a
 * speculative parse on those picks the positions worth a full walk, the exact parse falls back where the guess was wrong.
 * It pays only if the guess is right well above 90 % of the time.  This program measures that on BASELINE configs[4]'s
 * slices with the oracle's deflate_slow loop (oracle/deflate_l6_ref.c, its hooks DREF_TRACE_REQ / DREF_CHAIN): the exact
 * run records the positions asked for; the predicted run is the same loop with the chain limit D; hit rate = asked-for
 * positions the predicted run asked for too (also with their neighbours p +- 1 added to the predicted set).
 *   gcc -O2 -o tools/deflate_predict_model tools/deflate_predict_model.c && tools/deflate_predict_model [slices per class]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static unsigned g_chain_limit = 0;            /* 0 = the level's own */
static uint8_t* g_mark; static size_t g_req;
#define DREF_TRACE_REQ(pos) do { g_mark[(pos)] = 1; g_req++; } while (0)
#define DREF_CHAIN(c) (g_chain_limit ? (g_chain_limit < (c) ? g_chain_limit : (c)) : (c))
#include "../oracle/deflate_l6_ref.c"
#define CAPI
#include "../kompressor_amd/csrc/corpus.c"

int main(int argc, char** argv)
{
    int const per = argc > 1 ? atoi(argv[1]) : 24;
    size_t const S = 65536;
    uint8_t* buf = (uint8_t*)malloc(S); uint8_t* out = (uint8_t*)malloc(2 * S);
    uint8_t* exact = (uint8_t*)calloc(2 * S, 1); uint8_t* pred = (uint8_t*)calloc(2 * S, 1);
    static const char cls[] = "TXSBDIZR"; static const unsigned depths[] = { 2, 4, 8, 16, 32 };
    printf("raw DEFLATE level 6, %d slices of 64 KiB per class; per class: positions asked for by zlib's parse per slice, then per depth D: hit %% (hit %% with p+-1) | positions the predicted parse asks for, as %% of the exact count\n", per);
    double tot_req = 0, tot_hit[5] = { 0 }, tot_hit1[5] = { 0 }, tot_pred[5] = { 0 };
    for (int c = 0; c < 8; c++) {
        double req = 0, hit[5] = { 0 }, hit1[5] = { 0 }, npred[5] = { 0 };
        for (int i = 0; i < per; i++) {
            kmp_corpus_fill(buf, 400000 + (uint64_t)c * 1000 + i, 1, S, cls[c]);
            memset(exact, 0, 2 * S); g_mark = exact; g_req = 0; g_chain_limit = 0;
            dref_deflate_raw_level(out, 2 * S, buf, S, 6);
            size_t const nreq = g_req; req += nreq;
            for (int d = 0; d < 5; d++) {
                memset(pred, 0, 2 * S); g_mark = pred; g_req = 0; g_chain_limit = depths[d];
                dref_deflate_raw_level(out, 2 * S, buf, S, 6);
                npred[d] += g_req;
                size_t h = 0, h1 = 0;
                for (size_t p = 1; p + 1 < 2 * S; p++) if (exact[p]) { if (pred[p]) h++; if (pred[p] || pred[p - 1] || pred[p + 1]) h1++; }
                hit[d] += h; hit1[d] += h1;
            }
        }
        printf("class %c: %7.0f asked |", cls[c], req / per);
        for (int d = 0; d < 5; d++) printf("  D=%-2u %5.1f%% (%5.1f%%) | %5.1f%%", depths[d], 100.0 * hit[d] / req, 100.0 * hit1[d] / req, 100.0 * npred[d] / req);
        printf("\n");
        tot_req += req; for (int d = 0; d < 5; d++) { tot_hit[d] += hit[d]; tot_hit1[d] += hit1[d]; tot_pred[d] += npred[d]; }
    }
    printf("all     : %7.0f asked |", tot_req / (8.0 * per));
    for (int d = 0; d < 5; d++) printf("  D=%-2u %5.1f%% (%5.1f%%) | %5.1f%%", depths[d], 100.0 * tot_hit[d] / tot_req, 100.0 * tot_hit1[d] / tot_req, 100.0 * tot_pred[d] / tot_req);
    printf("\n");
    return 0;
}
