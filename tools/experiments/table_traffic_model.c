/* table_traffic_model.c -- how much of k_zstd_match's hash-table traffic could ANY on-chip scheme keep away from HBM?
 *
 * The level-3 parser's run time is the number of 64-byte table lines it touches cold (every
 * insert dirties a line = a fill + a write-back, every other probe is a read).  This tool replays the exact table
 * traffic of the parse (the oracle's loop with its event trace switched on) for slices of BASELINE configs[1] and
 * prices three families of schemes, each as the number of line transfers that would still reach HBM:
 *   log(C)   a write-combining log of the last C inserts of a slice (LDS), probed before the table, flushed in line
 *            order when full: an insert overwritten while still in the log never leaves, inserts of one line leave as
 *            one write (VERDICT r1 item 2 / DESIGN section 8.1c);
 *   lru(K)   K table lines of a slice cached on chip (LDS or registers), write-back, least recently used out;
 *   ideal    every line transferred once per slice and written back once if dirty (tables entirely on chip).
 * Output: per-slice averages by class and over the mix.
 *   gcc -O2 -o tools/table_traffic_model tools/table_traffic_model.c && ./tools/table_traffic_model [slices]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define KREF_TRACE
static void kref_trace_event(int table, uint32_t bucket, int kind);
#include "../oracle/zstd_l3_ref.c"
#define CAPI
#include "../kompressor_amd/csrc/corpus.c"

/* one event = line id (table line of 16 four-byte entries; short table behind the long one), kind */
typedef struct { uint32_t line; uint32_t bucketkey; uint8_t kind; } ev_t;
static ev_t* g_ev; static size_t g_nev, g_cap;
static void kref_trace_event(int table, uint32_t bucket, int kind)
{
    if (g_nev == g_cap) { g_cap = g_cap ? 2 * g_cap : 1 << 18; g_ev = (ev_t*)realloc(g_ev, g_cap * sizeof(ev_t)); }
    uint32_t const key = (uint32_t)table << 20 | bucket;
    g_ev[g_nev].line = key >> 4; g_ev[g_nev].bucketkey = key; g_ev[g_nev].kind = (uint8_t)kind; g_nev++;
}

typedef struct { double probes, ins_search, ins_other, base_reads, base_writes; double log_reads[4], log_writes[4]; double lru_reads[6], lru_writes[6]; double ideal_reads, ideal_writes; double n; } acc_t;
static const int LOGC[4] = { 64, 256, 1024, 4096 };
static const int LRUK[6] = { 8, 32, 128, 512, 2048, 6144 };

/* today's kernel: every probe reads its line unless the same line was touched by the previous event pair (the probe
 * of a searched position and its own insert share a line: one fill, one write-back); every insert dirties its line */
static void price_base(acc_t* a)
{
    double reads = 0, writes = 0; size_t i;
    for (i = 0; i < g_nev; i++) {
        if (g_ev[i].kind == 0) reads++;                       /* probe: cold line */
        else if (g_ev[i].kind == 1) writes++;                 /* own insert: the probe's line, already fetched */
        else { reads++; writes++; }                           /* blind insert: fill + write-back */
    }
    a->base_reads += reads; a->base_writes += writes;
}

static int cmp_u32(const void* x, const void* y) { uint32_t a = *(const uint32_t*)x, b = *(const uint32_t*)y; return a < b ? -1 : a > b; }

/* log of C pending inserts: a probe whose bucket is in the log is served there; flush = sort by line, one
 * read-modify-write per distinct line (the line has to be merged with its other 15 entries) */
static void price_log(acc_t* a, int which)
{
    int const C = LOGC[which];
    uint32_t* keys = (uint32_t*)malloc((size_t)C * 4); int n = 0, j; size_t i;
    uint32_t* lines = (uint32_t*)malloc((size_t)C * 4);
    double reads = 0, writes = 0;
    for (i = 0; i <= g_nev; i++) {
        int const flush = (i == g_nev) || (g_ev[i].kind != 0 && n == C);
        if (flush && n) {
            int d = 0;
            for (j = 0; j < n; j++) lines[j] = keys[j] >> 4;
            qsort(lines, (size_t)n, 4, cmp_u32);
            for (j = 0; j < n; j++) if (j == 0 || lines[j] != lines[j - 1]) d++;
            reads += d; writes += d; n = 0;
        }
        if (i == g_nev) break;
        if (g_ev[i].kind == 0) {
            int hit = 0;
            for (j = 0; j < n; j++) if (keys[j] == g_ev[i].bucketkey) { hit = 1; break; }
            if (!hit) reads++;
        } else {
            int found = 0;
            for (j = 0; j < n; j++) if (keys[j] == g_ev[i].bucketkey) { found = 1; break; }      /* overwritten in the log */
            if (!found) keys[n++] = g_ev[i].bucketkey;
        }
    }
    a->log_reads[which] += reads; a->log_writes[which] += writes;
    free(keys); free(lines);
}

/* K lines cached per slice, LRU, write-back */
static void price_lru(acc_t* a, int which)
{
    int const K = LRUK[which];
    uint32_t* line = (uint32_t*)malloc((size_t)K * 4); uint8_t* dirty = (uint8_t*)calloc((size_t)K, 1);
    uint64_t* stamp = (uint64_t*)calloc((size_t)K, 8); int used = 0, j; size_t i; uint64_t t = 1;
    /* index: line -> slot (direct map over all 6144 lines of a slice's two tables, plus the high table bit) */
    static int slot_of[1 << 18]; memset(slot_of, 0xFF, sizeof(slot_of));
    double reads = 0, writes = 0;
    for (i = 0; i < g_nev; i++, t++) {
        uint32_t const L = g_ev[i].line & ((1u << 18) - 1);
        int s = slot_of[L];
        if (s < 0) {
            reads++;
            if (used < K) s = used++;
            else {
                uint64_t best = ~0ull; s = 0;
                for (j = 0; j < K; j++) if (stamp[j] < best) { best = stamp[j]; s = j; }
                if (dirty[s]) writes++;
                slot_of[line[s]] = -1;
            }
            line[s] = L; dirty[s] = 0; slot_of[L] = s;
        }
        stamp[s] = t;
        if (g_ev[i].kind != 0) dirty[s] = 1;
    }
    for (j = 0; j < used; j++) if (dirty[j]) writes += 0;      /* tables die with the slice: nothing to write back */
    a->lru_reads[which] += reads; a->lru_writes[which] += writes;
    free(line); free(dirty); free(stamp);
}

int main(int argc, char** argv)
{
    int const nslices = argc > 1 ? atoi(argv[1]) : 256;
    size_t const S = 65536;
    u8* src = (u8*)malloc(S); u8* dst = (u8*)malloc(S + 4096);
    static acc_t by_class[256], all;
    static const double RD_PS = 18.5, RMW_PS = 50.0;      /* profiles/r01_random_access.txt: 54 G reads/s, 20 G fill+write-back pairs/s */
    int i, k;
    for (i = 0; i < nslices; i++) {
        int const cls = kmp_corpus_class((uint64_t)i, 0);
        kmp_corpus_fill(src, (uint64_t)i, 1, S, 0);
        g_nev = 0;
        kref_zstd_l3_compress(dst, S + 4096, src, S);
        acc_t* dstacc[2] = { &by_class[cls], &all };
        for (k = 0; k < 2; k++) {
            acc_t* a = dstacc[k]; size_t e; int w;
            a->n++;
            for (e = 0; e < g_nev; e++) { if (g_ev[e].kind == 0) a->probes++; else if (g_ev[e].kind == 1) a->ins_search++; else a->ins_other++; }
            price_base(a);
            for (w = 0; w < 4; w++) price_log(a, w);
            for (w = 0; w < 6; w++) price_lru(a, w);
        }
    }
    printf("# table traffic of the level-3 parse, per 64 KiB slice, %d slices of BASELINE configs[1]'s mix (exact replay of the oracle's loop;\n", nslices);
    printf("# the GPU kernel adds its speculative probes, about a quarter more).  cost = reads x %.1f ps + dirty lines x %.1f ps\n", RD_PS, RMW_PS);
    printf("# (a dirty line = fill + write-back; rates of profiles/r01_random_access.txt, profiles/r02_random_store.txt)\n");
    printf("%-6s %8s %8s %8s | %-22s", "class", "probes", "ins own", "ins oth", "today: rd  dirty   us");
    for (k = 0; k < 4; k++) printf(" | log %-5d rd  dirty   us", LOGC[k]);
    printf("\n");
    for (i = 0; i <= 256; i++) {
        acc_t* a = i < 256 ? &by_class[i] : &all; double n = a->n; if (n == 0) continue;
        double base_us = ((a->base_reads - a->base_writes) * RD_PS + a->base_writes * RMW_PS) / n * 1e-6;
        printf("%-6s %8.0f %8.0f %8.0f | %9.0f %6.0f %5.2f", i < 256 ? (char[]){ (char)i, 0 } : "mix", a->probes / n, a->ins_search / n, a->ins_other / n,
               a->base_reads / n, a->base_writes / n, base_us);
        for (k = 0; k < 4; k++) {
            double us = ((a->log_reads[k] - a->log_writes[k]) * RD_PS + a->log_writes[k] * RMW_PS) / n * 1e-6;
            printf(" | %12.0f %6.0f %5.2f", a->log_reads[k] / n, a->log_writes[k] / n, us);
        }
        printf("\n");
    }
    printf("\n%-6s", "class");
    for (k = 0; k < 6; k++) printf(" | lru %-4d lines (%4d KiB) rd  dirty   us", LRUK[k], LRUK[k] * 64 / 1024);
    printf("\n");
    for (i = 0; i <= 256; i++) {
        acc_t* a = i < 256 ? &by_class[i] : &all; double n = a->n; if (n == 0) continue;
        printf("%-6s", i < 256 ? (char[]){ (char)i, 0 } : "mix");
        for (k = 0; k < 6; k++) {
            double us = ((a->lru_reads[k] - a->lru_writes[k] > 0 ? a->lru_reads[k] - a->lru_writes[k] : 0) * RD_PS + a->lru_writes[k] * RMW_PS) / n * 1e-6;
            printf(" | %24.0f %6.0f %5.2f", a->lru_reads[k] / n, a->lru_writes[k] / n, us);
        }
        printf("\n");
    }
    return 0;
}
