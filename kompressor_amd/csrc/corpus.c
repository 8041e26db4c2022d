/*
 * Seeded synthetic slice corpus: the reference ships
 * only a seeded random-byte source (kompressor-test/src/commonMain/kotlin/
 * com/ensody/kompressor/test/RandomSource.kt:13-37), which at level 3 only
 * exercises the raw-block fallback.  This generator plays the same role
 * (deterministic input from a seed) but produces compressible classes:
 *   T text, X json-ish records, S source-like lines, B binary structs,
 *   D fixed-width rows, I 16-bit smooth samples, Z sparse, R uniform random.
 * Slice i of a batch uses class mix16[i % 16] and PRNG seed 0x4B6F6D70 ^ i.
 * Integer arithmetic only, so every machine regenerates identical bytes.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#define CAPI __attribute__((visibility("default")))

typedef struct { uint64_t s[4]; } rng_t;
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static uint64_t splitmix(uint64_t* x)
{
    uint64_t z = (*x += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static void rng_seed(rng_t* r, uint64_t seed) { int i; for (i = 0; i < 4; i++) r->s[i] = splitmix(&seed); }
static inline uint64_t rng_next(rng_t* r)
{
    uint64_t* s = r->s; uint64_t const result = rotl(s[1] * 5, 7) * 9; uint64_t const t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return result;
}
static inline uint32_t rng_below(rng_t* r, uint32_t n) { return (uint32_t)(((rng_next(r) >> 32) * (uint64_t)n) >> 32); }

/* ---- shared vocabulary (fixed seed, built once per call; cheap) ---- */
#define NWORDS 4096
typedef struct { char w[NWORDS][10]; uint8_t len[NWORDS]; uint32_t cdf[NWORDS]; } vocab_t;
static void vocab_build(vocab_t* v)
{
    static const char letters[] = "eeeeeeeeeeeetttttttttaaaaaaaaooooooooiiiiiiinnnnnnnsssssshhhhhhrrrrrrddddllllcccuuummmwwffggyyppbbvkjxqz";
    rng_t r; int i, j; uint64_t acc = 0; uint64_t tot = 0;
    rng_seed(&r, 0x766F636162ULL);
    for (i = 0; i < NWORDS; i++) {
        int len = 2 + (int)rng_below(&r, 8);
        if (i < 64) len = 1 + (int)rng_below(&r, 4);   /* frequent words are short */
        v->len[i] = (uint8_t)len;
        for (j = 0; j < len; j++) v->w[i][j] = letters[rng_below(&r, sizeof(letters) - 1)];
    }
    for (i = 0; i < NWORDS; i++) tot += (1u << 24) / (uint32_t)(i + 1);
    for (i = 0; i < NWORDS; i++) { acc += (1u << 24) / (uint32_t)(i + 1); v->cdf[i] = (uint32_t)((acc << 32) / (tot + 1)); }
    v->cdf[NWORDS - 1] = 0xFFFFFFFFu;
}
static int vocab_pick(const vocab_t* v, rng_t* r)
{
    uint32_t const x = (uint32_t)(rng_next(r) >> 32); int lo = 0, hi = NWORDS - 1;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (v->cdf[mid] < x) lo = mid + 1; else hi = mid; }
    return lo;
}

typedef struct { uint8_t* p; size_t n, cap; } out_t;
static inline void put(out_t* o, const void* s, size_t k) { if (o->n + k > o->cap) k = o->cap - o->n; memcpy(o->p + o->n, s, k); o->n += k; }
static inline void putc_(out_t* o, int c) { if (o->n < o->cap) o->p[o->n++] = (uint8_t)c; }
static inline void puts_(out_t* o, const char* s) { put(o, s, strlen(s)); }
static void putu(out_t* o, uint64_t v) { char b[24]; int n = 0; do { b[n++] = (char)('0' + v % 10); v /= 10; } while (v); while (n) putc_(o, b[--n]); }
static void putword(out_t* o, const vocab_t* v, int w, int cap) { if (cap && v->w[w][0] >= 'a') { putc_(o, v->w[w][0] - 32); put(o, v->w[w] + 1, v->len[w] - 1u); } else put(o, v->w[w], v->len[w]); }

static void gen_text(out_t* o, rng_t* r, const vocab_t* v)
{
    int cap = 1;
    while (o->n < o->cap) {
        uint32_t const x = rng_below(r, 100);
        putword(o, v, vocab_pick(v, r), cap); cap = 0;
        if (x < 80) putc_(o, ' ');
        else if (x < 88) puts_(o, ", ");
        else if (x < 95) { puts_(o, ". "); cap = 1; }
        else if (x < 98) { puts_(o, ".\n"); cap = 1; }
        else { puts_(o, ".\n\n"); cap = 1; }
    }
}
static void gen_json(out_t* o, rng_t* r, const vocab_t* v)
{
    static const char* status[] = { "active", "pending", "closed", "error" };
    uint64_t id = 100000 + rng_below(r, 900000);
    while (o->n < o->cap) {
        int i, nt;
        puts_(o, "{\"id\": "); putu(o, id); id += 1 + rng_below(r, 3);
        puts_(o, ", \"name\": \""); putword(o, v, vocab_pick(v, r), 1); putc_(o, ' '); putword(o, v, vocab_pick(v, r), 1);
        puts_(o, "\", \"value\": "); putu(o, rng_below(r, 10000)); putc_(o, '.'); putu(o, 10 + rng_below(r, 90));
        puts_(o, ", \"status\": \""); puts_(o, status[rng_below(r, 4)]);
        puts_(o, "\", \"tags\": ["); nt = (int)rng_below(r, 4);
        for (i = 0; i < nt; i++) { if (i) puts_(o, ", "); putc_(o, '"'); putword(o, v, (int)rng_below(r, 48), 0); putc_(o, '"'); }
        puts_(o, "], \"enabled\": "); puts_(o, rng_below(r, 2) ? "true" : "false"); puts_(o, "}\n");
    }
}
static void gen_source(out_t* o, rng_t* r, const vocab_t* v)
{
    int ids[64]; int i, depth = 0;
    for (i = 0; i < 64; i++) ids[i] = 64 + (int)rng_below(r, NWORDS - 64);
#define ID() putword(o, v, ids[rng_below(r, 64)], 0)
    while (o->n < o->cap) {
        uint32_t const x = rng_below(r, 100);
        for (i = 0; i < depth; i++) puts_(o, "    ");
        if (x < 30) { ID(); puts_(o, " = "); ID(); puts_(o, (rng_below(r, 2) ? " + " : " * ")); putu(o, rng_below(r, 256)); puts_(o, ";\n"); }
        else if (x < 42 && depth < 6) { puts_(o, "if ("); ID(); puts_(o, " < "); ID(); puts_(o, ") {\n"); depth++; }
        else if (x < 50 && depth < 6) { puts_(o, "for (int i = 0; i < "); ID(); puts_(o, "; i++) {\n"); depth++; }
        else if (x < 66 && depth > 0) { o->n -= (o->n >= 4 ? 4 : 0); puts_(o, "}\n"); depth--; }
        else if (x < 76) { ID(); putc_(o, '('); ID(); puts_(o, ", "); ID(); puts_(o, ");\n"); }
        else if (x < 84) { puts_(o, "return "); ID(); puts_(o, ";\n"); }
        else if (x < 92) { puts_(o, "// "); for (i = 0; i < 5; i++) { putword(o, v, vocab_pick(v, r), 0); putc_(o, ' '); } putc_(o, '\n'); }
        else if (depth == 0) { puts_(o, "static int "); ID(); puts_(o, "(int "); ID(); puts_(o, ") {\n"); depth++; }
        else { puts_(o, "int "); ID(); puts_(o, " = 0;\n"); }
    }
#undef ID
}
static void gen_binary(out_t* o, rng_t* r, const vocab_t* v)
{
    static const uint8_t ops[48] = { 0x48,0x89,0x8b,0xe8,0xff,0x0f,0x85,0x84,0x74,0x75,0xc3,0x55,0x5d,0x41,0x83,0xc0,
        0x00,0x01,0x24,0x10,0x20,0x08,0x4c,0x8d,0x05,0x3d,0xeb,0xe9,0x31,0x39,0x66,0x90,
        0xc7,0x45,0xf8,0xfc,0x40,0x44,0x49,0x50,0x58,0x5b,0x5c,0x5e,0x5f,0x80,0xb8,0xba };
    uint8_t idiom[512][8]; uint8_t ilen[512]; int i, j;
    for (i = 0; i < 512; i++) {   /* per-slice pool of instruction-like idioms */
        ilen[i] = (uint8_t)(2 + rng_below(r, 6));
        for (j = 0; j < ilen[i]; j++) idiom[i][j] = (rng_below(r, 100) < 75) ? ops[rng_below(r, 48)] : (uint8_t)rng_below(r, 256);
    }
    while (o->n < o->cap) {
        uint32_t const kind = rng_below(r, 100);
        if (kind < 55) {            /* code: idioms (skewed pick) + immediates */
            int n = 64 + (int)rng_below(r, 448);
            for (i = 0; i < n; i++) {
                uint32_t a = rng_below(r, 512), b = rng_below(r, 512); uint32_t const k = a < b ? a : b;   /* triangular skew */
                uint32_t const k2 = (rng_below(r, 4) == 0) ? k : (k >> 3);
                put(o, idiom[k2], ilen[k2]);
                if (rng_below(r, 100) < 30) { uint32_t imm = rng_below(r, 4096) * (rng_below(r, 4) ? 8u : 0x1001u); put(o, &imm, (rng_below(r, 2) ? 4 : 1)); }
            }
        } else if (kind < 80) {     /* tables: little-endian u32 random walks */
            uint32_t x = (uint32_t)rng_next(r) & 0x00FFFFFFu; int n = 32 + (int)rng_below(r, 224);
            uint32_t const stride = 1 + rng_below(r, 64);
            for (i = 0; i < n; i++) { x += (rng_below(r, 8) == 0) ? rng_below(r, 1024) : stride; put(o, &x, 4); }
        } else if (kind < 92) {     /* string table */
            int n = 8 + (int)rng_below(r, 56);
            for (i = 0; i < n; i++) { putword(o, v, vocab_pick(v, r), 0); if (rng_below(r, 3) == 0) { putc_(o, '_'); putword(o, v, vocab_pick(v, r), 0); } putc_(o, 0); }
        } else {                    /* padding / raw data */
            int n = 16 + (int)rng_below(r, 240);
            if (rng_below(r, 2)) for (i = 0; i < n; i++) putc_(o, 0); else for (i = 0; i < n; i++) putc_(o, (int)rng_below(r, 256));
        }
    }
}
static void gen_rows(out_t* o, rng_t* r, const vocab_t* v)
{
    static const char status[8][9] = { "NEW     ", "OPEN    ", "PENDING ", "SHIPPED ", "CLOSED  ", "RETURNED", "ERROR   ", "HOLD    " };
    uint32_t id = rng_below(r, 1u << 24); uint64_t ts = 1700000000000ULL + rng_below(r, 1u << 30);
    while (o->n < o->cap) {
        uint8_t row[64]; uint16_t cat = (uint16_t)rng_below(r, 16); uint32_t val = rng_below(r, 100000); int w = 64 + (int)rng_below(r, 256);
        memset(row, 0, sizeof(row));
        memcpy(row, &id, 4); id++;
        memcpy(row + 4, &cat, 2);
        memcpy(row + 8, status[rng_below(r, 8)], 8);
        ts += rng_below(r, 5000); memcpy(row + 16, &ts, 8);
        memcpy(row + 24, &val, 4);
        memset(row + 32, ' ', 16); memcpy(row + 32, v->w[w], v->len[w]);
        put(o, row, 64);
    }
}
static void gen_image(out_t* o, rng_t* r)
{
    int32_t v = 2000 + (int32_t)rng_below(r, 4000);
    while (o->n < o->cap) {
        uint64_t const x = rng_next(r); uint16_t s;
        int32_t const g = (int32_t)((x & 7) + ((x >> 3) & 7) + ((x >> 6) & 7) + ((x >> 9) & 7)) - 14;   /* ~N(0, 4.6) */
        v += g / 2; if (v < 0) v = 0; if (v > 65535) v = 65535;
        s = (uint16_t)v; put(o, &s, 2);
    }
}
static void gen_sparse(out_t* o, rng_t* r)
{
    memset(o->p, 0, o->cap);
    { size_t i; for (i = 0; i < o->cap; i++) if (rng_below(r, 50) == 0) o->p[i] = (uint8_t)rng_below(r, 256); }
    o->n = o->cap;
}
static void gen_random(out_t* o, rng_t* r)
{
    while (o->n + 8 <= o->cap) { uint64_t const x = rng_next(r); memcpy(o->p + o->n, &x, 8); o->n += 8; }
    while (o->n < o->cap) putc_(o, (int)rng_below(r, 256));
}

static const char mix16[17] = "TXSBTDTBIXTSZBTR";
static const char mixTB[3] = "TB";

/* cls: 0 => config[1] mix (mix16[i%16]); 1 => config[3] mix (T/B alternating);
 * otherwise an explicit class letter. */
CAPI int kmp_corpus_class(uint64_t index, int cls)
{
    if (cls == 0) return mix16[index % 16];
    if (cls == 1) return mixTB[index % 2];
    return cls;
}

CAPI void kmp_corpus_fill(uint8_t* dst, uint64_t first_index, uint64_t count, size_t slice_size, int cls)
{
    static vocab_t vocab; static int vocab_ready = 0;
    vocab_t local; const vocab_t* v;
    uint64_t k;
    /* benign race: every thread would build identical contents */
    if (!vocab_ready) { vocab_build(&local); memcpy(&vocab, &local, sizeof(vocab)); __sync_synchronize(); vocab_ready = 1; }
    v = &vocab;
    for (k = 0; k < count; k++) {
        uint64_t const i = first_index + k; rng_t r; out_t o;
        o.p = dst + k * slice_size; o.n = 0; o.cap = slice_size;
        rng_seed(&r, 0x4B6F6D70ULL ^ i);
        switch (kmp_corpus_class(i, cls)) {
        case 'T': gen_text(&o, &r, v); break;
        case 'X': gen_json(&o, &r, v); break;
        case 'S': gen_source(&o, &r, v); break;
        case 'B': gen_binary(&o, &r, v); break;
        case 'D': gen_rows(&o, &r, v); break;
        case 'I': gen_image(&o, &r); break;
        case 'Z': gen_sparse(&o, &r); break;
        default:  gen_random(&o, &r); break;
        }
    }
}
