/*
 * oracle/deflate_l6_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of what the reference's
 * ZlibCompressor(ZlibFormat.Raw, compressionLevel = 6, windowBits = 15, memLevel = 8)
 * computes for a one-shot slice: deflateInit2(6, Z_DEFLATED, -15, 8, 0) then
 * deflate(Z_FINISH) with all input present
 * (kompressor-zlib--nativelib/src/jvmCommonMain/jni/Wrapper.cpp:20,73, driven by
 *  .../zlib/ZlibCompressor.jvm.kt:27-45 under SliceTransform.kt:33-45).
 *
 * The arithmetic lives in a third-party dependency that is ABSENT from
 * /root/reference: com.ensody.nativebuilds:zlib-libz:1.3.1.8 == upstream zlib
 * 1.3.1 (gradle/libs.versions.toml:8,43-44).  This file restates its published
 * algorithm (RFC 1951 + zlib's level-6 "deflate_slow": 32 KiB sliding window in
 * a 64 KiB buffer, 15-bit rolling hash with chains, lazy matching with
 * good/lazy/nice/chain = 8/16/128/128, block flush every 16383 symbols,
 * stored / fixed / dynamic block choice).
 *
 * Parity pin: byte equality with the zlib on this machine (1.2.11, through
 * Python's zlib module: tests/golden/make_golden_deflate.py).  zlib 1.3.1
 * itself is not available here; its deflate_slow / trees.c are believed to
 * produce the same streams -- recorded as a residual risk in DESIGN.md.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this. The product path never links or calls it.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>

#define DREF_API __attribute__((visibility("default")))
typedef uint8_t u8; typedef uint16_t u16; typedef uint32_t u32; typedef uint64_t u64;

/* deflateInit2's windowBits (9 .. 15) and memLevel (1 .. 9) are fields of the state: w_size = 1 << windowBits, hash_bits = memLevel + 7,
 * hash_shift = (hash_bits + MIN_MATCH - 1) / MIN_MATCH, lit_bufsize = 1 << (memLevel + 6)  (zlib deflate.c deflateInit2_).  The arrays
 * have the sizes of the largest setting. */
#define W_BITS_MAX 15
#define HASH_BITS_MAX 16
#define LIT_BUFSIZE_MAX 32768
#define W_SIZE (s->w_size)
#define W_MASK (s->w_size - 1)
#define HASH_SIZE (s->hash_size)
#define HASH_MASK (s->hash_size - 1)
#define HASH_SHIFT (s->hash_shift)
#define MIN_MATCH 3
#define MAX_MATCH 258
#define MIN_LOOKAHEAD (MAX_MATCH + MIN_MATCH + 1)
#define MAX_DIST (W_SIZE - MIN_LOOKAHEAD)
#define TOO_FAR 4096
#define LIT_BUFSIZE (s->lit_bufsize)
#define NIL 0

#define L_CODES 286
#define D_CODES 30
#define BL_CODES 19
#define HEAP_SIZE (2 * L_CODES + 1)
#define LITERALS 256
#define END_BLOCK 256
#define MAX_BITS 15

static const int extra_lbits[29] = { 0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0 };
static const int extra_dbits[30] = { 0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13 };
static const int extra_blbits[19] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,2,3,7 };
static const u8 bl_order[19] = { 16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15 };

typedef struct { u16 freq; u16 code; u16 dad; u16 len; } ct;

static u8 length_code[256], dist_code[512]; static int base_length[29], base_dist[30];
static ct static_ltree[L_CODES + 2], static_dtree[D_CODES]; static int tables_ready = 0;

static unsigned bi_reverse(unsigned code, int len) { unsigned res = 0; do { res |= code & 1; code >>= 1; res <<= 1; } while (--len > 0); return res >> 1; }

typedef struct {
    const ct* stree; const int* extra; int base, elems, max_length;
} sdesc;
typedef struct { ct* tree; int max_code; sdesc sd; } tdesc;

typedef struct {
    /* input / window */
    const u8* in; size_t in_len, in_pos;
    u32 w_size, hash_size, hash_shift, lit_bufsize;
    u8 window[2 << W_BITS_MAX]; u16 prev[1 << W_BITS_MAX]; u16 head[1 << HASH_BITS_MAX];
    u32 strstart, lookahead, match_start, match_length, prev_length, prev_match, ins_h, insert; long block_start; int match_available;
    u32 good_match, max_lazy, nice_match, max_chain;      /* zlib's configuration_table row of the level (deflate_slow levels 4 .. 9) */
    /* symbols */
    u16 d_buf[LIT_BUFSIZE_MAX]; u8 l_buf[LIT_BUFSIZE_MAX]; u32 last_lit;
    ct dyn_ltree[HEAP_SIZE], dyn_dtree[2 * D_CODES + 1], bl_tree[2 * BL_CODES + 1];
    tdesc l_desc, d_desc, bl_desc;
    u16 bl_count[MAX_BITS + 1]; int heap[2 * L_CODES + 1]; int heap_len, heap_max; u8 depth[2 * L_CODES + 1];
    u64 opt_len, static_len;
    /* output */
    u8* out; size_t out_cap, out_pos; int overflow; u32 bi_buf; int bi_valid;
} dstate;

static void gen_codes(ct* tree, int max_code, u16* bl_count)
{
    u16 next_code[MAX_BITS + 1]; unsigned code = 0; int bits, n;
    for (bits = 1; bits <= MAX_BITS; bits++) { code = (code + bl_count[bits - 1]) << 1; next_code[bits] = (u16)code; }
    for (n = 0; n <= max_code; n++) { int len = tree[n].len; if (len == 0) continue; tree[n].code = (u16)bi_reverse(next_code[len]++, len); }
}

static void tr_static_init(void)
{
    int n, bits, length, code, dist; u16 bl_count[MAX_BITS + 1];
    /* (once, whatever the number of threads that arrive here together: the tests' helpers may call from a pool) */
    static int lock = 0;
    if (__atomic_load_n(&tables_ready, __ATOMIC_ACQUIRE)) return;
    while (__atomic_exchange_n(&lock, 1, __ATOMIC_ACQUIRE)) { }
    if (tables_ready) { __atomic_store_n(&lock, 0, __ATOMIC_RELEASE); return; }
    length = 0;
    for (code = 0; code < 28; code++) { base_length[code] = length; for (n = 0; n < (1 << extra_lbits[code]); n++) length_code[length++] = (u8)code; }
    length_code[length - 1] = (u8)code;
    dist = 0;
    for (code = 0; code < 16; code++) { base_dist[code] = dist; for (n = 0; n < (1 << extra_dbits[code]); n++) dist_code[dist++] = (u8)code; }
    dist >>= 7;
    for (; code < D_CODES; code++) { base_dist[code] = dist << 7; for (n = 0; n < (1 << (extra_dbits[code] - 7)); n++) dist_code[256 + dist++] = (u8)code; }
    for (bits = 0; bits <= MAX_BITS; bits++) bl_count[bits] = 0;
    n = 0;
    while (n <= 143) static_ltree[n++].len = 8, bl_count[8]++;
    while (n <= 255) static_ltree[n++].len = 9, bl_count[9]++;
    while (n <= 279) static_ltree[n++].len = 7, bl_count[7]++;
    while (n <= 287) static_ltree[n++].len = 8, bl_count[8]++;
    gen_codes(static_ltree, L_CODES + 1, bl_count);
    for (n = 0; n < D_CODES; n++) { static_dtree[n].len = 5; static_dtree[n].code = (u16)bi_reverse((unsigned)n, 5); }
    __atomic_store_n(&tables_ready, 1, __ATOMIC_RELEASE);
    __atomic_store_n(&lock, 0, __ATOMIC_RELEASE);
}
#define d_code(dist) ((dist) < 256 ? dist_code[dist] : dist_code[256 + ((dist) >> 7)])

/* ---- bit output -------------------------------------------------------- */
static void put_byte(dstate* s, unsigned c) { if (s->out_pos < s->out_cap) s->out[s->out_pos] = (u8)c; else s->overflow = 1; s->out_pos++; }
static void send_bits(dstate* s, unsigned value, int length)
{
    s->bi_buf |= (u32)value << s->bi_valid; s->bi_valid += length;
    while (s->bi_valid >= 8) { put_byte(s, s->bi_buf & 0xFF); s->bi_buf >>= 8; s->bi_valid -= 8; }
}
static void bi_windup(dstate* s) { if (s->bi_valid > 0) put_byte(s, s->bi_buf & 0xFF); s->bi_buf = 0; s->bi_valid = 0; }
#define send_code(s, c, tree) send_bits(s, (tree)[c].code, (tree)[c].len)

/* ---- trees ------------------------------------------------------------- */
static void init_block(dstate* s)
{
    int n;
    for (n = 0; n < L_CODES; n++) s->dyn_ltree[n].freq = 0;
    for (n = 0; n < D_CODES; n++) s->dyn_dtree[n].freq = 0;
    for (n = 0; n < BL_CODES; n++) s->bl_tree[n].freq = 0;
    s->dyn_ltree[END_BLOCK].freq = 1;
    s->opt_len = s->static_len = 0; s->last_lit = 0;
}
#define smaller(tree, n, m, depth) (tree[n].freq < tree[m].freq || (tree[n].freq == tree[m].freq && depth[n] <= depth[m]))
static void pqdownheap(dstate* s, ct* tree, int k)
{
    int v = s->heap[k]; int j = k << 1;
    while (j <= s->heap_len) {
        if (j < s->heap_len && smaller(tree, s->heap[j + 1], s->heap[j], s->depth)) j++;
        if (smaller(tree, v, s->heap[j], s->depth)) break;
        s->heap[k] = s->heap[j]; k = j; j <<= 1;
    }
    s->heap[k] = v;
}
static void gen_bitlen(dstate* s, tdesc* desc)
{
    ct* tree = desc->tree; int max_code = desc->max_code; const ct* stree = desc->sd.stree; const int* extra = desc->sd.extra;
    int base = desc->sd.base, max_length = desc->sd.max_length; int h, n, m, bits, xbits; u16 f; int overflow = 0;
    for (bits = 0; bits <= MAX_BITS; bits++) s->bl_count[bits] = 0;
    tree[s->heap[s->heap_max]].len = 0;
    for (h = s->heap_max + 1; h < HEAP_SIZE; h++) {
        n = s->heap[h]; bits = tree[tree[n].dad].len + 1;
        if (bits > max_length) bits = max_length, overflow++;
        tree[n].len = (u16)bits;
        if (n > max_code) continue;
        s->bl_count[bits]++;
        xbits = 0; if (n >= base) xbits = extra[n - base];
        f = tree[n].freq;
        s->opt_len += (u64)f * (unsigned)(bits + xbits);
        if (stree) s->static_len += (u64)f * (unsigned)(stree[n].len + xbits);
    }
    if (overflow == 0) return;
    do {
        bits = max_length - 1;
        while (s->bl_count[bits] == 0) bits--;
        s->bl_count[bits]--; s->bl_count[bits + 1] += 2; s->bl_count[max_length]--;
        overflow -= 2;
    } while (overflow > 0);
    for (bits = max_length; bits != 0; bits--) {
        n = s->bl_count[bits];
        while (n != 0) {
            m = s->heap[--h];
            if (m > max_code) continue;
            if ((unsigned)tree[m].len != (unsigned)bits) { s->opt_len += ((u64)bits - tree[m].len) * tree[m].freq; tree[m].len = (u16)bits; }
            n--;
        }
    }
}
static void build_tree(dstate* s, tdesc* desc)
{
    ct* tree = desc->tree; const ct* stree = desc->sd.stree; int elems = desc->sd.elems; int n, m, max_code = -1, node;
    s->heap_len = 0; s->heap_max = HEAP_SIZE;
    for (n = 0; n < elems; n++) {
        if (tree[n].freq != 0) { s->heap[++(s->heap_len)] = max_code = n; s->depth[n] = 0; } else tree[n].len = 0;
    }
    while (s->heap_len < 2) {
        node = s->heap[++(s->heap_len)] = (max_code < 2 ? ++max_code : 0);
        tree[node].freq = 1; s->depth[node] = 0; s->opt_len--; if (stree) s->static_len -= stree[node].len;
    }
    desc->max_code = max_code;
    for (n = s->heap_len / 2; n >= 1; n--) pqdownheap(s, tree, n);
    node = elems;
    do {
        n = s->heap[1]; s->heap[1] = s->heap[s->heap_len--]; pqdownheap(s, tree, 1);
        m = s->heap[1];
        s->heap[--(s->heap_max)] = n; s->heap[--(s->heap_max)] = m;
        tree[node].freq = (u16)(tree[n].freq + tree[m].freq);
        s->depth[node] = (u8)((s->depth[n] >= s->depth[m] ? s->depth[n] : s->depth[m]) + 1);
        tree[n].dad = tree[m].dad = (u16)node;
        s->heap[1] = node++;
        pqdownheap(s, tree, 1);
    } while (s->heap_len >= 2);
    s->heap[--(s->heap_max)] = s->heap[1];
    gen_bitlen(s, desc);
    gen_codes(tree, max_code, s->bl_count);
}
static void scan_tree(dstate* s, ct* tree, int max_code)
{
    int n, prevlen = -1, curlen, nextlen = tree[0].len, count = 0, max_count = 7, min_count = 4;
    if (nextlen == 0) max_count = 138, min_count = 3;
    tree[max_code + 1].len = (u16)0xffff;
    for (n = 0; n <= max_code; n++) {
        curlen = nextlen; nextlen = tree[n + 1].len;
        if (++count < max_count && curlen == nextlen) continue;
        else if (count < min_count) s->bl_tree[curlen].freq += (u16)count;
        else if (curlen != 0) { if (curlen != prevlen) s->bl_tree[curlen].freq++; s->bl_tree[16].freq++; }
        else if (count <= 10) s->bl_tree[17].freq++;
        else s->bl_tree[18].freq++;
        count = 0; prevlen = curlen;
        if (nextlen == 0) max_count = 138, min_count = 3;
        else if (curlen == nextlen) max_count = 6, min_count = 3;
        else max_count = 7, min_count = 4;
    }
}
static void send_tree(dstate* s, ct* tree, int max_code)
{
    int n, prevlen = -1, curlen, nextlen = tree[0].len, count = 0, max_count = 7, min_count = 4;
    if (nextlen == 0) max_count = 138, min_count = 3;
    for (n = 0; n <= max_code; n++) {
        curlen = nextlen; nextlen = tree[n + 1].len;
        if (++count < max_count && curlen == nextlen) continue;
        else if (count < min_count) { do { send_code(s, curlen, s->bl_tree); } while (--count != 0); }
        else if (curlen != 0) {
            if (curlen != prevlen) { send_code(s, curlen, s->bl_tree); count--; }
            send_code(s, 16, s->bl_tree); send_bits(s, (unsigned)(count - 3), 2);
        } else if (count <= 10) { send_code(s, 17, s->bl_tree); send_bits(s, (unsigned)(count - 3), 3); }
        else { send_code(s, 18, s->bl_tree); send_bits(s, (unsigned)(count - 11), 7); }
        count = 0; prevlen = curlen;
        if (nextlen == 0) max_count = 138, min_count = 3;
        else if (curlen == nextlen) max_count = 6, min_count = 3;
        else max_count = 7, min_count = 4;
    }
}
static int build_bl_tree(dstate* s)
{
    int max_blindex;
    scan_tree(s, s->dyn_ltree, s->l_desc.max_code);
    scan_tree(s, s->dyn_dtree, s->d_desc.max_code);
    build_tree(s, &s->bl_desc);
    for (max_blindex = BL_CODES - 1; max_blindex >= 3; max_blindex--) if (s->bl_tree[bl_order[max_blindex]].len != 0) break;
    s->opt_len += 3 * ((u64)max_blindex + 1) + 5 + 5 + 4;
    return max_blindex;
}
static void compress_block(dstate* s, const ct* ltree, const ct* dtree)
{
    unsigned dist; int lc; unsigned lx = 0; unsigned code; int extra;
    if (s->last_lit != 0) do {
        dist = s->d_buf[lx]; lc = s->l_buf[lx++];
        if (dist == 0) { send_code(s, lc, ltree); }
        else {
            code = length_code[lc];
            send_code(s, code + LITERALS + 1, ltree);
            extra = extra_lbits[code];
            if (extra != 0) { lc -= base_length[code]; send_bits(s, (unsigned)lc, extra); }
            dist--;
            code = d_code(dist);
            send_code(s, code, dtree);
            extra = extra_dbits[code];
            if (extra != 0) { dist -= (unsigned)base_dist[code]; send_bits(s, dist, extra); }
        }
    } while (lx < s->last_lit);
    send_code(s, END_BLOCK, ltree);
}
static void tr_flush_block(dstate* s, const u8* buf, u32 stored_len, int last)
{
    u64 opt_lenb, static_lenb; int max_blindex, rank;
    build_tree(s, &s->l_desc);
    build_tree(s, &s->d_desc);
    max_blindex = build_bl_tree(s);
    opt_lenb = (s->opt_len + 3 + 7) >> 3;
    static_lenb = (s->static_len + 3 + 7) >> 3;
    if (static_lenb <= opt_lenb) opt_lenb = static_lenb;
    if ((u64)stored_len + 4 <= opt_lenb && buf != NULL) {
        u32 i;
        send_bits(s, (0 << 1) + (unsigned)last, 3);
        bi_windup(s);
        put_byte(s, stored_len & 0xFF); put_byte(s, (stored_len >> 8) & 0xFF);
        put_byte(s, ~stored_len & 0xFF); put_byte(s, (~stored_len >> 8) & 0xFF);
        for (i = 0; i < stored_len; i++) put_byte(s, buf[i]);
    } else if (static_lenb == opt_lenb) {
        send_bits(s, (1 << 1) + (unsigned)last, 3);
        compress_block(s, static_ltree, static_dtree);
    } else {
        send_bits(s, (2 << 1) + (unsigned)last, 3);
        send_bits(s, (unsigned)(s->l_desc.max_code + 1 - 257), 5);
        send_bits(s, (unsigned)(s->d_desc.max_code + 1 - 1), 5);
        send_bits(s, (unsigned)(max_blindex + 1 - 4), 4);
        for (rank = 0; rank < max_blindex + 1; rank++) send_bits(s, s->bl_tree[bl_order[rank]].len, 3);
        send_tree(s, s->dyn_ltree, s->l_desc.max_code);
        send_tree(s, s->dyn_dtree, s->d_desc.max_code);
        compress_block(s, s->dyn_ltree, s->dyn_dtree);
    }
    init_block(s);
    if (last) bi_windup(s);
}
static int tally(dstate* s, unsigned dist, unsigned lc)
{
    s->d_buf[s->last_lit] = (u16)dist; s->l_buf[s->last_lit++] = (u8)lc;
    if (dist == 0) s->dyn_ltree[lc].freq++;
    else { dist--; s->dyn_ltree[length_code[lc] + LITERALS + 1].freq++; s->dyn_dtree[d_code(dist)].freq++; }
    return s->last_lit == LIT_BUFSIZE - 1;
}

/* ---- matcher ------------------------------------------------------------- */
#define UPDATE_HASH(s, h, c) (h = (((h) << HASH_SHIFT) ^ (c)) & HASH_MASK)
#define INSERT_STRING(s, str, match_head) \
    (UPDATE_HASH(s, s->ins_h, s->window[(str) + (MIN_MATCH - 1)]), \
     match_head = s->prev[(str) & W_MASK] = s->head[s->ins_h], s->head[s->ins_h] = (u16)(str))

static void fill_window(dstate* s)
{
    unsigned n; u32 more;
    do {
        more = (u32)(2 * W_SIZE - s->lookahead - s->strstart);
        if (s->strstart >= W_SIZE + MAX_DIST) {
            unsigned m; unsigned k;
            memcpy(s->window, s->window + W_SIZE, W_SIZE - more);
            s->match_start -= W_SIZE; s->strstart -= W_SIZE; s->block_start -= (long)W_SIZE;
            for (k = 0; k < HASH_SIZE; k++) { m = s->head[k]; s->head[k] = (u16)(m >= W_SIZE ? m - W_SIZE : NIL); }
            for (k = 0; k < W_SIZE; k++) { m = s->prev[k]; s->prev[k] = (u16)(m >= W_SIZE ? m - W_SIZE : NIL); }
            more += W_SIZE;
        }
        if (s->in_pos == s->in_len) break;
        n = (unsigned)(s->in_len - s->in_pos); if (n > more) n = more;
        memcpy(s->window + s->strstart + s->lookahead, s->in + s->in_pos, n); s->in_pos += n;
        s->lookahead += n;
        if (s->lookahead + s->insert >= MIN_MATCH) {
            u32 str = s->strstart - s->insert;
            s->ins_h = s->window[str];
            UPDATE_HASH(s, s->ins_h, s->window[str + 1]);
            while (s->insert) {
                UPDATE_HASH(s, s->ins_h, s->window[str + MIN_MATCH - 1]);
                s->prev[str & W_MASK] = s->head[s->ins_h]; s->head[s->ins_h] = (u16)str;
                str++; s->insert--;
                if (s->lookahead + s->insert < MIN_MATCH) break;
            }
        }
    } while (s->lookahead < MIN_LOOKAHEAD && s->in_pos != s->in_len);
    /* zlib then zeroes up to 258 bytes past the data it has ever written (its window is not
     * zero-allocated); this window starts zeroed and bytes once written keep their value, as there */
}

/* hooks of tools/deflate_predict_model.c (a model of a predicted-parse scheme); nothing by default */
#ifndef DREF_TRACE_REQ
#define DREF_TRACE_REQ(pos) ((void)0)
#define DREF_CHAIN(c) (c)
#endif
static u32 longest_match(dstate* s, u32 cur_match)
{
    unsigned chain_length = DREF_CHAIN(s->max_chain); const u8* scan = s->window + s->strstart; const u8* match; int len;
    int best_len = (int)s->prev_length; int nice_match = (int)s->nice_match;
    u32 limit = s->strstart > MAX_DIST ? s->strstart - MAX_DIST : NIL;
    const u8* strend = s->window + s->strstart + MAX_MATCH;
    u8 scan_end1 = scan[best_len - 1], scan_end = scan[best_len];
    if (s->prev_length >= s->good_match) chain_length >>= 2;
    if ((u32)nice_match > s->lookahead) nice_match = (int)s->lookahead;
    do {
        match = s->window + cur_match;
        if (match[best_len] != scan_end || match[best_len - 1] != scan_end1 || match[0] != scan[0] || match[1] != scan[1]) continue;
        { const u8* sc = scan + 2; const u8* mt = match + 2;
          while (sc < strend && *sc == *mt) { sc++; mt++; }
          len = MAX_MATCH - (int)(strend - sc); }
        if (len > best_len) {
            s->match_start = cur_match; best_len = len;
            if (len >= nice_match) break;
            scan_end1 = scan[best_len - 1]; scan_end = scan[best_len];
        }
    } while ((cur_match = s->prev[cur_match & W_MASK]) > limit && --chain_length != 0);
    if ((u32)best_len <= s->lookahead) return (u32)best_len;
    return s->lookahead;
}

#define FLUSH_BLOCK(s, last) { \
    tr_flush_block(s, (s->block_start >= 0L ? &s->window[(unsigned)s->block_start] : NULL), (u32)((long)s->strstart - s->block_start), (last)); \
    s->block_start = (long)s->strstart; }

static void deflate_slow_finish(dstate* s)
{
    u32 hash_head; int bflush;
    for (;;) {
        if (s->lookahead < MIN_LOOKAHEAD) { fill_window(s); if (s->lookahead == 0) break; }
        hash_head = NIL;
        if (s->lookahead >= MIN_MATCH) { INSERT_STRING(s, s->strstart, hash_head); }
        s->prev_length = s->match_length; s->prev_match = s->match_start;
        s->match_length = MIN_MATCH - 1;
        if (hash_head != NIL && s->prev_length < s->max_lazy && s->strstart - hash_head <= MAX_DIST) {
            DREF_TRACE_REQ(s->strstart);
            s->match_length = longest_match(s, hash_head);
            if (s->match_length <= 5 && (s->match_length == MIN_MATCH && s->strstart - s->match_start > TOO_FAR)) s->match_length = MIN_MATCH - 1;
        }
        if (s->prev_length >= MIN_MATCH && s->match_length <= s->prev_length) {
            u32 max_insert = s->strstart + s->lookahead - MIN_MATCH;
            bflush = tally(s, s->strstart - 1 - s->prev_match, s->prev_length - MIN_MATCH);
            s->lookahead -= s->prev_length - 1;
            s->prev_length -= 2;
            do { if (++s->strstart <= max_insert) { INSERT_STRING(s, s->strstart, hash_head); } } while (--s->prev_length != 0);
            s->match_available = 0; s->match_length = MIN_MATCH - 1; s->strstart++;
            if (bflush) FLUSH_BLOCK(s, 0);
        } else if (s->match_available) {
            bflush = tally(s, 0, s->window[s->strstart - 1]);
            if (bflush) FLUSH_BLOCK(s, 0);
            s->strstart++; s->lookahead--;
        } else { s->match_available = 1; s->strstart++; s->lookahead--; }
    }
    if (s->match_available) { tally(s, 0, s->window[s->strstart - 1]); s->match_available = 0; }
    FLUSH_BLOCK(s, 1);
}

/* zlib deflate.c deflate_fast (levels 1 .. 3): no lazy evaluation; the strings inside a match enter the hash chains
 * only when the match is at most max_insert_length long (the max_lazy column of the table), so the chains depend on the parse */
static void deflate_fast_finish(dstate* s)
{
    u32 hash_head; int bflush;
    for (;;) {
        if (s->lookahead < MIN_LOOKAHEAD) { fill_window(s); if (s->lookahead == 0) break; }
        hash_head = NIL;
        if (s->lookahead >= MIN_MATCH) { INSERT_STRING(s, s->strstart, hash_head); }
        if (hash_head != NIL && s->strstart - hash_head <= MAX_DIST) s->match_length = longest_match(s, hash_head);
        if (s->match_length >= MIN_MATCH) {
            bflush = tally(s, s->strstart - s->match_start, s->match_length - MIN_MATCH);
            s->lookahead -= s->match_length;
            if (s->match_length <= s->max_lazy && s->lookahead >= MIN_MATCH) {
                s->match_length--;
                do { s->strstart++; INSERT_STRING(s, s->strstart, hash_head); } while (--s->match_length != 0);
                s->strstart++;
            } else {
                s->strstart += s->match_length; s->match_length = 0;
                s->ins_h = s->window[s->strstart];
                UPDATE_HASH(s, s->ins_h, s->window[s->strstart + 1]);
            }
        } else {
            bflush = tally(s, 0, s->window[s->strstart]);
            s->lookahead--; s->strstart++;
        }
        if (bflush) FLUSH_BLOCK(s, 0);
    }
    s->insert = s->strstart < MIN_MATCH - 1 ? s->strstart : MIN_MATCH - 1;
    FLUSH_BLOCK(s, 1);
}

DREF_API size_t dref_deflate_bound(size_t n) { return n + (n >> 12) + (n >> 14) + (n >> 25) + 13 + 64; }

/* raw deflate at level 1 .. 9 (1 .. 3 deflate_fast, 4 .. 9 deflate_slow; zlib's configuration_table: good_length, max_lazy, nice_length, max_chain),
 * windowBits 15, memLevel 8, strategy 0, one shot. returns size or (size_t)-1 */
DREF_API size_t dref_deflate_raw_level(u8* dst, size_t cap, const u8* src, size_t n, int level);
DREF_API size_t dref_deflate_raw_params(u8* dst, size_t cap, const u8* src, size_t n, int level, int window_bits, int mem_level);
DREF_API size_t dref_deflate_l6_raw(u8* dst, size_t cap, const u8* src, size_t n) { return dref_deflate_raw_level(dst, cap, src, n, 6); }
DREF_API size_t dref_deflate_raw_level(u8* dst, size_t cap, const u8* src, size_t n, int level) { return dref_deflate_raw_params(dst, cap, src, n, level, 15, 8); }
/* zlib's bound for settings other than the default ones (deflateBound's conservative branch: fixed blocks at nine bits a literal) */
DREF_API size_t dref_deflate_bound_params(size_t n) { return n + ((n + 7) >> 3) + ((n + 63) >> 6) + 5 + 64; }
/* ... with deflateInit2's windowBits 9 .. 15 (8 is served as 9, as zlib does) and memLevel 1 .. 9 */
DREF_API size_t dref_deflate_raw_params(u8* dst, size_t cap, const u8* src, size_t n, int level, int window_bits, int mem_level)
{
    static const u32 cfg[10][4] = { {0,0,0,0}, { 4, 4, 8, 4 }, { 4, 5, 16, 8 }, { 4, 6, 32, 32 },
        { 4, 4, 16, 16 }, { 8, 16, 32, 32 }, { 8, 16, 128, 128 }, { 8, 32, 128, 256 }, { 32, 128, 258, 1024 }, { 32, 258, 258, 4096 } };
    dstate* s; size_t r;
    if (level < 1 || level > 9 || window_bits < 8 || window_bits > 15 || mem_level < 1 || mem_level > 9) return (size_t)-1;
    if (window_bits == 8) window_bits = 9;
    s = (dstate*)calloc(1, sizeof(dstate));
    if (!s) return (size_t)-1;
    s->w_size = 1u << window_bits; s->hash_size = 1u << (mem_level + 7); s->hash_shift = (u32)(mem_level + 7 + MIN_MATCH - 1) / MIN_MATCH;
    s->lit_bufsize = 1u << (mem_level + 6);
    s->good_match = cfg[level][0]; s->max_lazy = cfg[level][1]; s->nice_match = cfg[level][2]; s->max_chain = cfg[level][3];
    tr_static_init();
    s->in = src; s->in_len = n; s->out = dst; s->out_cap = cap;
    s->l_desc.tree = s->dyn_ltree; s->l_desc.sd.stree = static_ltree; s->l_desc.sd.extra = extra_lbits; s->l_desc.sd.base = LITERALS + 1; s->l_desc.sd.elems = L_CODES; s->l_desc.sd.max_length = MAX_BITS;
    s->d_desc.tree = s->dyn_dtree; s->d_desc.sd.stree = static_dtree; s->d_desc.sd.extra = extra_dbits; s->d_desc.sd.base = 0; s->d_desc.sd.elems = D_CODES; s->d_desc.sd.max_length = MAX_BITS;
    s->bl_desc.tree = s->bl_tree; s->bl_desc.sd.stree = NULL; s->bl_desc.sd.extra = extra_blbits; s->bl_desc.sd.base = 0; s->bl_desc.sd.elems = BL_CODES; s->bl_desc.sd.max_length = 7;
    init_block(s);
    s->match_length = s->prev_length = MIN_MATCH - 1;
    if (level < 4) deflate_fast_finish(s); else deflate_slow_finish(s);
    r = s->overflow ? (size_t)-1 : s->out_pos;
    free(s);
    return r;
}
