// deflate_match.h -- the LZ77 half of zlib level 6 ("deflate_slow": 15-bit hash of
// 3 bytes, hash chains, lazy matching, good/lazy/nice/chain = 8/16/128/128) for many
// independent slices (positions are 32-bit; match candidates are kept as distances, which zlib's 32 KiB window bounds).
//
// Replaces zlib's deflate() behind the reference's ZlibCompressor(ZlibFormat.Raw, 6)
// (kompressor-zlib--nativelib/src/jvmCommonMain/jni/Wrapper.cpp:20,73; Kotlin side
// .../zlib/ZlibCompressor.jvm.kt:19-47).  The stream must equal zlib's byte for byte.
//
// Unlike zstd's double-fast parse, zlib inserts EVERY position into its hash chains,
// so the candidate list of a position does not depend on the parse.  That splits the
// work into three kernels:
//   k_deflate_chains : link[p] = previous position with the same 3-byte hash
//                      (one workgroup of 1 .. 4 waves per slice, head table in LDS)
//   k_deflate_best   : for every position, the match zlib's longest_match would
//                      return after 32 and after 128 chain steps (1024 threads per
//                      slice, chain links + source bytes of the 32 KiB window in LDS)
//   k_deflate_parse  : the lazy-evaluation decision chain, one LANE per slice, which
//                      now only looks results up; emits the symbol list and the
//                      block boundaries (every 16383 symbols)
#pragma once
#include "zstd_common.h"

#define KD_MIN_MATCH 3
#define KD_MAX_MATCH 258
#define KD_MIN_LOOKAHEAD 262
#define KD_WSIZE 32768                                   /* the largest window (windowBits 15); a.wsize is the stream's */
#define KD_MAX_DIST (KD_WSIZE - KD_MIN_LOOKAHEAD)       /* 32506: the largest distance; a.max_dist is the stream's */
#define KD_TOO_FAR 4096
#define KD_LIT_BUFSIZE 16384                            /* memLevel 8; a.lit_buf is the stream's (128 .. 32 768) */
#define KD_MAX_SLICE (1u << 30)

// per position: what longest_match returns with a full chain (128 steps) and with the
// shortened chain (32 steps, used when the previous match is >= good_match); dist = position - match start (<= MAX_DIST)
struct KdBest { u16 len128, dist128, len32, dist32; };

struct KdBlockInfo { u32 nsym_end; u32 end_pos; u32 start_pos; u32 stored_ok; };   // symbols [prev nsym_end, nsym_end)
struct KdSliceMeta { u32 nblocks; u32 nsym; u32 pad[2]; };
// blocks a slice of n bytes can have: one per lit_bufsize - 1 symbols (a symbol covers at least one byte), the last one may be short
KX_DEV u32 kd_block_cap(u32 n, u32 lit_buf = KD_LIT_BUFSIZE) { return n / (lit_buf - 1u) + 2u; }

struct KdArgs {
    const u8* src; const u64* in_off; const u32* in_len; u32 n_slices;
    u32 pos_cap;        // positions of the per-slice arrays below (the context's max_slice_bytes, rounded up to 64)
    u16* link;          // per slice: pos_cap entries: distance to the previous position with the same hash, 0 = none
    KdBest* best;       // per slice: pos_cap entries
    u32* wr = nullptr;  // per slice: pos_cap entries (deflate_lazy.h: where | rank << 16)
    u32* order_key = nullptr; u32* order_hist = nullptr; const u32* order = nullptr;      // deflate_lazy.h: per slice a cost class (0 .. 255) and their histogram, written by k_deflate_sort; the slices ordered by it, largest first
    u32* syms;          // per slice: pos_cap entries: dist | lc << 16
    KdSliceMeta* meta;
    KdBlockInfo* blocks; u32 blk_cap;    // per slice: blk_cap entries
    u8* dst; const u64* out_off; u32* out_len;
    u32 flags;          // timing-only ablations (results wrong): 1 = no match extension, 2 = at most 16 chain steps; 4 = chunks of 8192 positions (results right); bits 8.. = refill threshold
    u32 format;         // 0 = raw deflate, 1 = zlib wrapper (78 9C header, Adler-32 trailer)
    u32 good, lazy, nice, chain;     // zlib's configuration_table row of the level (deflate_slow: levels 4 .. 9; level 6 = 8, 16, 128, 128)
    u32 zflg, gxfl;                  // what the wrappers say about the level: the zlib header's FLG byte (5E / 9C / DA), gzip's XFL (2 at level 9)
    // deflateInit2's windowBits and memLevel (zlib deflate.c deflateInit2_): w_size = 1 << windowBits and MAX_DIST = w_size - 262;
    // hash_bits = memLevel + 7, hash_shift = (hash_bits + 2) / 3; lit_bufsize = 1 << (memLevel + 6) (a block is closed at lit_bufsize - 1 symbols)
    u32 wsize = KD_WSIZE, max_dist = KD_MAX_DIST, hshift = 5, hmask = 0x7FFFu, lit_buf = KD_LIT_BUFSIZE, zcmf = 0x78u;
    // deflate_lazy.h, slices above 64 KiB: the segment a launch works on (its 64 KiB span starts at 32 768 * seg) and where the parse
    // keeps its state between two segments (KDL_STATE_WORDS words per slice)
    u32 seg = 0; u32* seg_state = nullptr; u16* seg_rank = nullptr;       // (seg_rank: the sort's ranks, 65 536 per slice -- up to 64 KiB they sit in the symbol array)
};
// zlib's UPDATE_HASH over three bytes: ((b0 << 2 * hash_shift) ^ (b1 << hash_shift) ^ b2) & hash_mask
KX_DEV u32 kd_hash3(const KdArgs& a, u32 b0, u32 b1, u32 b2) { return ((b0 << (2u * a.hshift)) ^ (b1 << a.hshift) ^ b2) & a.hmask; }
// (KdBest's fields are named after level 6: len128 / dist128 = the result of a full chain, len32 / dist32 = of a quarter of it,
// what longest_match walks when the previous match was at least `good` long)
static inline void kd_level_config(KdArgs& a, int level, int window_bits = 15, int mem_level = 8)        // (host side: fills the kernel arguments)
{
    static const u32 cfg[10][4] = { {8,16,128,128}, { 4, 4, 8, 4 }, { 4, 5, 16, 8 }, { 4, 6, 32, 32 },     // (1 .. 3: deflate_fast, lazy = max_insert_length)
        { 4, 4, 16, 16 }, { 8, 16, 32, 32 }, { 8, 16, 128, 128 }, { 8, 32, 128, 256 }, { 32, 128, 258, 1024 }, { 32, 258, 258, 4096 } };
    int const l = (level < 1 || level > 9) ? 6 : level;
    a.good = cfg[l][0]; a.lazy = cfg[l][1]; a.nice = cfg[l][2]; a.chain = cfg[l][3];
    a.gxfl = l == 9 ? 2u : l == 1 ? 4u : 0u;
    int const wb = window_bits < 9 ? 9 : window_bits > 15 ? 15 : window_bits, ml = mem_level < 1 ? 1 : mem_level > 9 ? 9 : mem_level;   // (windowBits 8 is served as 9, as zlib does)
    a.wsize = 1u << wb; a.max_dist = a.wsize - KD_MIN_LOOKAHEAD;
    a.hshift = (u32)(ml + 7 + KD_MIN_MATCH - 1) / KD_MIN_MATCH; a.hmask = (1u << (ml + 7)) - 1u; a.lit_buf = 1u << (ml + 6);
    // the zlib wrapper's two bytes (deflate.c deflate(), INIT_STATE): CMF = 8 + ((windowBits - 8) << 4), FLG = level flags << 6 plus the
    // check bits that make the pair a multiple of 31 (78 01 / 5E / 9C / DA with a 32 KiB window)
    u32 const lf = l < 2 ? 0u : l < 6 ? 1u : l == 6 ? 2u : 3u;
    u32 hdr = ((8u + ((u32)(wb - 8) << 4)) << 8) | (lf << 6); hdr += 31u - hdr % 31u;
    a.zcmf = hdr >> 8; a.zflg = hdr & 0xFFu;
}

// ---------------------------------------------------------------------------
// k_deflate_chains: 256 threads per workgroup, head[32768] in LDS
// ---------------------------------------------------------------------------
// HEAD: u16 while every slice fits 64 KiB (64 KiB of LDS: two workgroups per CU), u32 for longer ones (128 KiB)
template <class HEAD, int HB = 15>                             // HB: hash bits the table has room for (memLevel 9: 16)
KX_DEV void deflate_chains_body(const KdArgs& a)
{
    KX_SHARED HEAD head[1 << HB];                                // position + 1 of the last string with that hash, 0 = none
    int const lane = kx_lane(); int const wv = kx_wave(); int const nw = kx_nwaves(); int const tid = wv * 64 + lane; int const nthreads = nw * 64;
    for (u32 it = kx_block(); it < a.n_slices; it += kx_nblocks()) {
        u32 const slice = kx_xcd_chunk(it, a.n_slices);
        const u8* const src = a.src + a.in_off[slice]; u32 const n = a.in_len[slice];
        u16* const link = a.link + (size_t)slice * a.pos_cap;
        u32 const nIns = n >= 3 ? n - 2 : 0;                // positions 0 .. n-3 enter the chains, in order
        // a hash wider than the table (memLevel 9's 16 bits): one pass over the slice per value of the bits above the table's
        u32 const npass = ((a.hmask + 1u) >> HB) ? ((a.hmask + 1u) >> HB) : 1u;
        for (u32 pass = 0; pass < npass; pass++) {
        for (int i = tid; i < (1 << HB) && i <= (int)a.hmask; i += nthreads) head[i] = 0;
        kx_block_sync();
        // The turns on the LDS head table are short (three LDS round trips); what a wave would wait for is the global
        // load of its source bytes.  So the bytes are fetched a GROUP of four rounds ahead: the loads of group g + 1
        // are in flight while the turns of group g are taken (the hash is computed when the bytes are used).
        u32 hq[4]; bool vq[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            u32 const p0 = (u32)k * (u32)nthreads + (u32)wv * 64u + (u32)lane;
            vq[k] = p0 < nIns; hq[k] = 0;
            if (vq[k]) hq[k] = (p0 + 4 <= n) ? kx_ld32(src + p0) : ((u32)src[p0] | ((u32)src[p0 + 1] << 8) | ((u32)src[p0 + 2] << 16));
        }
        for (u32 gbase = 0; gbase < nIns; gbase += 4u * (u32)nthreads) {
            u32 hn[4]; bool vn[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                u32 const pn = gbase + (4u + (u32)k) * (u32)nthreads + (u32)wv * 64u + (u32)lane;
                vn[k] = pn < nIns; hn[k] = 0;
                if (vn[k]) hn[k] = (pn + 4 <= n) ? kx_ld32(src + pn) : ((u32)src[pn] | ((u32)src[pn + 1] << 8) | ((u32)src[pn + 2] << 16));
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                u32 const base = gbase + (u32)k * (u32)nthreads;
                if (base >= nIns) break;                    // uniform over the workgroup
                u32 const p = base + (u32)wv * 64u + (u32)lane;
                u32 const hfull = kd_hash3(a, hq[k] & 0xFFu, (hq[k] >> 8) & 0xFFu, (hq[k] >> 16) & 0xFFu);      // the bytes were fetched a group ago
                bool const valid = vq[k] && (hfull >> HB) == pass;
                u32 const h = valid ? (hfull & ((1u << HB) - 1u)) : 0x10000u + (u32)lane;
                u32 lk = 0;                                  // previous position + 1, 0 = none
                for (int w = 0; w < nw; w++) {
                    if (wv == w) {
                        u32 const old = valid ? head[h] : 0u;
                        kx_lockstep();
                        if (valid) head[h] = (HEAD)(p + 1u);
                        kx_lockstep();
                        u32 const chk = valid ? (u32)head[h] : p + 1u;
                        lk = old;
                        // lanes of this wave that share a bucket: one bucket per round; inside a bucket the nearest
                        // lower lane is the predecessor and the highest lane is the one left in the table
                        for (u64 losers = kx_ballot(valid && chk != p + 1u); losers; ) {
                            int const L = (int)kx_ctz64(losers);
                            u32 const hL = kx_bcast(h, L);                       // v_readlane: L is uniform
                            u64 const grp = kx_ballot(valid && h == hL);
                            if (valid && h == hL) {
                                u64 const below = grp & ((1ull << lane) - 1ull);
                                if (below) lk = p + 1u - (u32)(lane - (63 - (int)__builtin_clzll(below)));
                                if ((grp >> lane) == 1ull) head[h] = (HEAD)(p + 1u);  // highest lane of the bucket
                            }
                            losers &= ~grp;
                        }
                    }
                    if (nw > 1) kx_block_sync();
                }
                // zlib's NIL is position 0: a string there is never a candidate; chains end beyond MAX_DIST anyway
                if (valid) { u32 const d = (lk > 1u) ? p - (lk - 1u) : 0u; link[p] = (u16)(d <= a.max_dist ? d : 0u); }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) { hq[k] = hn[k]; vq[k] = vn[k]; }
        }
        kx_block_sync();
        }
    }
}

// ---------------------------------------------------------------------------
// k_deflate_best: 1024 threads per workgroup; 8192-position chunks with their
// 32 KiB history (links + bytes) staged in LDS
// ---------------------------------------------------------------------------
#define KD_CHUNK 16384                                  /* positions searched per staging phase (flags bit 2: 8192, the earlier setting) */
#define KD_HIST 32512                                   /* >= MAX_DIST, multiple of 64 */
struct KdBestLds { u16 lnk[KD_CHUNK + KD_HIST]; u32 sw[(KD_CHUNK + KD_HIST + 272 + 16) / 4]; u32 next; };   // sw = staged bytes, read as aligned words; next = position counter

// 2 / 4 / 8 bytes at byte offset `o` of the staged window.  gfx950 serves unaligned LDS reads (the compiler emits one
// ds_read_u16 / _b32 / _b64 for these), so no word pairs and funnel shifts are needed.
KX_DEV u32 kd_ld16(const u32* sw, int o) { u16 v; __builtin_memcpy(&v, (const u8*)sw + o, 2); return v; }
KX_DEV u32 kd_ld32(const u32* sw, int o) { u32 v; __builtin_memcpy(&v, (const u8*)sw + o, 4); return v; }
KX_DEV u64 kd_ld64(const u32* sw, int o) { u64 v; __builtin_memcpy(&v, (const u8*)sw + o, 8); return v; }

KX_DEV void deflate_best_body(const KdArgs& a)
{
    KX_SHARED KdBestLds lds;
    int const tid = kx_wave() * 64 + kx_lane(); int const nthreads = kx_nwaves() * 64;
    for (u32 it = kx_block(); it < a.n_slices; it += kx_nblocks()) {
        u32 const slice = kx_xcd_chunk(it, a.n_slices);
        const u8* const src = a.src + a.in_off[slice]; int const n = (int)a.in_len[slice];
        const u16* const link = a.link + (size_t)slice * a.pos_cap;
        KdBest* const best = a.best + (size_t)slice * a.pos_cap;
        int const chunk = (a.flags & 4u) ? 8192 : KD_CHUNK;
        for (int cb = 0; cb < n; cb += chunk) {
            int const lo = cb > KD_HIST ? cb - KD_HIST : 0;                 // first position staged
            int const hiP = (cb + chunk < n) ? cb + chunk : n;        // positions [cb, hiP) are searched
            int const hiB = (hiP + 264 < n) ? hiP + 264 : n;                // bytes staged up to here
            kx_block_sync();
            for (int i = lo + tid; i < hiP; i += nthreads) lds.lnk[i - lo] = (i + 2 < n) ? link[i] : (u16)0;
            {   // bytes [lo, hiB) as little-endian words; bytes past n read as 0 and never count (lengths are capped by lookahead)
                int const nwords = (hiB - lo + 3 + 16) >> 2;
                for (int wi = tid; wi < nwords; wi += nthreads) {
                    int const o = lo + 4 * wi; u32 v = 0;
                    if (o + 4 <= n) v = kx_ld32(src + o);
                    else for (int k = 0; o + k < n; k++) v |= (u32)src[o + k] << (8 * k);
                    lds.sw[wi] = v;
                }
            }
            if (tid == 0) lds.next = (u32)cb;
            kx_block_sync();
            // Chains have very different lengths (0 .. 128 steps) and a wave that owns 64 fixed positions runs as long as
            // its longest one.  Instead every lane walks one chain at a time and idle lanes are handed new positions
            // from a counter as soon as 16 of them are free: a wave step then carries ~56 useful lanes instead of ~30.
            {
                bool active = false, drained = false;
                int p = 0, c = 0, steps = 0, bestLen = 2, bestPos = 0, so = 0, maxlen = 0, nice = 0, limit = 0;
                u32 scan01 = 0, scanEnd = 0; KdBest r; r.len128 = 0; r.dist128 = 0; r.len32 = 0; r.dist32 = 0;
                int const lane = kx_lane();
                int const maxSteps = (a.flags & 2u) ? 16 : (int)a.chain;
                int const quarter = maxSteps >> 2, niceMax = (int)a.nice;
                int const refillAt = (a.flags >> 8) ? (int)(a.flags >> 8) : 16;      // idle lanes that trigger a refill (tuning switch)
                int const reps = ((a.flags >> 4) & 15u) ? (int)((a.flags >> 4) & 15u) : 8;   // flags bits 4..7: candidate steps per refill check (1: 349 ms, 2: 332, 4: 325, 8: 320 per 16 384 slices)
                for (;;) {
                    u64 const idle = kx_ballot(!active);
                    int const nidle = (int)kx_popc64(idle);
                    if (!drained && (nidle >= refillAt)) {
                        int const first = (int)kx_ctz64(idle);
                        u32 base = 0;
                        if (lane == first) base = kx_lds_add(&lds.next, (u32)nidle);
                        base = kx_bcast(base, first);                // v_readlane: base, and with it `drained`, stay on the scalar unit
                        if (!active) {
                            int const np = (int)base + (int)kx_popc64(idle & ((1ull << lane) - 1ull));
                            if (np < hiP) {
                                p = np;
                                r.len128 = 0; r.dist128 = 0; r.len32 = 0; r.dist32 = 0;
                                int const lookahead = n - p;
                                bool go = false;
                                if (lookahead >= KD_MIN_MATCH) {
                                    limit = p > (int)a.max_dist ? p - (int)a.max_dist : 0;
                                    int const d0 = lds.lnk[p - lo];
                                    c = p - d0;
                                    if (d0 != 0) {
                                        nice = lookahead < niceMax ? lookahead : niceMax;
                                        maxlen = lookahead < KD_MAX_MATCH ? lookahead : KD_MAX_MATCH;
                                        so = p - lo;                              // scan offset in the staged window
                                        scan01 = kd_ld16(lds.sw, so);
                                        bestLen = 2; bestPos = 0; steps = 0;
                                        scanEnd = kd_ld16(lds.sw, so + 1);      // scan[best-1], scan[best]
                                        go = true;
                                    }
                                }
                                if (go) active = true; else best[p] = r;          // no candidate at all: done already
                            }
                        }
                        if ((int)base + nidle >= hiP) drained = true;
                    }
                    if (!kx_any(active)) { if (drained) break; continue; }
                    // `reps` candidate steps between two looks at the refill counter: the kernel is bound by instruction issue and
                    // the bookkeeping above costs as much as a step (a lane that finishes early idles for <= reps - 1 steps)
                    for (int rep = 0; rep < reps; rep++) {
                        if (active) {
                            int const mo = c - lo;
                            int const dn = lds.lnk[mo];
                            int const cnext = dn ? c - dn : 0;                 // 0: the chain ends (position 0 is zlib's NIL)
                            bool done = false;
                            steps++;
                            // the candidate can only win if it matches at the current best length too (most fail here)
                            u32 const mEnd = kd_ld16(lds.sw, mo + bestLen - 1);
                            bool const endOk = (bestLen < maxlen) ? (mEnd == scanEnd) : ((mEnd & 0xFFu) == (scanEnd & 0xFFu));
                            if (endOk && kd_ld16(lds.sw, mo) == scan01) {
                                int len = 2;
                                if (a.flags & 1u) len = 3; else
                                for (;;) {                                  // 8 bytes per step
                                    u64 const d = kd_ld64(lds.sw, mo + len) ^ kd_ld64(lds.sw, so + len);
                                    if (d) { len += (int)(kx_ctz64(d) >> 3); break; }
                                    len += 8;
                                    if (len >= maxlen) break;
                                }
                                if (len > maxlen) len = maxlen;
                                if (len > bestLen) {
                                    bestLen = len; bestPos = c;
                                    if (len >= nice) done = true;
                                    else scanEnd = kd_ld16(lds.sw, so + bestLen - 1);
                                }
                            }
                            if (steps == quarter || (done && steps < quarter)) { r.len32 = (u16)(bestLen > 2 ? bestLen : 0); r.dist32 = (u16)(p - bestPos); }
                            c = cnext;
                            if (done || !(c > limit && steps < maxSteps)) {
                                if (steps < quarter && !done) { r.len32 = (u16)(bestLen > 2 ? bestLen : 0); r.dist32 = (u16)(p - bestPos); }
                                r.len128 = (u16)(bestLen > 2 ? bestLen : 0); r.dist128 = (u16)(p - bestPos);
                                best[p] = r;
                                active = false;
                            }
                        }
                    }
                }
            }
        }
        kx_block_sync();
    }
}

// ---------------------------------------------------------------------------
// k_deflate_parse: one lane per slice (64 slices per wave)
// ---------------------------------------------------------------------------
KX_DEV void deflate_parse_body(const KdArgs& a)
{
    u32 const slice = kx_block() * 64u + (u32)kx_lane();
    if (slice >= a.n_slices) return;
    const u8* const src = a.src + a.in_off[slice]; int const n = (int)a.in_len[slice];
    const KdBest* const best = a.best + (size_t)slice * a.pos_cap;
    u32* const syms = a.syms + (size_t)slice * a.pos_cap;
    KdBlockInfo* const blocks = a.blocks + (size_t)slice * a.blk_cap;
    KdSliceMeta mm; mm.nblocks = 0; mm.nsym = 0; mm.pad[0] = 0; mm.pad[1] = 0;
    int strstart = 0; int match_length = 2, prev_length = 2; int match_dist = 0, prev_dist = 0; bool match_available = false;
    u32 nsym = 0, blockSyms = 0; int block_start = 0;
    // zlib's 64 KiB window buffer: it holds the bytes [base, dataEnd) of the slice; fill_window tops it up when fewer than
    // MIN_LOOKAHEAD bytes lie ahead, after moving everything down by 32 KiB once strstart has reached WSIZE + MAX_DIST.
    // Only one thing depends on it here: a block whose start has left the buffer cannot be emitted as a stored block.
    int const W = (int)a.wsize, MD = (int)a.max_dist; u32 const LB = a.lit_buf;
    int base = 0, dataEnd = n < 2 * W ? n : 2 * W;
#define KD_FLUSH(last_) { KdBlockInfo b_; \
        b_.nsym_end = nsym; b_.end_pos = (u32)strstart; b_.start_pos = (u32)block_start; b_.stored_ok = (block_start - base >= 0) ? 1u : 0u; \
        if (mm.nblocks < a.blk_cap) blocks[mm.nblocks] = b_; \
        mm.nblocks++; block_start = strstart; blockSyms = 0; }
#define KD_TALLY(dist_, lc_) { syms[nsym++] = (u32)(dist_) | ((u32)(lc_) << 16); blockSyms++; }
    for (;;) {
        if (dataEnd - strstart < KD_MIN_LOOKAHEAD) {
            // fill_window: one pass is enough (it brings at least 65 536 - strstart bytes, or all that is left)
            int const rel = strstart - base;
            int const slide = (rel >= W + MD) ? W : 0;
            base += slide;
            int const more = 2 * W - (dataEnd - base);
            dataEnd += (n - dataEnd < more) ? n - dataEnd : more;
            if (dataEnd == strstart) break;
        }
        int const lookahead = n - strstart;          // (what lies ahead in the buffer is this, or at least MIN_LOOKAHEAD > MAX_MATCH)
        prev_length = match_length; prev_dist = match_dist;
        match_length = KD_MIN_MATCH - 1;
        if (lookahead >= KD_MIN_MATCH && prev_length < (int)a.lazy) {
            // longest_match starts from best_len = prev_length, so only a longer match changes anything;
            // a previous match >= good_match (8) shortens the chain walk to 32 steps
            KdBest const r = best[strstart];
            int const len = prev_length >= (int)a.good ? r.len32 : r.len128, dist = prev_length >= (int)a.good ? r.dist32 : r.dist128;
            // (a string at the window's base is zlib's NIL -- slide_hash turns position w_size into 0 --: a head of the chain exactly
            // MAX_DIST back that lies there is no candidate, and it was the only one: the next link is farther.  Only at the end of the
            // input can strstart sit exactly MAX_DIST above the base: fill_window then runs, and slides, at every step.)
            if (len > prev_length && !(dist == MD && strstart - dist <= base)) {
                match_length = len; match_dist = dist;
                if (match_length == KD_MIN_MATCH && match_dist > KD_TOO_FAR) match_length = KD_MIN_MATCH - 1;
            }
        }
        if (prev_length >= KD_MIN_MATCH && match_length <= prev_length) {
            KD_TALLY(prev_dist, prev_length - KD_MIN_MATCH)
            bool const bflush = blockSyms == LB - 1u;
            strstart += prev_length - 1;
            match_available = false; match_length = KD_MIN_MATCH - 1;
            if (bflush) KD_FLUSH(0)
        } else if (match_available) {
            KD_TALLY(0, src[strstart - 1])
            if (blockSyms == LB - 1u) KD_FLUSH(0)
            strstart++;
        } else { match_available = true; strstart++; }
        KX_OPAQUE(match_length); KX_OPAQUE(strstart);          // one flat loop: see kx_wave.h
    }
    if (match_available) KD_TALLY(0, src[strstart - 1])
    KD_FLUSH(1)
    mm.nsym = nsym;
    a.meta[slice] = mm;
#undef KD_TALLY
#undef KD_FLUSH
}

// ---------------------------------------------------------------------------
// k_deflate_fast: zlib's deflate_fast (levels 1 .. 3), one lane per slice
// ---------------------------------------------------------------------------
// deflate_fast takes the first acceptable match at a position and, when that match is longer than max_insert_length (the
// max_lazy column: 4 / 5 / 6), does NOT enter the strings inside it into the hash chains: the chains depend on the parse, so
// the split into a chain kernel and a parse kernel of the lazy levels does not apply.  The chains are short instead (4 / 8 / 32
// steps), so one lane per slice walks them straight in HBM (head and prev tables of position + 1, 32 768 words each per
// slice, in the workspace that holds KdBest at the lazy levels); tens of thousands of slices in flight hide the latency.
// Positions are absolute; zlib's window-relative ones are position - base, where base follows fill_window's slides:
// an entry is NIL when it is 0 or not above base (slide_hash would have zeroed it).
// One loop iteration handles one SYMBOL: the <= 5 insert-only positions a short match left behind and the next search
// position.  Their head-table loads are issued together (one memory latency instead of up to six) and strings of equal hash
// among them are put in order in registers, as the stores then are in memory.
KX_DEV void deflate_fast_body(const KdArgs& a)
{
    u32 const slice = kx_block() * 64u + (u32)kx_lane();
    if (slice >= a.n_slices) return;
    const u8* const src = a.src + a.in_off[slice]; int const n = (int)a.in_len[slice];
    u32* const head = (u32*)a.best + (size_t)slice * (a.hmask + 1u);            // cleared by the host before the launch
    u32* const prev = (u32*)a.best + (size_t)a.n_slices * (a.hmask + 1u) + (size_t)slice * KD_WSIZE;     // read only where written
    u32* const syms = a.syms + (size_t)slice * a.pos_cap;
    KdBlockInfo* const blocks = a.blocks + (size_t)slice * a.blk_cap;
    KdSliceMeta mm; mm.nblocks = 0; mm.nsym = 0; mm.pad[0] = 0; mm.pad[1] = 0;
    int strstart = 0, run_n = 0;                                                // run_n: insert-only positions right before strstart
    u32 nsym = 0, blockSyms = 0; int block_start = 0;
    int const W = (int)a.wsize, MD = (int)a.max_dist; u32 const LB = a.lit_buf;
    int base = 0, dataEnd = n < 2 * W ? n : 2 * W;                             // zlib's window buffer holds [base, dataEnd), see k_deflate_parse
    int const maxChain = (int)a.chain, niceMax = (int)a.nice, maxInsert = (int)a.lazy;
#define KDF_FLUSH(last_, end_) { KdBlockInfo b_; \
        b_.nsym_end = nsym; b_.end_pos = (u32)(end_); b_.start_pos = (u32)block_start; b_.stored_ok = (block_start - base >= 0) ? 1u : 0u; \
        if (mm.nblocks < a.blk_cap) blocks[mm.nblocks] = b_; \
        mm.nblocks++; block_start = (end_); blockSyms = 0; }
    // symbols leave in groups of four (one 16-byte store: the kernel is bound by the number of memory requests)
    u32 sq0 = 0, sq1 = 0, sq2 = 0;
#define KDF_SYM(v_) { u32 const v__ = (v_); u32 const k__ = nsym & 3u; \
        if (k__ == 0) sq0 = v__; else if (k__ == 1) sq1 = v__; else if (k__ == 2) sq2 = v__; \
        else kx_st128(syms + (nsym - 3u), (u64)sq0 | (u64)sq1 << 32, (u64)sq2 | (u64)v__ << 32); \
        nsym++; blockSyms++; }
    for (;;) {
        if (dataEnd - strstart < KD_MIN_LOOKAHEAD) {           // (fill_window runs after the inserts of the previous round; they do not depend on it)
            int const rel = strstart - base;
            int const slide = (rel >= W + MD) ? W : 0;
            base += slide;
            int const more = 2 * W - (dataEnd - base);
            dataEnd += (n - dataEnd < more) ? n - dataEnd : more;
            if (dataEnd == strstart) break;
        }
        int const lookahead = dataEnd - strstart;
        u32 w = 0;                                              // the bytes at strstart
        if (strstart + 4 <= n) w = kx_ld32(src + strstart);
        else for (int k = 0; strstart + k < n; k++) w |= (u32)src[strstart + k] << (8 * k);
        // the strings of the run: positions strstart - run_n .. strstart - 1 (a run is only left when 3 bytes follow the match,
        // so the bytes up to strstart + 2 exist; the match before it started at position 1 or later, so strstart >= 4)
        u64 x = 0;                                              // bytes strstart - 5 .. strstart + 2
        if (run_n > 0) {
            if (strstart >= 5) x = kx_ld64(src + strstart - 5);
            else x = (u64)kx_ld32(src) << 8 | (u64)kx_ld16(src + 4) << 40 | (u64)src[6] << 56;     // strstart == 4: bytes -1 .. 6
        }
        u32 rh[5], rv[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {                           // slot k holds position strstart - 5 + k; the run is the last run_n slots
            u32 const b = (u32)(x >> (8 * k));
            rh[k] = kd_hash3(a, b & 0xFFu, (b >> 8) & 0xFFu, (b >> 16) & 0xFFu);
            rv[k] = 0;
            if (k >= 5 - run_n) rv[k] = head[rh[k]];
        }
        u32 const h = kd_hash3(a, w & 0xFFu, (w >> 8) & 0xFFu, (w >> 16) & 0xFFu);
        u32 hh = 0;                                             // INSERT_STRING: position + 1 of the previous string with this hash
        if (lookahead >= KD_MIN_MATCH) hh = head[h];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            if (k >= 5 - run_n) {
                u32 v = rv[k];
#pragma unroll
                for (int j = 0; j < k; j++) if (j >= 5 - run_n && rh[j] == rh[k]) v = (u32)(strstart - 5 + j) + 1u;
                prev[(strstart - 5 + k) & (W - 1)] = v; head[rh[k]] = (u32)(strstart - 5 + k) + 1u;
                if (rh[k] == h) hh = (u32)(strstart - 5 + k) + 1u;
            }
        }
        run_n = 0;
        if (lookahead >= KD_MIN_MATCH) { prev[strstart & (W - 1)] = hh; head[h] = (u32)strstart + 1u; }
        int match_length = KD_MIN_MATCH - 1, match_start = 0;
        if (hh != 0 && (int)(hh - 1u) > base && strstart - (int)(hh - 1u) <= MD) {
            // longest_match with prev_length = 2 (deflate_fast never changes it, so good_match never shortens the chain)
            int const maxlen = lookahead < KD_MAX_MATCH ? lookahead : KD_MAX_MATCH;
            int const nice = lookahead < niceMax ? lookahead : niceMax;
            int const limit = (strstart - MD > base) ? strstart - MD : base;
            int best_len = 2, chain = maxChain; int cur = (int)(hh - 1u);
            u32 scanEnd = (w >> 8) & 0xFFFFu;                   // scan[best_len - 1], scan[best_len]
            for (;;) {
                const u8* const m = src + cur;
                u32 nx = 0;                                     // the next link, fetched beside the candidate's bytes (not behind the last step)
                if (chain > 1) nx = prev[cur & (W - 1)];
                if (kx_ld16(m + best_len - 1) == scanEnd && kx_ld16(m) == (w & 0xFFFFu)) {
                    int len = 2;                                // (byte 2 is equal when bytes 0, 1 and the hash are)
                    while (len < maxlen) {
                        if (strstart + len + 8 <= n) {
                            u64 const d = kx_ld64(m + len) ^ kx_ld64(src + strstart + len);
                            if (d) { len += (int)(kx_ctz64(d) >> 3); break; }
                            len += 8;
                        } else { if (m[len] != src[strstart + len]) break; len++; }
                    }
                    if (len > maxlen) len = maxlen;
                    if (len > best_len) {
                        match_start = cur; best_len = len;
                        if (len >= nice) break;
                        scanEnd = kx_ld16(src + strstart + best_len - 1);
                    }
                }
                if (nx == 0 || (int)(nx - 1u) <= limit || --chain == 0) break;
                cur = (int)(nx - 1u);
            }
            match_length = best_len;                            // (best_len <= maxlen <= lookahead)
        }
        if (match_length >= KD_MIN_MATCH) {
            KDF_SYM((u32)(strstart - match_start) | ((u32)(match_length - KD_MIN_MATCH) << 16))
            int const after = strstart + match_length;
            if (match_length <= maxInsert && dataEnd - after >= KD_MIN_MATCH) run_n = match_length - 1;
            strstart = after;
            if (blockSyms == LB - 1u) KDF_FLUSH(0, after)
        } else {
            KDF_SYM((w & 0xFFu) << 16)
            strstart++;
            if (blockSyms == LB - 1u) KDF_FLUSH(0, strstart)
        }
    }
    { u32 const k = nsym & 3u; u32* const q = syms + (nsym - k);                // the open group
      if (k > 0) q[0] = sq0; if (k > 1) q[1] = sq1; if (k > 2) q[2] = sq2; }
    KDF_FLUSH(1, strstart)
    mm.nsym = nsym;
    a.meta[slice] = mm;
#undef KDF_FLUSH
#undef KDF_SYM
}
