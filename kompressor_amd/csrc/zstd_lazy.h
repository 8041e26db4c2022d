// zstd_lazy.h -- zstd levels 5 .. 10 (and 4 .. 8 for slices up to 16 KiB): libzstd's strategies "greedy", "lazy", "lazy2"
// (zstd_lazy.c ZSTD_compressBlock_lazy_generic, depth 0 / 1 / 2) for slices of one block (<= 128 KiB), no dictionary.
//
// What libzstd does: at a position it asks a match finder for the longest match among the newest earlier positions that hash like
// it -- the row-based finder (window above 2^14: rows of 16 / 32 / 64 slots, one of them the row's head counter, entries tagged with 8
// more hash bits, 1 << min(searchLog, rowLog) tag hits looked at) or the hash-chain finder (windows up to 2^14: 1 << searchLog chain
// steps) --, prefers a repeat offset when that pays, and with depth 1 / 2 looks one / two positions further before it commits.
// Both finders insert EVERY position they pass, in order -- so, as in zlib (deflate_lazy.h), a position's candidates do not depend on
// the parse: they are the run of earlier positions in its hash bucket.  Two exceptions, both ranges of positions that never enter
// the tables: "lazy skipping" (after 2 KiB without a match the parser strides, and only searched positions are inserted) and, for the
// row finder, the middle of a gap of more than 384 positions behind a long match (its first 96 and last 32 go in).  A bitmap in LDS
// holds them.
//
// Two kernels over a workspace of 20 bytes a position:
//   k_zstd_lazy_sort  256 threads a slice: stable counting sort of the positions by bucket (row index, or chain hash) -> rec[] (position |
//                     tag << 24 and the 12 bytes there, bucket by bucket, ascending), wr[p] (where p stands | its rank in the bucket << 18)
//   k_zstd_lazy       a wave per slice walks the parse; a search is one wave-wide step per 64 candidates
// then k_zstd_entropy (literals gathered there; the sequence tables chosen by price from strategy "lazy" on).
#pragma once
#include "zstd_common.h"

// one sorted position: the position and its tag, and the 12 bytes there -- everything a search needs of a candidate comes with one 16-byte
// load (the searches are bound by memory round trips: counters in profiles/r04_lazy_levels.txt)
struct alignas(16) KLazyRec { u32 pt; u32 b0; u64 b4; };      // pt = position | tag << 24; b0 = bytes 0 .. 3; b4 = bytes 4 .. 11

struct KLazyArgs {
    const u8* src; const u64* in_off; const u32* in_len; u32 n_slices;
    KLazyRec* rec; u32* wr; u32 pos_cap;              // per slice: pos_cap entries each
    KSeq* seqs; u32 seq_cap; KSliceMeta* meta;
    u32 level;
    u32* order_key = nullptr; u32* order_hist = nullptr;     // the sort files each slice under a cost class (occupied buckets / 64, at most 255) ...
    const u32* order = nullptr;                              // ... and the parse takes the slices in this order: the costliest first (null: as they come)
};

// ZSTD_getCParams(level, n, 0) where it is one of the three strategies; strat 0: not served at this size
struct KLazyPar { u32 W, H, S, strat, rows, rowLog, buckLog; };
KX_DEV KLazyPar kx_lazy_params(u32 level, u32 n)
{
    KLazyPar p; p.W = 0; p.H = 0; p.S = 0; p.strat = 0; p.rows = 0; p.rowLog = 0; p.buckLog = 0;
    if (n > 131072u) return p;
    u32 tW, H;
    if (n < 8u) { p.strat = 3; p.W = 10; p.H = 11; p.S = 3; p.rowLog = 4; p.buckLog = 11; return p; }     // (nothing to parse: a raw block at every level)
    if (n <= 16384u) {
        if (level < 4 || level > 8) return p;
        tW = 14; H = 14; p.S = level == 4 ? 4u : level == 5 ? 3u : level == 6 ? 4u : level == 7 ? 6u : 8u; p.strat = level == 4 ? 3u : level == 5 ? 4u : 5u;
    } else {
        if (level < 5 || level > 10) return p;
        tW = 17; H = 17; p.S = level <= 7 ? 3u : level - 4u; p.strat = level == 5 ? 3u : level == 6 ? 4u : 5u;
    }
    u32 const srcLog = (n < 64u) ? 6u : kx_hb32(n - 1u) + 1u;
    p.W = tW < srcLog ? tW : srcLog;
    if (H > p.W + 1) H = p.W + 1;
    if (p.W < 10) p.W = 10;
    p.H = H;
    p.rows = p.W > 14 ? 1u : 0u;
    p.rowLog = p.S < 4 ? 4u : p.S > 6 ? 6u : p.S;
    p.buckLog = p.rows ? H - p.rowLog : H;             // row index bits / chain hash bits
    return p;
}
// 4 bytes at position p of a slice of n bytes, bytes past the end read as zero
KX_DEV u32 kx_ld32_clamped(const u8* src, u32 p, u32 n)
{
    if (p + 4u <= n) return kx_ld32(src + p);
    u32 v = 0;
    for (u32 k = 0; p + k < n; k++) v |= (u32)src[p + k] << (8u * k);
    return v;
}
// minMatch is 4 at all these levels: ZSTD_hash4 of the four bytes, hBits wide
KX_DEV u32 kx_lazy_hash4(u32 v, u32 hBits) { return (v * 2654435761u) >> (32u - hBits); }

// ---------------------------------------------------------------------------
// k_zstd_lazy_sort (the scheme of k_deflate_sort: ranks in position order, bucket scan, scatter)
// ---------------------------------------------------------------------------
#define KZL_MAXBUCK 32768u
KX_DEV void zstd_lazy_sort_body(const KLazyArgs& a)
{
    KX_SHARED u32 cnt[KZL_MAXBUCK / 2u];          // bucket sizes, then bucket starts: 8 192 rows of 32 bits, or (chains: slices <= 16 KiB) 32 768 of 16
    KX_SHARED u32 part[256];
    u16* const cnt16 = (u16*)cnt;
    int const lane = kx_lane(); int const wv = kx_wave(); int const nw = kx_nwaves(); int const tid = wv * 64 + lane; int const nthreads = nw * 64;
    for (u32 it = kx_block(); it < a.n_slices; it += kx_nblocks()) {
        u32 const slice = kx_xcd_chunk(it, a.n_slices);
        const u8* const src = a.src + a.in_off[slice]; u32 const n = a.in_len[slice];
        KLazyPar const P = kx_lazy_params(a.level, n);
        if (P.strat == 0 || n < 8u) { if (a.order_key && tid == 0) { a.order_key[slice] = 0; kx_atomic_add(a.order_hist, 1u); } continue; }      // (uniform over the workgroup)
        u32* const wr = a.wr + (size_t)slice * a.pos_cap; KLazyRec* const rec = a.rec + (size_t)slice * a.pos_cap;
        u32 const nb = 1u << P.buckLog; bool const wide = P.rows != 0;            // wide: 32-bit counters
        u32 const hBits = P.rows ? P.buckLog + 8u : P.buckLog;
        for (u32 i = (u32)tid; i < KZL_MAXBUCK / 2u; i += (u32)nthreads) cnt[i] = 0;
        kx_block_sync();
        u32 const nIns = n - 7u;                 // positions 0 .. n - 8: the parse stops 8 (rows: 16) bytes before the end, nothing later is ever inserted
#define KZL_CNT(h_) (wide ? cnt[h_] : (u32)cnt16[h_])
#define KZL_SET(h_, v_) { if (wide) cnt[h_] = (v_); else cnt16[h_] = (u16)(v_); }
        // ---- pass 1: rank[p] (kept in wr[p]) = how many earlier positions share p's bucket; the waves take turns in position order
        for (u32 base = 0; base < nIns; base += (u32)nthreads) {
            u32 const p = base + (u32)wv * 64u + (u32)lane; bool const valid = p < nIns;
            u32 const h = valid ? (kx_lazy_hash4(kx_ld32(src + p), hBits) >> (P.rows ? 8u : 0u)) : 0u;
            u32 rk = 0;
            for (int w = 0; w < nw; w++) {
                if (wv == w) {
                    u32 const old = valid ? KZL_CNT(h) : 0u;
                    u32 below = 0, group = 1;
                    // lanes of this wave in one bucket: a lane's rank counts the lower ones, the highest one leaves the total.  (A lane that is
                    // alone in its bucket -- most are -- reads its own mark back and takes no turn; one that reads its own mark although lower
                    // lanes share the bucket gets its numbers in their turn: the ballot is over all valid lanes of the bucket.)
                    kx_lockstep();
                    if (valid) KZL_SET(h, (u32)lane)
                    kx_lockstep();
                    bool const shared = valid && KZL_CNT(h) != (u32)lane;
                    for (u64 todo = kx_ballot(shared); todo; ) {
                        int const L = (int)kx_ctz64(todo);
                        u32 const hL = kx_bcast(h, L);
                        u64 const grp = kx_ballot(valid && h == hL);
                        if (valid && h == hL) { below = kx_popc64(grp & ((1ull << lane) - 1ull)); group = kx_popc64(grp); }
                        todo &= ~grp;
                    }
                    rk = old + below;
                    kx_lockstep();
                    if (valid && below + 1u == group) KZL_SET(h, old + group)
                    kx_lockstep();
                }
                if (nw > 1) kx_block_sync();
            }
            if (valid) wr[p] = rk;
        }
        kx_block_sync();
        // ---- pass 2: bucket starts
        {
            u32 const per = nb / (u32)nthreads ? nb / (u32)nthreads : 1u;            // (nb >= 256 at every served size? no: small slices have few buckets)
            u32 const first = (u32)tid * per; u32 s = 0, occ = 0;
            for (u32 i = 0; i < per && first + i < nb; i++) { u32 const v = KZL_CNT(first + i); s += v; occ += v ? 1u : 0u; }
            // how many buckets the slice uses says how varied its bytes are: a stand-in for the number of searches the parse will make
            // (text and binary slices: thousands; sparse ones: tens), by which the parse kernel orders its slices -- the launch does not
            // end with a few waves walking its costliest slices alone
            if (a.order_key) {
                part[tid] = occ;
                kx_block_sync();
                if (tid == 0) { u32 tot = 0; for (int t = 0; t < nthreads; t++) tot += part[t]; u32 k = tot >> 5; if (k > 255u) k = 255u; a.order_key[slice] = k; kx_atomic_add(a.order_hist + k, 1u); }
                kx_block_sync();
            }
            part[tid] = s;
            kx_block_sync();
            if (tid == 0) { u32 run = 0; for (int t = 0; t < nthreads; t++) { u32 const v = part[t]; part[t] = run; run += v; } }
            kx_block_sync();
            u32 run = part[tid];
            for (u32 i = 0; i < per && first + i < nb; i++) { u32 const v = KZL_CNT(first + i); KZL_SET(first + i, run) run += v; }
        }
        kx_block_sync();
        // ---- pass 3: scatter
        for (u32 p = (u32)tid; p < nIns; p += (u32)nthreads) {
            u64 const w8 = kx_ld64(src + p);                    // (p + 8 <= n)
            u32 const hh = kx_lazy_hash4((u32)w8, hBits);
            u32 const h = P.rows ? hh >> 8 : hh, tag = P.rows ? hh & 0xFFu : 0u;
            u32 const rk = wr[p];
            u32 const where = KZL_CNT(h) + rk;
            wr[p] = where | ((rk > 16383u ? 16383u : rk) << 18);
            KLazyRec r; r.pt = p | (tag << 24); r.b0 = (u32)w8; r.b4 = (w8 >> 32) | ((u64)kx_ld32_clamped(src, p + 8u, n) << 32);
            rec[where] = r;
        }
        kx_block_sync();
#undef KZL_CNT
#undef KZL_SET
    }
}

// ---------------------------------------------------------------------------
// k_zstd_lazy: one wave per slice
// ---------------------------------------------------------------------------
// one bit per position: 1 = it entered (or will enter) the finder's tables.  WORDS: 2 048 for contexts of slices up to 64 KiB (8 KiB of LDS a
// wave: twenty waves a CU), 4 096 up to 128 KiB
template <int WORDS> struct KLazyLds { u32 ins[WORDS]; };

// length of the common prefix of src[a ..) and src[b ..) (b < a), not past n: the whole wave compares 512 bytes a step
KX_DEV u32 kzl_count_wave(const u8* src, u32 a, u32 b, u32 n, int lane)
{
    u32 len = 0;
    for (;;) {
        u32 const o = len + 8u * (u32)lane;
        u32 eq = 8;                                  // bytes of my 8 that agree (positions past the end disagree)
        if (a + o >= n) eq = 0;
        else {
            u64 const x = kx_ld64_clamped(src, (int)(a + o), (int)n) ^ kx_ld64_clamped(src, (int)(b + o), (int)n);
            u32 const room = n - (a + o) < 8u ? n - (a + o) : 8u;
            u32 const same = x ? (u32)(kx_ctz64(x) >> 3) : 8u;
            eq = same < room ? same : room;
        }
        u64 const stop = kx_ballot(eq < 8u);
        if (stop) { int const L = (int)kx_ctz64(stop); return len + 8u * (u32)L + kx_bcast(eq, L); }
        len += 512u;
    }
}

template <int WORDS>
KX_DEV void zstd_lazy_slice(const KLazyArgs& a, KLazyLds<WORDS>& lds, u32 slice, int lane)
{
    const u8* const src = a.src + a.in_off[slice]; u32 const n = a.in_len[slice];
    KLazyPar P = kx_lazy_params(a.level, n);
    if (n > 32u * (u32)WORDS) P.strat = 0;                                     // (cannot happen: the host picks WORDS by the context's slice size)
    KSeq* const seqs = a.seqs + (size_t)slice * a.seq_cap;
    KSliceMeta mm; mm.nbSeq = 0; mm.litSize = 0; mm.lastLL = n; mm.longType = 0; mm.longPos = 0; mm.status = 0; mm.pad[0] = 0; mm.pad[1] = 0;
    // another strategy at this size: at level 4 the double-fast kernel has served it (or refused it) and its record stands; else "not served"
    // (k_len_guard_finish voids the frame: KMP_STATUS_LEVEL_SIZE)
    if (P.strat == 0) { if (a.level != 4u) { mm.status = 3; if (lane == 0) a.meta[slice] = mm; } return; }
    if (n < 8u) { if (lane == 0) a.meta[slice] = mm; return; }
    const u32* const wr = a.wr + (size_t)slice * a.pos_cap; const KLazyRec* const rec = a.rec + (size_t)slice * a.pos_cap;
    u32 const depth = P.strat - 3u;
    bool const rows = P.rows != 0;
    u32 const rowCap = (1u << P.rowLog) - 1u;                                  // entries a row holds (one slot is its head counter)
    u32 const nbAttempts = 1u << (rows ? (P.S < P.rowLog ? P.S : P.rowLog) : P.S);
    u32 const hBits = rows ? P.buckLog + 8u : P.buckLog;
    u32 const ilimit = rows ? n - 16u : n - 8u;                                // (n >= 8; rows: n > 16 384)
    for (u32 i = (u32)lane; i < (u32)WORDS; i += 64u) lds.ins[i] = 0xFFFFFFFFu;
    kx_sync();
    u32 ip = 1, anchor = 0, off1 = 1, off2 = 4, saved1 = 0, saved2 = 0;        // first block: position 0 is only ever a match source
    if (off2 > 1u) { saved2 = off2; off2 = 0; }                                // maxRep = 1: repeat offset 4 waits
    u32 ntu = 0; bool skipping = false;                                        // ms->nextToUpdate (a position), ms->lazySkipping
    u32 nseq = 0, nlit = 0, longType = 0, longPos = 0;
    u32 guard = 0;

    // The parse is bound by memory round trips (counters: profiles/r04_lazy_levels.txt), so every step asks for all it can know the addresses
    // of in ONE batch of loads: a search's own 12 bytes and its candidates' records, and with them the four bytes a repeat offset back that
    // the step's repeat-offset test needs (`x`).  LOAD has no side effects (a greedy step that takes the repeat offset never searches);
    // EVAL marks what the finder would have inserted and compares.
    struct Loaded { u64 scan; u32 scanHi, where, rk, x; KLazyRec e; };
    int wbase = -(1 << 20); u32 wrv = 0;       // where / rank of the 64 positions from wbase on (lane i: wr[wbase + i])
    auto clear_bits = [&](u32 lo, u32 hi) {    // positions [lo, hi) never enter the tables
        if (lo < hi) {
            for (u32 w = (lo >> 5) + (u32)lane; w <= ((hi - 1u) >> 5); w += 64u) {
                u32 m = 0xFFFFFFFFu;
                if (w == (lo >> 5)) m &= 0xFFFFFFFFu << (lo & 31u);
                if (w == ((hi - 1u) >> 5)) m &= 0xFFFFFFFFu >> (31u - ((hi - 1u) & 31u));
                lds.ins[w] &= ~m;
            }
            kx_sync();
        }
    };
    auto load = [&](u32 cur, u32 xaddr, Loaded& L) {
        if ((int)cur < wbase || (int)cur >= wbase + 64) { wbase = (int)cur; wrv = (cur + (u32)lane < n - 7u) ? wr[cur + (u32)lane] : 0u; }
        u32 const w0 = kx_bcast(wrv, (int)cur - wbase);
        L.where = w0 & 0x3FFFFu; L.rk = w0 >> 18;
        L.e.pt = 0; L.e.b0 = 0; L.e.b4 = 0;
        if ((u32)lane < L.where && (L.rk == 16383u || (u32)lane < L.rk)) L.e = rec[L.where - 1u - (u32)lane];
        L.scan = kx_ld64(src + cur); L.scanHi = kx_ld32_clamped(src, cur + 8u, n);
        L.x = kx_ld32(src + xaddr);
    };
    // the finder's answer for position cur: ml (3 = nothing), off (the distance)
    auto eval = [&](u32 cur, const Loaded& L, u32& mlOut, u32& offOut) {
        if (rows) { if (!skipping) { if (cur - ntu > 384u) clear_bits(ntu + 96u, cur - 32u); } else clear_bits(ntu, cur); ntu = cur + 1u; }
        else { if (skipping && ntu < cur) clear_bits(ntu + 1u, cur); ntu = cur; }
        u32 const where = L.where, rk = L.rk; u64 const scan = L.scan;
        u32 const hsh = kx_lazy_hash4((u32)scan, hBits); u32 const tag = rows ? hsh & 0xFFu : 0u;
        u32 best = 3, bestPos = 0, insSeen = 0, attSeen = 0;
        for (u32 cb = 0; cb < rk || (rk == 16383u && cb < where); cb += 64u) {
            u32 const j = cb + (u32)lane;
            bool v = j < where && (rk == 16383u || j < rk);
            KLazyRec e = L.e;
            if (cb != 0) { e.pt = 0; e.b0 = 0; e.b4 = 0; if (v) e = rec[where - 1u - j]; }
            u32 const cp = e.pt & 0xFFFFFFu;
            if (rk == 16383u && v) { u32 const ch = kx_lazy_hash4(e.b0, hBits); v = rows ? (ch >> 8) == (hsh >> 8) : ch == hsh; }
            u64 const outside = (rk == 16383u) ? kx_ballot(j < where && !v) : 0ull;      // (a capped rank: the bucket ends where another hash starts)
            if (outside) { int const Lo = (int)kx_ctz64(outside); if (lane >= Lo) v = false; }
            bool const in = v && ((lds.ins[cp >> 5] >> (cp & 31u)) & 1u);
            u64 const inM = kx_ballot(in);
            bool const inRow = in && (!rows || insSeen + kx_popc64(inM & ((1ull << lane) - 1ull)) < rowCap);
            bool const hit = inRow && (!rows || (e.pt >> 24) == tag);
            u64 const hitM = kx_ballot(hit);
            bool const cand = hit && attSeen + kx_popc64(hitM & ((1ull << lane) - 1ull)) < nbAttempts;
            u32 len = 0;
            if (cand) {
                u32 const d0 = e.b0 ^ (u32)scan;
                if (d0) len = (u32)(kx_ctz32(d0) >> 3);
                else {
                    u64 const d4 = e.b4 ^ ((scan >> 32) | ((u64)L.scanHi << 32));
                    if (d4) len = 4u + (u32)(kx_ctz64(d4) >> 3);
                    else {
                        len = 12;
                        for (;;) {
                            if (cur + len >= n) break;
                            u64 const x = kx_ld64_clamped(src, (int)(cur + len), (int)n) ^ kx_ld64_clamped(src, (int)(cp + len), (int)n);
                            if (x) { len += (u32)(kx_ctz64(x) >> 3); break; }
                            len += 8u;
                        }
                    }
                }
                if (len > n - cur) len = n - cur;
            }
            // the longest wins, the newer one among equals: lanes are in order of age, blocks too.  (Few candidates -- 8 up to level 7 -- are
            // looked at one by one through the scalar unit; many by a butterfly of shuffles.)
            if (nbAttempts <= 16u) {
                for (u64 c = kx_ballot(cand && len > 3u); c; c &= c - 1) { int const Lc = (int)kx_ctz64(c); u32 const l = kx_bcast(len, Lc); if (l > best) { best = l; bestPos = kx_bcast(cp, Lc); } }
            } else {
                u32 m = len;
                for (int o = 32; o >= 1; o >>= 1) { u32 const t = kx_shfl(m, lane ^ o); m = t > m ? t : m; }
                if (m > best) { u64 const who = kx_ballot(cand && len == m); best = m; bestPos = kx_bcast(cp, (int)kx_ctz64(who)); }
            }
            insSeen += kx_popc64(inM); attSeen += kx_popc64(hitM);
            if (outside || (rows && insSeen >= rowCap) || attSeen >= nbAttempts) break;
        }
        mlOut = best; offOut = cur - bestPos;
    };
    auto store = [&](u32 ll, u32 offBase, u32 ml) {
        u32 const mlb = ml - 3u;
        if (ll > 0xFFFFu) { longType = 1; longPos = nseq; }
        if (mlb > 0xFFFFu) { longType = 2; longPos = nseq; }
        if (lane == 0 && nseq < a.seq_cap) { KSeq q; q.offBase = offBase; q.litLength = (u16)ll; q.mlBase = (u16)mlb; seqs[nseq] = q; }
        nseq++; nlit += ll;
    };

    Loaded L;
    while (ip < ilimit) {
        if (++guard > 400000u) { mm.status = 2; break; }
        u32 matchLength = 0, offBase = 1, start = ip + 1u;
        bool stored = false;
        // the search's loads, and with them the bytes a repeat offset before ip + 1 (off1 == 0: any address)
        load(ip, off1 ? ip + 1u - off1 : ip, L);
        if (off1 > 0 && L.x == (u32)(L.scan >> 8)) {                               // (bytes 1 .. 4 of what lies at ip)
            matchLength = kzl_count_wave(src, ip + 1u + 4u, ip + 1u + 4u - off1, n, lane) + 4u;
            if (depth == 0) stored = true;
        }
        if (!stored) {
            u32 ml2, of2; eval(ip, L, ml2, of2);
            if (ml2 > matchLength) { matchLength = ml2; start = ip; offBase = of2 + 3u; }
            if (matchLength < 4u) {
                u32 const step = ((ip - anchor) >> 8) + 1u;               // kSearchStrength
                ip += step;
                skipping = step > 8u;                                     // kLazySkippingStep
                continue;
            }
            // one / two positions further: something better?
            if (depth >= 1u)
            while (ip < ilimit) {
                ip++;
                load(ip, off1 ? ip - off1 : ip, L);
                if (off1 > 0 && L.x == (u32)L.scan) {
                    u32 const mlRep = kzl_count_wave(src, ip + 4u, ip + 4u - off1, n, lane) + 4u;
                    int const gain2 = (int)(mlRep * 3u), gain1 = (int)(matchLength * 3u - kx_hb32(offBase) + 1u);
                    if (mlRep >= 4u && gain2 > gain1) { matchLength = mlRep; offBase = 1; start = ip; }
                }
                {
                    u32 ml3, of3; eval(ip, L, ml3, of3);
                    int const gain2 = (int)(ml3 * 4u - kx_hb32(ml3 > 3u ? of3 + 3u : 999999999u)), gain1 = (int)(matchLength * 4u - kx_hb32(offBase) + 4u);
                    if (ml3 >= 4u && gain2 > gain1) { matchLength = ml3; offBase = of3 + 3u; start = ip; continue; }
                }
                if (depth == 2u && ip < ilimit) {
                    ip++;
                    load(ip, off1 ? ip - off1 : ip, L);
                    if (off1 > 0 && L.x == (u32)L.scan) {
                        u32 const mlRep = kzl_count_wave(src, ip + 4u, ip + 4u - off1, n, lane) + 4u;
                        int const gain2 = (int)(mlRep * 4u), gain1 = (int)(matchLength * 4u - kx_hb32(offBase) + 1u);
                        if (mlRep >= 4u && gain2 > gain1) { matchLength = mlRep; offBase = 1; start = ip; }
                    }
                    {
                        u32 ml3, of3; eval(ip, L, ml3, of3);
                        int const gain2 = (int)(ml3 * 4u - kx_hb32(ml3 > 3u ? of3 + 3u : 999999999u)), gain1 = (int)(matchLength * 4u - kx_hb32(offBase) + 7u);
                        if (ml3 >= 4u && gain2 > gain1) { matchLength = ml3; offBase = of3 + 3u; start = ip; continue; }
                    }
                }
                break;
            }
        }
        // where the parse goes on, and the repeat offset that is tried there first: both known before the catch-up (which moves the match's
        // start and its length by the same amount), so its four bytes are asked for together with the catch-up's
        u32 const nextIp = start + matchLength;
        u32 const nOff2 = (!stored && offBase > 3u) ? off1 : off2;
        u32 repNow = 0, repBack = 1;
        if (nextIp <= ilimit && nOff2 > 0) { repNow = kx_ld32(src + nextIp); repBack = kx_ld32(src + nextIp - nOff2); }
        if (!stored && offBase > 3u) {
            // catch up: bytes before the match that agree too (not before the anchor, not before the slice's first byte)
            u32 const off = offBase - 3u;
            u32 const maxBack = (start - anchor) < (start - off) ? (start - anchor) : (start - off);
            u32 back = 0;
            for (u32 done = 0; done < maxBack; done += 64u) {
                u32 const k = done + (u32)lane;
                bool const ne = k >= maxBack || src[start - 1u - k] != src[start - off - 1u - k];
                u64 const stop = kx_ballot(ne);
                if (stop) { back = done + (u32)kx_ctz64(stop); break; }
                back = done + 64u;
            }
            if (back > maxBack) back = maxBack;
            start -= back; matchLength += back;
            off2 = off1; off1 = off;
        }
        store(start - anchor, offBase, matchLength);
        anchor = ip = nextIp;
        skipping = false;
        // repeat offset 2 right behind the match
        bool first = true;
        while (ip <= ilimit && off2 > 0) {
            bool const same = first ? (repNow == repBack) : (kx_ld32(src + ip) == kx_ld32(src + ip - off2));
            first = false;
            if (!same) break;
            u32 const ml = kzl_count_wave(src, ip + 4u, ip + 4u - off2, n, lane) + 4u;
            { u32 const t = off2; off2 = off1; off1 = t; }
            store(0u, 1u, ml);
            ip += ml; anchor = ip;
        }
    }
    (void)saved1; (void)saved2;
    mm.nbSeq = nseq; mm.litSize = nlit; mm.lastLL = n - anchor; mm.longType = longType; mm.longPos = longPos;
    if (nseq > a.seq_cap) mm.status = 2;
    if (lane == 0) a.meta[slice] = mm;
}

template <int WORDS>
KX_DEV void zstd_lazy_body(const KLazyArgs& a)
{
    KX_SHARED KLazyLds<WORDS> lds;
    int const lane = kx_lane();
    for (u32 it = kx_block(); it < a.n_slices; it += kx_nblocks()) {
        u32 const slice = a.order ? a.order[it] : kx_xcd_chunk(it, a.n_slices);
        zstd_lazy_slice(a, lds, slice, lane);
        kx_sync();
    }
}
