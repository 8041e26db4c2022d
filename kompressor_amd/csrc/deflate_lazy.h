// deflate_lazy.h -- zlib's lazy levels (4 .. 9: "deflate_slow") for slices up to 64 KiB as TWO kernels that only do the work
// zlib does (round 4).  Replaces k_deflate_chains + k_deflate_best + k_deflate_parse of deflate_match.h for such slices.
//
// Why.  k_deflate_best walked the whole hash chain of EVERY position (a lane per position, 25 - 85 dependent LDS steps
// each); zlib's lazy parse asks longest_match for one position in 5.6 -- the positions inside an accepted match are
// inserted into the chains but never searched.  Which positions are searched is the parse's serial decision chain, so the
// search has to be driven BY the parse; what makes that affordable on a GPU is that a search need not be serial:
//   * zlib inserts every position into its chains, so the candidates of a position do not depend on the parse: they are the
//     earlier positions with the same 3-byte hash, nearest first.  k_deflate_sort lays the positions of a slice out SORTED
//     BY HASH (a stable counting sort: srt[]), and records for every position where it sits (where) and how many same-hash
//     positions precede it (rank): the candidates of p are srt[where - 1], srt[where - 2], ... -- contiguous in memory.
//   * longest_match's result does not depend on the order effects of its walk: it is the first candidate (nearest first)
//     that reaches nice_length if there is one within max_chain steps, else the longest candidate, the nearest among equals
//     (best_len only ever grows; a candidate is skipped when it cannot beat it).  So k_deflate_lazy, one WAVE per slice,
//     measures 64 candidates of the position at once -- a coalesced load of their positions, one 8-byte compare per lane,
//     more only for the lanes that still match -- and picks the winner with two ballots.  A search is two dependent memory
//     round trips instead of a chain of up to 128, and only the positions the parse really asks about are searched.
// The parse itself (fill_window and its slides, lazy evaluation, TOO_FAR, the 16 383-symbol blocks, stored-block
// eligibility) is deflate_parse_body's, run wave-uniformly; the symbols leave 64 at a time.  Output: the symbol list, block
// list and slice record k_deflate_encode takes, as before.  Reference: the call at kompressor-zlib--nativelib/src/
// jvmCommonMain/jni/Wrapper.cpp:73 (deflate) under deflateInit2(level, 8, -15, 8, 0) (:20).
#pragma once
#include "deflate_match.h"

// The per-slice arrays of the two kernels: srt (the positions in hash order, 2 bytes each) where the chain links lived; sb (the
// first 16 bytes of every position, in the same order: a search reads its candidates' bytes as ONE contiguous run, and most
// candidates are measured without a second trip to memory -- a kernel that waits for memory three quarters of its cycles lives on
// the number of dependent round trips per search, not on bytes) where KdBest lived (twice its size); wr (where | rank << 16,
// 4 bytes per position) in an array of its own; the ranks of the first pass sit in the symbol array until the parse fills it.
struct alignas(16) KdlBytes { u64 lo, hi; };
// (SEG: the arrays of a slice hold one 64 KiB span whatever the slice's length -- see deflate_sort_body; the symbols keep pos_cap)
template <bool SEG> KX_DEV size_t kdl_stride(const KdArgs& a) { return SEG ? (size_t)65536u : (size_t)a.pos_cap; }
template <bool SEG> KX_DEV u32* kdl_wr(const KdArgs& a, u32 slice) { return a.wr + (size_t)slice * kdl_stride<SEG>(a); }
template <bool SEG> KX_DEV u16* kdl_srt(const KdArgs& a, u32 slice) { return a.link + (size_t)slice * kdl_stride<SEG>(a); }
template <bool SEG> KX_DEV KdlBytes* kdl_sb(const KdArgs& a, u32 slice) { return (KdlBytes*)(a.best + (size_t)slice * kdl_stride<SEG>(a) * 2u); }
template <bool SEG> KX_DEV u16* kdl_rank(const KdArgs& a, u32 slice) { return SEG ? a.seg_rank + (size_t)slice * 65536u : (u16*)(a.syms + (size_t)slice * a.pos_cap); }

// 8 bytes at position q of a slice of n bytes, bytes past the end read as zero (q < n)
KX_DEV u64 kdl_ld64(const u8* src, int q, int n)
{
    if (q + 8 <= n) return kx_ld64(src + q);
    u64 v = 0;
    for (int k = 0; q + k < n; k++) v |= (u64)src[q + k] << (8 * k);
    return v;
}
// (a branch-free form -- one load that ends at the slice's last byte, then a 64-bit shift -- was measured on one box against this one:
// the parse kernel 151 ms per 16 384 text slices against 129, 721 ms per 65 536 mixed slices against 627: the shifts cost more than the branch)
template <bool TINY> KX_DEV u64 kdl_get64(const u8* src, int q, int n) { return q < n ? kdl_ld64(src, q, n) : 0ull; }

// ---------------------------------------------------------------------------
// k_deflate_sort: 256 threads per workgroup, one slice at a time, cnt[32768] in LDS
// ---------------------------------------------------------------------------
// Pass 1, in position order (the waves of the workgroup take turns on the table, as in k_deflate_chains): rank[p] = how many
// earlier positions share p's hash.  Pass 2: exclusive scan of the bucket sizes.  Pass 3, any order: where[p] = start of the
// bucket + rank[p]; srt[where[p]] = p.
// SEG (slices above 64 KiB, late round 4): a long slice goes through in SEGMENTS.  A candidate lies less than 32 KiB back, so the
// positions of [32 768 (k + 1), 32 768 (k + 2)) find all of theirs in the 64 KiB span that starts at 32 768 k: launch k of this kernel
// sorts that span of every slice (positions relative to its start, so the arrays and their 16-bit entries are those of a 64 KiB slice,
// whatever the slice's length), launch k of k_deflate_lazy parses those positions (segment 0: the whole first span) and leaves its
// state for launch k + 1.  Every slice of a piece is in flight in every launch; the workspace no longer grows with the slice.
#define KDL_SEG_STEP 32768u
#define KDL_SEG_SPAN 65536u
#define KDL_STATE_WORDS 12u
template <int HB = 15, bool SEG = false>      // hash bits the table has room for (memLevel 9: 16)
KX_DEV void deflate_sort_body(const KdArgs& a)
{
    KX_SHARED u16 cnt[1 << HB];                     // bucket sizes, then bucket starts (a slice has at most 65 534 chained positions)
    KX_SHARED u32 part[256];
    KX_SHARED u32 occp[256];
    int const lane = kx_lane(); int const wv = kx_wave(); int const nw = kx_nwaves(); int const tid = wv * 64 + lane; int const nthreads = nw * 64;
    for (u32 it = kx_block(); it < a.n_slices; it += kx_nblocks()) {
        u32 const slice = kx_xcd_chunk(it, a.n_slices);
        u32 const s0 = SEG ? a.seg * KDL_SEG_STEP : 0u;
        if (SEG && a.seg > 0 && a.in_len[slice] <= s0 + KDL_SEG_STEP) continue;        // this slice was finished by an earlier segment (uniform over the workgroup)
        const u8* const src = a.src + a.in_off[slice] + s0; u32 const n = a.in_len[slice] - s0;      // (n: the bytes from the span's start to the slice's end)
        u16* const rank = kdl_rank<SEG>(a, slice);
        u32* const wr = kdl_wr<SEG>(a, slice); u16* const srt = kdl_srt<SEG>(a, slice); KdlBytes* const sb = kdl_sb<SEG>(a, slice);
        int const hsize = (int)a.hmask + 1;
        for (int i = tid; i < hsize; i += nthreads) cnt[i] = 0;
        kx_block_sync();
        u32 nIns = n >= 3 ? n - 2 : 0;                      // positions 0 .. n-3 enter the chains, in order
        if (SEG && nIns > KDL_SEG_SPAN) nIns = KDL_SEG_SPAN; // (... those of the span; their bytes may lie beyond it)
        // ---- pass 1 (the source bytes of a group of four rounds are fetched a group ahead, as in k_deflate_chains)
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
                u32 const p = base + (u32)wv * 64u + (u32)lane; bool const valid = vq[k];
                u32 const h = valid ? kd_hash3(a, hq[k] & 0xFFu, (hq[k] >> 8) & 0xFFu, (hq[k] >> 16) & 0xFFu) : 0x10000u + (u32)lane;
                u32 rk = 0;
                for (int w = 0; w < nw; w++) {
                    if (wv == w) {
                        u32 const old = valid ? (u32)cnt[h] : 0u;
                        // lanes of this wave that share a bucket: a lane's rank counts the lower ones, the highest one leaves the total
                        u32 below = 0, group = 1;
                        {
                            // (one bucket per round, for the lanes that are not alone in theirs)
                            kx_lockstep();
                            if (valid) cnt[h] = (u16)lane;           // a mark: who is alone reads its own lane back
                            kx_lockstep();
                            bool const shared = valid && (u32)cnt[h] != (u32)lane;
                            for (u64 todo = kx_ballot(shared); todo; ) {
                                int const L = (int)kx_ctz64(todo);
                                u32 const hL = kx_bcast(h, L);
                                u64 const grp = kx_ballot(valid && h == hL);
                                if (valid && h == hL) { below = kx_popc64(grp & ((1ull << lane) - 1ull)); group = kx_popc64(grp); }
                                todo &= ~grp;
                            }
                            // (a lane that read its own mark can still share its bucket with LOWER lanes whose marks it overwrote: they were
                            // `shared`, so their round set this lane's below / group too -- the ballot is over all valid lanes of the bucket)
                        }
                        rk = old + below;
                        kx_lockstep();
                        if (valid && below + 1u == group) cnt[h] = (u16)(old + group);      // the highest lane of the bucket
                        kx_lockstep();
                    }
                    if (nw > 1) kx_block_sync();
                }
                if (valid) rank[p] = (u16)rk;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) { hq[k] = hn[k]; vq[k] = vn[k]; }
        }
        kx_block_sync();
        // ---- pass 2: bucket starts (each thread scans 32768 / nthreads consecutive buckets; the threads' sums through LDS)
        {
            int const per = hsize / nthreads;           // (256 threads, 256 buckets and more)
            u32 s = 0, occ = 0;
            for (int i = 0; i < per; i++) { u32 const v = cnt[tid * per + i]; s += v; occ += v ? 1u : 0u; }
            // how many of the 32 768 buckets the slice uses says how varied its bytes are: a cheap stand-in for what the parse will
            // cost (binary and text slices use thousands, structured records hundreds), by which the parse kernel orders its slices
            if (!SEG && a.order_key) {
                occp[tid] = occ;
                kx_block_sync();
                if (tid == 0) {
                    u32 tot = 0; for (int t = 0; t < nthreads; t++) tot += occp[t];
                    u32 k = tot >> 6; if (k > 255u) k = 255u;
                    a.order_key[slice] = k; kx_atomic_add(a.order_hist + k, 1u);
                }
            }
            part[tid] = s;
            kx_block_sync();
            if (tid == 0) { u32 run = 0; for (int t = 0; t < nthreads; t++) { u32 const v = part[t]; part[t] = run; run += v; } }
            kx_block_sync();
            u32 run = part[tid];
            for (int i = 0; i < per; i++) { u32 const v = cnt[tid * per + i]; cnt[tid * per + i] = (u16)run; run += v; }
        }
        kx_block_sync();
        // ---- pass 3
        for (u32 p = (u32)tid; p < nIns; p += (u32)nthreads) {
            u64 const w8 = kdl_ld64(src, (int)p, (int)n);
            u32 const w = (u32)w8;
            u32 const h = kd_hash3(a, w & 0xFFu, (w >> 8) & 0xFFu, (w >> 16) & 0xFFu);
            u32 const rk = rank[p];
            u32 const where = (u32)cnt[h] + rk;
            wr[p] = where | (rk << 16);
            srt[where] = (u16)p;
            KdlBytes e; e.lo = w8; e.hi = ((int)p + 8 < (int)n) ? kdl_ld64(src, (int)p + 8, (int)n) : 0ull;
            sb[where] = e;
        }
        kx_block_sync();
    }
}

// ---------------------------------------------------------------------------
// k_deflate_lazy: one wave per slice
// ---------------------------------------------------------------------------
template <bool SEG>        // SEG: one segment of a slice above 64 KiB (see deflate_sort_body): positions stay absolute, the arrays are the span's
KX_DEV void deflate_lazy_slice(const KdArgs& a, u32 slice, int lane)
{
    constexpr bool TINY = false;
    {
        const u8* const src = a.src + a.in_off[slice]; int const n = (int)a.in_len[slice];
        int const s0 = SEG ? (int)(a.seg * KDL_SEG_STEP) : 0;                        // the span's first position
        if (SEG && a.seg > 0 && n <= s0 + (int)KDL_SEG_STEP) return;                  // finished by an earlier segment
        bool const to_end = !SEG || n <= s0 + (int)KDL_SEG_SPAN;                     // this launch parses the slice to its end
        int const segEnd = s0 + (int)KDL_SEG_SPAN;                                   // else: up to here, the next launch goes on
        const u32* const wr = kdl_wr<SEG>(a, slice) - s0; const u16* const srt = kdl_srt<SEG>(a, slice); const KdlBytes* const sb = kdl_sb<SEG>(a, slice);   // (wr is taken at absolute positions)
        u32* const syms = a.syms + (size_t)slice * a.pos_cap;
        KdBlockInfo* const blocks = a.blocks + (size_t)slice * a.blk_cap;
        KdSliceMeta mm; mm.nblocks = 0; mm.nsym = 0; mm.pad[0] = 0; mm.pad[1] = 0;
        int strstart = 0; int match_length = 2, prev_length = 2; int match_dist = 0, prev_dist = 0; bool match_available = false;
        u32 nsym = 0, blockSyms = 0; int block_start = 0;
        u32 symq = 0;                                      // lane (nsym & 63) holds symbol nsym until 64 are there
        // zlib's 64 KiB window buffer (see deflate_parse_body): only the stored-block eligibility of a block depends on it
        int const W = (int)a.wsize, MD = (int)a.max_dist; u32 const LB = a.lit_buf;
        int base = 0, dataEnd = n < 2 * W ? n : 2 * W;
        int const maxChain = (int)a.chain, niceMax = (int)a.nice;
        // the positions the sort has placed (absolute): 0 .. n - 3, in segment mode those of the span
        int nIns = n >= 3 ? n - 2 : 0;
        if (SEG && nIns > segEnd) nIns = segEnd;
        u32* const st_ = SEG ? a.seg_state + (size_t)slice * KDL_STATE_WORDS : nullptr;
        if (SEG && a.seg > 0) {
            // where the previous segment stopped (the open group of symbols waits in the symbol array)
            strstart = (int)st_[0]; match_length = (int)st_[1]; match_dist = (int)st_[2]; match_available = st_[3] != 0u;
            nsym = st_[4]; blockSyms = st_[5]; block_start = (int)st_[6]; base = (int)st_[7]; dataEnd = (int)st_[8]; mm.nblocks = st_[9];
            if ((u32)lane < (nsym & 63u)) symq = syms[(nsym & ~63u) + (u32)lane];
        }
        // The parse is one decision chain per slice and runs on the scalar unit, of which a compute unit has ONE: with 32 slices per CU
        // in flight the kernel is bound by scalar instructions per position (counters: profiles/r04_deflate_lazy.txt), so the parse
        // touches as few positions as it can and does as little as it can at each.
        // where / rank of the 64 positions from wbase on sit in a register window (lane i: wr[wbase + i]), and with them a verdict for
        // each, worked out by all lanes at once when the window is loaded: can a search there find anything?  A position with no
        // candidate, or whose one or two candidates differ from it within the first three bytes, cannot (longest_match would return
        // what it started with): the parse passes it without a search, and a RUN of such positions -- incompressible data is nothing
        // else -- in one step that emits their literals 64 lanes wide.
        int wbase = -(1 << 20); u32 wrv = 0; u64 maybe = 0;
        // The candidates of a search are one memory round trip that the eight waves of a SIMD only half hide (counters: waves wait for memory
        // in half of their cycles).  A search at p that has no match behind it is followed by a step at p + 1 whatever it finds, and that
        // step usually searches too (the lazy parse's second look): its candidates are asked for together with p's and wait in registers.
        int pf_pos = -1, pf_c = 0; KdlBytes pf_cb; pf_cb.lo = 0; pf_cb.hi = 0; u64 pf_s0 = 0, pf_s1 = 0;
#define KDL_FLUSH() { KdBlockInfo b_; \
        b_.nsym_end = nsym; b_.end_pos = (u32)strstart; b_.start_pos = (u32)block_start; b_.stored_ok = (block_start - base >= 0) ? 1u : 0u; \
        if (lane == 0 && mm.nblocks < a.blk_cap) blocks[mm.nblocks] = b_; \
        mm.nblocks++; block_start = strstart; blockSyms = 0; }
        // (a symbol whose value a single lane fetches -- a literal -- is loaded by that lane only, and nobody waits for it before the 64 are stored)
#define KDL_TALLY_MATCH(dist_, lc_) { u32 const v__ = (u32)(dist_) | ((u32)(lc_) << 16); \
        if ((u32)lane == (nsym & 63u)) symq = v__; \
        nsym++; blockSyms++; \
        if ((nsym & 63u) == 0) syms[nsym - 64u + (u32)lane] = symq; }
#define KDL_TALLY_LIT(pos_) { if ((u32)lane == (nsym & 63u)) symq = (u32)src[pos_] << 16; \
        nsym++; blockSyms++; \
        if ((nsym & 63u) == 0) syms[nsym - 64u + (u32)lane] = symq; }
        for (;;) {
            if (lane == 0) KX_STAT(40, 1);                          // steps of the parse
            if (SEG && !to_end && strstart >= segEnd) {
                // the next segment's launch goes on from here
                if (lane == 0) {
                    st_[0] = (u32)strstart; st_[1] = (u32)match_length; st_[2] = (u32)match_dist; st_[3] = match_available ? 1u : 0u;
                    st_[4] = nsym; st_[5] = blockSyms; st_[6] = (u32)block_start; st_[7] = (u32)base; st_[8] = (u32)dataEnd; st_[9] = mm.nblocks;
                }
                if ((u32)lane < (nsym & 63u)) syms[(nsym & ~63u) + (u32)lane] = symq;
                return;
            }
            if (dataEnd - strstart < KD_MIN_LOOKAHEAD) {
                // fill_window: one pass is enough (it brings at least 65 536 - strstart bytes, or all that is left)
                int const rel = strstart - base;
                int const slide = (rel >= W + MD) ? W : 0;
                base += slide;
                int const more = 2 * W - (dataEnd - base);
                dataEnd += (n - dataEnd < more) ? n - dataEnd : more;
                if (dataEnd == strstart) break;
            }
            int const lookahead = n - strstart;
            bool const search_here = lookahead >= KD_MIN_MATCH && match_length < (int)a.lazy;       // (match_length: what becomes prev_length below)
            if (search_here && (strstart < wbase || strstart >= wbase + 64)) {
                if (lane == 0) KX_STAT(41, 1);                      // window loads
                wbase = strstart; wrv = (strstart + lane < nIns) ? wr[strstart + lane] : 0u;
                u32 const wh = wrv & 0xFFFFu, rk_ = wrv >> 16;
                // the position's own entry and its nearest candidates are neighbours in the sorted array (three loads, no branch between them)
                u32 const own = (u32)sb[wh].lo & 0xFFFFFFu;
                u32 const c1 = (u32)sb[wh - (rk_ >= 1u ? 1u : 0u)].lo & 0xFFFFFFu, c2 = (u32)sb[wh - (rk_ >= 2u ? 2u : 0u)].lo & 0xFFFFFFu;
                bool const m = rk_ > 2u || (rk_ >= 1u && c1 == own) || (rk_ == 2u && c2 == own);      // more than two candidates: ask the search
                maybe = kx_ballot(m && strstart + lane < nIns);
            }
            int const off = strstart - wbase;
            if (search_here && match_length == KD_MIN_MATCH - 1 && !((maybe >> off) & 1ull)) {
                // ---- a run of positions at which nothing can be found, after a position at which nothing was: literals, all at once.
                // Bounded by the window, by fill_window's next turn, by the symbol buffer's 64 and by the block's 16 383 symbols (the
                // step that fills a block is left to the ordinary path below).
                u64 const rest = maybe >> off;
                int K = rest ? (int)kx_ctz64(rest) : 64; if (K > 64 - off) K = 64 - off;
                int const room_w = dataEnd - KD_MIN_LOOKAHEAD - strstart + 1; if (K > room_w) K = room_w;
                if (SEG && !to_end && K > segEnd - strstart) K = segEnd - strstart;       // (the positions beyond belong to the next segment)
                int const first_lit = match_available ? strstart - 1 : strstart;          // the first position passed emits the byte behind it, if that is still owed
                int L = match_available ? K : K - 1;
                int const room_q = 64 - (int)(nsym & 63u), room_b = (int)(LB - 2u) - (int)blockSyms;
                int const Lmax = room_q < room_b ? room_q : room_b;
                if (L > Lmax) { K -= L - Lmax; L = Lmax; }
                if (K >= 2 && L >= 1) {
                    if (lane == 0) { KX_STAT(42, 1); KX_STAT(43, K); }   // run steps, positions they pass
                    int const i = lane - (int)(nsym & 63u);
                    if (i >= 0 && i < L) symq = (u32)src[first_lit + i] << 16;
                    nsym += (u32)L; blockSyms += (u32)L;
                    if ((nsym & 63u) == 0) syms[nsym - 64u + (u32)lane] = symq;
                    strstart += K; match_available = true;
                    prev_length = KD_MIN_MATCH - 1;                               // (match_length stays MIN_MATCH - 1: nothing was found at the last of them either)
                    continue;
                }
            }
            prev_length = match_length; prev_dist = match_dist;
            match_length = KD_MIN_MATCH - 1;
            if (search_here && ((maybe >> off) & 1ull)) {
                // ---- longest_match(strstart), starting from best_len = prev_length: only a longer match changes anything; a previous
                // match >= good_match shortens the walk to a quarter of max_chain
                u32 const w0 = kx_bcast(wrv, off);
                int const where = (int)(w0 & 0xFFFFu), rk = (int)(w0 >> 16);
                int const chain = prev_length >= (int)a.good ? (maxChain >> 2) : maxChain;
                int const ncand = rk < chain ? rk : chain;
                int const maxlen = lookahead < KD_MAX_MATCH ? lookahead : KD_MAX_MATCH;
                int const nice = lookahead < niceMax ? lookahead : niceMax;
                int bestLen = prev_length, bestPos = -1;
                // the string at the position: every lane reads the same sixteen bytes (one request), beside the candidates' loads
                u64 scan0, scan1; int c0 = 0; KdlBytes cb0; cb0.lo = 0; cb0.hi = 0;
                if (lane == 0) { KX_STAT(44, 1); KX_STAT(45, pf_pos == strstart ? 1 : 0); KX_STAT(46, prev_length >= KD_MIN_MATCH ? 1 : 0); KX_STAT(47, (ncand + 63) / 64); }   // searches, served from the step before, second looks, candidate blocks
                if (pf_pos == strstart) { scan0 = pf_s0; scan1 = pf_s1; c0 = pf_c; cb0 = pf_cb; }     // asked for a step ago
                else {
                    scan0 = kdl_get64<TINY>(src, strstart, n); scan1 = kdl_get64<TINY>(src, strstart + 8, n);
                    if (lane < ncand) { c0 = s0 + (int)srt[where - 1 - lane]; cb0 = sb[where - 1 - lane]; }
                }
                pf_pos = -1;
                if (prev_length < KD_MIN_MATCH && off + 1 < 64 && ((maybe >> (off + 1)) & 1ull)) {
                    // (the next step's chain may be quartered by what this search finds: the loads take the full first block, the step masks it)
                    u32 const w1 = kx_bcast(wrv, off + 1);
                    int const where1 = (int)(w1 & 0xFFFFu), rk1 = (int)(w1 >> 16);
                    int const n1 = rk1 < maxChain ? rk1 : maxChain;
                    pf_c = 0; pf_cb.lo = 0; pf_cb.hi = 0;
                    if (lane < n1) { pf_c = s0 + (int)srt[where1 - 1 - lane]; pf_cb = sb[where1 - 1 - lane]; }
                    pf_s0 = kdl_get64<TINY>(src, strstart + 1, n); pf_s1 = kdl_get64<TINY>(src, strstart + 9, n);
                    pf_pos = strstart + 1;
                    if (lane == 0) KX_STAT(49, 1);                      // searches asked for ahead
                }
                for (int cb = 0; cb < ncand; cb += 64) {
                    int const j = cb + lane;
                    bool valid = j < ncand;
                    int c = valid ? c0 : 0;
                    KdlBytes cbytes; cbytes.lo = 0; cbytes.hi = 0;
                    if (valid) cbytes = cb0;
                    if (cb > 0) {
                        c = valid ? s0 + (int)srt[where - 1 - j] : 0;
                        if (valid) cbytes = sb[where - 1 - j];               // (the candidates' first bytes lie next to each other, like their positions)
                    }
                    // the window's base is zlib's NIL (position 0 at first; slide_hash turns position w_size into 0 later); beyond MAX_DIST the
                    // chain ends (and with it every later candidate: they lie further back).  The head of the chain may lie exactly MAX_DIST
                    // back (deflate_slow's test) -- which can be the base when the input ends: fill_window then runs, and slides, at every
                    // step --, the others must be nearer (longest_match's limit).
                    bool const inWin = valid && c > base && (j == 0 ? strstart - c <= MD : strstart - c < MD);
                    u64 const ended = kx_ballot(valid && !inWin);
                    valid = inWin;
                    int len = 0;
                    if (valid) {
                        u64 const d0 = cbytes.lo ^ scan0, d1 = cbytes.hi ^ scan1;
                        len = d0 ? (int)(kx_ctz64(d0) >> 3) : d1 ? 8 + (int)(kx_ctz64(d1) >> 3) : 16;
                    }
                    // the lanes whose first sixteen bytes agree go on, sixteen bytes at a time
                    bool more = valid && len == 16 && len < maxlen;
                    for (int done16 = 16; kx_any(more); done16 += 16) {
                        if (lane == 0) KX_STAT(48, 1);                  // extension rounds
                        if (more) {
                            int const q = c + done16, r = strstart + done16;
                            u64 const d0 = kdl_get64<TINY>(src, q, n) ^ kdl_get64<TINY>(src, r, n);
                            u64 const d1 = kdl_get64<TINY>(src, q + 8, n) ^ kdl_get64<TINY>(src, r + 8, n);
                            if (d0) { len = done16 + (int)(kx_ctz64(d0) >> 3); more = false; }
                            else if (d1) { len = done16 + 8 + (int)(kx_ctz64(d1) >> 3); more = false; }
                            else { len = done16 + 16; if (len >= maxlen) more = false; }
                        }
                    }
                    if (len > maxlen) len = maxlen;
                    // the walk stops at the first candidate that reaches nice_length
                    u64 const niceM = kx_ballot(valid && len >= nice);
                    bool const take = valid && (niceM == 0 || lane <= (int)kx_ctz64(niceM));
                    // ... and keeps the longest it has seen, the earlier one among equals: successive improvements in lane order
                    for (u64 up = kx_ballot(take && len > bestLen); up; up = kx_ballot(take && len > bestLen)) {
                        int const L = (int)kx_ctz64(up);
                        bestLen = (int)kx_bcast((u32)len, L); bestPos = (int)kx_bcast((u32)c, L);
                    }
                    if (niceM != 0 || ended != 0) break;
                }
                if (bestPos >= 0) {
                    match_length = bestLen; match_dist = strstart - bestPos;
                    if (match_length == KD_MIN_MATCH && match_dist > KD_TOO_FAR) match_length = KD_MIN_MATCH - 1;
                }
            }
            if (prev_length >= KD_MIN_MATCH && match_length <= prev_length) {
                KDL_TALLY_MATCH(prev_dist, prev_length - KD_MIN_MATCH)
                bool const bflush = blockSyms == LB - 1u;
                strstart += prev_length - 1;
                match_available = false; match_length = KD_MIN_MATCH - 1;
                if (bflush) KDL_FLUSH()
            } else if (match_available) {
                KDL_TALLY_LIT(strstart - 1)
                if (blockSyms == LB - 1u) KDL_FLUSH()
                strstart++;
            } else { match_available = true; strstart++; }
        }
        if (match_available) KDL_TALLY_LIT(strstart - 1)
        KDL_FLUSH()
        if ((nsym & 63u) != 0 && (u32)lane < (nsym & 63u)) syms[(nsym & ~63u) + (u32)lane] = symq;
        mm.nsym = nsym;
        if (lane == 0) a.meta[slice] = mm;
#undef KDL_TALLY_MATCH
#undef KDL_TALLY_LIT
#undef KDL_FLUSH
    }
}

template <bool SEG = false>
KX_DEV void deflate_lazy_body(const KdArgs& a)
{
    int const lane = kx_lane();
    for (u32 it = kx_block(); it < a.n_slices; it += kx_nblocks()) {
        // the slices that will take longest first (a.order: k_deflate_sort's estimate, largest first), so that the launch does not end
        // with a few waves walking its heaviest slices alone
        u32 const slice = a.order ? a.order[it] : kx_xcd_chunk(it, a.n_slices);
        deflate_lazy_slice<SEG>(a, slice, lane);
    }
}

// ---------------------------------------------------------------------------
// k_deflate_parse_wave: one wave per slice over k_deflate_best's records (slices above 64 KiB)
// ---------------------------------------------------------------------------
// The older kernels' parse was a LANE per slice: every step a dependent 8-byte load of best[strstart] (a round trip to memory per
// position visited), and a piece of long slices has few of them -- 4 096 slices of 256 KiB are 64 waves.  Here the decision chain is
// deflate_lazy_slice's, wave-uniform on the scalar unit, and the records of the 64 positions from wbase on sit in registers (lane i:
// best[wbase + i], one coalesced load per window); a step reads its record with two v_readlane, runs of positions without any match
// leave as literals 64 lanes wide, the symbols 64 at a time.  Same output as deflate_parse_body (the emulator runs both).
KX_DEV void deflate_parse_wave_slice(const KdArgs& a, u32 slice, int lane)
{
    const u8* const src = a.src + a.in_off[slice]; int const n = (int)a.in_len[slice];
    const KdBest* const best = a.best + (size_t)slice * a.pos_cap;
    u32* const syms = a.syms + (size_t)slice * a.pos_cap;
    KdBlockInfo* const blocks = a.blocks + (size_t)slice * a.blk_cap;
    KdSliceMeta mm; mm.nblocks = 0; mm.nsym = 0; mm.pad[0] = 0; mm.pad[1] = 0;
    int strstart = 0; int match_length = 2, prev_length = 2; int match_dist = 0, prev_dist = 0; bool match_available = false;
    u32 nsym = 0, blockSyms = 0; int block_start = 0;
    u32 symq = 0;                                      // lane (nsym & 63) holds symbol nsym until 64 are there
    int const W = (int)a.wsize, MD = (int)a.max_dist; u32 const LB = a.lit_buf;
    int base = 0, dataEnd = n < 2 * W ? n : 2 * W;     // zlib's window buffer (see deflate_parse_body)
    int wbase = -(1 << 20); u32 r_full = 0, r_quarter = 0; u64 maybe = 0;       // len | dist << 16 of a full chain and of a quarter of it
#define KDW_FLUSH() { KdBlockInfo b_; \
        b_.nsym_end = nsym; b_.end_pos = (u32)strstart; b_.start_pos = (u32)block_start; b_.stored_ok = (block_start - base >= 0) ? 1u : 0u; \
        if (lane == 0 && mm.nblocks < a.blk_cap) blocks[mm.nblocks] = b_; \
        mm.nblocks++; block_start = strstart; blockSyms = 0; }
#define KDW_TALLY_MATCH(dist_, lc_) { u32 const v__ = (u32)(dist_) | ((u32)(lc_) << 16); \
        if ((u32)lane == (nsym & 63u)) symq = v__; \
        nsym++; blockSyms++; \
        if ((nsym & 63u) == 0) syms[nsym - 64u + (u32)lane] = symq; }
#define KDW_TALLY_LIT(pos_) { if ((u32)lane == (nsym & 63u)) symq = (u32)src[pos_] << 16; \
        nsym++; blockSyms++; \
        if ((nsym & 63u) == 0) syms[nsym - 64u + (u32)lane] = symq; }
    for (;;) {
        if (dataEnd - strstart < KD_MIN_LOOKAHEAD) {
            int const rel = strstart - base;
            int const slide = (rel >= W + MD) ? W : 0;
            base += slide;
            int const more = 2 * W - (dataEnd - base);
            dataEnd += (n - dataEnd < more) ? n - dataEnd : more;
            if (dataEnd == strstart) break;
        }
        int const lookahead = n - strstart;
        bool const search_here = lookahead >= KD_MIN_MATCH && match_length < (int)a.lazy;       // (match_length: what becomes prev_length below)
        if (search_here && (strstart < wbase || strstart >= wbase + 64)) {
            wbase = strstart;
            KdBest r; r.len128 = 0; r.dist128 = 0; r.len32 = 0; r.dist32 = 0;
            if (strstart + lane + 2 < n) r = best[strstart + lane];         // (k_deflate_best writes the positions that have three bytes)
            r_full = (u32)r.len128 | ((u32)r.dist128 << 16); r_quarter = (u32)r.len32 | ((u32)r.dist32 << 16);
            maybe = kx_ballot(r.len128 != 0);                                // (a quarter of the chain finds nothing where the full chain does not)
        }
        int const off = strstart - wbase;
        if (search_here && match_length == KD_MIN_MATCH - 1 && !((maybe >> off) & 1ull)) {
            // a run of positions at which nothing is found, after one at which nothing was: literals, all at once (deflate_lazy_slice)
            u64 const rest = maybe >> off;
            int K = rest ? (int)kx_ctz64(rest) : 64; if (K > 64 - off) K = 64 - off;
            int const room_w = dataEnd - KD_MIN_LOOKAHEAD - strstart + 1; if (K > room_w) K = room_w;
            int const first_lit = match_available ? strstart - 1 : strstart;
            int L = match_available ? K : K - 1;
            int const room_q = 64 - (int)(nsym & 63u), room_b = (int)(LB - 2u) - (int)blockSyms;
            int const Lmax = room_q < room_b ? room_q : room_b;
            if (L > Lmax) { K -= L - Lmax; L = Lmax; }
            if (K >= 2 && L >= 1) {
                int const i = lane - (int)(nsym & 63u);
                if (i >= 0 && i < L) symq = (u32)src[first_lit + i] << 16;
                nsym += (u32)L; blockSyms += (u32)L;
                if ((nsym & 63u) == 0) syms[nsym - 64u + (u32)lane] = symq;
                strstart += K; match_available = true;
                prev_length = KD_MIN_MATCH - 1;
                continue;
            }
        }
        prev_length = match_length; prev_dist = match_dist;
        match_length = KD_MIN_MATCH - 1;
        if (search_here) {
            u32 const rec = prev_length >= (int)a.good ? kx_bcast(r_quarter, off) : kx_bcast(r_full, off);
            int const len = (int)(rec & 0xFFFFu), dist = (int)(rec >> 16);
            // (the window's base is zlib's NIL: see deflate_parse_body)
            if (len > prev_length && !(dist == MD && strstart - dist <= base)) {
                match_length = len; match_dist = dist;
                if (match_length == KD_MIN_MATCH && match_dist > KD_TOO_FAR) match_length = KD_MIN_MATCH - 1;
            }
        }
        if (prev_length >= KD_MIN_MATCH && match_length <= prev_length) {
            KDW_TALLY_MATCH(prev_dist, prev_length - KD_MIN_MATCH)
            bool const bflush = blockSyms == LB - 1u;
            strstart += prev_length - 1;
            match_available = false; match_length = KD_MIN_MATCH - 1;
            if (bflush) KDW_FLUSH()
        } else if (match_available) {
            KDW_TALLY_LIT(strstart - 1)
            if (blockSyms == LB - 1u) KDW_FLUSH()
            strstart++;
        } else { match_available = true; strstart++; }
    }
    if (match_available) KDW_TALLY_LIT(strstart - 1)
    KDW_FLUSH()
    if ((nsym & 63u) != 0 && (u32)lane < (nsym & 63u)) syms[(nsym & ~63u) + (u32)lane] = symq;
    mm.nsym = nsym;
    if (lane == 0) a.meta[slice] = mm;
#undef KDW_TALLY_MATCH
#undef KDW_TALLY_LIT
#undef KDW_FLUSH
}

KX_DEV void deflate_parse_wave_body(const KdArgs& a)
{
    int const lane = kx_lane();
    for (u32 it = kx_block(); it < a.n_slices; it += kx_nblocks()) deflate_parse_wave_slice(a, kx_xcd_chunk(it, a.n_slices), lane);
}
