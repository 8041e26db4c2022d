// zstd_match_fast.h -- the LZ stage of zstd levels 1 and 2 (strategy "fast": one hash table) for slices of at most
// 128 KiB.  Level 1 is what the reference's Ktor encoder asks for (kompressor-zstd-ktor ZstdContentEncoder.kt:11;
// SURVEY 8f rank 4); the arithmetic is libzstd 1.5.7's ZSTD_compressBlock_fast_noDict_generic: positions are searched
// in adjacent pairs (ip0, ip0+1), the repcode is tried at ip0 + step before ip0's own candidate, the pair distance
// grows by one after every 128 bytes without a match, and which position gets inserted after a hit depends on the
// phase.  One team of G lanes per slice as in zstd_match.h; the search is not speculative (lane 0 walks the pairs,
// the team extends matches).  All cross-lane primitives are called from wave-uniform control flow.
#pragma once
#include "zstd_match.h"

struct KFastArgs { KMatchArgs m; u32 level; u32 step0 = 2; };     // level 1 or 2 (block mode: level 1; m.flags bit 8 = stream of unknown size), or 0 = a negative
                                                                  // level (row 0 of libzstd's tables) with step0 = 1 - level (ZSTD_compressBlock_fast: targetLength + 1)

// ZSTD_getCParams(level, n, 0) for the fast rows: windowLog, hashLog, minMatch
KX_DEV void kx_params_fast_w(u32 level, u32 n, u32& W, u32& hashLog, u32& mml)
{
    if (level == 0) {            // negative levels
        if (n <= 16384) { W = 14; hashLog = 13; mml = 5; } else if (n <= 131072) { W = 17; hashLog = 12; mml = 5; }
        else if (n <= 262144) { W = 18; hashLog = 13; mml = 5; } else { W = 19; hashLog = 13; mml = 6; }
    } else
    if (level == 1) {
        if (n <= 16384) { W = 14; hashLog = 15; mml = 5; } else if (n <= 131072) { W = 17; hashLog = 13; mml = 6; }
        else if (n <= 262144) { W = 18; hashLog = 14; mml = 6; } else { W = 19; hashLog = 14; mml = 7; }
    } else {
        if (n <= 16384) { W = 14; hashLog = 15; mml = 4; } else if (n <= 131072) { W = 17; hashLog = 15; mml = 5; }
        else { W = 20; hashLog = 16; mml = 6; }         // (128 KiB < n <= 256 KiB is a double-fast row: kx_params_l2_dfast)
    }
    u32 const srcLog = (n < 64) ? 6 : kx_hb32(n - 1) + 1;
    if (W > srcLog) W = srcLog;
    if (hashLog > W + 1) hashLog = W + 1;
    if (W < 10) W = 10;
}
KX_DEV void kx_params_fast(u32 level, u32 n, u32& hashLog, u32& mml) { u32 W; kx_params_fast_w(level, n, W, hashLog, mml); }
// the window log a frame at a "fast" level is written with: the size's row when the size is known, else the row of the largest sizes
KX_DEV u32 kx_window_log_fast(u32 level, u32 n, bool unknown_size)
{
    if (unknown_size) return level == 2u ? 20u : 19u;
    u32 W, h, m; kx_params_fast_w(level, n, W, h, m);
    return W;
}

enum { KFS_IDLE = 0, KFS_START = 1, KFS_PAIR = 2, KFS_REPLOOP = 3, KFS_MATCH = 4, KFS_CLEANUP = 5, KFS_DONE = 6 };

// BLK as in zstd_match.h: the block [ipos, ipos + blockSize) of every unfinished slice, per-slice table, repcodes from
// and back to the frame state.
template <int G, bool BLK = false>
KX_DEV void zstd_match_fast_body(const KFastArgs& f)
{
    bool const wide = BLK && (f.m.flags & 16u);             // slices of 4 MiB and more: plain 32-bit indices, no check bits
    u32 const IDXM = BLK ? (wide ? 0xFFFFFFFFu : KX_BLK_IDX_MASK) : KX_IDX_MASK;
    constexpr u32 TAGM = BLK ? 0u : KX_TAG_MASK;            // block mode: plain indices, no epoch
    u32 const CHKM = BLK ? (wide ? 0u : KX_BLK_CHK_MASK) : KX_CHK_MASK;      // check bits (first 4 bytes of the position, what a candidate is compared on)
    u32 lowIdx = 2u;                                        // lowest valid index (position + 2): 2 unless the window has slid (block mode, long slices)
#define KFS_CK(bytes4_) (wide ? 0u : kx_chk_short((u64)(bytes4_)) << (BLK ? KX_BLK_IDX_BITS : KX_CHK_SHIFT))
    constexpr int NT = 64 / G;
    const KMatchArgs& a = f.m;
    int const lane = kx_lane();
    int const k = lane & (G - 1);
    int const tbase = lane - k;
    u32 const team = kx_block() * NT + (u32)(lane / G);
    u32* H = BLK ? a.big_tables : kx_team_tables(a, team);
    int bstart = 0; u32 saved1 = 0, saved2 = 0;
    u64 const tmask = (G == 64) ? ~0ull : ((1ull << G) - 1ull);

    int state = KFS_IDLE;
    const u8* src = a.src; int n = 0, ilimit = 0; u32 slice = 0;
    int ip0 = 0, anchor = 0; u32 rep1 = 1, rep2 = 0; u32 nseq = 0, nlit = 0, tag = 0, hlog = 13, mls = 6;
    int step = 2, gap = 2, nextStep = 0; u32 hash0 = 0, hash1 = 0, matchIdx = 0; int current0 = 0;     // gap = distance from the pair to the next one
    u32 longType = 0, longPos = 0, guard = 0, status = 0;
    KSeq* seqs = a.seqs; u64 sq0 = 0, sq1 = 0;
    int m_start = 0, m_mpos = 0; u32 m_len0 = 0, m_off = 0; bool m_back = false, m_fill = false;

    for (;;) {
        if (kx_any(state == KFS_IDLE)) {
            u32 s = 0, ep = 0;
            if (state == KFS_IDLE && k == 0) {
                s = kx_atomic_add(a.counter, 1u);
                if (!BLK && s < a.n_slices) {
                    ep = a.team_epoch[team] + 1;
                    if (ep > KX_EPOCH_MAX) ep = 0;
                    a.team_epoch[team] = ep ? ep : 1u;
                }
            }
            s = kx_shfl(s, tbase); ep = kx_shfl(ep, tbase);
            if (state == KFS_IDLE) {
                if (s >= a.n_slices) state = KFS_DONE;
                else if (BLK) {
                    KFrameState const fs = a.fstate[s];
                    // (a block that libzstd parses with the extDict variant -- behind a wrap of its staging buffer -- is left to zstd_match_fast_ext_body)
                    KBlockWin const bw = kx_block_window(fs.lowLimit, fs.dictLimit, fs.ipos, fs.blockSize, kx_window_log_fast(f.level, a.in_len[s], (a.flags & 8u) != 0));
                    if (fs.blockSize != 0 && !bw.ext && kx_in_class((a.flags >> 6) & 3u, a.in_len[s])) {
                        slice = s;
                        src = a.src + a.in_off[s];
                        seqs = a.seqs + (size_t)s * a.seq_cap;
                        H = a.big_tables + (size_t)s * KX_BIG_TBL_ENTRIES;
                        kx_params_fast(f.level, a.in_len[s], hlog, mls);
                        if (a.flags & 8u) { hlog = f.level == 2 ? 16 : f.level == 0 ? 13 : 14; mls = f.level == 1 ? 7 : 6; }      // size unknown: level 1 window 19, hash 14, minMatch 7; level 2 window 20, hash 16, minMatch 6; negative levels window 19, hash 13, minMatch 6
                        nseq = 0; nlit = 0; longType = 0; longPos = 0; guard = 0; status = 0; tag = 0;
                        bstart = (int)fs.ipos; n = bstart + (int)fs.blockSize;   // n = end of the block
                        anchor = bstart; ilimit = n - 8;
                        // candidates: valid from ZSTD_getLowestPrefixIndex at the block's END on; the repcodes are checked against the
                        // one at the first searched position (ZSTD_compressBlock_fast_noDict_generic: prefixStartIndex / windowLow)
                        lowIdx = kx_lowest_prefix((u32)n + 2u, bw.dictLimit, bw.maxDist);
                        ip0 = bstart + ((u32)bstart + 2u == lowIdx ? 1 : 0);
                        rep1 = fs.rep[0]; rep2 = fs.rep[1]; saved1 = 0; saved2 = 0;
                        u32 const maxRep = ((u32)ip0 + 2u) - kx_lowest_prefix((u32)ip0 + 2u, bw.dictLimit, bw.maxDist);
                        if (rep2 > maxRep) { saved2 = rep2; rep2 = 0; }
                        if (rep1 > maxRep) { saved1 = rep1; rep1 = 0; }
                        state = (fs.blockSize < 8) ? KFS_CLEANUP : KFS_START;
                    }
                } else {
                    slice = s;
                    src = a.src + a.in_off[s]; n = (int)a.in_len[s];
                    seqs = a.seqs + (size_t)s * a.seq_cap;
                    kx_params_fast(f.level, (u32)n, hlog, mls);
                    nseq = 0; nlit = 0; longType = 0; longPos = 0; guard = 0; status = 0;
                    if (ep == 0) {
                        for (u32 i = (u32)k; i < KX_TBL_ENTRIES; i += G) H[i] = 0;
                        ep = 1;
                    }
                    tag = ep << KX_TAG_SHIFT;
                    anchor = 0; ilimit = n - 8;
                    ip0 = 1; rep1 = 1; rep2 = 0;             // rep {1,4,8}: 4 exceeds the 1 byte of history at ip0 = 1
                    state = (n < 8) ? KFS_CLEANUP : KFS_START;
                }
            }
        }
        if (kx_all(state == KFS_DONE)) break;

        // ================= "_start": a new run of pairs =====================
        if (kx_any(state == KFS_START)) {
            if (state == KFS_START) {
                step = (int)f.step0; gap = (int)f.step0; nextStep = ip0 + 128;
                if (ip0 + step + 1 >= ilimit) state = KFS_CLEANUP;             // (ip3 = ip0 + step + 1)
                else {
                    u32 e = 0, h0 = 0, h1 = 0, ck0 = 0;
                    if (k == 0) {
                        u64 const w0 = kx_ld64(src + ip0);
                        h0 = kx_hash_short_any(w0, hlog, mls);
                        h1 = kx_hash_short_any(kx_ld64(src + ip0 + 1), hlog, mls);
                        e = H[h0]; ck0 = KFS_CK(w0);
                    }
                    hash0 = h0; hash1 = h1;                  // lane 0's copies are the ones used
                    matchIdx = ((e & TAGM) == tag && (e & CHKM) == ck0 && (e & IDXM) >= lowIdx) ? (e & IDXM) : 0u;
                    state = KFS_PAIR;
                }
            }
        }

        // ================= one pair (lane 0 of the team walks it) ===========
        if (kx_any(state == KFS_PAIR)) {
            bool const pr = state == KFS_PAIR;
            u32 kind = 0;            // 0 no hit, 1 repcode at ip0 + step, 2 candidate of the pair's first position, 3 of its second
            int n_ip0 = ip0, n_step = step, n_gap = gap, n_next = nextStep, n_cur = current0; u32 n_mi = matchIdx;
            if (pr && k == 0) {
                int const ip1 = ip0 + 1, ip2 = ip0 + gap, ip3 = ip2 + 1;
                // Everything the pair can ask for is requested up front (the step is a chain of memory latencies, not of
                // instructions): the table slots of ip1 and ip2, the bytes of all four positions, the repcode bytes.  libzstd
                // reads those slots AFTER it has written ip0 (and ip1) into the table; where the slots coincide the value it
                // would have seen is put back by hand.
                // Two pairs out of three end without a hit, so the NEXT pair (B: first position ip2, second ip3, probe
                // ip2 + step) is walked in the same step, on the assumption that this one (A) finds nothing: its loads
                // travel with A's, its table writes and results only count when A indeed found nothing.
                bool const canB = ip2 + 1 + step < ilimit;          // the condition under which the run would go on to B
                int const ip2B = canB ? ip2 + step : ip2, ip3B = ip2B + 1;
                u32 const e1raw = H[hash1];
                u64 const w2 = kx_ld64(src + ip2), w3 = kx_ld64(src + ip3);
                u64 const w2B = kx_ld64(src + ip2B), w3B = kx_ld64(src + ip3B);
                u32 const rval = kx_ld32(src + ip2 - (int)rep1), rvalB = kx_ld32(src + ip2B - (int)rep1);
                u32 const c0 = kx_ld32(src + (matchIdx >= 2u ? (int)matchIdx - 2 : 0));
                u32 const s0 = kx_ld32(src + ip0), s1 = kx_ld32(src + ip1);
                u32 const hash2 = kx_hash_short_any(w2, hlog, mls), hash3 = kx_hash_short_any(w3, hlog, mls);
                u32 const hash2B = kx_hash_short_any(w2B, hlog, mls), hash3B = kx_hash_short_any(w3B, hlog, mls);
                u32 const e2raw = H[hash2], e3raw = H[hash3], e2Braw = H[hash2B];
                u32 const k0 = KFS_CK(s0), k1 = KFS_CK(s1), k2 = KFS_CK(w2), k3 = KFS_CK(w3), k2B = KFS_CK(w2B);
                u32 const t0 = tag | k0 | (u32)(ip0 + 2), t1 = tag | k1 | (u32)(ip1 + 2), t2 = tag | k2 | (u32)(ip2 + 2), t3 = tag | k3 | (u32)(ip3 + 2);
                u32 const e1 = (hash1 == hash0) ? t0 : e1raw;
                u32 const e2 = (hash2 == hash1) ? t1 : (hash2 == hash0) ? t0 : e2raw;
                u32 const e3 = (hash3 == hash2) ? t2 : (hash3 == hash1) ? t1 : (hash3 == hash0) ? t0 : e3raw;
                u32 const e2B = (hash2B == hash3) ? t3 : (hash2B == hash2) ? t2 : (hash2B == hash1) ? t1 : (hash2B == hash0) ? t0 : e2Braw;
                // an entry whose check bits differ from the position's cannot pass the 4-byte compare: no candidate, no fetch
                u32 const mi1 = ((e1 & TAGM) == tag && (e1 & CHKM) == k1 && (e1 & IDXM) >= lowIdx) ? (e1 & IDXM) : 0u;
                u32 const mi2 = ((e2 & TAGM) == tag && (e2 & CHKM) == k2 && (e2 & IDXM) >= lowIdx) ? (e2 & IDXM) : 0u;
                u32 const mi3 = ((e3 & TAGM) == tag && (e3 & CHKM) == k3 && (e3 & IDXM) >= lowIdx) ? (e3 & IDXM) : 0u;
                u32 const mi2B = ((e2B & TAGM) == tag && (e2B & CHKM) == k2B && (e2B & IDXM) >= lowIdx) ? (e2B & IDXM) : 0u;
                u32 const c1 = kx_ld32(src + (mi1 >= 2u ? (int)mi1 - 2 : 0));
                u32 const c0B = kx_ld32(src + (mi2 >= 2u ? (int)mi2 - 2 : 0));
                u32 const c1B = kx_ld32(src + (mi3 >= 2u ? (int)mi3 - 2 : 0));
                n_cur = ip0;
                H[hash0] = t0;
                H[hash1] = t1;                                   // every branch below stores ip1 (as the pair's second position)
                if ((u32)w2 == rval && rep1 > 0) { kind = 1; n_ip0 = ip2; }
                else if (matchIdx >= 2u && c0 == s0) { kind = 2; n_mi = matchIdx; }
                else {
                    n_cur = ip1;
                    if (mi1 >= 2u && c1 == s1) {
                        kind = 3; n_ip0 = ip1; n_mi = mi1;
                        if (step <= 4) H[hash2] = t2;
                        hash0 = hash1; hash1 = hash2;
                    } else {
                        // pair A found nothing: the state the next step would start from ...
                        int stepB = step, nextB = nextStep;
                        if (ip2 + step >= nextStep) { stepB = step + 1; nextB = nextStep + 128; }
                        n_mi = mi2; hash0 = hash2; hash1 = hash3;
                        n_ip0 = ip2;                              // the next pair starts `step` behind this one's second half
                        n_gap = step; n_step = stepB; n_next = nextB;
                        if (canB) {
                            // ... and that step itself: pair B
                            n_cur = ip2;
                            H[hash2] = t2;
                            H[hash3] = t3;
                            if ((u32)w2B == rvalB && rep1 > 0) { kind = 1; n_ip0 = ip2B; }
                            else if (mi2 >= 2u && c0B == (u32)w2) { kind = 2; n_ip0 = ip2; n_mi = mi2; }
                            else {
                                n_cur = ip3;
                                if (mi3 >= 2u && c1B == (u32)w3) {
                                    kind = 3; n_ip0 = ip3; n_mi = mi3;
                                    if (stepB <= 4) H[hash2B] = tag | k2B | (u32)(ip2B + 2);
                                    hash0 = hash3; hash1 = hash2B;
                                } else {
                                    n_mi = mi2B; hash0 = hash2B; hash1 = hash3B;
                                    n_ip0 = ip2B;
                                    n_gap = stepB;
                                    if (ip2B + stepB >= nextB) { n_step = stepB + 1; n_next = nextB + 128; }
                                }
                            }
                        }
                    }
                }
            }
            kind = kx_shfl(kind, tbase); n_ip0 = (int)kx_shfl((u32)n_ip0, tbase); n_step = (int)kx_shfl((u32)n_step, tbase);
            n_next = (int)kx_shfl((u32)n_next, tbase); n_cur = (int)kx_shfl((u32)n_cur, tbase); n_mi = kx_shfl(n_mi, tbase);
            n_gap = (int)kx_shfl((u32)n_gap, tbase);
            if (pr) {
                guard++;
                current0 = n_cur;
                if (kind == 0) {
                    // ip0 = old ip2, ip1 = old ip3 = ip0 + 1 again; ip3 of the new pair decides whether the run goes on
                    ip0 = n_ip0; matchIdx = n_mi; gap = n_gap; step = n_step; nextStep = n_next;
                    if (!(ip0 + 1 + gap < ilimit)) state = KFS_CLEANUP;
                    if (guard > 2u * (u32)n + 64u) { status = 1; state = KFS_CLEANUP; }
                } else if (kind == 1) {
                    int const mp = n_ip0 - (int)rep1;
                    bool const b1 = mp >= 1 && src[n_ip0 - 1] == src[mp - 1];
                    m_start = n_ip0 - (b1 ? 1 : 0); m_mpos = mp - (b1 ? 1 : 0); m_len0 = 4u + (b1 ? 1u : 0u); m_back = false; m_fill = true; m_off = 0;
                    state = KFS_MATCH;
                } else {
                    m_start = n_ip0; m_mpos = (int)n_mi - 2; m_len0 = 4; m_back = true; m_fill = true;
                    m_off = (u32)(m_start - m_mpos);
                    state = KFS_MATCH;
                }
            }
        }

        // ================= immediate repcode =================================
        if (kx_any(state == KFS_REPLOOP)) {
            bool const inrep = state == KFS_REPLOOP;
            bool hit = false;
            if (inrep && ip0 <= ilimit && rep2 > 0) hit = kx_ld32(src + ip0) == kx_ld32(src + ip0 - (int)rep2);
            if (inrep) {
                if (hit) {
                    if (k == 0) { u64 const wr = kx_ld64(src + ip0); H[kx_hash_short_any(wr, hlog, mls)] = tag | KFS_CK(wr) | (u32)(ip0 + 2); }
                    u32 const t = rep2; rep2 = rep1; rep1 = t;
                    m_start = ip0; m_mpos = ip0 - (int)rep1; m_len0 = 4; m_back = false; m_fill = false; m_off = 0;
                    state = KFS_MATCH;
                } else state = KFS_START;
            }
        }

        // ================= take the match ====================================
        if (kx_any(state == KFS_MATCH)) {
            bool const mt = state == KFS_MATCH;
            u32 lenA = kx_team_extend<G>(mt, src, n, m_start, m_mpos, m_len0, k, tbase, tmask);
            int const mlow = m_mpos - ((int)lowIdx - 2);               // the match may grow backwards down to the lowest valid position
            int const mb = (m_start - anchor < mlow) ? m_start - anchor : mlow;
            u32 const back = kx_team_backward<G>(mt && m_back, src, m_start, m_mpos, mb, k, tbase, tmask);
            if (mt) {
                u32 offBase = 1;
                if (m_back) { m_start -= (int)back; lenA += back; rep2 = rep1; rep1 = m_off; offBase = m_off + 3; }
                int const ll = m_start - anchor;
                {
                    u64 const q = (u64)offBase | ((u64)(u16)ll << 32) | ((u64)(u16)(lenA - 3) << 48);   // KSeq
                    u32 const slot = nseq & (2u * G - 1u);
                    if ((u32)k == (slot >> 1)) { if (slot & 1u) sq1 = q; else sq0 = q; }
                    if (slot == 2u * G - 1u) kx_st128(seqs + (nseq - slot) + 2u * (u32)k, sq0, sq1);
                }
                if (ll > 0xFFFF) { longType = 1; longPos = nseq; }
                if (lenA - 3 > 0xFFFF) { longType = 2; longPos = nseq; }
                nseq++; nlit += (u32)ll;
                ip0 = m_start + (int)lenA; anchor = ip0;
                if (m_fill && ip0 <= ilimit && k == 0) {
                    // fill: current0 + 2 and ip0 - 2
                    u64 const wf0 = kx_ld64(src + current0 + 2), wf1 = kx_ld64(src + ip0 - 2);
                    H[kx_hash_short_any(wf0, hlog, mls)] = tag | KFS_CK(wf0) | (u32)(current0 + 2 + 2);
                    H[kx_hash_short_any(wf1, hlog, mls)] = tag | KFS_CK(wf1) | (u32)(ip0 - 2 + 2);
                }
                if (++guard > 2u * (u32)n + 64u) { status = 2; state = KFS_CLEANUP; }
                else state = (ip0 <= ilimit) ? KFS_REPLOOP : KFS_START;
            }
        }

        // ================= finish the slice ==================================
        if (kx_any(state == KFS_CLEANUP)) {
            if (state == KFS_CLEANUP) {
                {
                    u32 const cnt = nseq & (2u * G - 1u);
                    u64* const sp = (u64*)(seqs + (nseq - cnt));
                    if (2u * (u32)k < cnt) sp[2 * k] = sq0;
                    if (2u * (u32)k + 1u < cnt) sp[2 * k + 1] = sq1;
                }
                if (k == 0) {
                    KSliceMeta mm;
                    mm.nbSeq = nseq; mm.litSize = nlit; mm.lastLL = (u32)(n - anchor);
                    mm.longType = longType; mm.longPos = longPos; mm.status = status; mm.pad[0] = 0; mm.pad[1] = 0;
                    if (BLK) {
                        u32 const s2 = (saved1 != 0 && rep1 != 0) ? saved1 : saved2;
                        mm.pad[0] = rep1 ? rep1 : saved1; mm.pad[1] = rep2 ? rep2 : s2;
                    }
                    a.meta[slice] = mm;
                }
                state = KFS_IDLE;
            }
        }
    }
}
#undef KFS_CK

// ---- the same levels once the stream is longer than libzstd's staging buffer (round 4) -------------------------------------
// Window + 128 KiB bytes at these levels (640 KiB at level 1 and the negative levels, 1 MiB + 128 KiB at level 2).  When that
// buffer wraps, the lap before becomes an older segment ("extDict") and libzstd 1.5.7 parses the blocks with
// ZSTD_compressBlock_fast_extDict_generic until ZSTD_window_enforceMaxDist has moved the window past the segment -- four blocks
// in five of a long level-1 stream.  The loop is the regular variant's (pairs of positions, the repcode tried one step ahead,
// the pair distance growing after 128 bytes without a match); what differs is the index rules: candidates are valid from
// dictStartIndex on, a match that starts in the older segment does not grow backwards past its start, a repcode that would
// straddle the boundary is refused (the unsigned test prefixStartIndex - repIndex >= 4), the second position of a pair goes into
// the table after the match, only if it lies before the match's end, and the repcodes set aside at the block's start compare with
// >=.  The bytes of the stream are contiguous here, so both segments are one pointer and ZSTD_count_2segments is an ordinary
// extension.  Block mode only; lane 0 of a team walks the pairs -- in program order: a slot read after a slot write sees it --,
// the team extends matches.  Blocks the regular variant parses are skipped here and the other way round (kx_block_window).
enum { KFX_IDLE = 0, KFX_START = 1, KFX_PAIR = 2, KFX_REPLOOP = 3, KFX_MATCH = 4, KFX_CLEANUP = 5, KFX_DONE = 6 };

template <int G>
KX_DEV void zstd_match_fast_ext_body(const KFastArgs& f)
{
    const KMatchArgs& a = f.m;
    bool const wide = (a.flags & 16u) != 0;
    u32 const IDXM = wide ? 0xFFFFFFFFu : KX_BLK_IDX_MASK;
#define KFX_E(bytes4_, idx_) ((u32)(idx_) | (wide ? 0u : kx_chk_short((u64)(bytes4_)) << KX_BLK_IDX_BITS))
    int const lane = kx_lane();
    int const k = lane & (G - 1);
    int const tbase = lane - k;
    u64 const tmask = (G == 64) ? ~0ull : ((1ull << G) - 1ull);

    int state = KFX_IDLE;
    const u8* src = a.src; u32* H = a.big_tables; KSeq* seqs = a.seqs;
    int n = 0, ilimit = 0, ip0 = 0, anchor = 0; u32 slice = 0; u32 off1 = 1, off2 = 4, saved1 = 0, saved2 = 0;
    u32 nseq = 0, nlit = 0, longType = 0, longPos = 0, guard = 0, status = 0, hlog = 14, mls = 7;
    u32 dsi = 2, psi = 2;                       // dictStartIndex, prefixStartIndex of the block
    int step = 2, gap = 2, nextStep = 0; u32 hash0 = 0, hash1 = 0, idx = 0;         // gap = distance from the pair to the next one (the step in force when it was laid out)
    u64 sq0 = 0, sq1 = 0;
    int m_start = 0, m_mpos = 0, m_low = 0, m_cur0 = 0, m_ip1 = 0; u32 m_len0 = 0, m_off = 0, m_hash1 = 0; bool m_back = false, m_fill = false;

    for (;;) {
        // ================= next slice whose block is an extDict block ==================
        if (kx_any(state == KFX_IDLE)) {
            u32 s = 0;
            if (state == KFX_IDLE && k == 0) s = kx_atomic_add(a.counter, 1u);
            s = kx_shfl(s, tbase);
            if (state == KFX_IDLE) {
                if (s >= a.n_slices) state = KFX_DONE;
                else {
                    KFrameState const fs = a.fstate[s];
                    KBlockWin const bw = kx_block_window(fs.lowLimit, fs.dictLimit, fs.ipos, fs.blockSize, kx_window_log_fast(f.level, a.in_len[s], (a.flags & 8u) != 0));
                    if (fs.blockSize != 0 && bw.ext && kx_in_class((a.flags >> 6) & 3u, a.in_len[s])) {
                        slice = s;
                        src = a.src + a.in_off[s];
                        seqs = a.seqs + (size_t)s * a.seq_cap;
                        H = a.big_tables + (size_t)s * KX_BIG_TBL_ENTRIES;
                        kx_params_fast(f.level, a.in_len[s], hlog, mls);
                        if (a.flags & 8u) { hlog = f.level == 2 ? 16 : f.level == 0 ? 13 : 14; mls = f.level == 1 ? 7 : 6; }
                        dsi = bw.dictStartIndex; psi = bw.prefixStartIndex;
                        nseq = 0; nlit = 0; longType = 0; longPos = 0; guard = 0; status = 0;
                        ip0 = (int)fs.ipos; anchor = ip0; n = ip0 + (int)fs.blockSize; ilimit = n - 8;
                        off1 = fs.rep[0]; off2 = fs.rep[1]; saved1 = 0; saved2 = 0;
                        u32 const maxRep = ((u32)ip0 + 2u) - dsi;
                        if (off2 >= maxRep) { saved2 = off2; off2 = 0; }
                        if (off1 >= maxRep) { saved1 = off1; off1 = 0; }
                        state = (fs.blockSize < 8) ? KFX_CLEANUP : KFX_START;
                    }
                }
            }
        }
        if (kx_all(state == KFX_DONE)) break;

        // ================= "_start": a new run of pairs =====================
        if (kx_any(state == KFX_START)) {
            if (state == KFX_START) {
                step = (int)f.step0; gap = (int)f.step0; nextStep = ip0 + 128;
                if (ip0 + step + 1 >= ilimit) state = KFX_CLEANUP;
                else {
                    u32 h0 = 0, h1 = 0, e = 0;
                    if (k == 0) {
                        h0 = kx_hash_short_any(kx_ld64(src + ip0), hlog, mls);
                        h1 = kx_hash_short_any(kx_ld64(src + ip0 + 1), hlog, mls);
                        e = H[h0] & IDXM;
                    }
                    hash0 = h0; hash1 = h1; idx = e;             // lane 0's copies are the ones used
                    state = KFX_PAIR;
                }
            }
        }

        // ================= one pair (lane 0 of the team walks it, in libzstd's order) ===========
        if (kx_any(state == KFX_PAIR)) {
            bool const pr = state == KFX_PAIR;
            u32 kind = 0;            // 0 no hit, 1 repcode at ip2, 2 candidate of the pair's first position, 3 of its second
            int n_ip0 = ip0, n_step = step, n_gap = gap, n_next = nextStep, n_cur = 0, n_ip1 = 0; u32 n_idx = 0, n_h0 = hash0, n_h1 = hash1, n_rep = 0;
            if (pr && k == 0) {
                int const ip1 = ip0 + 1, ip2 = ip0 + gap, ip3 = ip2 + 1;
                u64 const w2 = kx_ld64(src + ip2);
                u32 const s0 = kx_ld32(src + ip0), s1 = kx_ld32(src + ip1);
                u32 const repIndex = (u32)ip2 + 2u - off1;
                bool const repOk = ((u32)(psi - repIndex) >= 4u) && off1 > 0;
                u32 const rval = repOk ? kx_ld32(src + (int)repIndex - 2) : ((u32)w2 ^ 1u);
                u32 const c0 = idx >= dsi ? kx_ld32(src + (int)idx - 2) : (s0 ^ 1u);
                n_cur = ip0; n_ip1 = ip1;
                H[hash0] = KFX_E(s0, ip0 + 2);
                if ((u32)w2 == rval) { kind = 1; n_ip0 = ip2; n_rep = repIndex; }
                else if (c0 == s0) { kind = 2; n_idx = idx; }
                else {
                    // second position of the pair (a slot this pair has just written is taken from the registers: libzstd reads it after its write)
                    u32 const i1 = (hash1 == hash0) ? (u32)ip0 + 2u : H[hash1] & IDXM;
                    u32 const h2 = kx_hash_short_any(w2, hlog, mls);
                    n_cur = ip1; n_ip1 = ip2;
                    H[hash1] = KFX_E(s1, ip1 + 2);
                    u32 const c1 = i1 >= dsi ? kx_ld32(src + (int)i1 - 2) : (s1 ^ 1u);
                    if (c1 == s1) { kind = 3; n_ip0 = ip1; n_idx = i1; n_h0 = hash1; n_h1 = h2; }
                    else {
                        u32 const i2 = (h2 == hash1) ? (u32)ip1 + 2u : (h2 == hash0) ? (u32)ip0 + 2u : H[h2] & IDXM;
                        u32 const h3 = kx_hash_short_any(kx_ld64(src + ip3), hlog, mls);
                        n_idx = i2; n_h0 = h2; n_h1 = h3;
                        n_ip0 = ip2;                              // the next pair: (ip2, ip3); its successor lies `step` behind it
                        n_gap = step;
                        if (ip2 + step >= nextStep) { n_step = step + 1; n_next = nextStep + 128; }
                    }
                }
            }
            kind = kx_shfl(kind, tbase); n_ip0 = (int)kx_shfl((u32)n_ip0, tbase); n_step = (int)kx_shfl((u32)n_step, tbase);
            n_next = (int)kx_shfl((u32)n_next, tbase); n_cur = (int)kx_shfl((u32)n_cur, tbase); n_idx = kx_shfl(n_idx, tbase);
            n_gap = (int)kx_shfl((u32)n_gap, tbase);
            n_ip1 = (int)kx_shfl((u32)n_ip1, tbase); n_h0 = kx_shfl(n_h0, tbase); n_h1 = kx_shfl(n_h1, tbase); n_rep = kx_shfl(n_rep, tbase);
            if (pr) {
                guard++;
                if (kind == 0) {
                    // libzstd: ip0 = ip1; ip1 = ip2; ip2 = ip0 + step; ip3 = ip1 + step (the step that was in force: gap), THEN the step may grow
                    ip0 = n_ip0; idx = n_idx; hash0 = n_h0; hash1 = n_h1; gap = n_gap; step = n_step; nextStep = n_next;
                    if (!(ip0 + 1 + gap < ilimit)) state = KFX_CLEANUP;               // while (ip3 < ilimit), ip3 = ip1 + step
                    if (guard > 600000u) { status = 1; state = KFX_CLEANUP; }
                } else {
                    m_cur0 = n_cur; m_ip1 = n_ip1; m_hash1 = kind == 3 ? n_h1 : hash1; m_fill = true;
                    if (kind == 1) {
                        int const mp = (int)n_rep - 2;
                        bool const b1 = src[n_ip0 - 1] == src[mp - 1];
                        m_start = n_ip0 - (b1 ? 1 : 0); m_mpos = mp - (b1 ? 1 : 0); m_len0 = 4u + (b1 ? 1u : 0u); m_back = false; m_off = 0; m_low = 0;
                    } else {
                        m_start = n_ip0; m_mpos = (int)n_idx - 2; m_len0 = 4; m_back = true;
                        m_off = (u32)(m_start - m_mpos);
                        m_low = (int)(n_idx < psi ? dsi : psi) - 2;      // lowMatchPtr: the start of the match's own segment
                    }
                    state = KFX_MATCH;
                }
            }
        }

        // ================= immediate repcode =================================
        if (kx_any(state == KFX_REPLOOP)) {
            bool const inrep = state == KFX_REPLOOP;
            bool hit = false; int rp = 0;
            if (inrep && ip0 <= ilimit) {
                u32 const repIndex2 = (u32)ip0 + 2u - off2;
                bool const ok = ((u32)((psi - 1u) - repIndex2) >= 3u) && off2 > 0;
                rp = (int)repIndex2 - 2;
                if (ok) hit = kx_ld32(src + rp) == kx_ld32(src + ip0);
            }
            if (inrep) {
                if (hit) {
                    if (k == 0) { u64 const wr = kx_ld64(src + ip0); H[kx_hash_short_any(wr, hlog, mls)] = KFX_E(wr, ip0 + 2); }
                    u32 const t = off2; off2 = off1; off1 = t;
                    m_start = ip0; m_mpos = rp; m_len0 = 4; m_back = false; m_fill = false; m_off = 0; m_low = 0;
                    state = KFX_MATCH;
                } else state = KFX_START;
            }
        }

        // ================= take the match ====================================
        if (kx_any(state == KFX_MATCH)) {
            bool const mt = state == KFX_MATCH;
            u32 lenA = kx_team_extend<G>(mt, src, n, m_start, m_mpos, m_len0, k, tbase, tmask);
            int const mb = (m_start - anchor < m_mpos - m_low) ? m_start - anchor : m_mpos - m_low;
            u32 const back = kx_team_backward<G>(mt && m_back, src, m_start, m_mpos, mb, k, tbase, tmask);
            if (mt) {
                u32 offBase = 1;
                if (m_back) { m_start -= (int)back; lenA += back; off2 = off1; off1 = m_off; offBase = m_off + 3; }
                int const ll = m_start - anchor;
                {
                    u64 const q = (u64)offBase | ((u64)(u16)ll << 32) | ((u64)(u16)(lenA - 3) << 48);   // KSeq
                    u32 const slot = nseq & (2u * G - 1u);
                    if ((u32)k == (slot >> 1)) { if (slot & 1u) sq1 = q; else sq0 = q; }
                    if (slot == 2u * G - 1u) kx_st128(seqs + (nseq - slot) + 2u * (u32)k, sq0, sq1);
                }
                if (ll > 0xFFFF) { longType = 1; longPos = nseq; }
                if (lenA - 3 > 0xFFFF) { longType = 2; longPos = nseq; }
                nseq++; nlit += (u32)ll;
                ip0 = m_start + (int)lenA; anchor = ip0;
                if (m_fill && k == 0) {
                    // the pair's other position, if the match has not swallowed it; then the fill: current0 + 2 and ip0 - 2
                    if (m_ip1 < ip0) H[m_hash1] = KFX_E(kx_ld32(src + m_ip1), m_ip1 + 2);
                    if (ip0 <= ilimit) {
                        u64 const wf0 = kx_ld64(src + m_cur0 + 2), wf1 = kx_ld64(src + ip0 - 2);
                        H[kx_hash_short_any(wf0, hlog, mls)] = KFX_E(wf0, m_cur0 + 2 + 2);
                        H[kx_hash_short_any(wf1, hlog, mls)] = KFX_E(wf1, ip0 - 2 + 2);
                    }
                }
                if (++guard > 600000u) { status = 2; state = KFX_CLEANUP; }
                else state = (ip0 <= ilimit) ? KFX_REPLOOP : KFX_START;
            }
        }

        // ================= finish the block ==================================
        if (kx_any(state == KFX_CLEANUP)) {
            if (state == KFX_CLEANUP) {
                {
                    u32 const cnt = nseq & (2u * G - 1u);
                    u64* const sp = (u64*)(seqs + (nseq - cnt));
                    if (2u * (u32)k < cnt) sp[2 * k] = sq0;
                    if (2u * (u32)k + 1u < cnt) sp[2 * k + 1] = sq1;
                }
                if (k == 0) {
                    KSliceMeta mm;
                    mm.nbSeq = nseq; mm.litSize = nlit; mm.lastLL = (u32)(n - anchor);
                    mm.longType = longType; mm.longPos = longPos; mm.status = status;
                    u32 const s2 = (saved1 != 0 && off1 != 0) ? saved1 : saved2;
                    mm.pad[0] = off1 ? off1 : saved1; mm.pad[1] = off2 ? off2 : s2;
                    a.meta[slice] = mm;
                }
                state = KFX_IDLE;
            }
        }
    }
#undef KFX_E
}
