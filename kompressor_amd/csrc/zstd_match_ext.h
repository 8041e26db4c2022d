// zstd_match_ext.h -- the level-3 parse of one block once the stream is longer than libzstd's staging buffer.
//
// The reference hands ZSTD_compressStream2 (kompressor-zstd--nativelib/src/jvmCommonMain/jni/Wrapper.cpp:112) output
// slices smaller than ZSTD_compressBound, so libzstd stages the input in a buffer of window + 128 KiB bytes
// (2 MiB + 128 KiB at level 3).  When that buffer wraps, the lap before becomes an older segment ("extDict") and
// libzstd 1.5.7 parses the following blocks with ZSTD_compressBlock_doubleFast_extDict_generic: a plain
// position-by-position loop (no pipelined look-ahead), whose results differ from the regular variant's where the segment
// boundary shows: a match found in the older segment does not grow backwards past its start, repcodes that would
// straddle the boundary are refused, the step grows with the distance from the last match.  The bytes of the stream are
// contiguous here (the slice lies in HBM as the caller gave it), so both segments are addressed through one pointer and
// ZSTD_count_2segments is an ordinary extension; only the index rules are kept.
//
// Block mode only (frames of several blocks, per-slice tables, KFrameState); one team of G lanes per slice, lane 0
// decides, the team extends matches -- the shape of zstd_match_dict.h.  Blocks the regular variant parses are skipped
// here and the other way round (kx_block_window says which).
#pragma once
#include "zstd_match.h"

enum { KXS_IDLE = 0, KXS_SEARCH = 1, KXS_REPLOOP = 2, KXS_MATCH = 3, KXS_CLEANUP = 4, KXS_DONE = 5 };

template <int G>
KX_DEV void zstd_match_ext_body(const KMatchArgs& a)
{
    int const lane = kx_lane();
    int const k = lane & (G - 1);
    int const tbase = lane - k;
    u64 const tmask = (G == 64) ? ~0ull : ((1ull << G) - 1ull);
    bool const wide = (a.flags & 16u) != 0;
    u32 const IDXM = wide ? 0xFFFFFFFFu : KX_BLK_IDX_MASK;
    constexpr u32 CHKS = KX_BLK_IDX_BITS;

    int state = KXS_IDLE;
    const u8* src = a.src; u32* L = a.big_tables; u32* S = a.big_tables; KSeq* seqs = a.seqs; u8* lits = a.lits;
    int n = 0, ilimit = 0, ip = 0, anchor = 0; u32 slice = 0; u32 off1 = 1, off2 = 4;
    u32 nseq = 0, nlit = 0, longType = 0, longPos = 0, guard = 0, status = 0; u32 hbL = 17, hbS = 16, mls = 5;
    u32 dsi = 2, psi = 2;                       // dictStartIndex, prefixStartIndex of the block
    u64 sq0 = 0, sq1 = 0;
    int m_start = 0, m_mpos = 0, m_low = 0, m_curr = 0; u32 m_len0 = 0, m_off = 0; bool m_back = false;

    for (;;) {
        // ================= next slice whose block is an extDict block ==================
        if (kx_any(state == KXS_IDLE)) {
            u32 s = 0;
            if (state == KXS_IDLE && k == 0) s = kx_atomic_add(a.counter, 1u);
            s = kx_shfl(s, tbase);
            if (state == KXS_IDLE) {
                if (s >= a.n_slices) state = KXS_DONE;
                else {
                    KFrameState const fs = a.fstate[s];
                    bool ok4 = true;
                    KParams P = (a.level == 4u) ? kx_params_l4(a.in_len[s], ok4) : kx_params_l3(a.in_len[s]);
                    if (a.flags & 8u) { P.windowLog = 21; P.chainLog = a.level == 4u ? 18 : 16; P.hashLog = a.level == 4u ? 18 : 17; P.minMatch = 5; }
                    KBlockWin const bw = kx_block_window(fs.lowLimit, fs.dictLimit, fs.ipos, fs.blockSize, P.windowLog);
                    if (fs.blockSize != 0 && bw.ext) {
                        slice = s;
                        src = a.src + a.in_off[s];
                        seqs = a.seqs + (size_t)s * a.seq_cap; lits = a.lits + (size_t)s * a.lit_cap;
                        L = a.big_tables + (size_t)s * a.big_stride; S = L + a.big_long;
                        hbL = P.hashLog; hbS = P.chainLog; mls = P.minMatch;
                        dsi = bw.dictStartIndex; psi = bw.prefixStartIndex;
                        nseq = 0; nlit = 0; longType = 0; longPos = 0; guard = 0; status = 0;
                        ip = (int)fs.ipos; anchor = ip; n = ip + (int)fs.blockSize; ilimit = n - 8;
                        off1 = fs.rep[0]; off2 = fs.rep[1];
                        state = (fs.blockSize < 8 || ip >= ilimit) ? KXS_CLEANUP : KXS_SEARCH;
                    }
                }
            }
        }
        if (kx_all(state == KXS_DONE)) break;

        // ================= one search position (lane 0 of the team decides) ==========
        if (kx_any(state == KXS_SEARCH)) {
            bool const srch = state == KXS_SEARCH;
            u32 kind = 0;                // 0 none, 1 rep at ip+1, 2 long at ip, 3 long at ip+1, 4 short at ip
            u32 mIdx = 0;
            if (srch && k == 0) {
                u64 const w0 = kx_ld64(src + ip);
                u64 const w1 = kx_ld64_clamped(src, ip + 1, n);
                u32 const hL = kx_hash_long(w0, hbL), hS = kx_hash_short(w0, hbS, mls);
                u32 const curr = (u32)ip + 2u;
                u32 const idxL = L[hL] & IDXM, idxS = S[hS] & IDXM;
                u32 const repIndex = curr + 1u - off1;
                // ZSTD_index_overlap_check(prefixStartIndex, repIndex) & (offset_1 <= curr + 1 - dictStartIndex)
                bool const repOk = ((u32)((psi - 1u) - repIndex) >= 3u) && off1 <= curr + 1u - dsi;
                L[hL] = curr | (wide ? 0u : kx_chk_long(w0, hbL) << CHKS);
                S[hS] = curr | (wide ? 0u : kx_chk_short(w0) << CHKS);
                bool const lok = idxL > dsi, sok = idxS > dsi;
                u32 const cR = kx_ld32(src + (repOk ? (int)repIndex - 2 : ip));
                u64 const cL = kx_ld64(src + (lok ? (int)idxL - 2 : ip));
                u32 const cS = kx_ld32(src + (sok ? (int)idxS - 2 : ip));
                if (repOk && cR == (u32)(w0 >> 8)) { kind = 1; mIdx = repIndex; }
                else if (lok && cL == w0) { kind = 2; mIdx = idxL; }
                else if (sok && cS == (u32)w0) {
                    // a long match at ip + 1 is preferred; that position is inserted either way
                    u32 const h3 = kx_hash_long(w1, hbL);
                    u32 idx3 = L[h3] & IDXM;
                    if (h3 == hL) idx3 = curr;                           // this step's own insert
                    L[h3] = (curr + 1u) | (wide ? 0u : kx_chk_long(w1, hbL) << CHKS);
                    if (idx3 > dsi && kx_ld64(src + (int)idx3 - 2) == w1) { kind = 3; mIdx = idx3; } else { kind = 4; mIdx = idxS; }
                }
            }
            kind = kx_shfl(kind, tbase); mIdx = kx_shfl(mIdx, tbase);
            if (srch) {
                guard++;
                if (kind == 0) {
                    ip += ((ip - anchor) >> 8) + 1;
                    if (ip >= ilimit) state = KXS_CLEANUP;
                    if (guard > 600000u) { status = 1; state = KXS_CLEANUP; }          // a block has at most 128 Ki positions
                } else {
                    m_curr = ip;
                    m_mpos = (int)mIdx - 2;
                    m_low = (int)(mIdx < psi ? dsi : psi) - 2;          // lowMatchPtr: the start of the match's own segment
                    u32 const curr = (u32)ip + 2u;
                    if (kind == 1) { m_start = ip + 1; m_len0 = 4; m_off = 0; m_back = false; }
                    else if (kind == 2) { m_start = ip; m_len0 = 8; m_off = curr - mIdx; m_back = true; }
                    else if (kind == 3) { m_start = ip + 1; m_len0 = 8; m_off = curr + 1u - mIdx; m_back = true; }
                    else { m_start = ip; m_len0 = 4; m_off = curr - mIdx; m_back = true; }
                    state = KXS_MATCH;
                }
            }
        }

        // ================= immediate repcode =================================
        if (kx_any(state == KXS_REPLOOP)) {
            bool const inrep = state == KXS_REPLOOP;
            bool hit = false; int rp = 0;
            if (inrep && ip <= ilimit) {
                u32 const current2 = (u32)ip + 2u;
                u32 const repIndex2 = current2 - off2;
                bool const ok = ((u32)((psi - 1u) - repIndex2) >= 3u) && off2 <= current2 - dsi;
                rp = (int)repIndex2 - 2;
                if (ok) hit = kx_ld32(src + rp) == kx_ld32(src + ip);
            }
            if (inrep) {
                if (hit) {
                    if (k == 0) {
                        u64 const w = kx_ld64(src + ip);
                        u32 const t = (u32)ip + 2u;
                        S[kx_hash_short(w, hbS, mls)] = t | (wide ? 0u : kx_chk_short(w) << CHKS);
                        L[kx_hash_long(w, hbL)] = t | (wide ? 0u : kx_chk_long(w, hbL) << CHKS);
                    }
                    u32 const tmp = off2; off2 = off1; off1 = tmp;
                    m_start = ip; m_mpos = rp; m_low = 0; m_len0 = 4; m_off = 0; m_back = false; m_curr = -1;      // -1: no complementary insertion
                    state = KXS_MATCH;
                } else state = (ip >= ilimit) ? KXS_CLEANUP : KXS_SEARCH;
            }
        }

        // ================= take the match ====================================
        if (kx_any(state == KXS_MATCH)) {
            bool const mt = state == KXS_MATCH;
            u32 lenA = kx_team_extend<G>(mt, src, n, m_start, m_mpos, m_len0, k, tbase, tmask);
            int const mb = (m_start - anchor < m_mpos - m_low) ? m_start - anchor : m_mpos - m_low;
            u32 const back = kx_team_backward<G>(mt && m_back, src, m_start, m_mpos, mb, k, tbase, tmask);
            if (mt) {
                u32 offBase = 1;
                if (m_back) { m_start -= (int)back; lenA += back; off2 = off1; off1 = m_off; offBase = m_off + 3; }
                int const ll = m_start - anchor;
                // the literals in front of the match (the block-chain kernel codes them from this buffer)
                if (!(a.flags & 4u)) for (int c = 8 * k; c < ll; c += 8 * G) kx_st64(lits + nlit + c, kx_ld64_clamped(src, anchor + c, n));
                {
                    u64 const q = (u64)offBase | ((u64)(u16)ll << 32) | ((u64)(u16)(lenA - 3) << 48);   // KSeq
                    u32 const slot = nseq & (2u * G - 1u);
                    if ((u32)k == (slot >> 1)) { if (slot & 1u) sq1 = q; else sq0 = q; }
                    if (slot == 2u * G - 1u) kx_st128(seqs + (nseq - slot) + 2u * (u32)k, sq0, sq1);
                }
                if (ll > 0xFFFF) { longType = 1; longPos = nseq; }
                if (lenA - 3 > 0xFFFF) { longType = 2; longPos = nseq; }
                nseq++; nlit += (u32)ll;
                ip = m_start + (int)lenA; anchor = ip;
                if (m_curr >= 0 && ip <= ilimit && k == 0) {
                    // complementary insertion: curr+2 into both tables, then ip-2 (long) and ip-1 (short)
                    u64 const wa = kx_ld64(src + m_curr + 2), wb = kx_ld64(src + ip - 2), wc = kx_ld64(src + ip - 1);
                    u32 const va = (u32)m_curr + 4u;
                    L[kx_hash_long(wa, hbL)] = va | (wide ? 0u : kx_chk_long(wa, hbL) << CHKS);
                    L[kx_hash_long(wb, hbL)] = (u32)ip | (wide ? 0u : kx_chk_long(wb, hbL) << CHKS);               // index of ip - 2
                    S[kx_hash_short(wa, hbS, mls)] = va | (wide ? 0u : kx_chk_short(wa) << CHKS);
                    S[kx_hash_short(wc, hbS, mls)] = ((u32)ip + 1u) | (wide ? 0u : kx_chk_short(wc) << CHKS);       // index of ip - 1
                }
                if (++guard > 600000u) { status = 2; state = KXS_CLEANUP; }
                else state = (ip <= ilimit) ? KXS_REPLOOP : KXS_CLEANUP;
            }
        }

        // ================= finish the block ==================================
        if (kx_any(state == KXS_CLEANUP)) {
            if (state == KXS_CLEANUP) {
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
                    mm.pad[0] = off1; mm.pad[1] = off2;                 // the extDict variant sets no repcode aside
                    a.meta[slice] = mm;
                }
                state = KXS_IDLE;
            }
        }
    }
}
