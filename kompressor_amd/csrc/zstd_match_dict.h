// zstd_match_dict.h -- the LZ stage of zstd level 3 when the context holds a raw-content dictionary
// (reference: ZstdCompressor(level, dictionary) -> loadCompressorDictionary, jni/Wrapper.cpp:41-56 =
// ZSTD_CCtx_loadDictionary, then the one-shot ZSTD_compressStream2 at :112).
//
// libzstd 1.5.7 turns the dictionary into a CDict (tables sized for the dictionary, built once: here by the host,
// kmp_api.hip) and then parses every input with one of two double-fast variants, both a plain position-by-position
// loop (no pipelined look-ahead as in the dictionary-less parser):
//   * input <= 16 KiB: the CDict stays attached; its (tagged) tables are consulted when the working tables miss
//     ("dictMatchState");
//   * input  > 16 KiB: the CDict's tables are copied into the working tables first ("extDict").  Here the copy is
//     never made: a working-table entry that is not from this slice falls back to the shared CDict table, which is
//     the same thing and keeps the dictionary's tables read-only, shared by all slices and cache resident.
// Index space as in libzstd: dictionary = indices [2, 2 + D), input from 2 + D.  Byte access goes through a virtual
// string V = dictionary ++ input, so a match that starts in the dictionary runs on into the input's start exactly
// like ZSTD_count_2segments.
//
// One team of G lanes per slice as in zstd_match.h; the search itself is not speculative (lane 0 decides), the team
// cooperates on match extension.  All cross-lane primitives are called from wave-uniform control flow.
#pragma once
#include "zstd_match.h"

struct KDictArgs {
    KMatchArgs m;                     // slices, sequences, meta, per-team working tables + epochs, work counter
    const u8* dict; u32 dict_size;    // D: 8 .. KX_MAX_DICT (a formatted dictionary's content part)
    u32 rep0 = 1, rep1 = 4;           // the repeat offsets a frame starts with (a formatted dictionary brings its own)
    const u32* dictL; const u32* dictS;   // CDict tables: entries index << 8 | tag, 1 << dHashLog / 1 << dChainLog of them
    u32 dWindowLog, dHashLog, dChainLog, dMinMatch;      // the CDict's parameters (ZSTD_getCParams for the dictionary alone)
};
#define KX_MAX_DICT (128u * 1024u - 512u)      /* keeps chainLog <= 15 and every index below 1 << KX_IDX_BITS */

struct KV { const u8* dict; int D; const u8* src; int n; };
// 8 bytes of V at virtual position p (0 <= p < D + n); bytes past the end read as zero
KX_DEV u64 kv_ld64(const KV& v, int p)
{
    if (p >= v.D) return kx_ld64_clamped(v.src, p - v.D, v.n);
    if (p + 8 <= v.D) return kx_ld64(v.dict + p);
    int const k = v.D - p;                                   // 1..7 bytes left in the dictionary
    u64 const lo = kx_ld64(v.dict + v.D - 8) >> (8 * (8 - k));
    u64 const hi = (v.n >= 8) ? kx_ld64(v.src) : 0ull;
    return lo | (hi << (8 * k));
}
KX_DEV u32 kv_ld32(const KV& v, int p) { return (u32)kv_ld64(v, p); }
KX_DEV u32 kv_byte(const KV& v, int p) { return p < v.D ? v.dict[p] : v.src[p - v.D]; }

// common prefix of V[s+len..) and V[m+len..), m < s, the s side ends at vend; 8 bytes per lane per round
template <int G>
KX_DEV u32 kv_team_extend(bool act, const KV& v, int vend, int s, int m, u32 len, int k, int tbase, u64 tmask)
{
    bool running = act;
    while (kx_any(running)) {
        u32 eq = 8;
        if (running) {
            int const p = s + (int)len + 8 * k, q = m + (int)len + 8 * k;
            int const avail = vend - p;
            if (avail <= 0) eq = 0;
            else {
                u64 const d = kv_ld64(v, p) ^ kv_ld64(v, q);
                u32 e = d ? (kx_ctz64(d) >> 3) : 8u;
                if (avail < 8 && e > (u32)avail) e = (u32)avail;
                eq = e;
            }
        }
        u64 const b = kx_ballot(running && eq < 8);
        u64 const tb = (b >> tbase) & tmask;
        int const f = tb ? (int)kx_ctz64(tb) : 0;
        u32 const eqf = kx_shfl(eq, tbase + f);
        if (running) {
            if (tb) { len += 8u * (u32)f + eqf; running = false; }
            else len += 8u * G;
        }
    }
    return len;
}
template <int G>
KX_DEV u32 kv_team_backward(bool act, const KV& v, int s, int m, int maxback, int k, int tbase, u64 tmask)
{
    u32 back = 0;
    bool running = act && maxback > 0;
    while (kx_any(running)) {
        bool ne = true;
        if (running) {
            int const o = (int)back + k;
            if (o < maxback) ne = kv_byte(v, s - 1 - o) != kv_byte(v, m - 1 - o);
        }
        u64 const b = kx_ballot(running && ne);
        u64 const tb = (b >> tbase) & tmask;
        if (running) {
            if (tb) { back += kx_ctz64(tb); running = false; }
            else back += G;
        }
    }
    return back;
}

enum { KDS_IDLE = 0, KDS_SEARCH = 1, KDS_REPLOOP = 2, KDS_MATCH = 3, KDS_CLEANUP = 4, KDS_DONE = 5 };

template <int G>
KX_DEV void zstd_match_dict_body(const KDictArgs& d)
{
    constexpr int NT = 64 / G;
    const KMatchArgs& a = d.m;
    int const lane = kx_lane();
    int const k = lane & (G - 1);
    int const tbase = lane - k;
    u32 const team = kx_block() * NT + (u32)(lane / G);
    u32* const L = kx_team_tables(a, team);
    u32* const S = L + KX_TBL_LONG;
    u64 const tmask = (G == 64) ? ~0ull : ((1ull << G) - 1ull);
    int const D = (int)d.dict_size;
    u32 const P = 2u + (u32)D;                            // index of the input's first byte

    int state = KDS_IDLE;
    KV v; v.dict = d.dict; v.D = D; v.src = a.src; v.n = 0;
    int n = 0, ilimit = 0; u32 slice = 0; bool attach = false;
    int ip = 0, anchor = 0; u32 off1 = 1, off2 = 4; u32 nseq = 0, nlit = 0, tag = 0; u32 hbL = 16, hbS = 15, mls = 5;
    u32 longType = 0, longPos = 0, guard = 0, status = 0;
    KSeq* seqs = a.seqs; u64 sq0 = 0, sq1 = 0;
    // pending match (virtual positions)
    int m_start = 0, m_mv = 0, m_low = 0, m_curr = 0; u32 m_len0 = 0, m_off = 0; bool m_back = false;

    for (;;) {
        // ================= next slice ==================================
        if (kx_any(state == KDS_IDLE)) {
            u32 s = 0, ep = 0;
            if (state == KDS_IDLE && k == 0) {
                s = kx_atomic_add(a.counter, 1u);
                if (s < a.n_slices) {
                    ep = a.team_epoch[team] + 1;
                    if (ep > KX_EPOCH_MAX) ep = 0;
                    a.team_epoch[team] = ep ? ep : 1u;
                }
            }
            s = kx_shfl(s, tbase); ep = kx_shfl(ep, tbase);
            if (state == KDS_IDLE) {
                if (s >= a.n_slices) state = KDS_DONE;
                else {
                    slice = s;
                    v.src = a.src + a.in_off[s]; n = (int)a.in_len[s]; v.n = n;
                    seqs = a.seqs + (size_t)s * a.seq_cap;
                    attach = n <= 16 * 1024;                     // attachDictSizeCutoffs[ZSTD_dfast]
                    hbL = d.dHashLog; hbS = d.dChainLog; mls = d.dMinMatch;
                    if (attach) {
                        // working tables resized for the input alone (ZSTD_adjustCParams_internal, attach mode)
                        u32 const srcLog = (n < 64) ? 6u : kx_hb32((u32)n - 1u) + 1u;
                        u32 const W = d.dWindowLog < srcLog ? d.dWindowLog : srcLog;
                        if (hbL > W + 1) hbL = W + 1;
                        if (hbS > W) hbS = W;
                    }
                    nseq = 0; nlit = 0; longType = 0; longPos = 0; guard = 0; status = 0;
                    if (ep == 0) {
                        for (u32 i = (u32)k; i < KX_TBL_ENTRIES; i += G) L[i] = 0;
                        ep = 1;
                    }
                    tag = ep << KX_TAG_SHIFT;
                    anchor = 0; ip = 0; ilimit = n - 8; off1 = d.rep0; off2 = d.rep1;
                    state = (n < 8 || ip >= ilimit) ? KDS_CLEANUP : KDS_SEARCH;
                }
            }
        }
        if (kx_all(state == KDS_DONE)) break;

        // ================= one search position (lane 0 of the team decides) ==========
        if (kx_any(state == KDS_SEARCH)) {
            bool const srch = state == KDS_SEARCH;
            u32 kind = 0;                // 0 none, 1 rep at ip+1, 2 long at ip, 3 long at ip+1, 4 short at ip
            u32 mIdx = 0;                // index of the match start
            if (srch && k == 0) {
                int const sv = D + ip;
                u64 const w0 = kv_ld64(v, sv);
                u64 const w1 = kv_ld64(v, sv + 1);
                u32 const hL = kx_hash_long(w0, hbL), hS = kx_hash_short(w0, hbS, mls);
                u32 const curr = P + (u32)ip;
                // Every table slot of this position is requested at once -- the slice's working tables and the CDict's -- and,
                // once they are known, the candidates' bytes (below): the step is a chain of memory latencies.  The decisions
                // keep libzstd's order.
                u32 const eL = L[hL], eS = S[hS];
                u32 hl = 0, hs = 0;
                if (attach) { hl = kx_hash_long(w0, d.dHashLog + 8); hs = kx_hash_short(w0, d.dChainLog + 8, mls); }
                u32 const xL = attach ? d.dictL[hl >> 8] : d.dictL[hL];
                u32 const xS = attach ? d.dictS[hs >> 8] : d.dictS[hS];
                u32 idxL = ((eL & KX_TAG_MASK) == tag) ? (eL & KX_IDX_MASK) : 0u;
                u32 idxS = ((eS & KX_TAG_MASK) == tag) ? (eS & KX_IDX_MASK) : 0u;
                // dictionary side
                u32 dIdxL = 0, dIdxS = 0; bool dTagL = false, dTagS = false;
                if (attach) {
                    dTagL = (xL & 0xFFu) == (hl & 0xFFu); dTagS = (xS & 0xFFu) == (hs & 0xFFu);
                    dIdxL = xL >> 8; dIdxS = xS >> 8;
                } else {
                    if (idxL == 0) idxL = xL >> 8;       // "copied" tables: the CDict entry until this slice overwrites it
                    if (idxS == 0) idxS = xS >> 8;
                }
                // repcode at ip + 1
                u32 const repIndex = curr + 1u - off1;
                // ZSTD_index_overlap_check + the lower bound (attach mode: libzstd only asserts it, it always holds)
                bool const repOk = ((u32)((P - 1u) - repIndex) >= 3u) && off1 <= curr - 1u;
                u32 const t = tag | curr;
                L[hL] = t; S[hS] = t;
                // copied tables (slices above 16 KiB): the three candidates' bytes are requested together; attached CDict
                // (small slices, two candidates per table): one after the other, as measured faster there
                bool const l1ok = !attach && idxL > 2u, s1ok = !attach && idxS > 2u;
                u32 cR = 0, cS1 = 0; u64 cL1 = 0;
                if (!attach) {
                    cR = kv_ld32(v, repOk ? (int)repIndex - 2 : 0);
                    cL1 = kv_ld64(v, l1ok ? (int)idxL - 2 : 0);
                    cS1 = kv_ld32(v, s1ok ? (int)idxS - 2 : 0);
                } else if (repOk) cR = kv_ld32(v, (int)repIndex - 2);
                if (repOk && cR == (u32)(w0 >> 8)) { kind = 1; mIdx = repIndex; }
                else {
                    bool longHit = false;
                    if (attach) {
                        if (idxL >= P && kv_ld64(v, (int)idxL - 2) == w0) { longHit = true; mIdx = idxL; }
                        else if (dTagL && dIdxL > 2u && kv_ld64(v, (int)dIdxL - 2) == w0) { longHit = true; mIdx = dIdxL; }
                    } else if (l1ok && cL1 == w0) { longHit = true; mIdx = idxL; }
                    if (longHit) kind = 2;
                    else {
                        bool shortHit = false; u32 sIdx = 0;
                        if (attach) {
                            if (idxS > P) { if (kv_ld32(v, (int)idxS - 2) == (u32)w0) { shortHit = true; sIdx = idxS; } }
                            else if (dTagS && dIdxS > 2u && kv_ld32(v, (int)dIdxS - 2) == (u32)w0) { shortHit = true; sIdx = dIdxS; }
                        } else if (s1ok && cS1 == (u32)w0) { shortHit = true; sIdx = idxS; }
                        if (shortHit) {
                            // look for a long match at ip + 1 first (and always insert that position)
                            u32 const h3 = kx_hash_long(w1, hbL);
                            u32 const e3 = L[h3];
                            u32 idx3 = ((e3 & KX_TAG_MASK) == tag) ? (e3 & KX_IDX_MASK) : 0u;
                            if (h3 == hL) idx3 = curr;                       // this step's own insert
                            bool hit3 = false; u32 m3 = 0;
                            if (attach) {
                                u32 const hl3 = kx_hash_long(w1, d.dHashLog + 8); u32 const x3 = d.dictL[hl3 >> 8];
                                if (idx3 >= P && kv_ld64(v, (int)idx3 - 2) == w1) { hit3 = true; m3 = idx3; }
                                else if ((x3 & 0xFFu) == (hl3 & 0xFFu) && (x3 >> 8) > 2u && kv_ld64(v, (int)(x3 >> 8) - 2) == w1) { hit3 = true; m3 = x3 >> 8; }
                            } else {
                                if (idx3 == 0) idx3 = d.dictL[h3] >> 8;
                                if (idx3 > 2u && kv_ld64(v, (int)idx3 - 2) == w1) { hit3 = true; m3 = idx3; }
                            }
                            L[h3] = tag | (curr + 1u);
                            if (hit3) { kind = 3; mIdx = m3; } else { kind = 4; mIdx = sIdx; }
                        }
                    }
                }
            }
            kind = kx_shfl(kind, tbase); mIdx = kx_shfl(mIdx, tbase);
            if (srch) {
                guard++;
                if (kind == 0) {
                    ip += ((ip - anchor) >> 8) + 1;
                    if (ip >= ilimit) state = KDS_CLEANUP;
                    if (guard > 2u * (u32)n + 64u) { status = 1; state = KDS_CLEANUP; }
                } else {
                    m_curr = ip;
                    m_mv = (int)mIdx - 2;
                    m_low = (m_mv >= D) ? D : 0;
                    if (kind == 1) { m_start = ip + 1; m_len0 = 4; m_off = 0; m_back = false; }
                    else if (kind == 2) { m_start = ip; m_len0 = 8; m_off = (P + (u32)ip) - mIdx; m_back = true; }
                    else if (kind == 3) { m_start = ip + 1; m_len0 = 8; m_off = (P + (u32)ip + 1u) - mIdx; m_back = true; }
                    else { m_start = ip; m_len0 = 4; m_off = (P + (u32)ip) - mIdx; m_back = true; }
                    state = KDS_MATCH;
                }
            }
        }

        // ================= immediate repcode =================================
        if (kx_any(state == KDS_REPLOOP)) {
            bool const inrep = state == KDS_REPLOOP;
            bool hit = false; int rv = 0;
            if (inrep && ip <= ilimit) {
                u32 const current2 = P + (u32)ip;
                u32 const repIndex2 = current2 - off2;
                bool ok = ((u32)((P - 1u) - repIndex2) >= 3u) && off2 <= current2 - 2u;
                rv = (int)repIndex2 - 2;
                if (ok) hit = kv_ld32(v, rv) == kv_ld32(v, D + ip);
            }
            if (inrep) {
                if (hit) {
                    if (k == 0) {
                        u64 const w = kv_ld64(v, D + ip);
                        u32 const t = tag | (P + (u32)ip);
                        S[kx_hash_short(w, hbS, mls)] = t; L[kx_hash_long(w, hbL)] = t;
                    }
                    u32 const tmp = off2; off2 = off1; off1 = tmp;
                    m_start = ip; m_mv = rv; m_low = 0; m_len0 = 4; m_off = 0; m_back = false; m_curr = -1;      // -1: no complementary insertion
                    state = KDS_MATCH;
                } else state = (ip >= ilimit) ? KDS_CLEANUP : KDS_SEARCH;
            }
        }

        // ================= take the match ====================================
        if (kx_any(state == KDS_MATCH)) {
            bool const mt = state == KDS_MATCH;
            u32 lenA = kv_team_extend<G>(mt, v, D + n, D + m_start, m_mv, m_len0, k, tbase, tmask);
            int const mb = (m_start - anchor < m_mv - m_low) ? m_start - anchor : m_mv - m_low;
            u32 const back = kv_team_backward<G>(mt && m_back, v, D + m_start, m_mv, mb, k, tbase, tmask);
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
                ip = m_start + (int)lenA; anchor = ip;
                if (m_curr >= 0 && ip <= ilimit && k == 0) {
                    // complementary insertion: curr+2 into both tables, then ip-2 (long) and ip-1 (short)
                    u64 const wa = kv_ld64(v, D + m_curr + 2), wb = kv_ld64(v, D + ip - 2), wc = kv_ld64(v, D + ip - 1);
                    u32 const va = tag | (P + (u32)m_curr + 2u);
                    L[kx_hash_long(wa, hbL)] = va;
                    L[kx_hash_long(wb, hbL)] = tag | (P + (u32)ip - 2u);
                    S[kx_hash_short(wa, hbS, mls)] = va;
                    S[kx_hash_short(wc, hbS, mls)] = tag | (P + (u32)ip - 1u);
                }
                if (++guard > 2u * (u32)n + 64u) { status = 2; state = KDS_CLEANUP; }
                else state = (ip <= ilimit) ? KDS_REPLOOP : KDS_CLEANUP;
            }
        }

        // ================= finish the slice ==================================
        if (kx_any(state == KDS_CLEANUP)) {
            if (state == KDS_CLEANUP) {
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
                    a.meta[slice] = mm;
                }
                state = KDS_IDLE;
            }
        }
    }
}
