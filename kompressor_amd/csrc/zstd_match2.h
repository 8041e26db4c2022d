// zstd_match2.h -- the level-3 ("double-fast") parse of zstd_match.h as a SPLIT-PHASE stage machine.
//
// Same decisions as zstd_match.h (and so as the third-party libzstd 1.5.7 behind the reference's
// ZSTD_compressStream2 call, kompressor-zstd--nativelib/src/jvmCommonMain/jni/Wrapper.cpp:112): teams of G lanes,
// one slice per team, G-1 search positions speculated per step, epoch-tagged tables in HBM.  What differs is how a
// wave spends its time.  zstd_match.h runs its blocks (repcode check, search, match, ...) one after the other and
// each block waits for its own dependent loads: one outer iteration is ~10 memory round trips in which every team
// advances one phase.  Here an outer iteration is
//
//      TOP      every team computes the addresses of whatever stage it is in and ISSUES its loads
//      (one wait)
//      CONSUME  every team takes its loads and moves to its next stage
//
// so an iteration costs ONE round trip and every team advances in every iteration.  To get there:
//   * the bytes at and ahead of the current position come from a per-team window in LDS (a ring of R bytes that the
//     team's lanes refill 16 bytes each, a line or two per iteration, ahead of the parse): hashing the search
//     positions, the immediate-repcode test, the complementary inserts and the first 16 bytes of every match
//     extension need no trip to memory, and the source is fetched once instead of once per step;
//   * the table probes of a step and its repcode candidates are one stage (SRCH); the candidates' bytes a second one
//     (CAND) that is skipped when no probed entry survives its epoch / check bits (then the step is decided by the
//     repcode bytes alone: a step without a hit is one round trip);
//   * the immediate repcode test after a match rides on the next step's probes (they are discarded if it hits);
//   * candidates arrive with 16 bytes forward and 8 bytes backward, so most matches are measured without another
//     trip; longer ones take EXT rounds of 16 bytes per lane, longer backward runs BACK rounds.
// Steps too wide for the window (incompressible data: the step grows by one per 256 bytes without a match) fetch
// their positions from memory first (WLOAD).
//
// All cross-lane primitives are called from wave-uniform control flow.
#pragma once
#include "zstd_match.h"

enum { K2_IDLE = 0, K2_SRCH = 1, K2_CAND = 2, K2_EXT = 3, K2_BACK = 4, K2_CLEANUP = 5, K2_DONE = 6,
       K2_WLOAD = 7, K2_STALL = 8 };      // (the last two only as "what this iteration does", never as a team's state)

struct K2W { u64 lo, hi; };               // sixteen bytes

// equal leading bytes of two 16-byte strings (0 .. 16)
KX_DEV u32 k2_eq16(K2W a, K2W b)
{
    u64 const dl = a.lo ^ b.lo, dh = a.hi ^ b.hi;
    return dl ? (kx_ctz64(dl) >> 3) : (dh ? 8u + (kx_ctz64(dh) >> 3) : 16u);
}
// equal trailing bytes of two 8-byte strings (0 .. 8)
KX_DEV u32 k2_eqback8(u64 a, u64 b) { u64 const d = a ^ b; return d ? (kx_clz64(d) >> 3) : 8u; }

// Sixteen bytes at src[p ..) (0 <= p < n, n >= 8) without touching a byte at or beyond n: the body loads the two halves from
// min(p, n - 8) and min(p + 8, n - 8) (the second only if p + 8 < n); after the wave's one wait k2_fix16 shifts what a clamped
// load brought into place.  Bytes at or beyond n come out as anything; callers cap by n.
KX_DEV K2W k2_fix16(K2W v, int p, int n)
{
    if (p + 16 > n) {
        if (p + 8 > n) v.lo >>= 8 * (p + 8 - n);
        if (p + 8 < n) v.hi >>= 8 * (p + 16 - n);
    }
    return v;
}
// The eight bytes that END at src[p) (p >= 1, n >= 8), nothing below src[0) touched; bytes below 0 come out as zero.
KX_DEV u64 k2_ldb8(const u8* src, int p) { return kx_ld64(src + (p >= 8 ? p - 8 : 0)); }
KX_DEV u64 k2_fixb8(u64 v, int p) { return p >= 8 ? v : v << (8 * (8 - p)); }

template <int G, int R = 256>
KX_DEV void zstd_match2_body(const KMatchArgs& a)
{
    static_assert(G == 2 || G == 4 || G == 8, "team width");
    constexpr int NT = 64 / G;
    constexpr int FB = 16 * G;                       // bytes one refill round brings (16 per lane)
    constexpr int RS = R + 16;                       // the ring and a copy of its first 16 bytes behind it (reads never wrap)
    constexpr u32 lowIdx = 2u;
    KX_SHARED KxQuad ring_store[NT * RS / 16];
    int const lane = kx_lane();
    int const k = lane & (G - 1);
    int const tbase = lane - k;
    u8* const ring = (u8*)ring_store + (lane / G) * RS;
    u32 const team = kx_block() * NT + (u32)(lane / G);
    u32* const L = kx_team_tables(a, team);
    u32* const S = L + KX_TBL_LONG;
    u64 const tmask = (1ull << G) - 1ull;
    bool const nt_st = (a.flags & 2u) != 0;

    // ---- team state (uniform across the team's lanes) -------------------
    int state = K2_IDLE;
    const u8* src = a.src; int n = 0; int ilimit = 0; u32 slice = 0;
    int ip = 0, anchor = 0; u32 off1 = 0, off2 = 0; int step = 1; int nextStep = 0;
    u32 nseq = 0, nlit = 0; u32 tag = 0; u32 hbL = 16, hbS = 15, mls = 5;
    u32 longType = 0, longPos = 0; u32 guard = 0; u32 status = 0;
    KSeq* seqs = a.seqs;
    u64 sq0 = 0, sq1 = 0;
    int wlo = 0, whi = 0;                            // the window holds src[max(wlo, whi - R), whi)
    bool chk0 = false;                               // a match just ended at ip: the immediate repcode test is due
    bool compl_due = false; u64 wa = 0; int c_pos = 0;   // ... and its complementary inserts (wa = the bytes at c_pos + 2)
    bool have_w = false;                             // wide step: the positions' bytes were fetched (WLOAD)
    // the match being measured
    int m_type = 0, m_pos = 0, m_start = 0, m_mpos = 0; u32 lenA = 0, m_off = 0;
    bool l1ok = false; int s1 = 0, m1 = 0; u32 lenB = 0; bool openB = false; int ext_b = 0;
    u32 backA = 0, backB = 0;                        // bytes already matched backwards | 0x100 if more may follow
    u32 back = 0;
    // ---- per lane, kept from SRCH to CAND --------------------------------
    u64 w = 0, w_hi = 0; u32 hl = 0, hs = 0, idxl = 0, idxs = 0; int predL = -1, predS = -1; u32 rlen = 0; bool repHit = false; int lim = 0;

    for (;;) {
        // ================= fetch the next slice (rare: its own round trip) =======================
        if (kx_any(state == K2_IDLE)) {
            u32 s = 0, ep = 0;
            if (state == K2_IDLE && k == 0) {
                s = kx_atomic_add(a.counter, 1u);
                if (s < a.n_slices) {
                    ep = a.team_epoch[team] + 1;
                    if (ep > KX_EPOCH_MAX) ep = 0;          // 0 = "clear the tables, restart at 1"
                    a.team_epoch[team] = ep ? ep : 1u;
                }
            }
            s = kx_shfl(s, tbase); ep = kx_shfl(ep, tbase);
            if (state == K2_IDLE) {
                if (s >= a.n_slices) state = K2_DONE;
                else {
                    slice = s;
                    src = a.src + a.in_off[s];
                    n = (int)a.in_len[s];
                    seqs = a.seqs + (size_t)s * a.seq_cap;
                    KParams const P = kx_params_l3((u32)n);
                    hbL = P.hashLog; hbS = P.chainLog; mls = P.minMatch;
                    nseq = 0; nlit = 0; longType = 0; longPos = 0; guard = 0; status = 0;
                    if (ep == 0) {
                        for (u32 i = (u32)k; i < KX_TBL_ENTRIES; i += G) L[i] = 0;
                        ep = 1;
                    }
                    tag = ep << KX_TAG_SHIFT;
                    anchor = 0; ilimit = n - 8;
                    ip = 1; off1 = 1; off2 = 0;     // rep {1,4,8}: 4 exceeds the 1 byte of history at ip=1
                    step = 1; nextStep = ip + 256;
                    wlo = 0; whi = 0; chk0 = false; compl_due = false; have_w = false;
                    state = (n < 8 || ip + 1 > ilimit) ? K2_CLEANUP : K2_SRCH;
                }
            }
        }
        if (kx_all(state == K2_DONE)) break;
        if (lane == 0) KX_STAT(0, 1);

        // =====================================================================================
        // TOP: what does this iteration do for the team, and its loads
        // =====================================================================================
        int const pos = ip + k * step;
        bool const cand = (k < G - 1) && (k == 0 || pos < nextStep) && (pos + step <= ilimit);
        bool const prov = k == 0 || ((k == 1 || pos - step < nextStep) && pos <= ilimit);
        int st0 = state;
        int const lo_need = ip >= 2 ? ip - 2 : 0;
        bool const far = (G - 1) * step + 16 + 2 + 64 + FB > R;
        if (state == K2_SRCH) {
            if (far && !compl_due && !chk0) { if (!have_w) st0 = K2_WLOAD; }
            else {
                if (lo_need >= whi) { wlo = lo_need & ~63; whi = wlo; }       // the parse left the window behind: start it again
                int const vlo = (wlo > whi - R) ? wlo : whi - R;
                int const last = ip + (G - 1) * step + 16;
                if (lo_need < vlo || (whi < n && whi < last)) st0 = K2_STALL;
            }
        }
        // Every load slot below has ONE issue site, fed by per-state address arithmetic: two pending loads into the same
        // registers (a write after a write) would make the compiler wait between them, and the iteration would no longer
        // be one round trip.  A 16-byte slot is two 8-byte loads whose addresses are clamped to the slice (k2_fix16 shifts
        // the bytes into place after the wait).
        // ---- window refill: up to two rounds of 16 bytes per lane ----
        int const fpos = whi + 16 * k;
        bool const fill0 = state != K2_DONE && state != K2_CLEANUP && whi < n && whi + FB <= lo_need + R;
        bool const fill1 = fill0 && whi + FB < n && whi + 2 * FB <= lo_need + R;
        int f0p = (fill0 && fpos < n) ? fpos : -1, f1p = (fill1 && fpos + FB < n) ? fpos + FB : -1;
        int x0p = -1, x1p = -1, y0p = 0, y1p = 0;               // 16 bytes at x?p (-1: none); the 8 bytes that end at y?p (0: none)
        bool p0v = false, p1v = false;
        if (st0 == K2_WLOAD) {
            if (prov) x0p = pos;
        } else if (st0 == K2_SRCH) {
            if (compl_due) {
                // complementary insertion: curr+2 into both tables, then ip-2 (long) and ip-1 (short)
                if (k == 0) {
                    u64 const wb = kx_ld64(ring + ((ip - 2) & (R - 1)));
                    u64 const wc = kx_ld64(ring + ((ip - 1) & (R - 1)));
                    u32 const va = tag | (u32)(c_pos + 2 + 2);
                    u32 const e0 = va | kx_chk_long(wa, hbL) << KX_CHK_SHIFT, e1 = (tag | (u32)(ip - 2 + 2)) | kx_chk_long(wb, hbL) << KX_CHK_SHIFT;
                    u32 const e2 = va | kx_chk_short(wa) << KX_CHK_SHIFT, e3 = (tag | (u32)(ip - 1 + 2)) | kx_chk_short(wc) << KX_CHK_SHIFT;
                    L[kx_hash_long(wa, hbL)] = e0; L[kx_hash_long(wb, hbL)] = e1;
                    S[kx_hash_short(wa, hbS, mls)] = e2; S[kx_hash_short(wc, hbS, mls)] = e3;
                }
                compl_due = false;
            }
            if (!have_w && prov) { int const o = pos & (R - 1); w = kx_ld64(ring + o); w_hi = kx_ld64(ring + o + 8); }
            if (prov) { hl = kx_hash_long(w, hbL); hs = kx_hash_short(w, hbS, mls); p0v = true; p1v = cand; }
            if (cand && off1 > 0) x0p = pos + 1 - (int)off1;
            if (chk0 && k == 0 && off2 > 0) x1p = ip - (int)off2;
        } else if (st0 == K2_CAND) {
            if (idxl >= lowIdx && k <= lim + 1) { x0p = (int)idxl - 2; y0p = x0p; }
            if (cand && idxs >= lowIdx && k <= lim) { x1p = (int)idxs - 2; y1p = x1p; }
        } else if (st0 == K2_EXT) {
            int const es_ = (ext_b ? s1 + (int)lenB : m_start + (int)lenA) + 16 * k;
            if (es_ < n) { x0p = es_; x1p = (ext_b ? m1 + (int)lenB : m_mpos + (int)lenA) + 16 * k; }
        } else if (st0 == K2_BACK) {
            int const bs_ = m_start - (int)back - 8 * k, bm_ = m_mpos - (int)back - 8 * k;
            if (bs_ > 0 && bm_ > 0) { y0p = bs_; y1p = bm_; }
        }
        K2W F0, F1, X0, X1; F0.lo = F0.hi = F1.lo = F1.hi = X0.lo = X0.hi = X1.lo = X1.hi = 0; u32 P0 = 0, P1 = 0; u64 Y0 = 0, Y1 = 0;
        int const n8 = n - 8;
        if (f0p >= 0) { F0.lo = kx_ld64(src + (f0p < n8 ? f0p : n8)); if (f0p + 8 < n) F0.hi = kx_ld64(src + (f0p + 8 < n8 ? f0p + 8 : n8)); }
        if (f1p >= 0) { F1.lo = kx_ld64(src + (f1p < n8 ? f1p : n8)); if (f1p + 8 < n) F1.hi = kx_ld64(src + (f1p + 8 < n8 ? f1p + 8 : n8)); }
        if (p0v) P0 = L[hl];
        if (p1v) P1 = S[hs];
        if (x0p >= 0) { X0.lo = kx_ld64(src + (x0p < n8 ? x0p : n8)); if (x0p + 8 < n) X0.hi = kx_ld64(src + (x0p + 8 < n8 ? x0p + 8 : n8)); }
        if (x1p >= 0) { X1.lo = kx_ld64(src + (x1p < n8 ? x1p : n8)); if (x1p + 8 < n) X1.hi = kx_ld64(src + (x1p + 8 < n8 ? x1p + 8 : n8)); }
        if (y0p > 0) Y0 = k2_ldb8(src, y0p);
        if (y1p > 0) Y1 = k2_ldb8(src, y1p);

        // =====================================================================================
        // CONSUME (the first use of a loaded value below is where the wave waits, once)
        // =====================================================================================
        // ---- window refill lands ----
        if (fill0) {
            if (fpos < n) {
                K2W const f = k2_fix16(F0, fpos, n); int const o = fpos & (R - 1);
                kx_st128(ring + o, f.lo, f.hi); if (o == 0) kx_st128(ring + R, f.lo, f.hi);
            }
            if (fill1 && fpos + FB < n) {
                K2W const f = k2_fix16(F1, fpos + FB, n); int const o = (fpos + FB) & (R - 1);
                kx_st128(ring + o, f.lo, f.hi); if (o == 0) kx_st128(ring + R, f.lo, f.hi);
            }
            whi += fill1 ? 2 * FB : FB;
        }
        kx_lockstep();
        bool const active = st0 == K2_SRCH || st0 == K2_CAND || st0 == K2_EXT || st0 == K2_BACK || st0 == K2_WLOAD || st0 == K2_STALL;
        if (active && ++guard > 8u * (u32)n + 4096u) { status = 1; state = K2_CLEANUP; st0 = K2_CLEANUP; }

        if (st0 == K2_WLOAD) {
            if (prov) { K2W const v = k2_fix16(X0, pos, n); w = v.lo; w_hi = v.hi; }
            have_w = true;
        }

        bool do_resolve = false, do_choose = false, do_final = false;
        bool longHit = false, shortHit = false, longRaw = false;
        u32 lenL = 0, lenS = 0, bkL = 0, bkS = 0;            // measured so far | 0x100 if the measure is open at its end
        int const avail = n - pos;                             // bytes from this lane's position to the end

        // ---- SRCH: the probes (and the repcode bytes) are here ----
        if (kx_any(st0 == K2_SRCH)) {
            if (lane == 0) KX_STAT(1, 1);
            bool const srch = st0 == K2_SRCH;
            // immediate repcode at ip (a match just ended there)
            bool hit0 = false; u32 len0 = 0;
            if (srch && chk0 && k == 0 && off2 > 0) {
                K2W const c = k2_fix16(X1, ip - (int)off2, n);
                if ((u32)c.lo == (u32)w) {
                    hit0 = true;
                    K2W ww; ww.lo = w; ww.hi = w_hi;
                    u32 e = k2_eq16(ww, c);
                    if ((int)e >= avail) e = (u32)avail; else if (e == 16) e |= 0x100u;
                    len0 = e;
                    u32 const v = tag | (u32)(ip + 2);
                    S[hs] = v | kx_chk_short(w) << KX_CHK_SHIFT;
                    L[hl] = v | kx_chk_long(w, hbL) << KX_CHK_SHIFT;
                }
            }
            hit0 = kx_shfl((u32)hit0, tbase) != 0u; len0 = kx_shfl(len0, tbase);
            int const K = (int)kx_popc64((kx_ballot(srch && cand) >> tbase) & tmask);
            // the table entries as lane k finds them: epoch, check bits, and what lanes < k of this step insert first
            u32 const ckl = kx_chk_long(w, hbL) << KX_CHK_SHIFT, cks = kx_chk_short(w) << KX_CHK_SHIFT;
            u32 il = ((P0 & KX_TAG_MASK) == tag && (P0 & KX_CHK_MASK) == ckl) ? (P0 & KX_IDX_MASK) : 0u;
            u32 is = ((P1 & KX_TAG_MASK) == tag && (P1 & KX_CHK_MASK) == cks) ? (P1 & KX_IDX_MASK) : 0u;
            u32 const hpack = hl | (hs << 16);
            int pL = -1, pS = -1;
#pragma unroll
            for (int d = 1; d < G; d++) {
                bool const ok = srch && prov && k >= d;
                u32 const hp = kx_shfl(hpack, lane - d);
                if (ok && pL < 0 && (hp & 0xFFFFu) == hl) pL = k - d;
                if (ok && pS < 0 && (hp >> 16) == hs) pS = k - d;
            }
            if (pL >= 0) il = (u32)(pos - (k - pL) * step) + 2u;
            if (pS >= 0) is = (u32)(pos - (k - pS) * step) + 2u;
            bool rh = false; u32 rl = 0;
            if (srch && cand && off1 > 0) {
                K2W const c = k2_fix16(X0, pos + 1 - (int)off1, n);
                K2W ww; ww.lo = (w >> 8) | (w_hi << 56); ww.hi = w_hi >> 8;         // the 15 bytes from pos + 1 on
                if ((u32)c.lo == (u32)ww.lo) {
                    rh = true;
                    u32 e = k2_eq16(ww, c); if (e > 15) e = 15;                             // (ww holds 15 bytes)
                    if ((int)e >= avail - 1) e = (u32)(avail - 1); else if (e == 15) e = 15u | 0x100u;
                    rl = e;
                }
            }
            u64 const cmask = (kx_ballot(srch && ((cand && (il >= lowIdx || is >= lowIdx)))) >> tbase) & tmask;
            u64 const rmask = (kx_ballot(rh) >> tbase) & tmask;
            if (srch) {
                if (hit0) {
                    // the repcode wins: the step that was probed alongside never happened
                    m_type = KMT_REP0; m_pos = ip; m_start = ip; m_mpos = ip - (int)off2; lenA = len0 & 0xFFu; l1ok = false; chk0 = false;
                    if (len0 & 0x100u) { state = K2_EXT; ext_b = 0; } else do_final = true;
                } else if (K == 0) { chk0 = false; state = K2_CLEANUP; }
                else {
                    chk0 = false;
                    idxl = il; idxs = is; predL = pL; predS = pS; repHit = rh; rlen = rl;
                    int const fR = rmask ? (int)kx_ctz64(rmask) : G, fC = cmask ? (int)kx_ctz64(cmask) : G;
                    lim = rmask ? fR : K - 1;
                    if (fC < fR) state = K2_CAND;               // a table candidate stands before the first repcode hit: its bytes decide
                    else do_resolve = true;
                }
            }
        }

        // ---- CAND: the candidates' bytes are here ----
        if (kx_any(st0 == K2_CAND)) {
            if (lane == 0) KX_STAT(2, 1);
            if (st0 == K2_CAND) {
                int const vlo = (wlo > whi - R) ? wlo : whi - R;
                int sb_n = have_w ? 0 : pos - vlo; if (sb_n > 8) sb_n = 8; if (sb_n < 0) sb_n = 0;       // bytes before pos the window still holds
                u64 const sb = kx_ld64(ring + ((pos - 8) & (R - 1)));
                K2W ww; ww.lo = w; ww.hi = w_hi;
                int const mbs = pos - anchor;
                if (idxl >= lowIdx && k <= lim + 1) {
                    int const c = (int)idxl - 2;
                    K2W const f = k2_fix16(X0, c, n);
                    if (f.lo == w) {
                        longRaw = true; longHit = cand;
                        u32 e = k2_eq16(ww, f);
                        if ((int)e >= avail) e = (u32)avail; else if (e == 16) e |= 0x100u;
                        lenL = e;
                        int const mb = mbs < c ? mbs : c;                   // how far the match may grow backwards
                        if (mb > 0) {
                            int const cmpn = mb < sb_n ? mb : sb_n;
                            u32 b = cmpn > 0 ? k2_eqback8(sb, k2_fixb8(Y0, c)) : 0u;
                            if ((int)b >= cmpn) { b = (u32)cmpn; if (cmpn < mb) b |= 0x100u; }
                            bkL = b;
                        }
                    }
                }
                if (cand && idxs >= lowIdx && k <= lim) {
                    int const c = (int)idxs - 2;
                    K2W const f = k2_fix16(X1, c, n);
                    if ((u32)f.lo == (u32)w) {
                        shortHit = true;
                        u32 e = k2_eq16(ww, f);
                        if ((int)e >= avail) e = (u32)avail; else if (e == 16) e |= 0x100u;
                        lenS = e;
                        int const mb = mbs < c ? mbs : c;
                        if (mb > 0) {
                            int const cmpn = mb < sb_n ? mb : sb_n;
                            u32 b = cmpn > 0 ? k2_eqback8(sb, k2_fixb8(Y1, c)) : 0u;
                            if ((int)b >= cmpn) { b = (u32)cmpn; if (cmpn < mb) b |= 0x100u; }
                            bkS = b;
                        }
                    }
                }
                do_resolve = true;
            }
        }

        // ---- the step is decided: inserts, then the winner's match ----
        if (kx_any(do_resolve)) {
            if (lane == 0) KX_STAT(3, 1);
            bool const rs = do_resolve;
            bool const cnd = rs && cand;
            bool const hit = cnd && (repHit | longHit | shortHit);
            u64 const th = (kx_ballot(hit) >> tbase) & tmask;
            int const K = (int)kx_popc64((kx_ballot(cnd) >> tbase) & tmask);
            int const wl = th ? (int)kx_ctz64(th) : -1;
            int const wmax = th ? wl : K - 1;
            // commit inserts of lanes <= wmax; a lane is superseded when a later committing lane of the team hits the same bucket
            bool const ins = cnd && k <= wmax;
            u32 const m = kx_team_or<G>((ins && predL >= 0 ? (1u << predL) : 0u) | (ins && predS >= 0 ? (1u << (16 + predS)) : 0u), lane);
            bool const supL = (m >> k) & 1u, supS = (m >> (16 + k)) & 1u;
            if (ins) {
                u32 const v = tag | (u32)(pos + 2);
                u32 const c1 = kx_chk_long(w, hbL) << KX_CHK_SHIFT, c2 = kx_chk_short(w) << KX_CHK_SHIFT;
                if (nt_st) { if (!supL) kx_st_nt(&L[hl], v | c1); if (!supS) kx_st_nt(&S[hs], v | c2); }
                else { if (!supL) L[hl] = v | c1; if (!supS) S[hs] = v | c2; }
            }
            // the winner's data, broadcast inside the team
            int const wsrc = tbase + (wl < 0 ? 0 : wl);
            int const mtLane = repHit ? KMT_REP : (longHit ? KMT_LONG : KMT_SHORT);
            u32 const myLen = repHit ? rlen : (longHit ? lenL : lenS);
            u32 const myMpos = repHit ? (u32)(pos + 1 - (int)off1) : (longHit ? idxl - 2u : idxs - 2u);
            u32 const myBack = longHit ? bkL : bkS;
            u32 const b_info = kx_shfl((u32)mtLane | myLen << 8 | myBack << 20, wsrc);        // type | len (9 bits) | back (9 bits)
            u32 const b_mpos = kx_shfl(myMpos, wsrc);
            u64 const mywa = (w >> 16) | (w_hi << 48);
            u32 const b_walo = kx_shfl((u32)mywa, wsrc), b_wahi = kx_shfl((u32)(mywa >> 32), wsrc);
            // the lane after the winner: its long-table entry is libzstd's "long match at ip + 1"
            bool const myL1 = longRaw && idxl > lowIdx;
            u32 const n_info = kx_shfl((u32)myL1 | lenL << 8 | bkL << 20, wsrc + 1);
            u32 const n_m1 = kx_shfl(idxl - 2u, wsrc + 1);
            u32 const n_hl = kx_shfl(hl | (kx_chk_long(w, hbL) << 16), wsrc + 1);
            if (rs) {
                have_w = false;
                if (!th) {
                    ip += K * step;
                    if (ip >= nextStep) { step++; nextStep += 256; }
                    state = (ip + step > ilimit) ? K2_CLEANUP : K2_SRCH;
                } else {
                    int const b_type = (int)(b_info & 0xFFu);
                    m_type = b_type; m_pos = ip + wl * step; c_pos = m_pos;
                    wa = (u64)b_walo | (u64)b_wahi << 32;
                    lenA = (b_info >> 8) & 0xFFu; bool const openA = (b_info >> 16) & 1u;
                    m_mpos = (int)b_mpos; l1ok = false; backA = 0; backB = 0;
                    if (b_type == KMT_REP) m_start = m_pos + 1;                     // (rlen was measured from pos + 1 on)
                    else {
                        m_start = m_pos; m_off = (u32)(m_start - m_mpos);
                        backA = (b_info >> 20) & 0x1FFu;
                        if (b_type == KMT_SHORT && (n_info & 1u)) {
                            l1ok = true; s1 = m_pos + step; m1 = (int)n_m1;
                            lenB = (n_info >> 8) & 0xFFu; openB = (n_info >> 16) & 1u; backB = (n_info >> 20) & 0x1FFu;
                        }
                        if (step < 4 && k == 0) L[n_hl & 0xFFFFu] = tag | (u32)(m_pos + step + 2) | (n_hl >> 16) << KX_CHK_SHIFT;
                    }
                    if (openA) { state = K2_EXT; ext_b = 0; }
                    else if (l1ok && openB) { state = K2_EXT; ext_b = 1; }
                    else do_choose = true;
                }
            }
        }

        // ---- EXT: sixteen more bytes per lane ----
        if (kx_any(st0 == K2_EXT)) {
            if (lane == 0) KX_STAT(4, 1);
            bool const ex = st0 == K2_EXT;
            int const cs = (ext_b ? s1 + (int)lenB : m_start + (int)lenA) + 16 * k;
            int const cm = (ext_b ? m1 + (int)lenB : m_mpos + (int)lenA) + 16 * k;
            u32 eq = 16;
            if (ex) {
                int const av = n - cs;
                if (av <= 0) eq = 0;
                else {
                    u32 e = k2_eq16(k2_fix16(X0, cs, n), k2_fix16(X1, cm, n));
                    if ((int)e > av) e = (u32)av;
                    eq = e;
                }
            }
            u64 const tb = (kx_ballot(ex && eq < 16) >> tbase) & tmask;
            int const f = tb ? (int)kx_ctz64(tb) : 0;
            u32 const eqf = kx_shfl(eq, tbase + f);
            if (ex) {
                u32 const add = tb ? 16u * (u32)f + eqf : 16u * G;
                if (ext_b) lenB += add; else lenA += add;
                if (tb) {
                    if (!ext_b && l1ok && openB) ext_b = 1;
                    else do_choose = true;
                }
            }
        }

        // ---- both candidates measured: take the longer, then its backward growth ----
        if (do_choose) {
            if (l1ok && lenB > lenA) { m_start = s1; m_mpos = m1; lenA = lenB; m_off = (u32)(s1 - m1); backA = backB; }
            back = backA & 0xFFu;
            if ((m_type == KMT_LONG || m_type == KMT_SHORT) && (backA & 0x100u)) state = K2_BACK;
            else do_final = true;
        }

        // ---- BACK: eight more bytes per lane, backwards ----
        if (kx_any(st0 == K2_BACK)) {
            if (lane == 0) KX_STAT(5, 1);
            bool const bk = st0 == K2_BACK;
            int const mb0 = (m_start - anchor < m_mpos) ? m_start - anchor : m_mpos;
            int const limk = mb0 - (int)back - 8 * k;               // bytes this lane may still add
            u32 c8 = 0;
            if (bk && limk > 0) {
                int const bs_ = m_start - (int)back - 8 * k, bm_ = m_mpos - (int)back - 8 * k;
                u32 b = k2_eqback8(k2_fixb8(Y0, bs_), k2_fixb8(Y1, bm_));
                c8 = (int)b > limk ? (u32)limk : b;
            }
            u64 const tb = (kx_ballot(bk && c8 < 8) >> tbase) & tmask;
            int const f = tb ? (int)kx_ctz64(tb) : 0;
            u32 const cf = kx_shfl(c8, tbase + f);
            if (bk) {
                if (tb) { back += 8u * (u32)f + cf; do_final = true; }
                else back += 8u * G;
            }
        }

        // ---- the sequence ----
        if (kx_any(do_final)) {
            if (lane == 0) KX_STAT(6, 1);
            if (do_final) {
                bool const bw = m_type == KMT_LONG || m_type == KMT_SHORT;
                u32 offBase = 1;
                if (bw) { m_start -= (int)back; m_mpos -= (int)back; lenA += back; off2 = off1; off1 = m_off; offBase = m_off + 3; }
                else if (m_type == KMT_REP0) { u32 const t = off2; off2 = off1; off1 = t; }
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
                compl_due = m_type != KMT_REP0 && ip <= ilimit;
                chk0 = true; step = 1; nextStep = ip + 256; have_w = false;
                state = ip <= ilimit ? K2_SRCH : K2_CLEANUP;
            }
        }

        // ================= finish the slice ===========================
        if (kx_any(state == K2_CLEANUP)) {
            if (state == K2_CLEANUP) {
                {
                    u32 const cnt = nseq & (2u * G - 1u);       // sequences still in registers
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
                state = K2_IDLE;
            }
        }
    }
}
