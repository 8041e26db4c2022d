// zstd_match.h -- the LZ stage of zstd level 3 ("double-fast": two hash
// tables, greedy parse, 3 repcodes) for many independent slices.
//
// Replaces, for the reference's one-shot ZstdCompressor(level=3).transform()
// (kompressor-zstd--nativelib/src/jvmCommonMain/kotlin/com/ensody/kompressor/zstd/ZstdCompressor.jvm.kt:27-42
//  -> jni/Wrapper.cpp:112 ZSTD_compressStream2), the match-finding half of the
// third-party libzstd 1.5.7 block compressor.  Output must equal that
// library's sequence list exactly, so the parse below is the same greedy
// decision chain -- only its execution is re-shaped for a 64-lane wave:
//
//  * a wave is split into 64/G "teams" of G lanes; each team owns one slice
//    at a time and pulls the next slice index from a global counter;
//  * one search step examines up to G-1 consecutive search positions at once
//    (lane k speculates that lanes < k found nothing); the first lane with a
//    hit wins, lanes before it commit their table inserts, the rest discard;
//  * match extension, backward catch-up and literal copy are team-cooperative;
//    sequences leave in whole 16-byte pieces;
//  * hash tables live in global memory (one pair per team, tagged with an
//    epoch so they are never cleared between slices).
//
// All cross-lane primitives are called from wave-uniform control flow.
#pragma once
#include "zstd_common.h"

struct KMatchArgs {
    const u8* src; const u64* in_off; const u32* in_len; u32 n_slices;
    KSeq* seqs; u32 seq_cap;         // per slice
    u8* lits; u32 lit_cap;           // per slice: the literals, in order
    KSliceMeta* meta;                // per slice
    u32* tables;                     // per team: KX_TBL_ENTRIES
    u32* team_epoch;                 // per team
    u32* counter;                    // work queue head (zeroed by the host)
    u32 flags;                       // 1 = non-temporal table loads, 2 = non-temporal table stores (the default),
                                     // 4 = copy no literals (the entropy kernel gathers them; A/B switch),
                                     // 8 = block mode with the parameters of a stream of unknown size,
                                     // 16 = block mode, slices of 4 MiB and more: table entries are plain 32-bit indices
    // block mode (frames of several blocks): one block of every unfinished slice per launch
    const KFrameState* fstate;       // per slice
    u32* big_tables;                 // per slice: KX_BIG_TBL_ENTRIES
    // the team tables in four pieces far apart in the HBM (team t: tseg[t & 3] + (t >> 2) tables), or tseg_n == 1: `tables` alone.
    // Random accesses confined to a few dozen GiB of this device's HBM reach 27.5 G/s, spread over most of it 37.
    u32* tseg[4] = { nullptr, nullptr, nullptr, nullptr }; u32 tseg_n = 1;
    // geometry of a team's tables (entries): the level-3 one unless the batch runs level 4's double-fast row (level = 4)
    u32 tbl_stride = KX_TBL_ENTRIES, tbl_long = KX_TBL_LONG, level = 3;
    u32 big_stride = KX_BIG_TBL_ENTRIES, big_long = KX_BIG_TBL_LONG;      // block mode: a slice's tables (level 4: KX_BIG4_*)
    // a piece of a batch on a stream of its own (kmp_zstd_compress_batch_pieces): this launch's workgroups own the teams from
    // block_base * (64 / G) on, so that the pieces of one batch run side by side over disjoint team tables
    u32 block_base = 0;
};

KX_DEV u32* kx_team_tables(const KMatchArgs& a, u32 team)
{
    if (a.tseg_n == 4) return a.tseg[team & 3u] + (size_t)(team >> 2) * a.tbl_stride;
    return a.tables + (size_t)team * a.tbl_stride;
}

enum { KST_IDLE = 0, KST_SEARCH = 1, KST_REPCHECK = 2, KST_MATCH = 3, KST_CLEANUP = 4, KST_DONE = 5 };
enum { KMT_REP = 0, KMT_LONG = 1, KMT_SHORT = 2, KMT_REP0 = 3 };

template <int G>
KX_DEV u32 kx_team_or(u32 v, int lane)
{
#pragma unroll
    for (int o = 1; o < G; o <<= 1) v |= kx_shfl(v, lane ^ o);
    return v;
}

template <int G>
KX_DEV u64 kx_team_or64(u64 v, int lane)
{
    u32 lo = kx_team_or<G>((u32)v, lane);
    u32 hi = (G > 32) ? kx_team_or<G>((u32)(v >> 32), lane) : 0u;
    return (u64)lo | ((u64)hi << 32);
}

// Length of the common prefix of src[s+len..) and src[m+len..), added to len.
// m < s. Each lane compares 8 bytes per round.
template <int G>
KX_DEV u32 kx_team_extend(bool act, const u8* src, int n, int s, int m, u32 len, int k, int tbase, u64 tmask)
{
    bool running = act;
    while (kx_any(running)) {
        if (kx_lane() == 0) KX_STAT(4, 1);                      // extension rounds of the wave
        u32 eq = 8;
        if (running) {
            int const p = s + (int)len + 8 * k;
            int const q = m + (int)len + 8 * k;
            int const avail = n - p;
            if (avail <= 0) eq = 0;
            else {
                u64 const d = kx_ld64_clamped(src, p, n) ^ kx_ld64_clamped(src, q, n);
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

// Number of equal bytes walking backwards from src[s-1] / src[m-1], at most maxback.
// have0: the bytes of the first round (offset k) were requested earlier (s0 / m0), together with other loads of the match
template <int G>
KX_DEV u32 kx_team_backward(bool act, const u8* src, int s, int m, int maxback, int k, int tbase, u64 tmask, bool have0 = false, u32 s0 = 0, u32 m0 = 0)
{
    u32 back = 0;
    bool running = act && maxback > 0;
    bool first = true;
    while (kx_any(running)) {
        if (kx_lane() == 0) KX_STAT(7, 1);                      // backward rounds of the wave
        bool ne = true;
        if (running) {
            int const o = (int)back + k;
            if (o < maxback) ne = (first && have0) ? (s0 != m0) : (src[s - 1 - o] != src[m - 1 - o]);
        }
        first = false;
        u64 const b = kx_ballot(running && ne);
        u64 const tb = (b >> tbase) & tmask;
        if (running) {
            if (tb) { back += kx_ctz64(tb); running = false; }
            else back += G;
        }
    }
    return back;
}

// BLK = false: every slice is one block (<= 128 KiB); tables per team, epoch-tagged.
// BLK = true:  the block [ipos, ipos + blockSize) of every slice whose frame is unfinished; tables per slice
//              (they persist from block to block), repcodes come from and go back to the frame state.
// DONE: what the wave does with a slice the moment one of its teams has finished it (k_zstd_l3_fused: entropy-code it, all 64
// lanes, while the other waves of the SIMD keep waiting for their table loads); called from wave-uniform control flow, after
// the team's stores have been waited for.
struct KNoDone { static constexpr bool on = false; KX_MEMBER void operator()(u32) const {} };

template <int G, bool BLK = false, class DONE = KNoDone>
KX_DEV void zstd_match_body(const KMatchArgs& a, DONE const& done = DONE())
{
    constexpr int NT = 64 / G;
    bool const wide = BLK && (a.flags & 16u);               // entries without check bits (indices need all 32 bits)
    u32 const IDXM = BLK ? (wide ? 0xFFFFFFFFu : KX_BLK_IDX_MASK) : KX_IDX_MASK;
    constexpr u32 TAGM = BLK ? 0u : KX_TAG_MASK;            // block mode: no epoch (tag stays 0)
    constexpr u32 CHKS = BLK ? KX_BLK_IDX_BITS : KX_CHK_SHIFT;
    u32 const CHKM = BLK ? (wide ? 0u : KX_BLK_CHK_MASK) : KX_CHK_MASK;
    // lowest valid index (position + 2) of the running block: 2 unless the window has slid (block mode, long slices)
    u32 lowIdx = 2u;
    int const lane = kx_lane();
    int const k = lane & (G - 1);
    int const tbase = lane - k;
    u32 const team = (kx_block() + a.block_base) * NT + (u32)(lane / G);
    u32* L = BLK ? a.big_tables : kx_team_tables(a, team);
    u32* S = L + (BLK ? a.big_long : a.tbl_long);
    int bstart = 0; u32 saved1 = 0, saved2 = 0;          // block mode: block start, repcodes set aside at block start
    u64 const tmask = (G == 64) ? ~0ull : ((1ull << G) - 1ull);

    // ---- team state (uniform across the team's lanes) -------------------
    int state = KST_IDLE;
    const u8* src = a.src; int n = 0; int ilimit = 0; u32 slice = 0;
    int ip = 0, anchor = 0; u32 off1 = 0, off2 = 0; int step = 1; int nextStep = 0;
    u32 nseq = 0, nlit = 0; u32 tag = 0; u32 hbL = 16, hbS = 15, mls = 5;
    u32 longType = 0, longPos = 0; u32 guard = 0; u32 status = 0;
    KSeq* seqs = a.seqs; u8* lits = a.lits;
    // pending match
    int m_type = 0, m_pos = 0, m_start = 0, m_mpos = 0; u32 m_len0 = 0, m_off = 0, m_idxl1 = 0; u64 m_w1 = 0;
    // sequences wait in registers (two per lane) until the team can store whole 16-byte pieces of a line
    u64 sq0 = 0, sq1 = 0;
    // long-table lookup of the first position of the next step, when the last step's extra lane already made it
    bool carry = false; u32 carry_idxl = 0;
    // flags bit 7 (experiment): the speculation width follows the last hit -- after a hit at lane w the next step looks at
    // w + 1 positions, after a step without a hit at all G - 1 again: the probes of lanes behind a winner are reads that buy nothing
    int kmax = G - 1; bool const adaptive = !BLK && (a.flags & 128u) != 0;
    // Loads that depend on the new position only are requested together, in the repcode-check block: the bytes of the immediate
    // repcode test, the bytes of the complementary inserts of the match that just ended (compl_due; wa = the bytes at its
    // search position + 2, requested when the match was taken) and the words of the first search step (pw: lane k's
    // position ip + k) -- one round trip instead of three in the chain of every sequence.
    bool compl_due = false; u64 wa = 0; int c_pos = 0;
    u64 pw = 0; bool have_pw = false;

    for (;;) {
        // ================= fetch the next slice =======================
        if (kx_any(state == KST_IDLE)) {
            u32 s = 0, ep = 0;
            if (state == KST_IDLE && k == 0) {
                s = kx_atomic_add(a.counter, 1u);
                if (!BLK && s < a.n_slices) {
                    ep = a.team_epoch[team] + 1;
                    if (ep > KX_EPOCH_MAX) ep = 0;          // 0 = "clear the tables, restart at 1"
                    a.team_epoch[team] = ep ? ep : 1u;
                }
            }
            s = kx_shfl(s, tbase); ep = kx_shfl(ep, tbase);
            if (state == KST_IDLE) {
                if (s >= a.n_slices) state = KST_DONE;
                else if (BLK) {
                    KFrameState const fs = a.fstate[s];
                    bool ok4 = true;
                    KParams P0 = (a.flags & 32u) ? kx_params_l2_dfast() : (a.level == 4u) ? kx_params_l4(a.in_len[s], ok4) : kx_params_l3(a.in_len[s]);          // (flags bit 5: level 2's double-fast row)
                    if (a.flags & 8u) { P0.windowLog = 21; P0.chainLog = a.level == 4u ? 18 : 16; P0.hashLog = a.level == 4u ? 18 : 17; P0.minMatch = 5; }   // streaming frame: size unknown when it starts
                    KBlockWin const bw = kx_block_window(fs.lowLimit, fs.dictLimit, fs.ipos, fs.blockSize, P0.windowLog);
                    // (a block that libzstd parses with the extDict variant is left to zstd_match_ext_body)
                    if (fs.blockSize != 0 && !bw.ext && kx_in_class((a.flags >> 6) & 3u, a.in_len[s])) {           // else: frame finished (or not this launch's), fetch the next slice
                        slice = s;
                        src = a.src + a.in_off[s];
                        seqs = a.seqs + (size_t)s * a.seq_cap; lits = a.lits + (size_t)s * a.lit_cap;
                        L = a.big_tables + (size_t)s * a.big_stride; S = L + a.big_long;
                        KParams const P = P0;
                        hbL = P.hashLog; hbS = P.chainLog; mls = P.minMatch;
                        nseq = 0; nlit = 0; longType = 0; longPos = 0; guard = 0; status = 0; tag = 0;
                        bstart = (int)fs.ipos; n = bstart + (int)fs.blockSize;       // n = end of the block
                        anchor = bstart; ilimit = n - 8;
                        // candidates: valid from ZSTD_getLowestPrefixIndex at the block's END on ...
                        lowIdx = kx_lowest_prefix((u32)n + 2u, bw.dictLimit, bw.maxDist);
                        ip = bstart + ((u32)bstart + 2u == lowIdx ? 1 : 0);
                        // ... repcodes longer than the history (at the first searched position) are set aside
                        // (ZSTD_compressBlock_doubleFast: offsetSaved)
                        off1 = fs.rep[0]; off2 = fs.rep[1]; saved1 = 0; saved2 = 0;
                        u32 const maxRep = ((u32)ip + 2u) - kx_lowest_prefix((u32)ip + 2u, bw.dictLimit, bw.maxDist);
                        if (off2 > maxRep) { saved2 = off2; off2 = 0; }
                        if (off1 > maxRep) { saved1 = off1; off1 = 0; }
                        step = 1; nextStep = ip + 256; carry = false; compl_due = false; have_pw = false;
                        state = (fs.blockSize < 8 || ip + 1 > ilimit) ? KST_CLEANUP : KST_SEARCH;
                    }
                } else {
                    slice = s;
                    src = a.src + a.in_off[s];
                    n = (int)a.in_len[s];
                    seqs = a.seqs + (size_t)s * a.seq_cap; lits = a.lits + (size_t)s * a.lit_cap;
                    bool lvl_ok = true;
                    KParams const P = (a.level == 4u) ? kx_params_l4((u32)n, lvl_ok) : kx_params_l3((u32)n);
                    hbL = P.hashLog; hbS = P.chainLog; mls = P.minMatch;
                    nseq = 0; nlit = 0; longType = 0; longPos = 0; guard = 0; status = lvl_ok ? 0u : 3u;      // 3: no double-fast row for this size at this level
                    if (ep == 0) {
                        for (u32 i = (u32)k; i < a.tbl_stride; i += G) L[i] = 0;
                        ep = 1;
                    }
                    tag = ep << KX_TAG_SHIFT;
                    anchor = 0; ilimit = n - 8;
                    ip = 1; off1 = 1; off2 = 0;     // rep {1,4,8}: 4 exceeds the 1 byte of history at ip=1
                    step = 1; nextStep = ip + 256; carry = false; compl_due = false; have_pw = false;
                    state = (n < 8 || ip + 1 > ilimit || !lvl_ok) ? KST_CLEANUP : KST_SEARCH;
                }
            }
        }
        if (kx_all(state == KST_DONE)) break;
        if (lane == 0) KX_STAT(0, 1);                           // outer iterations of the wave

        // ================= immediate repcode check ====================
        if (kx_any(state == KST_REPCHECK)) {
            if (lane == 0) KX_STAT(1, 1);
            bool const inrep = state == KST_REPCHECK;
            bool const chk = inrep && ip <= ilimit && off2 > 0;
            bool const docompl = inrep && compl_due && k == 0;            // (compl_due implies ip <= ilimit)
            u32 a4 = 0, b4 = 0; u64 wb = 0, wc = 0; bool pwv = false;
            if (chk) { a4 = kx_ld32(src + ip); b4 = kx_ld32(src + ip - (int)off2); }
            if (inrep && ip + k <= ilimit) { pw = kx_ld64(src + ip + k); pwv = true; }
            if (docompl) { wb = kx_ld64(src + ip - 2); wc = kx_ld64(src + ip - 1); }
            if (docompl) {
                // complementary insertion: curr+2 into both tables, then ip-2 (long) and ip-1 (short)
                u32 const va = tag | (u32)(c_pos + 2 + 2);
                L[kx_hash_long(wa, hbL)] = va | (wide ? 0u : kx_chk_long(wa, hbL) << CHKS);
                L[kx_hash_long(wb, hbL)] = (tag | (u32)(ip - 2 + 2)) | (wide ? 0u : kx_chk_long(wb, hbL) << CHKS);
                S[kx_hash_short(wa, hbS, mls)] = va | (wide ? 0u : kx_chk_short(wa) << CHKS);
                S[kx_hash_short(wc, hbS, mls)] = (tag | (u32)(ip - 1 + 2)) | (wide ? 0u : kx_chk_short(wc) << CHKS);
            }
            bool const hit = chk && a4 == b4;
            if (inrep) {
                compl_due = false;
                if (hit) {
                    if (k == 0) {
                        u64 const w = pw;                                  // (lane 0's word is the one at ip, and ip <= ilimit)
                        u32 const v = tag | (u32)(ip + 2);
                        S[kx_hash_short(w, hbS, mls)] = v | (wide ? 0u : kx_chk_short(w) << CHKS);
                        L[kx_hash_long(w, hbL)] = v | (wide ? 0u : kx_chk_long(w, hbL) << CHKS);
                    }
                    m_type = KMT_REP0; m_pos = ip; m_start = ip; m_mpos = ip - (int)off2; m_len0 = 4;
                    have_pw = false;
                    state = KST_MATCH;
                } else {
                    step = 1; nextStep = ip + 256; carry = false;
                    have_pw = true;                                        // (every lane the first step asks has its word: its position is <= ilimit)
                    state = (ip + 1 > ilimit) ? KST_CLEANUP : KST_SEARCH;
                }
            }
            (void)pwv;
        }

        // ================= speculative search step ====================
        if (kx_any(state == KST_SEARCH)) {
            if (lane == 0) KX_STAT(2, 1);
            if (k == 0 && state == KST_SEARCH) KX_STAT(5, 1);   // team search steps
            bool const srch = state == KST_SEARCH;
            int const pos = ip + k * step;
            bool const cand = srch && (k < kmax) && (k == 0 || pos < nextStep) && (pos + step <= ilimit);
            bool const prov = srch && (k == 0 || ((k == 1 || pos - step < nextStep) && pos <= ilimit && k <= kmax));
            u64 w = 0; u32 hl = 0, hs = 0, el = 0, es = 0;
            if (prov) {
                w = have_pw ? pw : kx_ld64(src + pos);
                hl = kx_hash_long(w, hbL); hs = kx_hash_short(w, hbS, mls);
                // the lane after the last candidate only provides the long-table lookup of "ip1"
                bool const haveL = carry && k == 0;
                if (a.flags & 1u) { if (!haveL) el = kx_ld_nt(&L[hl]); if (cand) es = kx_ld_nt(&S[hs]); }
                else { if (!haveL) el = L[hl]; if (cand) es = S[hs]; }
                KX_STAT(8, haveL ? 0 : 1); KX_STAT(9, cand ? 1 : 0);      // table probes issued (long, short)
            }
            u32 idxl = ((el & TAGM) == tag) ? (el & IDXM) : 0u;
            u32 idxs = ((es & TAGM) == tag) ? (es & IDXM) : 0u;
            u32 ckl = 0, cks = 0;                    // this position's check bits (also stored with its inserts)
            {
                ckl = wide ? 0u : kx_chk_long(w, hbL) << CHKS; cks = wide ? 0u : kx_chk_short(w) << CHKS;
                // an entry with other check bits cannot pass the 8- / 4-byte compare: no candidate, no source line fetched
                if ((el & CHKM) != ckl) idxl = 0;
                if ((es & CHKM) != cks) idxs = 0;
            }
            if (srch && carry && k == 0) idxl = carry_idxl;
            // what lanes < k of this team would have inserted before lane k looks up
            int predL = -1, predS = -1;
            if (BLK || a.level == 4u) {              // hashLog 17 / chainLog 16 or 17: two shuffles
#pragma unroll
                for (int d = 1; d < G; d++) {
                    bool const ok = prov && k >= d;
                    u32 const pl = kx_shfl(hl, lane - d), ps = kx_shfl(hs, lane - d);
                    if (ok && predL < 0 && pl == hl) predL = k - d;
                    if (ok && predS < 0 && ps == hs) predS = k - d;
                }
            } else {
                u32 const hpack = hl | (hs << 16);   // level 3, one block: hashLog <= 16 and chainLog <= 15
#pragma unroll
                for (int d = 1; d < G; d++) {
                    bool const ok = prov && k >= d;
                    u32 const hp = kx_shfl(hpack, lane - d);
                    if (ok && predL < 0 && (hp & 0xFFFFu) == hl) predL = k - d;
                    if (ok && predS < 0 && (hp >> 16) == hs) predS = k - d;
                }
            }
            if (predL >= 0) idxl = (u32)(pos - (k - predL) * step) + 2u;
            if (predS >= 0) idxs = (u32)(pos - (k - predS) * step) + 2u;

            bool repHit = false, longHit = false, shortHit = false;
            if (cand) {
                int const pr = (off1 > 0) ? pos + 1 - (int)off1 : pos;
                int const pl = (idxl >= lowIdx) ? (int)idxl - 2 : pos;
                int const ps = (idxs >= lowIdx) ? (int)idxs - 2 : pos;
                u32 const vr = kx_ld32(src + pr);
                u64 const vl = kx_ld64(src + pl);
                u32 const vs = kx_ld32(src + ps);
                repHit = (off1 > 0) && vr == (u32)(w >> 8);
                longHit = (idxl >= lowIdx) && vl == w;
                shortHit = (idxs >= lowIdx) && vs == (u32)w;
            }
            bool const hit = repHit | longHit | shortHit;
            u64 const th = (kx_ballot(hit) >> tbase) & tmask;
            int const K = (int)kx_popc64((kx_ballot(cand) >> tbase) & tmask);
            int const wl = th ? (int)kx_ctz64(th) : -1;
            int const wmax = th ? wl : K - 1;

            // commit inserts of lanes <= wmax; a lane is superseded when a later
            // committing lane of the team hits the same bucket
            bool const ins = cand && k <= wmax;
            bool supL, supS;
            if (G <= 16) {
                u32 const m = kx_team_or<G>((ins && predL >= 0 ? (1u << predL) : 0u) | (ins && predS >= 0 ? (1u << (16 + predS)) : 0u), lane);
                supL = (m >> k) & 1u; supS = (m >> (16 + k)) & 1u;
            } else {
                u64 const rl = kx_team_or64<G>((ins && predL >= 0) ? (1ull << predL) : 0ull, lane);
                u64 const rs = kx_team_or64<G>((ins && predS >= 0) ? (1ull << predS) : 0ull, lane);
                supL = (rl >> k) & 1ull; supS = (rs >> k) & 1ull;
            }
            if (ins) {
                KX_STAT(10, (supL ? 0 : 1) + (supS ? 0 : 1));             // inserts of searched positions
                u32 const v = tag | (u32)(pos + 2);
                if (a.flags & 2u) { if (!supL) kx_st_nt(&L[hl], v | ckl); if (!supS) kx_st_nt(&S[hs], v | cks); }
                else { if (!supL) L[hl] = v | ckl; if (!supS) S[hs] = v | cks; }
            }

            // winner data, broadcast inside the team
            int const wsrc = tbase + (wl < 0 ? 0 : wl);
            int const mtLane = repHit ? KMT_REP : (longHit ? KMT_LONG : KMT_SHORT);
            int const b_type = (int)kx_shfl((u32)mtLane, wsrc);
            u32 const b_idxl = kx_shfl(idxl, wsrc);
            u32 const b_idxs = kx_shfl(idxs, wsrc);
            u32 const n_wlo = kx_shfl((u32)w, wsrc + 1);
            u32 const n_whi = kx_shfl((u32)(w >> 32), wsrc + 1);
            u32 const n_idxl = kx_shfl(idxl, wsrc + 1);
            u32 const n_hl = kx_shfl(hl, wsrc + 1);
            // after a step without a hit the next step starts where the extra lane (lane K) looked
            u32 const p_idxl = kx_shfl(idxl, tbase + K);
            bool const p_prov = kx_shfl((u32)prov, tbase + K) != 0u;

            if (srch) {
                guard++;
                have_pw = false;
                if (adaptive) { kmax = th ? wl + 1 : G - 1; if (kmax > G - 1) kmax = G - 1; }
                if (!th) {
                    ip += K * step;
                    carry = p_prov && K > 0; carry_idxl = p_idxl;
                    if (ip >= nextStep) { step++; nextStep += 256; }
                    if (ip + step > ilimit) state = KST_CLEANUP;
                    if (K == 0 || guard > 2u * (u32)n + 64u) { status = 1; state = KST_CLEANUP; }
                } else {
                    m_type = b_type; m_pos = ip + wl * step;
                    if (b_type == KMT_REP) { m_start = m_pos + 1; m_mpos = m_start - (int)off1; m_len0 = 4; }
                    else {
                        if (b_type == KMT_LONG) { m_start = m_pos; m_mpos = (int)b_idxl - 2; m_len0 = 8; }
                        else { m_start = m_pos; m_mpos = (int)b_idxs - 2; m_len0 = 4; }
                        m_off = (u32)(m_start - m_mpos);
                        m_idxl1 = n_idxl; m_w1 = (u64)n_wlo | ((u64)n_whi << 32);
                        if (step < 4 && k == 0) L[n_hl] = tag | (u32)(m_pos + step + 2) | (wide ? 0u : kx_chk_long(m_w1, hbL) << CHKS);
                    }
                    carry = false;
                    state = KST_MATCH;
                }
            }
        }

        // ================= take the match =============================
        if (kx_any(state == KST_MATCH)) {
            if (lane == 0) KX_STAT(3, 1);
            if (k == 0 && state == KST_MATCH) KX_STAT(6, 1);    // team sequences
            bool const mt = state == KST_MATCH;
            // requested up front, beside the first round of the extension: the long candidate at ip + 1, the bytes of the
            // complementary inserts at the search position + 2, and the first round of the backward growth of this candidate
            bool l1cand = false; int s1 = 0, m1 = 0; u64 l1v = 0;
            if (mt && m_type == KMT_SHORT && m_idxl1 > lowIdx) {
                m1 = (int)m_idxl1 - 2; s1 = m_pos + step;
                l1v = kx_ld64(src + m1); l1cand = true;
            }
            u64 wa_new = 0;
            if (mt && m_type != KMT_REP0 && k == 0 && m_pos + 10 <= n) wa_new = kx_ld64(src + m_pos + 2);     // (only then can the new position lie at or below ilimit)
            bool const bw0 = mt && (m_type == KMT_LONG || m_type == KMT_SHORT);
            int const mlow0 = m_mpos - ((int)lowIdx - 2);
            int const mb0 = (m_start - anchor < mlow0) ? m_start - anchor : mlow0;
            u32 bs0 = 0, bm0 = 0;
            if (bw0 && k < mb0) { bs0 = src[m_start - 1 - k]; bm0 = src[m_mpos - 1 - k]; }
            u32 lenA = kx_team_extend<G>(mt, src, n, m_start, m_mpos, m_len0, k, tbase, tmask);
            bool const l1ok = l1cand && l1v == m_w1;
            bool tookB = false;
            if (kx_any(l1ok)) {
                u32 const lenB = kx_team_extend<G>(l1ok, src, n, s1, m1, 8u, k, tbase, tmask);
                if (l1ok && lenB > lenA) { m_start = s1; m_mpos = m1; lenA = lenB; m_off = (u32)(s1 - m1); tookB = true; }
            }
            bool const bw = mt && (m_type == KMT_LONG || m_type == KMT_SHORT);
            int const mlow = m_mpos - ((int)lowIdx - 2);               // the match may grow backwards down to the lowest valid position
            int const mb = (m_start - anchor < mlow) ? m_start - anchor : mlow;
            u32 const back = kx_team_backward<G>(bw, src, m_start, m_mpos, mb, k, tbase, tmask, !tookB, bs0, bm0);
            if (mt) {
                u32 offBase = 1;
                if (bw) { m_start -= (int)back; m_mpos -= (int)back; lenA += back; off2 = off1; off1 = m_off; offBase = m_off + 3; }
                else if (m_type == KMT_REP0) { u32 const t = off2; off2 = off1; off1 = t; }
                int const ll = m_start - anchor;
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
                // the complementary inserts of this match wait for the bytes at the new position: the repcode-check block asks for them
                compl_due = m_type != KMT_REP0 && ip <= ilimit;
                if (compl_due) { wa = wa_new; c_pos = m_pos; }
                if (++guard > 2u * (u32)n + 64u) { status = 2; state = KST_CLEANUP; }
                else state = KST_REPCHECK;
            }
        }

        // ================= finish the slice ===========================
        if (kx_any(state == KST_CLEANUP)) {
            bool const fin = state == KST_CLEANUP;
            if (fin) {
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
                    if (!BLK && (a.flags & 256u)) { u64 const t = kx_realtime(); mm.pad[0] = (u32)t; mm.pad[1] = (u32)(t >> 32); }   // diagnostics: when the slice was done (100 MHz ticks)
                    if (BLK) {
                        // repcodes this block leaves behind (taken over by the frame only if the block is emitted compressed)
                        u32 const s2 = (saved1 != 0 && off1 != 0) ? saved1 : saved2;
                        mm.pad[0] = off1 ? off1 : saved1; mm.pad[1] = off2 ? off2 : s2;
                    }
                    a.meta[slice] = mm;
                }
                state = KST_IDLE;
            }
            if (DONE::on) {
                u64 fm = kx_ballot(fin && k == 0);
                kx_sync();                                      // the team's sequences, literals and meta are in memory
                // `done` is a real call into code that wants every register: the parse state goes to private memory by hand and
                // comes back afterwards, so that nothing of it lives across the call (left to the register allocator, the spills
                // land inside the search loop).
                u32 sv[64];
#define KX_MS32(X) X(state) X(n) X(ilimit) X(slice) X(ip) X(anchor) X(off1) X(off2) X(step) X(nextStep) X(nseq) X(nlit) X(tag) X(hbL) X(hbS) X(mls) \
                   X(longType) X(longPos) X(guard) X(status) X(m_type) X(m_pos) X(m_start) X(m_mpos) X(m_len0) X(m_off) X(m_idxl1) X(carry_idxl) X(kmax) \
                   X(c_pos) X(lowIdx) X(bstart) X(saved1) X(saved2)
#define KX_MS64(X) X(m_w1) X(sq0) X(sq1) X(wa) X(pw)
#define KX_MSP(X) X(src) X(seqs) X(lits) X(L) X(S)
                {
                    int q = 0;
#define KX_SAVE32(x) sv[q++] = (u32)(x);
#define KX_SAVE64(x) sv[q++] = (u32)(u64)(x); sv[q++] = (u32)((u64)(x) >> 32);
#define KX_SAVEP(x) sv[q++] = (u32)(uintptr_t)(x); sv[q++] = (u32)((u64)(uintptr_t)(x) >> 32);
                    KX_MS32(KX_SAVE32) KX_MS64(KX_SAVE64) KX_MSP(KX_SAVEP)
                    sv[q++] = (carry ? 1u : 0u) | (compl_due ? 2u : 0u) | (have_pw ? 4u : 0u);
                }
                KX_ESCAPE(sv);
                while (fm) { int const j = (int)kx_ctz64(fm); fm &= fm - 1ull; done(kx_bcast(sv[3], j)); KX_ESCAPE(sv); }
                {
                    int q = 0;
#define KX_LOAD32(x) x = (decltype(x))sv[q++];
#define KX_LOAD64(x) { u32 const lo_ = sv[q++]; u32 const hi_ = sv[q++]; x = (u64)lo_ | ((u64)hi_ << 32); }
#define KX_LOADP(x) { u32 const lo_ = sv[q++]; u32 const hi_ = sv[q++]; x = (decltype(x))(uintptr_t)((u64)lo_ | ((u64)hi_ << 32)); }
                    KX_MS32(KX_LOAD32) KX_MS64(KX_LOAD64) KX_MSP(KX_LOADP)
                    u32 const fl = sv[q++]; carry = fl & 1u; compl_due = (fl & 2u) != 0; have_pw = (fl & 4u) != 0;
                }
#undef KX_MS32
#undef KX_MS64
#undef KX_MSP
#undef KX_SAVE32
#undef KX_SAVE64
#undef KX_SAVEP
#undef KX_LOAD32
#undef KX_LOAD64
#undef KX_LOADP
            }
        }
    }
}
