// deflate_predecode.h -- the serial half of inflate, a LANE per stream.
//
// k_inflate (deflate_decode.h) decodes the symbols of a stream on lane 0 of a wave: 63 lanes wait while one walks
// the Huffman codes.  Here KIP_STREAMS streams share a wave, one lane each, with everything a lane touches per symbol in
// LDS -- a ring of input words, an 8-bit first-level table for literal/length codes and a 6-bit one for distances,
// the canonical lists for longer codes -- and what a stream decodes to is left in HBM in the form the zstd decoder's
// window executor takes (zstd_decode.h kxd_exec_window): the literal bytes, and per match one 8-byte record
// literals-before | (length - 3) << 16 | distance << 32.  k_inflate_exec then runs a wave per stream over that.
//
// Replaces zlib's inflate() behind the reference's ZlibDecompressor (kompressor-zlib--nativelib/src/jvmCommonMain/jni/
// Wrapper.cpp:84-153, inflate at :144) for the streams it covers; a stream it does not cover -- stored blocks, incomplete
// or over-subscribed codes, any error, more output than the staging holds -- is marked and decoded by inflate_stream
// as before, which also decides its status.  So this file only has to be right about well-formed Huffman blocks.
//
// The loop is made of phases so that the lanes of a wave touch HBM together: every lane tops its ring up (128-bit loads,
// one wait for the wave), the lanes that stand at a block header read it and build their tables, then every lane
// decodes up to KIP_SYMS symbols out of LDS (stores only: nothing waits for them until the next phase).  Streams
// written by zlib change blocks every 16 383 symbols, so the lanes of a wave reach their headers in the same phase.
#pragma once
#include "deflate_decode.h"

#define KIP_STREAMS 16          /* streams (active lanes) per wave */
#define KIP_RING 32             /* input words per stream */
#define KIP_SYMS 48             /* symbols per phase */

struct KipArgs {
    const u8* src; const u64* in_off; const u32* in_len; u32 n_slices; const u32* out_cap; u32 format;
    u64* stage; u32 seq_cap;    // per stream: seq_cap match records
    u8* lits; u32 lit_cap;      // per stream: lit_cap literal bytes
    u32* nseq; u32* nlit;       // per stream: counts; bit 31 of nseq = covered (the executor may run)
    const u32* perm;            // slot -> stream (streams of similar size side by side), or null
};

struct KipStream {
    u32 ring[KIP_RING];
    union { u16 l1[256]; u8 lens[320]; } t;     // literal/length first level: symbol | length << 9, 0 = longer than 8 bits; the code lengths of a header first
    u16 d1[64];                                  // distance first level (6 bits), same packing
    u8 lsortLo[288]; u8 lsortHi[36];             // literal/length symbols in canonical order (9 bits each)
    u8 dsort[32];
    u16 lcount[16]; u16 dcount[16];
    u16 offs[16];                                // next free place per code length while a list is sorted
};

struct KipBits { u64 buf; int cnt; u32 rp, wp; };    // rp / wp: words of the stream consumed from / put into the ring

// the words [w, w + 4) of a stream of nbytes bytes (zero past its end)
KX_DEV KxQuad kip_load4(const u8* sp, u32 nbytes, u32 w)
{
    u32 const o = 4u * w;
    if (o + 16u <= nbytes) return kx_ld128u(sp + o);
    KxQuad q; q.x = 0; q.y = 0; q.z = 0; q.w = 0;
    u32 v[4] = { 0, 0, 0, 0 };
    for (u32 k = 0; k < 16u && o + k < nbytes; k++) v[k >> 2] |= (u32)sp[o + k] << (8u * (k & 3u));
    q.x = v[0]; q.y = v[1]; q.z = v[2]; q.w = v[3];
    return q;
}
KX_DEV void kip_put4(KipStream& S, KipBits& br, const KxQuad& q)
{
    u32 const i = br.wp & (KIP_RING - 1);        // wp is a multiple of 4: the four words do not wrap
    S.ring[i] = q.x; S.ring[i + 1] = q.y; S.ring[i + 2] = q.z; S.ring[i + 3] = q.w;
    br.wp += 4;
}
// at least 32 valid bits in the container afterwards (a dry ring is topped up on the spot: rare, long headers)
#define KIP_FILL() { if (br.cnt <= 32) { \
        if (br.rp == br.wp) { KxQuad const q_ = kip_load4(sp, nbytes, br.wp); kip_put4(S, br, q_); } \
        br.buf |= (u64)S.ring[br.rp & (KIP_RING - 1)] << br.cnt; br.cnt += 32; br.rp++; } }
#define KIP_TAKE(dst_, n_) { u32 const n__ = (n_); dst_ = (u32)(br.buf & ((1ull << n__) - 1ull)); br.buf >>= n__; br.cnt -= (int)n__; }

// canonical code from lens[0, n): counts, sorted symbols, first-level table of tbits bits.  false unless the code is
// complete (zlib also takes a few incomplete ones; those streams go to inflate_stream)
KX_DEV bool kip_counts(u16* count, const u8* lens, int n)
{
    for (int l = 0; l < 16; l++) count[l] = 0;
    for (int s = 0; s < n; s++) count[lens[s]]++;
    count[0] = 0;
    int left = 1;
    for (int l = 1; l < 16; l++) { left <<= 1; left -= count[l]; if (left < 0) return false; }
    return left == 0;
}

KX_DEV u32 kip_lsym(const KipStream& S, u32 i) { return (u32)S.lsortLo[i] | ((((u32)S.lsortHi[i >> 3] >> (i & 7u)) & 1u) << 8); }

// a code longer than the first level: canonical walk over the lengths (bits of the code arrive most significant first)
KX_DEV bool kip_walk(u64 buf, const u16* count, u32& index, u32& clen)
{
    u32 code = 0, first = 0, idx = 0;
    for (u32 l = 1; l <= 15; l++) {
        code |= (u32)((buf >> (l - 1)) & 1u);
        u32 const cnt = count[l];
        if (code < first + cnt) { index = idx + (code - first); clen = l; return true; }
        idx += cnt; first += cnt; first <<= 1; code <<= 1;
    }
    return false;
}

// The codes the first-level tables do not hold, without touching LDS.  A wave takes this path whenever one of its lanes
// meets a long code -- most rounds -- so it must be short.  In a canonical code the codes of one length are a contiguous
// range of the 15-bit left-aligned code space, and the ranges follow each other by length: with the range ends of the
// lengths above the table width in registers (16 bits each), the length of a code is one comparison per length, and its
// place in the canonical order is (code - start of its range) >> (15 - length) + the symbols of the shorter lengths.
// TB: index width of the first-level table; W: bits per "symbols before this length" field.
struct KipLong { u32 e[4]; u32 lo0; u64 ib; };
template <int TB, int W>
KX_DEV void kip_long_build(const u16* count, KipLong& K)
{
    u32 first = 0, idx = 0;
    for (int l = 1; l <= TB; l++) { u32 const cnt = count[l]; idx += cnt; first = (first + cnt) << 1; }
    K.lo0 = first << (15 - (TB + 1));
    K.ib = 0; K.e[0] = 0; K.e[1] = 0; K.e[2] = 0; K.e[3] = 0;
    for (int l = TB + 1; l <= 15; l++) {
        int const j = l - TB - 1; u32 const cnt = count[l];
        K.ib |= (u64)idx << (W * j);
        u32 const end = (first + cnt) << (15 - l);
        if (j < 8) K.e[j >> 1] |= end << (16 * (j & 1));
        idx += cnt; first = (first + cnt) << 1;
    }
}
template <int TB, int W>
KX_DEV void kip_long(u64 buf, const KipLong& K, u32& index, u32& clen)
{
    u32 const c15 = kx_brev32((u32)buf) >> 17;
    u32 k = 0, lo = K.lo0;
#pragma unroll
    for (int j = 0; j < 14 - TB; j++) {
        u32 const ej = (K.e[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
        bool const ge = c15 >= ej;
        k += ge ? 1u : 0u; lo = ge ? ej : lo;
    }
    u32 const l = (u32)TB + 1u + k;
    index = ((u32)(K.ib >> ((u32)W * k)) & ((1u << W) - 1u)) + ((c15 - lo) >> (15u - l));
    clen = l;
}

KX_DEV void inflate_predecode_body(const KipArgs& a)
{
    KX_SHARED KipStream lds[KIP_STREAMS];
    int const lane = kx_lane();
    bool const mine = lane < KIP_STREAMS;
    u32 const slot = kx_block() * (u32)KIP_STREAMS + (u32)(mine ? lane : 0);
    bool live = mine && slot < a.n_slices;
    u32 const f = (live && a.perm) ? a.perm[slot] : slot;
    KipStream& S = lds[mine ? lane : 0];
    const u8* sp = a.src; u32 nbytes = 0, cap = 0;
    bool ok = live;                                          // false: not covered (whatever the reason)
    if (live) {
        const u8* const src = a.src + a.in_off[f]; u32 const srcSize = a.in_len[f];
        cap = a.out_cap[f];
        // (a stream that decodes to more than the staging holds is not covered; the literals of one phase may run past cap
        //  before the check at its end, so cap keeps KIP_SYMS bytes of the staging free)
        if (cap > a.lit_cap - KIP_SYMS) cap = (a.lit_cap - KIP_SYMS) & ~7u;
        u32 spos = 0, send = srcSize, fmt = a.format & 0xFFu; u32 const wmax = (a.format >> 8) ? (a.format >> 8) : 15u;
        if (fmt == 3) fmt = (srcSize >= 2 && src[0] == 0x1F && src[1] == 0x8B) ? 2u : 1u;
        if (fmt == 2) {
            // only the plain ten-byte header zlib writes (no FEXTRA / FNAME / FCOMMENT / FHCRC): the others go the long way
            if (srcSize < 18 || src[0] != 0x1F || src[1] != 0x8B || src[2] != 8 || src[3] != 0) ok = false;
            spos = 10; send = srcSize - 8;
        } else if (fmt == 1) {
            if (srcSize < 6) ok = false;
            else {
                u32 const cmf = src[0], flg = src[1];
                if ((cmf & 0x0F) != 8 || (cmf >> 4) + 8u > wmax || ((cmf << 8) | flg) % 31 != 0 || (flg & 0x20)) ok = false;
            }
            spos = 2; send = srcSize - 4;
        }
        if (!ok || send <= spos) { ok = false; spos = 0; send = 0; }
        sp = src + spos; nbytes = send - spos;
    }
    u64* const stage = a.stage + (size_t)f * a.seq_cap;
    u8* const lits = a.lits + (size_t)f * a.lit_cap;
    KipBits br; br.buf = 0; br.cnt = 0; br.rp = 0; br.wp = 0;
    u32 const nwords = (nbytes + 3u) >> 2;
    bool fin = !ok;                                          // this lane has nothing more to do
    bool inBlock = false, last = false;
    u32 op = 0, nseq = 0, nlit = 0, ll = 0;                  // output position, records, literal bytes, literals since the last match
    u64 lq = 0;                                              // literal bytes not yet stored (nlit & 7 of them)
    KipLong LL, DL;                                          // the long codes of the block
    LL.e[0] = LL.e[1] = LL.e[2] = LL.e[3] = 0; LL.lo0 = 0; LL.ib = 0; DL = LL;
    while (kx_any(!fin)) {
        // ---- top the ring up: every free group of four words, up to eight groups ----------------------------------------
        {
            KxQuad q[8]; u32 const wp0 = br.wp; u32 const room = KIP_RING - (br.wp - br.rp);
#pragma unroll
            for (int u = 0; u < 8; u++) { q[u].x = 0; q[u].y = 0; q[u].z = 0; q[u].w = 0; if (!fin && room >= 4u * (u32)(u + 1) && wp0 + 4u * (u32)u < nwords) q[u] = kip_load4(sp, nbytes, wp0 + 4u * (u32)u); }
#pragma unroll
            for (int u = 0; u < 8; u++) if (!fin && room >= 4u * (u32)(u + 1) && wp0 + 4u * (u32)u < nwords) kip_put4(S, br, q[u]);
        }
        // ---- block headers -----------------------------------------------------------------------------------------------
        if (!fin && !inBlock) {
            if (last) {
                // the stream must end inside its last byte (inflate_stream's rule) and hold nothing after it
                u32 const bitsUsed = 32u * br.rp - (u32)br.cnt;
                if (((bitsUsed + 7u) >> 3) != nbytes) ok = false;
                fin = true;
            } else {
                u32 hdr; KIP_FILL() KIP_TAKE(hdr, 3)
                last = hdr & 1u; u32 const btype = hdr >> 1;
                if (btype == 0 || btype == 3) { ok = false; fin = true; }         // stored blocks (and the invalid type): the long way
                else {
                    u32 hlit = 288, hdist = 30;
                    if (btype == 1) {
                        for (int s = 0; s < 288; s++) S.t.lens[s] = (u8)kd_static_llen((u32)s);
                        for (int s = 0; s < 30; s++) S.t.lens[288 + s] = 5;
                        // (the fixed distance code has 30 symbols of 5 bits: incomplete by two codes, which inflate accepts)
                        S.t.lens[318] = 5; S.t.lens[319] = 5; hdist = 32;
                    } else {
                        u32 hclen; KIP_FILL() KIP_TAKE(hlit, 5) KIP_TAKE(hdist, 5) KIP_TAKE(hclen, 4)
                        hlit += 257; hdist += 1; hclen += 4;
                        if (hlit > 286 || hdist > 30) { ok = false; fin = true; }
                        else {
                            // the code-length code: at most 7 bits, read by the canonical walk
                            // (its lengths, counts and sorted list borrow the distance lists, which are built after it)
                            u8* const cl = S.dsort; u16* const ccount = S.dcount; u8* const csort = S.lsortHi;
                            for (int i = 0; i < 19; i++) cl[i] = 0;
                            for (u32 i = 0; i < hclen; i++) {
                                u32 v; KIP_FILL() KIP_TAKE(v, 3)
                                u32 const pos = i < 3 ? 16u + i : (i == 3 ? 0u : ((i & 1u) ? 7u - ((i - 5u) >> 1) : 8u + ((i - 4u) >> 1)));   // 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
                                cl[pos] = (u8)v;
                            }
                            if (!kip_counts(ccount, cl, 19)) { ok = false; fin = true; }
                            else {
                                S.offs[1] = 0;
                                for (int l = 1; l < 15; l++) S.offs[l + 1] = (u16)(S.offs[l] + ccount[l]);
                                for (u32 s = 0; s < 19; s++) if (cl[s]) csort[S.offs[cl[s]]++] = (u8)s;
                                u32 idx = 0, prevLen = 0;
                                while (idx < hlit + hdist && !fin) {
                                    KIP_FILL()
                                    u32 index = 0, clen = 0;
                                    if (!kip_walk(br.buf, ccount, index, clen) || clen > 7) { ok = false; fin = true; break; }
                                    br.buf >>= clen; br.cnt -= (int)clen;
                                    u32 const sym = csort[index];
                                    if (sym < 16) { S.t.lens[idx++] = (u8)sym; prevLen = sym; }
                                    else {
                                        u32 rep, val = 0, x;
                                        if (sym == 16) { if (idx == 0) { ok = false; fin = true; break; } KIP_TAKE(x, 2) rep = 3 + x; val = prevLen; }
                                        else if (sym == 17) { KIP_TAKE(x, 3) rep = 3 + x; }
                                        else { KIP_TAKE(x, 7) rep = 11 + x; }
                                        if (idx + rep > hlit + hdist) { ok = false; fin = true; break; }
                                        while (rep--) S.t.lens[idx++] = (u8)val;
                                        prevLen = val;
                                    }
                                }
                                if (!fin && S.t.lens[256] == 0) { ok = false; fin = true; }       // no end-of-block code
                            }
                        }
                    }
                    if (!fin) {
                        // distance side first (its lengths sit behind the literal/length ones, whose table then takes their place)
                        const u8* const dl = S.t.lens + (btype == 1 ? 288u : hlit);
                        bool good = kip_counts(S.dcount, dl, (int)hdist);
                        if (good) {
                            S.offs[1] = 0;
                            for (int l = 1; l < 15; l++) S.offs[l + 1] = (u16)(S.offs[l] + S.dcount[l]);
                            for (u32 s = 0; s < hdist; s++) if (dl[s]) S.dsort[S.offs[dl[s]]++] = (u8)s;
                            for (int i = 0; i < 64; i++) S.d1[i] = 0;
                            u32 code = 0, k = 0;
                            for (u32 l = 1; l <= 15; l++) {
                                for (u32 c = 0; c < S.dcount[l]; c++, k++, code++) {
                                    if (l <= 6) { u32 const rev = kd_bi_reverse(code, (int)l); for (u32 i = rev; i < 64u; i += 1u << l) S.d1[i] = (u16)((u32)S.dsort[k] | (l << 9)); }
                                }
                                code <<= 1;
                            }
                            kip_long_build<6, 5>(S.dcount, DL);
                            good = kip_counts(S.lcount, S.t.lens, (int)(btype == 1 ? 288u : hlit));
                        }
                        if (good) {
                            u32 const nl = btype == 1 ? 288u : hlit;
                            S.offs[1] = 0;
                            for (int l = 1; l < 15; l++) S.offs[l + 1] = (u16)(S.offs[l] + S.lcount[l]);
                            for (int i = 0; i < 36; i++) S.lsortHi[i] = 0;
                            for (u32 s = 0; s < nl; s++) {
                                u32 const l = S.t.lens[s];
                                if (l) { u32 const o = S.offs[l]++; S.lsortLo[o] = (u8)s; if (s & 256u) S.lsortHi[o >> 3] |= (u8)(1u << (o & 7u)); }
                            }
                            for (int i = 0; i < 256; i++) S.t.l1[i] = 0;                  // (the lengths are gone from here on)
                            u32 code = 0, k = 0;
                            for (u32 l = 1; l <= 15; l++) {
                                u32 const cnt = S.lcount[l];
                                if (l <= 8) for (u32 c = 0; c < cnt; c++, k++, code++) {
                                    u32 const rev = kd_bi_reverse(code, (int)l); u32 const e = kip_lsym(S, k) | (l << 9);
                                    for (u32 i = rev; i < 256u; i += 1u << l) S.t.l1[i] = (u16)e;
                                }
                                else { k += cnt; code += cnt; }
                                code <<= 1;
                            }
                            kip_long_build<8, 9>(S.lcount, LL);
                            inBlock = true;
                        } else { ok = false; fin = true; }
                    }
                }
            }
        }
        // ---- symbols -----------------------------------------------------------------------------------------------------
        for (int n = 0; n < KIP_SYMS; n++) {
            bool const go = !fin && inBlock && ((br.wp - br.rp) >= 2u || br.wp >= nwords);      // 64 bits at hand besides the container, or the stream's tail
            if (!kx_any(go)) break;
            if (go) {
                KIP_FILL()
                u32 e = S.t.l1[br.buf & 255u]; u32 sym, clen = e >> 9;
                if (clen) sym = e & 511u;
                else {
                    u32 index = 0;
                    kip_long<8, 9>(br.buf, LL, index, clen);
                    sym = index < 288u ? kip_lsym(S, index) : 999u;
                }
                br.buf >>= clen; br.cnt -= (int)clen;
                if (sym < 256) {
                    lq |= (u64)sym << (8u * (nlit & 7u)); nlit++; ll++; op++;
                    if ((nlit & 7u) == 0) { kx_st64(lits + (nlit - 8u), lq); lq = 0; }  // (nlit <= op <= cap + KIP_SYMS <= lit_cap: checked per phase)
                } else if (sym == 256) inBlock = false;
                else if (sym > 285) { ok = false; fin = true; }
                else if (op > cap) { ok = false; fin = true; }                           // (literals ran past the capacity earlier in this phase)
                else {
                    u32 const lc = sym - 257; u32 len, x;
                    if (lc < 8) len = 3 + lc; else if (lc == 28) len = 258;
                    else { u32 const eb = (lc - 4) >> 2; KIP_TAKE(x, eb) len = 3 + ((4 + (lc & 3u)) << eb) + x; }
                    KIP_FILL()
                    u32 de = S.d1[br.buf & 63u]; u32 dsym, dlen = de >> 9;
                    if (dlen) dsym = de & 511u;
                    else {
                        u32 index = 0;
                        kip_long<6, 5>(br.buf, DL, index, dlen);
                        dsym = index < 32u ? S.dsort[index] : 99u;
                    }
                    br.buf >>= dlen; br.cnt -= (int)dlen;
                    if (dsym > 29) { ok = false; fin = true; }
                    else {
                        u32 dist;
                        if (dsym < 4) dist = dsym + 1; else { u32 const eb = (dsym - 2) >> 1; KIP_TAKE(x, eb) dist = 1 + ((2 + (dsym & 1u)) << eb) + x; }
                        if (dist > op || (u64)op + len > cap || ll > 0xFFFFu || nseq >= a.seq_cap) { ok = false; fin = true; }
                        else { stage[nseq++] = (u64)ll | ((u64)(len - 3u) << 16) | ((u64)dist << 32); ll = 0; op += len; }
                    }
                }
            }
        }
        if (!fin && op > cap) { ok = false; fin = true; }                                 // more output than the caller's capacity (or the staging's)
    }
    if (live) {
        if (ok) { u32 const k = nlit & 7u; for (u32 i = 0; i < k; i++) lits[(nlit - k) + i] = (u8)(lq >> (8u * i)); }
        a.nseq[f] = ok ? (nseq | 0x80000000u) : 0u;
        a.nlit[f] = ok ? nlit : 0u;
    }
}
#undef KIP_FILL
#undef KIP_TAKE

// ---------------------------------------------------------------------------
// k_inflate_exec: one wave per stream.  A covered stream is put together by the zstd decoder's window executor from
// the staged literals and match records, its checksum verified as inflate_stream does; any other stream -- and any
// covered one the executor or the checksum objects to -- is decoded by inflate_stream from the start.
// ---------------------------------------------------------------------------
#include "zstd_decode.h"
struct KieArgs { KiArgs i; const u64* stage; u32 seq_cap; const u8* lits; u32 lit_cap; const u32* nseq; const u32* nlit; };

KX_DEV void inflate_exec_body(const KieArgs& a)
{
    KX_SHARED KiLds lds;
    int const lane = kx_lane();
    for (u32 it = kx_block(); it < a.i.n_slices; it += kx_nblocks()) {
        u32 const f = kx_xcd_chunk(it, a.i.n_slices);
        u32 const ns = a.nseq[f];
        bool done = false;
        if (ns >> 31) {
            u32 const nseq = ns & 0x7FFFFFFFu, nlit = a.nlit[f];
            const u8* const src = a.i.src + a.i.in_off[f]; u32 const srcSize = a.i.in_len[f];
            u8* const dst = a.i.dst + a.i.out_off[f]; u32 const cap = a.i.out_cap[f];
            const u8* const lp = a.lits + (size_t)f * a.lit_cap;
            KxdExecResult xr; xr.err = 0; xr.op = 0; xr.litUsed = 0;
            if (nseq) xr = kxd_exec_window(a.i.src, 0u, 0u, a.stage + (size_t)f * a.seq_cap, nseq, dst, 0u, cap, lp, nlit, 0u);
            u32 op = xr.op; bool good = xr.err == 0 && xr.litUsed <= nlit;
            if (good) {
                u32 const rest = nlit - xr.litUsed;
                if ((u64)op + rest > cap) good = false;
                else { kx_sync(); kx_wave_copy(dst + op, lp + xr.litUsed, rest, lane); op += rest; }
            }
            kx_sync();
            u32 fmt = a.i.format & 0xFFu;
            if (fmt == 3) fmt = (srcSize >= 2 && src[0] == 0x1F && src[1] == 0x8B) ? 2u : 1u;
            if (good && fmt == 2) {
                u32 const got = kx_wave_crc32(dst, op, lds.inw, lane);
                if (got != kx_ld32(src + srcSize - 8) || op != kx_ld32(src + srcSize - 4)) good = false;
            }
            if (good && fmt == 1) {
                u32 const got = kx_wave_adler32(dst, op, lane);
                u32 const want = ((u32)src[srcSize - 4] << 24) | ((u32)src[srcSize - 3] << 16) | ((u32)src[srcSize - 2] << 8) | src[srcSize - 1];
                if (got != want) good = false;
            }
            if (good) { if (lane == 0) { a.i.status[f] = 0; a.i.out_len[f] = op; } done = true; }
            kx_sync();
        }
        if (!done) inflate_stream(a.i, lds, f, lane);
        kx_sync();
    }
}
