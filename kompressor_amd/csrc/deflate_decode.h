// deflate_decode.h -- inflate (RFC 1951, optional RFC 1950 zlib wrapper), one wave per
// stream.  Replaces zlib's inflate() behind the reference's ZlibDecompressor
// (kompressor-zlib--nativelib/src/jvmCommonMain/jni/Wrapper.cpp:84-153, inflate at :144).
//
// Lane 0 parses block headers, builds the Huffman decoding tables in LDS (10-bit direct
// look-up + canonical walk for longer codes) and decodes symbols from an LDS-staged
// copy of the input; literals are stored as they are decoded, matches are queued 64 at
// a time and then copied by all lanes in dependency rounds (a match whose source lies
// before the earliest pending match's destination is copied by its own lane).
#pragma once
#include "deflate_encode.h"

struct KiArgs {
    const u8* src; const u64* in_off; const u32* in_len; u32 n_slices;
    u8* dst; const u64* out_off; const u32* out_cap; u32* out_len; int* status;
    u32 format;                 // bits 0 .. 7: 0 = raw deflate, 1 = zlib wrapper (2-byte header, Adler-32 trailer),
                                // 2 = gzip (RFC 1952 header, CRC-32 + ISIZE trailer), 3 = zlib or gzip by the first bytes;
                                // bits 8 .. 15: the windowBits declared to inflateInit2 (0 = 15): a zlib header that names a larger
                                // window is "invalid window size" (inflate.c HEAD state)
};

enum { KI_OK = 0, KI_DATA_ERROR = -3, KI_BUF_ERROR = -5 };

struct KiLds {
    u16 ltab[1024];             // literal/length: symbol | codeLen << 9 (codeLen 0 = walk the canonical code)
    u16 dtab[512];              // distance, same packing, 9-bit index
    u16 lsorted[288]; u16 dsorted[32];      // symbols in canonical order, for codes longer than the table index
    u16 lcount[16]; u16 dcount[16];
    u8 lens[320];               // code lengths while a dynamic header is read
    u32 inw[640];               // staged input words
    u32 stage[64 * 3];          // queued matches: dst position, length, distance
    u32 bc[16];
};

// canonical Huffman decode table from code lengths (lane 0). Returns false if over-subscribed.
KX_DEV bool ki_build(u16* tab, int tbits, u16* sorted, u16* count, const u8* lens, int n)
{
    u16 offs[16]; u32 code = 0; u32 next[16];
    for (int l = 0; l < 16; l++) count[l] = 0;
    for (int s = 0; s < n; s++) count[lens[s]]++;
    count[0] = 0;
    int left = 1;
    for (int l = 1; l < 16; l++) { left <<= 1; left -= count[l]; if (left < 0) return false; }
    offs[1] = 0;
    for (int l = 1; l < 15; l++) offs[l + 1] = (u16)(offs[l] + count[l]);
    for (int s = 0; s < n; s++) if (lens[s]) sorted[offs[lens[s]]++] = (u16)s;
    for (int l = 1; l < 16; l++) { code = (code + count[l - 1]) << 1; next[l] = code; }
    int const tsize = 1 << tbits;
    for (int i = 0; i < tsize; i++) tab[i] = 0;
    for (int s = 0; s < n; s++) {
        int const l = lens[s];
        if (!l) continue;
        u32 const c = next[l]++;
        if (l <= tbits) {
            u32 const rev = kd_bi_reverse(c, l);
            for (u32 k = rev; k < (u32)tsize; k += 1u << l) tab[k] = (u16)((u32)s | ((u32)l << 9));
        }
    }
    return true;
}

struct KiBits { u64 buf; int cnt; int wp; };     // wp = next staged word (index into the stream's words)

KX_DEV void inflate_stream(const KiArgs& a, KiLds& lds, u32 f, int lane)
{
    const u8* const src = a.src + a.in_off[f]; u32 const srcSize = a.in_len[f];
    u8* const dst = a.dst + a.out_off[f]; u32 const cap = a.out_cap[f];
    int err = 0; u32 op = 0;
    u32 spos = 0, send = srcSize;                 // deflate data = src[spos, send)
    u32 fmt = a.format & 0xFFu; u32 const wmax = (a.format >> 8) ? (a.format >> 8) : 15u;
    if (fmt == 3) fmt = (srcSize >= 2 && src[0] == 0x1F && src[1] == 0x8B) ? 2u : 1u;     // inflateInit2(windowBits + 32)
    if (fmt == 2) {
        // gzip member header: ID1 ID2 CM FLG MTIME(4) XFL OS [FEXTRA] [FNAME] [FCOMMENT] [FHCRC]
        if (srcSize < 18 || src[0] != 0x1F || src[1] != 0x8B || src[2] != 8 || (src[3] & 0xE0)) err = KI_DATA_ERROR;
        else {
            u32 const flg = src[3]; u32 p = 10; u32 const lim = srcSize - 8;
            if (flg & 4u) { if (p + 2 > lim) err = KI_DATA_ERROR; else p += 2u + ((u32)src[p] | ((u32)src[p + 1] << 8)); }
            if (!err && (flg & 8u)) { while (p < lim && src[p]) p++; p++; }
            if (!err && (flg & 16u)) { while (p < lim && src[p]) p++; p++; }
            if (!err && (flg & 2u)) p += 2;
            if (err || p > lim) err = KI_DATA_ERROR;
            spos = p; send = lim;
        }
    }
    if (fmt == 1) {
        if (srcSize < 6) err = KI_DATA_ERROR;
        else {
            u32 const cmf = src[0], flg = src[1];
            if ((cmf & 0x0F) != 8 || (cmf >> 4) + 8u > wmax || ((cmf << 8) | flg) % 31 != 0 || (flg & 0x20)) err = KI_DATA_ERROR;
            spos = 2; send = srcSize - 4;
        }
    }
    const u8* const sp = src + spos; int const nbytes = (int)(send - spos); int const nwords = (nbytes + 3) >> 2;
    // lane 0's bit reader state; stage window [swLo, swLo + 640) words (uniform swLo)
    KiBits br; br.buf = 0; br.cnt = 0; br.wp = 0; int swLo = -1000000;
    bool last = false; bool inBlock = false; int btype = 0; u32 storedLeft = 0;
    // loop of "batches": each batch decodes until 64 matches are queued, 384 words are consumed, or the block ends
    while (!err) {
        int const curW = (int)kx_bcast((u32)br.wp, 0);
        {   // restage 640 words from the read position (a batch consumes at most ~530)
            kx_sync();
            int const lo = curW > 2 ? curW - 2 : 0;
            for (int i = lo + lane; i < lo + 640; i += 64) {
                u32 v = 0; int const o = 4 * i;
                if (i < nwords) { if (o + 4 <= nbytes) v = kx_ld32(sp + o); else for (int k = 0; o + k < nbytes; k++) v |= (u32)sp[o + k] << (8 * k); }
                lds.inw[i - lo] = v;
            }
            swLo = lo;
            kx_sync();
        }
#define KI_NEED(n_) { if (br.cnt < (int)(n_)) { br.buf |= (u64)lds.inw[br.wp - swLo] << br.cnt; br.cnt += 32; br.wp++; } }
#define KI_TAKE(dst_, n_) { u32 const n__ = (n_); KI_NEED(n__) dst_ = (u32)(br.buf & ((1ull << n__) - 1ull)); br.buf >>= n__; br.cnt -= (int)n__; }
        u32 nq = 0;                                   // matches queued by lane 0 in this batch
        if (lane == 0) {
            int e = 0; u32 opL = op; int const wpStart = br.wp; bool endBatch = false; u32 storedCopy = 0, storedFrom = 0;
            while (!e && !endBatch) {
                if (!inBlock) {
                    if (last) { endBatch = true; break; }
                    u32 hdr; KI_TAKE(hdr, 3)
                    last = hdr & 1; btype = (int)(hdr >> 1);
                    if (btype == 3) { e = KI_DATA_ERROR; break; }
                    if (btype == 0) {
                        u32 const drop = (u32)br.cnt & 7u; br.buf >>= drop; br.cnt -= (int)drop;
                        u32 len, nlen; KI_TAKE(len, 16) KI_TAKE(nlen, 16)
                        if ((len ^ 0xFFFFu) != nlen) { e = KI_DATA_ERROR; break; }
                        storedLeft = len; inBlock = true;
                    } else {
                        if (btype == 1) {
                            for (int s = 0; s < 288; s++) lds.lens[s] = (u8)kd_static_llen((u32)s);
                            if (!ki_build(lds.ltab, 10, lds.lsorted, lds.lcount, lds.lens, 288)) { e = KI_DATA_ERROR; break; }
                            for (int s = 0; s < 30; s++) lds.lens[s] = 5;
                            if (!ki_build(lds.dtab, 9, lds.dsorted, lds.dcount, lds.lens, 30)) { e = KI_DATA_ERROR; break; }
                        } else {
                            u32 hlit, hdist, hclen; KI_TAKE(hlit, 5) KI_TAKE(hdist, 5) KI_TAKE(hclen, 4)
                            hlit += 257; hdist += 1; hclen += 4;
                            if (hlit > 286 || hdist > 30) { e = KI_DATA_ERROR; break; }
                            const u8 order[19] = { 16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15 };
                            u8 cl[19]; for (int i = 0; i < 19; i++) cl[i] = 0;
                            for (u32 i = 0; i < hclen; i++) { u32 v; KI_TAKE(v, 3) cl[order[i]] = (u8)v; }
                            // the code-length code (<= 7 bits) goes through the distance table's storage
                            if (!ki_build(lds.dtab, 7, lds.dsorted, lds.dcount, cl, 19)) { e = KI_DATA_ERROR; break; }
                            u32 idx = 0;
                            while (idx < hlit + hdist && !e) {
                                KI_NEED(7)
                                u32 const ent = lds.dtab[br.buf & 127u]; u32 const cl_len = ent >> 9, sym = ent & 511u;
                                if (!cl_len) { e = KI_DATA_ERROR; break; }
                                br.buf >>= cl_len; br.cnt -= (int)cl_len;
                                if (sym < 16) lds.lens[idx++] = (u8)sym;
                                else {
                                    u32 rep, val = 0;
                                    if (sym == 16) { if (!idx) { e = KI_DATA_ERROR; break; } val = lds.lens[idx - 1]; KI_TAKE(rep, 2) rep += 3; }
                                    else if (sym == 17) { KI_TAKE(rep, 3) rep += 3; }
                                    else { KI_TAKE(rep, 7) rep += 11; }
                                    if (idx + rep > hlit + hdist) { e = KI_DATA_ERROR; break; }
                                    while (rep--) lds.lens[idx++] = (u8)val;
                                }
                            }
                            if (e) break;
                            if (lds.lens[256] == 0) { e = KI_DATA_ERROR; break; }
                            // distance lengths follow the literal/length lengths in lds.lens: build the distance table first
                            // from a copy placed above, since building the literal table does not touch lds.lens
                            u8 dl[32]; for (u32 i = 0; i < 32; i++) dl[i] = (i < hdist) ? lds.lens[hlit + i] : (u8)0;
                            if (!ki_build(lds.ltab, 10, lds.lsorted, lds.lcount, lds.lens, (int)hlit)) { e = KI_DATA_ERROR; break; }
                            if (!ki_build(lds.dtab, 9, lds.dsorted, lds.dcount, dl, (int)hdist)) { e = KI_DATA_ERROR; break; }
                        }
                        inBlock = true;
                    }
                }
                if (btype == 0) {
                    // stored bytes are copied by the whole wave after this batch; pending bits are whole bytes here
                    u32 const bufBytes = (u32)br.cnt >> 3;
                    u32 const bytePos = (u32)(4 * br.wp) - bufBytes;                 // stream byte offset of the next unread byte
                    if (bytePos + storedLeft > (u32)nbytes) { e = KI_DATA_ERROR; break; }
                    if ((u64)opL + storedLeft > cap) { e = KI_BUF_ERROR; break; }
                    storedCopy = storedLeft; storedFrom = bytePos;
                    u32 const newPos = bytePos + storedLeft;
                    br.wp = (int)(newPos >> 2); br.buf = 0; br.cnt = 0;
                    if (newPos & 3u) { /* re-prime from the word holding newPos: it may lie outside the staged window, so read memory */
                        u32 v = 0; u32 const o = newPos & ~3u; for (u32 k = 0; o + k < (u32)nbytes && k < 4; k++) v |= (u32)sp[o + k] << (8 * k);
                        br.buf = (u64)v >> (8 * (newPos & 3u)); br.cnt = 32 - (int)(8 * (newPos & 3u)); br.wp++; }
                    storedLeft = 0; inBlock = false; endBatch = true;
                    lds.bc[4] = storedCopy; lds.bc[5] = storedFrom; lds.bc[6] = opL;
                    opL += storedCopy;
                    break;
                }
                // compressed data
                for (;;) {
                    if (br.wp - wpStart >= 384 || nq == 64) { endBatch = true; break; }
                    KI_NEED(15)
                    u32 ent = lds.ltab[br.buf & 1023u]; u32 sym, clen = ent >> 9;
                    if (clen) { sym = ent & 511u; }
                    else {      // code longer than 10 bits: canonical walk
                        u32 code = 0, first = 0, index = 0; sym = 0xFFFFu;
                        for (u32 l = 1; l <= 15; l++) {
                            code |= (u32)((br.buf >> (l - 1)) & 1u);
                            u32 const cnt = lds.lcount[l];
                            if (code < first + cnt) { sym = lds.lsorted[index + (code - first)]; clen = l; break; }
                            index += cnt; first += cnt; first <<= 1; code <<= 1;
                        }
                        if (sym == 0xFFFFu) { e = KI_DATA_ERROR; break; }
                    }
                    br.buf >>= clen; br.cnt -= (int)clen;
                    if (sym < 256) {
                        if (opL >= cap) { e = KI_BUF_ERROR; break; }
                        dst[opL++] = (u8)sym;
                    } else if (sym == 256) { inBlock = false; break; }
                    else {
                        u32 const lc = sym - 257;
                        if (lc > 28) { e = KI_DATA_ERROR; break; }
                        u32 len, x;
                        if (lc < 8) len = 3 + lc; else if (lc == 28) len = 258;
                        else { u32 const eb = (lc - 4) >> 2; KI_TAKE(x, eb) len = 3 + ((4 + (lc & 3u)) << eb) + x; }
                        KI_NEED(15)
                        u32 dent = lds.dtab[br.buf & 511u]; u32 dsym, dlen = dent >> 9;
                        if (dlen) dsym = dent & 511u;
                        else {
                            u32 code = 0, first = 0, index = 0; dsym = 0xFFFFu;
                            for (u32 l = 1; l <= 15; l++) {
                                code |= (u32)((br.buf >> (l - 1)) & 1u);
                                u32 const cnt = lds.dcount[l];
                                if (code < first + cnt) { dsym = lds.dsorted[index + (code - first)]; dlen = l; break; }
                                index += cnt; first += cnt; first <<= 1; code <<= 1;
                            }
                            if (dsym == 0xFFFFu) { e = KI_DATA_ERROR; break; }
                        }
                        br.buf >>= dlen; br.cnt -= (int)dlen;
                        if (dsym > 29) { e = KI_DATA_ERROR; break; }
                        u32 dist;
                        if (dsym < 4) dist = dsym + 1; else { u32 const eb = (dsym - 2) >> 1; KI_TAKE(x, eb) dist = 1 + ((2 + (dsym & 1u)) << eb) + x; }
                        if (dist > opL) { e = KI_DATA_ERROR; break; }
                        if ((u64)opL + len > cap) { e = KI_BUF_ERROR; break; }
                        lds.stage[3 * nq] = opL; lds.stage[3 * nq + 1] = len; lds.stage[3 * nq + 2] = dist; nq++;
                        opL += len;
                    }
                }
                if (!inBlock && last) endBatch = true;
                // consumed more input than exists?
                if (br.wp > nwords + 2) e = KI_DATA_ERROR;        // ran past the input
            }
            lds.bc[0] = (u32)e; lds.bc[1] = nq; lds.bc[2] = opL; lds.bc[3] = (last && !inBlock) ? 1u : 0u;
            if (btype != 0 || !endBatch) { lds.bc[4] = 0; }
        }
        kx_sync();
        err = (int)lds.bc[0]; nq = lds.bc[1];
        u32 const opNew = lds.bc[2]; bool const finished = lds.bc[3] != 0;
        if (err) break;
        if (lds.bc[4]) { kx_wave_copy(dst + lds.bc[6], sp + lds.bc[5], lds.bc[4], lane); }
        // ---- copy the queued matches: lane i owns match i ---------------------------------
        if (nq) {
            bool const own = (u32)lane < nq;
            u32 const dmat = own ? lds.stage[3 * lane] : 0u, ml = own ? lds.stage[3 * lane + 1] : 0u, off = own ? lds.stage[3 * lane + 2] : 8u;
            for (u64 P = kx_ballot(own); P; ) {
                int const e2 = (int)kx_ctz64(P);
                u32 const mlE = kx_bcast(ml, e2), offE = kx_bcast(off, e2), dE = kx_bcast(dmat, e2);
                if (mlE > 32 || offE < 8) {
                    const u8* const ms = dst + dE - offE;
                    if (offE >= 64) {
                        for (u32 base = 0; base < mlE; base += 64) { u32 const k = base + (u32)lane; if (k < mlE) dst[dE + k] = ms[k]; kx_lockstep(); }
                    } else {
                        u32 const chunk = (64 / offE) * offE; u32 const m = (u32)lane % offE; u8 v = 0;
                        if ((u32)lane < chunk) v = ms[m];
                        for (u32 base = 0; base < mlE; base += chunk) { u32 const k = base + (u32)lane; if ((u32)lane < chunk && k < mlE) dst[dE + k] = v; }
                        kx_lockstep();
                    }
                    P &= P - 1;
                    continue;
                }
                bool const safe = ((P >> lane) & 1ull) && ml <= 32 && off >= 8 && (lane == e2 || dmat - off + ml <= dE);
                if (safe) {
                    const u8* const s_ = dst + dmat - off; u8* const d_ = dst + dmat; u32 k = 0;
                    for (; k + 8 <= ml; k += 8) kx_st64(d_ + k, kx_ld64(s_ + k));
                    for (; k < ml; k++) d_[k] = s_[k];
                }
                P &= ~kx_ballot(safe);
                kx_lockstep();
            }
        }
        op = opNew;
        kx_sync();
        if (finished) break;
    }
#undef KI_TAKE
#undef KI_NEED
    if (!err) {
        // the final block must end inside the last byte of the deflate data: bits past the end read as zero and would
        // otherwise pass for codes (a truncated stream is Z_BUF_ERROR in zlib); bytes left over are not accepted either
        u32 const bitsUsed = 32u * kx_shfl((u32)br.wp, 0) - kx_shfl((u32)br.cnt, 0);
        u32 const bytesUsed = (bitsUsed + 7u) >> 3;
        if (bytesUsed > (u32)nbytes) err = KI_BUF_ERROR; else if (bytesUsed < (u32)nbytes) err = KI_DATA_ERROR;
    }
    if (!err && fmt == 2) {
        // CRC-32 and length (mod 2^32) of the output against the little-endian trailer
        u32 const got = kx_wave_crc32(dst, op, lds.inw, lane);
        if (got != kx_ld32(src + srcSize - 8) || op != kx_ld32(src + srcSize - 4)) err = KI_DATA_ERROR;
    }
    if (!err && fmt == 1) {
        // Adler-32 of the output against the big-endian trailer
        u32 const got = kx_wave_adler32(dst, op, lane);
        u32 const want = ((u32)src[srcSize - 4] << 24) | ((u32)src[srcSize - 3] << 16) | ((u32)src[srcSize - 2] << 8) | src[srcSize - 1];
        if (got != want) err = KI_DATA_ERROR;
    }
    if (lane == 0) { a.status[f] = err; a.out_len[f] = err ? 0u : op; }
}

KX_DEV void inflate_body(const KiArgs& a)
{
    KX_SHARED KiLds lds;
    int const lane = kx_lane();
    for (u32 it = kx_block(); it < a.n_slices; it += kx_nblocks()) {
        u32 const f = kx_xcd_chunk(it, a.n_slices);
        inflate_stream(a, lds, f, lane);
        kx_sync();
    }
}
