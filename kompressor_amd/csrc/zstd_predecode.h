// zstd_predecode.h -- the sequence bitstreams of a batch of frames, decoded one LANE per frame ahead of k_zstd_decode.
//
// In k_zstd_decode (one wave per frame) the three tANS state chains of a block's sequences section are single-lane
// work: 86 % of that kernel's instructions served three lanes of a wave, and instruction issue
// was what bound it.  A sequence costs the same instructions whether one lane or sixty-four execute them, so this
// kernel turns the work sideways: a lane owns a whole frame, walks its blocks, builds the three FSE decoding tables of
// each compressed block in its own slice of the LDS and decodes the block's sequences -- literal length, match length
// and the offset with the repeat-offset rules already applied -- into a staging area in HBM.  k_zstd_decode then loads
// 64 finished sequences per step instead of decoding them and keeps everything else (headers, literals, execution,
// every check and error code).
//
// This kernel is an accelerator, not an authority: it records a block as staged only if everything about it was
// regular (the bitstream consumed to the last bit included); at the first doubt it stops, the remaining blocks of the
// entry stay unstaged and k_zstd_decode decodes them itself, as before, and reports whatever is wrong with them.
// Only the first frame of an entry is looked at.  Reads never leave the entry, table indices never leave the lane's
// tables, whatever the bytes are.
//
// Replaces, with k_zstd_decode, libzstd's ZSTD_decompressStream behind the reference's ZstdDecompressor
// (kompressor-zstd--nativelib/src/jvmCommonMain/jni/Wrapper.cpp:178).
#pragma once
#include "zstd_decode.h"

struct KPreArgs {
    const u8* src; const u64* in_off; const u32* in_len; u32 n_slices;
    u32* stage; u32 seq_cap;            // per entry: seq_cap x (litLength, matchLength, offset)
    KPreBlk* blk; u32 blk_cap;          // per entry: blk_cap records (compressed blocks in frame order)
    u32* nblk;                          // per entry: records written
};

#define KXP_FRAMES 32                   /* frames per workgroup: what the LDS holds */
#define KXP_WAVES 4                     /* ... spread over 4 waves (8 lanes each busy), one per SIMD */
#define KXP_WIN 64                      /* words of a lane's bitstream window in LDS */

struct KPreLaneLds {
    u16 fb[1280]; u8 fc[1280];          // FSE decoding tables LL [0,512) ML [512,1024) OF [1024,1280), as in KDecodeLds
    short keepNorm[3][56]; u16 symnext[56];
    u32 win[KXP_WIN];                   // the sequence bitstream around the read position: word i of the stream at win[i & 63]
};
struct KPreLds { KPreLaneLds f[KXP_FRAMES]; u32 llx[36]; u32 mlx[53]; };

// One symbol type's table for the block, into the lane's LDS; keep* remember what a later "repeat" mode rebuilds from.
// Returns bytes consumed or KXD_FAIL.
KX_DEV u32 kxp_seq_table(KPreLaneLds& L, int t, u32 mode, const u8* p, u32 size, u32* tableLog, u32* keepKind, u32* keepLog, u32* keepMax)
{
    static const short LL_defaultNorm[36] = { 4,3,2,2,2,2,2,2, 2,2,2,2,2,1,1,1, 2,2,2,2,2,2,2,2, 2,3,2,1,1,1,1,1, -1,-1,-1,-1 };
    static const short ML_defaultNorm[53] = { 1,4,3,2,2,2,2,2, 2,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1,
                                              1,1,1,1,1,1,1,1, 1,1,1,1,1,1,-1,-1, -1,-1,-1,-1,-1 };
    static const short OF_defaultNorm[29] = { 1,1,1,1,1,1,2,2, 2,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1, -1,-1,-1,-1,-1 };
    u32 const maxSym = (t == 0) ? 35 : (t == 1) ? 31 : 52;
    u32 const maxLog = (t == 0) ? 9 : (t == 1) ? 8 : 9;
    u16* const db = L.fb + kxd_seq_base(t); u8* const dc = L.fc + kxd_seq_base(t);
    u32 used = 0;
    if (mode == 0) {
        const short* dn = (t == 0) ? LL_defaultNorm : (t == 1) ? OF_defaultNorm : ML_defaultNorm;
        u32 const dmax = (t == 0) ? 35 : (t == 1) ? 28 : 52; u32 const dlog = (t == 1) ? 5 : 6;
        for (u32 s = 0; s <= dmax; s++) L.keepNorm[t][s] = dn[s];
        keepKind[t] = 2; keepLog[t] = dlog; keepMax[t] = dmax;
    } else if (mode == 1) {
        if (size < 1 || p[0] > maxSym) return KXD_FAIL;
        keepKind[t] = 1; keepLog[t] = 0; keepMax[t] = p[0];
        used = 1;
    } else if (mode == 2) {
        u32 maxSV = maxSym, tl = 0;
        u32 const h = kfse_read_ncount(L.keepNorm[t], &maxSV, &tl, p, size, maxLog);
        if (h == 0) { keepKind[t] = 0; return KXD_FAIL; }
        keepKind[t] = 2; keepLog[t] = tl; keepMax[t] = maxSV;
        used = h;
    } else if (keepKind[t] == 0) return KXD_FAIL;        // repeat without a previous table
    if (keepKind[t] == 1) { db[0] = 0; dc[0] = (u8)keepMax[t]; }
    else kfse_build_dtable(db, dc, L.keepNorm[t], keepMax[t], keepLog[t], L.symnext, dc);     // the spread IS the symbol table
    *tableLog = keepLog[t];
    return used;
}

KX_DEV void zstd_seq_predecode_body(const KPreArgs& a)
{
    KX_SHARED KPreLds lds;
    int const lane = kx_lane(), wv = kx_wave();
    {
        int const t = wv * 64 + lane;
        if (t < 36) lds.llx[t] = kx_ll_base((u32)t) | (kxd_ll_bits((u32)t) << 24);
        if (t >= 64 && t < 64 + 53) lds.mlx[t - 64] = kx_ml_base((u32)(t - 64)) | (kxd_ml_bits((u32)(t - 64)) << 24);
    }
    kx_block_sync();
    constexpr int PER = KXP_FRAMES / KXP_WAVES;
    if (lane >= PER) return;
    KPreLaneLds& L = lds.f[wv * PER + lane];
    for (u32 f = (kx_block() * KXP_WAVES + (u32)wv) * PER + (u32)lane; f < a.n_slices; f += kx_nblocks() * KXP_FRAMES) {
        const u8* const src = a.src + a.in_off[f];
        u32 const srcSize = a.in_len[f];
        u32* const stage = a.stage + (size_t)f * a.seq_cap * 3u;
        KPreBlk* const blk = a.blk + (size_t)f * a.blk_cap;
        u32 nb = 0, nstaged = 0;              // compressed blocks seen / sequences staged so far
        u32 pos = 0;
        // ---- frame header (first frame of the entry only) ----
        bool ok = srcSize >= 5 && kx_ld32(src) == 0xFD2FB528u;
        if (ok) {
            u32 const fhd = src[4]; u32 const dictId = fhd & 3, single = (fhd >> 5) & 1, fcsId = fhd >> 6;
            if (fhd & 0x08) ok = false;
            u32 const didSize = dictId == 3 ? 4 : dictId;
            u32 const fcsSize = fcsId == 0 ? single : (fcsId == 1 ? 2 : fcsId == 2 ? 4 : 8);
            pos = 5 + (single ? 0u : 1u) + didSize + fcsSize;
            if (pos > srcSize) ok = false;
        }
        u32 rep1 = 1, rep2 = 4, rep3 = 8;
        u32 keepKind[3] = { 0, 0, 0 }, keepLog[3] = { 0, 0, 0 }, keepMax[3] = { 0, 0, 0 };
        bool last = false;
        while (ok && !last && nb < a.blk_cap) {
            if (pos + 3 > srcSize) break;
            u32 const bh = (u32)src[pos] | ((u32)src[pos + 1] << 8) | ((u32)src[pos + 2] << 16);
            last = bh & 1; u32 const btype = (bh >> 1) & 3; u32 const bsize = bh >> 3;
            pos += 3;
            if (btype == 3) break;
            if (btype == 0) { if (pos + bsize > srcSize) break; pos += bsize; continue; }
            if (btype == 1) { if (pos + 1 > srcSize) break; pos += 1; continue; }
            if (pos + bsize > srcSize || bsize > 128u * 1024u || bsize < 2) break;
            const u8* const bp = src + pos; u32 const bend = bsize;
            // literals section: only its size matters here
            u32 const lh0 = bp[0]; u32 const ltype = lh0 & 3, sf = (lh0 >> 2) & 3;
            u32 lpos;
            if (ltype < 2) {
                u32 lhSize, regen;
                if (sf == 0 || sf == 2) { lhSize = 1; regen = lh0 >> 3; }
                else if (sf == 1) { lhSize = 2; regen = kx_ld16(bp) >> 4; }
                else { if (bend < 3) break; lhSize = 3; regen = ((u32)bp[0] | ((u32)bp[1] << 8) | ((u32)bp[2] << 16)) >> 4; }
                lpos = lhSize + (ltype == 0 ? regen : 1u);
            } else {
                if (bend < 5) break;
                u32 const w = kx_ld32(bp); u32 lhSize, comp;
                if (sf < 2) { lhSize = 3; comp = (w >> 14) & 0x3FF; }
                else if (sf == 2) { lhSize = 4; comp = w >> 18; }
                else { lhSize = 5; comp = (w >> 22) + ((u32)bp[4] << 10); }
                lpos = lhSize + comp;
            }
            if (lpos >= bend) break;
            // sequences header
            u32 nbSeq, p2 = lpos;
            {
                u32 const b0 = bp[p2++];
                if (b0 < 128) nbSeq = b0;
                else if (b0 < 255) { if (p2 >= bend) break; nbSeq = ((b0 - 128) << 8) + bp[p2++]; }
                else { if (p2 + 2 > bend) break; nbSeq = kx_ld16(bp + p2) + 0x7F00; p2 += 2; }
            }
            KPreBlk rec; rec.seq_off = nstaged; rec.nbSeq = nbSeq; rec.ok = 0; rec.rep[0] = rep1; rec.rep[1] = rep2; rec.rep[2] = rep3; rec.pad[0] = 0; rec.pad[1] = 0;
            if (nbSeq == 0) { rec.ok = 1; blk[nb++] = rec; pos += bsize; continue; }
            if (p2 >= bend) break;
            u32 const modes = bp[p2++];
            if (modes & 3) break;
            if (nstaged + nbSeq > a.seq_cap) break;
            u32 tlLL = 0, tlOF = 0, tlML = 0; bool tok = true;
            for (int t = 0; t < 3 && tok; t++) {
                u32 const mode = (modes >> (6 - 2 * t)) & 3u;
                u32 const r = kxp_seq_table(L, t, mode, bp + p2, bend - p2, t == 0 ? &tlLL : t == 1 ? &tlOF : &tlML, keepKind, keepLog, keepMax);
                if (r == KXD_FAIL) tok = false; else p2 += r;
            }
            if (!tok || p2 >= bend) break;
            // ---- the bitstream, read backwards from its last set bit ----
            const u8* const sq = bp + p2; u32 const ssz = bend - p2;
            u32 const lastByte = sq[ssz - 1];
            if (lastByte == 0) break;
            int const totalWords = (int)((ssz + 3) >> 2);
            int bitPos = (int)(8 * (ssz - 1) + kx_hb32(lastByte));     // unread bits
            bool bad = false;
            // The window holds the words [lo, lo + 64) of the stream (words below 0 read as zero: the container of an
            // exhausted stream reaches word -3), circularly.  The stream is read downwards, a sequence takes at most 89
            // bits = words [top - 4, top]; the next eight words below the window wait in registers, loaded one refill
            // ahead, so no sequence waits for memory.
            int lo;
            {
                int const curWord = bitPos >> 5;
                int hiW = curWord + 2; if (hiW > totalWords) hiW = totalWords;
                lo = hiW - KXP_WIN;
                for (int w0 = lo; w0 < hiW; w0 += 8) {
                    u32 t[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        int const w = w0 + j; int const o = 4 * w; u32 v = 0;
                        if (w >= 0 && w < hiW) {
                            if (o + 4 <= (int)ssz) v = kx_ld32(sq + o);
                            else for (int k = 0; o + k < (int)ssz; k++) v |= (u32)sq[o + k] << (8 * k);
                        }
                        t[j] = v;
                    }
#pragma unroll
                    for (int j = 0; j < 8; j++) L.win[(w0 + j) & (KXP_WIN - 1)] = t[j];
                }
            }
            u32 pf[8];
#define KXP_PREFETCH() { _Pragma("unroll") for (int j = 0; j < 8; j++) { int const w = lo - 8 + j; pf[j] = (w >= 0) ? kx_ld32(sq + 4 * w) : 0u; } }
            KXP_PREFETCH()
#define KXP_WORD(i) L.win[(i) & (KXP_WIN - 1)]
#define KXP_CONTAINER(C_) u64 C_; { int const topw_ = (bitPos - 1) >> 5; \
                u32 const hi_ = KXP_WORD(topw_), mid_ = KXP_WORD(topw_ - 1), lo_ = KXP_WORD(topw_ - 2); \
                u32 const used_ = (u32)(32 * (topw_ + 1) - bitPos); \
                C_ = ((((u64)hi_ << 32) | mid_) << used_) | ((u64)lo_ >> (32u - used_)); }
#define KXP_AT(C_, c_, n_) ((u32)((((C_) << (c_)) >> 1) >> (63u - (n_))))
            u32 sLL = 0, sOF = 0, sML = 0; bool primed = false;
            u32* const out = stage + (size_t)nstaged * 3u;
            for (u32 i = 0; i < nbSeq; i++) {
                if ((bitPos >> 5) - lo < 8 && lo > -8) {
                    // the registers move into the window (over words that lie above the read position by now) ...
#pragma unroll
                    for (int j = 0; j < 8; j++) L.win[(lo - 8 + j) & (KXP_WIN - 1)] = pf[j];
                    lo -= 8;
                    KXP_PREFETCH()               // ... and the next eight are requested
                }
                if (!primed) {
                    u32 const need0 = tlLL + tlOF + tlML;
                    bad |= bitPos < (int)need0;
                    KXP_CONTAINER(C0)
                    sLL = KXP_AT(C0, 0u, tlLL); sOF = KXP_AT(C0, tlLL, tlOF); sML = KXP_AT(C0, tlLL + tlOF, tlML);    // stream order LL, OF, ML
                    if (tlLL == 0) sLL = 0; if (tlOF == 0) sOF = 0; if (tlML == 0) sML = 0;
                    bitPos -= (int)need0; bitPos = bitPos < 0 ? 0 : bitPos;
                    primed = true;
                }
                KXP_CONTAINER(C)
                u32 const eL = L.fb[KXD_LL0 + sLL], cL = L.fc[KXD_LL0 + sLL];
                u32 const eO = L.fb[KXD_OF0 + sOF], cO = L.fc[KXD_OF0 + sOF];
                u32 const eM = L.fb[KXD_ML0 + sML], cM = L.fc[KXD_ML0 + sML];
                if (cL > 35 || cM > 52 || cO > 31) { bad = true; break; }
                u32 const xL = lds.llx[cL], xM = lds.mlx[cM];
                u32 const aO = cO, aM = xM >> 24, aL = xL >> 24;
                bool const upd = i + 1 < nbSeq;                        // the block's final sequence updates no state
                u32 const nL = upd ? eL >> 12 : 0u, nM = upd ? eM >> 12 : 0u, nO = upd ? eO >> 12 : 0u;
                u32 const needA = aO + aM + aL, needB = nL + nM + nO;
                bad |= bitPos < (int)(needA + needB);
                // bit order inside a sequence: OF extra, ML extra, LL extra, then LL state, ML state, OF state
                u32 const xo = aO ? KXP_AT(C, 0u, aO) : 0u;
                u32 const xm = aM ? KXP_AT(C, aO, aM) : 0u;
                u32 const xl = aL ? KXP_AT(C, aO + aM, aL) : 0u;
                u64 Cs = C; u32 offB = needA;
                if (needA + needB > 64) { bitPos -= (int)needA; bitPos = bitPos < 0 ? 0 : bitPos; KXP_CONTAINER(C2) bitPos += (int)needA; Cs = C2; offB = 0; }
                u32 const yL = nL ? KXP_AT(Cs, offB, nL) : 0u;
                u32 const yM = nM ? KXP_AT(Cs, offB + nL, nM) : 0u;
                u32 const yO = nO ? KXP_AT(Cs, offB + nL + nM, nO) : 0u;
                bitPos -= (int)(needA + needB); bitPos = bitPos < 0 ? 0 : bitPos;
                u32 const ofv = (1u << cO) + xo, ml = (xM & 0xFFFFFFu) + xm, ll = (xL & 0xFFFFFFu) + xl;
                // repeat-offset rules
                bool const isRep = ofv <= 3;
                u32 const idx = ofv - 1 + (ll == 0);
                u32 const rm1 = (rep1 - 1) ? rep1 - 1 : 1u;
                u32 const roff = idx == 0 ? rep1 : idx == 1 ? rep2 : idx == 2 ? rep3 : rm1;
                u32 const off = isRep ? roff : ofv - 3;
                bool const sh2 = !isRep || idx >= 2, sh1 = !isRep || idx >= 1;
                rep3 = sh2 ? rep2 : rep3; rep2 = sh1 ? rep1 : rep2; rep1 = off;
                out[3 * i] = ll; out[3 * i + 1] = ml; out[3 * i + 2] = off;
                sLL = ((eL & 0xFFFu) + yL) & 511u; sML = ((eM & 0xFFFu) + yM) & 511u; sOF = ((eO & 0xFFFu) + yO) & 255u;
            }
#undef KXP_AT
#undef KXP_CONTAINER
#undef KXP_WORD
#undef KXP_PREFETCH
            if (bad || bitPos != 0) break;          // irregular: this block and the rest are left to k_zstd_decode
            rec.ok = 1; rec.rep[0] = rep1; rec.rep[1] = rep2; rec.rep[2] = rep3;
            blk[nb++] = rec;
            nstaged += nbSeq;
            pos += bsize;
        }
        a.nblk[f] = nb;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_zstd_lit_predecode -- the Huffman-coded literals of a batch of frames, one LANE per stream.
//
// In k_zstd_decode four lanes of a wave walk the four Huffman streams of a block while sixty sit idle.  Here a
// workgroup takes 32 frames: in phase A thread j parses frame j up to its next Huffman-coded literals section and
// builds that block's decoding table in the frame's slice of the LDS; after a workgroup barrier, in phase B thread
// 4 j + s decodes stream s of frame j into the entry's literal staging area in HBM; another barrier, and phase A moves on
// to the frame's next block.  k_zstd_decode then reads the literals instead of decoding them.  Like the sequence
// pre-decoder above this is an accelerator: a block is recorded only if its streams decoded to the last bit; anything
// irregular ends the frame's pre-decoding and k_zstd_decode handles (and reports) the rest as before.
struct KLitArgs {
    const u8* src; const u64* in_off; const u32* in_len; u32 n_slices;
    u8* lits; u32 lit_cap;              // per entry: lit_cap bytes of literal staging (the literals of all blocks, back to back)
    KPreLit* rec; u32 blk_cap;          // per entry: blk_cap records (compressed blocks of the first frame, in order)
    u32* nrec;                          // per entry: records written
};

#define KXL_FRAMES 32
struct KLitFrameLds {
    union { u16 huf[2048]; struct { u16 wb[64]; u8 wc[64]; u8 tsym[64]; } b; } u;     // as KDecodeLds: the weights' FSE table lives where the decoding table goes
    u8 weights[256]; short norm[64]; u16 symnext[64]; u32 rank[16];
    // this round's job for the four stream lanes
    u32 valid, tableLog, soff[4], ssz[4], cnt[4], dst, bad;
};
struct KLitLds { KLitFrameLds f[KXL_FRAMES]; u32 more; };

KX_DEV void zstd_lit_predecode_body(const KLitArgs& a)
{
    KX_SHARED KLitLds lds;
    int const tid = kx_wave() * 64 + kx_lane();                 // 128 threads
    for (u32 base = kx_block() * KXL_FRAMES; base < a.n_slices; base += kx_nblocks() * KXL_FRAMES) {
        // ---- phase A state of thread tid < 32: frame base + tid ----
        u32 const fa = base + (u32)tid;
        bool const owner = tid < KXL_FRAMES && fa < a.n_slices;
        const u8* srcA = a.src; u32 srcSize = 0, pos = 0, nb = 0, litUsed = 0; bool go = false, last = false;
        u32 hufLog = 0; bool hufValid = false;
        if (owner) {
            srcA = a.src + a.in_off[fa]; srcSize = a.in_len[fa];
            go = srcSize >= 5 && kx_ld32(srcA) == 0xFD2FB528u;
            if (go) {
                u32 const fhd = srcA[4]; u32 const dictId = fhd & 3, single = (fhd >> 5) & 1, fcsId = fhd >> 6;
                if (fhd & 0x08) go = false;
                u32 const didSize = dictId == 3 ? 4 : dictId;
                u32 const fcsSize = fcsId == 0 ? single : (fcsId == 1 ? 2 : fcsId == 2 ? 4 : 8);
                pos = 5 + (single ? 0u : 1u) + didSize + fcsSize;
                if (pos > srcSize) go = false;
            }
        }
        // ---- phase B identity of every thread: stream sB of frame base + jB ----
        int const jB = tid >> 2, sB = tid & 3;
        u32 const fb = base + (u32)jB;
        for (;;) {
            if (tid == 0) lds.more = 0;
            if (tid < KXL_FRAMES) { lds.f[tid].valid = 0; lds.f[tid].bad = 0; }
            kx_block_sync();
            // ---- phase A: up to the frame's next Huffman-coded literals section ----
            u32 regenA = 0;
            if (owner && go) {
                KLitFrameLds& L = lds.f[tid];
                bool found = false;
                while (go && !last && !found) {
                    if (nb >= a.blk_cap || pos + 3 > srcSize) { go = false; break; }
                    u32 const bh = (u32)srcA[pos] | ((u32)srcA[pos + 1] << 8) | ((u32)srcA[pos + 2] << 16);
                    last = bh & 1; u32 const btype = (bh >> 1) & 3; u32 const bsize = bh >> 3;
                    pos += 3;
                    if (btype == 3) { go = false; break; }
                    if (btype == 0) { if (pos + bsize > srcSize) { go = false; break; } pos += bsize; continue; }
                    if (btype == 1) { if (pos + 1 > srcSize) { go = false; break; } pos += 1; continue; }
                    if (pos + bsize > srcSize || bsize > 128u * 1024u || bsize < 2) { go = false; break; }
                    const u8* const bp = srcA + pos; u32 const bend = bsize;
                    u32 const lh0 = bp[0]; u32 const ltype = lh0 & 3, sf = (lh0 >> 2) & 3;
                    KPreLit r; r.off = litUsed; r.regen = 0; r.ok = 0; r.pad = 0;
                    if (ltype < 2) { a.rec[(size_t)fa * a.blk_cap + nb++] = r; pos += bsize; continue; }      // raw / RLE literals: nothing to decode
                    if (bend < 5) { go = false; break; }
                    u32 const w = kx_ld32(bp); u32 lhSize, regen, comp, nstreams;
                    if (sf < 2) { lhSize = 3; regen = (w >> 4) & 0x3FF; comp = (w >> 14) & 0x3FF; nstreams = sf ? 4 : 1; }
                    else if (sf == 2) { lhSize = 4; regen = (w >> 4) & 0x3FFF; comp = w >> 18; nstreams = 4; }
                    else { lhSize = 5; regen = (w >> 4) & 0x3FFFF; comp = (w >> 22) + ((u32)bp[4] << 10); nstreams = 4; }
                    if (regen > 128u * 1024u || lhSize + comp > bend || comp == 0 || litUsed + regen > a.lit_cap) { go = false; break; }
                    u32 hused = 0;
                    if (ltype == 2) {
                        u32 tl = 0, nw = 0;
                        hused = khuf_read_dtable(L, bp + lhSize, comp, &tl, &nw);
                        if (hused == 0) { go = false; break; }
                        hufLog = tl; (void)nw; hufValid = true;
                    } else {
                        if (!hufValid) { go = false; break; }
                        // (the table of the previous Huffman block is still in this frame's slice of the LDS)
                    }
                    u32 const so = (u32)(bp - srcA) + lhSize + hused, ssize = comp - hused;
                    if (nstreams == 1) {
                        L.soff[0] = so; L.ssz[0] = ssize; L.cnt[0] = regen;
                        for (int k = 1; k < 4; k++) { L.soff[k] = 0; L.ssz[k] = 0; L.cnt[k] = 0xFFFFFFFFu; }       // no such stream
                    } else {
                        if (ssize < 10) { go = false; break; }
                        u32 const c0 = kx_ld16(srcA + so), c1 = kx_ld16(srcA + so + 2), c2 = kx_ld16(srcA + so + 4);
                        if (6 + c0 + c1 + c2 > ssize) { go = false; break; }
                        u32 const seg = (regen + 3) / 4;
                        if (3 * seg > regen) { go = false; break; }
                        L.soff[0] = so + 6; L.soff[1] = so + 6 + c0; L.soff[2] = so + 6 + c0 + c1; L.soff[3] = so + 6 + c0 + c1 + c2;
                        L.ssz[0] = c0; L.ssz[1] = c1; L.ssz[2] = c2; L.ssz[3] = ssize - 6 - c0 - c1 - c2;
                        L.cnt[0] = seg; L.cnt[1] = seg; L.cnt[2] = seg; L.cnt[3] = regen - 3 * seg;
                    }
                    L.tableLog = hufLog; L.dst = litUsed; L.valid = 1;
                    regenA = regen; found = true;
                    pos += bsize;
                }
                if (found) lds.more = 1;
            }
            kx_block_sync();
            if (lds.more == 0) break;
            // ---- phase B: one lane per stream ----
            if (fb < a.n_slices && lds.f[jB].valid) {
                KLitFrameLds const& L = lds.f[jB];
                u32 const cnt = L.cnt[sB];
                if (cnt != 0xFFFFFFFFu) {
                    u32 const seg = L.cnt[0];                                   // stream s writes behind s full segments
                    u8* const out = a.lits + (size_t)fb * a.lit_cap + L.dst + (L.cnt[1] == 0xFFFFFFFFu ? 0u : (u32)sB * seg);
                    bool const ok = khuf_decode_stream(L, L.tableLog, a.src + a.in_off[fb] + L.soff[sB], L.ssz[sB], out, cnt);
                    if (!ok) lds.f[jB].bad = 1;
                }
            }
            kx_block_sync();
            // ---- the owner records the block ----
            if (owner && lds.f[tid].valid) {
                KPreLit r; r.off = litUsed; r.regen = regenA; r.ok = lds.f[tid].bad ? 0u : 1u; r.pad = 0;
                a.rec[(size_t)fa * a.blk_cap + nb++] = r;
                if (lds.f[tid].bad) go = false; else litUsed += regenA;
            }
            kx_block_sync();
        }
        if (owner) a.nrec[fa] = nb;
        kx_block_sync();
    }
}
