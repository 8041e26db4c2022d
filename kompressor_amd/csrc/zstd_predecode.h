// zstd_predecode.h -- the sequence bitstreams of a batch of frames, decoded one LANE per frame ahead of k_zstd_decode.
//
// In k_zstd_decode (one wave per frame) the three tANS state chains of a block's sequences section are single-lane
// work: 86 % of that kernel's instructions served three lanes of a wave, and instruction issue
// was what bound it.  A sequence costs the same instructions whether one lane or sixty-four execute them, so this
// kernel turns the work sideways: a lane owns a whole frame, walks its blocks, builds the three FSE decoding tables of
// each compressed block (in HBM: 5 KiB per frame, so that nothing limits how many frames are in flight; the first
// version kept them in LDS, 32 frames per CU, and ran at one wave per SIMD with every latency exposed) and decodes the
// block's sequences -- literal length, match length and the offset with the repeat-offset rules already applied --
// into a staging area in HBM.  What it costs is three random table words per sequence: it is bound by memory
// transactions where k_zstd_decode is bound by instruction issue, so the two run side by side on the same CUs.  k_zstd_decode then loads
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
    u32* tables;                        // per entry: KXP_TBL_WORDS words: the three FSE decoding tables + their construction scratch
};

// per entry in HBM: one word per FSE state (newStateBase | nbBits << 16 | symbol << 24), LL [0,512) ML [512,1024) OF [1024,1280),
// then the normalised counts and the per-symbol cursor of the table under construction (56 x i16 + 56 x u16)
#define KXP_TBL_WORDS (1280 + 56)

// FSE table description -> the entry's table in HBM (same construction as kfse_build_dtable)
KX_DEV void kxp_build_dtable(u32* tb, const short* norm, u32 maxSymbolValue, u32 tableLog, u16* symnext)
{
    u32 const tableSize = 1u << tableLog, mask = tableSize - 1;
    u32 const step = (tableSize >> 1) + (tableSize >> 3) + 3;
    u32 high = tableSize - 1;
    for (u32 s = 0; s <= maxSymbolValue; s++) {
        if (norm[s] == -1) { tb[high--] = s << 24; symnext[s] = 1; }
        else symnext[s] = (u16)norm[s];
    }
    u32 pos = 0;
    for (u32 s = 0; s <= maxSymbolValue; s++) {
        for (int i = 0; i < norm[s]; i++) {
            tb[pos] = s << 24;
            pos = (pos + step) & mask;
            while (pos > high) pos = (pos + step) & mask;
        }
    }
    for (u32 u = 0; u < tableSize; u++) {
        u32 const s = tb[u] >> 24; u32 const next = symnext[s]++;
        u32 const nb = tableLog - kx_hb32(next);
        tb[u] = (((next << nb) - tableSize) & 0xFFFFu) | (nb << 16) | (s << 24);
    }
}

// One symbol type's table for the block (a "repeat" mode keeps what is there).  kind[t]: 0 none yet, 1 RLE, 2 FSE.
// Returns bytes consumed or KXD_FAIL.
KX_DEV u32 kxp_seq_table(u32* tables, int t, u32 mode, const u8* p, u32 size, u32* tableLog, u32* kind, u32* klog)
{
    static const short LL_defaultNorm[36] = { 4,3,2,2,2,2,2,2, 2,2,2,2,2,1,1,1, 2,2,2,2,2,2,2,2, 2,3,2,1,1,1,1,1, -1,-1,-1,-1 };
    static const short ML_defaultNorm[53] = { 1,4,3,2,2,2,2,2, 2,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1,
                                              1,1,1,1,1,1,1,1, 1,1,1,1,1,1,-1,-1, -1,-1,-1,-1,-1 };
    static const short OF_defaultNorm[29] = { 1,1,1,1,1,1,2,2, 2,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1, -1,-1,-1,-1,-1 };
    u32 const maxSym = (t == 0) ? 35 : (t == 1) ? 31 : 52;
    u32 const maxLog = (t == 0) ? 9 : (t == 1) ? 8 : 9;
    u32* const tb = tables + kxd_seq_base(t);
    short* const norm = (short*)(tables + 1280); u16* const symnext = (u16*)(tables + 1280 + 28);
    u32 used = 0;
    if (mode == 0) {
        const short* dn = (t == 0) ? LL_defaultNorm : (t == 1) ? OF_defaultNorm : ML_defaultNorm;
        u32 const dmax = (t == 0) ? 35 : (t == 1) ? 28 : 52; u32 const dlog = (t == 1) ? 5 : 6;
        for (u32 s = 0; s <= dmax; s++) norm[s] = dn[s];
        kxp_build_dtable(tb, norm, dmax, dlog, symnext);
        kind[t] = 2; klog[t] = dlog;
    } else if (mode == 1) {
        if (size < 1 || p[0] > maxSym) return KXD_FAIL;
        tb[0] = (u32)p[0] << 24;                             // nbBits 0, next state 0
        kind[t] = 1; klog[t] = 0;
        used = 1;
    } else if (mode == 2) {
        u32 maxSV = maxSym, tl = 0;
        u32 const h = kfse_read_ncount(norm, &maxSV, &tl, p, size, maxLog);
        if (h == 0) { kind[t] = 0; return KXD_FAIL; }
        kxp_build_dtable(tb, norm, maxSV, tl, symnext);
        kind[t] = 2; klog[t] = tl;
        used = h;
    } else if (kind[t] == 0) return KXD_FAIL;            // repeat without a previous table
    *tableLog = klog[t];
    return used;
}

KX_DEV void zstd_seq_predecode_body(const KPreArgs& a)
{
    KX_SHARED u32 llx[36]; KX_SHARED u32 mlx[53];          // per code: baseValue | extraBits << 24
    {
        int const lane = kx_lane();
        if (lane < 36) llx[lane] = kx_ll_base((u32)lane) | (kxd_ll_bits((u32)lane) << 24);
        if (lane < 53) mlx[lane] = kx_ml_base((u32)lane) | (kxd_ml_bits((u32)lane) << 24);
    }
    kx_sync();
    u32 const f = kx_block() * 64u + (u32)kx_lane();
    if (f >= a.n_slices) return;
    const u8* const src = a.src + a.in_off[f];
    u32 const srcSize = a.in_len[f];
    u32* const stage = a.stage + (size_t)f * a.seq_cap * 3u;
    KPreBlk* const blk = a.blk + (size_t)f * a.blk_cap;
    u32* const tables = a.tables + (size_t)f * KXP_TBL_WORDS;
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
    u32 kind[3] = { 0, 0, 0 }, klog[3] = { 0, 0, 0 };
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
            u32 const r = kxp_seq_table(tables, t, mode, bp + p2, bend - p2, t == 0 ? &tlLL : t == 1 ? &tlOF : &tlML, kind, klog);
            if (r == KXD_FAIL) tok = false; else p2 += r;
        }
        if (!tok || p2 >= bend) break;
        // ---- the bitstream, read backwards from its last set bit ----
        const u8* const sq = bp + p2; u32 const ssz = bend - p2;
        u32 const lastByte = sq[ssz - 1];
        if (lastByte == 0) break;
        int bitPos = (int)(8 * (ssz - 1) + kx_hb32(lastByte));     // unread bits
        bool bad = false;
        // The reader keeps 128 bits of the stream in registers: lo = word k (bits [64 k, 64 k + 64)), hi = word k + 1, and the
        // two words below them already requested (q0, q1), so a field never waits for the stream.  Fields come in two
        // groups per sequence, each at most 64 bits: ENSURE moves the window down until the group lies inside it.
        int k = ((bitPos - 1) >> 6) - 1;
        u64 hi = kxp_word(sq, ssz, k + 1), lo = kxp_word(sq, ssz, k), q0 = kxp_word(sq, ssz, k - 1), q1 = kxp_word(sq, ssz, k - 2);
#define KXP_DOWN() { hi = lo; lo = q0; q0 = q1; k--; q1 = kxp_word(sq, ssz, k - 2); }
#define KXP_ENSURE(m_) { if (bitPos - (int)(m_) < 64 * k) KXP_DOWN() if (bitPos - (int)(m_) < 64 * k) KXP_DOWN() }
// the n (<= 32) bits below bit position P_ (P_ - n >= 64 k, P_ <= 64 k + 128); n = 0 gives 0
#define KXP_BITS(P_, n_) kxp_bits(hi, lo, (int)(P_) - (int)(n_) - 64 * k, (n_))
        u32 sLL, sOF, sML;
        {
            u32 const need0 = tlLL + tlOF + tlML;                  // initial states, stream order LL, OF, ML
            if (bitPos < (int)need0) break;
            KXP_ENSURE(need0)
            sLL = KXP_BITS(bitPos, tlLL); sOF = KXP_BITS(bitPos - (int)tlLL, tlOF); sML = KXP_BITS(bitPos - (int)(tlLL + tlOF), tlML);
            bitPos -= (int)need0;
        }
        u32* const out = stage + (size_t)nstaged * 3u;
        u32 eL = tables[KXD_LL0 + sLL], eO = tables[KXD_OF0 + sOF], eM = tables[KXD_ML0 + sML];
        for (u32 i = 0; i < nbSeq; i++) {
            u32 const cL = eL >> 24, cO = eO >> 24, cM = eM >> 24;
            if (cL > 35 || cM > 52 || cO > 31) { bad = true; break; }
            u32 const xL = llx[cL], xM = mlx[cM];
            u32 const aL = xL >> 24, aM = xM >> 24, aO = cO;
            bool const upd = i + 1 < nbSeq;                        // the block's final sequence updates no state
            u32 const nL = upd ? (eL >> 16) & 0xFFu : 0u, nM = upd ? (eM >> 16) & 0xFFu : 0u, nO = upd ? (eO >> 16) & 0xFFu : 0u;
            u32 const needA = aO + aM + aL, needB = nL + nM + nO;
            if (bitPos < (int)(needA + needB)) { bad = true; break; }
            // bit order inside a sequence: OF extra, ML extra, LL extra, then LL state, ML state, OF state
            KXP_ENSURE(needA)
            u32 const xo = KXP_BITS(bitPos, aO), xm = KXP_BITS(bitPos - (int)aO, aM), xl = KXP_BITS(bitPos - (int)(aO + aM), aL);
            bitPos -= (int)needA;
            KXP_ENSURE(needB)
            u32 const yL = KXP_BITS(bitPos, nL), yM = KXP_BITS(bitPos - (int)nL, nM), yO = KXP_BITS(bitPos - (int)(nL + nM), nO);
            bitPos -= (int)needB;
            // the next states' table words are requested before this sequence is finished
            sLL = ((eL & 0xFFFFu) + yL) & 511u; sML = ((eM & 0xFFFFu) + yM) & 511u; sOF = ((eO & 0xFFFFu) + yO) & 255u;
            u32 const nLe = tables[KXD_LL0 + sLL], nOe = tables[KXD_OF0 + sOF], nMe = tables[KXD_ML0 + sML];
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
            eL = nLe; eO = nOe; eM = nMe;
        }
#undef KXP_BITS
#undef KXP_ENSURE
#undef KXP_DOWN
        if (bad || bitPos != 0) break;          // irregular: this block and the rest are left to k_zstd_decode
        rec.ok = 1; rec.rep[0] = rep1; rec.rep[1] = rep2; rec.rep[2] = rep3;
        blk[nb++] = rec;
        nstaged += nbSeq;
        pos += bsize;
    }
    a.nblk[f] = nb;
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
