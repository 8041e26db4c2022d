// zstd_predecode.h -- the sequence bitstreams of a batch of frames, decoded one LANE per frame ahead of k_zstd_decode.
//
// In k_zstd_decode (one wave per frame) the three tANS state chains of a block's sequences section are single-lane
// work: 86 % of that kernel's instructions served three lanes of a wave, and instruction issue
// was what bound it.  A sequence costs the same instructions whether one lane or sixty-four execute them, so this
// kernel turns the work sideways: a lane owns a whole frame, walks its blocks, builds the three FSE decoding tables of
// each compressed block in its share of the LDS (2.5 KiB: one 16-bit word per state) and decodes the block's sequences
// -- literal length, match length and the offset with the repeat-offset rules already applied -- into a staging area
// in HBM, 8 bytes per sequence.  Sixteen frames share a wave (LDS capacity sets how many frames a CU has in flight;
// tables in HBM instead -- tried, profiles/ -- lift that limit and put 1 to 2 us of memory latency into every state
// transition).  k_zstd_decode then loads 64 finished sequences per step instead of decoding them and keeps everything
// else (headers, literals, execution, every check and error code).
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
    u64* stage; u32 seq_cap;            // per entry: seq_cap sequences: litLength | (matchLength - 3) << 16 | offset << 32
    KPreBlk* blk; u32 blk_cap;          // per entry: blk_cap records (compressed blocks in frame order)
    u32* nblk;                          // per entry: records written
    const u32* perm;                    // which entry each lane slot takes (k_zstd_seq_perm), or null: slot i takes entry i
    u32 rep[3] = { 1, 4, 8 };           // the repeat offsets a frame starts with (a formatted dictionary brings its own)
};

// Per frame in LDS: one 16-bit word per FSE state -- the state's rank among its symbol's states, as "next" = count +
// rank (10 bits; FSE_buildDTable's symbolNext), and the symbol (6 bits): nbBits = tableLog - highbit(next) and
// newStateBase = (next << nbBits) - tableSize follow from it -- LL [0,512) ML [512,1024) OF [1024,1280), then the
// normalised counts and the per-symbol cursor of the table under construction.
#ifndef KXP_FRAMES
#define KXP_FRAMES 16
#endif
// ... and the two queues that keep HBM out of the sequence loop (see zstd_seq_predecode_body): 64 words of the
// bitstream, eight finished sequences.
#define KXP_RING 64
struct KPreFrameLds { u16 tb[1280]; short norm[56]; u16 symnext[56]; alignas(16) u32 ring[KXP_RING + 4]; alignas(16) u64 outq[8]; };      // (ring[64], ring[65] repeat ring[0], ring[1]: three words in a row never wrap)
struct KPreLds { KPreFrameLds f[KXP_FRAMES]; u32 llx[36]; u32 mlx[53]; };       // (llx, mlx: per code baseValue | extraBits << 24)

// FSE table description -> the frame's table (same construction as kfse_build_dtable)
KX_DEV void kxp_build_dtable(u16* tb, const short* norm, u32 maxSymbolValue, u32 tableLog, u16* symnext)
{
    u32 const tableSize = 1u << tableLog, mask = tableSize - 1;
    u32 const step = (tableSize >> 1) + (tableSize >> 3) + 3;
    u32 high = tableSize - 1;
    for (u32 s = 0; s <= maxSymbolValue; s++) {
        if (norm[s] == -1) { tb[high--] = (u16)s; symnext[s] = 1; }
        else symnext[s] = (u16)norm[s];
    }
    u32 pos = 0;
    for (u32 s = 0; s <= maxSymbolValue; s++) {
        for (int i = 0; i < norm[s]; i++) {
            tb[pos] = (u16)s;
            pos = (pos + step) & mask;
            while (pos > high) pos = (pos + step) & mask;
        }
    }
    for (u32 u = 0; u < tableSize; u++) {
        u32 const s = tb[u]; u32 const next = symnext[s]++;
        tb[u] = (u16)(next | (s << 10));
    }
}

// One symbol type's table for the block (a "repeat" mode keeps what is there).  kind[t]: 0 none yet, 1 RLE, 2 FSE.
// Returns bytes consumed or KXD_FAIL.
KX_DEV u32 kxp_seq_table(KPreFrameLds& fl, int t, u32 mode, const u8* p, u32 size, u32* tableLog, u32* kind, u32* klog)
{
    static const short LL_defaultNorm[36] = { 4,3,2,2,2,2,2,2, 2,2,2,2,2,1,1,1, 2,2,2,2,2,2,2,2, 2,3,2,1,1,1,1,1, -1,-1,-1,-1 };
    static const short ML_defaultNorm[53] = { 1,4,3,2,2,2,2,2, 2,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1,
                                              1,1,1,1,1,1,1,1, 1,1,1,1,1,1,-1,-1, -1,-1,-1,-1,-1 };
    static const short OF_defaultNorm[29] = { 1,1,1,1,1,1,2,2, 2,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1, -1,-1,-1,-1,-1 };
    u32 const maxSym = (t == 0) ? 35 : (t == 1) ? 31 : 52;
    u32 const maxLog = (t == 0) ? 9 : (t == 1) ? 8 : 9;
    u16* const tb = fl.tb + kxd_seq_base(t);
    u32 used = 0;
    if (mode == 0) {
        const short* dn = (t == 0) ? LL_defaultNorm : (t == 1) ? OF_defaultNorm : ML_defaultNorm;
        u32 const dmax = (t == 0) ? 35 : (t == 1) ? 28 : 52; u32 const dlog = (t == 1) ? 5 : 6;
        for (u32 s = 0; s <= dmax; s++) fl.norm[s] = dn[s];
        kxp_build_dtable(tb, fl.norm, dmax, dlog, fl.symnext);
        kind[t] = 2; klog[t] = dlog;
    } else if (mode == 1) {
        if (size < 1 || p[0] > maxSym) return KXD_FAIL;
        tb[0] = (u16)(1u | ((u32)p[0] << 10));               // tableLog 0: nbBits 0, next state 0
        kind[t] = 1; klog[t] = 0;
        used = 1;
    } else if (mode == 2) {
        u32 maxSV = maxSym, tl = 0;
        u32 const h = kfse_read_ncount(fl.norm, &maxSV, &tl, p, size, maxLog);
        if (h == 0) { kind[t] = 0; return KXD_FAIL; }
        kxp_build_dtable(tb, fl.norm, maxSV, tl, fl.symnext);
        kind[t] = 2; klog[t] = tl;
        used = h;
    } else if (kind[t] == 0) return KXD_FAIL;            // repeat without a previous table
    *tableLog = klog[t];
    return used;
}

// ---- which frames share a wave ----------------------------------------------------------------------------------
// The sixteen lanes of a wave run until the longest of their frames is done: with frames taken in batch order a wave's
// longest frame has 1.7 times the average number of sequences (the bench corpus), and that factor is the kernel's
// time.  Three small kernels order the lane slots by sequence count (a counting sort over count / 64, most first, so
// the waves that start last are the short ones): count (a thread per entry reads the first compressed block's sequence
// count), rank (bucket sizes to bucket starts), perm (an entry takes the next slot of its bucket; the order inside a
// bucket is whatever the atomics give, which changes nothing but who shares a wave).
#define KXP_SORT_BUCKETS 256
struct KSeqSortArgs { const u8* src; const u64* in_off; const u32* in_len; u32 n_slices; u32* key; u32* hist; u32* perm;
                      u32 len_shift; };   // 0: key = sequences of the first block / 64; else key = entry bytes >> len_shift (inflate: symbols go with the compressed size)

// sequences of the entry's first compressed block; 0 when there is none or anything is irregular (the pre-decoder will see that itself)
KX_DEV u32 kxp_first_nbseq(const u8* src, u32 srcSize)
{
    if (srcSize < 9 || kx_ld32(src) != 0xFD2FB528u) return 0;
    u32 const fhd = src[4]; u32 const dictId = fhd & 3, single = (fhd >> 5) & 1, fcsId = fhd >> 6;
    u32 const didSize = dictId == 3 ? 4 : dictId;
    u32 const fcsSize = fcsId == 0 ? single : (fcsId == 1 ? 2 : fcsId == 2 ? 4 : 8);
    u32 pos = 5 + (single ? 0u : 1u) + didSize + fcsSize;
    for (int guard = 0; guard < 4; guard++) {                   // raw / RLE blocks in front of it: a few at most are followed
        if (pos + 3 > srcSize) return 0;
        u32 const bh = (u32)src[pos] | ((u32)src[pos + 1] << 8) | ((u32)src[pos + 2] << 16);
        u32 const btype = (bh >> 1) & 3, bsize = bh >> 3;
        pos += 3;
        if (btype == 2) {
            if (pos + bsize > srcSize || bsize < 5) return 0;
            const u8* const bp = src + pos;
            u32 const lh0 = bp[0]; u32 const ltype = lh0 & 3, sf = (lh0 >> 2) & 3; u32 lpos;
            if (ltype < 2) {
                u32 lhSize, regen;
                if (sf == 0 || sf == 2) { lhSize = 1; regen = lh0 >> 3; }
                else if (sf == 1) { lhSize = 2; regen = kx_ld16(bp) >> 4; }
                else { lhSize = 3; regen = ((u32)bp[0] | ((u32)bp[1] << 8) | ((u32)bp[2] << 16)) >> 4; }
                lpos = lhSize + (ltype == 0 ? regen : 1u);
            } else {
                u32 const w = kx_ld32(bp); u32 lhSize, comp;
                if (sf < 2) { lhSize = 3; comp = (w >> 14) & 0x3FF; }
                else if (sf == 2) { lhSize = 4; comp = w >> 18; }
                else { lhSize = 5; comp = (w >> 22) + ((u32)bp[4] << 10); }
                lpos = lhSize + comp;
            }
            if (lpos + 3 > bsize) return 0;
            u32 const b0 = bp[lpos];
            if (b0 < 128) return b0;
            if (b0 < 255) return ((b0 - 128) << 8) + bp[lpos + 1];
            return kx_ld16(bp + lpos + 1) + 0x7F00;
        }
        if (btype == 3 || (bh & 1)) return 0;
        pos += btype == 0 ? bsize : 1u;
    }
    return 0;
}

KX_DEV void zstd_seq_count_body(const KSeqSortArgs& a)          // 256 threads per workgroup, a thread per entry
{
    u32 const f = kx_block() * 256u + (u32)kx_wave() * 64u + (u32)kx_lane();
    if (f >= a.n_slices) return;
    u32 k = a.len_shift ? (a.in_len[f] >> a.len_shift) : (kxp_first_nbseq(a.src + a.in_off[f], a.in_len[f]) >> 6);
    if (k >= KXP_SORT_BUCKETS) k = KXP_SORT_BUCKETS - 1;
    a.key[f] = k;
    kx_atomic_add(a.hist + k, 1u);
}

KX_DEV void zstd_seq_rank_body(const KSeqSortArgs& a)           // one workgroup of 256 threads: thread k owns bucket k
{
    u32 const k = (u32)kx_wave() * 64u + (u32)kx_lane();
    u32 before = 0;                                             // entries in the buckets that come first (the larger counts)
    for (u32 i = k + 1; i < KXP_SORT_BUCKETS; i++) before += a.hist[i];
    kx_block_sync();
    a.hist[k] = before;
}

KX_DEV void zstd_seq_perm_body(const KSeqSortArgs& a)
{
    u32 const f = kx_block() * 256u + (u32)kx_wave() * 64u + (u32)kx_lane();
    if (f >= a.n_slices) return;
    a.perm[kx_atomic_add(a.hist + a.key[f], 1u)] = f;
}

KX_DEV void zstd_seq_predecode_body(const KPreArgs& a)
{
    KX_SHARED KPreLds lds;
    {
        int const lane = kx_lane();
        for (int i = lane; i < 36; i += 4 * KXP_FRAMES) lds.llx[i] = kx_ll_base((u32)i) | (kxd_ll_bits((u32)i) << 24);
        for (int i = lane; i < 53; i += 4 * KXP_FRAMES) lds.mlx[i] = kx_ml_base((u32)i) | (kxd_ml_bits((u32)i) << 24);
    }
    kx_sync();
    if ((kx_lane() >> 2) >= KXP_FRAMES) return;          // (a launch wider than 4 lanes a frame: the emulator's 64)
    // Four lanes per frame.  All four walk the frame's blocks (same loads, same decisions); lane 0 of the quad builds the
    // tables and writes the records; in the sequence loop lane 0 is the offset's tANS chain, lane 1 the match length's,
    // lane 2 the literal length's (lane 3 shadows lane 2): a table look-up, an extra-bits field and a state update each,
    // the bit counts and values exchanged inside the quad (DPP), the bitstream container the same three LDS words for all.
    u32 const role = (u32)kx_lane() & 3u;
    u32 const slot = kx_block() * (u32)KXP_FRAMES + ((u32)kx_lane() >> 2);
    if (slot >= a.n_slices) return;
    u32 const f = a.perm ? a.perm[slot] : slot;
    KPreFrameLds& fl = lds.f[kx_lane() >> 2];
    const u8* const src = a.src + a.in_off[f];
    u32 const srcSize = a.in_len[f];
    u64* const stage = a.stage + (size_t)f * a.seq_cap;
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
    u32 rep1 = a.rep[0], rep2 = a.rep[1], rep3 = a.rep[2];
    u32 kind[3] = { 0, 0, 0 }, klog[3] = { 0, 0, 0 };        // (lane 0 of the quad)
    bool last = false;
    u32 covered = 0;                      // set when the frame's last block has been taken: every compressed block has a record
    while (ok && !last && nb < a.blk_cap) {
        if (pos + 3 > srcSize) break;
        u32 const bh = (u32)src[pos] | ((u32)src[pos + 1] << 8) | ((u32)src[pos + 2] << 16);
        last = bh & 1; u32 const btype = (bh >> 1) & 3; u32 const bsize = bh >> 3;
        pos += 3;
        if (btype == 3) break;
        if (btype == 0) { if (pos + bsize > srcSize) break; pos += bsize; if (last) covered = 0x80000000u; continue; }
        if (btype == 1) { if (pos + 1 > srcSize) break; pos += 1; if (last) covered = 0x80000000u; continue; }
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
        if (nbSeq == 0) { rec.ok = 1; if (role == 0) blk[nb] = rec; nb++; pos += bsize; if (last) covered = 0x80000000u; continue; }
        if (p2 >= bend) break;
        u32 const modes = bp[p2++];
        if (modes & 3) break;
        if (nstaged + nbSeq > a.seq_cap) break;
        u32 tlLL = 0, tlOF = 0, tlML = 0; bool tok = true;
        for (int t = 0; t < 3 && tok; t++) {
            u32 const mode = (modes >> (6 - 2 * t)) & 3u;
            u32 r = 0, tl = 0;
            if (role == 0) r = kxp_seq_table(fl, t, mode, bp + p2, bend - p2, &tl, kind, klog);
            r = kx_quad_bcast<0>(r); tl = kx_quad_bcast<0>(tl);
            if (t == 0) tlLL = tl; else if (t == 1) tlOF = tl; else tlML = tl;
            if (r == KXD_FAIL) tok = false; else p2 += r;
        }
        if (!tok || p2 >= bend) break;
        // ---- the bitstream, read backwards from its last set bit ----
        // The reader has no state but `remaining`, the number of unread bits: every sequence builds a 64-bit container
        // (top bit = next unread bit) from the three 32-bit words of the stream that hold it -- one address, three LDS
        // reads, two funnel shifts -- and each role lane takes its extra-bits field and its state field out of it.
        // (More than 64 bits in one sequence -- offsets beyond 64 KiB with long lengths -- takes a second container.)
        //
        // No HBM access inside the loop waits for memory.  On gfx9 loads and stores share one counter, and a wait in a
        // loop with conditional accesses is a wait for all of them: with the stream read straight from HBM and the
        // sequence stored straight to it, every refill waited a full round trip (measured: 2 250 cycles a sequence).
        // So the stream comes through a ring in LDS, filled sixteen words (64 bytes, aligned to the stream's 16-word
        // grid; a quarter per lane) at a time, and sequences leave through a queue of eight: every eighth sequence the
        // batch requested eight sequences earlier is written to the ring, the next one is requested, and the eight
        // sequences go out as one 64-byte run.  The one wait there finds everything long done.  A stream that outruns
        // the ring (more than 64 bits a sequence for a while) fills it on the spot.
        const u8* const sq = bp + p2; u32 const ssz = bend - p2;
        u32 const lastByte = sq[ssz - 1];
        if (lastByte == 0) break;
        int remaining = (int)(8 * (ssz - 1) + kx_hb32(lastByte));      // unread bits; the stream must end at exactly 0
        int lowFetched;                                                 // words [lowFetched, top] of the stream are in the ring
        int reqM = -1;                                                  // batch (16 words from word 16 reqM) on its way, -1: none
        KxQuad bq; bq.x = bq.y = bq.z = bq.w = 0;                       // this lane's quarter of it
#define KXP_RING_PUT(m_, q_) { u32 const o_ = (u32)(16 * (m_)) & (KXP_RING - 1); *(KxQuad*)(fl.ring + o_ + 4u * role) = q_; \
            if (o_ == 0 && role == 0) { fl.ring[KXP_RING] = (q_).x; fl.ring[KXP_RING + 1] = (q_).y; } }
        {
            // the top batch holds the stream's last word (1 to 4 bytes of it exist): its words come in one by one (a
            // 64-byte load could leave the entry); the batch below it, whole, with them; the one below that is requested
            int const jt = (int)((ssz - 1) >> 2), mt = jt >> 4;
            u32 wt = 0;
            for (u32 k = 4u * (u32)jt; k < ssz; k++) wt |= (u32)sq[k] << (8u * (k - 4u * (u32)jt));
            u32 tw[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { int const wi = 16 * mt + 4 * (int)role + k; tw[k] = wi < jt ? kx_ld32(sq + 4 * wi) : (wi == jt ? wt : 0u); }
            KxQuad t0; t0.x = tw[0]; t0.y = tw[1]; t0.z = tw[2]; t0.w = tw[3];
            if (mt >= 1) bq = kx_ld128u(sq + 64 * (mt - 1) + 16 * (int)role);
            KXP_RING_PUT(mt, t0)
            lowFetched = 16 * mt;
            if (mt >= 1) { KXP_RING_PUT(mt - 1, bq) lowFetched = 16 * (mt - 1); }
            if (mt >= 2) { bq = kx_ld128u(sq + 64 * (mt - 2) + 16 * (int)role); reqM = mt - 2; }
            kx_quad_sync();
        }
        u32 bad = 0;
// the 64 bits below bit position pos_ of the stream; words below the stream's first read as zero (only its last sequences get there)
#define KXP_CONTAINER(C_, pos_) u64 C_; { \
            int const w_ = (pos_) >> 5; u32 const s_ = (u32)(pos_) & 31u; \
            if (w_ - 2 < lowFetched && lowFetched > 0) { \
                /* the ring ran dry: the batch on its way is taken now, further ones are loaded on the spot -- the only waits for HBM the loop has */ \
                if (reqM >= 0) { KXP_RING_PUT(reqM, bq) lowFetched = 16 * reqM; reqM = -1; } \
                while (w_ - 2 < lowFetched && lowFetched > 0) { int const m_ = (lowFetched >> 4) - 1; \
                    KxQuad const c_ = kx_ld128u(sq + 64 * m_ + 16 * (int)role); KXP_RING_PUT(m_, c_) lowFetched = 16 * m_; } \
                kx_quad_sync(); } \
            u32 x2_, x1_, x0_; \
            if (w_ >= 2) { const u32* const r_ = fl.ring + ((w_ - 2) & (KXP_RING - 1)); x0_ = r_[0]; x1_ = r_[1]; x2_ = r_[2]; } \
            else { x2_ = w_ >= 0 ? fl.ring[w_ & (KXP_RING - 1)] : 0u; x1_ = w_ >= 1 ? fl.ring[(w_ - 1) & (KXP_RING - 1)] : 0u; x0_ = 0u; } \
            C_ = ((u64)kx_alignbit(x2_, x1_, s_) << 32) | kx_alignbit(x1_, x0_, s_); }
// n (<= 31) bits of container C_ that start o_ (<= 63) bits below its top
#define KXP_FIELD(C_, o_, n_) ((((u32)(((C_) << (o_)) >> 32)) >> 1) >> (31u - (n_)))
        // this lane's chain: 0 offset, 1 match length, 2 (and 3) literal length
        u32 const tbase = role == 0 ? (u32)KXD_OF0 : role == 1 ? (u32)KXD_ML0 : (u32)KXD_LL0;
        u32 const tl = role == 0 ? tlOF : role == 1 ? tlML : tlLL;
        u32 const sz = 1u << tl, smask = role == 0 ? 255u : 511u, maxSym = role == 0 ? 31u : role == 1 ? 52u : 35u;
        const u32* const xt = role == 1 ? lds.mlx : lds.llx;
        u32 st;
        {
            KXP_CONTAINER(C0, remaining)                            // initial states, stream order LL, OF, ML
            u32 const o0 = role == 0 ? tlLL : role == 1 ? tlLL + tlOF : 0u;
            st = KXP_FIELD(C0, o0, tl);
            remaining -= (int)(tlLL + tlOF + tlML);
        }
        if (remaining < 0) break;
        u64* const out = stage + nstaged;
        u32 e = fl.tb[tbase + st];
        for (u32 i = 0; i < nbSeq; i++) {
            u32 const sym = e >> 10, k = e & 1023u;
            bool const upd = i + 1 < nbSeq;                        // the block's final sequence updates no state
            u32 const n = upd ? tl - kx_hb32(k) : 0u;
            u32 const symc = sym > maxSym ? maxSym : sym;
            bad |= (sym > maxSym) | (remaining < 0);
            u32 const x = xt[symc];
            u32 const ax = role == 0 ? symc : x >> 24, base = role == 0 ? (1u << symc) : (x & 0xFFFFFFu);
            u32 const aO = kx_quad_bcast<0>(ax), aM = kx_quad_bcast<1>(ax), aL = kx_quad_bcast<2>(ax);
            u32 const nO = kx_quad_bcast<0>(n), nM = kx_quad_bcast<1>(n), nL = kx_quad_bcast<2>(n);
            // bit order inside a sequence: OF extra, ML extra, LL extra, then LL state, ML state, OF state
            u32 const needA = aO + aM + aL, needB = nL + nM + nO;
            u32 const offA = role == 0 ? 0u : role == 1 ? aO : aO + aM;
            u32 const offB = role == 0 ? nL + nM : role == 1 ? nL : 0u;
            int const rem0 = remaining < 0 ? 0 : remaining;
            KXP_CONTAINER(C, rem0)
            u32 const xv = KXP_FIELD(C, offA, ax);
            u32 yv;
            if (needA + needB <= 64u) { u32 const o = needA + offB; yv = KXP_FIELD(C, o > 63u ? 63u : o, n); }
            else { int const rem2 = rem0 - (int)needA < 0 ? 0 : rem0 - (int)needA; KXP_CONTAINER(C2, rem2) yv = KXP_FIELD(C2, offB, n); }
            remaining -= (int)(needA + needB);
            // the next state's table word is requested before this sequence is finished
            st = (((k << n) - sz) + yv) & smask;
            u32 const en = fl.tb[tbase + st];
            u32 const val = base + xv;
            u32 const ofv = kx_quad_bcast<0>(val), ml = kx_quad_bcast<1>(val), ll = kx_quad_bcast<2>(val);
            // repeat-offset rules
            bool const isRep = ofv <= 3;
            u32 const idx = ofv - 1 + (ll == 0);
            u32 const rm1 = (rep1 - 1) ? rep1 - 1 : 1u;
            u32 const roff = idx == 0 ? rep1 : idx == 1 ? rep2 : idx == 2 ? rep3 : rm1;
            u32 const off = isRep ? roff : ofv - 3;
            bool const sh2 = !isRep || idx >= 2, sh1 = !isRep || idx >= 1;
            rep3 = sh2 ? rep2 : rep3; rep2 = sh1 ? rep1 : rep2; rep1 = off;
            bad |= (ll > 0xFFFFu) | (ml - 3u > 0xFFFFu);              // does not fit the staging word: the block is left to k_zstd_decode
            if (role == 0) fl.outq[i & 7u] = (u64)((ll & 0xFFFFu) | ((ml - 3u) << 16)) | ((u64)off << 32);
            e = en;
            if ((i & 7u) == 7u) {
                // the batch requested eight sequences ago goes into the ring (its slots held words long taken)
                if (reqM >= 0) { KXP_RING_PUT(reqM, bq) lowFetched = 16 * reqM; reqM = -1; }
                // the next one, if its slots are free: the ring holds words [lowFetched, lowFetched + 64), the batch takes the
                // slots of the top sixteen
                if (lowFetched >= 16 && (remaining >> 5) < lowFetched + (KXP_RING - 16)) {
                    reqM = (lowFetched >> 4) - 1;
                    bq = kx_ld128u(sq + 64 * reqM + 16 * (int)role);
                }
                // eight sequences leave, a quarter per lane
                kx_quad_sync();
                KxQuad const q = ((const KxQuad*)fl.outq)[role];
                kx_st128u((u8*)(out + (i - 7u)) + 16u * role, q);
            }
        }
#undef KXP_FIELD
#undef KXP_CONTAINER
#undef KXP_RING_PUT
        bad = kx_quad_bcast<0>(bad) | kx_quad_bcast<1>(bad) | kx_quad_bcast<2>(bad);
        if (!bad && role == 0) for (u32 k = nbSeq & ~7u; k < nbSeq; k++) out[k] = fl.outq[k & 7u];
        if (bad || remaining != 0) break;          // irregular: this block and the rest are left to k_zstd_decode
        rec.ok = 1; rec.rep[0] = rep1; rec.rep[1] = rep2; rec.rep[2] = rep3;
        if (role == 0) blk[nb] = rec;
        nb++;
        nstaged += nbSeq;
        pos += bsize;
        if (last) covered = 0x80000000u;
    }
    if (role == 0) a.nblk[f] = nb | covered;
}

// ---------------------------------------------------------------------------------------------------------------------
// k_zstd_lit_predecode -- the Huffman-coded literals of a batch of frames, one LANE per stream.
//
// In k_zstd_decode four lanes of a wave walk the four Huffman streams of a block while sixty sit idle.  Here a wave
// takes 16 frames, a round per block: lanes 0..15 parse their frame up to its next Huffman-coded literals section;
// four lanes per frame bring the tree description into LDS; lanes 0..15 turn it into the frame's decoding table; then
// lane 4 j + s decodes stream s of frame j into the entry's literal staging area in HBM, and lanes 0..15 record the
// block.  k_zstd_decode then reads the literals instead of decoding them.
//
// The table is two-level so that sixteen of them and the stream rings fit three times into a CU's LDS: codes of up to
// 9 bits (all of them when the table log is 9 or less) are looked up by the top 9 bits of the window; the longer codes
// are the symbols of the one or two smallest weights, and zstd's table puts those first: a window below T (where the
// 9-bit codes start) indexes that prefix of the full table directly, at most 256 symbols x 2 entries.  Exact for every
// valid tree.
//
// As in the sequence pre-decoder no HBM access inside the symbol loop waits for memory: a lane's stream comes through
// a ring of 32 words in LDS, sixteen words per batch, requested 32 symbols before they are written to the ring; the
// 32 symbols leave as two 16-byte stores at the same point.
//
// Like the sequence pre-decoder this is an accelerator: a block is recorded only if its streams decoded to the last
// bit; anything irregular ends the frame's pre-decoding and k_zstd_decode handles (and reports) the rest as before.
struct KLitArgs {
    const u8* src; const u64* in_off; const u32* in_len; u32 n_slices;
    u8* lits; u32 lit_cap;              // per entry: lit_cap bytes of literal staging (the literals of all blocks, back to back)
    KPreLit* rec; u32 blk_cap;          // per entry: blk_cap records (compressed blocks of the first frame, in order)
    u32* nrec;                          // per entry: records written
    const u32* perm;                    // which entry each frame slot takes (k_zstd_seq_perm: neighbours in size share a wave), or null
};

#define KXL_FRAMES 16
#define KXL_RING 32
#define KXL_TREE 144                    // a tree description is at most 1 + 127 bytes; eight bytes in front for the reader's 8-byte loads
struct KLitFrameLds {
    union { struct { u16 l1[512]; u16 lg[512]; } t; struct { u16 wb[64]; u8 wc[64]; u8 tsym[64]; } b; } u;     // the weights' FSE table lives where the decoding table goes
    union { alignas(16) u32 ring[4][KXL_RING]; u8 tree[8 + KXL_TREE]; } r;
    u8 weights[256]; short norm[64]; u16 symnext[64]; u32 rank[16];
    // this round's job for the four stream lanes
    u32 valid, tableLog, T, soff[4], ssz[4], cnt[4], dst, bad, treeOff, treeLen;
};
struct KLitLds { KLitFrameLds f[KXL_FRAMES]; u32 more; };

// weights[0..nw) -> the two-level table; false on an invalid weight set
KX_DEV bool khuf_fill_compact(KLitFrameLds& L, u32 nw, u32 tableLog)
{
    for (u32 i = 0; i < 16; i++) L.rank[i] = 0;
    for (u32 i = 0; i < nw; i++) L.rank[L.weights[i]]++;
    if (L.rank[1] < 2 || (L.rank[1] & 1)) return false;
    u32 const sh = tableLog > 9 ? tableLog - 9 : 0u;
    u32 next = 0, T = 0;
    for (u32 w = 1; w <= tableLog; w++) { u32 const cur = next; next += L.rank[w] << (w - 1); L.rank[w] = cur; if (w == sh) T = next; }
    for (u32 s = 0; s < nw; s++) {
        u32 const w = L.weights[s];
        if (!w) continue;
        u32 const len = 1u << (w - 1); u32 const start = L.rank[w]; u32 const e = s | ((tableLog + 1 - w) << 8);
        if (w <= sh) { for (u32 i = 0; i < len; i++) L.u.t.lg[start + i] = (u16)e; }
        else { u32 const n = len >> sh, o = start >> sh; for (u32 i = 0; i < n; i++) L.u.t.l1[o + i] = (u16)e; }
        L.rank[w] += len;
    }
    L.T = T;
    return true;
}

// one lane, one stream, through the lane's ring (see the kernel's header); returns false on corruption
KX_DEV bool khuf_decode_stream_ring(const KLitFrameLds& L, u32* ring, const u8* p, u32 size, u8* out, u32 count)
{
    if (size == 0) return false;
    u32 const lastByte = p[size - 1];
    if (lastByte == 0) return false;
    u32 const tableLog = L.tableLog, T = L.T;
    u32 const sh = 32u - tableLog, sh1 = tableLog > 9 ? tableLog - 9 : 0u;
    u32 const totalBits = 8 * (size - 1) + kx_hb32(lastByte);
    const u16* const tbl = L.u.t.l1;                                // (lg follows l1: one array of 1024)
    int low;                                                        // words [low, low + 32) of the stream are in the ring
    int reqM = -1;                                                  // batch (16 words from word 16 reqM = low - 16) on its way, -1: none
    KxQuad b0, b1, b2, b3;
    b0.x = b0.y = b0.z = b0.w = 0; b1 = b0; b2 = b0; b3 = b0;
#define KHR_PUT(m_, q0_, q1_, q2_, q3_) { u32* const r_ = ring + ((16 * (m_)) & (KXL_RING - 1)); \
        *(KxQuad*)r_ = q0_; *(KxQuad*)(r_ + 4) = q1_; *(KxQuad*)(r_ + 8) = q2_; *(KxQuad*)(r_ + 12) = q3_; }
#define KHR_LOAD(m_, q0_, q1_, q2_, q3_) { const u8* const g_ = p + 64 * (m_); q0_ = kx_ld128u(g_); q1_ = kx_ld128u(g_ + 16); q2_ = kx_ld128u(g_ + 32); q3_ = kx_ld128u(g_ + 48); }
    int j = (int)((size - 1) >> 2);                                // the top word (1 to 4 bytes of it exist)
    u32 wt = 0;
    for (u32 k = 4u * (u32)j; k < size; k++) wt |= (u32)p[k] << (8u * (k - 4u * (u32)j));
    u32 const vb = totalBits - 32u * (u32)j;                       // its bits below the end mark: 0..31
    u32 hi = vb ? wt << (32u - vb) : 0u, lo = 0, avail = vb;
    {
        // the top batch's words below the top word come in one by one (a 64-byte load could leave the entry); the batch
        // below it, whole, with them; the one below that is requested
        int const mt = j >> 4;
        u32 tw[16];
#pragma unroll
        for (int k = 0; k < 16; k++) { int const wi = 16 * mt + k; tw[k] = wi < j ? kx_ld32(p + 4 * wi) : 0u; }
        KxQuad t0, t1, t2, t3;
        t0.x = tw[0]; t0.y = tw[1]; t0.z = tw[2]; t0.w = tw[3]; t1.x = tw[4]; t1.y = tw[5]; t1.z = tw[6]; t1.w = tw[7];
        t2.x = tw[8]; t2.y = tw[9]; t2.z = tw[10]; t2.w = tw[11]; t3.x = tw[12]; t3.y = tw[13]; t3.z = tw[14]; t3.w = tw[15];
        if (mt >= 1) KHR_LOAD(mt - 1, b0, b1, b2, b3)
        KHR_PUT(mt, t0, t1, t2, t3)
        low = 16 * mt;
        if (mt >= 1) { KHR_PUT(mt - 1, b0, b1, b2, b3) low = 16 * (mt - 1); }
        if (mt >= 2) { KHR_LOAD(mt - 2, b0, b1, b2, b3) reqM = mt - 2; }
    }
    j--;                                                            // the next word to take ...
    u32 w1;                                                         // ... read ahead of its use
// (the ring ran dry -- it cannot at 11 bits a symbol, the code is there for the reader's sake: the batch on its way is
// taken now, further ones are loaded on the spot)
#define KHR_FETCH { if (j >= 0 && j < low) { \
                        if (reqM >= 0) { KHR_PUT(reqM, b0, b1, b2, b3) low = 16 * reqM; reqM = -1; } \
                        while (j < low) { int const m_ = (low >> 4) - 1; KxQuad c0_, c1_, c2_, c3_; KHR_LOAD(m_, c0_, c1_, c2_, c3_) KHR_PUT(m_, c0_, c1_, c2_, c3_) low = 16 * m_; } } \
                    w1 = j < 0 ? 0u : ring[j & (KXL_RING - 1)]; }
    KHR_FETCH
    u32 consumed = 0;
#define KHR_REFILL if (avail <= 32u) { u64 const c_ = (((u64)hi << 32) | lo) | ((u64)w1 << (32u - avail)); hi = (u32)(c_ >> 32); lo = (u32)c_; \
                                       avail += 32u; j--; KHR_FETCH }
#define KHR_SYM(acc_, k_) { u32 const ix_ = hi >> sh; u32 const e_ = tbl[ix_ < T ? 512u + ix_ : ix_ >> sh1]; u32 const nb_ = e_ >> 8; \
                            hi = kx_alignbit(hi, lo, 32u - nb_); lo <<= nb_; avail -= nb_; consumed += nb_; acc_ |= (e_ & 0xFFu) << (8 * (k_)); }
#define KHR_FOUR(acc_) { acc_ = 0; KHR_REFILL KHR_SYM(acc_, 0) KHR_SYM(acc_, 1) KHR_REFILL KHR_SYM(acc_, 2) KHR_SYM(acc_, 3) }
    u32 i = 0;
    for (; i + 32 <= count; i += 32) {
        KxQuad qa, qb;
        KHR_FOUR(qa.x) KHR_FOUR(qa.y) KHR_FOUR(qa.z) KHR_FOUR(qa.w) KHR_FOUR(qb.x) KHR_FOUR(qb.y) KHR_FOUR(qb.z) KHR_FOUR(qb.w)
        // the batch requested 32 symbols ago goes into the ring once the words in its slots (the ring's top half) are taken;
        // then the next one is requested; the 32 symbols leave
        if (reqM >= 0 && j < low + 16) { KHR_PUT(reqM, b0, b1, b2, b3) low = 16 * reqM; reqM = -1; }
        if (reqM < 0 && low >= 16) { reqM = (low >> 4) - 1; KHR_LOAD(reqM, b0, b1, b2, b3) }
        kx_st128u(out + i, qa); kx_st128u(out + i + 16, qb);
    }
    for (; i < count; i++) { u32 acc = 0; KHR_REFILL KHR_SYM(acc, 0) out[i] = (u8)acc; }
#undef KHR_FOUR
#undef KHR_SYM
#undef KHR_REFILL
#undef KHR_FETCH
#undef KHR_LOAD
#undef KHR_PUT
    return consumed == totalBits;
}

KX_DEV void zstd_lit_predecode_body(const KLitArgs& a)
{
    KX_SHARED KLitLds lds;
    int const tid = kx_lane();                                  // 64 threads
    for (u32 base = kx_block() * KXL_FRAMES; base < a.n_slices; base += kx_nblocks() * KXL_FRAMES) {
        // ---- the parsing state of lane tid < 16: frame base + tid ----
        bool const owner = tid < KXL_FRAMES && base + (u32)tid < a.n_slices;
        u32 const fa = !owner ? 0u : a.perm ? a.perm[base + (u32)tid] : base + (u32)tid;
        const u8* srcA = a.src; u32 srcSize = 0, pos = 0, nb = 0, litUsed = 0; bool go = false, last = false;
        u32 hufLog = 0; bool hufValid = false;
        u32 covered = 0;                  // set when the frame's last block has been taken: every compressed block has a record
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
        // ---- the decoding identity of every lane: stream sB of frame base + jB ----
        int const jB = tid >> 2, sB = tid & 3;
        bool const haveB = base + (u32)jB < a.n_slices;
        u32 const fb = !haveB ? 0u : a.perm ? a.perm[base + (u32)jB] : base + (u32)jB;
        const u8* const srcB = haveB ? a.src + a.in_off[fb] : a.src;
        u32 const srcSizeB = haveB ? a.in_len[fb] : 0u;
        for (;;) {
            if (tid == 0) lds.more = 0;
            if (tid < KXL_FRAMES) { lds.f[tid].valid = 0; lds.f[tid].bad = 0; lds.f[tid].treeLen = 0; }
            kx_sync();
            // ---- the owner: up to the frame's next Huffman-coded literals section ----
            u32 regenA = 0, lhSizeA = 0, compA = 0, nstreamsA = 0, ltypeA = 0, bposA = 0; bool found = false;
            if (owner && go) {
                KLitFrameLds& L = lds.f[tid];
                while (go && !last && !found) {
                    if (nb >= a.blk_cap || pos + 3 > srcSize) { go = false; break; }
                    u32 const bh = (u32)srcA[pos] | ((u32)srcA[pos + 1] << 8) | ((u32)srcA[pos + 2] << 16);
                    last = bh & 1; u32 const btype = (bh >> 1) & 3; u32 const bsize = bh >> 3;
                    pos += 3;
                    if (btype == 3) { go = false; break; }
                    if (btype == 0) { if (pos + bsize > srcSize) { go = false; break; } pos += bsize; if (last) covered = 0x80000000u; continue; }
                    if (btype == 1) { if (pos + 1 > srcSize) { go = false; break; } pos += 1; if (last) covered = 0x80000000u; continue; }
                    if (pos + bsize > srcSize || bsize > 128u * 1024u || bsize < 2) { go = false; break; }
                    const u8* const bp = srcA + pos; u32 const bend = bsize;
                    u32 const lh0 = bp[0]; u32 const ltype = lh0 & 3, sf = (lh0 >> 2) & 3;
                    KPreLit r; r.off = litUsed; r.regen = 0; r.ok = 0; r.pad = 0;
                    if (ltype < 2) { a.rec[(size_t)fa * a.blk_cap + nb++] = r; pos += bsize; if (last) covered = 0x80000000u; continue; }      // raw / RLE literals: nothing to decode
                    if (bend < 5) { go = false; break; }
                    u32 const w = kx_ld32(bp); u32 lhSize, regen, comp, nstreams;
                    if (sf < 2) { lhSize = 3; regen = (w >> 4) & 0x3FF; comp = (w >> 14) & 0x3FF; nstreams = sf ? 4 : 1; }
                    else if (sf == 2) { lhSize = 4; regen = (w >> 4) & 0x3FFF; comp = w >> 18; nstreams = 4; }
                    else { lhSize = 5; regen = (w >> 4) & 0x3FFFF; comp = (w >> 22) + ((u32)bp[4] << 10); nstreams = 4; }
                    if (regen > 128u * 1024u || lhSize + comp > bend || comp == 0 || litUsed + regen > a.lit_cap) { go = false; break; }
                    if (ltype == 3 && !hufValid) { go = false; break; }
                    // (a tree-less block: the table of the previous Huffman block is still in this frame's slice of the LDS)
                    if (ltype == 2) { L.treeOff = pos + lhSize; L.treeLen = comp < (u32)KXL_TREE ? comp : (u32)KXL_TREE; }
                    regenA = regen; lhSizeA = lhSize; compA = comp; nstreamsA = nstreams; ltypeA = ltype; bposA = pos;
                    found = true;
                    pos += bsize;
                }
                if (found) lds.more = 1;
            }
            kx_sync();
            if (lds.more == 0) break;
            // ---- four lanes per frame: the tree description into LDS ----
            {
                KLitFrameLds& L = lds.f[jB];
                u32 const tl_ = L.treeLen, to_ = L.treeOff;
                for (u32 k = (u32)sB * 4u; k < tl_; k += 16u) {
                    if (k + 4 <= tl_) { u32 const v = kx_ld32(srcB + to_ + k); __builtin_memcpy(L.r.tree + 8 + k, &v, 4); }
                    else for (u32 q = k; q < tl_; q++) L.r.tree[8 + q] = srcB[to_ + q];
                }
            }
            kx_sync();
            // ---- the owner: the table, the streams ----
            if (owner && found) {
                KLitFrameLds& L = lds.f[tid];
                u32 hused = 0; bool okA = true;
                if (ltypeA == 2) {
                    u32 tl = 0, nw = 0;
                    hused = khuf_read_dtable<KLitFrameLds, false>(L, L.r.tree + 8, L.treeLen, &tl, &nw);
                    if (hused == 0 || !khuf_fill_compact(L, nw, tl)) okA = false;
                    else { hufLog = tl; hufValid = true; }
                }
                u32 const so = bposA + lhSizeA + hused, ssize = compA - hused;
                if (okA && hused >= compA) okA = false;
                if (okA) {
                    if (nstreamsA == 1) {
                        L.soff[0] = so; L.ssz[0] = ssize; L.cnt[0] = regenA;
                        for (int k = 1; k < 4; k++) { L.soff[k] = 0; L.ssz[k] = 0; L.cnt[k] = 0xFFFFFFFFu; }       // no such stream
                    } else if (ssize < 10) okA = false;
                    else {
                        u32 const c0 = kx_ld16(srcA + so), c1 = kx_ld16(srcA + so + 2), c2 = kx_ld16(srcA + so + 4);
                        u32 const seg = (regenA + 3) / 4;
                        if (6 + c0 + c1 + c2 > ssize || 3 * seg > regenA) okA = false;
                        else {
                            L.soff[0] = so + 6; L.soff[1] = so + 6 + c0; L.soff[2] = so + 6 + c0 + c1; L.soff[3] = so + 6 + c0 + c1 + c2;
                            L.ssz[0] = c0; L.ssz[1] = c1; L.ssz[2] = c2; L.ssz[3] = ssize - 6 - c0 - c1 - c2;
                            L.cnt[0] = seg; L.cnt[1] = seg; L.cnt[2] = seg; L.cnt[3] = regenA - 3 * seg;
                        }
                    }
                }
                if (okA) { L.tableLog = hufLog; L.dst = litUsed; L.valid = 1; }
                else { go = false; found = false; }
            }
            kx_sync();
            // ---- one lane per stream ----
            if (haveB && lds.f[jB].valid) {
                KLitFrameLds& L = lds.f[jB];
                u32 const cnt = L.cnt[sB];
                if (cnt != 0xFFFFFFFFu) {
                    u32 const seg = L.cnt[0];                                   // stream s writes behind s full segments
                    u8* const out = a.lits + (size_t)fb * a.lit_cap + L.dst + (L.cnt[1] == 0xFFFFFFFFu ? 0u : (u32)sB * seg);
                    bool const ok = khuf_decode_stream_ring(L, L.r.ring[sB], srcB + L.soff[sB], L.ssz[sB], out, cnt);
                    if (!ok) L.bad = 1;
                }
            }
            (void)srcSizeB;
            kx_sync();
            // ---- the owner records the block ----
            if (owner && found) {
                KPreLit r; r.off = litUsed; r.regen = regenA; r.ok = lds.f[tid].bad ? 0u : 1u; r.pad = 0;
                a.rec[(size_t)fa * a.blk_cap + nb++] = r;
                if (lds.f[tid].bad) go = false; else { litUsed += regenA; if (last) covered = 0x80000000u; }
            }
            kx_sync();
        }
        if (owner) a.nrec[fa] = nb | covered;
        kx_sync();
    }
}
