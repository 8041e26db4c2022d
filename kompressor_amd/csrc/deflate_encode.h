// deflate_encode.h -- the block writer of zlib level 6 (trees.c): per block of at
// most 16383 symbols the literal/length, distance and code-length Huffman trees
// (zlib's heap construction with depth tie-break, 15/7-bit length limiting),
// the stored / fixed / dynamic decision, the dynamic header, and the symbol bits.
//
// One wave per slice.  Histograms and symbol packing are wave-parallel (each lane
// packs one symbol per step into an LDS bit buffer at a scanned offset); the tree
// constructions (<= 573 nodes) and the dynamic header are lane 0's.
// Replaces zlib's _tr_flush_block behind the reference's
// ZlibCompressor(ZlibFormat.Raw, 6) (kompressor-zlib--nativelib/.../jni/Wrapper.cpp:73).
#pragma once
#include "deflate_match.h"
#include "zstd_entropy.h"      // kx_wave_copy, kx_lds_or users

#define KD_L_CODES 286
#define KD_D_CODES 30
#define KD_BL_CODES 19
#define KD_HEAP_SIZE 573

struct KdEncLds {
    u32 lfreq[288]; u32 dfreq[32]; u32 blfreq[20];
    u32 lcode[288]; u32 dcode[32]; u32 blcode[20];      // code | len << 16
    u16 tfreq[576]; u16 tdad[576]; u16 tlen[576]; u16 heap[576]; u8 depth[576];
    u16 bl_count[16];
    u16 llen[290]; u16 dlen[34];                        // code lengths (+ sentinel) for the header
    u32 cbuf[128];
    u32 bc[16];
};

KX_DEV u32 kd_bi_reverse(u32 code, int len) { u32 res = 0; do { res |= code & 1; code >>= 1; res <<= 1; } while (--len > 0); return res >> 1; }

// length symbol (lc = length - 3): code index 0..28, extra bits, extra value
KX_DEV void kd_len_code(u32 lc, u32& code, u32& extra, u32& xval)
{
    if (lc < 8) { code = lc; extra = 0; xval = 0; }
    else if (lc == 255) { code = 28; extra = 0; xval = 0; }
    else { u32 const k = kx_hb32(lc); extra = k - 2; code = 4 * (k - 1) + ((lc >> (k - 2)) & 3u); xval = lc & ((1u << extra) - 1u); }
}
// distance symbol (d = distance - 1): code 0..29
KX_DEV void kd_dist_code(u32 d, u32& code, u32& extra, u32& xval)
{
    if (d < 4) { code = d; extra = 0; xval = 0; }
    else { u32 const k = kx_hb32(d); extra = k - 1; code = 2 * k + ((d >> (k - 1)) & 1u); xval = d & ((1u << extra) - 1u); }
}
KX_DEV u32 kd_extra_lbits(u32 code) { return (code < 8 || code == 28) ? 0u : (code - 4) >> 2; }
KX_DEV u32 kd_extra_dbits(u32 code) { return code < 4 ? 0u : (code - 2) >> 1; }
KX_DEV u32 kd_static_llen(u32 n) { return n <= 143 ? 8u : n <= 255 ? 9u : n <= 279 ? 7u : 8u; }
KX_DEV u32 kd_static_lcode(u32 n)
{
    if (n <= 143) return kd_bi_reverse(0x30 + n, 8) | (8u << 16);
    if (n <= 255) return kd_bi_reverse(0x190 + (n - 144), 9) | (9u << 16);
    if (n <= 279) return kd_bi_reverse(n - 256, 7) | (7u << 16);
    return kd_bi_reverse(0xC0 + (n - 280), 8) | (8u << 16);
}

// ---- lane 0: zlib's build_tree over lds.tfreq[0..elems) -> lds.tlen, codes into `codes` ----
struct KdTreeCost { u32 opt_len, static_len; };

KX_DEV bool kd_smaller(const KdEncLds& l, int n, int m) { return l.tfreq[n] < l.tfreq[m] || (l.tfreq[n] == l.tfreq[m] && l.depth[n] <= l.depth[m]); }
KX_DEV void kd_pqdownheap(KdEncLds& l, int heap_len, int k)
{
    int const v = l.heap[k]; int j = k << 1;
    while (j <= heap_len) {
        if (j < heap_len && kd_smaller(l, l.heap[j + 1], l.heap[j])) j++;
        if (kd_smaller(l, v, l.heap[j])) break;
        l.heap[k] = l.heap[j]; k = j; j <<= 1;
    }
    l.heap[k] = (u16)v;
}
// kind: 0 = literal/length, 1 = distance, 2 = code lengths.  Returns max_code.
KX_DEV int kd_build_tree(KdEncLds& l, int kind, u32* codes, KdTreeCost& cost)
{
    int const elems = kind == 0 ? KD_L_CODES : kind == 1 ? KD_D_CODES : KD_BL_CODES;
    int const max_length = kind == 2 ? 7 : 15;
    int heap_len = 0, heap_max = KD_HEAP_SIZE, max_code = -1, n, m, node;
    for (n = 0; n < elems; n++) {
        if (l.tfreq[n] != 0) { l.heap[++heap_len] = (u16)(max_code = n); l.depth[n] = 0; } else l.tlen[n] = 0;
    }
    while (heap_len < 2) {
        node = l.heap[++heap_len] = (u16)(max_code < 2 ? ++max_code : 0);
        l.tfreq[node] = 1; l.depth[node] = 0; cost.opt_len--;
        if (kind == 0) cost.static_len -= kd_static_llen((u32)node); else if (kind == 1) cost.static_len -= 5;
    }
    for (n = heap_len / 2; n >= 1; n--) kd_pqdownheap(l, heap_len, n);
    node = elems;
    do {
        n = l.heap[1]; l.heap[1] = l.heap[heap_len--]; kd_pqdownheap(l, heap_len, 1);
        m = l.heap[1];
        l.heap[--heap_max] = (u16)n; l.heap[--heap_max] = (u16)m;
        l.tfreq[node] = (u16)(l.tfreq[n] + l.tfreq[m]);
        l.depth[node] = (u8)((l.depth[n] >= l.depth[m] ? l.depth[n] : l.depth[m]) + 1);
        l.tdad[n] = l.tdad[m] = (u16)node;
        l.heap[1] = (u16)node++;
        kd_pqdownheap(l, heap_len, 1);
    } while (heap_len >= 2);
    l.heap[--heap_max] = l.heap[1];
    // gen_bitlen
    int h, bits, overflow = 0;
    for (bits = 0; bits <= 15; bits++) l.bl_count[bits] = 0;
    l.tlen[l.heap[heap_max]] = 0;
    for (h = heap_max + 1; h < KD_HEAP_SIZE; h++) {
        n = l.heap[h]; bits = l.tlen[l.tdad[n]] + 1;
        if (bits > max_length) { bits = max_length; overflow++; }
        l.tlen[n] = (u16)bits;
        if (n > max_code) continue;
        l.bl_count[bits]++;
        u32 xbits = 0;
        if (kind == 0) { if (n >= 257) xbits = kd_extra_lbits((u32)n - 257); }
        else if (kind == 1) xbits = kd_extra_dbits((u32)n);
        else xbits = n == 16 ? 2u : n == 17 ? 3u : n == 18 ? 7u : 0u;
        u32 const f = l.tfreq[n];
        cost.opt_len += f * ((u32)bits + xbits);
        if (kind == 0) cost.static_len += f * (kd_static_llen((u32)n) + xbits); else if (kind == 1) cost.static_len += f * (5 + xbits);
    }
    if (overflow > 0) {
        do {
            bits = max_length - 1;
            while (l.bl_count[bits] == 0) bits--;
            l.bl_count[bits]--; l.bl_count[bits + 1] += 2; l.bl_count[max_length]--;
            overflow -= 2;
        } while (overflow > 0);
        for (bits = max_length; bits != 0; bits--) {
            n = l.bl_count[bits];
            while (n != 0) {
                m = l.heap[--h];
                if (m > max_code) continue;
                if ((int)l.tlen[m] != bits) { cost.opt_len += (u32)(bits - (int)l.tlen[m]) * l.tfreq[m]; l.tlen[m] = (u16)bits; }
                n--;
            }
        }
    }
    // gen_codes
    u32 next_code[16]; u32 code = 0;
    for (bits = 1; bits <= 15; bits++) { code = (code + l.bl_count[bits - 1]) << 1; next_code[bits] = code; }
    for (n = 0; n < elems; n++) {
        int const len = n <= max_code ? l.tlen[n] : 0;
        codes[n] = len ? (kd_bi_reverse(next_code[len]++, len) | ((u32)len << 16)) : 0u;
    }
    return max_code;
}

// run-length statistics of a code-length array (scan_tree)
KX_DEV void kd_scan_tree(KdEncLds& l, u16* lens, int max_code)
{
    int prevlen = -1, curlen, nextlen = lens[0], count = 0, max_count = 7, min_count = 4;
    if (nextlen == 0) { max_count = 138; min_count = 3; }
    lens[max_code + 1] = 0xFFFF;
    for (int n = 0; n <= max_code; n++) {
        curlen = nextlen; nextlen = lens[n + 1];
        if (++count < max_count && curlen == nextlen) continue;
        else if (count < min_count) l.blfreq[curlen] += (u32)count;
        else if (curlen != 0) { if (curlen != prevlen) l.blfreq[curlen]++; l.blfreq[16]++; }
        else if (count <= 10) l.blfreq[17]++;
        else l.blfreq[18]++;
        count = 0; prevlen = curlen;
        if (nextlen == 0) { max_count = 138; min_count = 3; }
        else if (curlen == nextlen) { max_count = 6; min_count = 3; }
        else { max_count = 7; min_count = 4; }
    }
}

// serial LSB-first bit writer that continues a stream (lane 0): `acc` holds the nb (< 32)
// pending bits of the 32-bit word number `word` of the stream
struct KdBitW { u8* base; u32 word; u64 acc; u32 nb; };
KX_DEV void kdw_init(KdBitW& w, u8* base, u32 bitpos, u32 pending) { w.base = base; w.word = bitpos >> 5; w.nb = bitpos & 31u; w.acc = pending; }
KX_DEV void kdw_put(KdBitW& w, u32 v, u32 n)
{
    if (!n) return;
    w.acc |= (u64)(v & ((n >= 32) ? 0xFFFFFFFFu : ((1u << n) - 1u))) << w.nb; w.nb += n;
    if (w.nb >= 32) { kx_st32(w.base + 4u * w.word, (u32)w.acc); w.word++; w.acc >>= 32; w.nb -= 32; }
}
KX_DEV u32 kdw_bitpos(const KdBitW& w) { return 32u * w.word + w.nb; }
KX_DEV void kd_send_tree(KdBitW& w, const KdEncLds& l, const u16* lens, int max_code)
{
    int prevlen = -1, curlen, nextlen = lens[0], count = 0, max_count = 7, min_count = 4;
    if (nextlen == 0) { max_count = 138; min_count = 3; }
#define KD_SENDBL(c_) kdw_put(w, l.blcode[c_] & 0xFFFFu, l.blcode[c_] >> 16)
    for (int n = 0; n <= max_code; n++) {
        curlen = nextlen; nextlen = lens[n + 1];
        if (++count < max_count && curlen == nextlen) continue;
        else if (count < min_count) { do { KD_SENDBL(curlen); } while (--count != 0); }
        else if (curlen != 0) {
            if (curlen != prevlen) { KD_SENDBL(curlen); count--; }
            KD_SENDBL(16); kdw_put(w, (u32)(count - 3), 2);
        } else if (count <= 10) { KD_SENDBL(17); kdw_put(w, (u32)(count - 3), 3); }
        else { KD_SENDBL(18); kdw_put(w, (u32)(count - 11), 7); }
        count = 0; prevlen = curlen;
        if (nextlen == 0) { max_count = 138; min_count = 3; }
        else if (curlen == nextlen) { max_count = 6; min_count = 3; }
        else { max_count = 7; min_count = 4; }
    }
#undef KD_SENDBL
}

KX_DEV void kd_cbuf_put(u32* cbuf, u32 pos, u64 v, u32 n)
{
    if (n == 0) return;
    u32 const sh = pos & 31u; u32 const wi = pos >> 5;
    u64 const lo = v << sh;
    kx_lds_or(&cbuf[wi], (u32)lo);
    if ((u32)(lo >> 32)) kx_lds_or(&cbuf[wi + 1], (u32)(lo >> 32));
    if (sh + n > 64) { u32 const hi = (u32)(v >> (64 - sh)); if (hi) kx_lds_or(&cbuf[wi + 2], hi); }
}

// Adler-32 of p[0..n), all lanes get the value
KX_DEV u32 kx_wave_adler32(const u8* p, u32 n, int lane)
{
    u32 const chunk = (n + 63) / 64; u32 const b0 = (u32)lane * chunk; u32 b1 = b0 + chunk; if (b1 > n) b1 = n;
    u32 s1 = 0, s2 = 0;
    for (u32 i = (b0 < n ? b0 : n); i < b1; i++) { s1 += p[i]; s2 += s1; if ((i & 2047u) == 2047u) { s1 %= 65521u; s2 %= 65521u; } }
    s1 %= 65521u; s2 %= 65521u;
    // A = 1 + sum a_i ; B = n + sum (b_i + a_i * bytes after chunk i)
    u32 const after = (b1 < n) ? n - b1 : 0u;
    u32 A = s1, B = (s2 + (u32)(((u64)s1 * (after % 65521u)) % 65521u)) % 65521u;
    for (int o = 1; o < 64; o <<= 1) { A = (A + kx_shfl(A, lane ^ o)) % 65521u; B = (B + kx_shfl(B, lane ^ o)) % 65521u; }
    A = (A + 1u) % 65521u; B = (B + n % 65521u) % 65521u;
    return (B << 16) | A;
}

// CRC-32 (gzip trailer, polynomial 0xEDB88320 reflected) of p[0..n), all lanes get the value.
// Every lane runs the byte-table recurrence over its own chunk (lane 0 takes the ragged first one),
// then the 64 chunk CRCs are folded pairwise: crc(A||B) = crc(A) * x^(8|B|) mod P  xor  crc(B),
// products of polynomials by shift-and-add (zlib's crc32_combine identity).  tab = 256 LDS words.
KX_DEV u32 kx_crc_mul(u32 a, u32 b)
{
    u32 p = 0;
    for (int i = 0; i < 32; i++) { if (a & 0x80000000u) p ^= b; a <<= 1; b = (b >> 1) ^ ((b & 1u) ? 0xEDB88320u : 0u); }
    return p;
}
KX_DEV u32 kx_wave_crc32(const u8* p, u32 n, u32* tab, int lane)
{
    kx_sync();
    for (int i = lane; i < 256; i += 64) { u32 c = (u32)i; for (int k = 0; k < 8; k++) c = (c >> 1) ^ ((c & 1u) ? 0xEDB88320u : 0u); tab[i] = c; }
    kx_sync();
    u32 const L = n / 64u, r = n - 63u * L;
    u32 const b0 = lane == 0 ? 0u : r + (u32)(lane - 1) * L, b1 = lane == 0 ? r : b0 + L;
    u32 c = 0xFFFFFFFFu; u32 i = b0;
    for (; i + 8 <= b1; i += 8) {
        u64 w = kx_ld64(p + i);
        for (int k = 0; k < 8; k++) { c = tab[(c ^ (u32)w) & 0xFFu] ^ (c >> 8); w >>= 8; }
    }
    for (; i < b1; i++) c = tab[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
    c ^= 0xFFFFFFFFu;
    u32 X = 0x80000000u, base = 0x00800000u;            // x^0, x^8
    for (u32 e = L; e; e >>= 1) { if (e & 1u) X = kx_crc_mul(base, X); base = kx_crc_mul(base, base); }
    for (int o = 1; o < 64; o <<= 1) {
        u32 const right = kx_shfl(c, lane + o);
        if ((lane & (2 * o - 1)) == 0) c = kx_crc_mul(X, c) ^ right;
        X = kx_crc_mul(X, X);
    }
    return kx_shfl(c, 0);
}

KX_DEV void deflate_encode_slice(const KdArgs& a, KdEncLds& lds, u32 slice, int lane)
{
    const u8* const src = a.src + a.in_off[slice];
    u8* const dst = a.dst + a.out_off[slice];
    const u32* const syms = a.syms + (size_t)slice * a.pos_cap;
    KdSliceMeta const mm = a.meta[slice];
    const KdBlockInfo* const blocks = a.blocks + (size_t)slice * a.blk_cap;
    // bits written so far; lds.cbuf[0] holds the pending partial word. The zlib wrapper's header for
    // level 6 / 32 KiB window is 78 9C (CMF 0x78, FLG: level flags 2 -- 1 below level 6: 5E, 3 above: DA --, check bits so that
    // CMF*256+FLG % 31 == 0); gzip (format 2): 1F 8B, CM 8, no flags, MTIME 0, XFL 0 (2 at level 9), OS 3 (what zlib writes on Linux)
    u32 bitpos = a.format == 1 ? 16u : (a.format == 2 ? 80u : 0u);
    for (int i = lane; i < 128; i += 64) lds.cbuf[i] = (i != 0) ? 0u : (a.format == 1 ? (a.zcmf | (a.zflg << 8)) : (a.format == 2 ? (0x0300u | a.gxfl) : 0u));
    if (a.format == 2 && lane == 0) { kx_st32(dst, 0x00088B1Fu); kx_st32(dst + 4, 0u); }
    kx_sync();
    u32 s0 = 0;
    for (u32 b = 0; b < mm.nblocks; b++) {
        KdBlockInfo const bi = blocks[b];
        u32 const s1 = bi.nsym_end; int const last = (b + 1 == mm.nblocks) ? 1 : 0;
        u32 const stored_len = bi.end_pos - bi.start_pos;
        // ---- symbol statistics --------------------------------------------------
        for (int i = lane; i < 288; i += 64) lds.lfreq[i] = (i == 256) ? 1u : 0u;
        if (lane < 32) lds.dfreq[lane] = 0;
        if (lane < 20) lds.blfreq[lane] = 0;
        kx_sync();
        for (u32 i = s0 + (u32)lane; i < s1; i += 64) {
            u32 const sy = syms[i]; u32 const dist = sy & 0xFFFFu, lc = sy >> 16;
            if (dist == 0) kx_lds_inc(&lds.lfreq[lc]);
            else { u32 c, e, x; kd_len_code(lc, c, e, x); kx_lds_inc(&lds.lfreq[257 + c]); kd_dist_code(dist - 1, c, e, x); kx_lds_inc(&lds.dfreq[c]); }
        }
        kx_sync();
        // ---- trees + block type (lane 0) -------------------------------------------
        if (lane == 0) {
            KdTreeCost cost; cost.opt_len = 0; cost.static_len = 0;
            for (int n = 0; n < KD_L_CODES; n++) lds.tfreq[n] = (u16)lds.lfreq[n];
            int const lmax = kd_build_tree(lds, 0, lds.lcode, cost);
            for (int n = 0; n <= lmax; n++) lds.llen[n] = (u16)(lds.lcode[n] >> 16);
            for (int n = 0; n < KD_D_CODES; n++) lds.tfreq[n] = (u16)lds.dfreq[n];
            int const dmax = kd_build_tree(lds, 1, lds.dcode, cost);
            for (int n = 0; n <= dmax; n++) lds.dlen[n] = (u16)(lds.dcode[n] >> 16);
            kd_scan_tree(lds, lds.llen, lmax);
            kd_scan_tree(lds, lds.dlen, dmax);
            for (int n = 0; n < KD_BL_CODES; n++) lds.tfreq[n] = (u16)lds.blfreq[n];
            int const blmaxcode = kd_build_tree(lds, 2, lds.blcode, cost); (void)blmaxcode;
            int max_blindex;
            {
                const u8 order[19] = { 16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15 };
                for (max_blindex = 18; max_blindex >= 3; max_blindex--) if ((lds.blcode[order[max_blindex]] >> 16) != 0) break;
                cost.opt_len += 3 * ((u32)max_blindex + 1) + 5 + 5 + 4;
                u32 opt_lenb = (cost.opt_len + 3 + 7) >> 3; u32 const static_lenb = (cost.static_len + 3 + 7) >> 3;
                if (static_lenb <= opt_lenb) opt_lenb = static_lenb;
                u32 type;                                 // 0 stored, 1 fixed, 2 dynamic
                if (stored_len + 4 <= opt_lenb && bi.stored_ok) type = 0;
                else if (static_lenb == opt_lenb) type = 1;
                else type = 2;
                // header bits through the serial writer, continuing the stream
                KdBitW w; kdw_init(w, dst, bitpos, lds.cbuf[0]);
                kdw_put(w, (type << 1) + (u32)last, 3);
                if (type == 2) {
                    kdw_put(w, (u32)(lmax + 1 - 257), 5); kdw_put(w, (u32)(dmax + 1 - 1), 5); kdw_put(w, (u32)(max_blindex + 1 - 4), 4);
                    for (int r = 0; r <= max_blindex; r++) kdw_put(w, lds.blcode[order[r]] >> 16, 3);
                    kd_send_tree(w, lds, lds.llen, lmax);
                    kd_send_tree(w, lds, lds.dlen, dmax);
                }
                lds.bc[0] = type; lds.bc[1] = kdw_bitpos(w); lds.bc[2] = (u32)w.acc;
            }
        }
        kx_sync();
        u32 const type = lds.bc[0];
        bitpos = lds.bc[1];
        for (int i = lane; i < 128; i += 64) lds.cbuf[i] = (i == 0) ? lds.bc[2] : 0u;
        kx_sync();
        if (type == 0) {
            // stored: pad to a byte, LEN, NLEN, raw bytes
            u32 const bytepos = (bitpos + 7) >> 3;
            if (lane == 0) {
                if (bitpos & 31u) { u32 const wb = 4u * (bitpos >> 5); u32 const v = lds.cbuf[0]; for (u32 k = wb; k < bytepos; k++) dst[k] = (u8)(v >> (8 * (k - wb))); }
                dst[bytepos] = (u8)stored_len; dst[bytepos + 1] = (u8)(stored_len >> 8);
                dst[bytepos + 2] = (u8)~stored_len; dst[bytepos + 3] = (u8)(~stored_len >> 8);
            }
            kx_wave_copy(dst + bytepos + 4, src + bi.start_pos, stored_len, lane);
            bitpos = 8u * (bytepos + 4 + stored_len);
            kx_sync();
            // re-prime the pending partial word (the stream is byte aligned but maybe not word aligned)
            if (lane == 0) { u32 v = 0; u32 const wb = 4u * (bitpos >> 5); for (u32 k = wb; k < (bitpos >> 3); k++) v |= (u32)dst[k] << (8 * (k - wb)); lds.cbuf[0] = v; }
            kx_sync();
        } else {
            if (type == 1) {
                for (int i = lane; i < 288; i += 64) lds.lcode[i] = kd_static_lcode((u32)i);
                if (lane < 30) lds.dcode[lane] = kd_bi_reverse((u32)lane, 5) | (5u << 16);
                kx_sync();
            }
            // symbols, 64 per step, plus END_BLOCK as one more pseudo-symbol
            u32 const total = (s1 - s0) + 1;
            for (u32 base = 0; base < total; base += 64) {
                u32 const i = base + (u32)lane; u64 v = 0; u32 nb = 0;
                if (i < total - 1) {
                    u32 const sy = syms[s0 + i]; u32 const dist = sy & 0xFFFFu, lc = sy >> 16;
                    if (dist == 0) { u32 const c = lds.lcode[lc]; v = c & 0xFFFFu; nb = c >> 16; }
                    else {
                        u32 c, e, x; kd_len_code(lc, c, e, x);
                        u32 const cl = lds.lcode[257 + c]; v = cl & 0xFFFFu; nb = cl >> 16;
                        v |= (u64)x << nb; nb += e;
                        kd_dist_code(dist - 1, c, e, x);
                        u32 const cd = lds.dcode[c]; v |= (u64)(cd & 0xFFFFu) << nb; nb += cd >> 16;
                        v |= (u64)x << nb; nb += e;
                    }
                } else if (i == total - 1) { u32 const c = lds.lcode[256]; v = c & 0xFFFFu; nb = c >> 16; }
                u32 sc = nb;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { u32 const t = kx_shfl(sc, lane - o); if (lane >= o) sc += t; }
                u32 const tot = kx_bcast(sc, 63);
                kd_cbuf_put(lds.cbuf, (bitpos & 31u) + (sc - nb), v, nb);
                kx_sync();
                u32 const nbits = (bitpos & 31u) + tot, nfull = nbits >> 5;
                u8* const wbase = dst + 4u * (bitpos >> 5);
                for (u32 wv = (u32)lane; wv < nfull; wv += 64) kx_st32(wbase + 4u * wv, lds.cbuf[wv]);
                u32 const carry = lds.cbuf[nfull];
                kx_sync();
                for (u32 wv = (u32)lane; wv <= nfull; wv += 64) lds.cbuf[wv] = (wv == 0) ? carry : 0u;
                kx_sync();
                bitpos += tot;
            }
        }
        s0 = s1;
    }
    // bi_windup of the last block: flush the pending partial word
    u32 total_bytes = (bitpos + 7) >> 3;
    u32 const adler = (a.format == 1) ? kx_wave_adler32(src, a.in_len[slice], lane) : 0u;
    u32 const crc = (a.format == 2) ? kx_wave_crc32(src, a.in_len[slice], lds.lfreq, lane) : 0u;
    if (lane == 0) {
        u32 const wb = 4u * (bitpos >> 5); u32 const v = lds.cbuf[0];
        for (u32 k = wb; k < total_bytes; k++) dst[k] = (u8)(v >> (8 * (k - wb)));
        if (a.format == 1) { dst[total_bytes] = (u8)(adler >> 24); dst[total_bytes + 1] = (u8)(adler >> 16); dst[total_bytes + 2] = (u8)(adler >> 8); dst[total_bytes + 3] = (u8)adler; total_bytes += 4; }
        if (a.format == 2) { kx_st32(dst + total_bytes, crc); kx_st32(dst + total_bytes + 4, a.in_len[slice]); total_bytes += 8; }
        a.out_len[slice] = total_bytes;
    }
}

KX_DEV void deflate_encode_body(const KdArgs& a)
{
    KX_SHARED KdEncLds lds;
    int const lane = kx_lane();
    for (u32 it = kx_block(); it < a.n_slices; it += kx_nblocks()) {
        u32 const slice = kx_xcd_chunk(it, a.n_slices);
        deflate_encode_slice(a, lds, slice, lane);
        kx_sync();
    }
}
