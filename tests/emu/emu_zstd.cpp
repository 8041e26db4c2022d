// TEST INFRASTRUCTURE: runs the product kernel bodies on the CPU wave emulator.
#include "kx_wave.h"
#include "emu_core.h"
#include "zstd_match.h"
#include <stdlib.h>
#include <vector>

extern "C" __attribute__((visibility("default")))
int emu_zstd_match(const u8* src, const u64* in_off, const u32* in_len, u32 n, int G, u32 nblocks,
                   KSeq* seqs, u32 seq_cap, u8* lits, u32 lit_cap, KSliceMeta* meta, u32 start_epoch)
{
    u32 const nteams = nblocks * (64 / G);
    bool const l4 = getenv("KXEMU_LEVEL") && atoi(getenv("KXEMU_LEVEL")) == 4;          // level 4's double-fast row: larger tables
    std::vector<u32> tables((size_t)nteams * (l4 ? KX_TBL4_ENTRIES : KX_TBL_ENTRIES), 0xDEADBEEFu & 0x0003FFFFu);   // stale junk with epoch 0
    std::vector<u32> epoch(nteams, start_epoch);
    u32 counter = 0;
    KMatchArgs a;
    a.src = src; a.in_off = in_off; a.in_len = in_len; a.n_slices = n;
    a.seqs = seqs; a.seq_cap = seq_cap; a.lits = lits; a.lit_cap = lit_cap; a.meta = meta;
    a.tables = tables.data(); a.team_epoch = epoch.data(); a.counter = &counter; a.flags = 0; a.fstate = nullptr; a.big_tables = nullptr;
    if (l4) { a.tbl_stride = KX_TBL4_ENTRIES; a.tbl_long = KX_TBL4_LONG; a.level = 4; }
    if (getenv("KXEMU_MATCH_FLAGS")) a.flags |= (u32)atoi(getenv("KXEMU_MATCH_FLAGS")) & 128u;       // (bit 7: adaptive speculation width)
    kxemu::failed = 0;
    switch (G) {
    case 2:  kxemu::launch(nblocks, [&]() { zstd_match_body<2>(a); }); break;
    case 4:  kxemu::launch(nblocks, [&]() { zstd_match_body<4>(a); }); break;
    case 8:  kxemu::launch(nblocks, [&]() { zstd_match_body<8>(a); }); break;
    case 16: kxemu::launch(nblocks, [&]() { zstd_match_body<16>(a); }); break;
    case 32: kxemu::launch(nblocks, [&]() { zstd_match_body<32>(a); }); break;
    case 64: kxemu::launch(nblocks, [&]() { zstd_match_body<64>(a); }); break;
    default: return -2;
    }
    return kxemu::failed ? -1 : 0;
}

#include "zstd_match2.h"
// The split-phase parser (zstd_match2.h): same interface, same results.  ring = bytes of a team's window in LDS.
extern "C" __attribute__((visibility("default")))
int emu_zstd_match2(const u8* src, const u64* in_off, const u32* in_len, u32 n, int G, u32 nblocks,
                    KSeq* seqs, u32 seq_cap, u8* lits, u32 lit_cap, KSliceMeta* meta, u32 start_epoch)
{
    int const ring = getenv("KXEMU_RING") ? atoi(getenv("KXEMU_RING")) : 256;
    u32 const nteams = nblocks * (64 / G);
    std::vector<u32> tables((size_t)nteams * KX_TBL_ENTRIES, 0xDEADBEEFu & 0x0003FFFFu);   // stale junk with epoch 0
    std::vector<u32> epoch(nteams, start_epoch);
    u32 counter = 0;
    KMatchArgs a;
    a.src = src; a.in_off = in_off; a.in_len = in_len; a.n_slices = n;
    a.seqs = seqs; a.seq_cap = seq_cap; a.lits = lits; a.lit_cap = lit_cap; a.meta = meta;
    a.tables = tables.data(); a.team_epoch = epoch.data(); a.counter = &counter; a.flags = 0; a.fstate = nullptr; a.big_tables = nullptr;
    kxemu::failed = 0;
    switch (G * 1000 + ring) {
    case 2256:  kxemu::launch(nblocks, [&]() { zstd_match2_body<2, 256>(a); }); break;
    case 4256:  kxemu::launch(nblocks, [&]() { zstd_match2_body<4, 256>(a); }); break;
    case 4512:  kxemu::launch(nblocks, [&]() { zstd_match2_body<4, 512>(a); }); break;
    case 8512:  kxemu::launch(nblocks, [&]() { zstd_match2_body<8, 512>(a); }); break;
    default: return -2;
    }
    return kxemu::failed ? -1 : 0;
}

#include "zstd_entropy.h"

// Full compress pipeline (match kernel + entropy kernel) on the emulator.
extern "C" __attribute__((visibility("default")))
int emu_zstd_compress(const u8* src, const u64* in_off, const u32* in_len, u32 n, int G, u32 nblocks,
                      u8* dst, const u64* out_off, u32* out_len, u32 slice_cap)
{
    u32 const seq_cap = (slice_cap / 4 + 8 + 15) & ~15u, lit_cap = slice_cap + 64, scratch_words = slice_cap / 4 + 64;
    std::vector<KSeq> seqs((size_t)n * seq_cap);
    std::vector<u8> lits((size_t)n * lit_cap, 0xEE);
    std::vector<KSliceMeta> meta(n);
    std::vector<u32> scratch((size_t)n * scratch_words, 0xA5A5A5A5u);
    // KXEMU_MATCH_V2=1: the split-phase parser (zstd_match2.h; it copies no literals, the entropy kernel gathers them)
    bool const v2 = getenv("KXEMU_MATCH_V2") && atoi(getenv("KXEMU_MATCH_V2")) != 0;
    // KXEMU_FUSE=1: k_zstd_l3_fused's body (the entropy stage inside the parse kernel's waves)
    if (getenv("KXEMU_FUSE") && atoi(getenv("KXEMU_FUSE")) != 0 && (G == 4 || G == 8)) {
        u32 const nteams = nblocks * (64 / G);
        std::vector<u32> tables((size_t)nteams * KX_TBL_ENTRIES, 0xDEADBEEFu & 0x0003FFFFu);
        std::vector<u32> epoch(nteams, 7);
        u32 counter = 0;
        KMatchArgs a;
        a.src = src; a.in_off = in_off; a.in_len = in_len; a.n_slices = n;
        a.seqs = seqs.data(); a.seq_cap = seq_cap; a.lits = lits.data(); a.lit_cap = lit_cap; a.meta = meta.data();
        a.tables = tables.data(); a.team_epoch = epoch.data(); a.counter = &counter; a.flags = 0; a.fstate = nullptr; a.big_tables = nullptr;
        KEntropyArgs e;
        e.src = src; e.in_off = in_off; e.in_len = in_len; e.n_slices = n;
        e.seqs = seqs.data(); e.seq_cap = seq_cap; e.lits = lits.data(); e.lit_cap = lit_cap; e.meta = meta.data();
        e.scratch = scratch.data(); e.scratch_words = scratch_words;
        e.dst = dst; e.out_off = out_off; e.out_len = out_len; e.flags = 0u;
        kxemu::failed = 0;
        if (G == 4) kxemu::launch(nblocks, [&]() { zstd_l3_fused_body<4>(a, e); });
        else kxemu::launch(nblocks, [&]() { zstd_l3_fused_body<8>(a, e); });
        if (kxemu::failed) return -1;
        for (u32 i = 0; i < n; i++) if (meta[i].status) return -3;
        return 0;
    }
    int r = v2 ? emu_zstd_match2(src, in_off, in_len, n, G, nblocks, seqs.data(), seq_cap, lits.data(), lit_cap, meta.data(), 7)
               : emu_zstd_match(src, in_off, in_len, n, G, nblocks, seqs.data(), seq_cap, lits.data(), lit_cap, meta.data(), 7);
    if (r) return r;
    for (u32 i = 0; i < n; i++) if (meta[i].status) return -3;
    KEntropyArgs e;
    e.src = src; e.in_off = in_off; e.in_len = in_len; e.n_slices = n;
    e.seqs = seqs.data(); e.seq_cap = seq_cap; e.lits = lits.data(); e.lit_cap = lit_cap; e.meta = meta.data();
    e.scratch = scratch.data(); e.scratch_words = scratch_words;
    e.dst = dst; e.out_off = out_off; e.out_len = out_len; e.flags = v2 ? 8u : 0u;
    kxemu::failed = 0;
    kxemu::launch(nblocks, [&]() { zstd_entropy_body(e); });
    return kxemu::failed ? -1 : 0;
}

#include "zstd_match_fast.h"
// Levels 1 and 2 (strategy fast): fast match kernel + entropy kernel on the emulator.
extern "C" __attribute__((visibility("default")))
int emu_zstd_compress_level(const u8* src, const u64* in_off, const u32* in_len, u32 n, int G, u32 nblocks,
                            u8* dst, const u64* out_off, u32* out_len, u32 slice_cap, int level)
{
    u32 const seq_cap = (slice_cap / 4 + 8 + 15) & ~15u, lit_cap = slice_cap + 64, scratch_words = slice_cap / 4 + 64;
    std::vector<KSeq> seqs((size_t)n * seq_cap);
    std::vector<u8> lits((size_t)n * lit_cap, 0xEE);
    std::vector<KSliceMeta> meta(n);
    std::vector<u32> scratch((size_t)n * scratch_words, 0xA5A5A5A5u);
    u32 const nteams = nblocks * (64 / G);
    std::vector<u32> tables((size_t)nteams * KX_TBL_ENTRIES, 0xDEADBEEFu & 0x0003FFFFu);
    std::vector<u32> epoch(nteams, 7);
    u32 counter = 0;
    KFastArgs g;
    g.m.src = src; g.m.in_off = in_off; g.m.in_len = in_len; g.m.n_slices = n;
    g.m.seqs = seqs.data(); g.m.seq_cap = seq_cap; g.m.lits = lits.data(); g.m.lit_cap = lit_cap; g.m.meta = meta.data();
    g.m.tables = tables.data(); g.m.team_epoch = epoch.data(); g.m.counter = &counter; g.m.flags = 6; g.m.fstate = nullptr; g.m.big_tables = nullptr;
    g.level = level < 0 ? 0u : (u32)level; g.step0 = level < 0 ? (u32)(1 - level) : 2u;     // negative levels: row 0, a step of 1 - level
    kxemu::failed = 0;
    switch (G) {
    case 2:  kxemu::launch(nblocks, [&]() { zstd_match_fast_body<2>(g); }); break;
    case 4:  kxemu::launch(nblocks, [&]() { zstd_match_fast_body<4>(g); }); break;
    case 8:  kxemu::launch(nblocks, [&]() { zstd_match_fast_body<8>(g); }); break;
    case 16: kxemu::launch(nblocks, [&]() { zstd_match_fast_body<16>(g); }); break;
    default: return -2;
    }
    if (kxemu::failed) return -1;
    for (u32 i = 0; i < n; i++) if (meta[i].status) return -3;
    KEntropyArgs e;
    e.src = src; e.in_off = in_off; e.in_len = in_len; e.n_slices = n;
    e.seqs = seqs.data(); e.seq_cap = seq_cap; e.lits = lits.data(); e.lit_cap = lit_cap; e.meta = meta.data();
    e.scratch = scratch.data(); e.scratch_words = scratch_words;
    e.dst = dst; e.out_off = out_off; e.out_len = out_len; e.flags = 8u | 32u | (level < 0 ? 64u : 0u);
    kxemu::launch(nblocks, [&]() { zstd_entropy_body(e); });
    return kxemu::failed ? -1 : 0;
}

#include "zstd_match_dict.h"
#include "zstd_cdict_host.h"
// Compress with a raw-content dictionary: dictionary match kernel + entropy kernel on the emulator.
extern "C" __attribute__((visibility("default")))
int emu_zstd_compress_dict(const u8* src, const u64* in_off, const u32* in_len, u32 n, int G, u32 nblocks,
                           u8* dst, const u64* out_off, u32* out_len, u32 slice_cap, const u8* dict, u32 dict_size)
{
    u32 const seq_cap = (slice_cap / 4 + 8 + 15) & ~15u, lit_cap = slice_cap + 64, scratch_words = slice_cap / 4 + 64;
    std::vector<KSeq> seqs((size_t)n * seq_cap);
    std::vector<u8> lits((size_t)n * lit_cap, 0xEE);
    std::vector<KSliceMeta> meta(n);
    std::vector<u32> scratch((size_t)n * scratch_words, 0xA5A5A5A5u);
    u32 const nteams = nblocks * (64 / G);
    std::vector<u32> tables((size_t)nteams * KX_TBL_ENTRIES, 0xDEADBEEFu & 0x0003FFFFu);
    std::vector<u32> epoch(nteams, 7);
    u32 counter = 0, W, C, H, M;
    // (a formatted dictionary: the host code's steps of kmp_zstd_compress_batch_dict)
    KDictPrior prior; size_t content_off = 0;
    int const formatted = cdict_parse_formatted(dict, dict_size, &prior, &content_off);
    if (formatted < 0) return -4;
    cdict_params(dict_size, &W, &C, &H, &M);
    dict += content_off; dict_size -= (u32)content_off;
    std::vector<u32> tl, ts;
    cdict_fill(tl, H, ts, C, M, dict, dict_size);
    KDictArgs g;
    if (formatted) { g.rep0 = prior.rep[0]; g.rep1 = prior.rep[1]; }
    g.m.src = src; g.m.in_off = in_off; g.m.in_len = in_len; g.m.n_slices = n;
    g.m.seqs = seqs.data(); g.m.seq_cap = seq_cap; g.m.lits = lits.data(); g.m.lit_cap = lit_cap; g.m.meta = meta.data();
    g.m.tables = tables.data(); g.m.team_epoch = epoch.data(); g.m.counter = &counter; g.m.flags = 6; g.m.fstate = nullptr; g.m.big_tables = nullptr;
    g.dict = dict; g.dict_size = dict_size; g.dictL = tl.data(); g.dictS = ts.data();
    g.dWindowLog = W; g.dHashLog = H; g.dChainLog = C; g.dMinMatch = M;
    kxemu::failed = 0;
    switch (G) {
    case 2:  kxemu::launch(nblocks, [&]() { zstd_match_dict_body<2>(g); }); break;
    case 4:  kxemu::launch(nblocks, [&]() { zstd_match_dict_body<4>(g); }); break;
    case 8:  kxemu::launch(nblocks, [&]() { zstd_match_dict_body<8>(g); }); break;
    case 16: kxemu::launch(nblocks, [&]() { zstd_match_dict_body<16>(g); }); break;
    default: return -2;
    }
    if (kxemu::failed) return -1;
    for (u32 i = 0; i < n; i++) if (meta[i].status) return -3;
    KEntropyArgs e;
    e.src = src; e.in_off = in_off; e.in_len = in_len; e.n_slices = n;
    e.seqs = seqs.data(); e.seq_cap = seq_cap; e.lits = lits.data(); e.lit_cap = lit_cap; e.meta = meta.data();
    e.scratch = scratch.data(); e.scratch_words = scratch_words;
    e.dst = dst; e.out_off = out_off; e.out_len = out_len; e.flags = 8u;
    if (formatted) { e.prior = &prior; kxemu::launch(nblocks, [&]() { zstd_entropy_body<true>(e); }); }
    else kxemu::launch(nblocks, [&]() { zstd_entropy_body(e); });
    return kxemu::failed ? -1 : 0;
}

#include "zstd_lazy.h"
// Levels 5 .. 10 (greedy / lazy / lazy2): sort body (workgroups of four waves), parse body, entropy body -- zstd_compress_lazy's steps.
extern "C" __attribute__((visibility("default")))
int emu_zstd_compress_lazy(const u8* src, const u64* in_off, const u32* in_len, u32 n, u32 nblocks,
                           u8* dst, const u64* out_off, u32* out_len, u32 slice_cap, int level)
{
    u32 const seq_cap = (slice_cap / 4 + 8 + 15) & ~15u, lit_cap = slice_cap + 64, scratch_words = slice_cap / 4 + 64, pos_cap = (slice_cap + 63u) & ~63u;
    std::vector<KSeq> seqs((size_t)n * seq_cap);
    std::vector<u8> lits((size_t)n * lit_cap, 0xEE);
    std::vector<KSliceMeta> meta(n);
    std::vector<u32> scratch((size_t)n * scratch_words, 0xA5A5A5A5u);
    std::vector<u32> wr((size_t)n * pos_cap, 0xCCCCCCCCu); std::vector<KLazyRec> rec((size_t)n * pos_cap);
    memset(rec.data(), 0xBB, rec.size() * sizeof(KLazyRec));
    for (u32 i = 0; i < n; i++) { memset(&meta[i], 0, sizeof(meta[i])); meta[i].lastLL = in_len[i]; meta[i].status = 3; }      // (level 4: what the double-fast kernel's slices look like to this harness: not served here)
    KLazyArgs g;
    g.src = src; g.in_off = in_off; g.in_len = in_len; g.n_slices = n;
    g.rec = rec.data(); g.wr = wr.data(); g.pos_cap = pos_cap;
    g.seqs = seqs.data(); g.seq_cap = seq_cap; g.meta = meta.data(); g.level = (u32)level;
    kxemu::failed = 0;
    kxemu::launch_block(nblocks, 4, [&]() { zstd_lazy_sort_body(g); });
    if (kxemu::failed) return -1;
    if (slice_cap <= 65536u) kxemu::launch(nblocks, [&]() { zstd_lazy_body<2048>(g); }); else kxemu::launch(nblocks, [&]() { zstd_lazy_body<4096>(g); });
    if (kxemu::failed) return -2;
    for (u32 i = 0; i < n; i++) if (meta[i].status == 2) return -3;
    KEntropyArgs e;
    e.src = src; e.in_off = in_off; e.in_len = in_len; e.n_slices = n;
    e.seqs = seqs.data(); e.seq_cap = seq_cap; e.lits = lits.data(); e.lit_cap = lit_cap; e.meta = meta.data();
    e.scratch = scratch.data(); e.scratch_words = scratch_words;
    e.dst = dst; e.out_off = out_off; e.out_len = out_len; e.flags = 8u | ((u32)level << 12);
    kxemu::launch(nblocks, [&]() { zstd_entropy_body(e); });
    if (kxemu::failed) return -4;
    for (u32 i = 0; i < n; i++) if (meta[i].status == 3) out_len[i] = 0;          // (k_len_guard_finish: another strategy at this size)
    return 0;
}

// Frames of several blocks (slices above 128 KiB): the host-side round loop of kmp_api.hip restated for the emulator.
extern "C" __attribute__((visibility("default")))
int emu_zstd_compress_big_ex(const u8* src, const u64* in_off, const u32* in_len, u32 n, int G, u32 nblocks,
                             u8* dst, const u64* out_off, u32* out_len, u32* rounds_out, u32 stream);
extern "C" __attribute__((visibility("default")))
int emu_zstd_compress_big(const u8* src, const u64* in_off, const u32* in_len, u32 n, int G, u32 nblocks,
                          u8* dst, const u64* out_off, u32* out_len, u32* rounds_out)
{ return emu_zstd_compress_big_ex(src, in_off, in_len, n, G, nblocks, dst, out_off, out_len, rounds_out, 0); }
extern "C" __attribute__((visibility("default")))
int emu_zstd_compress_big_ex2(const u8* src, const u64* in_off, const u32* in_len, u32 n, int G, u32 nblocks,
                              u8* dst, const u64* out_off, u32* out_len, u32* rounds_out, u32 stream_and_strategy, u32 tail_or_chunk, u32 wide);
extern "C" __attribute__((visibility("default")))
int emu_zstd_compress_big_ex(const u8* src, const u64* in_off, const u32* in_len, u32 n, int G, u32 nblocks,
                             u8* dst, const u64* out_off, u32* out_len, u32* rounds_out, u32 stream_and_strategy)
{ return emu_zstd_compress_big_ex2(src, in_off, in_len, n, G, nblocks, dst, out_off, out_len, rounds_out, stream_and_strategy, 0, 0); }
// stream (low byte): KFrameArgs.stream, 0 .. 3; tail_or_chunk: tail_direct of a stream, out_chunk of the one-shot driver (mode 3);
// wide: table entries without check bits (what contexts for slices of 4 MiB and more use)
extern "C" __attribute__((visibility("default")))
int emu_zstd_compress_big_ex2(const u8* src, const u64* in_off, const u32* in_len, u32 n, int G, u32 nblocks,
                              u8* dst, const u64* out_off, u32* out_len, u32* rounds_out, u32 stream_and_strategy, u32 tail_or_chunk, u32 wide)
{
    u32 const stream = stream_and_strategy & 0xFFu; u32 strategy = (stream_and_strategy >> 8) & 0xFFu;     // strategy 1: level 1 (fast); 2: level 2 (as zstd_compress_big in kmp_api.hip)
    bool const level4 = strategy == 4u; if (level4) strategy = 0;                                           // 4: level 4's double-fast rows (the level-3 kernels, larger tables)
    u32 const fast_step0 = strategy == 1u ? stream_and_strategy >> 16 : 0u;                               // bits 16 ..: a negative level's step (1 - level), with strategy 1
    u32 const level2 = strategy == 2u ? 1u : 0u;
    bool const streaming = stream == 1 || stream == 2;
    u32 const block_cap = 128u * 1024u;
    u32 const seq_cap = (block_cap / 4 + 8 + 15) & ~15u, lit_cap = block_cap + 64, scratch_words = block_cap / 4 + 64;
    std::vector<KSeq> seqs((size_t)n * seq_cap);
    std::vector<u8> lits((size_t)n * lit_cap, 0xEE);
    std::vector<KSliceMeta> meta(n);
    std::vector<u32> scratch((size_t)n * scratch_words, 0xA5A5A5A5u);
    std::vector<KFrameState> fstate(n);
    std::vector<u32> hufct((size_t)n * 512, 0xDEADBEEFu);
    std::vector<u32> big_tables((size_t)n * (level4 ? KX_BIG4_ENTRIES : KX_BIG_TBL_ENTRIES), 0u);
    u32 remaining = 0, counter = 0;
    for (u32 i = 0; i < n; i++) {
        KFrameState s; memset(&s, 0, sizeof(s));
        s.blockSize = in_len[i] < KX_BLOCK_MAX ? in_len[i] : KX_BLOCK_MAX; s.first = 1; s.rep[0] = 1; s.rep[1] = 4; s.rep[2] = 8;
        s.lowLimit = 2; s.dictLimit = 2; s.chunkEnd = (stream != 0 && in_len[i] > KX_BLOCK_MAX) ? KX_BLOCK_MAX : in_len[i];
        if (level4 && !streaming && !kx_l4_served(in_len[i])) { s.blockSize = 0; s.chunkEnd = 0; fstate[i] = s; out_len[i] = 0; continue; }      // (k_zstd_frame_init: refused size class)
        fstate[i] = s;
        if (in_len[i] == 0) { u8* d = dst + out_off[i]; u32 const magic = 0xFD2FB528u; memcpy(d, &magic, 4); d[4] = streaming ? 0x00 : 0x20; d[5] = streaming ? (strategy == 1u ? 0x48 : strategy == 2u ? 0x50 : 0x58) : 0; d[6] = 1; d[7] = 0; d[8] = 0; out_len[i] = 9; }
        else remaining++;
    }
    KMatchArgs m;
    m.src = src; m.in_off = in_off; m.in_len = in_len; m.n_slices = n;
    m.seqs = seqs.data(); m.seq_cap = seq_cap; m.lits = lits.data(); m.lit_cap = lit_cap; m.meta = meta.data();
    m.tables = nullptr; m.team_epoch = nullptr; m.counter = &counter; m.flags = (streaming ? 8u : 0u) | (wide ? 16u : 0u);
    m.fstate = fstate.data(); m.big_tables = big_tables.data();
    if (level4) { m.level = 4; m.big_stride = KX_BIG4_ENTRIES; m.big_long = KX_BIG4_LONG; }
    KFrameArgs e;
    e.src = src; e.in_off = in_off; e.in_len = in_len; e.n_slices = n;
    e.seqs = seqs.data(); e.seq_cap = seq_cap; e.lits = lits.data(); e.lit_cap = lit_cap; e.meta = meta.data();
    e.scratch = scratch.data(); e.scratch_words = scratch_words;
    e.dst = dst; e.out_off = out_off; e.out_len = out_len;
    e.fstate = fstate.data(); e.hufct = hufct.data(); e.remaining = &remaining; e.stream = stream; e.strategy = strategy ? 1u : 0u; e.level2 = level2; e.cls = 0; e.fast_step0 = fast_step0;
    e.tail_direct = stream == 3 ? 0u : tail_or_chunk; e.out_chunk = stream == 3 ? tail_or_chunk : 0u;
    if (strategy && rounds_out) return -6;
    if (!rounds_out) {
        // product path: one wave per slice walks its chain of blocks
        std::vector<u32> counters(nblocks, 0u);
        KBigArgs g; g.m = m; g.e = e; g.counters = counters.data(); g.spw = (n > 2 && 64 / G >= 2) ? 2 : 1;
        kxemu::failed = 0;
        if (level2 && !streaming) {
            // level 2, sizes known: the slices of its double-fast row first (class 1), the others (class 2) through the fast parser below
            g.e.strategy = 0; g.e.cls = 1; g.m.flags = m.flags | 32u | (1u << 6);
            switch (G) {
            case 2:  kxemu::launch(nblocks, [&]() { zstd_big_body<2>(g); }); break;
            case 4:  kxemu::launch(nblocks, [&]() { zstd_big_body<4>(g); }); break;
            case 8:  kxemu::launch(nblocks, [&]() { zstd_big_body<8>(g); }); break;
            case 16: kxemu::launch(nblocks, [&]() { zstd_big_body<16>(g); }); break;
            case 32: kxemu::launch(nblocks, [&]() { zstd_big_body<32>(g); }); break;
            case 64: kxemu::launch(nblocks, [&]() { zstd_big_body<64>(g); }); break;
            default: return -2;
            }
            if (kxemu::failed) return -1;
            for (auto& x : counters) x = 0;
            g.e.strategy = 1; g.e.cls = 2; g.m.flags = m.flags | (2u << 6);
        }
        if (strategy) switch (G) {
        case 2:  kxemu::launch(nblocks, [&]() { zstd_big_body<2, true>(g); }); break;
        case 4:  kxemu::launch(nblocks, [&]() { zstd_big_body<4, true>(g); }); break;
        case 8:  kxemu::launch(nblocks, [&]() { zstd_big_body<8, true>(g); }); break;
        case 16: kxemu::launch(nblocks, [&]() { zstd_big_body<16, true>(g); }); break;
        case 32: kxemu::launch(nblocks, [&]() { zstd_big_body<32, true>(g); }); break;
        case 64: kxemu::launch(nblocks, [&]() { zstd_big_body<64, true>(g); }); break;
        default: return -2;
        }
        else switch (G) {
        case 2:  kxemu::launch(nblocks, [&]() { zstd_big_body<2>(g); }); break;
        case 4:  kxemu::launch(nblocks, [&]() { zstd_big_body<4>(g); }); break;
        case 8:  kxemu::launch(nblocks, [&]() { zstd_big_body<8>(g); }); break;
        case 16: kxemu::launch(nblocks, [&]() { zstd_big_body<16>(g); }); break;
        case 32: kxemu::launch(nblocks, [&]() { zstd_big_body<32>(g); }); break;
        case 64: kxemu::launch(nblocks, [&]() { zstd_big_body<64>(g); }); break;
        default: return -2;
        }
        if (kxemu::failed) return -1;
        for (u32 i = 0; i < n; i++) if (fstate[i].blockSize != 0) return -5;
        return 0;
    }
    u32 rounds = 0;
    while (remaining != 0) {
        if (++rounds > 20000) return -4;
        counter = 0; kxemu::failed = 0;
        switch (G) {
        case 2:  kxemu::launch(nblocks, [&]() { zstd_match_body<2, true>(m); }); break;
        case 4:  kxemu::launch(nblocks, [&]() { zstd_match_body<4, true>(m); }); break;
        case 8:  kxemu::launch(nblocks, [&]() { zstd_match_body<8, true>(m); }); break;
        case 16: kxemu::launch(nblocks, [&]() { zstd_match_body<16, true>(m); }); break;
        case 32: kxemu::launch(nblocks, [&]() { zstd_match_body<32, true>(m); }); break;
        case 64: kxemu::launch(nblocks, [&]() { zstd_match_body<64, true>(m); }); break;
        default: return -2;
        }
        if (kxemu::failed) return -1;
        // the blocks libzstd parses with the extDict variant (behind a wrap of its staging buffer)
        counter = 0;
        switch (G) {
        case 2:  kxemu::launch(nblocks, [&]() { zstd_match_ext_body<2>(m); }); break;
        case 4:  kxemu::launch(nblocks, [&]() { zstd_match_ext_body<4>(m); }); break;
        case 8:  kxemu::launch(nblocks, [&]() { zstd_match_ext_body<8>(m); }); break;
        case 16: kxemu::launch(nblocks, [&]() { zstd_match_ext_body<16>(m); }); break;
        case 32: kxemu::launch(nblocks, [&]() { zstd_match_ext_body<32>(m); }); break;
        default: kxemu::launch(nblocks, [&]() { zstd_match_ext_body<64>(m); }); break;
        }
        if (kxemu::failed) return -1;
        for (u32 i = 0; i < n; i++) if (fstate[i].blockSize >= 8 && meta[i].status) return -3;
        kxemu::launch(nblocks, [&]() { zstd_frame_body(e); });
        if (kxemu::failed) return -1;
    }
    if (rounds_out) *rounds_out = rounds;
    return 0;
}

#include "zstd_decode.h"
#include "zstd_predecode.h"
extern "C" __attribute__((visibility("default")))
int emu_zstd_decompress_dict(const u8* src, const u64* in_off, const u32* in_len, u32 n, u32 nblocks,
                             u8* dst, const u64* out_off, const u32* out_cap, u32* out_len, u32* status, u32 lit_cap,
                             const u8* dict, u32 dict_size);
extern "C" __attribute__((visibility("default")))
int emu_zstd_decompress(const u8* src, const u64* in_off, const u32* in_len, u32 n, u32 nblocks,
                        u8* dst, const u64* out_off, const u32* out_cap, u32* out_len, u32* status, u32 lit_cap)
{ return emu_zstd_decompress_dict(src, in_off, in_len, n, nblocks, dst, out_off, out_cap, out_len, status, lit_cap, nullptr, 0); }
extern "C" __attribute__((visibility("default")))
int emu_zstd_decompress_dict(const u8* src, const u64* in_off, const u32* in_len, u32 n, u32 nblocks,
                             u8* dst, const u64* out_off, const u32* out_cap, u32* out_len, u32* status, u32 lit_cap,
                             const u8* dict, u32 dict_size)
{
    std::vector<u8> lits((size_t)n * lit_cap, 0xEE);
    KDecodeArgs d;
    d.src = src; d.in_off = in_off; d.in_len = in_len; d.n_slices = n;
    d.dst = dst; d.out_off = out_off; d.out_cap = out_cap; d.out_len = out_len; d.status = status;
    d.lits = lits.data(); d.lit_cap = lit_cap; d.flags = 0; d.dict = dict; d.dict_size = dict ? dict_size : 0;
    // (a formatted dictionary: the host code's steps of zstd_decompress_impl)
    KDictDPrior dprior; u32 start_rep[3] = { 1, 4, 8 };
    if (dict && dict_size >= 8) {
        size_t off = 0;
        int const formatted = cdict_parse_formatted(dict, dict_size, nullptr, &off, &dprior);
        if (formatted < 0) return -7;
        if (formatted) { d.dict = dict + off; d.dict_size = dict_size - (u32)off; d.dprior = &dprior; d.dict_id = dprior.dictID; start_rep[0] = dprior.rep[0]; start_rep[1] = dprior.rep[1]; start_rep[2] = dprior.rep[2]; }
    }
    // sequences decoded ahead, one lane per frame (what the product does); KXEMU_NO_PRE=1: everything in the decode body
    u32 const seq_cap = lit_cap / 3u + 64u, blk_cap = lit_cap / 8192u + 16u;
    std::vector<u64> stage; std::vector<KPreBlk> pblk; std::vector<u32> nblk;
    d.pre_stage = nullptr; d.pre_seq_cap = 0; d.pre_blk = nullptr; d.pre_blk_cap = 0; d.pre_nblk = nullptr;
    kxemu::failed = 0;
    // as the product: the pre-decoders take their entries in order of sequence count (the product sorts batches of 1024 and more;
    // here every batch of 4 and more, so that the CPU suite covers the mapping)
    std::vector<u32> skey(n, 0u), sperm(n, 0u), shist(KXP_SORT_BUCKETS, 0u);
    bool const sorted = n >= 4 && !getenv("KXEMU_NO_PRE");
    if (sorted) {
        KSeqSortArgs sa;
        sa.src = src; sa.in_off = in_off; sa.in_len = in_len; sa.n_slices = n; sa.key = skey.data(); sa.hist = shist.data(); sa.perm = sperm.data(); sa.len_shift = 0;
        kxemu::launch_block((n + 255) / 256, 4, [&]() { zstd_seq_count_body(sa); });
        kxemu::launch_block(1, 4, [&]() { zstd_seq_rank_body(sa); });
        kxemu::launch_block((n + 255) / 256, 4, [&]() { zstd_seq_perm_body(sa); });
        if (kxemu::failed) return -5;
    }
    if (!getenv("KXEMU_NO_PRE")) {
        stage.assign((size_t)n * seq_cap, 0xCDCDCDCDCDCDCDCDull); pblk.resize((size_t)n * blk_cap); nblk.assign(n, 0u);
        KPreArgs p;
        p.src = src; p.in_off = in_off; p.in_len = in_len; p.n_slices = n;
        p.stage = stage.data(); p.seq_cap = seq_cap; p.blk = pblk.data(); p.blk_cap = blk_cap; p.nblk = nblk.data(); p.perm = sorted ? sperm.data() : nullptr;
        p.rep[0] = start_rep[0]; p.rep[1] = start_rep[1]; p.rep[2] = start_rep[2];
        kxemu::launch((n + KXP_FRAMES - 1) / KXP_FRAMES, [&]() { zstd_seq_predecode_body(p); });
        if (kxemu::failed) return -2;
        d.pre_stage = stage.data(); d.pre_seq_cap = seq_cap; d.pre_blk = pblk.data(); d.pre_blk_cap = blk_cap; d.pre_nblk = nblk.data();
    }
    std::vector<u8> plits; std::vector<KPreLit> plrec; std::vector<u32> nlit;
    d.pre_lits = nullptr; d.pre_lit_cap = 0; d.pre_lit = nullptr; d.pre_nlit = nullptr; d.pre_blk_cap = blk_cap;
    if (!getenv("KXEMU_NO_PRE")) {
        u32 const plcap = lit_cap + 64u;
        plits.assign((size_t)n * plcap, 0xABu); plrec.resize((size_t)n * blk_cap); nlit.assign(n, 0u);
        KLitArgs p;
        p.src = src; p.in_off = in_off; p.in_len = in_len; p.n_slices = n;
        p.lits = plits.data(); p.lit_cap = plcap; p.rec = plrec.data(); p.blk_cap = blk_cap; p.nrec = nlit.data(); p.perm = sorted ? sperm.data() : nullptr;
        kxemu::launch((n + KXL_FRAMES - 1) / KXL_FRAMES, [&]() { zstd_lit_predecode_body(p); });
        if (kxemu::failed) return -3;
        d.pre_lits = plits.data(); d.pre_lit_cap = plcap; d.pre_lit = plrec.data(); d.pre_nlit = nlit.data();
    }
    kxemu::launch(nblocks, [&]() { zstd_decode_body(d); });
    return kxemu::failed ? -1 : 0;
}

// the counting sort that orders the pre-decoders' lane slots by sequence count (zstd_predecode.h): count, rank, perm
extern "C" __attribute__((visibility("default")))
int emu_seq_sort_by(const u8* src, const u64* in_off, const u32* in_len, u32 n, u32* key, u32* perm, u32 len_shift);
extern "C" __attribute__((visibility("default")))
int emu_seq_sort(const u8* src, const u64* in_off, const u32* in_len, u32 n, u32* key, u32* perm) { return emu_seq_sort_by(src, in_off, in_len, n, key, perm, 0); }
// len_shift != 0: keyed by the entry's size (what inflate's pre-decoder is given)
extern "C" __attribute__((visibility("default")))
int emu_seq_sort_by(const u8* src, const u64* in_off, const u32* in_len, u32 n, u32* key, u32* perm, u32 len_shift)
{
    std::vector<u32> hist(KXP_SORT_BUCKETS, 0u);
    KSeqSortArgs sa;
    sa.src = src; sa.in_off = in_off; sa.in_len = in_len; sa.n_slices = n; sa.key = key; sa.hist = hist.data(); sa.perm = perm; sa.len_shift = len_shift;
    kxemu::failed = 0;
    kxemu::launch_block((n + 255) / 256, 4, [&]() { zstd_seq_count_body(sa); });
    if (kxemu::failed) return -1;
    kxemu::launch_block(1, 4, [&]() { zstd_seq_rank_body(sa); });
    if (kxemu::failed) return -2;
    kxemu::launch_block((n + 255) / 256, 4, [&]() { zstd_seq_perm_body(sa); });
    return kxemu::failed ? -3 : 0;
}

#include "deflate_match.h"
#include "deflate_lazy.h"
#include "deflate_encode.h"
// raw DEFLATE level 6 pipeline (chains -> best -> parse -> encode) on the emulator
extern "C" __attribute__((visibility("default")))
int emu_deflate_level(const u8* src, const u64* in_off, const u32* in_len, u32 n, u8* dst, const u64* out_off, u32* out_len,
                      u16* link_out, KdBest* best_out, u32 format, int level);
extern "C" __attribute__((visibility("default")))
int emu_deflate(const u8* src, const u64* in_off, const u32* in_len, u32 n, u8* dst, const u64* out_off, u32* out_len,
                u16* link_out, KdBest* best_out, u32 format)
{ return emu_deflate_level(src, in_off, in_len, n, dst, out_off, out_len, link_out, best_out, format, 6); }
extern "C" __attribute__((visibility("default")))
int emu_deflate_params(const u8* src, const u64* in_off, const u32* in_len, u32 n, u8* dst, const u64* out_off, u32* out_len,
                       u16* link_out, KdBest* best_out, u32 format, int level, int window_bits, int mem_level, int old_kernels);
extern "C" __attribute__((visibility("default")))
int emu_deflate_level(const u8* src, const u64* in_off, const u32* in_len, u32 n, u8* dst, const u64* out_off, u32* out_len,
                      u16* link_out, KdBest* best_out, u32 format, int level)
{ return emu_deflate_params(src, in_off, in_len, n, dst, out_off, out_len, link_out, best_out, format, level, 15, 8, 0); }
// ... with deflateInit2's windowBits / memLevel; old_kernels: the chain / all-positions search / lane-per-slice parse also for slices up to 64 KiB
extern "C" __attribute__((visibility("default")))
int emu_deflate_params(const u8* src, const u64* in_off, const u32* in_len, u32 n, u8* dst, const u64* out_off, u32* out_len,
                       u16* link_out, KdBest* best_out, u32 format, int level, int window_bits, int mem_level, int old_kernels)
{
    u32 maxlen = 65536u;
    for (u32 i = 0; i < n; i++) if (in_len[i] > maxlen) maxlen = in_len[i];
    u32 const pos_cap = (maxlen + 63u) & ~63u, blk_cap = kd_block_cap(pos_cap, 1u << (mem_level + 6));
    std::vector<u16> link((size_t)n * pos_cap, 0xEEEE);
    std::vector<KdBest> best((size_t)n * pos_cap * 2);
    std::vector<u32> syms((size_t)n * pos_cap, 0xDDDDDDDDu);
    std::vector<u32> wrv((size_t)n * pos_cap, 0xCCCCCCCCu);
    std::vector<KdSliceMeta> meta(n);
    std::vector<KdBlockInfo> blocks((size_t)n * blk_cap);
    KdArgs a;
    a.wr = wrv.data();
    a.src = src; a.in_off = in_off; a.in_len = in_len; a.n_slices = n;
    a.pos_cap = pos_cap; a.blk_cap = blk_cap; a.blocks = blocks.data();
    a.link = link.data(); a.best = best.data(); a.syms = syms.data(); a.meta = meta.data();
    a.dst = dst; a.out_off = out_off; a.out_len = out_len; a.flags = 0; a.format = format;
    kd_level_config(a, level, window_bits, mem_level);
    kxemu::failed = 0;
    if (level >= 1 && level <= 3) {
        memset((void*)best.data(), 0, (size_t)n * (a.hmask + 1u) * 4u);
        kxemu::launch((n + 63) / 64, [&]() { deflate_fast_body(a); });
        if (kxemu::failed) return -3;
        kxemu::launch(n < 3 ? n : 3, [&]() { deflate_encode_body(a); });
        return kxemu::failed ? -4 : 0;
    }
    // slices up to 64 KiB: the sort + wave-wide lazy parse of deflate_lazy.h (what the product runs there), unless the caller wants the
    // chain links / per-position records of the older kernels back (link_out / best_out) -- the tests keep both pipelines honest
    if (pos_cap > 65536u && !link_out && !best_out && !old_kernels) {
        // slices above 64 KiB: segment by segment, as deflate_batch_impl does (the search arrays hold one 64 KiB span per slice)
        std::vector<u16> rankv((size_t)n * 65536u, 0xBBBB);
        std::vector<u32> statev((size_t)n * KDL_STATE_WORDS, 0xAAAAAAAAu);
        a.seg_rank = rankv.data(); a.seg_state = statev.data();
        u32 const segs = (maxlen - KDL_SEG_SPAN + KDL_SEG_STEP - 1u) / KDL_SEG_STEP + 1u;
        for (u32 seg = 0; seg < segs; seg++) {
            a.seg = seg;
            if (a.hmask > 0x7FFFu) kxemu::launch_block(n < 2 ? n : 2, 4, [&]() { deflate_sort_body<16, true>(a); });
            else kxemu::launch_block(n < 2 ? n : 2, 4, [&]() { deflate_sort_body<15, true>(a); });
            if (kxemu::failed) return -1;
            kxemu::launch(n < 3 ? n : 3, [&]() { deflate_lazy_body<true>(a); });
            if (kxemu::failed) return -3;
        }
        kxemu::launch(n < 3 ? n : 3, [&]() { deflate_encode_body(a); });
        return kxemu::failed ? -4 : 0;
    }
    if (pos_cap <= 65536u && !link_out && !best_out && !old_kernels) {
        if (a.hmask > 0x7FFFu) kxemu::launch_block(n < 2 ? n : 2, 4, [&]() { deflate_sort_body<16>(a); });
        else kxemu::launch_block(n < 2 ? n : 2, 4, [&]() { deflate_sort_body<15>(a); });
        if (kxemu::failed) return -1;
        kxemu::launch(n < 3 ? n : 3, [&]() { deflate_lazy_body<false>(a); });
        if (kxemu::failed) return -3;
        kxemu::launch(n < 3 ? n : 3, [&]() { deflate_encode_body(a); });
        return kxemu::failed ? -4 : 0;
    }
    if (pos_cap <= 65536u) kxemu::launch_block(n < 2 ? n : 2, 4, [&]() { deflate_chains_body<u16>(a); });
    else kxemu::launch_block(n < 2 ? n : 2, 4, [&]() { deflate_chains_body<u32>(a); });
    if (kxemu::failed) return -1;
    kxemu::launch_block(n < 2 ? n : 2, 16, [&]() { deflate_best_body(a); });
    if (kxemu::failed) return -2;
    // the parse over the records: a wave per slice (what the product runs), or the earlier lane per slice (old_kernels == 2)
    if (old_kernels == 2) kxemu::launch((n + 63) / 64, [&]() { deflate_parse_body(a); });
    else kxemu::launch(n < 3 ? n : 3, [&]() { deflate_parse_wave_body(a); });
    if (kxemu::failed) return -3;
    kxemu::launch(n < 3 ? n : 3, [&]() { deflate_encode_body(a); });
    if (kxemu::failed) return -4;
    if (link_out) memcpy(link_out, link.data(), link.size() * 2);
    if (best_out) memcpy(best_out, best.data(), (size_t)n * pos_cap * sizeof(KdBest));
    return 0;
}

#include "deflate_predecode.h"
// the two-kernel inflate (lane-per-stream pre-decoder, then the executor / inflate_stream for what it did not cover);
// covered_out[i] = 1 where the pre-decoder's staging was executed.  stage_bytes: the size the staging is made for
// (max_slice_bytes of a context)
extern "C" __attribute__((visibility("default")))
int emu_inflate_pre(const u8* src, const u64* in_off, const u32* in_len, u32 n, u8* dst, const u64* out_off, const u32* out_cap,
                    u32* out_len, int* status, u32 format, u32 stage_bytes, u32* covered_out)
{
    u32 const seq_cap = stage_bytes / 3u + 64u, lit_cap = stage_bytes + 64u;
    std::vector<u64> stage((size_t)n * seq_cap, 0xDDDDDDDDDDDDDDDDull);
    std::vector<u8> lits((size_t)n * lit_cap, 0xEE);
    std::vector<u32> nseq(n, 0x12345678u), nlit(n, 0x12345678u);
    KipArgs p;
    p.src = src; p.in_off = in_off; p.in_len = in_len; p.n_slices = n; p.out_cap = out_cap; p.format = format;
    p.stage = stage.data(); p.seq_cap = seq_cap; p.lits = lits.data(); p.lit_cap = lit_cap; p.nseq = nseq.data(); p.nlit = nlit.data();
    std::vector<u32> perm(n), skey(n);                      // the host's order: by compressed size, largest first
    { u32 sh = 1; while ((stage_bytes >> sh) >= KXP_SORT_BUCKETS) sh++; if (emu_seq_sort_by(src, in_off, in_len, n, skey.data(), perm.data(), sh) != 0) return -3; }
    p.perm = perm.data();
    kxemu::failed = 0;
    kxemu::launch((n + KIP_STREAMS - 1) / KIP_STREAMS, [&]() { inflate_predecode_body(p); });
    if (kxemu::failed) return -1;
    for (u32 i = 0; i < n; i++) if (covered_out) covered_out[i] = nseq[i] >> 31;
    KieArgs e;
    e.i.src = src; e.i.in_off = in_off; e.i.in_len = in_len; e.i.n_slices = n;
    e.i.dst = dst; e.i.out_off = out_off; e.i.out_cap = out_cap; e.i.out_len = out_len; e.i.status = status; e.i.format = format;
    e.stage = stage.data(); e.seq_cap = seq_cap; e.lits = lits.data(); e.lit_cap = lit_cap; e.nseq = nseq.data(); e.nlit = nlit.data();
    kxemu::launch(n < 3 ? n : 3, [&]() { inflate_exec_body(e); });
    return kxemu::failed ? -2 : 0;
}

extern "C" __attribute__((visibility("default")))
int emu_inflate(const u8* src, const u64* in_off, const u32* in_len, u32 n, u8* dst, const u64* out_off, const u32* out_cap,
                u32* out_len, int* status, u32 format)
{
    KiArgs a;
    a.src = src; a.in_off = in_off; a.in_len = in_len; a.n_slices = n;
    a.dst = dst; a.out_off = out_off; a.out_cap = out_cap; a.out_len = out_len; a.status = status; a.format = format;
    kxemu::failed = 0;
    kxemu::launch(n < 3 ? n : 3, [&]() { inflate_body(a); });
    return kxemu::failed ? -1 : 0;
}

// kx_xcd_chunk (zstd_common.h): the slice a virtual workgroup index takes
extern "C" __attribute__((visibility("default")))
unsigned emu_xcd_chunk(unsigned it, unsigned n) { return kx_xcd_chunk(it, n); }

