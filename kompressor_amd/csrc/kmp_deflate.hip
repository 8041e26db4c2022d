// kmp_deflate.hip -- the zlib twin (BASELINE configs[4]): DEFLATE / inflate kernel entry points and the batched calls
// kmp_deflate_compress_batch* / kmp_zlib_compress_batch / kmp_gzip_compress_batch / kmp_inflate_batch over the batch
// context of kmp_batch.hip (reference: kompressor-zlib--nativelib/src/jvmCommonMain/jni/Wrapper.cpp:20,73,91,144).
#include "kx_wave.h"
#include "zstd_common.h"
#include "deflate_match.h"
#include "deflate_lazy.h"
#include "deflate_encode.h"
#include "deflate_decode.h"
#include "deflate_predecode.h"
#include "kmp_internal.h"

__global__ __launch_bounds__(256) void k_deflate_chains(KdArgs a) { deflate_chains_body<u16>(a); }          // slices <= 64 KiB
__global__ __launch_bounds__(256) void k_deflate_chains_long(KdArgs a) { deflate_chains_body<u32>(a); }
__global__ __launch_bounds__(1024) void k_deflate_best(KdArgs a) { deflate_best_body(a); }
__global__ __launch_bounds__(64) void k_deflate_parse(KdArgs a) { deflate_parse_body(a); }
__global__ __launch_bounds__(64, 8) void k_deflate_parse_wave(KdArgs a) { deflate_parse_wave_body(a); }   // the same parse, a wave per slice (deflate_lazy.h)
__global__ __launch_bounds__(64) void k_deflate_fast(KdArgs a) { deflate_fast_body(a); }
// the lazy levels for slices up to 64 KiB: positions sorted by hash, then the parse with a wave-wide longest_match (deflate_lazy.h)
__global__ __launch_bounds__(256) void k_deflate_sort(KdArgs a) { deflate_sort_body<15>(a); }
__global__ __launch_bounds__(256) void k_deflate_sort_wide(KdArgs a) { deflate_sort_body<16>(a); }        // memLevel 9: 65 536 buckets (128 KiB of LDS)
// slices above 64 KiB, one segment per launch (deflate_lazy.h: a 64 KiB span of every slice is sorted, half of it parsed)
__global__ __launch_bounds__(256) void k_deflate_sort_seg(KdArgs a) { deflate_sort_body<15, true>(a); }
__global__ __launch_bounds__(256) void k_deflate_sort_seg_wide(KdArgs a) { deflate_sort_body<16, true>(a); }
__global__ __launch_bounds__(64, 8) void k_deflate_lazy_seg(KdArgs a) { deflate_lazy_body<true>(a); }
__global__ __launch_bounds__(256) void k_max_len(const u32* len, u32 n, u32* out)
{
    u32 const i = blockIdx.x * 256u + threadIdx.x;
    u32 v = i < n ? len[i] : 0u;
    for (int o = 32; o; o >>= 1) { u32 const t = (u32)__shfl_xor((int)v, o); v = t > v ? t : v; }
    if ((threadIdx.x & 63u) == 0 && v) atomicMax(out, v);
}
__global__ __launch_bounds__(64, 8) void k_deflate_lazy(KdArgs a) { deflate_lazy_body<false>(a); }
__global__ __launch_bounds__(64, 2) void k_inflate_predecode(KipArgs a) { inflate_predecode_body(a); }
__global__ __launch_bounds__(64) void k_inflate_exec(KieArgs a) { inflate_exec_body(a); }
__global__ __launch_bounds__(64, 4) void k_deflate_encode(KdArgs a) { deflate_encode_body(a); }
__global__ __launch_bounds__(64, 5) void k_inflate(KiArgs a) { inflate_body(a); }

// zlib's deflateBound for the default parameters plus the largest wrapper: stored blocks (5 bytes each per 16 383-symbol
// block at worst) + 7 for the end of the stream + 18 for a gzip header and trailer (zlib wrapper: 6, raw: 0)
extern "C" size_t kmp_deflate_bound(size_t n) { return n + (n >> 12) + (n >> 14) + (n >> 25) + 7 + 18; }

/* any of zlib's levels 1 .. 9 (-1 = 6): 4 .. 9 (deflate_slow) are the same kernels with the level's good / lazy / nice / chain
 * numbers, 1 .. 3 (deflate_fast) one kernel that parses and keeps its hash chains a lane per slice */
extern "C" int kmp_deflate_compress_batch_level(kmp_batch_ctx* c, const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                                uint32_t n, void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len, int format, int level, void* hip_stream)
{
    if (level == -1) level = 6;
    if (format < 0 || format > 2 || level < 1 || level > 9) { g_last_error = "kmp_deflate_compress_batch_level: format 0 (raw), 1 (zlib) or 2 (gzip), level 1 .. 9"; return KMP_ERR_ARG; }
    return deflate_batch_impl(c, d_src, d_in_off, d_in_len, n, d_dst, d_out_off, d_out_len, (u32)format, hip_stream, level);
}
/* ... and any of deflateInit2's windowBits 9 .. 15 (8 is served as 9, as zlib does: deflate.c deflateInit2_) and memLevel 1 .. 9 */
extern "C" int kmp_deflate_compress_batch_params(kmp_batch_ctx* c, const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                                 uint32_t n, void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len, int format, int level,
                                                 int window_bits, int mem_level, void* hip_stream)
{
    if (level == -1) level = 6;
    if (format < 0 || format > 2 || level < 1 || level > 9 || window_bits < 8 || window_bits > 15 || mem_level < 1 || mem_level > 9 || (window_bits == 8 && format != 1)) {
        g_last_error = "kmp_deflate_compress_batch_params: format 0 (raw), 1 (zlib) or 2 (gzip), level 1 .. 9, windowBits 9 .. 15 (8 with the zlib wrapper only, as in zlib), memLevel 1 .. 9";
        return KMP_ERR_ARG;
    }
    return deflate_batch_impl(c, d_src, d_in_off, d_in_len, n, d_dst, d_out_off, d_out_len, (u32)format, hip_stream, level, window_bits, mem_level);
}
/* room for a stream of any of those settings: zlib's deflateBound for non-default parameters (every byte a nine-bit literal of a fixed
 * block, a block header per lit_bufsize - 1 symbols) + the largest wrapper */
extern "C" size_t kmp_deflate_bound_params(size_t n, int window_bits, int mem_level)
{
    if (window_bits == 15 && mem_level == 8) return kmp_deflate_bound(n);
    return n + ((n + 7) >> 3) + ((n + 63) >> 6) + 5 + 18;
}
extern "C" int kmp_deflate_compress_batch(kmp_batch_ctx* c, const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                          uint32_t n, void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len, void* hip_stream)
{ return deflate_batch_impl(c, d_src, d_in_off, d_in_len, n, d_dst, d_out_off, d_out_len, 0, hip_stream); }
extern "C" int kmp_zlib_compress_batch(kmp_batch_ctx* c, const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                       uint32_t n, void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len, void* hip_stream)
{ return deflate_batch_impl(c, d_src, d_in_off, d_in_len, n, d_dst, d_out_off, d_out_len, 1, hip_stream); }
extern "C" int kmp_gzip_compress_batch(kmp_batch_ctx* c, const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                       uint32_t n, void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len, void* hip_stream)
{ return deflate_batch_impl(c, d_src, d_in_off, d_in_len, n, d_dst, d_out_off, d_out_len, 2, hip_stream); }

extern "C" int kmp_inflate_batch(kmp_batch_ctx* c, const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len, uint32_t n,
                                 void* d_dst, const uint64_t* d_out_off, const uint32_t* d_out_cap, uint32_t* d_out_len, int32_t* d_status,
                                 int format, void* hip_stream)
{ return inflate_batch_impl(c, d_src, d_in_off, d_in_len, n, d_dst, d_out_off, d_out_cap, d_out_len, d_status, format, 0, hip_stream); }

// window_bits: what the caller declared to inflateInit2 (8 .. 15; 0 = 15): zlib streams whose header names a larger window are refused
int inflate_batch_impl(kmp_batch_ctx* c, const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len, uint32_t n,
                       void* d_dst, const uint64_t* d_out_off, const uint32_t* d_out_cap, uint32_t* d_out_len, int32_t* d_status,
                       int format, int window_bits, void* hip_stream)
{
    if (format < 0 || format > 3 || window_bits < 0 || window_bits > 15) { g_last_error = "kmp_inflate_batch: format must be 0 (raw), 1 (zlib), 2 (gzip) or 3 (zlib or gzip)"; return KMP_ERR_ARG; }
    if (!c || (n && (!d_src || !d_in_off || !d_in_len || !d_dst || !d_out_off || !d_out_cap || !d_out_len || !d_status))) { g_last_error = "kmp_inflate_batch: null argument"; return KMP_ERR_ARG; }
    if (n == 0) return KMP_OK;
    hipStream_t const st = (hipStream_t)hip_stream;
    HIP_TRY(hipSetDevice(c->device));
    KiArgs a;
    a.src = (const u8*)d_src; a.in_off = d_in_off; a.in_len = d_in_len; a.n_slices = n;
    a.dst = (u8*)d_dst; a.out_off = d_out_off; a.out_cap = d_out_cap; a.out_len = d_out_len; a.status = d_status; a.format = (u32)format | ((u32)window_bits << 8);
    KMP_TRY(batch_begin(c, st, nullptr, n, 0));
    bool const use_pre = c->knob.inflate_pre && n >= env_pre_min_batch();
    if (use_pre) ensure_pre_staging(c);
    if (use_pre && c->pre_stage && c->pre_lits && c->pre_nblk && c->pre_nlit && c->pre_slices) {
        // two kernels: a lane per stream decodes the Huffman codes into staged literals and match records (the staging of
        // the zstd decoder), a wave per stream executes them -- and decodes the streams the first kernel did not cover
        // One piece when the batch fits the staging; a larger batch goes through in pieces of the staging's size, one after the
        // other.  (KMP_INFLATE_PIECES=2..4 splits a batch that fits, each piece with its own part of the staging and its executor
        // on the context's second stream beside the next piece's pre-decoder: measured 55 / 44 / 36 GB/s against 64 in one
        // piece -- the launches get short and their tails long, as in the zstd decoder.)
        u32 pieces = (n <= c->pre_slices && n >= 16384u && c->knob.inflate_pieces > 1) ? c->knob.inflate_pieces : 1u;
        if (pieces > KMP_MAX_CHUNKS) pieces = KMP_MAX_CHUNKS;
        bool const overlap = pieces > 1;
        u32 const per = overlap ? (((n + pieces - 1) / pieces + 1023u) & ~1023u) : c->pre_slices;
        if (overlap) { HIP_TRY(hipEventRecord(c->ev_pre[0], st)); HIP_TRY(hipStreamWaitEvent(c->st2, c->ev_pre[0], 0)); }
        u32 pi = 0;
        for (u32 first = 0; first < n; first += per, pi++) {
            u32 const m = (n - first < per) ? n - first : per;
            size_t const so = overlap ? first : 0;                      // this piece's place in the staging
            // streams of similar compressed size (about as many symbols) share a wave: a wave lasts as long as its longest lane
            u32* const sort_key = c->pre_sort ? c->pre_sort + so : nullptr; u32* const sort_perm = c->pre_sort ? c->pre_sort + c->pre_slices + so : nullptr; u32* const sort_hist = c->pre_sort ? c->pre_sort + 2u * (size_t)c->pre_slices : nullptr;
            bool const sorted = c->pre_sort && c->knob.decode_sort != 0 && m >= 1024u;
            if (sorted) {
                u32 sh = 1; while ((c->max_slice_bytes >> sh) >= 256u) sh++;          // (256 = the sort's buckets)
                KMP_TRY(size_sort(c, st, a.src, a.in_off + first, a.in_len + first, m, sort_key, sort_hist, sort_perm, sh));
            }
            KipArgs p;
            p.perm = sorted ? sort_perm : nullptr;
            p.src = a.src; p.in_off = a.in_off + first; p.in_len = a.in_len + first; p.n_slices = m; p.out_cap = a.out_cap + first; p.format = a.format;
            p.stage = c->pre_stage + so * c->pre_seq_cap; p.seq_cap = c->pre_seq_cap; p.lits = c->pre_lits + so * c->pre_lit_cap; p.lit_cap = c->pre_lit_cap; p.nseq = c->pre_nblk + so; p.nlit = c->pre_nlit + so;
            hipLaunchKernelGGL(k_inflate_predecode, dim3((m + KIP_STREAMS - 1) / KIP_STREAMS), dim3(64), 0, st, p);
            HIP_TRY(hipGetLastError());
            hipStream_t es = st;
            if (overlap) { es = c->st2; HIP_TRY(hipEventRecord(c->ev_pre[1 + pi], st)); HIP_TRY(hipStreamWaitEvent(es, c->ev_pre[1 + pi], 0)); }
            KieArgs e;
            e.i = a; e.i.in_off += first; e.i.in_len += first; e.i.n_slices = m; e.i.out_off += first; e.i.out_cap += first; e.i.out_len += first; e.i.status += first;
            e.stage = p.stage; e.seq_cap = c->pre_seq_cap; e.lits = p.lits; e.lit_cap = c->pre_lit_cap; e.nseq = p.nseq; e.nlit = p.nlit;
            hipLaunchKernelGGL(k_inflate_exec, dim3(m), dim3(64), 0, es, e);
            HIP_TRY(hipGetLastError());
        }
        if (overlap) { HIP_TRY(hipEventRecord(c->ev_join, c->st2)); HIP_TRY(hipStreamWaitEvent(st, c->ev_join, 0)); }
        return batch_end(c, st, nullptr, n, 0, nullptr, nullptr);
    }
    hipLaunchKernelGGL(k_inflate, dim3(n), dim3(64), 0, st, a);
    HIP_TRY(hipGetLastError());
    return batch_end(c, st, nullptr, n, 0, nullptr, nullptr);
}

int deflate_batch_impl(kmp_batch_ctx* c, const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                       uint32_t n, void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len, u32 format, void* hip_stream, int level,
                       int window_bits, int mem_level)
{
    if (!c || (n && (!d_src || !d_in_off || !d_in_len || !d_dst || !d_out_off || !d_out_len))) { g_last_error = "kmp_deflate_compress_batch: null argument"; return KMP_ERR_ARG; }
    if (n > c->max_slices) { g_last_error = "kmp_deflate_compress_batch: n exceeds the context's max_slices"; return KMP_ERR_CAPACITY; }
    if (n == 0) return KMP_OK;
    hipStream_t const st = (hipStream_t)hip_stream;
    HIP_TRY(hipSetDevice(c->device));
    if (!c->dfl_link) {
        // Two workspace halves of up to 16 384 slices each: while the sort works on one piece of the batch, the parse and the encoder of
        // the previous piece run beside it on the context's second stream.  Per slice the search arrays hold 65 536 positions (26 bytes
        // each: srt, sb, wr, symbol) -- for a context of longer slices too, whose slices go through in segments of one 64 KiB span
        // (deflate_lazy.h; 28 bytes with the ranks, which cannot sit in the symbol array there), plus 4 bytes per position of the whole slice for the symbols.
        u32 const pos_cap = ((c->max_slice_bytes < 65536u ? 65536u : c->max_slice_bytes) + 63u) & ~63u;
        bool const lng = pos_cap > 65536u;
        size_t const span = 65536u;
        u32 cap = c->knob.dfl_chunk ? c->knob.dfl_chunk : 16384u;
        // (above 64 KiB the span arrays exist twice: the sort of segment k + 1 fills one copy while the parse of segment k reads the other)
        size_t const copies = lng ? 2u : 1u;
        size_t const per_slot = span * (copies * (sizeof(u16) + 2 * sizeof(KdBest) + sizeof(u32)) + (lng ? sizeof(u16) : 0)) + (lng ? (size_t)pos_cap * sizeof(u32) : 0);
        if (lng && (u64)cap * per_slot > (16ull << 30)) cap = (u32)((16ull << 30) / per_slot);          // 16 GiB a half
        if (cap < 1) cap = 1;
        u32 const chunk = c->max_slices < cap ? c->max_slices : cap;
        c->dfl_pos_cap = pos_cap; c->dfl_blk_cap = pos_cap / (KD_LIT_BUFSIZE - 1) + 2u;
        HIP_TRY(hipMalloc((void**)&c->dfl_link, (size_t)2 * copies * chunk * span * sizeof(u16)));
        HIP_TRY(hipMalloc((void**)&c->dfl_best, (size_t)2 * copies * chunk * span * sizeof(KdBest) * 2u));
        HIP_TRY(hipMalloc((void**)&c->dfl_syms, (size_t)2 * chunk * pos_cap * sizeof(u32)));
        HIP_TRY(hipMalloc((void**)&c->dfl_wr, (size_t)2 * copies * chunk * span * sizeof(u32)));       // (deflate_lazy.h: where / rank of every position)
        HIP_TRY(hipMalloc((void**)&c->dfl_order, ((size_t)2 * (2 * chunk + 256)) * sizeof(u32)));  // (per half: cost classes, their histogram, the slices in order)
        if (lng) {
            HIP_TRY(hipMalloc((void**)&c->dfl_rank, (size_t)2 * chunk * span * sizeof(u16)));
            HIP_TRY(hipMalloc((void**)&c->dfl_state, (size_t)2 * chunk * KDL_STATE_WORDS * sizeof(u32)));
            HIP_TRY(hipMalloc((void**)&c->dfl_maxlen, 64));
            for (int i = 0; i < 2; i++) {
                HIP_TRY(hipStreamCreateWithFlags(&c->dfl_sort_st[i], hipStreamNonBlocking));
                for (int k = 0; k < 2; k++) { HIP_TRY(hipEventCreateWithFlags(&c->dfl_sorted[i][k], hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&c->dfl_parsed[i][k], hipEventDisableTiming)); }
            }
            c->dfl_seg_sync = 1;
        }
        HIP_TRY(hipMalloc((void**)&c->dfl_meta, (size_t)2 * chunk * sizeof(KdSliceMeta)));
        HIP_TRY(hipMalloc((void**)&c->dfl_blocks, (size_t)2 * chunk * c->dfl_blk_cap * sizeof(KdBlockInfo)));
        for (int i = 0; i < 2; i++) {
            HIP_TRY(hipEventCreateWithFlags(&c->dfl_searched[i], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&c->dfl_done[i], hipEventDisableTiming));
        }
        c->dfl_events = 1;
        c->dfl_chunk = chunk;
    }
    {   // a smaller memLevel closes a block after fewer symbols (lit_bufsize - 1 = 127 at memLevel 1): room for the block lists
        u32 const need = c->dfl_pos_cap / ((1u << ((mem_level < 1 ? 1 : mem_level > 9 ? 9 : mem_level) + 6)) - 1u) + 2u;
        if (need > c->dfl_blk_cap) {
            HIP_TRY(hipDeviceSynchronize());
            (void)hipFree(c->dfl_blocks); c->dfl_blocks = nullptr;
            HIP_TRY(hipMalloc((void**)&c->dfl_blocks, (size_t)2 * c->dfl_chunk * need * sizeof(KdBlockInfo)));
            if (c->dfl_fblocks) { (void)hipFree(c->dfl_fblocks); c->dfl_fblocks = nullptr; HIP_TRY(hipMalloc((void**)&c->dfl_fblocks, (size_t)4 * c->dfl_chunk * need * sizeof(KdBlockInfo))); }
            c->dfl_blk_cap = need;
        }
    }
    u32 const dfl_cap = c->max_slice_bytes < 65536u ? 65536u : c->max_slice_bytes;       // a context for smaller slices still takes 64 KiB ones
    KMP_TRY(batch_begin(c, st, d_in_len, n, dfl_cap));
    if (c->profiling) HIP_TRY(hipEventRecord(c->ev[6], st));
    u32 chain_waves = c->knob.dfl_chain_waves; if (chain_waves < 1 || chain_waves > 4) chain_waves = 4;
    bool const serial = c->knob.dfl_serial != 0;          // experiment switch: everything on the caller's stream
    u32 piece = 0;
    if (level >= 1 && level <= 3) {
        // deflate_fast needs no link / best arrays: the head / prev tables of a slice (256 KiB) live where the KdBest entries of
        // the lazy levels do, which has room for 4 * dfl_chunk slices.  The kernel is a lane per slice and bound by memory
        // latency, so the more slices are in flight the better: symbols and block lists for that many slices are allocated
        // on the first call at these levels (16 GiB with the default sizes; if that fails, pieces of 2 * dfl_chunk slices use
        // the arrays of the lazy levels).  Everything runs on the caller's stream.
        if (!c->dfl_ftried) {
            c->dfl_ftried = 1;
            size_t const cap4 = (size_t)4 * c->dfl_chunk;
            if (c->max_slices > 2u * c->dfl_chunk && !c->knob.dfl_serial) {
                if (hipMalloc((void**)&c->dfl_fsyms, cap4 * c->dfl_pos_cap * sizeof(u32)) != hipSuccess ||
                    hipMalloc((void**)&c->dfl_fmeta, cap4 * sizeof(KdSliceMeta)) != hipSuccess ||
                    hipMalloc((void**)&c->dfl_fblocks, cap4 * c->dfl_blk_cap * sizeof(KdBlockInfo)) != hipSuccess) {
                    (void)hipGetLastError();
                    (void)hipFree(c->dfl_fsyms); (void)hipFree(c->dfl_fmeta); (void)hipFree(c->dfl_fblocks);
                    c->dfl_fsyms = nullptr; c->dfl_fmeta = nullptr; c->dfl_fblocks = nullptr;
                }
            }
        }
        bool const wide = c->dfl_fsyms && c->dfl_fmeta && c->dfl_fblocks;
        u32 span = (wide ? 4u : 2u) * c->dfl_chunk;
        {   // the head table of a slice has 1 << (memLevel + 7) entries, its prev table 32 768: as many slices at a time as the workspace holds
            size_t const ws = (size_t)2 * (c->dfl_pos_cap > 65536u ? 2u : 1u) * c->dfl_chunk * 65536u * sizeof(KdBest) * 2u;
            size_t const per = ((size_t)(1u << ((mem_level > 9 ? 9 : mem_level) + 7)) + KD_WSIZE) * sizeof(u32);
            if ((size_t)span * per > ws) span = (u32)(ws / per);
            if (span < 1) { g_last_error = "kmp_deflate_compress_batch: workspace too small for this memLevel"; return KMP_ERR_CAPACITY; }
        }
        for (u32 first = 0; first < n; first += span) {
            u32 const m = (n - first < span) ? n - first : span;
            KdArgs a;
            a.src = (const u8*)d_src; a.in_off = d_in_off + first; a.in_len = c->len_ok + first; a.n_slices = m;
            a.pos_cap = c->dfl_pos_cap; a.blk_cap = c->dfl_blk_cap;
            a.link = c->dfl_link; a.best = c->dfl_best;
            a.syms = wide ? c->dfl_fsyms : c->dfl_syms; a.meta = wide ? c->dfl_fmeta : c->dfl_meta; a.blocks = wide ? c->dfl_fblocks : c->dfl_blocks;
            a.dst = (u8*)d_dst; a.out_off = d_out_off + first; a.out_len = d_out_len + first; a.flags = c->knob.dfl_flags; a.format = format;
            kd_level_config(a, level, window_bits, mem_level);
            bool const prof = c->profiling && first == 0;
            if (prof) { HIP_TRY(hipEventRecord(c->ev[8], st)); HIP_TRY(hipEventRecord(c->ev[9], st)); HIP_TRY(hipEventRecord(c->ev[10], st)); HIP_TRY(hipEventRecord(c->ev[13], st)); }
            HIP_TRY(hipMemsetAsync(c->dfl_best, 0, (size_t)m * (a.hmask + 1u) * sizeof(u32), st));          // the head tables
            hipLaunchKernelGGL(k_deflate_fast, dim3((m + 63) / 64), dim3(64), 0, st, a);
            if (prof) HIP_TRY(hipEventRecord(c->ev[11], st));
            hipLaunchKernelGGL(k_deflate_encode, dim3(m), dim3(64), 0, st, a);
            if (prof) HIP_TRY(hipEventRecord(c->ev[12], st));
            HIP_TRY(hipGetLastError());
        }
        if (c->profiling) { HIP_TRY(hipEventRecord(c->ev[7], st)); c->ev_valid[3] = 1; }
        return batch_end(c, st, d_in_len, n, dfl_cap, d_out_len, nullptr);
    }
    if (c->dfl_pos_cap > 65536u) {
        // Slices above 64 KiB: segment by segment (deflate_lazy.h).  The longest slice of the batch says how many launches there are; the
        // two halves of the workspace take alternate pieces, each on its own stream from its first sort to its encoder.
        u32 maxlen = 0;
        HIP_TRY(hipMemsetAsync(c->dfl_maxlen, 0, sizeof(u32), st));
        hipLaunchKernelGGL(k_max_len, dim3((n + 255) / 256), dim3(256), 0, st, c->len_ok, n, c->dfl_maxlen);
        HIP_TRY(hipMemcpyAsync(&maxlen, c->dfl_maxlen, sizeof(u32), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        u32 const segs = maxlen <= KDL_SEG_SPAN ? 1u : (maxlen - KDL_SEG_SPAN + KDL_SEG_STEP - 1u) / KDL_SEG_STEP + 1u;
        // Per half: a parse stream (the caller's for half 0, the context's second one for half 1) and a sort stream.  The span arrays exist in
        // two copies, taken by the parity of the segment: sort(k + 1) runs beside parse(k) -- it waits for parse(k - 1), the last reader
        // of its copy --, parse(k) waits for sort(k).  The sort leaves a piece's critical path (it was a third of a segment).
        if (!serial) {
            HIP_TRY(hipEventRecord(c->dfl_searched[0], st));
            HIP_TRY(hipStreamWaitEvent(c->st2, c->dfl_searched[0], 0));
            for (int i = 0; i < 2; i++) HIP_TRY(hipStreamWaitEvent(c->dfl_sort_st[i], c->dfl_searched[0], 0));
        }
        size_t const span_slots = (size_t)c->dfl_chunk * 65536u;
        for (u32 first = 0; first < n; first += c->dfl_chunk, piece++) {
            u32 const m = (n - first < c->dfl_chunk) ? n - first : c->dfl_chunk;
            u32 const h = piece & 1u;
            hipStream_t const sx = (h == 0 || serial) ? st : c->st2;              // parse + encode
            hipStream_t const ss = serial ? sx : c->dfl_sort_st[h];               // sort
            size_t const half = (size_t)h * c->dfl_chunk;
            KdArgs a;
            a.src = (const u8*)d_src; a.in_off = d_in_off + first; a.in_len = c->len_ok + first; a.n_slices = m;
            a.pos_cap = c->dfl_pos_cap; a.blk_cap = c->dfl_blk_cap;
            a.seg_rank = c->dfl_rank + half * 65536u; a.seg_state = c->dfl_state + half * KDL_STATE_WORDS;
            a.syms = c->dfl_syms + half * c->dfl_pos_cap; a.meta = c->dfl_meta + half; a.blocks = c->dfl_blocks + half * c->dfl_blk_cap;
            a.dst = (u8*)d_dst; a.out_off = d_out_off + first; a.out_len = d_out_len + first; a.flags = c->knob.dfl_flags; a.format = format;
            kd_level_config(a, level, window_bits, mem_level);
            bool const prof = c->profiling && first == 0;
            if (prof) { HIP_TRY(hipEventRecord(c->ev[8], sx)); HIP_TRY(hipEventRecord(c->ev[9], sx)); HIP_TRY(hipEventRecord(c->ev[10], sx)); HIP_TRY(hipEventRecord(c->ev[13], sx)); }
            if (!serial && piece >= 2) for (int k = 0; k < 2; k++) HIP_TRY(hipStreamWaitEvent(ss, c->dfl_parsed[h][k], 0));     // the previous piece of this half has read its spans
            for (u32 seg = 0; seg < segs; seg++) {
                u32 const par = seg & 1u;
                size_t const copy = ((size_t)h * 2u + par) * span_slots;          // this half's copy of that parity
                a.seg = seg;
                a.link = c->dfl_link + copy; a.best = c->dfl_best + copy * 2u; a.wr = c->dfl_wr + copy;
                if (!serial && seg >= 2) HIP_TRY(hipStreamWaitEvent(ss, c->dfl_parsed[h][par], 0));
                if (a.hmask > 0x7FFFu) hipLaunchKernelGGL(k_deflate_sort_seg_wide, dim3(m), dim3(256), 0, ss, a);
                else hipLaunchKernelGGL(k_deflate_sort_seg, dim3(m), dim3(256), 0, ss, a);
                if (!serial) { HIP_TRY(hipEventRecord(c->dfl_sorted[h][par], ss)); HIP_TRY(hipStreamWaitEvent(sx, c->dfl_sorted[h][par], 0)); }
                hipLaunchKernelGGL(k_deflate_lazy_seg, dim3(m), dim3(64), 0, sx, a);
                if (!serial) HIP_TRY(hipEventRecord(c->dfl_parsed[h][par], sx));
            }
            if (prof) HIP_TRY(hipEventRecord(c->ev[11], sx));
            hipLaunchKernelGGL(k_deflate_encode, dim3(m), dim3(64), 0, sx, a);
            if (prof) HIP_TRY(hipEventRecord(c->ev[12], sx));
            HIP_TRY(hipGetLastError());
            if (!serial) HIP_TRY(hipEventRecord(c->dfl_done[h], sx));
        }
        if (!serial && piece >= 2) HIP_TRY(hipStreamWaitEvent(st, c->dfl_done[1], 0));       // (a single piece ran on the caller's stream alone)
        if (c->profiling) { HIP_TRY(hipEventRecord(c->ev[7], st)); c->ev_valid[3] = 1; }
        return batch_end(c, st, d_in_len, n, dfl_cap, d_out_len, nullptr);
    }
    for (u32 first = 0; first < n; first += c->dfl_chunk, piece++) {
        u32 const m = (n - first < c->dfl_chunk) ? n - first : c->dfl_chunk;
        u32 const h = piece & 1u;                                        // workspace half
        KdArgs a;
        a.src = (const u8*)d_src; a.in_off = d_in_off + first; a.in_len = c->len_ok + first; a.n_slices = m;
        size_t const half = (size_t)h * c->dfl_chunk;
        a.pos_cap = c->dfl_pos_cap; a.blk_cap = c->dfl_blk_cap;
        a.link = c->dfl_link + half * c->dfl_pos_cap; a.best = c->dfl_best + half * c->dfl_pos_cap * (c->dfl_wr ? 2u : 1u);
        a.syms = c->dfl_syms + half * c->dfl_pos_cap; a.meta = c->dfl_meta + half; a.blocks = c->dfl_blocks + half * c->dfl_blk_cap;
        a.wr = c->dfl_wr ? c->dfl_wr + half * c->dfl_pos_cap : nullptr;
        a.dst = (u8*)d_dst; a.out_off = d_out_off + first; a.out_len = d_out_len + first; a.flags = c->knob.dfl_flags; a.format = format;
        kd_level_config(a, level, window_bits, mem_level);
        bool const prof = c->profiling && first == 0;      // per-kernel events for the first piece
        hipStream_t const s2 = serial ? st : c->st2;
        if (!serial && piece >= 2) HIP_TRY(hipStreamWaitEvent(st, c->dfl_done[h], 0));      // this half's previous piece has been encoded
        // slices up to 64 KiB: k_deflate_sort + k_deflate_lazy (only the positions zlib's parse asks about are searched, each by a
        // whole wave: deflate_lazy.h); longer ones: the chain / all-positions search / lane-per-slice parse of deflate_match.h
        bool const lazy2 = c->dfl_wr != nullptr && !KMP_KNOB("KMP_DEFLATE_OLD", 0);
        if (prof) HIP_TRY(hipEventRecord(c->ev[8], st));
        if (lazy2) {
            u32* const ord = c->dfl_order + (size_t)h * (2 * c->dfl_chunk + 256);
            a.order_key = ord; a.order_hist = ord + c->dfl_chunk; a.order = ord + c->dfl_chunk + 256;
            HIP_TRY(hipMemsetAsync(a.order_hist, 0, 256 * sizeof(u32), st));
            if (a.hmask > 0x7FFFu) hipLaunchKernelGGL(k_deflate_sort_wide, dim3(m), dim3(256), 0, st, a);
            else hipLaunchKernelGGL(k_deflate_sort, dim3(m), dim3(256), 0, st, a);
            KMP_TRY(size_sort_keys(c, st, m, a.order_key, a.order_hist, (u32*)a.order));
        }
        else if (c->dfl_pos_cap <= 65536u) hipLaunchKernelGGL(k_deflate_chains, dim3(m), dim3(64u * chain_waves), 0, st, a);
        else hipLaunchKernelGGL(k_deflate_chains_long, dim3(m), dim3(64u * chain_waves), 0, st, a);
        if (prof) HIP_TRY(hipEventRecord(c->ev[9], st));
        if (!lazy2) hipLaunchKernelGGL(k_deflate_best, dim3(m), dim3(1024), 0, st, a);
        if (prof) HIP_TRY(hipEventRecord(c->ev[10], st));
        if (!serial) { HIP_TRY(hipEventRecord(c->dfl_searched[h], st)); HIP_TRY(hipStreamWaitEvent(s2, c->dfl_searched[h], 0)); }
        if (prof) HIP_TRY(hipEventRecord(c->ev[13], s2));
        if (lazy2) hipLaunchKernelGGL(k_deflate_lazy, dim3(m), dim3(64), 0, s2, a);
        else if (KMP_KNOB("KMP_DEFLATE_LANE_PARSE", 0)) hipLaunchKernelGGL(k_deflate_parse, dim3((m + 63) / 64), dim3(64), 0, s2, a);
        else hipLaunchKernelGGL(k_deflate_parse_wave, dim3(m), dim3(64), 0, s2, a);
        if (prof) HIP_TRY(hipEventRecord(c->ev[11], s2));
        // (the encoder on a third stream, beside the next piece's parse, was measured: 699 ms per 65 536 slices against 652 -- three
        // kernels at once stretch each other; it stays behind its piece's parse)
        hipLaunchKernelGGL(k_deflate_encode, dim3(m), dim3(64), 0, s2, a);
        if (prof) HIP_TRY(hipEventRecord(c->ev[12], s2));
        HIP_TRY(hipGetLastError());
        if (!serial) HIP_TRY(hipEventRecord(c->dfl_done[h], s2));
    }
    if (!serial) {                                                        // the caller's stream continues when every piece is out
        HIP_TRY(hipStreamWaitEvent(st, c->dfl_done[0], 0));
        if (piece >= 2) HIP_TRY(hipStreamWaitEvent(st, c->dfl_done[1], 0));
    }
    if (c->profiling) { HIP_TRY(hipEventRecord(c->ev[7], st)); c->ev_valid[3] = 1; }
    return batch_end(c, st, d_in_len, n, dfl_cap, d_out_len, nullptr);
}

// per-kernel milliseconds of the first workspace chunk of the last deflate batch: chains, best, parse, encode
extern "C" int kmp_deflate_last_kernel_ms(kmp_batch_ctx* c, float* ms4)
{
    if (!c || !ms4 || !c->ev_valid[3]) { g_last_error = "no deflate timing recorded"; return KMP_ERR_ARG; }
    HIP_TRY(hipEventSynchronize(c->ev[12]));
    static const int from[4] = { 8, 9, 13, 11 }, to[4] = { 9, 10, 11, 12 };
    for (int i = 0; i < 4; i++) HIP_TRY(hipEventElapsedTime(&ms4[i], c->ev[from[i]], c->ev[to[i]]));
    return KMP_OK;
}
