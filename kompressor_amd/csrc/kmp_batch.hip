// kmp_batch.hip -- the zstd half of libkompressor_hip.so's batched device API (include/kompressor_hip.h part 2): kernel entry
// points for gfx950, the batch context and its workspace, kmp_zstd_compress_batch* / kmp_zstd_decompress_batch*,
// kmp_compact_batch.  The DEFLATE half is kmp_deflate.hip, the streaming-compatible API kmp_stream.hip.
#include "kx_wave.h"
#include "zstd_match.h"
#ifdef KMP_ABLATIONS
#include "zstd_match2.h"
#endif
#include "zstd_entropy.h"
#include "zstd_match_dict.h"
#include "zstd_lazy.h"
#include "zstd_match_fast.h"
#include "zstd_cdict_host.h"
#include "zstd_decode.h"
#include "zstd_predecode.h"
#include "deflate_match.h"          // (KdBest & co.: the context frees the DEFLATE workspace)
#include "kmp_internal.h"

#include <mutex>
#include <new>
#include <vector>

// --------------------------------------------------------------------------
// kernels (one 64-lane wave per workgroup everywhere)
// --------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(64) void k_zstd_match(KMatchArgs a) { zstd_match_body<G>(a); }
#ifdef KMP_ABLATIONS
// the same parse as a split-phase stage machine (zstd_match2.h): one memory round trip per outer iteration
template <int G, int R>
__global__ __launch_bounds__(64, 4) void k_zstd_match2(KMatchArgs a) { zstd_match2_body<G, R>(a); }
#endif
__global__ __launch_bounds__(64, 4) void k_zstd_entropy(KEntropyArgs a) { zstd_entropy_body(a); }
// zstd levels 5 .. 10 (greedy / lazy / lazy2: zstd_lazy.h)
__global__ __launch_bounds__(256) void k_zstd_lazy_sort(KLazyArgs a) { zstd_lazy_sort_body(a); }
template <int WORDS> __global__ __launch_bounds__(64, 2) void k_zstd_lazy(KLazyArgs a) { zstd_lazy_body<WORDS>(a); }
// ... of a batch parsed against a formatted dictionary: its tables as the block's predecessor, its ID in the frame header
__global__ __launch_bounds__(64, 4) void k_zstd_entropy_prior(KEntropyArgs a) { zstd_entropy_body<true>(a); }
#ifdef KMP_ABLATIONS
// parse and entropy stage in one launch (zstd_entropy.h: zstd_l3_fused_body)
template <int G>
__global__ __launch_bounds__(64, 4) void k_zstd_l3_fused(KMatchArgs a, KEntropyArgs e) { zstd_l3_fused_body<G>(a, e); }
#endif
// levels 1 and 2 (strategy "fast")
template <int G>
__global__ __launch_bounds__(64) void k_zstd_match_fast(KFastArgs a) { zstd_match_fast_body<G>(a); }
// the parse when the context holds a raw-content dictionary
template <int G>
__global__ __launch_bounds__(64) void k_zstd_match_dict(KDictArgs a) { zstd_match_dict_body<G>(a); }
#ifdef KMP_ABLATIONS
// frames of several blocks (slices above 128 KiB): one block of every unfinished slice per launch
template <int G>
__global__ __launch_bounds__(64) void k_zstd_match_blk(KMatchArgs a)
{
    zstd_match_body<G, true>(a);
    a.counter += 1;                          // second work queue: the blocks libzstd parses with the extDict variant
    zstd_match_ext_body<G>(a);
}
__global__ __launch_bounds__(64, 4) void k_zstd_frame(KFrameArgs a) { zstd_frame_body(a); }
#endif
// ... or the whole chain of blocks of a slice by one wave (no host rounds)
template <int G>
__global__ __launch_bounds__(64, 3) void k_zstd_big(KBigArgs a) { zstd_big_body<G>(a); }
template <int G>
__global__ __launch_bounds__(64, 3) void k_zstd_big_fast(KBigArgs a) { zstd_big_body<G, true>(a); }
// first block size, repcodes {1,4,8}, no Huffman table, a fresh window; an empty slice is a header and an empty raw block
// chunked: the input is taken in chunks of 128 KiB (KFrameArgs.stream != 0); wd: window descriptor byte of a streaming frame, 0 = one-shot
// level4: slices of the size classes level 4 runs as "greedy" (up to 16 KiB, 128 - 256 KiB; sizes known) are refused: out_len 0, status bit
__global__ __launch_bounds__(256) void k_zstd_frame_init(const u32* in_len, u32 n, KFrameState* fs, u8* dst, const u64* out_off, u32* out_len, u32* remaining, u32 wd, u32 chunked, u32 level4 = 0, u32* status = nullptr)
{
    u32 const i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    KFrameState s; u32 const len = in_len[i];
    if (level4 && !wd && !kx_l4_served(len)) {
        s.ipos = 0; s.opos = 0; s.blockSize = 0; s.first = 1; s.rep[0] = 1; s.rep[1] = 4; s.rep[2] = 8; s.hufValid = 0; s.hufSel = 0; s.savings = 0;
        s.lowLimit = 2; s.dictLimit = 2; s.bufPos = 0; s.extBase = 0; s.wflags = 0; s.chunkEnd = 0;
        fs[i] = s; out_len[i] = 0; if (status) atomicOr(status, (u32)KMP_STATUS_LEVEL_SIZE);
        return;
    }
    s.ipos = 0; s.opos = 0; s.blockSize = len < KX_BLOCK_MAX ? len : KX_BLOCK_MAX; s.first = 1;
    s.rep[0] = 1; s.rep[1] = 4; s.rep[2] = 8; s.hufValid = 0; s.hufSel = 0; s.savings = 0;
    s.lowLimit = 2; s.dictLimit = 2; s.bufPos = 0; s.extBase = 0; s.wflags = 0;
    s.chunkEnd = (chunked && len > KX_BLOCK_MAX) ? KX_BLOCK_MAX : len;
    fs[i] = s;
    if (len == 0) { u8* d = dst + out_off[i]; kx_st32(d, 0xFD2FB528u); d[4] = wd ? 0x00 : 0x20; d[5] = (u8)wd; d[6] = 1; d[7] = 0; d[8] = 0; out_len[i] = 9; }
    else atomicAdd(remaining, 1u);
}
__global__ __launch_bounds__(64, 5) void k_zstd_decode(KDecodeArgs a) { zstd_decode_body(a); }
// the sequence bitstreams decoded ahead of it, one lane per frame (FSE tables in HBM)
__global__ __launch_bounds__(4 * KXP_FRAMES) void k_zstd_seq_predecode(KPreArgs a) { zstd_seq_predecode_body(a); }
// ... and the Huffman-coded literals, one lane per stream (32 frames per workgroup of 128 threads)
__global__ __launch_bounds__(64) void k_zstd_lit_predecode(KLitArgs a) { zstd_lit_predecode_body(a); }
__global__ __launch_bounds__(256) void k_zstd_seq_count(KSeqSortArgs a) { zstd_seq_count_body(a); }
__global__ __launch_bounds__(256) void k_zstd_seq_rank(KSeqSortArgs a) { zstd_seq_rank_body(a); }
__global__ __launch_bounds__(256) void k_zstd_seq_perm(KSeqSortArgs a) { zstd_seq_perm_body(a); }
// exclusive prefix sum of u32 lengths into u64 offsets, single workgroup
__global__ __launch_bounds__(1024) void k_scan_lengths(const u32* len, u32 n, u64* off)
{
    __shared__ u64 part[1024];
    u32 const t = threadIdx.x;
    u32 const per = (n + 1023) / 1024;
    u32 const b = t * per, e = (b + per < n) ? b + per : n;
    u64 s = 0;
    for (u32 i = b; i < e; i++) s += len[i];
    part[t] = s;
    __syncthreads();
    for (u32 o = 1; o < 1024; o <<= 1) {
        u64 v = (t >= o) ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    u64 run = (t == 0) ? 0 : part[t - 1];
    for (u32 i = b; i < e; i++) { off[i] = run; run += len[i]; }
    if (t == 1023) off[n] = part[1023];
}
// one wave per frame, 16 B per lane per step when both sides allow it
__global__ __launch_bounds__(64) void k_compact(const u8* src, const u64* in_off, const u32* len, u32 n, u8* dst, const u64* out_off)
{
    int const lane = threadIdx.x;
    for (u32 it = blockIdx.x; it < n; it += gridDim.x) {
        u32 const i = kx_xcd_chunk(it, n);                  // (frame sizes follow the content's period too: zstd_common.h)
        const u8* s = src + in_off[i]; u8* d = dst + out_off[i]; u32 const L = len[i];
        u32 k = (u32)lane * 8u;
        for (; k + 8 <= L; k += 512u) kx_st64(d + k, kx_ld64(s + k));
        u32 const tail = L & ~7u;
        if (lane < (int)(L - tail)) d[tail + lane] = s[tail + lane];
    }
}

// Per-slice lengths live on the device, so the host cannot check them against the workspace before it launches.
// Every batch entry point therefore runs its kernels on a sanitised copy: a slice longer than the context (or the
// codec) holds is processed as an empty one, then loses its frame again (out_len = 0: a real frame is never empty) and
// raises the context's status word, which kmp_batch_status hands to the host.
__global__ __launch_bounds__(256) void k_len_guard(const u32* in_len, u32 n, u32 cap, u32* len_ok, u32* status)
{
    u32 const i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    u32 const len = in_len[i];
    len_ok[i] = len > cap ? 0u : len;
    if (len > cap) atomicOr(status, (u32)KMP_STATUS_SLICE_TOO_LARGE);
}
__global__ void k_status_take(u32* status, u32* out) { *out = atomicExch(status, 0u); }
// meta: the one-block zstd parsers' per-slice record (a tripped loop guard there also voids the frame), or null
__global__ __launch_bounds__(256) void k_len_guard_finish(const u32* in_len, u32 n, u32 cap, u32* out_len, const KSliceMeta* meta, u32* status)
{
    u32 const i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    if (in_len[i] > cap) out_len[i] = 0;
    else if (meta && meta[i].status == 3u) { out_len[i] = 0; atomicOr(status, (u32)KMP_STATUS_LEVEL_SIZE); }
    else if (meta && in_len[i] >= 8 && meta[i].status != 0) { out_len[i] = 0; atomicOr(status, (u32)KMP_STATUS_KERNEL_GUARD); }
}

// --------------------------------------------------------------------------
// errors
// --------------------------------------------------------------------------
thread_local std::string g_last_error;
int hip_fail(hipError_t e, const char* what)
{
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return KMP_ERR_HIP;
}

extern "C" const char* kmp_last_error(void) { return g_last_error.c_str(); }

// Random 4-byte read-modify-writes over a region, the way the level-3 parser touches its tables: how fast is THIS piece
// of HBM?  (The rate differs by a sixth between regions of one device: DESIGN.md section 5a; tools/region_probe.py.)
__global__ __launch_bounds__(256) void k_region_probe(u32* p, u64 words, u32 iters, u32 salt)
{
    u64 x = ((u64)blockIdx.x * 256u + threadIdx.x) * 0x9E3779B97F4A7C15ull + salt;
    u32 acc = 0;
    for (u32 i = 0; i < iters; i++) {
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        u64 const w = x % words;
        u32 const v = kx_ld_nt(p + w);
        acc += v;
        kx_st_nt(p + ((w * 2654435761ull + 12345u) % words), v + i);
    }
    if (acc == 0x12345678u) p[0] = acc;
}
// milliseconds for blocks x 256 threads x iters random read + write pairs over [p, p + bytes)
extern "C" int kmp_debug_probe_region(void* p, size_t bytes, uint32_t blocks, uint32_t iters, float* ms, void* hip_stream)
{
    if (!p || bytes < 4096 || !ms) { g_last_error = "kmp_debug_probe_region: bad argument"; return KMP_ERR_ARG; }
    hipStream_t const st = (hipStream_t)hip_stream;
    hipEvent_t e0, e1; HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_region_probe, dim3(blocks), dim3(256), 0, st, (u32*)p, (u64)(bytes / 4), 8u, 1u);       // warm
    HIP_TRY(hipEventRecord(e0, st));
    hipLaunchKernelGGL(k_region_probe, dim3(blocks), dim3(256), 0, st, (u32*)p, (u64)(bytes / 4), iters, 7u);
    HIP_TRY(hipEventRecord(e1, st));
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipEventElapsedTime(ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return KMP_OK;
}
// What the memory system gives the level-3 parser's tables WHERE THEY LIE: random accesses over the context's table pieces,
// four independent chains per lane, 16 waves per CU.  mode 0: 4-byte loads (a probe that inserts nothing); mode 1: a load
// and a store into the same word (a probe + insert: the line comes in and goes back).  Run once when a context with large
// tables is created, before the tables are zeroed; bench.py prices the parser's measured memory requests with the two rates
// (DESIGN.md section 4.1).
__global__ __launch_bounds__(64) void k_table_probe(u32* p0, u32* p1, u32* p2, u32* p3, u32 pieces, u64 words, u32 iters, u32 mode, u32* sink)
{
    u64 s[4]; u32 acc = 0;
    u64 const gid = (u64)blockIdx.x * 64 + threadIdx.x;
    for (int j = 0; j < 4; j++) s[j] = (gid * 4 + j) * 0x9E3779B97F4A7C15ull + 12345;
    for (u32 i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            s[j] = s[j] * 6364136223846793005ull + 1442695040888963407ull;
            u64 const r = s[j] >> 11;
            u32 const pc = pieces == 4 ? (u32)(r & 3) : 0u;
            u32* const base = pc == 0 ? p0 : pc == 1 ? p1 : pc == 2 ? p2 : p3;
            u32* const q = base + __umul64hi(r << 11, words);
            u32 v = *q;
            if (mode) kx_st_nt(q, v + 1);
            acc += v; s[j] += v & 1;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
extern "C" const char* kmp_version(void) { return "kompressor_hip 0.4 (gfx950; zstd levels -131072 .. -1 and 1 .. 3: frames and streams up to 1 GiB, dictionaries (raw content and zstd format); level 4 up to 128 KiB, above 256 KiB and streams; levels 5 .. 10 up to 128 KiB; deflate / zlib / gzip levels 1-9, windowBits 9-15, memLevel 1-9; zstd and inflate decoders)"; }

u32 env_u32(const char* name, u32 dflt)
{
    const char* v = getenv(name);
    return (v && *v) ? (u32)strtoul(v, nullptr, 10) : dflt;
}

// Where a randomly accessed table lands matters a little: read + insert pairs confined to a few dozen GiB of this device's HBM
// run at 20 G/s, over a span of 72 GiB or more at 26 (tools/spanprobe, profiles/r03_match_floor.txt).  The level-3 team tables
// get their span from the context's arena (kmp_batch_create); for the other large tables a placement by trial is available
// as an option: with KMP_PLACE_TRIES = 2 .. 8 that many allocations are probed with the tables' access pattern
// (k_region_probe, ~30 ms each) until one is of the fast kind, the fastest stays, the others are freed.  The default is one
// plain allocation: the candidates of a trial are held until the choice is made, and a library must not take that much
// memory behind its caller's back.
static int place_alloc(u32** out, size_t bytes, float* kept_ms, u32* tried_out)
{
    u32 tries = KMP_KNOB("KMP_PLACE_TRIES", 1); if (tries < 1) tries = 1; if (tries > 8) tries = 8;
    if (bytes < ((size_t)4 << 30)) tries = 1;
    u32* cand[8] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr }; float ms[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    u32 got = 0, best = 0;
    for (u32 t = 0; t < tries; t++) {
        if (hipMalloc((void**)&cand[t], bytes) != hipSuccess) { (void)hipGetLastError(); cand[t] = nullptr; break; }
        got = t + 1;
        if (tries > 1 && kmp_debug_probe_region(cand[t], bytes, 4096u, 384u, &ms[t], nullptr) != KMP_OK) ms[t] = 1e30f;
        if (ms[t] < ms[best]) best = t;
        // 805 M accesses: the fast kind of region takes 23.5 ms (34 G/s), the slow kind 29.2 (27.6 G/s) on an MI355X
        if (ms[t] <= 26.0f) break;
    }
    if (got == 0) return hip_fail(hipErrorOutOfMemory, "hipMalloc(parser tables)");
    if (env_u32("KMP_VERBOSE", 0)) { fprintf(stderr, "place_alloc %zu MiB:", bytes >> 20); for (u32 t = 0; t < got; t++) fprintf(stderr, " %.1f ms%s", ms[t], t == best ? "*" : ""); fprintf(stderr, "\n"); }
    for (u32 t = 0; t < got; t++) if (t != best) (void)hipFree(cand[t]);
    *out = cand[best];
    if (kept_ms) *kept_ms = ms[best];
    if (tried_out) *tried_out = got;
    return KMP_OK;
}

struct create_opts { int span_gib; int retry; };          // -1 = the environment's / the default
static int batch_create_body(kmp_batch_ctx* c, int device, uint32_t max_slices, uint32_t max_slice_bytes, int team_lanes, create_opts const& o);
extern "C" void kmp_batch_destroy(kmp_batch_ctx* c);
static int batch_create_with(kmp_batch_ctx** out, int device, uint32_t max_slices, uint32_t max_slice_bytes, int team_lanes, create_opts const& o)
{
    if (!out || max_slices == 0) { g_last_error = "kmp_batch_create: bad argument"; return KMP_ERR_ARG; }
    if (max_slice_bytes > KMP_MAX_BIG_SLICE_BYTES) { g_last_error = "kmp_batch_create: slices above 1 GiB are not supported"; return KMP_ERR_CAPACITY; }
    int const team_fixed = (team_lanes != 0 || getenv("KMP_TEAM_LANES")) ? 1 : 0;
    if (team_lanes == 0) team_lanes = (int)env_u32("KMP_TEAM_LANES", 4);
    if (team_lanes != 2 && team_lanes != 4 && team_lanes != 8 && team_lanes != 16 && team_lanes != 32 && team_lanes != 64) { g_last_error = "team_lanes must be 2, 4, 8, 16, 32 or 64"; return KMP_ERR_ARG; }
    HIP_TRY(hipSetDevice(device));
    kmp_batch_ctx* c = new (std::nothrow) kmp_batch_ctx();
    if (!c) { g_last_error = "out of host memory"; return KMP_ERR_ARG; }
    memset(c, 0, sizeof(*c));
    c->team_fixed = team_fixed;
    int const rc = batch_create_body(c, device, max_slices, max_slice_bytes, team_lanes, o);
    if (rc != KMP_OK) { std::string const keep = g_last_error; kmp_batch_destroy(c); g_last_error = keep; return rc; }      // nothing half-built is left behind
    *out = c;
    return KMP_OK;
}
extern "C" int kmp_batch_create(kmp_batch_ctx** out, int device, uint32_t max_slices, uint32_t max_slice_bytes, int team_lanes)
{
    create_opts const o = { -1, -1 };
    return batch_create_with(out, device, max_slices, max_slice_bytes, team_lanes, o);
}
extern "C" int kmp_batch_create_ex(kmp_batch_ctx** out, int device, uint32_t max_slices, uint32_t max_slice_bytes, const kmp_batch_options* opts)
{
    if (!opts || opts->struct_bytes < sizeof(kmp_batch_options)) { g_last_error = "kmp_batch_create_ex: options missing or of another version"; return KMP_ERR_ARG; }
    create_opts const o = { opts->table_span_gib, opts->table_retry };
    return batch_create_with(out, device, max_slices, max_slice_bytes, opts->team_lanes, o);
}
// a context whose arena is packed (the engines of the host-memory batch: kmp_coalesce.h)
int batch_create_packed(kmp_batch_ctx** out, int device, uint32_t max_slices, uint32_t max_slice_bytes)
{
    create_opts const o = { 0, 0 };
    return batch_create_with(out, device, max_slices, max_slice_bytes, 0, o);
}
static int batch_create_body(kmp_batch_ctx* c, int device, uint32_t max_slices, uint32_t max_slice_bytes, int team_lanes, create_opts const& o)
{
    c->device = device; c->max_slices = max_slices; c->max_slice_bytes = max_slice_bytes < 64 ? 64 : max_slice_bytes; c->G = team_lanes;
    hipDeviceProp_t prop; HIP_TRY(hipGetDeviceProperties(&prop, device));
    // the context's own stream: the second stream of its pipelines, and where creation's probes and clears run (non-blocking:
    // nothing here serialises with the caller's streams, the NULL stream included)
    HIP_TRY(hipStreamCreateWithFlags(&c->st2, hipStreamNonBlocking));
    u32 const waves_per_cu = KMP_KNOB("KMP_MATCH_WAVES_PER_CU", 16);     // 16 x 16 teams x 256 CUs = 65 536 slices in flight at team width 4
    u32 const teams_per_wave = 64 / (u32)team_lanes;
    u32 blocks = (u32)prop.multiProcessorCount * waves_per_cu;
    u32 const need = (max_slices + teams_per_wave - 1) / teams_per_wave;
    if (blocks > need) blocks = need;
    c->match_blocks = blocks; c->nteams = blocks * teams_per_wave;
    // Every parser wants every slice in flight: 16 waves per CU.  (The level-3 parser ran with 12 for most of two rounds --
    // 49 152 team slots, a 65 536-slice batch in two launches with the entropy kernel of the first beside the second: 251 ms.
    // That optimum belonged to an entropy kernel that was a third slower than it had to be (zstd_common.h kx_xcd_chunk);
    // with it fixed, one launch of each kernel at 16 waves per CU takes 191 + 27 ms.)
    { u32 const l3 = (u32)prop.multiProcessorCount * KMP_KNOB("KMP_MATCH_WAVES_PER_CU_L3", 16); c->match_blocks_l3 = l3 < blocks ? l3 : blocks; c->l3_team_slots = l3 * teams_per_wave; }
    c->big = c->max_slice_bytes > KMP_MAX_SLICE_BYTES;
    u32 const block_cap = c->big ? KMP_MAX_SLICE_BYTES : c->max_slice_bytes;     // the sequence / literal workspaces hold one block
    c->seq_cap = (block_cap / 4 + 8 + 15) & ~15u; c->lit_cap = block_cap + 64; c->scratch_words = block_cap / 4 + 64;
    size_t const ns = max_slices;
    if (c->big) {
        c->big_G = (int)KMP_KNOB("KMP_BIG_TEAM_LANES", 0);      // 0 = by batch size (zstd_compress_big)
        if (c->big_G != 0 && c->big_G != 2 && c->big_G != 4 && c->big_G != 8 && c->big_G != 16 && c->big_G != 32 && c->big_G != 64) c->big_G = 0;
        HIP_TRY(hipMalloc((void**)&c->fstate, ns * sizeof(KFrameState)));
        HIP_TRY(hipMalloc((void**)&c->hufct, ns * 512 * sizeof(u32)));
        { int const rc = place_alloc(&c->big_tables, ns * KX_BIG_TBL_ENTRIES * sizeof(u32), nullptr, nullptr); if (rc != KMP_OK) return rc; }     // (the block-chain parser's tables: the same access pattern)
        HIP_TRY(hipMalloc((void**)&c->remaining, 64));
        HIP_TRY(hipMalloc((void**)&c->big_counters, ns * 4));
    }
    {
        // The workspace of a context with large team tables is ONE allocation (an arena): the tables in four pieces (team t:
        // piece t & 3) with the sequence, literal and staging buffers between them, laid out over a span of
        // KMP_TABLE_SPAN_GIB / kmp_batch_options.table_span_gib (default 100) GiB.  Why a span: read + insert pairs run a quarter
        // faster when they straddle the coarse blocks this device's HBM is laid out in -- 20 G pairs/s inside any 24 .. 36 GiB,
        // 25 - 26 from 72 GiB of span on (tools/spanprobe, profiles/r03_match_floor.txt) -- and the parser lives on that rate:
        // the same context takes 205 ms per batch with its 41 GiB packed, 183 - 190 over 100 - 140 GiB on every box tried.
        // The gaps between the parts are memory the context holds and does not use (59 GiB of a 65 536-slice context's 100), so
        // the span is BOUNDED: never more than half of what the device has free when the context is created (a second context,
        // the caller's own buffers and the host engines must still fit: ADVICE r3); below the parts' own size the arena is
        // packed.  Span 0 packs it (memory over ~11 % of parser time).  kmp_batch_memory reports what the context holds.
        // A second arena is tried only on request (KMP_TABLE_RETRY=1 / table_retry: when the first measures slow whatever the
        // layout -- some allocations land on slower memory, 212 ms instead of 189 -- and the device has room for both; one of
        // the two is freed before creation returns).  That is the one transient allocation creation can make, hence opt-in;
        // bench.py asks for it and says so in its line.
        size_t const tbytes = (size_t)c->nteams * KX_TBL_ENTRIES * sizeof(u32);
        size_t const A = (size_t)2 << 20;                                // every part starts on a 2 MiB boundary
        auto up = [&](size_t v) { return (v + A - 1) & ~(A - 1); };
        size_t const seqs_b = up(ns * c->seq_cap * sizeof(KSeq)), lits_b = up(ns * c->lit_cap), meta_b = up(ns * sizeof(KSliceMeta)), scr_b = up(ns * c->scratch_words * sizeof(u32));
        c->tseg_n = 1;
        if (KMP_KNOB("KMP_TABLE_ARENA", 1) && tbytes >= ((size_t)4 << 30) && (c->nteams & 3u) == 0) {
            size_t const piece = up(tbytes / 4);
            size_t const need = 4 * piece + seqs_b + lits_b + meta_b + scr_b;
            size_t want = (size_t)(o.span_gib >= 0 ? (u32)o.span_gib : env_u32("KMP_TABLE_SPAN_GIB", 100)) << 30;
            size_t fr = 0, tot = 0;
            if (hipMemGetInfo(&fr, &tot) != hipSuccess) fr = 0;
            if (want > fr / 2) want = fr / 2;                               // bounded: at most half of the free memory
            if (want > need && fr < want + ((size_t)16 << 30)) want = 0;    // not that much room: pack
            size_t const gap = want > need ? ((want - need) / 3) & ~(A - 1) : 0;     // unused bytes behind each of the three buffers between the pieces (rounded down: the span is a bound)
            size_t const total = need + 3 * gap;
            if (fr > total + ((size_t)1 << 30) && hipMalloc((void**)&c->arena, total) == hipSuccess) {
                c->arena_bytes = total;
                // Where in the arena the four pieces go is chosen by measurement, inside the arena (nothing else is allocated):
                // up to eight layouts are probed with the tables' own traffic (k_table_probe, read + insert pairs, ~12 ms each) and
                // the fastest stays.  Which physical blocks of the HBM an offset of the arena falls into differs from process to
                // process: with one fixed layout the same box gave 21.6 and 24.9 G pairs/s (parser 213 and 196 ms) in two runs
                // a minute apart.  KMP_TABLE_LAYOUT = 1 .. 8 fixes a layout (1: two pieces at either end; 2: evenly spread;
                // 3: the ends moved inwards by an eighth; 4: the second and third quarter; 5 .. 8: below).  The other buffers fill the space
                // the pieces leave.
                size_t const slackT = total - 4 * piece;                                   // bytes that are not table
                size_t lay[8][4]; u32 nlay = 0;
                // (a layout counts only if its pieces do not overlap and the other buffers fit into what it leaves free)
                auto fits = [&](const size_t* o, KSeq** ps, u8** pl, KSliceMeta** pm, u32** pc) -> bool {
                    size_t fs_[5], fe_[5]; u32 nf = 0; size_t cur = 0;
                    for (int i = 0; i < 4; i++) { if (o[i] < cur || o[i] + piece > total) return false; if (o[i] > cur) { fs_[nf] = cur; fe_[nf] = o[i]; nf++; } cur = o[i] + piece; }
                    if (cur < total) { fs_[nf] = cur; fe_[nf] = total; nf++; }
                    auto take = [&](size_t bytes) -> u8* { for (u32 f = 0; f < nf; f++) if (fe_[f] - fs_[f] >= bytes) { u8* const r = c->arena + fs_[f]; fs_[f] += bytes; return r; } return nullptr; };
                    u8* const a0 = take(seqs_b); u8* const a1 = take(lits_b); u8* const a2 = take(meta_b); u8* const a3 = take(scr_b);
                    if (!a0 || !a1 || !a2 || !a3) return false;
                    if (ps) { *ps = (KSeq*)a0; *pl = a1; *pm = (KSliceMeta*)a2; *pc = (u32*)a3; }
                    return true;
                };
                auto add = [&](size_t o0, size_t o1, size_t o2, size_t o3) {
                    size_t const o[4] = { o0 & ~(A - 1), o1 & ~(A - 1), o2 & ~(A - 1), o3 & ~(A - 1) };
                    if (!fits(o, nullptr, nullptr, nullptr, nullptr)) return;
                    for (int i = 0; i < 4; i++) lay[nlay][i] = o[i];
                    nlay++;
                };
                add(0, piece, total - 2 * piece, total - piece);
                if (gap) {
                    add(0, up(slackT / 3) + piece, 2 * up(slackT / 3) + 2 * piece, total - piece);
                    add(up(total / 8), up(total / 8) + piece, total - 2 * piece - up(total / 8), total - piece - up(total / 8));
                    add(up(total / 4), up(total / 4) + piece, total - 2 * piece - up(total / 4), total - piece - up(total / 4));
                    add(0, up(total / 2) - piece, up(total / 2), total - piece);                                    // 5: the ends and the middle
                    add(up(total / 16), up(total / 16) + piece, total - 2 * piece - up(total / 16), total - piece - up(total / 16));
                    add(up(total / 16 * 3), up(total / 16 * 3) + piece, total - 2 * piece - up(total / 16 * 3), total - piece - up(total / 16 * 3));
                    add(up(total / 16), up(total / 16 * 5), up(total / 16 * 9), up(total / 16 * 13));               // 8: one piece in every quarter
                }
                u32 pick = 0;
                u32 const fixed = KMP_KNOB("KMP_TABLE_LAYOUT", 0);
                // probes every layout on the arena at `base`: the fastest one (priced with the parser's own mix) and its pair rate
                auto probe_arena = [&](u8* base, u32* best_l, float* best_ms, float* best_pairs) -> int {
                    hipEvent_t e[3] = { nullptr, nullptr, nullptr };
                    for (int i = 0; i < 3; i++) if (hipEventCreate(&e[i]) != hipSuccess) { for (int j = 0; j < i; j++) (void)hipEventDestroy(e[j]); g_last_error = "kmp_batch_create: hipEventCreate failed"; return KMP_ERR_HIP; }
                    u64 const words = (u64)(piece / 4); u32 const pb = (u32)prop.multiProcessorCount * 16u;
                    int rc = KMP_OK; *best_ms = 1e30f; *best_l = 0; *best_pairs = 0;
                    for (u32 l = 0; l < nlay && rc == KMP_OK; l++) {
                        u32* t[4]; for (int i = 0; i < 4; i++) t[i] = (u32*)(base + lay[l][i]);
                        hipLaunchKernelGGL(k_table_probe, dim3(pb), dim3(64), 0, c->st2, t[0], t[1], t[2], t[3], 4u, words, 4u, 1u, t[0]);       // warm (TLB)
                        bool ok = hipEventRecord(e[0], c->st2) == hipSuccess;
                        hipLaunchKernelGGL(k_table_probe, dim3(pb), dim3(64), 0, c->st2, t[0], t[1], t[2], t[3], 4u, words, 96u, 1u, t[0]);
                        ok = ok && hipEventRecord(e[1], c->st2) == hipSuccess;
                        hipLaunchKernelGGL(k_table_probe, dim3(pb), dim3(64), 0, c->st2, t[0], t[1], t[2], t[3], 4u, words, 96u, 0u, t[0]);
                        ok = ok && hipEventRecord(e[2], c->st2) == hipSuccess && hipEventSynchronize(e[2]) == hipSuccess;
                        float msp = 0, msr = 0;
                        ok = ok && hipEventElapsedTime(&msp, e[0], e[1]) == hipSuccess && hipEventElapsedTime(&msr, e[1], e[2]) == hipSuccess;
                        if (!ok) { g_last_error = "kmp_batch_create: probing the arena failed"; rc = KMP_ERR_HIP; break; }
                        // the parser's own mix: per batch 3.97 G probe + insert pairs and 1.80 G reads that insert nothing (profiles/pmc_latest.json)
                        float const ms = 3.97f * msp + 1.80f * msr;
                        float const pairs = (float)((double)pb * 64.0 * 96.0 * 4.0 / (msp * 1e-3));
                        if (env_u32("KMP_VERBOSE", 0)) fprintf(stderr, "arena %p layout %u: %.1f G pairs/s, %.1f G reads/s\n", (void*)base, l + 1, pairs / 1e9, (double)pb * 64.0 * 96.0 * 4.0 / (msr * 1e-3) / 1e9);
                        if (ms < *best_ms) { *best_ms = ms; *best_l = l; *best_pairs = pairs; }
                    }
                    for (int i = 0; i < 3; i++) (void)hipEventDestroy(e[i]);
                    return rc;
                };
                if (fixed >= 1 && fixed <= nlay) pick = fixed - 1;
                else if (nlay > 1) {
                    float ms1 = 0, pr1 = 0;
                    { int const rc = probe_arena(c->arena, &pick, &ms1, &pr1); if (rc != KMP_OK) return rc; }
                    // Some allocations are slow whatever the layout (one box: every second process, 21.7 G pairs/s at best against
                    // 25; the parser then takes 212 ms instead of 189 -- which physical memory the allocator hands out is not ours to
                    // choose).  On request (table_retry / KMP_TABLE_RETRY = 1), when that happens and the device has room for a second
                    // arena beside the first, ONE more is allocated and probed, the faster of the two stays and the other is freed
                    // before creation returns (0: never, the default; 2: always, for the tests).
                    size_t fr2 = 0, tot2 = 0;
                    u32 const retry = o.retry >= 0 ? (u32)o.retry : env_u32("KMP_TABLE_RETRY", 0);     // 0 never (the default), 1 when slow, 2 always (tests)
                    if ((pr1 < (float)KMP_KNOB("KMP_TABLE_RETRY_BELOW", 235) * 1e8f || retry == 2) && retry && hipMemGetInfo(&fr2, &tot2) == hipSuccess && fr2 > total + ((size_t)16 << 30)) {
                        u8* second = nullptr;
                        if (hipMalloc((void**)&second, total) == hipSuccess) {
                            u32 pick2 = 0; float ms2 = 0, pr2 = 0;
                            int const rc = probe_arena(second, &pick2, &ms2, &pr2);
                            if (rc != KMP_OK) { (void)hipFree(second); return rc; }
                            if (env_u32("KMP_VERBOSE", 0)) fprintf(stderr, "arena retry: %.1f -> %.1f G pairs/s\n", pr1 / 1e9, pr2 / 1e9);
                            if (ms2 < ms1 * 0.98f) { (void)hipFree(c->arena); c->arena = second; pick = pick2; c->table_retry = 2; }
                            else { (void)hipFree(second); c->table_retry = 1; }
                        } else (void)hipGetLastError();
                    }
                }
                c->table_layout = pick + 1;
                for (int i = 0; i < 4; i++) c->tseg[i] = (u32*)(c->arena + lay[pick][i]);
                // the other buffers: first fit into what the pieces leave free
                if (nlay == 0 || !fits(lay[pick], &c->seqs, &c->lits, &c->meta, &c->scratch)) { g_last_error = "kmp_batch_create: arena layout failed"; return KMP_ERR_ARG; }
                c->tseg_n = 4; c->tables = c->tseg[0];
            } else { (void)hipGetLastError(); c->arena = nullptr; }
        }
        if (!c->arena) {
            HIP_TRY(hipMalloc((void**)&c->seqs, ns * c->seq_cap * sizeof(KSeq)));
            HIP_TRY(hipMalloc((void**)&c->lits, ns * c->lit_cap));
            HIP_TRY(hipMalloc((void**)&c->meta, ns * sizeof(KSliceMeta)));
            HIP_TRY(hipMalloc((void**)&c->scratch, ns * c->scratch_words * sizeof(u32)));
            int const rc = place_alloc(&c->tables, tbytes, &c->place_ms, &c->place_tried); if (rc != KMP_OK) return rc;
            c->tseg[0] = c->tables;
        }
    }
    HIP_TRY(hipMalloc((void**)&c->team_epoch, (size_t)c->nteams * sizeof(u32)));
    HIP_TRY(hipMalloc((void**)&c->counter, 64));
    if ((size_t)c->nteams * KX_TBL_ENTRIES * sizeof(u32) >= ((size_t)4 << 30) && KMP_KNOB("KMP_TABLE_PROBE", 1)) {
        // ~40 ms: the two rates the parser lives on, measured on these very tables (they are zeroed right below)
        hipEvent_t e[3]; for (int i = 0; i < 3; i++) HIP_TRY(hipEventCreate(&e[i]));
        u64 const words = (u64)c->nteams * KX_TBL_ENTRIES / c->tseg_n; u32 const blocks = (u32)prop.multiProcessorCount * 16u, iters = 96u;
        u32* const t1 = c->tseg_n == 4 ? c->tseg[1] : c->tseg[0]; u32* const t2 = c->tseg_n == 4 ? c->tseg[2] : c->tseg[0]; u32* const t3 = c->tseg_n == 4 ? c->tseg[3] : c->tseg[0];
        hipLaunchKernelGGL(k_table_probe, dim3(blocks), dim3(64), 0, c->st2, c->tseg[0], t1, t2, t3, c->tseg_n, words, 8u, 1u, c->counter);     // warm (TLB)
        HIP_TRY(hipEventRecord(e[0], c->st2));
        hipLaunchKernelGGL(k_table_probe, dim3(blocks), dim3(64), 0, c->st2, c->tseg[0], t1, t2, t3, c->tseg_n, words, iters, 0u, c->counter);
        HIP_TRY(hipEventRecord(e[1], c->st2));
        hipLaunchKernelGGL(k_table_probe, dim3(blocks), dim3(64), 0, c->st2, c->tseg[0], t1, t2, t3, c->tseg_n, words, iters, 1u, c->counter);
        HIP_TRY(hipEventRecord(e[2], c->st2));
        HIP_TRY(hipEventSynchronize(e[2]));
        float ms0 = 0, ms1 = 0; HIP_TRY(hipEventElapsedTime(&ms0, e[0], e[1])); HIP_TRY(hipEventElapsedTime(&ms1, e[1], e[2]));
        double const ops = (double)blocks * 64.0 * iters * 4.0;
        if (ms0 > 0) c->table_reads_per_s = (float)(ops / (ms0 * 1e-3));
        if (ms1 > 0) c->table_pairs_per_s = (float)(ops / (ms1 * 1e-3));
        for (int i = 0; i < 3; i++) (void)hipEventDestroy(e[i]);
    }
    for (u32 i = 0; i < c->tseg_n; i++) HIP_TRY(hipMemsetAsync(c->tseg[i], 0, (size_t)c->nteams * KX_TBL_ENTRIES * sizeof(u32) / c->tseg_n, c->st2));
    HIP_TRY(hipMemsetAsync(c->team_epoch, 0, (size_t)c->nteams * sizeof(u32), c->st2));
    HIP_TRY(hipMemsetAsync(c->meta, 0, ns * sizeof(KSliceMeta), c->st2));
    for (int i = 0; i < 14; i++) HIP_TRY(hipEventCreate(&c->ev[i]));
    for (int i = 0; i < KMP_MAX_CHUNKS; i++) for (int j = 0; j < 2; j++) { HIP_TRY(hipEventCreate(&c->evm[i][j])); HIP_TRY(hipEventCreate(&c->eve[i][j])); }
    HIP_TRY(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    HIP_TRY(hipEventCreate(&c->tune_ev[0])); HIP_TRY(hipEventCreate(&c->tune_ev[1]));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_last_match, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
    for (int i = 0; i < KMP_MAX_PIECES; i++) HIP_TRY(hipEventCreateWithFlags(&c->ev_piece[i], hipEventDisableTiming));
    for (int i = 0; i <= KMP_MAX_CHUNKS; i++) HIP_TRY(hipEventCreateWithFlags(&c->ev_pre[i], hipEventDisableTiming));
    c->cus = (u32)prop.multiProcessorCount;
    HIP_TRY(hipMalloc((void**)&c->len_ok, ns * sizeof(u32)));
    HIP_TRY(hipMalloc((void**)&c->d_status, 64));
    HIP_TRY(hipMemsetAsync(c->d_status, 0, 64, c->st2));
    c->knob.chunks = KMP_KNOB("KMP_ZSTD_CHUNKS", 0); c->knob.match_flags = KMP_KNOB("KMP_MATCH_FLAGS", 6); c->knob.entropy_pad = KMP_KNOB("KMP_ENTROPY_PAD_LDS", 0);
    c->knob.first_permille = KMP_KNOB("KMP_ZSTD_FIRST_PERMILLE", 500); c->knob.fast_first_permille = KMP_KNOB("KMP_ZSTD_FAST_FIRST_PERMILLE", 550); c->knob.entropy_flags = KMP_KNOB("KMP_ENTROPY_FLAGS", 0);
    c->knob.decode_flags = KMP_KNOB("KMP_DECODE_FLAGS", 0); c->knob.decode_pad = KMP_KNOB("KMP_DECODE_PAD_LDS", 0);
    c->knob.big_rounds = KMP_KNOB("KMP_BIG_ROUNDS", 0); c->knob.big_spw = KMP_KNOB("KMP_BIG_SLICES_PER_WAVE", 0);
    c->knob.dfl_chunk = env_u32("KMP_DEFLATE_CHUNK", 16384u); c->knob.dfl_chain_waves = KMP_KNOB("KMP_DEFLATE_CHAIN_WAVES", 4);
    c->knob.dfl_serial = KMP_KNOB("KMP_DEFLATE_SERIAL", 0); c->knob.dfl_flags = KMP_KNOB("KMP_DEFLATE_FLAGS", 0);
    // the decoder's pre-decode kernels (on by default for batches of KMP_PRE_MIN_BATCH = 256 entries or more, DESIGN.md section
    // 4.3): bit 0 = sequences decoded ahead of k_zstd_decode (k_zstd_seq_predecode, one lane per frame), bit 1 = literals
    // (k_zstd_lit_predecode, one lane per stream)
    c->knob.decode_pre = KMP_KNOB("KMP_DECODE_PRE", 3); c->knob.decode_sort = KMP_KNOB("KMP_DECODE_SORT", 1); c->knob.decode_pieces = KMP_KNOB("KMP_DECODE_PIECES", 1); c->knob.decode_stage_slices = env_u32("KMP_DECODE_STAGE_SLICES", 0); c->knob.inflate_pre = KMP_KNOB("KMP_INFLATE_PRE", 1); c->knob.inflate_pieces = KMP_KNOB("KMP_INFLATE_PIECES", 1); c->knob.autotune = KMP_KNOB("KMP_ZSTD_AUTOTUNE", 0);      // opt-in: one launch or two chunks, tried once each (two blocking event reads on the 2nd / 3rd batch)
    c->knob.fuse = KMP_KNOB("KMP_FUSE", 0);                             // 1: k_zstd_l3_fused (the entropy stage inside the parse kernel's waves)
    c->knob.match_v2 = KMP_KNOB("KMP_MATCH_V2", 0);                     // 0: zstd_match.h (the default: 4 % faster, both sit on the same memory floor); 1: zstd_match2.h; 2: with a 512-byte window at team width 4
    HIP_TRY(hipStreamSynchronize(c->st2));
    return KMP_OK;
}

extern "C" void kmp_batch_destroy(kmp_batch_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->arena) (void)hipFree(c->arena);                                        // (holds seqs, lits, meta, scratch and the table pieces)
    else { (void)hipFree(c->seqs); (void)hipFree(c->lits); (void)hipFree(c->meta); (void)hipFree(c->scratch); (void)hipFree(c->tables); }
    (void)hipFree(c->tables_flat); (void)hipFree(c->team_epoch_flat);
    (void)hipFree(c->team_epoch); (void)hipFree(c->tables4); (void)hipFree(c->epoch4); (void)hipFree(c->big_tables4);
    (void)hipFree(c->d_dict); (void)hipFree(c->d_dictL); (void)hipFree(c->d_dictS); (void)hipFree(c->d_prior); (void)hipFree(c->d_dprior);
    (void)hipFree(c->lz_srt); (void)hipFree(c->lz_wr); (void)hipFree(c->lz_order);
    (void)hipFree(c->fstate); (void)hipFree(c->hufct); (void)hipFree(c->big_tables); (void)hipFree(c->remaining); (void)hipFree(c->big_counters); (void)hipFree(c->counter);
    for (int i = 0; i < 14; i++) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    for (int i = 0; i < KMP_MAX_CHUNKS; i++) for (int j = 0; j < 2; j++) { if (c->evm[i][j]) (void)hipEventDestroy(c->evm[i][j]); if (c->eve[i][j]) (void)hipEventDestroy(c->eve[i][j]); }
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    for (int i = 0; i < 2; i++) if (c->tune_ev[i]) (void)hipEventDestroy(c->tune_ev[i]);
    if (c->ev_last_match) (void)hipEventDestroy(c->ev_last_match);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    for (int i = 0; i < KMP_MAX_PIECES; i++) if (c->ev_piece[i]) (void)hipEventDestroy(c->ev_piece[i]);
    for (int i = 0; i <= KMP_MAX_CHUNKS; i++) if (c->ev_pre[i]) (void)hipEventDestroy(c->ev_pre[i]);
    (void)hipFree(c->len_ok); (void)hipFree(c->d_status);
    (void)hipFree(c->pre_stage); (void)hipFree(c->pre_blk); (void)hipFree(c->pre_nblk); (void)hipFree(c->pre_sort);
    (void)hipFree(c->pre_lits); (void)hipFree(c->pre_lit); (void)hipFree(c->pre_nlit);
    if (c->st2) (void)hipStreamDestroy(c->st2);
    (void)hipFree(c->dfl_wr); (void)hipFree(c->dfl_order); (void)hipFree(c->dfl_link); (void)hipFree(c->dfl_best); (void)hipFree(c->dfl_syms); (void)hipFree(c->dfl_meta); (void)hipFree(c->dfl_blocks);
    (void)hipFree(c->dfl_fsyms); (void)hipFree(c->dfl_fmeta); (void)hipFree(c->dfl_fblocks); (void)hipFree(c->dfl_rank); (void)hipFree(c->dfl_state); (void)hipFree(c->dfl_maxlen);
    if (c->dfl_seg_sync) for (int i = 0; i < 2; i++) { (void)hipStreamDestroy(c->dfl_sort_st[i]); for (int k = 0; k < 2; k++) { (void)hipEventDestroy(c->dfl_sorted[i][k]); (void)hipEventDestroy(c->dfl_parsed[i][k]); } }
    if (c->dfl_events) for (int i = 0; i < 2; i++) { (void)hipEventDestroy(c->dfl_searched[i]); (void)hipEventDestroy(c->dfl_done[i]); }
    delete c;
}

/* What the context holds on the device right now, by part (bytes): the arena (or the separate workspace allocations), the other
 * table sets, the decoders' staging, the DEFLATE workspace, the block-chain state.  Sets allocated on first use count once they exist. */
extern "C" int kmp_batch_memory(kmp_batch_ctx* c, kmp_batch_memory_info* info)
{
    if (!c || !info || info->struct_bytes < sizeof(kmp_batch_memory_info)) { g_last_error = "kmp_batch_memory: bad argument"; return KMP_ERR_ARG; }
    size_t const ns = c->max_slices;
    kmp_batch_memory_info m; memset(&m, 0, sizeof m); m.struct_bytes = sizeof m;
    size_t const tbytes = (size_t)c->nteams * KX_TBL_ENTRIES * sizeof(u32);
    if (c->arena) { m.arena = c->arena_bytes; m.arena_used = tbytes + ns * c->seq_cap * sizeof(KSeq) + ns * c->lit_cap + ns * sizeof(KSliceMeta) + ns * c->scratch_words * sizeof(u32); }
    else { m.arena = 0; m.arena_used = 0; m.workspace = tbytes + ns * c->seq_cap * sizeof(KSeq) + ns * c->lit_cap + ns * sizeof(KSliceMeta) + ns * c->scratch_words * sizeof(u32); }
    m.workspace += (size_t)c->nteams * sizeof(u32) + ns * sizeof(u32) + 192;
    if (c->tables_flat) m.other_tables += tbytes + (size_t)c->nteams * sizeof(u32);
    if (c->tables4) m.other_tables += (size_t)c->teams4 * KX_TBL4_ENTRIES * sizeof(u32) + (size_t)c->teams4 * sizeof(u32);
    if (c->big_tables4) m.other_tables += ns * KX_BIG4_ENTRIES * sizeof(u32);
    if (c->d_dict) m.other_tables += c->dict_size + 64 + ((size_t)4 << c->cdH) + ((size_t)4 << c->cdC);
    if (c->lz_srt) m.other_tables += (size_t)c->lz_chunk * c->lz_pos_cap * (sizeof(KLazyRec) + sizeof(u32));       // (levels 5 .. 10: the sorted positions' records and where each stands)
    if (c->big) m.block_chain = ns * sizeof(KFrameState) + ns * 512 * sizeof(u32) + ns * KX_BIG_TBL_ENTRIES * sizeof(u32) + ns * 4 + 64;
    if (c->pre_stage) m.decode_staging += (size_t)c->pre_slices * c->pre_seq_cap * 8u + (size_t)c->pre_slices * c->pre_blk_cap * sizeof(KPreBlk) + (size_t)c->pre_slices * 4u + ((size_t)c->pre_slices * 2u + KXP_SORT_BUCKETS) * 4u;
    if (c->pre_lits) m.decode_staging += (size_t)c->pre_slices * c->pre_lit_cap + (size_t)c->pre_slices * c->pre_blk_cap * sizeof(KPreLit) + (size_t)c->pre_slices * 4u;
    if (c->dfl_link) {
        size_t const span = c->dfl_pos_cap > 65536u ? 65536u : c->dfl_pos_cap;          // (above 64 KiB the search arrays hold one 64 KiB span per slice: kmp_deflate.hip)
        m.deflate_workspace += (size_t)2 * c->dfl_chunk * (span * ((c->dfl_rank ? 2u : 1u) * (sizeof(u16) + sizeof(KdBest) * 2u + sizeof(u32)) + (c->dfl_rank ? sizeof(u16) : 0)) + (size_t)c->dfl_pos_cap * sizeof(u32)
                                                            + sizeof(KdSliceMeta) + (size_t)c->dfl_blk_cap * sizeof(KdBlockInfo));
    }
    if (c->dfl_fsyms) m.deflate_workspace += (size_t)4 * c->dfl_chunk * ((size_t)c->dfl_pos_cap * sizeof(u32) + sizeof(KdSliceMeta) + (size_t)c->dfl_blk_cap * sizeof(KdBlockInfo));
    m.total = m.arena + m.workspace + m.other_tables + m.block_chain + m.decode_staging + m.deflate_workspace;
    *info = m;
    return KMP_OK;
}

extern "C" int kmp_batch_set_profiling(kmp_batch_ctx* c, int on) { if (!c) return KMP_ERR_ARG; c->profiling = on; return KMP_OK; }
extern "C" int kmp_batch_last_kernel_ms(kmp_batch_ctx* c, int which, float* ms)
{
    if (!c || which < 0 || which > 6 || !ms || !c->ev_valid[which]) { g_last_error = "no timing recorded"; return KMP_ERR_ARG; }
    if (which <= 1) {
        // zstd compress: mean duration of the k_zstd_match (0) / k_zstd_entropy (1) launches of the last batch
        float sum = 0;
        for (u32 i = 0; i < c->last_chunks; i++) {
            hipEvent_t* const e = which == 0 ? c->evm[i] : c->eve[i]; float t = 0;
            HIP_TRY(hipEventSynchronize(e[1]));
            HIP_TRY(hipEventElapsedTime(&t, e[0], e[1]));
            sum += t;
        }
        *ms = sum / (float)c->last_chunks;
        return KMP_OK;
    }
    HIP_TRY(hipEventSynchronize(c->ev[2 * which + 1]));
    HIP_TRY(hipEventElapsedTime(ms, c->ev[2 * which], c->ev[2 * which + 1]));
    return KMP_OK;
}
/* random 4-byte loads per second and load + store pairs per second over this context's level-3 team tables, measured when
 * the context was created (0 when its tables are small: nothing to price) */
extern "C" int kmp_batch_table_rates(kmp_batch_ctx* c, float* reads_per_s, float* pairs_per_s)
{
    if (!c || !reads_per_s || !pairs_per_s) { g_last_error = "kmp_batch_table_rates: null argument"; return KMP_ERR_ARG; }
    *reads_per_s = c->table_reads_per_s; *pairs_per_s = c->table_pairs_per_s;
    return KMP_OK;
}
// diagnostics: the per-slice records of the last one-block zstd batch (KSliceMeta, 32 bytes each) copied to host memory
extern "C" int kmp_debug_copy_meta(kmp_batch_ctx* c, void* h_dst, uint32_t n)
{
    if (!c || !h_dst || n > c->max_slices) { g_last_error = "kmp_debug_copy_meta: bad argument"; return KMP_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(h_dst, c->meta, (size_t)n * sizeof(KSliceMeta), hipMemcpyDeviceToHost));
    return KMP_OK;
}
/* launches of each zstd compress kernel in the last batch (the batch is cut into that many chunks) */
extern "C" int kmp_batch_last_chunks(kmp_batch_ctx* c) { return c ? (int)c->last_chunks : 0; }

extern "C" int kmp_batch_status(kmp_batch_ctx* c, uint32_t* bits, void* hip_stream)
{
    if (!c) { g_last_error = "kmp_batch_status: null context"; return KMP_ERR_ARG; }
    hipStream_t const st = (hipStream_t)hip_stream;
    HIP_TRY(hipSetDevice(c->device));
    KMP_TRY(batch_wait_previous(c, st));
    // read and clear in ONE atomic exchange: a bit raised by a batch on another stream between a copy and a memset would be lost
    u32 v = 0;
    hipLaunchKernelGGL(k_status_take, dim3(1), dim3(1), 0, st, c->d_status, c->d_status + 1);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&v, c->d_status + 1, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (bits) *bits = v;
    if (v & KMP_STATUS_SLICE_TOO_LARGE) { g_last_error = "a slice is larger than the context was created for: its out_len is 0"; return KMP_ERR_CAPACITY; }
    if (v & KMP_STATUS_KERNEL_GUARD) { g_last_error = "a parser's loop guard tripped: the slice's out_len is 0"; return KMP_ERR_KERNEL; }
    if (v & KMP_STATUS_LEVEL_SIZE) { g_last_error = "the level is another strategy at a slice's size (levels 9 and 10 up to 16 KiB; level 4 between 128 and 256 KiB): that slice's out_len is 0"; return KMP_ERR_CAPACITY; }
    return KMP_OK;
}

// A batch begins: it waits for the previous batch of this context (whatever stream that ran on), and its kernels get
// the sanitised lengths (k_len_guard).  A batch ends: oversized slices lose their frames, the event is recorded.
int batch_wait_previous(kmp_batch_ctx* c, hipStream_t st)
{
    if (c->have_done) HIP_TRY(hipStreamWaitEvent(st, c->ev_done, 0));
    for (u32 p = 0; p < c->pieces_pending; p++) HIP_TRY(hipStreamWaitEvent(st, c->ev_piece[p], 0));
    return KMP_OK;
}
int batch_begin(kmp_batch_ctx* c, hipStream_t st, const u32* d_in_len, u32 n, u32 cap)
{
    KMP_TRY(batch_wait_previous(c, st));
    if (d_in_len) {
        hipLaunchKernelGGL(k_len_guard, dim3((n + 255) / 256), dim3(256), 0, st, d_in_len, n, cap, c->len_ok, c->d_status);
        HIP_TRY(hipGetLastError());
    }
    return KMP_OK;
}
int batch_end(kmp_batch_ctx* c, hipStream_t st, const u32* d_in_len, u32 n, u32 cap, u32* d_out_len, const KSliceMeta* meta)
{
    if (d_in_len) {
        hipLaunchKernelGGL(k_len_guard_finish, dim3((n + 255) / 256), dim3(256), 0, st, d_in_len, n, cap, d_out_len, meta, c->d_status);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(c->ev_done, st)); c->have_done = 1; c->pieces_pending = 0;      // (this batch waited for the pieces before it)
    return KMP_OK;
}

extern "C" size_t kmp_zstd_compress_bound(size_t n)
{
    return n + (n >> 8) + ((n < (128u << 10)) ? (((128u << 10) - n) >> 11) : 0);
}
// the tables of the one-position-per-step parsers (levels 1 / 2, dictionary): the level-3 tables when those are one piece,
// else a piece of their own with its own epochs (measured over the spread tables: level 1 19.8 GB/s against 23.9)
static int flat_tables(kmp_batch_ctx* c, u32** tables, u32** epochs)
{
    if (c->tseg_n == 1) { *tables = c->tables; *epochs = c->team_epoch; return KMP_OK; }
    if (!c->tables_flat) {
        size_t const tbytes = (size_t)c->nteams * KX_TBL_ENTRIES * sizeof(u32);
        int const rc = place_alloc(&c->tables_flat, tbytes, nullptr, nullptr); if (rc != KMP_OK) return rc;
        HIP_TRY(hipMalloc((void**)&c->team_epoch_flat, (size_t)c->nteams * sizeof(u32)));
        HIP_TRY(hipMemset(c->tables_flat, 0, tbytes));
        HIP_TRY(hipMemset(c->team_epoch_flat, 0, (size_t)c->nteams * sizeof(u32)));
    }
    *tables = c->tables_flat; *epochs = c->team_epoch_flat;
    return KMP_OK;
}

// ---- levels 1 and 2 -------------------------------------------------------------------------------------
static int zstd_compress_dfast(kmp_batch_ctx* c, const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                               uint32_t n, void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len, void* hip_stream, int level);
// ---- levels 5 .. 10: strategies greedy / lazy / lazy2 (zstd_lazy.h), slices of one block ------------------------------------------------
// The batch goes through in pieces that share one workspace (20 bytes a position: the sorted positions with their first bytes, where each
// position stands): sort, then the wave-per-slice parse; the entropy kernel runs once over the whole batch.
static int lazy_workspace(kmp_batch_ctx* c, u32 need_bytes);
static int lazy_parse(kmp_batch_ctx* c, hipStream_t st, const void* d_src, const uint64_t* d_in_off, u32 n, u32 first0, int level);
static int zstd_compress_lazy(kmp_batch_ctx* c, const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                              uint32_t n, void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len, void* hip_stream, int level)
{
    if (!c || (n && (!d_src || !d_in_off || !d_in_len || !d_dst || !d_out_off || !d_out_len))) { g_last_error = "kmp_zstd_compress_batch_level: null argument"; return KMP_ERR_ARG; }
    if (n > c->max_slices) { g_last_error = "kmp_zstd_compress_batch_level: n exceeds the context's max_slices"; return KMP_ERR_CAPACITY; }
    if (c->big) { g_last_error = "kmp_zstd_compress_batch_level: levels 5 .. 10 are served for slices up to 128 KiB"; return KMP_ERR_CAPACITY; }
    if (n == 0) return KMP_OK;
    hipStream_t const st = (hipStream_t)hip_stream;
    HIP_TRY(hipSetDevice(c->device));
    KMP_TRY(batch_begin(c, st, d_in_len, n, c->max_slice_bytes));          // (first: waits for the context's previous batch, whose workspace this may replace)
    KMP_TRY(lazy_workspace(c, c->max_slice_bytes));
    KMP_TRY(lazy_parse(c, st, d_src, d_in_off, n, 0, level));
    KEntropyArgs e;
    e.src = (const u8*)d_src; e.in_off = d_in_off; e.in_len = c->len_ok; e.n_slices = n;
    e.seqs = c->seqs; e.seq_cap = c->seq_cap; e.lits = c->lits; e.lit_cap = c->lit_cap; e.meta = c->meta;
    e.scratch = c->scratch; e.scratch_words = c->scratch_words;
    e.dst = (u8*)d_dst; e.out_off = d_out_off; e.out_len = d_out_len;
    e.flags = 8u | ((u32)level << 12);           // literals are gathered by the entropy kernel; the level: it derives each slice's strategy from it
    hipLaunchKernelGGL(k_zstd_entropy, dim3(n), dim3(64), 0, st, e);
    HIP_TRY(hipGetLastError());
    c->last_chunks = 1;
    return batch_end(c, st, d_in_len, n, c->max_slice_bytes, d_out_len, c->meta);
}
// need_bytes: the largest slice these kernels will parse (level 4 hands them only its slices up to 16 KiB: a quarter of the memory)
static int lazy_workspace(kmp_batch_ctx* c, u32 need_bytes)
{
    if (c->lz_srt && c->lz_pos_cap < ((need_bytes + 63u) & ~63u)) {          // made for a smaller need: once more, larger
        (void)hipFree(c->lz_srt); (void)hipFree(c->lz_wr); c->lz_srt = nullptr; c->lz_wr = nullptr;
    }
    if (!c->lz_srt) {
        u32 const pos_cap = (need_bytes + 63u) & ~63u;
        u32 cap = (u32)((1ull << 30) / pos_cap); if (cap > 16384u) cap = 16384u; if (cap < 1u) cap = 1u;
        u32 const chunk = c->max_slices < cap ? c->max_slices : cap;
        HIP_TRY(hipMalloc((void**)&c->lz_srt, (size_t)chunk * pos_cap * sizeof(KLazyRec)));
        if (hipMalloc((void**)&c->lz_wr, (size_t)chunk * pos_cap * sizeof(u32)) != hipSuccess) {
            (void)hipGetLastError(); (void)hipFree(c->lz_srt); c->lz_srt = nullptr;
            g_last_error = "kmp_zstd_compress_batch_level: no memory for the workspace of levels 5 .. 10"; return KMP_ERR_HIP;
        }
        (void)hipFree(c->lz_order); c->lz_order = nullptr;
        if (hipMalloc((void**)&c->lz_order, ((size_t)2 * chunk + 256) * sizeof(u32)) != hipSuccess) { (void)hipGetLastError(); c->lz_order = nullptr; }      // (without it the slices go as they come)
        c->lz_pos_cap = pos_cap; c->lz_chunk = chunk;
    }
    return KMP_OK;
}
// sort + parse of the slices [first0, first0 + n) of a batch, piece by piece (the slices' sanitised lengths are c->len_ok; sequences and
// the per-slice record go where the other parsers put theirs).  Slices the level does not parse this way at their size are skipped: at
// level 4 their record stays what k_zstd_match left (it serves 16 KiB < size <= 128 KiB), at the other levels it says "not served".
static int lazy_parse(kmp_batch_ctx* c, hipStream_t st, const void* d_src, const uint64_t* d_in_off, u32 n, u32 first0, int level)
{
    for (u32 first = first0; first < first0 + n; first += c->lz_chunk) {
        u32 const m = (first0 + n - first < c->lz_chunk) ? first0 + n - first : c->lz_chunk;
        KLazyArgs g;
        g.src = (const u8*)d_src; g.in_off = d_in_off + first; g.in_len = c->len_ok + first; g.n_slices = m;
        g.rec = (KLazyRec*)c->lz_srt; g.wr = c->lz_wr; g.pos_cap = c->lz_pos_cap;
        g.seqs = c->seqs + (size_t)first * c->seq_cap; g.seq_cap = c->seq_cap; g.meta = c->meta + first; g.level = (u32)level;
        // (the parse takes the costliest slices first: cost classes from the sort, a counting sort of the classes; batches worth ordering only)
        u32* const ord = (m >= 1024u && c->lz_order) ? c->lz_order : nullptr;
        if (ord) { g.order_key = ord; g.order_hist = ord + c->lz_chunk; HIP_TRY(hipMemsetAsync(g.order_hist, 0, 256 * sizeof(u32), st)); }
        hipLaunchKernelGGL(k_zstd_lazy_sort, dim3(m), dim3(256), 0, st, g);
        if (ord) { KMP_TRY(size_sort_keys(c, st, m, g.order_key, g.order_hist, ord + c->lz_chunk + 256)); g.order = ord + c->lz_chunk + 256; }
        if (c->max_slice_bytes <= 65536u) hipLaunchKernelGGL(k_zstd_lazy<2048>, dim3(m), dim3(64), 0, st, g);
        else hipLaunchKernelGGL(k_zstd_lazy<4096>, dim3(m), dim3(64), 0, st, g);
        HIP_TRY(hipGetLastError());
    }
    return KMP_OK;
}

extern "C" int kmp_zstd_compress_batch_level(kmp_batch_ctx* c, const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                             uint32_t n, void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len, int level, void* hip_stream)
{
    if (level == 3 || level == 0) return kmp_zstd_compress_batch(c, d_src, d_in_off, d_in_len, n, d_dst, d_out_off, d_out_len, hip_stream);
    if (level == 4) return zstd_compress_dfast(c, d_src, d_in_off, d_in_len, n, d_dst, d_out_off, d_out_len, hip_stream, 4);
    if (level >= 5 && level <= 10) return zstd_compress_lazy(c, d_src, d_in_off, d_in_len, n, d_dst, d_out_off, d_out_len, hip_stream, level);
    bool const neg = level < 0;                // negative levels: strategy "fast" with a step of 1 - level, literals left uncompressed
    if ((level != 1 && level != 2 && !neg) || level < -131072) { g_last_error = "kmp_zstd_compress_batch_level: levels -131072 .. -1 and 1 .. 10 are served"; return KMP_ERR_ARG; }
    if (!c || (n && (!d_src || !d_in_off || !d_in_len || !d_dst || !d_out_off || !d_out_len))) { g_last_error = "kmp_zstd_compress_batch_level: null argument"; return KMP_ERR_ARG; }
    if (n > c->max_slices) { g_last_error = "kmp_zstd_compress_batch_level: n exceeds the context's max_slices"; return KMP_ERR_CAPACITY; }
    if (n == 0) return KMP_OK;
    hipStream_t const st = (hipStream_t)hip_stream;
    HIP_TRY(hipSetDevice(c->device));
    if (c->big) {
        // frames of several blocks, any size the context holds: beyond the level's window (512 KiB at level 1 and at the negative levels,
        // 1 MiB at level 2) libzstd's staging buffer wraps and the window slides (zstd_match_fast_ext_body)
        return zstd_compress_big(c, d_src, d_in_off, d_in_len, n, d_dst, d_out_off, d_out_len, st, 0, neg ? 1u : (u32)level, 0, neg ? (u32)(1 - level) : 0u);
    }
    KMP_TRY(batch_begin(c, st, d_in_len, n, c->max_slice_bytes));
    HIP_TRY(hipMemsetAsync(c->counter, 0, 4 * KMP_MAX_CHUNKS, st));
    // One launch of each kernel.  (KMP_ZSTD_CHUNKS=2 splits the batch as at level 3, the entropy kernel of the first chunk on
    // the second stream beside the parse of the second: measured 16.3 - 17.1 GB/s against 21.9 in one piece at 65 536 x 64 KiB --
    // this parser needs all the slices in flight.)
    u32 const tpw = 64 / (u32)c->G;
    u32 chunks = c->knob.chunks ? c->knob.chunks : 1u;
    if (chunks > 2) chunks = 2;
    u32 starts[3] = { 0, n, n };
    if (chunks == 2) { u32 const pm = c->knob.fast_first_permille; starts[1] = (u32)((u64)n * (pm >= 100 && pm <= 950 ? pm : 500u) / 1000u) & ~63u; if (starts[1] == 0 || starts[1] >= n) chunks = 1, starts[1] = n; }
    bool forked = false;
    for (u32 ci = 0; ci < chunks; ci++) {
        u32 const first = starts[ci], m_n = starts[ci + 1] - first;
        if (m_n == 0) continue;
        KFastArgs g;
        g.m.src = (const u8*)d_src; g.m.in_off = d_in_off + first; g.m.in_len = c->len_ok + first; g.m.n_slices = m_n;
        g.m.seqs = c->seqs + (size_t)first * c->seq_cap; g.m.seq_cap = c->seq_cap; g.m.lits = c->lits + (size_t)first * c->lit_cap; g.m.lit_cap = c->lit_cap; g.m.meta = c->meta + first;
        { u32* ft_ = nullptr; u32* fe_ = nullptr; KMP_TRY(flat_tables(c, &ft_, &fe_)); g.m.tables = ft_; g.m.tseg_n = 1; g.m.team_epoch = fe_; } g.m.counter = c->counter + ci; g.m.flags = 6; g.m.fstate = nullptr; g.m.big_tables = nullptr;
        g.level = neg ? 0u : (u32)level; g.step0 = neg ? (u32)(1 - level) : 2u;
        u32 blocks = (m_n + tpw - 1) / tpw; if (blocks > c->match_blocks) blocks = c->match_blocks;
        switch (c->G) {
        case 2:  hipLaunchKernelGGL(k_zstd_match_fast<2>, dim3(blocks), dim3(64), 0, st, g); break;
        case 4:  hipLaunchKernelGGL(k_zstd_match_fast<4>, dim3(blocks), dim3(64), 0, st, g); break;
        case 8:  hipLaunchKernelGGL(k_zstd_match_fast<8>, dim3(blocks), dim3(64), 0, st, g); break;
        case 16: hipLaunchKernelGGL(k_zstd_match_fast<16>, dim3(blocks), dim3(64), 0, st, g); break;
        case 32: hipLaunchKernelGGL(k_zstd_match_fast<32>, dim3(blocks), dim3(64), 0, st, g); break;
        default: hipLaunchKernelGGL(k_zstd_match_fast<64>, dim3(blocks), dim3(64), 0, st, g); break;
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->evm[ci][1], st));
        KEntropyArgs e;
        e.src = (const u8*)d_src; e.in_off = d_in_off + first; e.in_len = c->len_ok + first; e.n_slices = m_n;
        e.seqs = g.m.seqs; e.seq_cap = c->seq_cap; e.lits = g.m.lits; e.lit_cap = c->lit_cap; e.meta = g.m.meta;
        e.scratch = c->scratch + (size_t)first * c->scratch_words; e.scratch_words = c->scratch_words;
        e.dst = (u8*)d_dst; e.out_off = d_out_off + first; e.out_len = d_out_len + first; e.flags = 8u | 32u | (neg ? 64u : 0u);   // gather literals; strategy "fast"; negative levels: literals stay raw
        hipStream_t es = st;
        if (ci + 1 < chunks) { es = c->st2; HIP_TRY(hipStreamWaitEvent(es, c->evm[ci][1], 0)); forked = true; }
        hipLaunchKernelGGL(k_zstd_entropy, dim3(m_n), dim3(64), 0, es, e);
        HIP_TRY(hipGetLastError());
    }
    if (forked) { HIP_TRY(hipEventRecord(c->ev_join, c->st2)); HIP_TRY(hipStreamWaitEvent(st, c->ev_join, 0)); }
    c->last_chunks = chunks;
    return batch_end(c, st, d_in_len, n, c->max_slice_bytes, d_out_len, c->meta);
}

// ---- compressing with a raw-content dictionary ------------------------------------------------------
// libzstd's CDict for the dictionary: parameters of ZSTD_getCParams(3, unknown source size, dictSize) in
// "create CDict" mode, then ZSTD_fillDoubleHashTableForCDict over the dictionary (tagged entries: index << 8 | tag).
int dict_header_state(const unsigned char* dict, size_t dict_size, int for_decoder)
{
    KDictPrior pr; KDictDPrior dp; size_t off = 0;
    return cdict_parse_formatted(dict, dict_size, &pr, &off, for_decoder ? &dp : nullptr);
}

extern "C" int kmp_zstd_compress_batch_dict(kmp_batch_ctx* c, const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                            uint32_t n, void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len,
                                            const void* h_dict, uint32_t dict_size, void* hip_stream)
{
    if (!c || !h_dict || (n && (!d_src || !d_in_off || !d_in_len || !d_dst || !d_out_off || !d_out_len))) { g_last_error = "kmp_zstd_compress_batch_dict: null argument"; return KMP_ERR_ARG; }
    if (n > c->max_slices) { g_last_error = "kmp_zstd_compress_batch_dict: n exceeds the context's max_slices"; return KMP_ERR_CAPACITY; }
    if (c->big) { g_last_error = "kmp_zstd_compress_batch_dict: slices above 128 KiB are not served with a dictionary"; return KMP_ERR_CAPACITY; }
    if (dict_size < 8 || dict_size > KX_MAX_DICT) { g_last_error = "kmp_zstd_compress_batch_dict: dictionary of 8 .. 130560 bytes expected"; return KMP_ERR_CAPACITY; }
    if (n == 0) return KMP_OK;
    hipStream_t const st = (hipStream_t)hip_stream;
    HIP_TRY(hipSetDevice(c->device));
    // (re)build the CDict when the dictionary changed
    u64 hsh = 1469598103934665603ull; for (u32 i = 0; i < dict_size; i++) { hsh ^= ((const u8*)h_dict)[i]; hsh *= 1099511628211ull; }
    if (!c->d_dict || c->dict_size != dict_size || c->dict_hash != hsh) {
        // A dictionary in zstd's own format (magic EC30A437) is loaded as libzstd loads it: entropy tables and repeat offsets for the first
        // block (KDictPrior), the bytes behind them as the content matches are searched in; anything else is content from its first byte.
        KDictPrior prior; size_t content_off = 0;
        int const formatted = cdict_parse_formatted((const u8*)h_dict, dict_size, &prior, &content_off);
        if (formatted < 0) { g_last_error = "kmp_zstd_compress_batch_dict: the dictionary starts with zstd's dictionary magic but its header is damaged (libzstd: Dictionary is corrupted)"; return KMP_ERR_ARG; }
        const u8* const content = (const u8*)h_dict + content_off; u32 const content_size = dict_size - (u32)content_off;
        HIP_TRY(hipStreamSynchronize(st));
        (void)hipFree(c->d_dict); (void)hipFree(c->d_dictL); (void)hipFree(c->d_dictS); (void)hipFree(c->d_prior); c->d_dict = nullptr; c->d_dictL = nullptr; c->d_dictS = nullptr; c->d_prior = nullptr;
        cdict_params(dict_size, &c->cdW, &c->cdC, &c->cdH, &c->cdM);        // (libzstd sizes the CDict and the frame's window by the whole dictionary, header included)
        std::vector<u32> tl, ts;
        cdict_fill(tl, c->cdH, ts, c->cdC, c->cdM, content, content_size);
        HIP_TRY(hipMalloc((void**)&c->d_dict, content_size + 64));
        HIP_TRY(hipMalloc((void**)&c->d_dictL, tl.size() * 4)); HIP_TRY(hipMalloc((void**)&c->d_dictS, ts.size() * 4));
        HIP_TRY(hipMemcpy(c->d_dict, content, content_size, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->d_dictL, tl.data(), tl.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->d_dictS, ts.data(), ts.size() * 4, hipMemcpyHostToDevice));
        if (formatted) {
            HIP_TRY(hipMalloc((void**)&c->d_prior, sizeof(KDictPrior)));
            HIP_TRY(hipMemcpy(c->d_prior, &prior, sizeof(KDictPrior), hipMemcpyHostToDevice));
            c->dict_rep[0] = prior.rep[0]; c->dict_rep[1] = prior.rep[1];
        } else { c->dict_rep[0] = 1; c->dict_rep[1] = 4; }
        c->dict_content = content_size;
        c->dict_size = dict_size; c->dict_hash = hsh;
    }
    KMP_TRY(batch_begin(c, st, d_in_len, n, c->max_slice_bytes));
    HIP_TRY(hipMemsetAsync(c->counter, 0, 4, st));
    KDictArgs g;
    g.m.src = (const u8*)d_src; g.m.in_off = d_in_off; g.m.in_len = c->len_ok; g.m.n_slices = n;
    g.m.seqs = c->seqs; g.m.seq_cap = c->seq_cap; g.m.lits = c->lits; g.m.lit_cap = c->lit_cap; g.m.meta = c->meta;
    { u32* ft_ = nullptr; u32* fe_ = nullptr; KMP_TRY(flat_tables(c, &ft_, &fe_)); g.m.tables = ft_; g.m.tseg_n = 1; g.m.team_epoch = fe_; } g.m.counter = c->counter; g.m.flags = 6; g.m.fstate = nullptr; g.m.big_tables = nullptr;
    g.dict = c->d_dict; g.dict_size = c->dict_content; g.dictL = c->d_dictL; g.dictS = c->d_dictS; g.rep0 = c->dict_rep[0]; g.rep1 = c->dict_rep[1];
    g.dWindowLog = c->cdW; g.dHashLog = c->cdH; g.dChainLog = c->cdC; g.dMinMatch = c->cdM;
    u32 const tpw = 64 / (u32)c->G;
    u32 blocks = (n + tpw - 1) / tpw; if (blocks > c->match_blocks) blocks = c->match_blocks;
    switch (c->G) {
    case 2:  hipLaunchKernelGGL(k_zstd_match_dict<2>, dim3(blocks), dim3(64), 0, st, g); break;
    case 4:  hipLaunchKernelGGL(k_zstd_match_dict<4>, dim3(blocks), dim3(64), 0, st, g); break;
    case 8:  hipLaunchKernelGGL(k_zstd_match_dict<8>, dim3(blocks), dim3(64), 0, st, g); break;
    case 16: hipLaunchKernelGGL(k_zstd_match_dict<16>, dim3(blocks), dim3(64), 0, st, g); break;
    case 32: hipLaunchKernelGGL(k_zstd_match_dict<32>, dim3(blocks), dim3(64), 0, st, g); break;
    default: hipLaunchKernelGGL(k_zstd_match_dict<64>, dim3(blocks), dim3(64), 0, st, g); break;
    }
    HIP_TRY(hipGetLastError());
    KEntropyArgs e;
    e.src = (const u8*)d_src; e.in_off = d_in_off; e.in_len = c->len_ok; e.n_slices = n;
    e.seqs = c->seqs; e.seq_cap = c->seq_cap; e.lits = c->lits; e.lit_cap = c->lit_cap; e.meta = c->meta;
    e.scratch = c->scratch; e.scratch_words = c->scratch_words;
    e.dst = (u8*)d_dst; e.out_off = d_out_off; e.out_len = d_out_len; e.flags = 8u;       // literals are gathered by the entropy kernel
    e.prior = c->d_prior;
    if (c->d_prior) hipLaunchKernelGGL(k_zstd_entropy_prior, dim3(n), dim3(64), 0, st, e);
    else hipLaunchKernelGGL(k_zstd_entropy, dim3(n), dim3(64), 0, st, e);
    HIP_TRY(hipGetLastError());
    c->last_chunks = 1;
    return batch_end(c, st, d_in_len, n, c->max_slice_bytes, d_out_len, c->meta);
}

// Slices above 128 KiB: frames of several blocks.  Every round runs the match kernel and the frame kernel over
// one block of every unfinished slice; block sizes depend on the bytes already produced (ZSTD_optimalBlockSize),
// so the rounds are sequential and the host only reads back how many frames are still open.
// stream: KFrameArgs.stream (0 ZSTD_compress2's frames, 1 / 2 streaming frames, 3 the reference's one-shot driver)
int zstd_compress_big(kmp_batch_ctx* c, const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                      uint32_t n, void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len, hipStream_t st, u32 stream, u32 strategy, u32 tail_direct, u32 fast_step0, bool level4)
{
    bool const streaming = stream == 1 || stream == 2;
    const uint32_t* const d_in_len_caller = d_in_len;
    KMP_TRY(batch_begin(c, st, d_in_len, n, c->max_slice_bytes));
    d_in_len = c->len_ok;
    if (level4) {
        // level 4's double-fast rows on this path (16 - 128 KiB: hash 17 / chain 17; above 256 KiB and streams: 18 / 18): per-slice tables of
        // 2 MiB, allocated by the first such batch of the context
        if (!c->big_tables4 && hipMalloc((void**)&c->big_tables4, (size_t)c->max_slices * KX_BIG4_ENTRIES * sizeof(u32)) != hipSuccess) {
            (void)hipGetLastError(); c->big_tables4 = nullptr; g_last_error = "kmp_zstd_compress_batch_level: no memory for level 4's tables"; return KMP_ERR_HIP;
        }
        HIP_TRY(hipMemsetAsync(c->big_tables4, 0, (size_t)n * KX_BIG4_ENTRIES * sizeof(u32), st));
    } else
    HIP_TRY(hipMemsetAsync(c->big_tables, 0, (size_t)n * KX_BIG_TBL_ENTRIES * sizeof(u32), st));
    HIP_TRY(hipMemsetAsync(c->remaining, 0, 4, st));
    // strategy: 0 level 3 (double-fast), 1 level 1 (fast), 2 level 2 (fast, but double-fast for 128 KiB < size <= 256 KiB when the size is known)
    u32 const level2 = strategy == 2u ? 1u : 0u;
    hipLaunchKernelGGL(k_zstd_frame_init, dim3((n + 255) / 256), dim3(256), 0, st, d_in_len, n, c->fstate, (u8*)d_dst, d_out_off, d_out_len, c->remaining, streaming ? (strategy == 1u ? 0x48u : strategy == 2u ? 0x50u : 0x58u) : 0u, stream != 0 ? 1u : 0u, level4 ? 1u : 0u, c->d_status);
    HIP_TRY(hipGetLastError());
    KMatchArgs m;
    m.src = (const u8*)d_src; m.in_off = d_in_off; m.in_len = d_in_len; m.n_slices = n;
    m.seqs = c->seqs; m.seq_cap = c->seq_cap; m.meta = c->meta; m.lits = c->lits; m.lit_cap = c->lit_cap;
    m.tables = c->tables; for (int ts_ = 0; ts_ < 4; ts_++) m.tseg[ts_] = c->tseg[ts_]; m.tseg_n = c->tseg_n; m.team_epoch = c->team_epoch; m.counter = c->counter;
    m.flags = 2u | (streaming ? 8u : 0u) | (c->max_slice_bytes >= KX_BLK_WIDE_FROM ? 16u : 0u);
    m.fstate = c->fstate; m.big_tables = c->big_tables;
    if (level4) { m.level = 4; m.big_tables = c->big_tables4; m.big_stride = KX_BIG4_ENTRIES; m.big_long = KX_BIG4_LONG; }
    KFrameArgs e;
    e.src = (const u8*)d_src; e.in_off = d_in_off; e.in_len = d_in_len; e.n_slices = n;
    e.seqs = c->seqs; e.seq_cap = c->seq_cap; e.lits = c->lits; e.lit_cap = c->lit_cap; e.meta = c->meta;
    e.scratch = c->scratch; e.scratch_words = c->scratch_words;
    e.dst = (u8*)d_dst; e.out_off = d_out_off; e.out_len = d_out_len;
    e.fstate = c->fstate; e.hufct = c->hufct; e.remaining = c->remaining; e.stream = stream; e.strategy = strategy ? 1u : 0u; e.level2 = level2; e.cls = 0;
    e.fast_step0 = strategy == 1u ? fast_step0 : 0u;         // a negative level: the level-1 machinery (window 2^19) on row 0 of the tables, a step of 1 - level, raw literals
    e.tail_direct = stream == 3 ? 0u : tail_direct; e.out_chunk = stream == 3 ? tail_direct : 0u;      // (one parameter: the mode says which it is)
    e.status_word = c->d_status;
#ifdef KMP_ABLATIONS
    if (strategy || c->knob.big_rounds == 0)
#endif
    {
        // one wave per slice walks its chain of blocks
        // few slices: one per wave (most waves); many: up to 64 / G per wave so that all of them are in flight
        KBigArgs g; g.m = m; g.e = e; g.counters = c->big_counters;
        // lanes per slice: every slice of the batch should be in flight (its block chain is serial), and a parse team
        // gains little beyond 8 lanes -- measured on 1 MiB and 256 KiB slices: 8 lanes up to 16 K slices, 4 above
        int const bigG = c->big_G ? c->big_G : (n >= 16384u ? 4 : 8);
        u32 const resident = 12u * c->cus;
        u32 spw = c->knob.big_spw ? c->knob.big_spw : (n + resident - 1) / resident;
        if (spw < 1) spw = 1; if (spw > 64u / (u32)bigG) spw = 64u / (u32)bigG;
        if (!c->knob.big_spw) {
            // ... and a grid that is a whole number of waves per CU when a few more slices per wave give one: the launch lasts as
            // long as its fullest CU (8 192 x 1 MiB: 3 slices per wave = 2 731 waves 6.5 GB/s, 4 = 2 048 waves 7.3)
            for (u32 t = spw; t <= 64u / (u32)bigG && t <= spw + 2u; t++) if (((n + t - 1) / t) % c->cus == 0) { spw = t; break; }
        }
        g.spw = spw;
        u32 const grid = (n + spw - 1) / spw;
        HIP_TRY(hipMemsetAsync(c->big_counters, 0, (size_t)n * 4, st));
        if (level2 && !streaming) {
            // level 2, sizes known: the slices of its double-fast row first (class 1), the others (class 2) through the fast parser below
            g.e.strategy = 0; g.e.cls = 1; g.m.flags = m.flags | 32u | (1u << 6);
            switch (bigG) {
            case 2:  hipLaunchKernelGGL(k_zstd_big<2>, dim3(grid), dim3(64), 0, st, g); break;
            case 4:  hipLaunchKernelGGL(k_zstd_big<4>, dim3(grid), dim3(64), 0, st, g); break;
            case 8:  hipLaunchKernelGGL(k_zstd_big<8>, dim3(grid), dim3(64), 0, st, g); break;
            case 16: hipLaunchKernelGGL(k_zstd_big<16>, dim3(grid), dim3(64), 0, st, g); break;
            case 32: hipLaunchKernelGGL(k_zstd_big<32>, dim3(grid), dim3(64), 0, st, g); break;
            default: hipLaunchKernelGGL(k_zstd_big<64>, dim3(grid), dim3(64), 0, st, g); break;
            }
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemsetAsync(c->big_counters, 0, (size_t)n * 4, st));
            g.e.strategy = 1; g.e.cls = 2; g.m.flags = m.flags | (2u << 6);
        }
        if (strategy) switch (bigG) {
        case 2:  hipLaunchKernelGGL(k_zstd_big_fast<2>, dim3(grid), dim3(64), 0, st, g); break;
        case 4:  hipLaunchKernelGGL(k_zstd_big_fast<4>, dim3(grid), dim3(64), 0, st, g); break;
        case 8:  hipLaunchKernelGGL(k_zstd_big_fast<8>, dim3(grid), dim3(64), 0, st, g); break;
        case 16: hipLaunchKernelGGL(k_zstd_big_fast<16>, dim3(grid), dim3(64), 0, st, g); break;
        case 32: hipLaunchKernelGGL(k_zstd_big_fast<32>, dim3(grid), dim3(64), 0, st, g); break;
        default: hipLaunchKernelGGL(k_zstd_big_fast<64>, dim3(grid), dim3(64), 0, st, g); break;
        }
        else switch (bigG) {
        case 2:  hipLaunchKernelGGL(k_zstd_big<2>, dim3(grid), dim3(64), 0, st, g); break;
        case 4:  hipLaunchKernelGGL(k_zstd_big<4>, dim3(grid), dim3(64), 0, st, g); break;
        case 8:  hipLaunchKernelGGL(k_zstd_big<8>, dim3(grid), dim3(64), 0, st, g); break;
        case 16: hipLaunchKernelGGL(k_zstd_big<16>, dim3(grid), dim3(64), 0, st, g); break;
        case 32: hipLaunchKernelGGL(k_zstd_big<32>, dim3(grid), dim3(64), 0, st, g); break;
        default: hipLaunchKernelGGL(k_zstd_big<64>, dim3(grid), dim3(64), 0, st, g); break;
        }
        HIP_TRY(hipGetLastError());
        c->last_rounds = 0; c->last_chunks = 1;
        return batch_end(c, st, d_in_len_caller, n, c->max_slice_bytes, d_out_len, nullptr);
    }
#ifdef KMP_ABLATIONS
    // (experiment switch KMP_BIG_ROUNDS=1) the same steps as separate launches per round of blocks
    int const bigR = c->big_G ? c->big_G : 8;
    u32 const tpw = 64 / (u32)bigR;
    u32 const blocks = (n + tpw - 1) / tpw;
    u32 rounds = 0;
    for (;;) {
        u32 left = 0;
        HIP_TRY(hipMemcpyAsync(&left, c->remaining, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (left == 0) break;
        if (++rounds > KMP_MAX_BIG_SLICE_BYTES / 64u) { g_last_error = "kmp_zstd_compress_batch: block rounds did not finish"; return KMP_ERR_KERNEL; }
        HIP_TRY(hipMemsetAsync(c->counter, 0, 8, st));
        switch (bigR) {
        case 2:  hipLaunchKernelGGL(k_zstd_match_blk<2>, dim3(blocks), dim3(64), 0, st, m); break;
        case 4:  hipLaunchKernelGGL(k_zstd_match_blk<4>, dim3(blocks), dim3(64), 0, st, m); break;
        case 8:  hipLaunchKernelGGL(k_zstd_match_blk<8>, dim3(blocks), dim3(64), 0, st, m); break;
        case 16: hipLaunchKernelGGL(k_zstd_match_blk<16>, dim3(blocks), dim3(64), 0, st, m); break;
        case 32: hipLaunchKernelGGL(k_zstd_match_blk<32>, dim3(blocks), dim3(64), 0, st, m); break;
        default: hipLaunchKernelGGL(k_zstd_match_blk<64>, dim3(blocks), dim3(64), 0, st, m); break;
        }
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_zstd_frame, dim3(n), dim3(64), 0, st, e);
        HIP_TRY(hipGetLastError());
    }
    c->last_rounds = rounds; c->last_chunks = 1;
    return batch_end(c, st, d_in_len_caller, n, c->max_slice_bytes, d_out_len, nullptr);
#else
    g_last_error = "zstd_compress_big: unreachable"; return KMP_ERR_ARG;
#endif
}
/* Streaming frames: what libzstd writes when a slice arrives through finish = false calls and is closed with
 * finish = true (size unknown when the frame starts).  empty_end: the closing calls brought no data.  The context must
 * have been created for slices above 128 KiB (the block-chain path and its 2^17 / 2^16 tables serve every size here). */
extern "C" int kmp_zstd_compress_batch_stream_level(kmp_batch_ctx* c, const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                                    uint32_t n, void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len, int empty_end, int level, void* hip_stream);
extern "C" int kmp_zstd_compress_batch_stream(kmp_batch_ctx* c, const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                              uint32_t n, void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len, int empty_end, void* hip_stream)
{ return kmp_zstd_compress_batch_stream_level(c, d_src, d_in_off, d_in_len, n, d_dst, d_out_off, d_out_len, empty_end, 3, hip_stream); }
/* level 3, or level 1 (the Ktor encoder's: streams <= 512 KiB, its window) or level 2 (<= 1 MiB) */
extern "C" int kmp_zstd_compress_batch_stream_level(kmp_batch_ctx* c, const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                                    uint32_t n, void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len, int empty_end, int level, void* hip_stream)
{
    if (level == 0) level = 3;
    bool const neg = level < 0;
    if ((level < 1 && !neg) || level > 4 || level < -131072) { g_last_error = "kmp_zstd_compress_batch_stream_level: levels -131072 .. -1 and 1 .. 4 are served"; return KMP_ERR_ARG; }
    if (!c || (n && (!d_src || !d_in_off || !d_in_len || !d_dst || !d_out_off || !d_out_len))) { g_last_error = "kmp_zstd_compress_batch_stream: null argument"; return KMP_ERR_ARG; }
    if (!c->big) { g_last_error = "kmp_zstd_compress_batch_stream: the context must be created with max_slice_bytes above 128 KiB"; return KMP_ERR_CAPACITY; }
    if (n > c->max_slices) { g_last_error = "kmp_zstd_compress_batch_stream: n exceeds the context's max_slices"; return KMP_ERR_CAPACITY; }
    if (n == 0) return KMP_OK;
    HIP_TRY(hipSetDevice(c->device));
    return zstd_compress_big(c, d_src, d_in_off, d_in_len, n, d_dst, d_out_off, d_out_len, (hipStream_t)hip_stream, empty_end ? 2u : 1u, (level == 3 || level == 4) ? 0u : neg ? 1u : (u32)level, 0, neg ? (u32)(1 - level) : 0u, level == 4);
}
/* What ZstdCompressor(level).transform(ByteArray) returns: above 128 KiB libzstd stages the input in chunks of 128 KiB
 * because the reference's output slices are smaller than ZSTD_compressBound (include/kompressor_hip.h). */
extern "C" int kmp_zstd_compress_batch_reference(kmp_batch_ctx* c, const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                                 uint32_t n, void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len, int level, uint32_t out_chunk, void* hip_stream)
{
    if (level == 0) level = 3;
    if (!c) { g_last_error = "kmp_zstd_compress_batch_reference: null argument"; return KMP_ERR_ARG; }
    if (!c->big) return kmp_zstd_compress_batch_level(c, d_src, d_in_off, d_in_len, n, d_dst, d_out_off, d_out_len, level, hip_stream);   // one block: one chunk
    bool const neg = level < 0;
    if ((level < 1 && !neg) || level > 4 || level < -131072) { g_last_error = "kmp_zstd_compress_batch_reference: levels -131072 .. -1 and 1 .. 4 are served"; return KMP_ERR_ARG; }
    if (n && (!d_src || !d_in_off || !d_in_len || !d_dst || !d_out_off || !d_out_len)) { g_last_error = "kmp_zstd_compress_batch_reference: null argument"; return KMP_ERR_ARG; }
    if (n > c->max_slices) { g_last_error = "kmp_zstd_compress_batch_reference: n exceeds the context's max_slices"; return KMP_ERR_CAPACITY; }
    if (n == 0) return KMP_OK;
    HIP_TRY(hipSetDevice(c->device));
    return zstd_compress_big(c, d_src, d_in_off, d_in_len, n, d_dst, d_out_off, d_out_len, (hipStream_t)hip_stream, 3u, (level == 3 || level == 4) ? 0u : neg ? 1u : (u32)level, out_chunk, neg ? (u32)(1 - level) : 0u, level == 4);
}
/* block rounds of the last batch of a context for slices above 128 KiB */
extern "C" int kmp_batch_last_rounds(kmp_batch_ctx* c) { return c ? (int)c->last_rounds : 0; }

// Level 4 where it is the double-fast parse (slices above 16 KiB up to 128 KiB: zstd_common.h kx_params_l4): its tables are
// 1 MiB per team, a set of their own allocated by the first such batch (as many teams as level 3 has, memory permitting; the
// batch's slices go through them by the work counter).
static int ensure_tables4(kmp_batch_ctx* c)
{
    if (c->tables4) return KMP_OK;
    // as many teams as level 3 has (65 536 x 64 KiB: 19.4 GB/s with 65 536 teams = 64 GiB of tables, 15.9 with 32 768, 14.6 with 16 384:
    // tools/r03_l4.sh), fewer when the device has less room (16 GiB stay free); KMP_L4_TEAMS caps it (the tests run many slices
    // through few teams)
    u32 const cap4 = env_u32("KMP_L4_TEAMS", 65536);
    u32 teams = c->nteams < cap4 ? c->nteams : cap4;
    { size_t fr = 0, tot = 0; if (hipMemGetInfo(&fr, &tot) == hipSuccess) { size_t const room = fr > ((size_t)16 << 30) ? (fr - ((size_t)16 << 30)) / ((size_t)KX_TBL4_ENTRIES * sizeof(u32)) : 0; if (teams > room) teams = (u32)room; } else (void)hipGetLastError(); }
    teams &= ~63u; if (teams == 0) teams = 64;
    size_t const bytes = (size_t)teams * KX_TBL4_ENTRIES * sizeof(u32);
    u32* t = nullptr; u32* e = nullptr;
    if (hipMalloc((void**)&t, bytes) != hipSuccess || hipMalloc((void**)&e, (size_t)teams * sizeof(u32)) != hipSuccess) {
        (void)hipGetLastError(); (void)hipFree(t); (void)hipFree(e);
        g_last_error = "kmp_zstd_compress_batch_level: no memory for level 4's tables"; return KMP_ERR_HIP;
    }
    if (hipMemset(t, 0, bytes) != hipSuccess || hipMemset(e, 0, (size_t)teams * sizeof(u32)) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(t); (void)hipFree(e); g_last_error = "kmp_zstd_compress_batch_level: hipMemset failed"; return KMP_ERR_HIP; }
    c->tables4 = t; c->epoch4 = e; c->teams4 = teams;
    return KMP_OK;
}

static int zstd_compress_dfast(kmp_batch_ctx* c, const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                               uint32_t n, void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len, void* hip_stream, int level)
{
    if (!c || (n && (!d_src || !d_in_off || !d_in_len || !d_dst || !d_out_off || !d_out_len))) { g_last_error = "kmp_zstd_compress_batch: null argument"; return KMP_ERR_ARG; }
    if (n > c->max_slices) { g_last_error = "kmp_zstd_compress_batch: n exceeds the context's max_slices"; return KMP_ERR_CAPACITY; }
    if (n == 0) return KMP_OK;
    hipStream_t const st = (hipStream_t)hip_stream;
    HIP_TRY(hipSetDevice(c->device));
    bool const l4 = level == 4;
    if (l4 && !c->big) KMP_TRY(ensure_tables4(c));
    if (c->big) return zstd_compress_big(c, d_src, d_in_off, d_in_len, n, d_dst, d_out_off, d_out_len, st, 0, 0, 0, 0, l4);
    // Chunks: the match kernel of chunk i+1 (memory-transaction bound) runs beside the entropy kernel of
    // chunk i (latency bound) on a second stream; the caller's stream sees everything finished.
    // (two chunks only when each still fills at least half of the match kernel's team slots: 65 536 x 64 KiB -> 2,
    // 32 768 x 128 KiB -> 1: 17.7 GB/s against 13.1 with two half-empty launches)
    u32 const team_slots = c->l3_team_slots;                 // what the device holds, whatever this context's max_slices
    u32 chunks = l4 ? 1u : c->knob.chunks ? c->knob.chunks : (n > team_slots ? 2u : 1u);       // measured: 49 152 slices (= the slots) 16.4 GB/s in one launch, 15.5 in two; 57 344: 15.2 / 16.2
    bool const tunable = !l4 && !c->knob.chunks && c->knob.autotune && n <= team_slots && 2u * n > team_slots;
    if (tunable) {
        if (c->tune_pending) {                       // the previous such batch was a trial: its whole-step time
            c->tune_pending = 0;
            float ms = 0;
            if (hipEventSynchronize(c->tune_ev[1]) == hipSuccess && hipEventElapsedTime(&ms, c->tune_ev[0], c->tune_ev[1]) == hipSuccess) c->tune_ms[c->tune_state++] = ms;
            else { (void)hipGetLastError(); c->tune_state = 2; c->tune_pick = 0; }
            if (c->tune_state == 2 && c->tune_ms[1] > 0) c->tune_pick = c->tune_ms[1] < c->tune_ms[0] ? 1u : 0u;
        }
        chunks = (c->tune_state < 2 ? (u32)c->tune_state : c->tune_pick) ? 2u : 1u;
    }
    if (chunks < 1) chunks = 1; if (chunks > KMP_MAX_CHUNKS) chunks = KMP_MAX_CHUNKS; if (chunks > n) chunks = 1;
    // One batch at a time per context (the sequence / literal / scratch workspaces and the team tables are shared): a
    // batch queued on another stream waits for the last kernel of the previous one.  (Letting the next batch's match
    // kernel start beside this batch's last entropy launch was measured and gave nothing: DESIGN.md section 8.)
    KMP_TRY(batch_begin(c, st, d_in_len, n, c->max_slice_bytes));
    if (tunable && c->tune_state < 2) HIP_TRY(hipEventRecord(c->tune_ev[0], st));
    HIP_TRY(hipMemsetAsync(c->counter, 0, 4 * KMP_MAX_CHUNKS, st));
    // Team width per batch: a small batch is bound by each slice's own chain, not by the memory system, and 8 lanes per slice
    // (7 positions per step) run it a quarter faster than 4 (64 slices: 29 ms against 39; 4 096: 43 against 49; 16 384 and up: 4
    // wins).  The tables belong to team slots, not to a width, so the choice is free per batch (KMP_TEAM_LANES or an explicit
    // team_lanes fixes it).
    int const G = (c->G == 4 && !c->team_fixed && n <= 8192u) ? 8 : c->G;
    u32 const tpw = 64 / (u32)G;
    u32 const match_flags = c->knob.match_flags, entropy_pad = c->knob.entropy_pad;   // experiments only
    u32 const per = (n + chunks - 1) / chunks;
    u32 starts[KMP_MAX_CHUNKS + 1];
    for (u32 ci = 0; ci <= chunks; ci++) starts[ci] = (ci * per < n) ? ci * per : n;
    // two chunks: the last entropy launch is the only one nothing runs beside, so the second chunk is the smaller one
    if (chunks == 2) { u32 const pm = c->knob.first_permille; if (pm >= 100 && pm <= 950) starts[1] = (u32)((u64)n * pm / 1000u) & ~63u; if (starts[1] == 0 || starts[1] >= n) starts[1] = per; }
    bool forked = false;
    for (u32 ci = 0; ci < chunks; ci++) {
        u32 const first = starts[ci], m_n = starts[ci + 1] - first;
        if (m_n == 0) continue;
        KMatchArgs m;
        m.src = (const u8*)d_src; m.in_off = d_in_off + first; m.in_len = c->len_ok + first; m.n_slices = m_n;
        m.seqs = c->seqs + (size_t)first * c->seq_cap; m.seq_cap = c->seq_cap; m.meta = c->meta + first;
        m.lits = c->lits + (size_t)first * c->lit_cap; m.lit_cap = c->lit_cap;
        m.tables = c->tables; for (int ts_ = 0; ts_ < 4; ts_++) m.tseg[ts_] = c->tseg[ts_]; m.tseg_n = c->tseg_n; m.team_epoch = c->team_epoch; m.counter = c->counter + ci; m.flags = match_flags;
        u32 max_blocks = (u32)((u64)c->match_blocks_l3 * (64u / (u32)c->G) / tpw);      // the context's team slots at this batch's width
        if (l4) {
            m.tables = c->tables4; m.tseg_n = 1; m.team_epoch = c->epoch4; m.tbl_stride = KX_TBL4_ENTRIES; m.tbl_long = KX_TBL4_LONG; m.level = 4;
            if (max_blocks > c->teams4 / tpw) max_blocks = c->teams4 / tpw;
        }
        u32 blocks = (m_n + tpw - 1) / tpw; if (blocks > max_blocks) blocks = max_blocks;
        if (c->profiling) HIP_TRY(hipEventRecord(c->evm[ci][0], st));
        KEntropyArgs e;
        e.src = (const u8*)d_src; e.in_off = d_in_off + first; e.in_len = c->len_ok + first; e.n_slices = m_n;
        e.seqs = m.seqs; e.seq_cap = c->seq_cap; e.lits = c->lits + (size_t)first * c->lit_cap; e.lit_cap = c->lit_cap; e.meta = m.meta;
        e.scratch = c->scratch + (size_t)first * c->scratch_words; e.scratch_words = c->scratch_words;
        e.dst = (u8*)d_dst; e.out_off = d_out_off + first; e.out_len = d_out_len + first; e.flags = c->knob.entropy_flags | ((m.flags & 4u) ? 8u : 0u);
#ifdef KMP_ABLATIONS
        // the ablation build's other two forms of the level-3 step: the split-phase parser, the fused kernel
        bool const fuse = c->knob.fuse && !c->knob.match_v2 && !l4 && (G == 4 || G == 8);
        if (fuse) {
            if (G == 4) hipLaunchKernelGGL(k_zstd_l3_fused<4>, dim3(blocks), dim3(64), 0, st, m, e);
            else hipLaunchKernelGGL(k_zstd_l3_fused<8>, dim3(blocks), dim3(64), 0, st, m, e);
        } else
        if (c->knob.match_v2 && !l4 && (G == 2 || G == 4 || G == 8)) {
            m.flags |= 4u;                                                // this parser never copies literals: the entropy kernel gathers them
            bool const r512 = c->knob.match_v2 == 2;
            switch (G) {
            case 2:  hipLaunchKernelGGL((k_zstd_match2<2, 256>), dim3(blocks), dim3(64), 0, st, m); break;
            case 4:  if (r512) hipLaunchKernelGGL((k_zstd_match2<4, 512>), dim3(blocks), dim3(64), 0, st, m); else hipLaunchKernelGGL((k_zstd_match2<4, 256>), dim3(blocks), dim3(64), 0, st, m); break;
            default: hipLaunchKernelGGL((k_zstd_match2<8, 512>), dim3(blocks), dim3(64), 0, st, m); break;
            }
        } else
#else
        bool const fuse = false;
#endif
        switch (G) {
        case 2:  hipLaunchKernelGGL(k_zstd_match<2>, dim3(blocks), dim3(64), 0, st, m); break;
        case 4:  hipLaunchKernelGGL(k_zstd_match<4>, dim3(blocks), dim3(64), 0, st, m); break;
        case 8:  hipLaunchKernelGGL(k_zstd_match<8>, dim3(blocks), dim3(64), 0, st, m); break;
        case 16: hipLaunchKernelGGL(k_zstd_match<16>, dim3(blocks), dim3(64), 0, st, m); break;
        case 32: hipLaunchKernelGGL(k_zstd_match<32>, dim3(blocks), dim3(64), 0, st, m); break;
        default: hipLaunchKernelGGL(k_zstd_match<64>, dim3(blocks), dim3(64), 0, st, m); break;
        }
        HIP_TRY(hipGetLastError());
        if (l4) {
            // level 4 up to 16 KiB is strategy "greedy" (ZSTD_getCParams(4, n <= 16 KiB)): those slices, which k_zstd_match has passed over, are
            // parsed by the kernels of levels 5 .. 10 (zstd_lazy.h), which pass over all the others
            KMP_TRY(lazy_workspace(c, c->max_slice_bytes < 16384u ? c->max_slice_bytes : 16384u));
            KMP_TRY(lazy_parse(c, st, d_src, d_in_off, m_n, first, 4));
        }
        HIP_TRY(hipEventRecord(c->evm[ci][1], st));
        e.flags = c->knob.entropy_flags | ((m.flags & 4u) ? 8u : 0u) | (l4 ? (4u << 12) : 0u);
        hipStream_t es = st;
        if (ci + 1 < chunks) { es = c->st2; HIP_TRY(hipStreamWaitEvent(es, c->evm[ci][1], 0)); forked = true; }
        if (c->profiling) HIP_TRY(hipEventRecord(c->eve[ci][0], es));
        if (!fuse) hipLaunchKernelGGL(k_zstd_entropy, dim3(m_n), dim3(64), entropy_pad, es, e);
        HIP_TRY(hipGetLastError());
        if (c->profiling) HIP_TRY(hipEventRecord(c->eve[ci][1], es));
    }
    if (forked) { HIP_TRY(hipEventRecord(c->ev_join, c->st2)); HIP_TRY(hipStreamWaitEvent(st, c->ev_join, 0)); }
    c->last_chunks = chunks;
    if (c->profiling) { c->ev_valid[0] = 1; c->ev_valid[1] = 1; }
    if (tunable && c->tune_state < 2) { HIP_TRY(hipEventRecord(c->tune_ev[1], st)); c->tune_pending = 1; }
    return batch_end(c, st, d_in_len, n, c->max_slice_bytes, d_out_len, c->meta);
}
// ---- a batch in pieces, each on a stream of its own ---------------------------------------------------------------------
// What a caller with HOST memory needs (every caller of the reference's boundary starts and ends there): the slices arrive over
// PCIe while earlier ones are being compressed, the frames leave while later ones still are.  One launch over the whole batch
// cannot start before its last slice is in; pieces run one after the other lose what makes this parser fast -- every slice in
// flight (16 384 slices alone take 73 ms, 65 536 together 187).  So the pieces of one batch run SIDE BY SIDE: piece p's kernels
// go to hip_streams[p] (behind whatever the caller queued there: its copy in), own the p-th part of the context's team slots
// (KMatchArgs.block_base) and of its workspace, and the device fills up as the pieces arrive.
extern "C" void kmp_batch_piece_range(uint32_t n, uint32_t pieces, uint32_t piece, uint32_t* first, uint32_t* count)
{
    if (pieces == 0) pieces = 1;
    auto cut = [&](u32 p) -> u32 { return p >= pieces ? n : (u32)(((u64)n * p / pieces) & ~63ull); };
    u32 const a = cut(piece), b = cut(piece + 1);
    if (first) *first = a;
    if (count) *count = b > a ? b - a : 0u;
}
// the three steps of a batch in pieces (kmp_coalesce.h queues a piece as soon as its copy in is queued)
int pieces_begin(kmp_batch_ctx* c, u32 pieces, void* const* hip_streams)
{
    if (!c || !hip_streams) { g_last_error = "kmp_zstd_compress_batch_pieces: null argument"; return KMP_ERR_ARG; }
    if (pieces < 1 || pieces > KMP_MAX_PIECES) { g_last_error = "kmp_zstd_compress_batch_pieces: 1 .. 8 pieces"; return KMP_ERR_ARG; }
    if (c->big) { g_last_error = "kmp_zstd_compress_batch_pieces: contexts for slices up to 128 KiB only"; return KMP_ERR_CAPACITY; }
    if (c->match_blocks_l3 / pieces == 0) { g_last_error = "kmp_zstd_compress_batch_pieces: more pieces than the context has workgroups"; return KMP_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    // every stream first waits for whatever ran on this context before (a batch, or the pieces of one)
    for (u32 p = 0; p < pieces; p++) KMP_TRY(batch_wait_previous(c, (hipStream_t)hip_streams[p]));
    return KMP_OK;
}
int piece_enqueue(kmp_batch_ctx* c, u32 p, u32 pieces, const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len, uint32_t n,
                  void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len, hipStream_t st)
{
    u32 const tpw = 64 / (u32)c->G;
    u32 const blocks_per_piece = c->match_blocks_l3 / pieces;
    u32 first = 0, m_n = 0; kmp_batch_piece_range(n, pieces, p, &first, &m_n);
    if (m_n == 0) { HIP_TRY(hipEventRecord(c->ev_piece[p], st)); return KMP_OK; }
    hipLaunchKernelGGL(k_len_guard, dim3((m_n + 255) / 256), dim3(256), 0, st, d_in_len + first, m_n, c->max_slice_bytes, c->len_ok + first, c->d_status);
    HIP_TRY(hipMemsetAsync(c->counter + p, 0, 4, st));
    KMatchArgs m;
    m.src = (const u8*)d_src; m.in_off = d_in_off + first; m.in_len = c->len_ok + first; m.n_slices = m_n;
    m.seqs = c->seqs + (size_t)first * c->seq_cap; m.seq_cap = c->seq_cap; m.meta = c->meta + first;
    m.lits = c->lits + (size_t)first * c->lit_cap; m.lit_cap = c->lit_cap;
    m.tables = c->tables; for (int ts_ = 0; ts_ < 4; ts_++) m.tseg[ts_] = c->tseg[ts_]; m.tseg_n = c->tseg_n; m.team_epoch = c->team_epoch;
    m.counter = c->counter + p; m.flags = c->knob.match_flags; m.block_base = p * blocks_per_piece;
    u32 blocks = (m_n + tpw - 1) / tpw; if (blocks > blocks_per_piece) blocks = blocks_per_piece;
    switch (c->G) {
    case 2:  hipLaunchKernelGGL(k_zstd_match<2>, dim3(blocks), dim3(64), 0, st, m); break;
    case 4:  hipLaunchKernelGGL(k_zstd_match<4>, dim3(blocks), dim3(64), 0, st, m); break;
    case 8:  hipLaunchKernelGGL(k_zstd_match<8>, dim3(blocks), dim3(64), 0, st, m); break;
    case 16: hipLaunchKernelGGL(k_zstd_match<16>, dim3(blocks), dim3(64), 0, st, m); break;
    case 32: hipLaunchKernelGGL(k_zstd_match<32>, dim3(blocks), dim3(64), 0, st, m); break;
    default: hipLaunchKernelGGL(k_zstd_match<64>, dim3(blocks), dim3(64), 0, st, m); break;
    }
    HIP_TRY(hipGetLastError());
    KEntropyArgs e;
    e.src = (const u8*)d_src; e.in_off = d_in_off + first; e.in_len = c->len_ok + first; e.n_slices = m_n;
    e.seqs = m.seqs; e.seq_cap = c->seq_cap; e.lits = m.lits; e.lit_cap = c->lit_cap; e.meta = m.meta;
    e.scratch = c->scratch + (size_t)first * c->scratch_words; e.scratch_words = c->scratch_words;
    e.dst = (u8*)d_dst; e.out_off = d_out_off + first; e.out_len = d_out_len + first; e.flags = c->knob.entropy_flags | ((m.flags & 4u) ? 8u : 0u);
    hipLaunchKernelGGL(k_zstd_entropy, dim3(m_n), dim3(64), 0, st, e);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_len_guard_finish, dim3((m_n + 255) / 256), dim3(256), 0, st, d_in_len + first, m_n, c->max_slice_bytes, d_out_len + first, c->meta + first, c->d_status);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev_piece[p], st));
    return KMP_OK;
}
void pieces_end(kmp_batch_ctx* c, u32 pieces) { c->have_done = 0; c->pieces_pending = pieces; c->last_chunks = pieces; }

extern "C" int kmp_zstd_compress_batch_pieces(kmp_batch_ctx* c, const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len, uint32_t n,
                                              void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len, uint32_t pieces, void* const* hip_streams)
{
    if (!c || !hip_streams || (n && (!d_src || !d_in_off || !d_in_len || !d_dst || !d_out_off || !d_out_len))) { g_last_error = "kmp_zstd_compress_batch_pieces: null argument"; return KMP_ERR_ARG; }
    if (n > c->max_slices) { g_last_error = "kmp_zstd_compress_batch_pieces: n exceeds the context's max_slices"; return KMP_ERR_CAPACITY; }
    KMP_TRY(pieces_begin(c, pieces, hip_streams));
    if (n == 0) return KMP_OK;
    for (u32 p = 0; p < pieces; p++) KMP_TRY(piece_enqueue(c, p, pieces, d_src, d_in_off, d_in_len, n, d_dst, d_out_off, d_out_len, (hipStream_t)hip_streams[p]));
    pieces_end(c, pieces);
    return KMP_OK;
}
// frame i of a piece straight into the caller's HOST memory (registered / pinned: h_dst is its device-visible address): what the
// dense packing + copy out + host-side memcpy do for pageable memory, in one kernel over PCIe.  A frame larger than its room is
// not written: out_len 0, status 70 (libzstd's "Destination buffer is too small").
__global__ __launch_bounds__(64) void k_scatter_frames(const u8* src, const u64* in_off, u32* len, u32 n, u8* h_dst, const u64* h_off, const u32* h_cap, u32* status)
{
    int const lane = threadIdx.x;
    for (u32 it = blockIdx.x; it < n; it += gridDim.x) {
        u32 const i = kx_xcd_chunk(it, n);
        u32 const L = len[i];
        bool const fits = L <= h_cap[i];
        if (lane == 0) status[i] = (L == 0) ? 1u : fits ? 0u : 70u;
        if (!fits) { if (lane == 0) len[i] = 0; continue; }
        const u8* s = src + in_off[i]; u8* d = h_dst + h_off[i];
        // 16 bytes per lane where both sides allow it (the strided frames start on 64-byte boundaries; the caller's offsets may not)
        u32 k = (u32)lane * 8u;
        for (; k + 8 <= L; k += 512u) kx_st64(d + k, kx_ld64(s + k));
        u32 const tail = L & ~7u;
        if (lane < (int)(L - tail)) d[tail + lane] = s[tail + lane];
    }
}
int scatter_frames(kmp_batch_ctx* c, hipStream_t st, const u8* d_src, const u64* d_in_off, u32* d_len, u32 n, u8* h_dst_dev, const u64* d_h_off, const u32* d_h_cap, u32* d_status)
{
    (void)c;
    if (!n) return KMP_OK;
    u32 const blocks = n < 16384u ? n : 16384u;
    hipLaunchKernelGGL(k_scatter_frames, dim3(blocks), dim3(64), 0, st, d_src, d_in_off, d_len, n, h_dst_dev, d_h_off, d_h_cap, d_status);
    HIP_TRY(hipGetLastError());
    return KMP_OK;
}

extern "C" int kmp_zstd_compress_batch(kmp_batch_ctx* c, const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                       uint32_t n, void* d_dst, const uint64_t* d_out_off, uint32_t* d_out_len, void* hip_stream)
{
    return zstd_compress_dfast(c, d_src, d_in_off, d_in_len, n, d_dst, d_out_off, d_out_len, hip_stream, 3);
}

// Staging for what the pre-decode kernels leave (8 bytes per sequence -- a frame of S bytes holds at most S / 3 -- and
// the literals): allocated on the first call that wants it, for as many entries as 48 GiB hold (all of them for the bench's
// batches; a larger batch goes through in pieces, one after the other, that reuse the staging).  Shared by the zstd decoder
// and inflate.
u32 env_pre_min_batch() { static u32 const v = env_u32("KMP_PRE_MIN_BATCH", 256); return v; }
void ensure_pre_staging(kmp_batch_ctx* c)
{
    if (c->pre_tried || !c->knob.decode_pre) return;
    c->pre_tried = 1;
    u32 const seq_cap = c->max_slice_bytes / 3u + 64u, blk_cap = c->max_slice_bytes / 8192u + 16u, lit_cap = c->max_slice_bytes + 64u;
    u64 const per_entry = (u64)seq_cap * 8u + (u64)lit_cap + (u64)blk_cap * (sizeof(KPreBlk) + sizeof(KPreLit)) + 64u;
    u64 fit = c->knob.decode_stage_slices ? c->knob.decode_stage_slices : (48ull << 30) / per_entry;
    if (fit > c->max_slices) fit = c->max_slices;
    if (fit >= 1024u) fit &= ~1023ull;                           // (whole workgroups of every kernel)
    u32 const ps = (u32)fit;
    c->pre_blk_cap = blk_cap;
    if (ps && (c->knob.decode_pre & 1u)) {
        if (hipMalloc((void**)&c->pre_stage, (size_t)ps * seq_cap * 8u) == hipSuccess &&
            hipMalloc((void**)&c->pre_blk, (size_t)ps * blk_cap * sizeof(KPreBlk)) == hipSuccess &&
            hipMalloc((void**)&c->pre_nblk, (size_t)ps * 4u) == hipSuccess &&
            hipMalloc((void**)&c->pre_sort, ((size_t)ps * 2u + KXP_SORT_BUCKETS) * 4u) == hipSuccess) { c->pre_seq_cap = seq_cap; c->pre_slices = ps; }
        else { (void)hipGetLastError(); (void)hipFree(c->pre_stage); (void)hipFree(c->pre_blk); (void)hipFree(c->pre_nblk); (void)hipFree(c->pre_sort); c->pre_stage = nullptr; c->pre_blk = nullptr; c->pre_nblk = nullptr; c->pre_sort = nullptr; }
    }
    if (ps && (c->knob.decode_pre & 2u)) {
        if (hipMalloc((void**)&c->pre_lits, (size_t)ps * lit_cap) == hipSuccess &&
            hipMalloc((void**)&c->pre_lit, (size_t)ps * blk_cap * sizeof(KPreLit)) == hipSuccess &&
            hipMalloc((void**)&c->pre_nlit, (size_t)ps * 4u) == hipSuccess) { c->pre_lit_cap = lit_cap; c->pre_slices = ps; }
        else { (void)hipGetLastError(); (void)hipFree(c->pre_lits); (void)hipFree(c->pre_lit); (void)hipFree(c->pre_nlit); c->pre_lits = nullptr; c->pre_lit = nullptr; c->pre_nlit = nullptr; }
    }
}

// the counting sort behind the lane-per-entry kernels' slot order (key, rank, permutation: three small launches)
int size_sort(kmp_batch_ctx* c, hipStream_t st, const u8* src, const u64* in_off, const u32* in_len, u32 m, u32* key, u32* hist, u32* perm, u32 len_shift)
{
    (void)c;
    HIP_TRY(hipMemsetAsync(hist, 0, (size_t)KXP_SORT_BUCKETS * 4u, st));
    KSeqSortArgs sa;
    sa.src = src; sa.in_off = in_off; sa.in_len = in_len; sa.n_slices = m; sa.key = key; sa.hist = hist; sa.perm = perm; sa.len_shift = len_shift;
    hipLaunchKernelGGL(k_zstd_seq_count, dim3((m + 255) / 256), dim3(256), 0, st, sa);
    hipLaunchKernelGGL(k_zstd_seq_rank, dim3(1), dim3(256), 0, st, sa);
    hipLaunchKernelGGL(k_zstd_seq_perm, dim3((m + 255) / 256), dim3(256), 0, st, sa);
    HIP_TRY(hipGetLastError());
    return KMP_OK;
}

int size_sort_keys(kmp_batch_ctx* c, hipStream_t st, u32 m, u32* key, u32* hist, u32* perm)
{
    (void)c;
    KSeqSortArgs sa;
    sa.src = nullptr; sa.in_off = nullptr; sa.in_len = nullptr; sa.n_slices = m; sa.key = key; sa.hist = hist; sa.perm = perm; sa.len_shift = 0;
    hipLaunchKernelGGL(k_zstd_seq_rank, dim3(1), dim3(256), 0, st, sa);
    hipLaunchKernelGGL(k_zstd_seq_perm, dim3((m + 255) / 256), dim3(256), 0, st, sa);
    HIP_TRY(hipGetLastError());
    return KMP_OK;
}

static int zstd_decompress_impl(kmp_batch_ctx* c, const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                uint32_t n, void* d_dst, const uint64_t* d_out_off, const uint32_t* d_out_cap,
                                uint32_t* d_out_len, uint32_t* d_status, const void* d_dict, uint32_t dict_size, void* hip_stream)
{
    if (!c || (n && (!d_src || !d_in_off || !d_in_len || !d_dst || !d_out_off || !d_out_cap || !d_out_len || !d_status))) { g_last_error = "kmp_zstd_decompress_batch: null argument"; return KMP_ERR_ARG; }
    if (n > c->max_slices) { g_last_error = "kmp_zstd_decompress_batch: n exceeds the context's max_slices"; return KMP_ERR_CAPACITY; }
    if (n == 0) return KMP_OK;
    hipStream_t const st = (hipStream_t)hip_stream;
    HIP_TRY(hipSetDevice(c->device));
    KDecodeArgs d;
    d.src = (const u8*)d_src; d.in_off = d_in_off; d.in_len = d_in_len; d.n_slices = n;
    d.dst = (u8*)d_dst; d.out_off = d_out_off; d.out_cap = d_out_cap; d.out_len = d_out_len; d.status = d_status;
    d.lits = c->lits; d.lit_cap = c->lit_cap; d.flags = c->knob.decode_flags;
    KMP_TRY(batch_begin(c, st, nullptr, n, 0));            // the decoder checks every length itself; lits is shared with the compressors
    d.dict = (const u8*)d_dict; d.dict_size = d_dict ? dict_size : 0u;
    u32 start_rep[3] = { 1, 4, 8 };
    if (d_dict && dict_size >= 8) {
        // A dictionary in zstd's own format brings tables, repeat offsets and an ID (ZSTD_loadDEntropy): its head is read back and parsed once
        // per dictionary (recognised by address, size and the head's bytes); the kernels get the content part and the parsed rest.
        u8 head[2048]; u32 const hn = dict_size < sizeof head ? dict_size : (u32)sizeof head;
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpy(head, d_dict, hn, hipMemcpyDeviceToHost));
        u64 hsh = 1469598103934665603ull; for (u32 i = 0; i < hn; i++) { hsh ^= head[i]; hsh *= 1099511628211ull; }
        if (c->ddict_ptr != d_dict || c->ddict_size != dict_size || c->ddict_hash != hsh) {
            KDictDPrior dp; size_t off = 0;
            int const formatted = cdict_parse_formatted(head, dict_size, nullptr, &off, &dp, hn);
            if (formatted < 0) { g_last_error = "kmp_zstd_decompress_batch_dict: the dictionary starts with zstd's dictionary magic but its header is damaged (libzstd: Dictionary is corrupted)"; return KMP_ERR_ARG; }
            (void)hipFree(c->d_dprior); c->d_dprior = nullptr;
            if (formatted) {
                HIP_TRY(hipMalloc((void**)&c->d_dprior, sizeof(KDictDPrior)));
                HIP_TRY(hipMemcpy(c->d_dprior, &dp, sizeof(KDictDPrior), hipMemcpyHostToDevice));
                c->ddict_off = (u32)off; c->ddict_id = dp.dictID; c->ddict_rep[0] = dp.rep[0]; c->ddict_rep[1] = dp.rep[1]; c->ddict_rep[2] = dp.rep[2];
            } else { c->ddict_off = 0; c->ddict_id = 0; }
            c->ddict_ptr = d_dict; c->ddict_size = dict_size; c->ddict_hash = hsh;
        }
        if (c->d_dprior) {
            d.dict = (const u8*)d_dict + c->ddict_off; d.dict_size = dict_size - c->ddict_off; d.dprior = c->d_dprior; d.dict_id = c->ddict_id;
            start_rep[0] = c->ddict_rep[0]; start_rep[1] = c->ddict_rep[1]; start_rep[2] = c->ddict_rep[2];
        }
    }
    // Staging for what the pre-decode kernels leave (8 bytes per sequence -- a frame of S bytes holds at most S / 3 -- and
    // the literals): allocated on the first call, for as many entries as 48 GiB hold (all of them for the bench's batches;
    // a larger batch goes through in pieces, one after the other, that reuse the staging).
    // (a small batch -- the one-frame contexts of the streaming entry points above all -- goes straight to k_zstd_decode: a quad
    // or a lane doing a whole frame's serial decode first is the old serial cost plus a second pass, and the staging is memory)
    bool const use_pre = n >= env_pre_min_batch();
    if (use_pre) ensure_pre_staging(c);
    d.pre_stage = nullptr; d.pre_seq_cap = 0; d.pre_blk = nullptr; d.pre_blk_cap = c->pre_blk_cap; d.pre_nblk = nullptr;
    d.pre_lits = nullptr; d.pre_lit_cap = 0; d.pre_lit = nullptr; d.pre_nlit = nullptr;
    if (c->profiling) HIP_TRY(hipEventRecord(c->ev[4], st));
    if (!use_pre || (!c->pre_stage && !c->pre_lits)) {
        hipLaunchKernelGGL(k_zstd_decode, dim3(n), dim3(64), c->knob.decode_pad, st, d);   // padding = occupancy experiment only
        HIP_TRY(hipGetLastError());
    } else {
        // piece by piece (one piece unless the batch is larger than the staging; KMP_DECODE_PIECES asks for more): the two
        // pre-decoders side by side -- literals on the caller's stream, sequences on the context's second one, their lane
        // slots in order of sequence count (neighbours in size share a wave) -- then k_zstd_decode
        u32 pieces = (n + c->pre_slices - 1) / c->pre_slices;
        if (c->knob.decode_pieces > pieces && n >= 8192u) pieces = c->knob.decode_pieces;
        u32 per = ((n + pieces - 1) / pieces + 63u) & ~63u;
        if (per > c->pre_slices) per = c->pre_slices;
        u32* const sort_key = c->pre_sort; u32* const sort_perm = c->pre_sort ? c->pre_sort + c->pre_slices : nullptr; u32* const sort_hist = c->pre_sort ? c->pre_sort + 2u * (size_t)c->pre_slices : nullptr;
        for (u32 first = 0; first < n; first += per) {
            u32 const m = (n - first < per) ? n - first : per;
            bool const sorted = c->pre_stage && c->knob.decode_sort != 0 && m >= 1024u;
            if (sorted) KMP_TRY(size_sort(c, st, d.src, d_in_off + first, d_in_len + first, m, sort_key, sort_hist, sort_perm, 0));
            KDecodeArgs q = d;
            q.in_off = d_in_off + first; q.in_len = d_in_len + first; q.n_slices = m;
            q.out_off = d_out_off + first; q.out_cap = d_out_cap + first; q.out_len = d_out_len + first; q.status = d_status + first;
            q.lits = c->lits + (size_t)first * c->lit_cap;
            if (c->pre_stage) {
                // (st2 starts where st stands: behind the previous piece's k_zstd_decode, which read the staging)
                HIP_TRY(hipEventRecord(c->ev_pre[0], st)); HIP_TRY(hipStreamWaitEvent(c->st2, c->ev_pre[0], 0));
                KPreArgs p;
                p.perm = sorted ? sort_perm : nullptr;
                p.src = d.src; p.in_off = q.in_off; p.in_len = q.in_len; p.n_slices = m;
                p.stage = c->pre_stage; p.seq_cap = c->pre_seq_cap; p.blk = c->pre_blk; p.blk_cap = c->pre_blk_cap; p.nblk = c->pre_nblk;
                p.rep[0] = start_rep[0]; p.rep[1] = start_rep[1]; p.rep[2] = start_rep[2];
                hipLaunchKernelGGL(k_zstd_seq_predecode, dim3((m + KXP_FRAMES - 1) / KXP_FRAMES), dim3(4 * KXP_FRAMES), 0, c->st2, p);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipEventRecord(c->ev_pre[1], c->st2));
                q.pre_stage = c->pre_stage; q.pre_seq_cap = c->pre_seq_cap; q.pre_blk = c->pre_blk; q.pre_nblk = c->pre_nblk;
            }
            if (c->pre_lits) {
                KLitArgs p;
                p.perm = sorted ? sort_perm : nullptr;
                p.src = d.src; p.in_off = q.in_off; p.in_len = q.in_len; p.n_slices = m;
                p.lits = c->pre_lits; p.lit_cap = c->pre_lit_cap; p.rec = c->pre_lit; p.blk_cap = c->pre_blk_cap; p.nrec = c->pre_nlit;
                hipLaunchKernelGGL(k_zstd_lit_predecode, dim3((m + KXL_FRAMES - 1) / KXL_FRAMES), dim3(64), 0, st, p);
                HIP_TRY(hipGetLastError());
                q.pre_lits = c->pre_lits; q.pre_lit_cap = c->pre_lit_cap; q.pre_lit = c->pre_lit; q.pre_nlit = c->pre_nlit;
            }
            if (c->pre_stage) HIP_TRY(hipStreamWaitEvent(st, c->ev_pre[1], 0));
            hipLaunchKernelGGL(k_zstd_decode, dim3(m), dim3(64), c->knob.decode_pad, st, q);
            HIP_TRY(hipGetLastError());
        }
    }
    if (c->profiling) { HIP_TRY(hipEventRecord(c->ev[5], st)); c->ev_valid[2] = 1; }
    return batch_end(c, st, nullptr, n, 0, nullptr, nullptr);
}

extern "C" int kmp_zstd_decompress_batch(kmp_batch_ctx* c, const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                         uint32_t n, void* d_dst, const uint64_t* d_out_off, const uint32_t* d_out_cap,
                                         uint32_t* d_out_len, uint32_t* d_status, void* hip_stream)
{ return zstd_decompress_impl(c, d_src, d_in_off, d_in_len, n, d_dst, d_out_off, d_out_cap, d_out_len, d_status, nullptr, 0, hip_stream); }
extern "C" int kmp_zstd_decompress_batch_dict(kmp_batch_ctx* c, const void* d_src, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                              uint32_t n, void* d_dst, const uint64_t* d_out_off, const uint32_t* d_out_cap,
                                              uint32_t* d_out_len, uint32_t* d_status, const void* d_dict, uint32_t dict_size, void* hip_stream)
{ return zstd_decompress_impl(c, d_src, d_in_off, d_in_len, n, d_dst, d_out_off, d_out_cap, d_out_len, d_status, d_dict, dict_size, hip_stream); }
extern "C" int kmp_compact_batch(kmp_batch_ctx* c, const void* d_src, const uint64_t* d_in_off, const uint32_t* d_len, uint32_t n,
                                 void* d_dst, uint64_t* d_out_off, void* hip_stream)
{
    if (!c || !d_src || !d_in_off || !d_len || !d_dst || !d_out_off) { g_last_error = "kmp_compact_batch: null argument"; return KMP_ERR_ARG; }
    hipStream_t const st = (hipStream_t)hip_stream;
    HIP_TRY(hipSetDevice(c->device));
    hipLaunchKernelGGL(k_scan_lengths, dim3(1), dim3(1024), 0, st, d_len, n, d_out_off);
    HIP_TRY(hipGetLastError());
    if (n) {
        u32 const blocks = n < 65536u ? n : 65536u;
        hipLaunchKernelGGL(k_compact, dim3(blocks), dim3(64), 0, st, (const u8*)d_src, d_in_off, d_len, n, (u8*)d_dst, (const u64*)d_out_off);
        HIP_TRY(hipGetLastError());
    }
    return KMP_OK;
}
