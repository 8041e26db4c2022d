// kmp_coalesce.h -- the batch for callers that hold HOST memory (a JVM does): host-batch engines, the pipelined bulk
// compressor and, on top of the small engine, the coalescer of the streaming entry point.  Included by kmp_stream.hip.
//
// Why: what this library measures is kmp_zstd_compress_batch over device pointers, and what the reference's Kotlin side can
// bind is the per-slice kmp_zstd_compress_stream (ZstdWrapper.kt:35-46), which ran every slice as a batch of ONE with
// synchronous copies.  Reference callers run many contexts at once (AsyncSliceTransform.kt:56-65: one transform per
// coroutine; BaseSliceTransformContentEncoder.kt:34-43: one per response).  Three things close the gap:
//   * kmp_zstd_compress_host_batch / kmp_zstd_decompress_host_batch: slices in host memory in, frames in host memory out
//     (jni/zstd/BatchWrapper.cpp binds them for direct ByteBuffers).  A small batch goes through pinned staging and one device
//     batch; a large one through the PIPELINED BULK COMPRESSOR below: the batch in pieces that run side by side on the device
//     (kmp_zstd_compress_batch_pieces), each piece's copy in, kernels and copy out overlapping the others';
//   * memory the caller has made page-stable (kmp_host_register -- a direct ByteBuffer is -- or its own pinned allocation) is
//     read and written by the device directly: no staging copy on either side (what Wrapper.cpp:92-118 borrows and releases per
//     call, borrowed once);
//   * inside kmp_zstd_compress_stream, closing calls of concurrent contexts that ask for the plain case (level 3, no
//     dictionary, the whole slice in one piece of at most 128 KiB) are gathered for a short window and compressed as ONE
//     batch; each caller gets its own frame back, bit-identical to what it would have got alone (frames do not depend
//     on batch position: tests/test_gpu_parity.py).
// kmp_host_engines_release gives everything these hold (pinned staging, device buffers, contexts) back.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

// the engines run on their own device whatever the calling thread's current one is, and leave that as they found it
struct device_guard {
    int prev; bool ok;
    explicit device_guard(int dev) : prev(0), ok(false) { if (hipGetDevice(&prev) == hipSuccess) ok = true; (void)hipSetDevice(dev); }
    ~device_guard() { if (ok) (void)hipSetDevice(prev); }
};

// The ROCm runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a
// hardware queue run in order.  The bulk compressor uses a stream per piece plus two copy streams and wants a queue for each
// (measured with 4 queues: a piece's parse waited behind another piece's whole parse, 285 ms per 65 536-slice batch instead of
// 228).  The variable is read when the runtime initialises: set here, when the library is loaded, unless the process already set
// it; a process whose runtime is up by then keeps its setting (INTEGRATION.md "Streams").
__attribute__((constructor)) static void kmp_want_hw_queues() { (void)setenv("GPU_MAX_HW_QUEUES", "16", 0); }

enum { KMP_RC_RANK_OK = 0, KMP_RC_RANK_CAPACITY = 1, KMP_RC_RANK_OTHER = 2 };
static int rc_rank(int rc) { return rc == KMP_OK ? KMP_RC_RANK_OK : rc == KMP_ERR_CAPACITY ? KMP_RC_RANK_CAPACITY : KMP_RC_RANK_OTHER; }
// keeps the most severe code (a benign "capacity" of one worker must not replace another worker's HIP or kernel error)
static void rc_raise(std::atomic<int>& acc, int rc)
{
    int cur = acc.load();
    while (rc_rank(rc) > rc_rank(cur) && !acc.compare_exchange_weak(cur, rc)) {}
}
// count items on up to `threads` host threads (the memcpy into and out of pinned staging)
template <class F> static void host_parallel(u32 threads, u32 count, F const& fn)
{
    if (threads <= 1 || count < 64) { for (u32 i = 0; i < count; i++) fn(i); return; }
    std::vector<std::thread> th;
    u32 const per = (count + threads - 1) / threads;
    for (u32 t = 0; t < threads; t++) {
        u32 const lo = t * per, hi = lo + per < count ? lo + per : count;
        if (lo >= hi) break;
        th.emplace_back([lo, hi, &fn] { for (u32 i = lo; i < hi; i++) fn(i); });
    }
    for (auto& t : th) t.join();
}

struct host_job { const u8* in; u32 len; u8* out; u32 out_cap; u32 out_len; u32 status; std::vector<u8>* out_vec; bool done; int batch_rc; };

struct host_engine {
    int device; kmp_batch_ctx* batch; u32 cap_slices; u32 slice_cap; size_t stride;
    u8 *h_in, *h_out; u64 *h_off, *h_doff; u32 *h_len, *h_cap, *h_st;             // pinned
    u8 *d_in, *d_out, *d_dense; u64 *d_off, *d_ooff, *d_doff; u32 *d_len, *d_olen, *d_cap, *d_st;
    size_t in_bytes, out_bytes;
    hipStream_t st;
    std::mutex run_mutex;                                                            // one run at a time (the staging is shared)
};

static void host_engine_free(host_engine* e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->batch) kmp_batch_destroy(e->batch);
    (void)hipHostFree(e->h_in); (void)hipHostFree(e->h_out); (void)hipHostFree(e->h_off); (void)hipHostFree(e->h_doff); (void)hipHostFree(e->h_len); (void)hipHostFree(e->h_cap); (void)hipHostFree(e->h_st);
    (void)hipFree(e->d_in); (void)hipFree(e->d_out); (void)hipFree(e->d_dense); (void)hipFree(e->d_off); (void)hipFree(e->d_ooff); (void)hipFree(e->d_doff);
    (void)hipFree(e->d_len); (void)hipFree(e->d_olen); (void)hipFree(e->d_cap); (void)hipFree(e->d_st);
    if (e->st) (void)hipStreamDestroy(e->st);
    delete e;
}

// Engines per device and process, made on first use.  Slot 0: batches of up to KMP_HOST_BATCH_SLICES (default 1024) slices of up to
// 128 KiB -- the coalescer's and the small calls'.  Slots 1 ..: the bulk engines of large DECODE batches (KMP_HOST_BULK_SLICES,
// default 8 192 entries each: 1.1 GiB of pinned staging each way), one per worker thread of such a call.  A creation that failed
// (pinned or device memory short) is remembered for a few seconds: the callers fall back to their solo paths without retrying
// allocations of several hundred MiB under the mutex on every call.
enum { KMP_HOST_ENGINES = 5, KMP_HOST_DEVICES = 16 };
struct engine_table {
    std::mutex m; host_engine* engines[KMP_HOST_DEVICES][KMP_HOST_ENGINES];
    std::chrono::steady_clock::time_point failed_at[KMP_HOST_DEVICES][KMP_HOST_ENGINES]; bool failed[KMP_HOST_DEVICES][KMP_HOST_ENGINES];
};
static engine_table& engine_tab() { static engine_table t = {}; return t; }
static host_engine* host_engine_get(int device, int slot = 0)
{
    engine_table& T = engine_tab();
    if (device < 0 || device >= KMP_HOST_DEVICES || slot < 0 || slot >= KMP_HOST_ENGINES) return nullptr;
    std::lock_guard<std::mutex> g(T.m);
    if (T.engines[device][slot]) return T.engines[device][slot];
    if (T.failed[device][slot] && std::chrono::steady_clock::now() - T.failed_at[device][slot] < std::chrono::seconds(5)) { g_last_error = "host batch engine: allocation failed a moment ago"; return nullptr; }
    device_guard const on(device);
    host_engine* e = new (std::nothrow) host_engine();
    if (!e) return nullptr;
    e->device = device; e->batch = nullptr; e->st = nullptr;
    e->h_in = e->h_out = nullptr; e->h_off = e->h_doff = nullptr; e->h_len = e->h_cap = e->h_st = nullptr;
    e->d_in = e->d_out = e->d_dense = nullptr; e->d_off = e->d_ooff = e->d_doff = nullptr; e->d_len = e->d_olen = e->d_cap = e->d_st = nullptr;
    e->cap_slices = slot == 0 ? env_u32("KMP_HOST_BATCH_SLICES", 1024) : env_u32("KMP_HOST_BULK_SLICES", 8192);
    if (e->cap_slices < 16) e->cap_slices = 16; if (e->cap_slices > 65536) e->cap_slices = 65536;
    e->slice_cap = KMP_MAX_SLICE_BYTES;
    e->stride = (kmp_zstd_compress_bound(e->slice_cap) + 8 + 63) & ~(size_t)63;
    size_t const n = e->cap_slices;
    e->in_bytes = n * ((size_t)e->slice_cap + 64); e->out_bytes = n * e->stride;
    bool ok = batch_create_packed(&e->batch, device, e->cap_slices, e->slice_cap) == KMP_OK;       // (its arena packed: an engine is not the place to spend a span on)
    ok = ok && hipStreamCreateWithFlags(&e->st, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&e->h_in, e->in_bytes) == hipSuccess && hipHostMalloc((void**)&e->h_out, e->out_bytes) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&e->h_off, (n + 1) * 8) == hipSuccess && hipHostMalloc((void**)&e->h_doff, (n + 1) * 8) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&e->h_len, n * 4) == hipSuccess && hipHostMalloc((void**)&e->h_cap, n * 4) == hipSuccess && hipHostMalloc((void**)&e->h_st, n * 4) == hipSuccess;
    ok = ok && hipMalloc((void**)&e->d_in, e->in_bytes) == hipSuccess && hipMalloc((void**)&e->d_out, e->out_bytes + 64) == hipSuccess && hipMalloc((void**)&e->d_dense, e->out_bytes + 64) == hipSuccess;
    ok = ok && hipMalloc((void**)&e->d_off, (n + 1) * 8) == hipSuccess && hipMalloc((void**)&e->d_ooff, (n + 1) * 8) == hipSuccess && hipMalloc((void**)&e->d_doff, (n + 1) * 8) == hipSuccess;
    ok = ok && hipMalloc((void**)&e->d_len, n * 4) == hipSuccess && hipMalloc((void**)&e->d_olen, n * 4) == hipSuccess && hipMalloc((void**)&e->d_cap, n * 4) == hipSuccess && hipMalloc((void**)&e->d_st, n * 4) == hipSuccess;
    if (ok) {
        // the strided output offsets never change
        for (size_t i = 0; i <= n; i++) e->h_doff[i] = i * e->stride;
        ok = hipMemcpy(e->d_ooff, e->h_doff, (n + 1) * 8, hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) {
        (void)hipGetLastError(); host_engine_free(e); g_last_error = "host batch engine: allocation failed";
        T.failed[device][slot] = true; T.failed_at[device][slot] = std::chrono::steady_clock::now();
        return nullptr;
    }
    T.failed[device][slot] = false;
    T.engines[device][slot] = e;
    return e;
}

// Compresses jobs[0 .. n) (n <= cap_slices, every len <= 128 KiB) at any level the batch call serves: one H2D copy, the device
// batch, the dense packing, two D2H copies.  A frame goes to job.out (out_cap bytes; status 70 = too small) or to job.out_vec;
// status 1 = the device refused the slice (level 4's greedy size classes: out_len 0, the call returns KMP_ERR_CAPACITY).
static int host_engine_compress(host_engine* e, host_job* jobs, u32 n, int level, bool* ran = nullptr)
{
    if (ran) *ran = false;
    std::lock_guard<std::mutex> g(e->run_mutex);
    device_guard const on(e->device);
    size_t pos = 0;
    for (u32 i = 0; i < n; i++) {
        if (jobs[i].len > e->slice_cap) { g_last_error = "host batch: a slice is larger than 128 KiB"; return KMP_ERR_CAPACITY; }
        e->h_off[i] = pos; e->h_len[i] = jobs[i].len;
        if (jobs[i].len) memcpy(e->h_in + pos, jobs[i].in, jobs[i].len);
        pos += ((size_t)jobs[i].len + 63) & ~(size_t)63;
    }
    if (pos) HIP_TRY(hipMemcpyAsync(e->d_in, e->h_in, pos, hipMemcpyHostToDevice, e->st));
    HIP_TRY(hipMemcpyAsync(e->d_off, e->h_off, (size_t)n * 8, hipMemcpyHostToDevice, e->st));
    HIP_TRY(hipMemcpyAsync(e->d_len, e->h_len, (size_t)n * 4, hipMemcpyHostToDevice, e->st));
    KMP_TRY(kmp_zstd_compress_batch_level(e->batch, e->d_in, e->d_off, e->d_len, n, e->d_out, e->d_ooff, e->d_olen, level, e->st));
    KMP_TRY(kmp_compact_batch(e->batch, e->d_out, e->d_ooff, e->d_olen, n, e->d_dense, e->d_doff, e->st));
    HIP_TRY(hipMemcpyAsync(e->h_off, e->d_doff, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost, e->st));
    HIP_TRY(hipStreamSynchronize(e->st));
    size_t const total = (size_t)e->h_off[n];
    if (total > e->out_bytes) { g_last_error = "host batch: frames exceed the staging"; return KMP_ERR_KERNEL; }
    if (total) HIP_TRY(hipMemcpyAsync(e->h_out, e->d_dense, total, hipMemcpyDeviceToHost, e->st));
    HIP_TRY(hipStreamSynchronize(e->st));
    if (ran) *ran = true;                                                // from here on every job gets its own outcome
    u32 bits = 0;
    (void)kmp_batch_status(e->batch, &bits, e->st);                       // (the bits of refused slices: cleared here, reported per slice below)
    int rc = KMP_OK;
    for (u32 i = 0; i < n; i++) {
        size_t const a = (size_t)e->h_off[i], b = (size_t)e->h_off[i + 1];
        u32 const fl = (u32)(b - a);
        jobs[i].status = 0; jobs[i].out_len = 0;
        if (fl == 0) {                                                   // a real frame is never empty: the device refused the slice or a guard tripped
            jobs[i].status = 1;
            if (bits & (KMP_STATUS_LEVEL_SIZE | KMP_STATUS_SLICE_TOO_LARGE)) { if (rc == KMP_OK) rc = KMP_ERR_CAPACITY; } else rc = KMP_ERR_KERNEL;
            continue;
        }
        if (jobs[i].out_vec) { jobs[i].out_vec->assign(e->h_out + a, e->h_out + b); jobs[i].out_len = fl; }
        else if (fl > jobs[i].out_cap) { jobs[i].status = 70; if (rc == KMP_OK) rc = KMP_ERR_CAPACITY; }
        else { memcpy(jobs[i].out, e->h_out + a, fl); jobs[i].out_len = fl; }
    }
    if (rc == KMP_ERR_CAPACITY) g_last_error = (bits & KMP_STATUS_LEVEL_SIZE) ? "host batch: levels 9 and 10 are served for slices above 16 KiB (smaller ones are another strategy there): those have out_len 0"
                                                                               : "host batch: an output region is smaller than its frame (kmp_zstd_compress_bound)";
    if (rc == KMP_ERR_KERNEL) g_last_error = "host batch: a slice came back without a frame";
    return rc;
}

// Decompresses jobs[0 .. n): frames in, content out (out_cap = room; status = libzstd's error number, 0 = fine).
static int host_engine_decompress(host_engine* e, host_job* jobs, u32 n)
{
    std::lock_guard<std::mutex> g(e->run_mutex);
    device_guard const on(e->device);
    size_t pos = 0, opos = 0;
    for (u32 i = 0; i < n; i++) {
        if (jobs[i].len > e->stride || jobs[i].out_cap > e->slice_cap) { g_last_error = "host batch: a frame or its content is larger than 128 KiB"; return KMP_ERR_CAPACITY; }
        e->h_off[i] = pos; e->h_len[i] = jobs[i].len;
        if (jobs[i].len) memcpy(e->h_in + pos, jobs[i].in, jobs[i].len);
        pos += ((size_t)jobs[i].len + 63) & ~(size_t)63;
        e->h_doff[i] = opos; e->h_cap[i] = jobs[i].out_cap;
        opos += ((size_t)jobs[i].out_cap + 63) & ~(size_t)63;
    }
    // (the strided offsets of the compress side live in d_ooff, untouched; h_doff is borrowed here and put back on every way out)
    struct restore { host_engine* e; ~restore() { for (size_t i = 0; i <= e->cap_slices; i++) e->h_doff[i] = i * e->stride; } } const put_back = { e };
    if (pos > e->in_bytes || opos > e->out_bytes) { g_last_error = "host batch: the batch exceeds the staging"; return KMP_ERR_CAPACITY; }
    if (pos) HIP_TRY(hipMemcpyAsync(e->d_in, e->h_in, pos, hipMemcpyHostToDevice, e->st));
    HIP_TRY(hipMemcpyAsync(e->d_off, e->h_off, (size_t)n * 8, hipMemcpyHostToDevice, e->st));
    HIP_TRY(hipMemcpyAsync(e->d_len, e->h_len, (size_t)n * 4, hipMemcpyHostToDevice, e->st));
    HIP_TRY(hipMemcpyAsync(e->d_doff, e->h_doff, (size_t)n * 8, hipMemcpyHostToDevice, e->st));
    HIP_TRY(hipMemcpyAsync(e->d_cap, e->h_cap, (size_t)n * 4, hipMemcpyHostToDevice, e->st));
    KMP_TRY(kmp_zstd_decompress_batch(e->batch, e->d_in, e->d_off, e->d_len, n, e->d_dense, e->d_doff, e->d_cap, e->d_olen, e->d_st, e->st));
    HIP_TRY(hipMemcpyAsync(e->h_len, e->d_olen, (size_t)n * 4, hipMemcpyDeviceToHost, e->st));
    HIP_TRY(hipMemcpyAsync(e->h_st, e->d_st, (size_t)n * 4, hipMemcpyDeviceToHost, e->st));
    if (opos) HIP_TRY(hipMemcpyAsync(e->h_out, e->d_dense, opos, hipMemcpyDeviceToHost, e->st));
    HIP_TRY(hipStreamSynchronize(e->st));
    for (u32 i = 0; i < n; i++) {
        jobs[i].status = e->h_st[i]; jobs[i].out_len = e->h_st[i] ? 0u : e->h_len[i];
        if (jobs[i].out_len) memcpy(jobs[i].out, e->h_out + e->h_doff[i], jobs[i].out_len);
    }
    return KMP_OK;
}

// ---- page-stable caller memory ------------------------------------------------------------------------------------------
// kmp_host_register pins [ptr, ptr + bytes) and maps it for the device (hipHostRegister); the host-batch calls then read slices
// from it and write frames into it directly.  For memory that outlives many calls: a direct ByteBuffer a Kotlin caller reuses,
// a native arena.  The caller unregisters before it frees the memory.
extern "C" int kmp_host_register(void* ptr, size_t bytes)
{
    if (!ptr || !bytes) { g_last_error = "kmp_host_register: bad argument"; return KMP_ERR_ARG; }
    HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterPortable | hipHostRegisterMapped));
    return KMP_OK;
}
extern "C" int kmp_host_unregister(void* ptr)
{
    if (!ptr) { g_last_error = "kmp_host_unregister: bad argument"; return KMP_ERR_ARG; }
    HIP_TRY(hipHostUnregister(ptr));
    return KMP_OK;
}
// the device-visible address of [p, p + bytes) when the whole range is pinned host memory (registered here or allocated pinned by
// the caller), else null
static u8* host_range_device_ptr(const void* p, size_t bytes)
{
    if (!p || !bytes) return nullptr;
    hipPointerAttribute_t a0, a1;
    if (hipPointerGetAttributes(&a0, p) != hipSuccess || hipPointerGetAttributes(&a1, (const u8*)p + bytes - 1) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (a0.type != hipMemoryTypeHost || a1.type != hipMemoryTypeHost) return nullptr;
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, const_cast<void*>(p), 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return (u8*)d;
}

// ---- the pipelined bulk compressor ------------------------------------------------------------------------------------------
// One per device, made by the first large kmp_zstd_compress_host_batch call for that call's shape (slice size class, slice
// count; a later call that needs more rebuilds it): a batch context, device buffers for the slices, the strided and the dense
// frames, a stream per piece and one for each copy direction.  A batch of n slices goes through as KMP_MAX_PIECES pieces:
//   copy in  (stream s_in, piece after piece: copies on different streams would share the link and all arrive at the end)
//            straight out of the caller's memory when that is pinned and the piece's slices lie one behind the other there,
//            else through pinned staging filled by KMP_HOST_BULK_WORKERS host threads;
//   kernels  (stream s_piece[p], behind the piece's copy: piece_enqueue -- the pieces share the context and run side by side);
//   frames   straight into the caller's memory when that is pinned (k_scatter_frames: one kernel over PCIe), else dense packing
//            on the piece's stream, copy out on s_out, and the worker threads hand the frames out.
// 65 536 x 64 KiB: 11.4 GB/s through the staged engines of round 3; pinned in and out, pipelined: see bench.py's host_batch_bulk.
struct bulk_pipe {
    int device; kmp_batch_ctx* batch; u32 cap_slices, slice_cap, pieces; size_t stride, in_bytes, out_bytes;
    hipStream_t s_in, s_out, s_piece[KMP_MAX_PIECES];
    hipEvent_t arrived[KMP_MAX_PIECES], packed[KMP_MAX_PIECES], copied[KMP_MAX_PIECES];
    u8 *d_in, *d_out, *d_dense; u64 *d_off, *d_ooff, *d_doff, *d_hoff; u32 *d_len, *d_olen, *d_cap, *d_st;
    u8 *h_in, *h_out;                                         // pinned staging, made when a call first needs it
    u64 *h_off, *h_doff, *h_tot; u32 *h_len, *h_st;           // pinned
    std::mutex run_mutex;
};
static void bulk_pipe_free(bulk_pipe* b)
{
    if (!b) return;
    (void)hipSetDevice(b->device);
    if (b->batch) kmp_batch_destroy(b->batch);
    (void)hipFree(b->d_in); (void)hipFree(b->d_out); (void)hipFree(b->d_dense); (void)hipFree(b->d_off); (void)hipFree(b->d_ooff); (void)hipFree(b->d_doff); (void)hipFree(b->d_hoff);
    (void)hipFree(b->d_len); (void)hipFree(b->d_olen); (void)hipFree(b->d_cap); (void)hipFree(b->d_st);
    (void)hipHostFree(b->h_in); (void)hipHostFree(b->h_out); (void)hipHostFree(b->h_off); (void)hipHostFree(b->h_doff); (void)hipHostFree(b->h_tot); (void)hipHostFree(b->h_len); (void)hipHostFree(b->h_st);
    if (b->s_in) (void)hipStreamDestroy(b->s_in);
    if (b->s_out) (void)hipStreamDestroy(b->s_out);
    for (int i = 0; i < KMP_MAX_PIECES; i++) {
        if (b->s_piece[i]) (void)hipStreamDestroy(b->s_piece[i]);
        if (b->arrived[i]) (void)hipEventDestroy(b->arrived[i]);
        if (b->packed[i]) (void)hipEventDestroy(b->packed[i]);
        if (b->copied[i]) (void)hipEventDestroy(b->copied[i]);
    }
    delete b;
}
struct bulk_table { std::mutex m; bulk_pipe* pipe[KMP_HOST_DEVICES]; std::mutex call[KMP_HOST_DEVICES]; };      // call[d]: one bulk call per device at a time (it may rebuild the pipe)
static bulk_table& bulk_tab() { static bulk_table t = {}; return t; }
// the pipe of `device`, large enough for n slices of up to max_len bytes (rebuilt when it is not)
static bulk_pipe* bulk_pipe_get(int device, u32 n, u32 max_len)
{
    bulk_table& T = bulk_tab();
    if (device < 0 || device >= KMP_HOST_DEVICES) return nullptr;
    std::lock_guard<std::mutex> g(T.m);
    u32 const limit = KMP_MAX_PIECES * env_u32("KMP_HOST_BULK_SLICES", 8192);            // slices of one pass (larger batches: several passes)
    u32 want_n = 1024; while (want_n < n && want_n < limit) want_n <<= 1; if (want_n > limit) want_n = limit;
    u32 const want_len = max_len <= 16384u ? 16384u : max_len <= 65536u ? 65536u : KMP_MAX_SLICE_BYTES;
    bulk_pipe* b = T.pipe[device];
    if (b && b->cap_slices >= want_n && b->slice_cap >= want_len) return b;
    if (b) { std::lock_guard<std::mutex> r(b->run_mutex); }                            // (nobody is inside it: calls hold run_mutex while they run)
    bulk_pipe_free(b); T.pipe[device] = nullptr;
    device_guard const on(device);
    b = new (std::nothrow) bulk_pipe();          // (value-initialised: every pointer null)
    if (!b) return nullptr;
    b->device = device; b->cap_slices = want_n; b->slice_cap = want_len; b->pieces = KMP_MAX_PIECES;
    b->stride = (kmp_zstd_compress_bound(want_len) + 8 + 63) & ~(size_t)63;
    size_t const ns = want_n;
    b->in_bytes = ns * ((size_t)want_len + 64); b->out_bytes = ns * b->stride;
    // (a large pipe gets a span for its tables like any large context -- the parser is 11 % faster over it -- but a modest one: a
    // quarter of what is free, and the packed form below 72 GiB of span, where a span buys nothing)
    bool ok;
    {
        size_t fr = 0, tot = 0; if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); fr = 0; }
        int span = (int)((fr / 4) >> 30); if (span > (int)env_u32("KMP_TABLE_SPAN_GIB", 100)) span = (int)env_u32("KMP_TABLE_SPAN_GIB", 100); if (span < 72) span = 0;
        kmp_batch_options o; o.struct_bytes = sizeof o; o.team_lanes = 0; o.table_span_gib = span; o.table_retry = 0;
        ok = kmp_batch_create_ex(&b->batch, device, want_n, want_len, &o) == KMP_OK;
    }
    ok = ok && hipStreamCreateWithFlags(&b->s_in, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&b->s_out, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ok && i < KMP_MAX_PIECES; i++)
        ok = hipStreamCreateWithFlags(&b->s_piece[i], hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&b->arrived[i], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&b->packed[i], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&b->copied[i], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipMalloc((void**)&b->d_in, b->in_bytes) == hipSuccess && hipMalloc((void**)&b->d_out, b->out_bytes + 64) == hipSuccess && hipMalloc((void**)&b->d_dense, b->out_bytes + 64) == hipSuccess;
    ok = ok && hipMalloc((void**)&b->d_off, (ns + 1) * 8) == hipSuccess && hipMalloc((void**)&b->d_ooff, (ns + 1) * 8) == hipSuccess && hipMalloc((void**)&b->d_doff, (ns + KMP_MAX_PIECES) * 8) == hipSuccess && hipMalloc((void**)&b->d_hoff, (ns + 1) * 8) == hipSuccess;
    ok = ok && hipMalloc((void**)&b->d_len, ns * 4) == hipSuccess && hipMalloc((void**)&b->d_olen, ns * 4) == hipSuccess && hipMalloc((void**)&b->d_cap, ns * 4) == hipSuccess && hipMalloc((void**)&b->d_st, ns * 4) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&b->h_off, (ns + 1) * 8) == hipSuccess && hipHostMalloc((void**)&b->h_doff, (ns + KMP_MAX_PIECES) * 8) == hipSuccess && hipHostMalloc((void**)&b->h_tot, 64 * 8) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&b->h_len, ns * 4) == hipSuccess && hipHostMalloc((void**)&b->h_st, ns * 4) == hipSuccess;
    if (ok) {
        for (size_t i = 0; i <= ns; i++) b->h_off[i] = i * b->stride;
        ok = hipMemcpy(b->d_ooff, b->h_off, (ns + 1) * 8, hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) { (void)hipGetLastError(); bulk_pipe_free(b); g_last_error = "host batch: the bulk compressor could not be made (memory)"; return nullptr; }
    T.pipe[device] = b;
    return b;
}

// one pass: n <= cap_slices slices.  src_dev / dst_dev: device-visible addresses of the caller's whole source / destination ranges
// when those are pinned (else null: staging)
static int bulk_pipe_run(bulk_pipe* b, int level, const u8* h_src, const u8* src_dev, const uint64_t* in_off, const uint32_t* in_len, u32 n,
                         u8* h_dst, u8* dst_dev, const uint64_t* out_off, const uint32_t* out_cap, uint32_t* out_len, u32 workers)
{
    std::lock_guard<std::mutex> g(b->run_mutex);
    device_guard const on(b->device);
    u32 const P = b->pieces;
    // where every slice sits in d_in: a piece whose slices lie one behind the other in the caller's pinned memory keeps their
    // spacing (one copy of the range), every other piece is packed at 64-byte steps (through staging)
    bool direct_in[KMP_MAX_PIECES]; size_t base[KMP_MAX_PIECES + 1], span[KMP_MAX_PIECES];
    size_t pos = 0; bool need_stage_in = false;
    for (u32 p = 0; p < P; p++) {
        u32 first = 0, cnt = 0; kmp_batch_piece_range(n, P, p, &first, &cnt);
        base[p] = pos; span[p] = 0; direct_in[p] = false;
        if (!cnt) continue;
        size_t const room = (size_t)cnt * ((size_t)b->slice_cap + 64);
        bool mono = src_dev != nullptr;
        for (u32 i = first; mono && i < first + cnt; i++) {
            if (in_len[i] > b->slice_cap) { g_last_error = "host batch: a slice is larger than the bulk compressor was made for"; return KMP_ERR_CAPACITY; }
            if (i > first && in_off[i] < in_off[i - 1] + in_len[i - 1]) mono = false;
        }
        size_t const range = mono ? (size_t)(in_off[first + cnt - 1] + in_len[first + cnt - 1] - in_off[first]) : 0;
        if (mono && range <= room) {
            direct_in[p] = true; span[p] = range;
            for (u32 i = first; i < first + cnt; i++) { b->h_off[i] = pos + (size_t)(in_off[i] - in_off[first]); b->h_len[i] = in_len[i]; }
        } else {
            need_stage_in = true;
            size_t q = pos;
            for (u32 i = first; i < first + cnt; i++) {
                if (in_len[i] > b->slice_cap) { g_last_error = "host batch: a slice is larger than the bulk compressor was made for"; return KMP_ERR_CAPACITY; }
                b->h_off[i] = q; b->h_len[i] = in_len[i]; q += ((size_t)in_len[i] + 63) & ~(size_t)63;
            }
            span[p] = q - pos;
        }
        pos += (span[p] + 63) & ~(size_t)63;
    }
    base[P] = pos;
    if (pos > b->in_bytes) { g_last_error = "host batch: the batch exceeds the bulk compressor's staging"; return KMP_ERR_CAPACITY; }
    if (need_stage_in && !b->h_in && hipHostMalloc((void**)&b->h_in, b->in_bytes) != hipSuccess) { (void)hipGetLastError(); b->h_in = nullptr; g_last_error = "host batch: no pinned memory for the input staging"; return KMP_ERR_HIP; }
    if (!dst_dev && !b->h_out && hipHostMalloc((void**)&b->h_out, b->out_bytes) != hipSuccess) { (void)hipGetLastError(); b->h_out = nullptr; g_last_error = "host batch: no pinned memory for the output staging"; return KMP_ERR_HIP; }
    // the small tables first (the kernels of every piece read them)
    HIP_TRY(hipMemcpyAsync(b->d_off, b->h_off, (size_t)n * 8, hipMemcpyHostToDevice, b->s_in));
    HIP_TRY(hipMemcpyAsync(b->d_len, b->h_len, (size_t)n * 4, hipMemcpyHostToDevice, b->s_in));
    if (dst_dev) {
        HIP_TRY(hipMemcpyAsync(b->d_hoff, out_off, (size_t)n * 8, hipMemcpyHostToDevice, b->s_in));      // (pageable sources: the runtime stages them before it returns)
        HIP_TRY(hipMemcpyAsync(b->d_cap, out_cap, (size_t)n * 4, hipMemcpyHostToDevice, b->s_in));
    }
    void* streams[KMP_MAX_PIECES]; for (u32 p = 0; p < P; p++) streams[p] = b->s_piece[p];
    KMP_TRY(pieces_begin(b->batch, P, streams));
    for (u32 p = 0; p < P; p++) {
        u32 first = 0, cnt = 0; kmp_batch_piece_range(n, P, p, &first, &cnt);
        if (cnt && span[p]) {
            if (direct_in[p]) HIP_TRY(hipMemcpyAsync(b->d_in + base[p], src_dev + in_off[first], span[p], hipMemcpyHostToDevice, b->s_in));
            else {
                host_parallel(workers, cnt, [&](u32 k) { u32 const i = first + k; if (in_len[i]) memcpy(b->h_in + b->h_off[i], h_src + in_off[i], in_len[i]); });
                HIP_TRY(hipMemcpyAsync(b->d_in + base[p], b->h_in + base[p], span[p], hipMemcpyHostToDevice, b->s_in));
            }
        }
        HIP_TRY(hipEventRecord(b->arrived[p], b->s_in));
        HIP_TRY(hipStreamWaitEvent(b->s_piece[p], b->arrived[p], 0));
        if (level == 3) KMP_TRY(piece_enqueue(b->batch, p, P, b->d_in, b->d_off, b->d_len, n, b->d_out, b->d_ooff, b->d_olen, b->s_piece[p]));
        if (!cnt) { HIP_TRY(hipEventRecord(b->packed[p], b->s_piece[p])); continue; }
        if (dst_dev) KMP_TRY(scatter_frames(b->batch, b->s_piece[p], b->d_out, b->d_ooff + first, b->d_olen + first, cnt, dst_dev, b->d_hoff + first, b->d_cap + first, b->d_st + first));
        else {
            // the piece's frames, densely packed, into its own part of d_dense (worst-case placement), the offsets behind those of the pieces before
            KMP_TRY(kmp_compact_batch(b->batch, b->d_out, b->d_ooff + first, b->d_olen + first, cnt, b->d_dense + (size_t)first * b->stride, b->d_doff + first + p, b->s_piece[p]));
            HIP_TRY(hipMemcpyAsync(b->h_doff + first + p, b->d_doff + first + p, ((size_t)cnt + 1) * 8, hipMemcpyDeviceToHost, b->s_piece[p]));
        }
        HIP_TRY(hipEventRecord(b->packed[p], b->s_piece[p]));
    }
    pieces_end(b->batch, P);
    int rc = KMP_OK;
    if (dst_dev) {
        // everything is queued: the frames land in the caller's memory as the pieces finish
        for (u32 p = 0; p < P; p++) HIP_TRY(hipStreamWaitEvent(b->s_out, b->packed[p], 0));
        HIP_TRY(hipMemcpyAsync(b->h_len, b->d_olen, (size_t)n * 4, hipMemcpyDeviceToHost, b->s_out));
        HIP_TRY(hipMemcpyAsync(b->h_st, b->d_st, (size_t)n * 4, hipMemcpyDeviceToHost, b->s_out));
        HIP_TRY(hipStreamSynchronize(b->s_out));
        for (u32 i = 0; i < n; i++) {
            out_len[i] = b->h_len[i];
            if (b->h_st[i] == 70u) { if (rc == KMP_OK) rc = KMP_ERR_CAPACITY; }
            else if (b->h_st[i]) rc = KMP_ERR_KERNEL;
        }
    } else {
        // piece after piece: its size is known -> copy its dense frames out -> (while the next piece's copy runs) hand the frames out
        auto hand_out = [&](u32 p) -> int {
            u32 first = 0, cnt = 0; kmp_batch_piece_range(n, P, p, &first, &cnt);
            if (!cnt) return KMP_OK;
            HIP_TRY(hipEventSynchronize(b->copied[p]));
            const u64* const off = b->h_doff + first + p; const u8* const from = b->h_out + (size_t)first * b->stride;
            std::atomic<int> prc(KMP_OK);
            host_parallel(workers, cnt, [&](u32 k) {
                u32 const i = first + k; u32 const fl = (u32)(off[k + 1] - off[k]);
                out_len[i] = 0;
                if (fl == 0) { rc_raise(prc, KMP_ERR_KERNEL); return; }
                if (fl > out_cap[i]) { rc_raise(prc, KMP_ERR_CAPACITY); return; }
                memcpy(h_dst + out_off[i], from + off[k], fl); out_len[i] = fl;
            });
            if (rc_rank(prc.load()) > rc_rank(rc)) rc = prc.load();
            return KMP_OK;
        };
        for (u32 p = 0; p < P; p++) {
            u32 first = 0, cnt = 0; kmp_batch_piece_range(n, P, p, &first, &cnt);
            if (cnt) {
                HIP_TRY(hipEventSynchronize(b->packed[p]));
                size_t const tot = (size_t)b->h_doff[first + p + cnt];
                if (tot > (size_t)cnt * b->stride) { g_last_error = "host batch: frames exceed the staging"; return KMP_ERR_KERNEL; }
                if (tot) HIP_TRY(hipMemcpyAsync(b->h_out + (size_t)first * b->stride, b->d_dense + (size_t)first * b->stride, tot, hipMemcpyDeviceToHost, b->s_out));
                HIP_TRY(hipEventRecord(b->copied[p], b->s_out));
            }
            if (p > 0) KMP_TRY(hand_out(p - 1));
        }
        KMP_TRY(hand_out(P - 1));
    }
    if (rc == KMP_ERR_CAPACITY) g_last_error = "host batch: an output region is smaller than its frame (kmp_zstd_compress_bound)";
    if (rc == KMP_ERR_KERNEL) g_last_error = "host batch: a slice came back without a frame";
    return rc;
}

/* Levels -131072 .. 10 (0 = 3): what kmp_zstd_compress_batch_level serves for one-block slices; in a batch of level 9 or 10 the slices of its
 * "btlazy2" size class (16 KiB or less) come back with out_len 0 and the call returns KMP_ERR_CAPACITY after every other slice has
 * been compressed.  out_len is written for every slice, whatever the return value. */
extern "C" int kmp_zstd_compress_host_batch(int device, int level, const void* h_src, const uint64_t* in_off, const uint32_t* in_len, uint32_t n,
                                            void* h_dst, const uint64_t* out_off, const uint32_t* out_cap, uint32_t* out_len)
{
    if (n && (!h_src || !in_off || !in_len || !h_dst || !out_off || !out_cap || !out_len)) { g_last_error = "kmp_zstd_compress_host_batch: null argument"; return KMP_ERR_ARG; }
    if (level == 0) level = 3;
    if (level < -131072 || level > 10) { g_last_error = "kmp_zstd_compress_host_batch: levels -131072 .. 10 are served (9 and 10: slices above 16 KiB)"; return KMP_ERR_ARG; }
    for (u32 i = 0; i < n; i++) out_len[i] = 0;
    if (n == 0) return KMP_OK;
    u32 const workers_wanted = env_u32("KMP_HOST_BULK_WORKERS", 4);
    u32 const workers = workers_wanted < 1 ? 1u : workers_wanted > 16 ? 16u : workers_wanted;
    u32 const small_cap = env_u32("KMP_HOST_BATCH_SLICES", 1024);
    // A large level-3 batch goes through the pipelined bulk compressor; the other levels and small batches through the staged engine.
    if (level == 3 && n > 2u * small_cap && workers_wanted >= 1) {
        u32 max_len = 0; u64 lo = ~0ull, hi = 0, olo = ~0ull, ohi = 0;
        for (u32 i = 0; i < n; i++) {
            if (in_len[i] > max_len) max_len = in_len[i];
            if (in_off[i] < lo) lo = in_off[i]; if (in_off[i] + in_len[i] > hi) hi = in_off[i] + in_len[i];
            if (out_off[i] < olo) olo = out_off[i]; if (out_off[i] + out_cap[i] > ohi) ohi = out_off[i] + out_cap[i];
        }
        if (max_len > KMP_MAX_SLICE_BYTES) { g_last_error = "host batch: a slice is larger than 128 KiB"; return KMP_ERR_CAPACITY; }
        if (device < 0 || device >= KMP_HOST_DEVICES) { g_last_error = "kmp_zstd_compress_host_batch: bad device"; return KMP_ERR_ARG; }
        std::lock_guard<std::mutex> one_call(bulk_tab().call[device]);
        bulk_pipe* b = bulk_pipe_get(device, n, max_len);
        if (b) {
            device_guard const on(device);
            // (device-visible addresses of the caller's ranges when those are pinned: offset back to the buffers' starts)
            u8* sd = host_range_device_ptr((const u8*)h_src + lo, (size_t)(hi - lo)); if (sd) sd -= lo;
            u8* dd = host_range_device_ptr((u8*)h_dst + olo, (size_t)(ohi - olo)); if (dd) dd -= olo;
            int rc = KMP_OK;
            for (u32 first = 0; first < n; first += b->cap_slices) {
                u32 const m = n - first < b->cap_slices ? n - first : b->cap_slices;
                int const r = bulk_pipe_run(b, level, (const u8*)h_src, sd, in_off + first, in_len + first, m, (u8*)h_dst, dd, out_off + first, out_cap + first, out_len + first, workers);
                if (rc_rank(r) > rc_rank(rc)) rc = r;
                if (rc_rank(r) == KMP_RC_RANK_OTHER) break;
            }
            return rc;
        }
        (void)hipGetLastError();                                      // no bulk compressor (memory): the staged engine, piece by piece
    }
    host_engine* e0 = host_engine_get(device);
    if (!e0) return KMP_ERR_HIP;
    u32 const piece = e0->cap_slices;
    std::vector<host_job> jobs(piece);
    int rc = KMP_OK;
    for (u32 first = 0; first < n; first += piece) {
        u32 const m = n - first < piece ? n - first : piece;
        for (u32 i = 0; i < m; i++) {
            host_job& j = jobs[i];
            j.in = (const u8*)h_src + in_off[first + i]; j.len = in_len[first + i]; j.out = (u8*)h_dst + out_off[first + i]; j.out_cap = out_cap[first + i];
            j.out_len = 0; j.status = 0; j.out_vec = nullptr; j.done = false; j.batch_rc = KMP_OK;
        }
        int const r = host_engine_compress(e0, jobs.data(), m, level);
        if (rc_rank(r) > rc_rank(rc)) rc = r;
        if (rc_rank(r) == KMP_RC_RANK_OTHER) break;                    // a HIP or kernel failure: stop; a refused slice or a small output region: go on
        for (u32 i = 0; i < m; i++) out_len[first + i] = jobs[i].out_len;
    }
    return rc;
}

extern "C" int kmp_zstd_decompress_host_batch(int device, const void* h_src, const uint64_t* in_off, const uint32_t* in_len, uint32_t n,
                                              void* h_dst, const uint64_t* out_off, const uint32_t* out_cap, uint32_t* out_len, uint32_t* status)
{
    if (n && (!h_src || !in_off || !in_len || !h_dst || !out_off || !out_cap || !out_len || !status)) { g_last_error = "kmp_zstd_decompress_host_batch: null argument"; return KMP_ERR_ARG; }
    for (u32 i = 0; i < n; i++) { out_len[i] = 0; status[i] = 1; }
    host_engine* e0 = host_engine_get(device);
    if (!e0) return KMP_ERR_HIP;
    // (large batches: the bulk engines on worker threads -- while one piece is on the device the next is being packed into pinned
    // memory and the last one's contents are being handed out)
    u32 const workers_wanted = env_u32("KMP_HOST_BULK_WORKERS", 4);
    bool bulk = n > 2u * e0->cap_slices && workers_wanted >= 1;
    u32 workers = !bulk ? 1u : (workers_wanted > (u32)KMP_HOST_ENGINES - 1u ? (u32)KMP_HOST_ENGINES - 1u : workers_wanted);
    std::vector<host_engine*> eng(workers, e0);
    if (bulk) {
        u32 got = 0;
        for (u32 w = 0; w < workers; w++) { host_engine* const x = host_engine_get(device, 1 + (int)w); if (!x) { (void)hipGetLastError(); break; } eng[got++] = x; }
        if (got == 0) { bulk = false; workers = 1; eng.assign(1, e0); } else { workers = got; eng.resize(got); }
    }
    host_engine* const e = eng[0];
    // pieces: as many entries as an engine's staging holds
    std::vector<u32> starts; starts.push_back(0);
    for (u32 first = 0; first < n; ) {
        u32 m = 0; size_t ib = 0, ob = 0;
        while (first + m < n && m < e->cap_slices) {
            size_t const a_ = ((size_t)in_len[first + m] + 63) & ~(size_t)63, b_ = ((size_t)out_cap[first + m] + 63) & ~(size_t)63;
            if (m && (ib + a_ > e->in_bytes || ob + b_ > e->out_bytes)) break;
            ib += a_; ob += b_; m++;
        }
        first += m; starts.push_back(first);
    }
    std::atomic<u32> next(0); std::atomic<int> rc(KMP_OK);
    auto work = [&](u32 w) {
        std::vector<host_job> jobs(e->cap_slices);
        for (;;) {
            u32 const pi = next.fetch_add(1);
            if (pi + 1 >= starts.size() || rc.load() != KMP_OK) return;
            u32 const first = starts[pi], m = starts[pi + 1] - first;
            for (u32 i = 0; i < m; i++) {
                host_job& j = jobs[i];
                j.in = (const u8*)h_src + in_off[first + i]; j.len = in_len[first + i]; j.out = (u8*)h_dst + out_off[first + i]; j.out_cap = out_cap[first + i];
                j.out_len = 0; j.status = 0; j.out_vec = nullptr; j.done = false; j.batch_rc = KMP_OK;
            }
            int const r = host_engine_decompress(eng[w], jobs.data(), m);
            if (r != KMP_OK) { rc_raise(rc, r); return; }
            for (u32 i = 0; i < m; i++) { out_len[first + i] = jobs[i].out_len; status[first + i] = jobs[i].status; }
        }
    };
    if (workers == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (u32 w = 0; w < workers; w++) th.emplace_back(work, w);
        for (auto& t : th) t.join();
    }
    if (rc.load() != KMP_OK) g_last_error = "kmp_zstd_decompress_host_batch: a piece failed";
    return rc.load();
}

/* Gives back what the host-batch calls and the coalescer hold on `device` (-1: on every device): the engines' pinned staging, their
 * device buffers and batch contexts, the bulk compressor.  They are made again by the next call that needs them.  The caller makes
 * sure no host-batch call and no closing kmp_zstd_*_stream call is running.  Memory registered with kmp_host_register stays
 * registered: it is the caller's. */
extern "C" int kmp_host_engines_release(int device)
{
    if (device >= KMP_HOST_DEVICES) { g_last_error = "kmp_host_engines_release: bad device"; return KMP_ERR_ARG; }
    for (int d = 0; d < KMP_HOST_DEVICES; d++) {
        if (device >= 0 && d != device) continue;
        {
            engine_table& T = engine_tab();
            std::lock_guard<std::mutex> g(T.m);
            for (int s = 0; s < KMP_HOST_ENGINES; s++) if (T.engines[d][s]) {
                { std::lock_guard<std::mutex> r(T.engines[d][s]->run_mutex); }
                host_engine_free(T.engines[d][s]); T.engines[d][s] = nullptr; T.failed[d][s] = false;
            }
        }
        {
            bulk_table& B = bulk_tab();
            std::lock_guard<std::mutex> one_call(B.call[d]);
            std::lock_guard<std::mutex> g(B.m);
            if (B.pipe[d]) { { std::lock_guard<std::mutex> r(B.pipe[d]->run_mutex); } bulk_pipe_free(B.pipe[d]); B.pipe[d] = nullptr; }
        }
    }
    return KMP_OK;
}

// ---- the coalescer of kmp_zstd_compress_stream ---------------------------------------------------------------------
// A closing call of the plain kind queues its slice and waits.  The first waiter leads: it gives others
// KMP_COALESCE_US (default 150) microseconds to join -- or until KMP_COALESCE_MAX (default 256) slices wait --, runs
// them as one batch, hands the frames out and passes the lead on if more have queued meanwhile.  The window is only spent
// when contexts have been seen closing at the same time lately (a lone caller does not pay it); 64 concurrent contexts share
// one launch of each kernel instead of queueing 64 batches of one.
struct coalescer {
    std::mutex m; std::condition_variable cv;
    std::vector<host_job*> queue; bool leading; int device; u32 window_us, max_batch; int decode;
    u32 crowd;                                              // > 0: callers arrived while a batch was forming or running, lately
};
// one per device and direction (decode = 1: the decoder's, over host_engine_decompress)
static coalescer* coalescer_get(int device, int decode = 0)
{
    static std::mutex m; static coalescer* cs[2][KMP_HOST_DEVICES] = { { nullptr }, { nullptr } };
    if (device < 0 || device >= KMP_HOST_DEVICES) return nullptr;
    std::lock_guard<std::mutex> g(m);
    if (!cs[decode][device]) {
        coalescer* c = new (std::nothrow) coalescer();
        if (!c) return nullptr;
        c->leading = false; c->device = device; c->decode = decode; c->window_us = env_u32("KMP_COALESCE_US", 150); c->max_batch = env_u32("KMP_COALESCE_MAX", 256);
        if (c->max_batch < 1) c->max_batch = 1;
        c->crowd = 0;
        cs[decode][device] = c;
    }
    return cs[decode][device];
}
// 0 = off: every context compresses alone as before
static bool coalesce_enabled() { static u32 const v = env_u32("KMP_COALESCE", 1); return v != 0; }

// One job through its direction's coalescer: queued, run in whatever batch forms, handed back.  Returns KMP_OK when the batch ran:
// job.status / job.out_len then say what became of this entry (a libzstd error number for a frame that does not decode, 70 for
// an output region that is too small).  Any other return value means the BATCH failed (a HIP error, the engine's capacity) and
// says nothing about this entry: the caller goes on alone, as it would without a coalescer.
static int coalesced_run(int device, int decode, host_job& job)
{
    coalescer* c = coalescer_get(device, decode);
    host_engine* e = host_engine_get(device);
    if (!c || !e) return KMP_ERR_ARG;
    job.out_len = 0; job.status = 0; job.done = false; job.batch_rc = KMP_OK;
    std::unique_lock<std::mutex> lk(c->m);
    if (c->leading || !c->queue.empty()) c->crowd = 64;                    // somebody else is in here right now
    c->queue.push_back(&job);
    c->cv.notify_all();                                                   // (a leader in its window counts the queue)
    for (;;) {
        if (job.done) return job.batch_rc;
        if (!c->leading) {
            c->leading = true;
            u32 const cap = c->max_batch < e->cap_slices ? c->max_batch : e->cap_slices;
            if (c->crowd) { c->crowd--; c->cv.wait_for(lk, std::chrono::microseconds(c->window_us), [&] { return c->queue.size() >= cap; }); }
            std::vector<host_job*> mine;
            u32 const take = c->queue.size() < cap ? (u32)c->queue.size() : cap;
            mine.assign(c->queue.begin(), c->queue.begin() + take);
            c->queue.erase(c->queue.begin(), c->queue.begin() + take);
            if (take > 1) c->crowd = 64;
            lk.unlock();
            std::vector<host_job> batch(take);
            for (u32 i = 0; i < take; i++) batch[i] = *mine[i];
            bool ran = false;
            int const rc = decode ? host_engine_decompress(e, batch.data(), take) : host_engine_compress(e, batch.data(), take, 3, &ran);
            // per-entry outcomes (a frame that does not decode, a small output region) are the entries' own: the batch itself ran
            bool const batch_ran = decode ? rc == KMP_OK : ran;
            lk.lock();
            for (u32 i = 0; i < take; i++) {
                mine[i]->out_len = batch[i].out_len;
                mine[i]->status = batch[i].status;
                mine[i]->batch_rc = batch_ran ? KMP_OK : rc;
                mine[i]->done = true;
            }
            c->leading = false;
            c->cv.notify_all();
            continue;                                                      // (my own job was in that batch unless the queue was longer than a batch)
        }
        c->cv.wait(lk);
    }
}
// the frame of `in` (len <= 128 KiB, level 3, no dictionary) into *out; KMP_OK = the frame is there, anything else: compress alone
static int coalesced_compress(int device, const u8* in, u32 len, std::vector<u8>* out)
{
    host_job job; job.in = in; job.len = len; job.out = nullptr; job.out_cap = 0; job.out_vec = out;
    int const rc = coalesced_run(device, 0, job);
    if (rc != KMP_OK) return rc;
    return (job.status || job.out_len == 0) ? KMP_ERR_KERNEL : KMP_OK;
}
// the content of the frame `in` (no dictionary; content size known and <= 128 KiB) into out[0 .. cap); *status = libzstd's error
// number for THIS frame when the call returns KMP_OK; any other return value: the batch failed, decode alone
static int coalesced_decompress(int device, const u8* in, u32 len, u8* out, u32 cap, u32* out_len, u32* status)
{
    host_job job; job.in = in; job.len = len; job.out = out; job.out_cap = cap; job.out_vec = nullptr;
    int const rc = coalesced_run(device, 1, job);
    *out_len = job.out_len; *status = job.status;
    return rc;
}
