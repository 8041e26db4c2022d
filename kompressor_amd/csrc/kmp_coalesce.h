// kmp_coalesce.h -- the batch for callers that hold HOST memory (a JVM does): a host-batch engine and, on top of it, the
// coalescer of the streaming entry point.  Included by kmp_api.hip behind the batched device API it drives.
//
// Why: what this library measures is kmp_zstd_compress_batch over device pointers, and what the reference's Kotlin side can
// bind is the per-slice kmp_zstd_compress_stream (ZstdWrapper.kt:35-46), which ran every slice as a batch of ONE with
// synchronous copies.  Reference callers run many contexts at once (AsyncSliceTransform.kt:56-65: one transform per
// coroutine; BaseSliceTransformContentEncoder.kt:34-43: one per response).  Two things close the gap:
//   * kmp_zstd_compress_host_batch / kmp_zstd_decompress_host_batch: slices in host memory in, frames in host memory out,
//     through pinned staging and the device batch (jni/zstd/BatchWrapper.cpp binds them for direct ByteBuffers);
//   * inside kmp_zstd_compress_stream, closing calls of concurrent contexts that ask for the plain case (level 3, no
//     dictionary, the whole slice in one piece of at most 128 KiB) are gathered for a short window and compressed as ONE
//     batch; each caller gets its own frame back, bit-identical to what it would have got alone (frames do not depend
//     on batch position: tests/test_gpu_parity.py).
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

struct host_job { const u8* in; u32 len; u8* out; u32 out_cap; u32 out_len; u32 status; std::vector<u8>* out_vec; bool done; };

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
// 128 KiB -- the coalescer's and the small calls'.  Slots 1 ..: the bulk engines of large host batches (KMP_HOST_BULK_SLICES,
// default 16 384 slices each), one per worker thread of such a call.
enum { KMP_HOST_ENGINES = 5 };
static host_engine* host_engine_get(int device, int slot = 0)
{
    static std::mutex m; static host_engine* engines[16][KMP_HOST_ENGINES] = { { nullptr } };
    if (device < 0 || device >= 16 || slot < 0 || slot >= KMP_HOST_ENGINES) return nullptr;
    std::lock_guard<std::mutex> g(m);
    if (engines[device][slot]) return engines[device][slot];
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
    if (!ok) { (void)hipGetLastError(); host_engine_free(e); g_last_error = "host batch engine: allocation failed"; return nullptr; }
    engines[device][slot] = e;
    return e;
}

// Compresses jobs[0 .. n) (n <= cap_slices, every len <= 128 KiB) at level 1 .. 3: one H2D copy, the device batch, the dense
// packing, two D2H copies.  A frame goes to job.out (out_cap bytes; status 70 = too small) or to job.out_vec.
static int host_engine_compress(host_engine* e, host_job* jobs, u32 n, int level)
{
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
    int rc = KMP_OK;
    for (u32 i = 0; i < n; i++) {
        size_t const a = (size_t)e->h_off[i], b = (size_t)e->h_off[i + 1];
        u32 const fl = (u32)(b - a);
        jobs[i].status = 0; jobs[i].out_len = 0;
        if (fl == 0) { jobs[i].status = 1; rc = KMP_ERR_KERNEL; continue; }                     // a real frame is never empty
        if (jobs[i].out_vec) { jobs[i].out_vec->assign(e->h_out + a, e->h_out + b); jobs[i].out_len = fl; }
        else if (fl > jobs[i].out_cap) { jobs[i].status = 70; if (rc == KMP_OK) rc = KMP_ERR_CAPACITY; }
        else { memcpy(jobs[i].out, e->h_out + a, fl); jobs[i].out_len = fl; }
    }
    if (rc == KMP_ERR_CAPACITY) g_last_error = "host batch: an output region is smaller than its frame (kmp_zstd_compress_bound)";
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
    // (the strided offsets of the compress side live in d_ooff, untouched; h_doff was borrowed: put it back)
    for (u32 i = 0; i < n; i++) {
        jobs[i].status = e->h_st[i]; jobs[i].out_len = e->h_st[i] ? 0u : e->h_len[i];
        if (jobs[i].out_len) memcpy(jobs[i].out, e->h_out + e->h_doff[i], jobs[i].out_len);
    }
    for (size_t i = 0; i <= e->cap_slices; i++) e->h_doff[i] = i * e->stride;
    return KMP_OK;
}

extern "C" int kmp_zstd_compress_host_batch(int device, int level, const void* h_src, const uint64_t* in_off, const uint32_t* in_len, uint32_t n,
                                            void* h_dst, const uint64_t* out_off, const uint32_t* out_cap, uint32_t* out_len)
{
    if (n && (!h_src || !in_off || !in_len || !h_dst || !out_off || !out_cap || !out_len)) { g_last_error = "kmp_zstd_compress_host_batch: null argument"; return KMP_ERR_ARG; }
    if (level == 0) level = 3;
    if (level < -131072 || level > 4) { g_last_error = "kmp_zstd_compress_host_batch: levels -131072 .. 3 are served, and 4 for slices above 16 KiB"; return KMP_ERR_ARG; }
    host_engine* e0 = host_engine_get(device);
    if (!e0) return KMP_ERR_ARG;
    // A large batch goes through the bulk engines: pieces of KMP_HOST_BULK_SLICES slices handed to KMP_HOST_BULK_WORKERS
    // (default 4) threads, each with an engine of its own -- while one piece is on the device the next is being
    // packed into pinned memory and the last one's frames are being handed out, and the device has two pieces in flight.
    // (measured, 65 536 x 64 KiB from pageable host memory to frames in host memory: 1.5 GB/s through the 1 024-slice engine piece
    // by piece, 5.7 with one bulk worker, 9.5 with two, 11.6 with three: tools/r03_hostbatch.py; pieces of 8 192 on four workers: 13.0
    // against 10.8 for 16 384 on three, 12.6 for 8 192 on six, 9.4 for 4 096 on eight: tools/r03_bulk.sh -- the default.  An engine that
    // cannot be made -- its pinned staging is 2.2 GiB -- is done without.)
    u32 const workers_wanted = env_u32("KMP_HOST_BULK_WORKERS", 4);
    bool bulk = n > 2u * e0->cap_slices && workers_wanted >= 1;
    u32 workers = !bulk ? 1u : (workers_wanted > (u32)KMP_HOST_ENGINES - 1u ? (u32)KMP_HOST_ENGINES - 1u : workers_wanted);
    std::vector<host_engine*> eng(workers, e0);
    if (bulk) {
        u32 got = 0;
        for (u32 w = 0; w < workers; w++) { host_engine* const x = host_engine_get(device, 1 + (int)w); if (!x) { (void)hipGetLastError(); break; } eng[got++] = x; }
        if (got == 0) { bulk = false; workers = 1; eng.assign(1, e0); } else { workers = got; eng.resize(got); }
    }
    u32 const piece = eng[0]->cap_slices;
    std::atomic<u32> next(0); std::atomic<int> rc(KMP_OK);
    auto work = [&](u32 w) {
        std::vector<host_job> jobs(piece);
        for (;;) {
            u32 const first = next.fetch_add(piece);
            if (first >= n || (rc.load() != KMP_OK && rc.load() != KMP_ERR_CAPACITY)) return;
            u32 const m = n - first < piece ? n - first : piece;
            for (u32 i = 0; i < m; i++) {
                host_job& j = jobs[i];
                j.in = (const u8*)h_src + in_off[first + i]; j.len = in_len[first + i]; j.out = (u8*)h_dst + out_off[first + i]; j.out_cap = out_cap[first + i];
                j.out_len = 0; j.status = 0; j.out_vec = nullptr; j.done = false;
            }
            int const r = host_engine_compress(eng[w], jobs.data(), m, level);
            if (r != KMP_OK) rc.store(r);
            if (r == KMP_OK || r == KMP_ERR_CAPACITY) for (u32 i = 0; i < m; i++) out_len[first + i] = jobs[i].out_len;
        }
    };
    if (workers == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (u32 w = 0; w < workers; w++) th.emplace_back(work, w);
        for (auto& t : th) t.join();
    }
    if (rc.load() != KMP_OK && rc.load() != KMP_ERR_CAPACITY) g_last_error = "kmp_zstd_compress_host_batch: a piece failed on a worker thread";
    else if (rc.load() == KMP_ERR_CAPACITY) g_last_error = "host batch: an output region is smaller than its frame (kmp_zstd_compress_bound)";
    return rc.load();
}

extern "C" int kmp_zstd_decompress_host_batch(int device, const void* h_src, const uint64_t* in_off, const uint32_t* in_len, uint32_t n,
                                              void* h_dst, const uint64_t* out_off, const uint32_t* out_cap, uint32_t* out_len, uint32_t* status)
{
    if (n && (!h_src || !in_off || !in_len || !h_dst || !out_off || !out_cap || !out_len || !status)) { g_last_error = "kmp_zstd_decompress_host_batch: null argument"; return KMP_ERR_ARG; }
    host_engine* e0 = host_engine_get(device);
    if (!e0) return KMP_ERR_ARG;
    // (large batches: the bulk engines on worker threads, as on the compress side)
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
                j.out_len = 0; j.status = 0; j.out_vec = nullptr; j.done = false;
            }
            int const r = host_engine_decompress(eng[w], jobs.data(), m);
            if (r != KMP_OK) { rc.store(r); return; }
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

// ---- the coalescer of kmp_zstd_compress_stream ---------------------------------------------------------------------
// A closing call of the plain kind queues its slice and waits.  The first waiter leads: it gives others
// KMP_COALESCE_US (default 150) microseconds to join -- or until KMP_COALESCE_MAX (default 256) slices wait --, runs
// them as one batch, hands the frames out and passes the lead on if more have queued meanwhile.  A lone caller pays the
// window once; 64 concurrent contexts share one launch of each kernel instead of queueing 64 batches of one.
struct coalescer {
    std::mutex m; std::condition_variable cv;
    std::vector<host_job*> queue; bool leading; int device; u32 window_us, max_batch; int decode;
};
// one per device and direction (decode = 1: the decoder's, over host_engine_decompress)
static coalescer* coalescer_get(int device, int decode = 0)
{
    static std::mutex m; static coalescer* cs[2][16] = { { nullptr }, { nullptr } };
    if (device < 0 || device >= 16) return nullptr;
    std::lock_guard<std::mutex> g(m);
    if (!cs[decode][device]) {
        coalescer* c = new (std::nothrow) coalescer();
        if (!c) return nullptr;
        c->leading = false; c->device = device; c->decode = decode; c->window_us = env_u32("KMP_COALESCE_US", 150); c->max_batch = env_u32("KMP_COALESCE_MAX", 256);
        if (c->max_batch < 1) c->max_batch = 1;
        cs[decode][device] = c;
    }
    return cs[decode][device];
}
// 0 = off: every context compresses alone as before
static bool coalesce_enabled() { static u32 const v = env_u32("KMP_COALESCE", 1); return v != 0; }

// One job through its direction's coalescer: queued, run in whatever batch forms, handed back.  Returns a KMP_* code for the
// batch; job.status / job.out_len say what became of this entry.
static int coalesced_run(int device, int decode, host_job& job)
{
    coalescer* c = coalescer_get(device, decode);
    host_engine* e = host_engine_get(device);
    if (!c || !e) return KMP_ERR_ARG;
    job.out_len = 0; job.status = 0; job.done = false;
    std::unique_lock<std::mutex> lk(c->m);
    c->queue.push_back(&job);
    c->cv.notify_all();                                                   // (a leader in its window counts the queue)
    for (;;) {
        if (job.done) return (job.status && !decode) ? KMP_ERR_KERNEL : KMP_OK;
        if (!c->leading) {
            c->leading = true;
            u32 const cap = c->max_batch < e->cap_slices ? c->max_batch : e->cap_slices;
            c->cv.wait_for(lk, std::chrono::microseconds(c->window_us), [&] { return c->queue.size() >= cap; });
            std::vector<host_job*> mine;
            u32 const take = c->queue.size() < cap ? (u32)c->queue.size() : cap;
            mine.assign(c->queue.begin(), c->queue.begin() + take);
            c->queue.erase(c->queue.begin(), c->queue.begin() + take);
            lk.unlock();
            std::vector<host_job> batch(take);
            for (u32 i = 0; i < take; i++) batch[i] = *mine[i];
            int const rc = decode ? host_engine_decompress(e, batch.data(), take) : host_engine_compress(e, batch.data(), take, 3);
            lk.lock();
            for (u32 i = 0; i < take; i++) {
                mine[i]->out_len = batch[i].out_len;
                mine[i]->status = (rc != KMP_OK && batch[i].out_len == 0 && batch[i].status == 0) ? 1u : batch[i].status;
                mine[i]->done = true;
            }
            c->leading = false;
            c->cv.notify_all();
            continue;                                                      // (my own job was in that batch unless the queue was longer than a batch)
        }
        c->cv.wait(lk);
    }
}
// the frame of `in` (len <= 128 KiB, level 3, no dictionary) into *out; returns a KMP_* code
static int coalesced_compress(int device, const u8* in, u32 len, std::vector<u8>* out)
{
    host_job job; job.in = in; job.len = len; job.out = nullptr; job.out_cap = 0; job.out_vec = out;
    return coalesced_run(device, 0, job);
}
// the content of the frame `in` (no dictionary; content size known and <= 128 KiB) into out[0 .. cap); *status = libzstd's error number
static int coalesced_decompress(int device, const u8* in, u32 len, u8* out, u32 cap, u32* out_len, u32* status)
{
    host_job job; job.in = in; job.len = len; job.out = out; job.out_cap = cap; job.out_vec = nullptr;
    int const rc = coalesced_run(device, 1, job);
    *out_len = job.out_len; *status = job.status;
    return rc;
}
