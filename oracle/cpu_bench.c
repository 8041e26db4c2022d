/* cpu_bench.c -- TEST / BENCH INFRASTRUCTURE (never on the product path).
 *
 * bench.py's cpu_baseline leg: times the reference's arithmetic on the GPU box's host cores from plain pthreads, so that
 * the figure does not carry a Python thread pool (all cores stated, a single-thread figure, a warm-up
 * and several passes).  What is timed is either the binary libzstd 1.5.7 the reference binds (gradle/libs.versions.toml:9,46;
 * the call behind kompressor-zstd--nativelib/src/jvmCommonMain/jni/Wrapper.cpp:112), looked up on the machine and passed in
 * by path -- driven with ZSTD_compress2 up to 128 KiB (what the reference's one-shot driver amounts to there) and with
 * ZSTD_compressStream2(e_end) into output slices of max(8192, n / 10) bytes above (SliceTransform.kt:33-56) --, or, when the
 * machine has none, the oracle's C restatement (oracle/zstd_l3_ref.c: kref_zstd_l3_compress).
 *
 *   gcc -O2 -shared -fPIC -o oracle/_build/libcpubench.so oracle/cpu_bench.c -ldl -lpthread
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct { void* p; size_t size, pos; } zbuf;
typedef void* (*create_t)(void);
typedef size_t (*setparam_t)(void*, int, int);
typedef size_t (*compress2_t)(void*, void*, size_t, const void*, size_t);
typedef size_t (*stream2_t)(void*, zbuf*, zbuf*, int);
typedef size_t (*free_t)(void*);
typedef size_t (*kref_t)(void*, size_t, const char*, size_t);

typedef struct {
    int id, threads, passes; uint32_t n, slice; const unsigned char* base;
    create_t create; setparam_t setp; compress2_t c2; stream2_t s2; free_t freec; kref_t kref;
    pthread_barrier_t* bar; double* secs; uint64_t* bytes; uint64_t* errors;
} job;

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }

/* the level the binary library runs at (default 3: BASELINE configs[1]; the bench's lines for other levels set theirs) */
static int g_zstd_level = 3;
__attribute__((visibility("default")))
void cpubench_set_zstd_level(int level) { g_zstd_level = level; }

static void* worker(void* arg)
{
    job* j = (job*)arg;
    size_t const cap = j->slice + j->slice / 128 + 1024;
    size_t const chunk = j->slice / 10 > 8192 ? j->slice / 10 : 8192;
    unsigned char* out = (unsigned char*)malloc(cap);
    void* cctx = j->create ? j->create() : NULL;
    if (cctx) j->setp(cctx, 100, g_zstd_level);                      /* ZSTD_c_compressionLevel = 100 (ZstdCompressor.jvm.kt:41) */
    uint32_t const per = (j->n + j->threads - 1) / j->threads;
    uint32_t const lo = (uint32_t)j->id * per, hi = lo + per < j->n ? lo + per : j->n;
    for (int p = 0; p < j->passes; p++) {
        pthread_barrier_wait(j->bar);
        double const t0 = now();
        uint64_t total = 0, bad = 0;
        for (uint32_t i = lo; i < hi; i++) {
            const unsigned char* src = j->base + (size_t)i * j->slice;
            size_t r;
            if (j->kref) r = j->kref(out, cap, (const char*)src, j->slice);
            else if (j->slice <= 131072) r = j->c2(cctx, out, cap, src, j->slice);
            else {
                zbuf ib = { (void*)src, j->slice, 0 };
                r = 0;
                for (;;) {
                    zbuf ob = { out, chunk, 0 };
                    size_t const q = j->s2(cctx, &ob, &ib, 2);          /* ZSTD_e_end from the first call, as the reference's driver */
                    r += ob.pos;
                    if (q == 0) break;
                    if (q > ((size_t)1 << 40)) { r = q; break; }
                }
            }
            if (r == 0 || r > ((size_t)1 << 40)) bad++; else total += r;
        }
        pthread_barrier_wait(j->bar);
        if (j->id == 0) j->secs[p] = now() - t0;
        j->bytes[(size_t)p * j->threads + j->id] = total;
        j->errors[j->id] += bad;
    }
    if (cctx && j->freec) j->freec(cctx);
    free(out);
    return NULL;
}

/* Compresses slices base[i * slice .. ), i < n, on `threads` host threads, `passes` times (the caller treats the first as a
 * warm-up).  libpath: a libzstd 1.5.7 (kind "reference"), or NULL with krefpath = the oracle's library (kind "port").
 * secs[p] = wall time of pass p, frame_bytes = the frames' total size (one pass).  0 on success. */
__attribute__((visibility("default")))
int cpubench_zstd_l3(const char* libpath, const char* krefpath, const unsigned char* base, uint32_t n, uint32_t slice,
                     int threads, int passes, double* secs, uint64_t* frame_bytes, uint64_t* errors)
{
    if (threads < 1 || passes < 1 || n == 0) return -1;
    job proto; memset(&proto, 0, sizeof proto);
    if (libpath) {
        void* h = dlopen(libpath, RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
        if (!h) return -2;
        proto.create = (create_t)dlsym(h, "ZSTD_createCCtx"); proto.setp = (setparam_t)dlsym(h, "ZSTD_CCtx_setParameter");
        proto.c2 = (compress2_t)dlsym(h, "ZSTD_compress2"); proto.s2 = (stream2_t)dlsym(h, "ZSTD_compressStream2");
        proto.freec = (free_t)dlsym(h, "ZSTD_freeCCtx");
        unsigned (*ver)(void) = (unsigned (*)(void))dlsym(h, "ZSTD_versionNumber");
        if (!proto.create || !proto.setp || !proto.c2 || !proto.s2 || !ver || ver() != 10507) return -3;
    } else {
        void* h = dlopen(krefpath, RTLD_NOW | RTLD_LOCAL);
        if (!h) return -2;
        proto.kref = (kref_t)dlsym(h, "kref_zstd_l3_compress");
        if (!proto.kref) return -3;
    }
    pthread_barrier_t bar; pthread_barrier_init(&bar, NULL, (unsigned)threads);
    uint64_t* bytes = (uint64_t*)calloc((size_t)passes * threads, sizeof(uint64_t));
    uint64_t* errs = (uint64_t*)calloc((size_t)threads, sizeof(uint64_t));
    job* jobs = (job*)calloc((size_t)threads, sizeof(job));
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    for (int t = 0; t < threads; t++) {
        jobs[t] = proto; jobs[t].id = t; jobs[t].threads = threads; jobs[t].passes = passes; jobs[t].n = n; jobs[t].slice = slice; jobs[t].base = base;
        jobs[t].bar = &bar; jobs[t].secs = secs; jobs[t].bytes = bytes; jobs[t].errors = errs;
        pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    uint64_t tot = 0, bad = 0;
    for (int t = 0; t < threads; t++) { tot += bytes[(size_t)(passes - 1) * threads + t]; bad += errs[t]; }
    *frame_bytes = tot; *errors = bad;
    pthread_barrier_destroy(&bar);
    free(bytes); free(errs); free(jobs); free(th);
    return 0;
}


/* ---- the other BASELINE configs: configs[2] (ZstdDecompressor), configs[4] (raw DEFLATE level 6) and its inverse -------------
 * Same harness (pthreads, static partition of entries over threads, barriers around a pass).  Entries are (offset, length)
 * pairs into one buffer; `cap` is the room per output.
 *   kind 1  libzstd 1.5.7 ZSTD_decompressStream (the call at kompressor-zstd--nativelib/src/jvmCommonMain/jni/Wrapper.cpp:178) driven as
 *           the reference's one-shot driver drives it: the whole frame as input, output slices of max(8192, frame / 10) bytes
 *           (SliceTransform.kt:33-45); stream_api = 0: ZSTD_decompressDCtx into one buffer instead (the library's fastest entry)
 *   kind 2  zlib deflateInit2(level, 8, -15, 8, 0) + deflate(Z_FINISH) per slice (kompressor-zlib--nativelib/.../jni/Wrapper.cpp:20,73)
 *   kind 3  zlib inflateInit2(-15) + inflate(Z_FINISH) per stream (Wrapper.cpp:91,144)
 * zlib is whatever libz.so.1 the machine has (its version is reported through *zlib_vernum). */
typedef struct {
    const unsigned char* next_in; unsigned avail_in; unsigned long total_in;
    unsigned char* next_out; unsigned avail_out; unsigned long total_out;
    const char* msg; void* state; void* zalloc; void* zfree; void* opaque; int data_type; unsigned long adler; unsigned long reserved;
} kz_stream;
typedef int (*dinit_t)(kz_stream*, int, int, int, int, int, const char*, int);
typedef int (*iinit_t)(kz_stream*, int, const char*, int);
typedef int (*zrun_t)(kz_stream*, int);
typedef int (*zend_t)(kz_stream*);
typedef size_t (*dstream_t)(void*, zbuf*, zbuf*);
typedef size_t (*dctx_t)(void*, void*, size_t, const void*, size_t);

typedef struct {
    int id, threads, passes, kind, level, stream_api; uint32_t n, cap; const unsigned char* base; const uint64_t* off; const uint32_t* len;
    create_t create; free_t freec; dstream_t dstream; dctx_t dctx;
    dinit_t dinit; iinit_t iinit; zrun_t zdeflate, zinflate; zend_t dend, iend; const char* zver;
    pthread_barrier_t* bar; double* secs; uint64_t* bytes; uint64_t* errors;
} job2;

static void* worker2(void* arg)
{
    job2* j = (job2*)arg;
    size_t const cap = (size_t)j->cap + 1024;
    unsigned char* out = (unsigned char*)malloc(cap);
    void* dctx = (j->kind == 1 && j->create) ? j->create() : NULL;
    uint32_t const per = (j->n + j->threads - 1) / j->threads;
    uint32_t const lo = (uint32_t)j->id * per, hi = lo + per < j->n ? lo + per : j->n;
    for (int p = 0; p < j->passes; p++) {
        pthread_barrier_wait(j->bar);
        double const t0 = now();
        uint64_t total = 0, bad = 0;
        for (uint32_t i = lo; i < hi && i < j->n; i++) {
            const unsigned char* src = j->base + j->off[i]; size_t const sl = j->len[i];
            size_t r = 0;
            if (j->kind == 1) {
                if (!j->stream_api) { r = j->dctx(dctx, out, cap, src, sl); if (r > ((size_t)1 << 40)) r = 0; }
                else {
                    size_t const chunk = sl / 10 > 8192 ? sl / 10 : 8192;
                    zbuf ib = { (void*)src, sl, 0 };
                    for (;;) {
                        if (r + chunk > cap) { r = 0; break; }
                        zbuf ob = { out + r, chunk, 0 };
                        size_t const q = j->dstream(dctx, &ob, &ib);
                        r += ob.pos;
                        if (q > ((size_t)1 << 40)) { r = 0; break; }
                        if (q == 0 && ib.pos == ib.size) break;
                        if (ob.pos == 0 && ib.pos == ib.size) { r = 0; break; }      /* truncated */
                    }
                }
            } else {
                kz_stream z; memset(&z, 0, sizeof z);
                int rc = j->kind == 2 ? j->dinit(&z, j->level, 8, -15, 8, 0, j->zver, (int)sizeof z) : j->iinit(&z, -15, j->zver, (int)sizeof z);
                if (rc == 0) {
                    z.next_in = src; z.avail_in = (unsigned)sl; z.next_out = out; z.avail_out = (unsigned)cap;
                    rc = j->kind == 2 ? j->zdeflate(&z, 4) : j->zinflate(&z, 4);           /* Z_FINISH */
                    if (rc == 1) r = z.total_out;                                           /* Z_STREAM_END */
                    (void)(j->kind == 2 ? j->dend(&z) : j->iend(&z));
                }
            }
            if (r == 0 && sl != 0) bad++; else total += r;
        }
        pthread_barrier_wait(j->bar);
        if (j->id == 0) j->secs[p] = now() - t0;
        j->bytes[(size_t)p * j->threads + j->id] = total;
        j->errors[j->id] += bad;
    }
    if (dctx && j->freec) j->freec(dctx);
    free(out);
    return NULL;
}

/* kind 1 .. 3 as above over entries base[off[i] .. + len[i]), i < n; cap = room per output; level: zlib level of kind 2;
 * libpath: the libzstd 1.5.7 of kind 1 (NULL otherwise).  secs[p] = wall time of pass p; out_bytes = what one pass produced.
 * 0 on success, -2 library not found, -3 symbol / version mismatch. */
__attribute__((visibility("default")))
int cpubench_codec(int kind, const char* libpath, int level, int stream_api, const unsigned char* base, const uint64_t* off, const uint32_t* len,
                   uint32_t n, uint32_t cap, int threads, int passes, double* secs, uint64_t* out_bytes, uint64_t* errors, char* zlib_version, int zlib_version_cap)
{
    if (threads < 1 || passes < 1 || n == 0 || kind < 1 || kind > 3) return -1;
    job2 proto; memset(&proto, 0, sizeof proto);
    proto.kind = kind; proto.level = level; proto.stream_api = stream_api;
    if (kind == 1) {
        void* h = dlopen(libpath, RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
        if (!h) return -2;
        proto.create = (create_t)dlsym(h, "ZSTD_createDCtx"); proto.freec = (free_t)dlsym(h, "ZSTD_freeDCtx");
        proto.dstream = (dstream_t)dlsym(h, "ZSTD_decompressStream"); proto.dctx = (dctx_t)dlsym(h, "ZSTD_decompressDCtx");
        unsigned (*ver)(void) = (unsigned (*)(void))dlsym(h, "ZSTD_versionNumber");
        if (!proto.create || !proto.freec || !proto.dstream || !proto.dctx || !ver || ver() != 10507) return -3;
    } else {
        void* h = dlopen("libz.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) return -2;
        proto.dinit = (dinit_t)dlsym(h, "deflateInit2_"); proto.iinit = (iinit_t)dlsym(h, "inflateInit2_");
        proto.zdeflate = (zrun_t)dlsym(h, "deflate"); proto.zinflate = (zrun_t)dlsym(h, "inflate");
        proto.dend = (zend_t)dlsym(h, "deflateEnd"); proto.iend = (zend_t)dlsym(h, "inflateEnd");
        const char* (*zv)(void) = (const char* (*)(void))dlsym(h, "zlibVersion");
        if (!proto.dinit || !proto.iinit || !proto.zdeflate || !proto.zinflate || !proto.dend || !proto.iend || !zv) return -3;
        proto.zver = zv();
        if (zlib_version && zlib_version_cap > 0) { strncpy(zlib_version, proto.zver, (size_t)zlib_version_cap - 1); zlib_version[zlib_version_cap - 1] = 0; }
    }
    pthread_barrier_t bar; pthread_barrier_init(&bar, NULL, (unsigned)threads);
    uint64_t* bytes = (uint64_t*)calloc((size_t)passes * threads, sizeof(uint64_t));
    uint64_t* errs = (uint64_t*)calloc((size_t)threads, sizeof(uint64_t));
    job2* jobs = (job2*)calloc((size_t)threads, sizeof(job2));
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    for (int t = 0; t < threads; t++) {
        jobs[t] = proto; jobs[t].id = t; jobs[t].threads = threads; jobs[t].passes = passes; jobs[t].n = n; jobs[t].cap = cap;
        jobs[t].base = base; jobs[t].off = off; jobs[t].len = len;
        jobs[t].bar = &bar; jobs[t].secs = secs; jobs[t].bytes = bytes; jobs[t].errors = errs;
        pthread_create(&th[t], NULL, worker2, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    uint64_t tot = 0, bad = 0;
    for (int t = 0; t < threads; t++) { tot += bytes[(size_t)(passes - 1) * threads + t]; bad += errs[t]; }
    *out_bytes = tot; *errors = bad;
    pthread_barrier_destroy(&bar);
    free(bytes); free(errs); free(jobs); free(th);
    return 0;
}
