// randstore.hip -- what does a random 4-byte hash-table insert cost on one MI355X, by store flavour and memory type?
// (k_zstd_match is bound by the rate of its table inserts: profiles/r01_random_access.txt priced a plain or non-temporal
// random store at 2.5 random loads.  This asks whether a write-through / system-scope store, or an allocation the L2
// does not cache, is any cheaper.)
//   hipcc --offload-arch=gfx950 -O3 -o randstore tools/randstore.hip && ./randstore
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t ld_plain(const uint32_t* p) { uint32_t v; asm volatile("global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ uint32_t ld_sc1(const uint32_t* p) { uint32_t v; asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ uint32_t ld_sc01(const uint32_t* p) { uint32_t v; asm volatile("global_load_dword %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ uint32_t ld_nt(const uint32_t* p) { uint32_t v; asm volatile("global_load_dword %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ void st_plain(uint32_t* p, uint32_t v) { asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_nt(uint32_t* p, uint32_t v) { asm volatile("global_store_dword %0, %1, off nt" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_sc1(uint32_t* p, uint32_t v) { asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_sc01(uint32_t* p, uint32_t v) { asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_sc01nt(uint32_t* p, uint32_t v) { asm volatile("global_store_dword %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_sc0(uint32_t* p, uint32_t v) { asm volatile("global_store_dword %0, %1, off sc0" :: "v"(p), "v"(v) : "memory"); }

// MODE: 0 nt load | 1 plain store | 2 nt store | 3 sc1 store | 4 sc0 sc1 store | 5 sc0 sc1 nt store | 6 sc0 store
//       7 plain load + plain store | 8 plain load + sc1 store | 9 plain load + sc0 sc1 store | 10 sc1 load + sc1 store
//       11 sc0 sc1 load | 12 nt load + sc0 sc1 store | 13 a lane group of 32 writes one whole 128-byte line (plain)
//       14 the same, sc0 sc1 | 15 a lane group of 16 writes one whole 64-byte line, sc0 sc1
template <int INDEP, int MODE>
__global__ void __launch_bounds__(64) k_rs(uint32_t* buf, uint64_t mask, int iters, uint32_t* sink)
{
    uint64_t s[INDEP]; uint32_t acc = 0;
    uint64_t gid = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    int const grp = MODE == 15 ? 16 : 32;
    if (MODE >= 13) gid /= grp;              // the lanes of a group share the random line
#pragma unroll
    for (int j = 0; j < INDEP; j++) s[j] = (gid * INDEP + j) * 0x9E3779B97F4A7C15ull + 12345;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < INDEP; j++) {
            s[j] = s[j] * 6364136223846793005ull + 1442695040888963407ull;
            uint64_t idx = ((s[j] >> 20) & mask);
            if (MODE >= 13) idx = (idx & ~(uint64_t)(grp - 1)) + (threadIdx.x & (grp - 1));
            uint32_t* const p = buf + idx;
            uint32_t v = 0;
            if (MODE == 0) v = ld_nt(p);
            if (MODE == 11) v = ld_sc01(p);
            if (MODE == 7 || MODE == 8 || MODE == 9) v = ld_plain(p);
            if (MODE == 10) v = ld_sc1(p);
            if (MODE == 12) v = ld_nt(p);
            if (MODE == 1 || MODE == 7 || MODE == 13) st_plain(p, v + (uint32_t)i);
            if (MODE == 2) st_nt(p, v + (uint32_t)i);
            if (MODE == 3 || MODE == 8 || MODE == 10) st_sc1(p, v + (uint32_t)i);
            if (MODE == 4 || MODE == 9 || MODE == 12 || MODE == 14 || MODE == 15) st_sc01(p, v + (uint32_t)i);
            if (MODE == 5) st_sc01nt(p, v + (uint32_t)i);
            if (MODE == 6) st_sc0(p, v + (uint32_t)i);
            acc += v; s[j] += v & 1;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

static const char* const names[] = { "nt load", "plain store", "nt store", "sc1 store", "sc0 sc1 store", "sc0 sc1 nt store", "sc0 store",
                                     "ld + plain st", "ld + sc1 st", "ld + sc0 sc1 st", "sc1 ld + sc1 st", "sc0 sc1 load", "nt ld + sc0 sc1 st",
                                     "128-B line plain st (x32 lanes)", "128-B line sc0 sc1 st (x32)", "64-B line sc0 sc1 st (x16)" };

template <int MODE>
static void run(const char* mem, uint32_t* buf, uint64_t bytes, uint32_t* sink)
{
    constexpr int INDEP = 4;
    int const iters = 400;
    uint64_t const mask = bytes / 4 - 1;
    int const blocks = 256 * 16;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_rs<INDEP, MODE>), dim3(blocks), dim3(64), 0, 0, buf, mask, iters / 4, sink);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_rs<INDEP, MODE>), dim3(blocks), dim3(64), 0, 0, buf, mask, iters, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double n = (double)blocks * 64 * iters * INDEP;
    if (MODE >= 13) n /= (MODE == 15 ? 16 : 32);
    printf("%-12s footprint %6.2f GiB  %-34s : %7.2f G %s/s\n", mem, bytes / 1073741824.0, names[MODE], n / ms / 1e6, MODE >= 13 ? "lines" : "ops");
    fflush(stdout);
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
}

static void sweep(const char* mem, uint32_t* buf, uint64_t f, uint32_t* sink)
{
    run<0>(mem, buf, f, sink); run<11>(mem, buf, f, sink);
    run<1>(mem, buf, f, sink); run<2>(mem, buf, f, sink); run<3>(mem, buf, f, sink); run<4>(mem, buf, f, sink); run<5>(mem, buf, f, sink); run<6>(mem, buf, f, sink);
    run<7>(mem, buf, f, sink); run<8>(mem, buf, f, sink); run<9>(mem, buf, f, sink); run<10>(mem, buf, f, sink); run<12>(mem, buf, f, sink);
    run<13>(mem, buf, f, sink); run<14>(mem, buf, f, sink); run<15>(mem, buf, f, sink);
}

int main()
{
    uint64_t const bytes = 8ull << 30;
    uint32_t *buf, *sink;
    CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
    sweep("hipMalloc", buf, bytes, sink);
    sweep("hipMalloc", buf, 128ull << 20, sink);
    sweep("hipMalloc", buf, 16ull << 20, sink);
    CK(hipFree(buf));
    if (hipExtMallocWithFlags((void**)&buf, bytes, hipDeviceMallocUncached) == hipSuccess) {
        CK(hipMemset(buf, 0, bytes));
        sweep("uncached", buf, bytes, sink);
        CK(hipFree(buf));
    } else printf("hipDeviceMallocUncached: not available\n");
    if (hipExtMallocWithFlags((void**)&buf, bytes, hipDeviceMallocFinegrained) == hipSuccess) {
        CK(hipMemset(buf, 0, bytes));
        sweep("finegrained", buf, bytes, sink);
        CK(hipFree(buf));
    } else printf("hipDeviceMallocFinegrained: not available\n");
    return 0;
}
