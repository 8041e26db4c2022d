// spanprobe.hip -- does the random-access rate of one MI355X depend on how much of the HBM the accesses span?
// (round 2 saw 27.5 G accesses/s inside any 8 GiB window and 37.5 over 144 GiB; this maps it out to the whole device, for
// reads, for read + insert into the same line -- the level-3 parser's table traffic -- and for stores alone, and for four
// 6 GiB pieces placed 1/4 of the span apart: what a parser's tables would be if they were spread over the whole device.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/spanprobe tools/spanprobe.hip && tools/spanprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// MODE 0: 4-byte load; 1: load + store into the same word; 2: 4-byte store; 3: load at A, store at B (k_region_probe's pair)
// MODE 4: load a word of each 32-byte sector of the line, store both back (both sectors dirty, 8 bytes of 64)
// MODE 5: four lanes share a line: each loads its 16 bytes and stores them back (the whole line dirty); counted per LINE
// MODE 6: four lanes share a line: each stores 16 bytes (whole-line store without a read); counted per LINE
// MODE 7: two lanes share a 32-byte sector: each loads its 16 bytes and stores them back (a whole sector dirty); per SECTOR
template <int INDEP, int MODE>
__global__ void __launch_bounds__(64) k_span(uint32_t* buf, uint64_t words, uint64_t piece_words, uint64_t piece_stride, int iters, uint32_t* sink)
{
    uint64_t s[INDEP]; uint32_t acc = 0;
    uint64_t gid = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    int const sub = (MODE == 5 || MODE == 6) ? (threadIdx.x & 3) : (MODE == 7 ? (threadIdx.x & 1) : 0);
    if (MODE == 5 || MODE == 6) gid >>= 2;
    if (MODE == 7) gid >>= 1;
#pragma unroll
    for (int j = 0; j < INDEP; j++) s[j] = (gid * INDEP + j) * 0x9E3779B97F4A7C15ull + 12345;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < INDEP; j++) {
            s[j] = s[j] * 6364136223846793005ull + 1442695040888963407ull;
            uint64_t const r = s[j] >> 11;
            uint64_t idx;
            if (piece_words) idx = (r & 3) * piece_stride + __umul64hi(r << 11, piece_words);
            else idx = __umul64hi(r << 11, words);
            uint32_t v = 0;
            if (MODE == 0) v = __builtin_nontemporal_load(&buf[idx]);
            if (MODE == 1) { v = buf[idx]; __builtin_nontemporal_store(v + 1, &buf[idx]); }
            if (MODE == 2) __builtin_nontemporal_store((uint32_t)i, &buf[idx]);
            if (MODE == 3) { v = __builtin_nontemporal_load(&buf[idx]); uint64_t const w2 = piece_words ? ((r >> 2) & 3) * piece_stride + __umul64hi(s[j] * 0x9E3779B97F4A7C15ull, piece_words) : __umul64hi(s[j] * 0x9E3779B97F4A7C15ull, words); __builtin_nontemporal_store(v + i, &buf[w2]); }
            if (MODE == 4) { uint64_t const a = idx & ~15ull; v = buf[a + (r & 7)]; uint32_t const v2 = buf[a + 8 + (r & 7)]; __builtin_nontemporal_store(v + 1, &buf[a + (r & 7)]); __builtin_nontemporal_store(v2, &buf[a + 8 + (r & 7)]); }
            if (MODE == 5) { uint4* const q = (uint4*)(buf + (idx & ~15ull)) + sub; uint4 t = *q; v = t.x; t.y += 1; *q = t; v = __shfl(v, threadIdx.x & ~3); }
            if (MODE == 6) { uint4* const q = (uint4*)(buf + (idx & ~15ull)) + sub; *q = make_uint4((uint32_t)i, 1, 2, 3); }
            if (MODE == 7) { uint4* const q = (uint4*)(buf + (idx & ~7ull)) + sub; uint4 t = *q; v = t.x; t.y += 1; *q = t; v = __shfl(v, threadIdx.x & ~1); }
            acc += v; s[j] += v & 1;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE>
static double run(uint32_t* buf, uint64_t bytes, uint64_t piece_bytes, uint32_t* sink)
{
    constexpr int INDEP = 4;
    int const iters = 500, blocks = 256 * 16;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    uint64_t const words = bytes / 4, pw = piece_bytes / 4, ps = piece_bytes ? (bytes / 4 / 4) : 0;
    hipLaunchKernelGGL((k_span<INDEP, MODE>), dim3(blocks), dim3(64), 0, 0, buf, words, pw, ps, iters / 8, sink);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_span<INDEP, MODE>), dim3(blocks), dim3(64), 0, 0, buf, words, pw, ps, iters, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    int const share = (MODE == 5 || MODE == 6) ? 4 : (MODE == 7 ? 2 : 1);
    return (double)blocks * 64 * iters * INDEP / share / ms / 1e6;
}

int main(int argc, char** argv)
{
    size_t fr = 0, tot = 0; CK(hipMemGetInfo(&fr, &tot));
    uint64_t gib = (fr >> 30) - 6; if (argc > 1) gib = strtoull(argv[1], nullptr, 10);
    printf("free %.1f GiB of %.1f; one allocation of %llu GiB\n", fr / 1073741824.0, tot / 1073741824.0, (unsigned long long)gib);
    uint32_t *buf, *sink;
    CK(hipMalloc(&buf, gib << 30)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0, gib << 30));
    printf("%-34s %10s %14s %12s %14s\n", "span (from the start)", "load", "ld+st same w", "store", "ld A + st B");
    uint64_t const spans[] = { 8, 24, 36, 72, 108, 144, 180, 216, 252, 272 };
    for (uint64_t sp : spans) {
        if (sp > gib) continue;
        printf("%4llu GiB                           %10.2f %14.2f %12.2f %14.2f   G ops/s\n", (unsigned long long)sp,
               run<0>(buf, sp << 30, 0, sink), run<1>(buf, sp << 30, 0, sink), run<2>(buf, sp << 30, 0, sink), run<3>(buf, sp << 30, 0, sink));
        fflush(stdout);
    }
    for (uint64_t sp : spans) {
        if (sp > gib || sp < 36) continue;
        printf("4 x 6 GiB pieces, %4llu GiB / 4 apart %10.2f %14.2f %12.2f %14.2f   G ops/s\n", (unsigned long long)sp,
               run<0>(buf, sp << 30, 6ull << 30, sink), run<1>(buf, sp << 30, 6ull << 30, sink), run<2>(buf, sp << 30, 6ull << 30, sink), run<3>(buf, sp << 30, 6ull << 30, sink));
        fflush(stdout);
    }
    printf("%-34s %12s %14s %14s %16s\n", "span", "2 sectors", "line ld+st", "line st", "sector ld+st");
    for (uint64_t sp : { 24ull, 144ull })
        if (sp <= gib) printf("%4llu GiB                           %12.2f %14.2f %14.2f %16.2f   G lines (sectors) /s\n", (unsigned long long)sp,
               run<4>(buf, sp << 30, 0, sink), run<5>(buf, sp << 30, 0, sink), run<6>(buf, sp << 30, 0, sink), run<7>(buf, sp << 30, 0, sink));
    // windows of 24 GiB at different places
    for (uint64_t at = 0; at + 24 <= gib; at += 48)
        printf("24 GiB window at %4llu GiB          %10.2f %14.2f\n", (unsigned long long)at, run<0>(buf + (at << 28), 24ull << 30, 0, sink), run<1>(buf + (at << 28), 24ull << 30, 0, sink));
    return 0;
}
