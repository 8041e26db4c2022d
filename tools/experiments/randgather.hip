// randgather.hip -- how many random 64-byte DRAM transactions per second does one MI355X sustain?
// (the ceiling of k_zstd_match's hash-table probes).  Every lane runs INDEP independent chains of
// dependent 4-byte loads at pseudo-random addresses inside a footprint of F bytes.
//   hipcc --offload-arch=gfx950 -O3 -o randgather tools/randgather.hip && ./randgather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// WR: 0 load only; 1 nt load + nt store (neighbour word); 2 plain load + plain store (same word);
//     3 nt load + nt store (same word); 4 plain load + nt store (same word); 5 atomic exchange (probe+insert in one op);
//     6 store only (nt); 7 store only (plain); 8 full 32-byte sector store (2 x dwordx4); 9 32-byte sector load,
//     then the same sector stored back whole; 10 full 64-byte line store; 11 16-byte store
template <int INDEP, int WR>
__global__ void __launch_bounds__(64) k_gather(uint32_t* buf, uint64_t mask, int iters, uint32_t* sink)
{
    uint64_t s[INDEP]; uint32_t acc = 0;
    uint64_t const gid = (uint64_t)blockIdx.x * 64 + threadIdx.x;
#pragma unroll
    for (int j = 0; j < INDEP; j++) s[j] = (gid * INDEP + j) * 0x9E3779B97F4A7C15ull + 12345;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < INDEP; j++) {
            s[j] = s[j] * 6364136223846793005ull + 1442695040888963407ull;
            uint64_t const idx = ((s[j] >> 20) & mask);
            uint32_t v = 0;
            if (WR == 0 || WR == 1 || WR == 3) v = __builtin_nontemporal_load(&buf[idx]);
            if (WR == 2 || WR == 4) v = buf[idx];
            if (WR == 1) __builtin_nontemporal_store(v + 1, &buf[idx ^ 1]);
            if (WR == 2) buf[idx] = v + 1;
            if (WR == 3 || WR == 4) __builtin_nontemporal_store(v + 1, &buf[idx]);
            if (WR == 5) v = __hip_atomic_exchange(&buf[idx], (uint32_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (WR == 6) __builtin_nontemporal_store((uint32_t)i, &buf[idx]);
            if (WR == 7) buf[idx] = (uint32_t)i;
            if (WR == 8 || WR == 9 || WR == 10 || WR == 11) {
                uint4* const sec = (uint4*)(buf + (idx & ~(uint64_t)(WR == 10 ? 15 : 7)));      // 32-byte (64-byte) aligned
                uint4 a = make_uint4((uint32_t)i, 1, 2, 3), b = a, c = a, d = a;
                if (WR == 9) { a = sec[0]; b = sec[1]; v = a.x + b.w; a.y += 1; }
                sec[0] = a;
                if (WR != 11) sec[1] = b;
                if (WR == 10) { sec[2] = c; sec[3] = d; }
            }
            acc += v; s[j] += v & 1;            // value-dependent chain
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int INDEP, int WR>
static void run(uint32_t* buf, uint64_t bytes, int wavesPerCu, uint32_t* sink)
{
    int const iters = 2000 / INDEP;
    uint64_t const mask = bytes / 4 - 1;
    int const blocks = 256 * wavesPerCu;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_gather<INDEP, WR>), dim3(blocks), dim3(64), 0, 0, buf, mask, iters / 4, sink);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_gather<INDEP, WR>), dim3(blocks), dim3(64), 0, 0, buf, mask, iters, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double const n = (double)blocks * 64 * iters * INDEP;
    static const char* const names[] = { "nt load", "nt ld + nt st neighbour", "ld + st same word", "nt ld + nt st same word",
                                         "ld + nt st same word", "atomic exchange", "nt store only", "store only",
                                         "32-byte sector store", "32-byte ld + st back whole", "64-byte line store", "16-byte store" };
    printf("footprint %6.2f GiB  waves/CU %2d  indep %d  %-24s : %7.2f G ops/s\n",
           bytes / 1073741824.0, wavesPerCu, INDEP, names[WR], n / ms / 1e6);
    fflush(stdout);
}

int main()
{
    uint64_t const maxBytes = 32ull << 30;
    uint32_t *buf, *sink;
    CK(hipMalloc(&buf, maxBytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0, maxBytes));
    uint64_t const sizes[] = { 8ull << 30, 32ull << 30 };
    for (uint64_t f : sizes) {
        run<4, 0>(buf, f, 16, sink); run<4, 7>(buf, f, 16, sink); run<4, 11>(buf, f, 16, sink); run<4, 8>(buf, f, 16, sink);
        run<4, 10>(buf, f, 16, sink); run<4, 2>(buf, f, 16, sink); run<4, 9>(buf, f, 16, sink);
    }
    return 0;
}
