// kx_wave.h -- wave64 primitives for gfx950 (CDNA4) used by every kernel body.
//
// Kernel bodies (zstd_match.h, zstd_entropy.h, zstd_decode.h) are written
// against this small vocabulary only, so that tests/emu/ can run the very
// same bodies lane-for-lane on the CPU (a lock-step fiber emulator that
// shadows this header) and diff every stage against oracle/.  This file is
// the product definition: plain HIP for gfx950, nothing else.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint8_t  u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

#define KX_DEV __device__ __forceinline__
#define KX_DEV_NOINLINE __device__ __noinline__
#define KX_MEMBER __device__ __forceinline__       // member functions (the emulator's KX_DEV says "static")
#define KX_SHARED __shared__

// ---- identity ---------------------------------------------------------
KX_DEV int kx_lane() { return (int)(threadIdx.x & 63); }
KX_DEV u32 kx_block() { return blockIdx.x; }
KX_DEV u32 kx_nblocks() { return gridDim.x; }
KX_DEV int kx_wave() { return (int)(threadIdx.x >> 6); }         // wave index inside the workgroup
KX_DEV int kx_nwaves() { return (int)(blockDim.x >> 6); }
KX_DEV void kx_block_sync() { __syncthreads(); }                 // workgroup barrier (multi-wave kernels)

// ---- cross-lane (must be called from wave-uniform control flow) -------
KX_DEV u64 kx_ballot(bool p) { return __ballot(p); }
KX_DEV bool kx_any(bool p) { return __ballot(p) != 0ull; }
KX_DEV bool kx_all(bool p) { return __ballot(!p) == 0ull; }
// value held by lane `src` (0..63); ds_bpermute_b32
KX_DEV u32 kx_shfl(u32 v, int src) { return (u32)__builtin_amdgcn_ds_bpermute((src & 63) << 2, (int)v); }
// orders LDS/global traffic between the lanes of the (single-wave) workgroup
// value held by lane k (a compile-time constant): v_readlane_b32, the result is wave-uniform
KX_DEV u32 kx_bcast(u32 v, int k) { return (u32)__builtin_amdgcn_readlane((int)v, k); }
// value held by lane K of the caller's quad (lanes 4 q .. 4 q + 3): one DPP move, no LDS
template <int K> KX_DEV u32 kx_quad_bcast(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, K * 0x55, 0xF, 0xF, true); }
// orders LDS traffic between the four lanes of a quad: on the GPU they are lanes of one wave (program order is enough, this only
// stops the compiler); the emulator's lanes are fibers and meet here
KX_DEV void kx_quad_sync() { __builtin_amdgcn_wave_barrier(); }
KX_DEV void kx_sync() { __syncthreads(); }
// The lanes of a wave execute in lock step and the vector-memory pipeline keeps
// one wave's accesses to an address in program order, so memory written by some
// lanes in one instruction is visible to other lanes of the same wave in a later
// one; this only stops the compiler from moving code across the point.
KX_DEV void kx_lockstep() { __builtin_amdgcn_wave_barrier(); }

// ---- memory -----------------------------------------------------------
// gfx950 global memory takes unaligned dword/dwordx2 accesses; hipcc emits
// one global_load_dwordx2 for these.
KX_DEV u64 kx_ld64(const u8* p) { u64 v; __builtin_memcpy(&v, p, 8); return v; }
KX_DEV u32 kx_ld32(const u8* p) { u32 v; __builtin_memcpy(&v, p, 4); return v; }
KX_DEV u32 kx_ld16(const u8* p) { u16 v; __builtin_memcpy(&v, p, 2); return v; }
KX_DEV void kx_st64(u8* p, u64 v) { __builtin_memcpy(p, &v, 8); }
KX_DEV void kx_st32(u8* p, u32 v) { __builtin_memcpy(p, &v, 4); }
KX_DEV void kx_st16(u8* p, u32 v) { u16 x = (u16)v; __builtin_memcpy(p, &x, 2); }
// 16-byte aligned store (one global_store_dwordx4): four neighbouring lanes fill a whole 64-byte line
struct alignas(16) KxU128 { u64 a, b; };
KX_DEV void kx_st128(void* p, u64 a, u64 b) { KxU128 v; v.a = a; v.b = b; *(KxU128*)p = v; }

// sixteen bytes as four words: the global side may be unaligned (one global_load/store_dwordx4), the LDS side is 16-byte aligned
struct alignas(16) KxQuad { u32 x, y, z, w; };
KX_DEV KxQuad kx_ld128u(const u8* p) { KxQuad q; __builtin_memcpy(&q, p, 16); return q; }
KX_DEV void kx_st128u(u8* p, const KxQuad& q) { __builtin_memcpy(p, &q, 16); }

KX_DEV u32 kx_ld_nt(const u32* p) { return __builtin_nontemporal_load(p); }
KX_DEV void kx_st_nt(u32* p, u32 v) { __builtin_nontemporal_store(v, p); }
KX_DEV u32 kx_atomic_add(u32* p, u32 v) { return atomicAdd(p, v); }
KX_DEV void kx_atomic_or(u32* p, u32 v) { atomicOr(p, v); }
KX_DEV void kx_lds_inc(u32* p) { atomicAdd(p, 1u); }
KX_DEV u32 kx_lds_add(u32* p, u32 v) { return atomicAdd(p, v); }   // returns the value before
KX_DEV void kx_lds_or(u32* p, u32 v) { atomicOr(p, v); }

// The optimiser must not look through this value (keeps a lane-serial loop ONE loop: when it threads the paths of an
// iteration into separate nested loops, the lanes of a wave wait for one another at every inner loop's exit).
#define KX_OPAQUE(x) __asm__ volatile("" : "+v"(x))
// The memory behind p is read and written here as far as the optimiser knows (a private array stays an array in memory).
#define KX_ESCAPE(p) __asm__ volatile("" : : "v"(p) : "memory")

// 100 MHz wall clock of the device (diagnostics only)
KX_DEV u64 kx_realtime() { return __builtin_amdgcn_s_memrealtime(); }
// statistics hook of the CPU emulator (tests/emu shadows this header); nothing on the GPU
#define KX_STAT(slot, v) ((void)0)

// ---- bit tricks -------------------------------------------------------
KX_DEV u32 kx_alignbit(u32 hi, u32 lo, u32 s) { return __builtin_amdgcn_alignbit(hi, lo, s); }      // low 32 bits of (hi : lo) >> (s & 31)
KX_DEV u32 kx_alignbyte(u32 hi, u32 lo, u32 bytes) { return __builtin_amdgcn_alignbyte(hi, lo, bytes); }   // ({hi,lo} >> 8*bytes) & 0xffffffff
KX_DEV u32 kx_umulhi(u32 a, u32 b) { return __umulhi(a, b); }
KX_DEV u32 kx_ctz32(u32 v) { return (u32)__builtin_ctz(v); }
KX_DEV u32 kx_ctz64(u64 v) { return (u32)__builtin_ctzll(v); }
KX_DEV u32 kx_brev32(u32 v) { return __builtin_bitreverse32(v); }      // v_bfrev_b32
KX_DEV u32 kx_clz32(u32 v) { return (u32)__builtin_clz(v); }
KX_DEV u32 kx_clz64(u64 v) { return (u32)__builtin_clzll(v); }
KX_DEV u32 kx_hb32(u32 v) { return 31u - (u32)__builtin_clz(v); }
KX_DEV u32 kx_popc64(u64 v) { return (u32)__builtin_popcountll(v); }
