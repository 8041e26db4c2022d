// tests/emu/kx_wave.h -- TEST INFRASTRUCTURE.  Shadows
// kompressor_amd/csrc/kx_wave.h (this directory comes first on the include
// path of the emulator build only) so that the kernel bodies compile with g++
// and run on a lock-step 64-lane fiber emulator.  Never part of the product.
#pragma once
#include <stdint.h>
#include <string.h>
#include <stddef.h>

typedef uint8_t  u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

#define KX_DEV static inline
#define KX_DEV_NOINLINE static
#define KX_MEMBER inline
#define KX_SHARED static

namespace kxemu {
enum { OP_BALLOT = 1, OP_SHFL = 2, OP_SYNC = 3, OP_BLOCK_SYNC = 4, OP_QUAD = 5 };
extern int cur_lane, cur_wave, waves_per_block; extern u32 cur_block, num_blocks;
u64 arrive(int op, u64 a, u64 b);
}

KX_DEV int kx_lane() { return kxemu::cur_lane; }
KX_DEV u32 kx_block() { return kxemu::cur_block; }
KX_DEV u32 kx_nblocks() { return kxemu::num_blocks; }
KX_DEV int kx_wave() { return kxemu::cur_wave; }                 // wave index inside the workgroup
KX_DEV int kx_nwaves() { return kxemu::waves_per_block; }
KX_DEV void kx_block_sync() { kxemu::arrive(kxemu::OP_BLOCK_SYNC, 0, 0); }   // workgroup barrier (multi-wave kernels)

KX_DEV u64 kx_ballot(bool p) { return kxemu::arrive(kxemu::OP_BALLOT, p ? 1 : 0, 0); }
KX_DEV bool kx_any(bool p) { return kx_ballot(p) != 0; }
KX_DEV bool kx_all(bool p) { return kx_ballot(!p) == 0; }
KX_DEV u32 kx_shfl(u32 v, int src) { return (u32)kxemu::arrive(kxemu::OP_SHFL, v, (u64)(src & 63)); }
KX_DEV u32 kx_bcast(u32 v, int k) { return kx_shfl(v, k); }
template <int K> KX_DEV u32 kx_quad_bcast(u32 v) { return (u32)kxemu::arrive(kxemu::OP_QUAD, v, (u64)K); }
KX_DEV void kx_quad_sync() { kxemu::arrive(kxemu::OP_QUAD, 0, 0); }
KX_DEV void kx_sync() { kxemu::arrive(kxemu::OP_SYNC, 0, 0); }
KX_DEV void kx_lockstep() { kxemu::arrive(kxemu::OP_SYNC, 1, 0); }

KX_DEV u64 kx_ld64(const u8* p) { u64 v; memcpy(&v, p, 8); return v; }
KX_DEV u32 kx_ld32(const u8* p) { u32 v; memcpy(&v, p, 4); return v; }
KX_DEV u32 kx_ld16(const u8* p) { u16 v; memcpy(&v, p, 2); return v; }
KX_DEV void kx_st64(u8* p, u64 v) { memcpy(p, &v, 8); }
KX_DEV void kx_st32(u8* p, u32 v) { memcpy(p, &v, 4); }
KX_DEV void kx_st16(u8* p, u32 v) { u16 x = (u16)v; memcpy(p, &x, 2); }
KX_DEV void kx_st128(void* p, u64 a, u64 b) { memcpy(p, &a, 8); memcpy((u8*)p + 8, &b, 8); }

struct alignas(16) KxQuad { u32 x, y, z, w; };
KX_DEV KxQuad kx_ld128u(const u8* p) { KxQuad q; memcpy(&q, p, 16); return q; }
KX_DEV void kx_st128u(u8* p, const KxQuad& q) { memcpy(p, &q, 16); }

KX_DEV u32 kx_ld_nt(const u32* p) { return *p; }
KX_DEV void kx_st_nt(u32* p, u32 v) { *p = v; }
KX_DEV u32 kx_atomic_add(u32* p, u32 v) { u32 o = *p; *p = o + v; return o; }
KX_DEV void kx_atomic_or(u32* p, u32 v) { *p |= v; }
KX_DEV void kx_lds_inc(u32* p) { *p += 1; }
KX_DEV u32 kx_lds_add(u32* p, u32 v) { u32 const o = *p; *p = o + v; return o; }
KX_DEV void kx_lds_or(u32* p, u32 v) { *p |= v; }

KX_DEV u64 kx_realtime() { return 0; }
namespace kxemu { extern u64 stat[64]; }
#define KX_STAT(slot, v) (kxemu::stat[slot] += (u64)(v))

#define KX_OPAQUE(x) __asm__ volatile("" : "+r"(x))
#define KX_ESCAPE(p) __asm__ volatile("" : : "r"(p) : "memory")

KX_DEV u32 kx_alignbit(u32 hi, u32 lo, u32 s) { return (u32)((((u64)hi << 32) | lo) >> (s & 31)); }
KX_DEV u32 kx_alignbyte(u32 hi, u32 lo, u32 bytes) { return (u32)((((u64)hi << 32) | lo) >> (8 * (bytes & 3))); }
KX_DEV u32 kx_umulhi(u32 a, u32 b) { return (u32)(((u64)a * b) >> 32); }
KX_DEV u32 kx_ctz32(u32 v) { return (u32)__builtin_ctz(v); }
KX_DEV u32 kx_ctz64(u64 v) { return (u32)__builtin_ctzll(v); }
KX_DEV u32 kx_brev32(u32 v) { u32 r = 0; for (int i = 0; i < 32; i++) { r = (r << 1) | (v & 1u); v >>= 1; } return r; }
KX_DEV u32 kx_clz32(u32 v) { return (u32)__builtin_clz(v); }
KX_DEV u32 kx_clz64(u64 v) { return (u32)__builtin_clzll(v); }
KX_DEV u32 kx_hb32(u32 v) { return 31u - (u32)__builtin_clz(v); }
KX_DEV u32 kx_popc64(u64 v) { return (u32)__builtin_popcountll(v); }
