// Device-side helpers shared by the gfx950 kernels.
//
// All kernels are written for 64-lane wavefronts: one wavefront owns one LZ4
// block at a time, control state is kept wave-uniform (scalar registers) and
// the 64 lanes do the byte moving.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lz4mi.h"

namespace lz4mi {

constexpr uint32_t kWave = 64;

__device__ __forceinline__ uint32_t lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v) {
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) {
  return __builtin_amdgcn_readlane(v, l);
}
__device__ __forceinline__ uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// Explicit "all vector-memory operations done" (vmcnt(0) alone).  Placed where the wavefront has
// just consumed its last outstanding load, it costs nothing and tells the compiler's wait
// insertion that no load is pending, so it does not add conservative drains of its own in front
// of later stores and atomics (loads, stores and atomics share one in-order counter on gfx950).
__device__ __forceinline__ void vm_quiesce() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// Work-queue pop for a persistent wavefront.  Every lane takes part in the atomic (lanes
// other than 0 add zero) so that no per-lane branch sits in front of the cross-lane read:
// with `if (lane == 0)` here the compiler may thread lanes 1..63 past the pop of the next
// loop iteration, after which they would read a stale index forever.
__device__ __forceinline__ uint32_t next_block(uint32_t *counter, uint32_t lane) {
  uint32_t v = atomicAdd(counter, lane == 0 ? 1u : 0u);
  return __builtin_amdgcn_readfirstlane(v);
}

// Unaligned little-endian accesses (gfx950 global memory handles any byte alignment).
struct __attribute__((packed)) U16U { uint16_t v; };
struct __attribute__((packed)) U32U { uint32_t v; };
struct __attribute__((packed)) U64U { uint64_t v; };
struct __attribute__((packed)) U128U { uint32_t a, b, c, d; };

__device__ __forceinline__ uint32_t ld32(const uint8_t *p) { return ((const U32U *)p)->v; }
__device__ __forceinline__ uint64_t ld64(const uint8_t *p) { return ((const U64U *)p)->v; }
__device__ __forceinline__ uint32_t ld16(const uint8_t *p) { return ((const U16U *)p)->v; }
__device__ __forceinline__ void st16(uint8_t *p, uint32_t v) { ((U16U *)p)->v = (uint16_t)v; }
__device__ __forceinline__ void st32(uint8_t *p, uint32_t v) { ((U32U *)p)->v = v; }
__device__ __forceinline__ void st64(uint8_t *p, uint64_t v) { ((U64U *)p)->v = v; }
__device__ __forceinline__ U128U ld128(const uint8_t *p) { return *(const U128U *)p; }
__device__ __forceinline__ void st128(uint8_t *p, U128U v) { *(U128U *)p = v; }

// LDS-typed pointers keep ring accesses on ds_read/ds_write (a generic pointer merged with a
// global one would turn them into flat accesses).
typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4 lds_u128;

// Wave-cooperative copy of n bytes between non-overlapping ranges.
// Destination is brought to 16-byte alignment first so the body uses aligned
// 16-byte stores (1 KiB per wave-instruction); loads may be unaligned.  Every load of a
// group is issued before the first store that depends on it, 4 KiB per group, so a copy
// costs one memory round trip per 4 KiB instead of one per instruction.
__device__ __forceinline__ void wave_copy(uint8_t *dst, const uint8_t *src, uint32_t n, uint32_t lane) {
  if (n <= kWave) {
    if (lane < n) dst[lane] = src[lane];
    return;
  }
  const uint32_t head = (uint32_t)(-(uintptr_t)dst) & 15u;
  const uint32_t body = (n - head) & ~15u;
  const uint32_t tail = n - head - body;
  const uint8_t *s = src + head;
  uint8_t *d = dst + head;
  uint8_t hb = 0, tb = 0;
  if (lane < head) hb = src[lane];
  if (lane < tail) tb = s[body + lane];
  uint32_t off = lane * 16;
  // long copies keep 8 KiB in flight per round trip (one wavefront alone is bound by the latency
  // of a round trip, not by bandwidth)
  for (; off + 7 * 1024 < body; off += 8192) {
    U128U v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = ld128(s + off + 1024 * k);
#pragma unroll
    for (int k = 0; k < 8; k++) st128(d + off + 1024 * k, v[k]);
  }
  for (; off + 3 * 1024 < body; off += 4096) {
    U128U a = ld128(s + off), b = ld128(s + off + 1024), c = ld128(s + off + 2048), e = ld128(s + off + 3072);
    st128(d + off, a);
    st128(d + off + 1024, b);
    st128(d + off + 2048, c);
    st128(d + off + 3072, e);
  }
  U128U r0 = {0, 0, 0, 0}, r1 = r0, r2 = r0;
  const bool h0 = off < body, h1 = off + 1024 < body, h2 = off + 2048 < body;
  if (h0) r0 = ld128(s + off);
  if (h1) r1 = ld128(s + off + 1024);
  if (h2) r2 = ld128(s + off + 2048);
  if (lane < head) dst[lane] = hb;
  if (lane < tail) d[body + lane] = tb;
  if (h0) st128(d + off, r0);
  if (h1) st128(d + off + 1024, r1);
  if (h2) st128(d + off + 2048, r2);
}

// The same for copies of 16 KiB and more (a literal-only block, a stored block): 12 KiB in flight per round trip.
// A function of its own, not inlined: its twelve 16-byte registers per lane would otherwise count against every
// kernel that copies anything.  (What is left behind the last 12 KiB goes through wave_copy: taking it in one
// predicated round trip here as well measured slower, bench decode launch 0.462 -> 0.473 ms.)
#ifndef LZ4MI_COPY_LONG_K
#define LZ4MI_COPY_LONG_K 12  // 16-byte registers per lane: KiB in flight per round trip
#endif
__device__ __attribute__((noinline)) void wave_copy_long(uint8_t *dst, const uint8_t *src, uint32_t n, uint32_t lane) {
  // Both ranges are device memory: global pointers, so that the loop is global_load / global_store (through the
  // generic pointers of a function that is not inlined it would be flat instructions, which count against the LDS
  // counter as well and go through the aperture check).
  typedef __attribute__((address_space(1))) uint8_t g_u8;
  typedef u32x4 u32x4_any __attribute__((aligned(1)));  // the source may start at any byte
  typedef __attribute__((address_space(1))) u32x4_any g_v4any;
  typedef __attribute__((address_space(1))) u32x4 g_v4;
  const uint32_t head = (uint32_t)(-(uintptr_t)dst) & 15u;
  if (lane < head) dst[lane] = src[lane];
  const uint8_t *s = src + head;
  uint8_t *d = dst + head;
  const g_u8 *gs = (const g_u8 *)s;
  g_u8 *gd = (g_u8 *)d;
  uint32_t off = 0;
  constexpr int K = LZ4MI_COPY_LONG_K;
  for (; off + 1024 * K <= n - head; off += 1024 * K) {
    // streamed once: nontemporal on both sides (scripts/micro/copy_tp.hip: 5.3 against 5.1 TB/s chip-wide)
    u32x4 v[K];
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = __builtin_nontemporal_load((const g_v4any *)(gs + off + 16 * lane + 1024 * k));
#pragma unroll
    for (int k = 0; k < K; k++) __builtin_nontemporal_store(v[k], (g_v4 *)(gd + off + 16 * lane + 1024 * k));
  }
  wave_copy(d + off, s + off, n - head - off, lane);
}
constexpr uint32_t kCopyLong = 16384;

// Inclusive prefix sum over the 64 lanes: row shifts inside the 16-lane rows, then the two row
// broadcasts (lanes without a source add zero).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add(uint32_t v) {
  return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  v = dpp_add<0x111, 0xF>(v);  // row_shr:1
  v = dpp_add<0x112, 0xF>(v);  // row_shr:2
  v = dpp_add<0x114, 0xF>(v);  // row_shr:4
  v = dpp_add<0x118, 0xF>(v);  // row_shr:8
  v = dpp_add<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3
  return v;
}
// xxHash32 constants (lib/src/xxhash/xxh32.dart:3-7).
constexpr uint32_t XP1 = 0x9E3779B1u, XP2 = 0x85EBCA77u, XP3 = 0xC2B2AE3Du, XP4 = 0x27D4EB2Fu,
                   XP5 = 0x165667B1u;
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__device__ __forceinline__ uint32_t xxh_round(uint32_t acc, uint32_t x) {
  return rotl32(acc + x * XP2, 13) * XP1;
}
__device__ __forceinline__ uint32_t xxh_avalanche(uint32_t h) {
  h ^= h >> 15;
  h *= XP2;
  h ^= h >> 13;
  h *= XP3;
  h ^= h >> 16;
  return h;
}

}  // namespace lz4mi
