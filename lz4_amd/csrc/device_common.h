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
__device__ __forceinline__ uint64_t ballot(bool p) { return __ballot(p); }

// Work-queue pop for a persistent wavefront.  Every lane takes part in the atomic (lanes
// other than 0 add zero) so that no per-lane branch sits in front of the cross-lane read:
// with `if (lane == 0)` here the compiler may thread lanes 1..63 past the pop of the next
// loop iteration, after which they would read a stale index forever.
__device__ __forceinline__ uint32_t next_block(uint32_t *counter, uint32_t lane) {
  uint32_t v = atomicAdd(counter, lane == 0 ? 1u : 0u);
  return __builtin_amdgcn_readfirstlane(v);
}

// Unaligned little-endian accesses (gfx950 global memory handles any byte alignment).
struct __attribute__((packed)) U32U { uint32_t v; };
struct __attribute__((packed)) U64U { uint64_t v; };
struct __attribute__((packed)) U128U { uint32_t a, b, c, d; };

__device__ __forceinline__ uint32_t ld32(const uint8_t *p) { return ((const U32U *)p)->v; }
__device__ __forceinline__ uint64_t ld64(const uint8_t *p) { return ((const U64U *)p)->v; }
__device__ __forceinline__ U128U ld128(const uint8_t *p) { return *(const U128U *)p; }
__device__ __forceinline__ void st128(uint8_t *p, U128U v) { *(U128U *)p = v; }

// Wave-cooperative copy of n bytes between non-overlapping ranges.
// Destination is brought to 16-byte alignment first so the body uses aligned
// 16-byte stores (1 KiB per wave-instruction); loads may be unaligned.
__device__ __forceinline__ void wave_copy(uint8_t *dst, const uint8_t *src, uint32_t n, uint32_t lane) {
  if (n <= kWave) {
    if (lane < n) dst[lane] = src[lane];
    return;
  }
  uint32_t head = (uint32_t)(-(uintptr_t)dst) & 15u;
  if (lane < head) dst[lane] = src[lane];
  dst += head;
  src += head;
  n -= head;
  uint32_t body = n & ~15u;
  for (uint32_t off = lane * 16; off < body; off += kWave * 16) st128(dst + off, ld128(src + off));
  uint32_t tail = n - body;
  if (lane < tail) dst[body + lane] = src[body + lane];
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
