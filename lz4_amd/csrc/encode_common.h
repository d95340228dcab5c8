// Pieces shared by the fast and HC block encoders: the dict||src position space, the
// reference's hash (lib/src/block/lz4_block_encoder.dart:167-170) and sequence emission
// (_writeSequence / _writeLastLiterals / _writeLength, :172-223; the HC encoder carries an
// identical copy, lib/src/hc/lz4_hc_block_encoder.dart:212-267).
#pragma once

#include "device_common.h"
#include "kernels.h"

namespace lz4mi {

struct EncIn {
  const uint8_t *dict;  // last <=65535 dictionary bytes (positions [0, D))
  const uint8_t *src;   // block bytes (positions [D, total))
  uint32_t D;
  uint32_t total;
  // End-of-block limits: a position p is searched while p + 4 <= plimit and a match may run up
  // to elimit.  Both equal `total` for the reference's behaviour (:95, :117-121); the optional
  // spec-conformant mode pulls them in (no match starts in the last 12 bytes, the last 5 bytes
  // stay literals, blocks under 13 bytes are literals only).
  uint32_t plimit;
  uint32_t elimit;
};

__device__ __forceinline__ void set_end_limits(EncIn &in, uint32_t conformant) {
  const uint32_t n = in.total - in.D;
  in.plimit = in.total;
  in.elimit = in.total;
  if (conformant) {
    in.plimit = n >= 13 ? in.total - 8 : 0u;
    in.elimit = n >= 5 ? in.total - 5 : 0u;
  }
}

template <bool HAS_DICT>
__device__ __forceinline__ uint32_t in_byte(const EncIn &in, uint32_t pos) {
  if (HAS_DICT && pos < in.D) return in.dict[pos];
  return in.src[pos - in.D];
}

template <bool HAS_DICT>
__device__ __forceinline__ uint32_t in_ld32(const EncIn &in, uint32_t pos) {
  if (!HAS_DICT) return ld32(in.src + pos);
  if (pos >= in.D) return ld32(in.src + (pos - in.D));
  if (pos + 4 <= in.D) return ld32(in.dict + pos);
  uint32_t v = 0;
  for (uint32_t k = 0; k < 4; k++) v |= in_byte<true>(in, pos + k) << (8 * k);
  return v;
}

#ifdef LZ4MI_TIMING_HASH_BITS  // timing-only builds (wrong output): the same work on a smaller table footprint
__device__ __forceinline__ uint32_t hash16(uint32_t seq) { return ((seq * 2654435761u) >> 16) & ((1u << LZ4MI_TIMING_HASH_BITS) - 1u); }
#else
__device__ __forceinline__ uint32_t hash16(uint32_t seq) { return (seq * 2654435761u) >> 16; }
#endif

__device__ __forceinline__ uint32_t sched_next(uint32_t o, uint32_t sh) {
  return o < 0x40000000u ? o + 1u + (o >> sh) : o;
}

// A table slot is only ever touched by the one wavefront that owns it, so lookups need no wider
// scope than the workgroup: the load is served by the XCD's L2, where the wavefront's own
// atomic-max inserts were performed (an agent-scope load would go past L2 to memory every time).
__device__ __forceinline__ uint32_t tbl_load(const uint32_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void tbl_store(uint32_t *p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void tbl_max(uint32_t *p, uint32_t v) {
  (void)__hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Lanes of one wave-instruction that hold the same 16-bit hash as this lane (among `vmask`).
__device__ __forceinline__ uint64_t same_hash_lanes(uint32_t h, uint64_t vmask) {
  uint64_t same = vmask;
  for (uint32_t bit = 0; bit < 16; bit++) {
    bool set = (h >> bit) & 1u;
    uint64_t m = ballot(set);
    same &= set ? m : ~m;
  }
  return same;
}

// Table insert for up to 64 ascending positions at once.  Every insert the encoder makes is at
// a larger position than the bucket holds, and wave-instructions of one wavefront reach memory
// in order, so a plain store is enough once the lanes of THIS instruction that share a bucket
// have been reduced to the last one (largest position).  Plain stores stay in the XCD's L2;
// device-scope atomics would execute at the memory side and evict the line.
__device__ __forceinline__ void tbl_insert(uint32_t *T, uint32_t h, uint32_t v, bool valid, uint32_t lane) {
  const uint64_t vmask = ballot(valid);
  const uint64_t same = same_hash_lanes(h, vmask);
  const uint64_t later = same & ~((2ull << lane) - 1ull);
  if (valid && !later) T[h] = v;
}

// 255-run length extension (:216-223): q bytes of 255 then the remainder.
__device__ __forceinline__ uint32_t emit_len(uint8_t *out, uint32_t op, uint32_t v, uint32_t lane) {
  uint32_t q = v / 255u, r = v - q * 255u;
  for (uint32_t k = lane; k <= q; k += kWave) out[op + k] = k < q ? (uint8_t)255 : (uint8_t)r;
  return op + q + 1;
}


// token, literal length extension, literals, offset, match length extension
__device__ __forceinline__ uint32_t emit_sequence(uint8_t *out, uint32_t op, const uint8_t *lit_src, uint32_t lit,
                                                  uint32_t dist, uint32_t ml, uint32_t lane) {
  const uint32_t m4 = ml - 4;
  if (lane == 0) out[op] = (uint8_t)(((lit < 15 ? lit : 15u) << 4) | (m4 < 15 ? m4 : 15u));
  op++;
  if (lit >= 15) op = emit_len(out, op, lit - 15, lane);
  if (lit) wave_copy(out + op, lit_src, lit, lane);
  op += lit;
  if (lane == 0) {
    out[op] = (uint8_t)dist;
    out[op + 1] = (uint8_t)(dist >> 8);
  }
  op += 2;
  if (m4 >= 15) op = emit_len(out, op, m4 - 15, lane);
  return op;
}

__device__ __forceinline__ uint32_t emit_last_literals(uint8_t *out, uint32_t op, const uint8_t *lit_src,
                                                       uint32_t last, uint32_t lane) {
  if (lane == 0) out[op] = (uint8_t)((last < 15 ? last : 15u) << 4);
  op++;
  if (last >= 15) op = emit_len(out, op, last - 15, lane);
  wave_copy(out + op, lit_src, last, lane);
  return op + last;
}

}  // namespace lz4mi
