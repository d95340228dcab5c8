// xxHash32 on gfx950 (lib/src/xxhash/xxh32.dart:166-247).
//
// The hash is four independent serial lanes (v1..v4) over 16-byte stripes with
// no combine operator, so the only parallelism is (a) the four accumulators and
// (b) independent streams.  Batch kernel: four lanes per block, sixteen blocks
// per wavefront.  Serial kernel (content checksum of one stream): four lanes,
// loads issued sixteen stripes ahead.
#include "device_common.h"
#include "kernels.h"

namespace lz4mi {

// Tail + avalanche by one lane: `h` already holds the merged accumulators (or seed+P5) plus len.
__device__ __forceinline__ uint32_t xxh_finish(uint32_t h, const uint8_t *p, uint32_t rem) {
  uint32_t k = 0;
  while (k + 4 <= rem) {  // :221-225
    h += ld32(p + k) * XP3;
    h = rotl32(h, 17) * XP4;
    k += 4;
  }
  while (k < rem) {  // :227-231
    h += (uint32_t)p[k] * XP5;
    h = rotl32(h, 11) * XP1;
    k++;
  }
  return xxh_avalanche(h);
}

__device__ __forceinline__ uint32_t xxh_init_acc(uint32_t acc, uint32_t seed) {
  switch (acc) {
    case 0: return seed + XP1 + XP2;
    case 1: return seed + XP2;
    case 2: return seed;
    default: return seed - XP1;
  }
}

// Runs `stripes` rounds on this lane's accumulator: word (4*s + acc) of the stream.
__device__ __forceinline__ uint32_t xxh_consume(uint32_t v, const uint8_t *p, uint64_t stripes, uint32_t acc) {
  const uint8_t *q = p + 4 * acc;
  uint64_t s = 0;
  for (; s + 16 <= stripes; s += 16) {
    uint32_t w[16];
#pragma unroll
    for (int u = 0; u < 16; u++) w[u] = ld32(q + (s + u) * 16);
#pragma unroll
    for (int u = 0; u < 16; u++) v = xxh_round(v, w[u]);
  }
  for (; s < stripes; s++) v = xxh_round(v, ld32(q + s * 16));
  return v;
}

__device__ __forceinline__ uint32_t xxh_merge4(uint32_t v, uint32_t acc, uint32_t lane) {
  // rotl(v1,1)+rotl(v2,7)+rotl(v3,12)+rotl(v4,18), summed across the 4 lanes of the group
  const int rot = acc == 0 ? 1 : acc == 1 ? 7 : acc == 2 ? 12 : 18;
  uint32_t r = rotl32(v, rot);
  r += __shfl_xor(r, 1);
  r += __shfl_xor(r, 2);
  (void)lane;
  return r;
}

__global__ __launch_bounds__(256) void xxh32_batch_kernel(BatchArgs a, uint32_t *hash_out) {
  const uint32_t lane = lane_id();
  const uint32_t wave = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
  const uint32_t b = wave * 16 + lane / 4;
  const uint32_t acc = lane & 3;
  const bool live = b < a.nblocks;
  const uint8_t *p = a.src;
  uint32_t len = 0;
  if (live) {
    p = a.src + (a.src_off ? a.src_off[b] : (uint64_t)b * a.src_stride);
    len = block_src_len(a, b);
  }
  uint32_t h;
  if (len >= 16) {
    uint32_t v = xxh_consume(xxh_init_acc(acc, 0), p, len / 16, acc);
    h = xxh_merge4(v, acc, lane);
  } else {
    h = XP5;  // seed 0
  }
  if (live && acc == 0) {
    h += len;
    uint32_t done = len & ~15u;
    hash_out[b] = xxh_finish(h, p + done, len - done);
  }
}

// One stream (content checksum).  The four accumulators are one dependent chain each
// (add, rotate, multiply per 16-byte stripe); everything else is taken off that chain: all 64
// lanes fetch the stream 4 KiB at a time with coalesced 16-byte loads (the next chunk is in flight
// while the current one is hashed), pre-multiply every word by PRIME32_2 and park the products in
// LDS; lanes 0-3 then only do  v = rotl(v + y, 13) * PRIME32_1  with y read from LDS.
constexpr uint32_t kXChunk = 4096;  // bytes per step: 64 lanes x 4 x 16

// Runs `stripes` 16-byte stripes of one stream through the four accumulators; lanes 0-3 pass in
// and get back v1..v4 (acc = lane & 3), the other lanes only fetch.
__device__ __forceinline__ uint32_t xxh_stream_rounds(uint32_t v, const uint8_t *data, uint64_t stripes,
                                                      uint32_t *s_y, uint32_t lane, uint32_t acc) {
  const uint64_t chunks = stripes / (kXChunk / 16);
  U128U nx[4];
  if (chunks) {
#pragma unroll
    for (int k = 0; k < 4; k++) nx[k] = ld128(data + 1024 * k + 16 * lane);
  }
  for (uint64_t c = 0; c < chunks; c++) {
#pragma unroll
    for (int k = 0; k < 4; k++) {  // word w of the chunk sits at s_y[w]
      const u32x4 y = {nx[k].a * XP2, nx[k].b * XP2, nx[k].c * XP2, nx[k].d * XP2};
      *(lds_u128 *)((lds_u8 *)s_y + 1024 * k + 16 * lane) = y;
    }
    if (c + 1 < chunks) {
      const uint8_t *q = data + (c + 1) * kXChunk;
#pragma unroll
      for (int k = 0; k < 4; k++) nx[k] = ld128(q + 1024 * k + 16 * lane);
    }
    if (lane < 4) {
      const lds_u32 *yy = (const lds_u32 *)s_y + acc;
      for (uint32_t st = 0; st < kXChunk / 16; st += 16) {
        uint32_t y[16];
#pragma unroll
        for (int u = 0; u < 16; u++) y[u] = yy[(st + u) * 4];
#pragma unroll
        for (int u = 0; u < 16; u++) v = rotl32(v + y[u], 13) * XP1;
      }
    }
  }
  if (lane < 4) v = xxh_consume(v, data + chunks * kXChunk, stripes - chunks * (kXChunk / 16), acc);
  return v;
}

__global__ __launch_bounds__(64) void xxh32_serial_kernel(const uint8_t *data, uint64_t len, uint32_t seed,
                                                          uint32_t *out) {
  __shared__ __attribute__((aligned(16))) uint32_t s_y[kXChunk / 4];
  const uint32_t lane = lane_id();
  const uint32_t acc = lane & 3;
  uint32_t h;
  if (len >= 16) {
    uint32_t v = xxh_stream_rounds(xxh_init_acc(acc, seed), data, len / 16, s_y, lane, acc);
    h = xxh_merge4(lane < 4 ? v : 0u, acc, lane);
  } else {
    h = seed + XP5;
  }
  if (lane == 0) {
    h += (uint32_t)len;
    uint64_t done = len & ~(uint64_t)15;
    *out = xxh_finish(h, data + done, (uint32_t)(len - done));
  }
}

// Streaming form (Xxh32.update, lib/src/xxhash/xxh32.dart:83-118): advances v1..v4 in `state`
// over whole stripes; carry bytes and the digest stay with the caller.
__global__ __launch_bounds__(64) void xxh32_stripes_kernel(const uint8_t *data, uint64_t stripes, uint32_t *state) {
  __shared__ __attribute__((aligned(16))) uint32_t s_y[kXChunk / 4];
  const uint32_t lane = lane_id();
  const uint32_t acc = lane & 3;
  uint32_t v = xxh_stream_rounds(state[acc], data, stripes, s_y, lane, acc);
  if (lane < 4) state[acc] = v;
}

// Few blocks (large ones, as a rule: a frame of 4 MiB blocks has 256 per GiB): a wavefront per block,
// each hashing its block the way the single-stream kernel does (64 lanes fetch ahead into LDS, four
// lanes run the chains).  Sixteen blocks to a wavefront would leave most of the chip idle and every
// lane waiting out its own memory latency.
__global__ __launch_bounds__(256) void xxh32_batch_wide_kernel(BatchArgs a, uint32_t *hash_out) {
  __shared__ __attribute__((aligned(16))) uint32_t s_y[4][kXChunk / 4];
  const uint32_t lane = lane_id();
  const uint32_t w = uni(threadIdx.x / kWave);
  const uint32_t b = uni(blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave);
  if (b >= a.nblocks) return;
  const uint8_t *p = a.src + uni64(a.src_off ? a.src_off[b] : (uint64_t)b * a.src_stride);
  const uint32_t len = uni(block_src_len(a, b));
  const uint32_t acc = lane & 3;
  uint32_t h;
  if (len >= 16) {
    uint32_t v = xxh_stream_rounds(xxh_init_acc(acc, 0), p, len / 16, s_y[w], lane, acc);
    h = xxh_merge4(lane < 4 ? v : 0u, acc, lane);
  } else {
    h = XP5;  // seed 0
  }
  if (lane == 0) {
    h += len;
    const uint32_t done = len & ~15u;
    hash_out[b] = xxh_finish(h, p + done, len - done);
  }
}

void launch_xxh32_batch(const BatchArgs &a, uint32_t *d_hash, hipStream_t s) {
  if (a.nblocks == 0) return;
  if (a.nblocks <= 2048) {
    hipLaunchKernelGGL(xxh32_batch_wide_kernel, dim3((a.nblocks + 3) / 4), dim3(256), 0, s, a, d_hash);
    return;
  }
  uint32_t waves = (a.nblocks + 15) / 16;
  hipLaunchKernelGGL(xxh32_batch_kernel, dim3((waves + 3) / 4), dim3(256), 0, s, a, d_hash);
}

void launch_xxh32_serial(const uint8_t *d_data, uint64_t len, uint32_t seed, uint32_t *d_out, hipStream_t s) {
  hipLaunchKernelGGL(xxh32_serial_kernel, dim3(1), dim3(64), 0, s, d_data, len, seed, d_out);
}

void launch_xxh32_stripes(const uint8_t *d_data, uint64_t stripes, uint32_t *d_state, hipStream_t s) {
  hipLaunchKernelGGL(xxh32_stripes_kernel, dim3(1), dim3(64), 0, s, d_data, stripes, d_state);
}

}  // namespace lz4mi
