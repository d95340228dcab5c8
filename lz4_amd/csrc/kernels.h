// Host-visible launch interface of the gfx950 kernels (internal to liblz4mi).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lz4mi {

// Resolved form of lz4mi_batch: every pointer is device memory.
struct BatchArgs {
  const uint8_t *src;
  const uint64_t *src_off;
  const uint32_t *src_len;
  uint64_t src_stride;
  uint32_t src_len_uniform;
  uint64_t src_total;
  uint8_t *dst;
  const uint64_t *dst_off;
  uint64_t dst_stride;
  const uint32_t *dst_cap;
  uint32_t dst_cap_uniform;
  const uint8_t *dict;
  uint32_t dict_len;
  uint32_t nblocks;
  uint32_t *out_len;
  int32_t *status;
  // Linked blocks (lz4_frame_encoder.dart:170-206): blocks are contiguous (src_stride == src_len_uniform)
  // and block b >= 1 takes the 65535 bytes in front of it as its dictionary; block 0 takes `dict`.
  uint32_t linked = 0;
  // Nonzero: blocks follow the lz4 block format's end-of-block rules instead of the reference's
  // (which ends blocks inside matches): see EncIn::plimit / elimit.
  uint32_t conformant = 0;
  // Decoder only: per block, nonzero = the block is stored (a frame record with bit 31 set,
  // lz4_frame_decoder.dart:268-271): that many source bytes are copied instead of decoded.
  const uint32_t *stored_len = nullptr;
};

__host__ __device__ inline uint32_t block_src_len(const BatchArgs &a, uint32_t b) {
  if (a.src_len) return a.src_len[b];
  uint64_t start = (uint64_t)b * a.src_len_uniform;
  uint64_t left = a.src_total > start ? a.src_total - start : 0;
  return left < a.src_len_uniform ? (uint32_t)left : a.src_len_uniform;
}

// Position field / generation tag split of a hash-table entry (encoder).
constexpr uint32_t kPosBits = 24;               // positions < 16 MiB (block + 64 KiB dictionary)
constexpr uint32_t kPosMask = (1u << kPosBits) - 1;
constexpr uint32_t kGenMax = (1u << (32 - kPosBits)) - 1;
constexpr uint32_t kHashEntries = 65536;

struct EncodeScratch {
  uint32_t *tables;    // nslots * 65536 entries
  uint32_t *slot_gen;  // nslots generation counters
  uint32_t *dict_table;// 65536 entries primed from the shared dictionary (or null)
  uint32_t nslots;
};

struct ChainArgs {
  const uint8_t *src;           // frame bytes (device)
  const uint64_t *payload_off;  // per block: payload offset in src
  const uint32_t *payload_len;
  const uint32_t *raw;          // per block: nonzero = stored uncompressed
  uint32_t nblocks;
  uint32_t block_max;
  uint8_t *dst;                 // contiguous frame output
  uint64_t dst_limit;           // no block starts behind this offset (the buffer holds block_max more)
  const uint8_t *dict;
  uint32_t dict_len;
  uint32_t *out_len;
  int32_t *status;
};

void launch_decode_chain(const ChainArgs &c, hipStream_t s);

// Linked blocks without the chain (decode_linked_kernel.hip): blocks [first, first + nblocks) of a frame.
struct LinkArgs {
  const uint8_t *src;           // frame bytes (device)
  const uint64_t *payload_off;  // per block of the frame: payload offset in src
  const uint32_t *payload_len;
  const uint32_t *raw;          // per block: nonzero = stored uncompressed
  uint32_t first, nblocks;
  uint32_t block_max;
  uint8_t *dst;                 // frame output (second walk)
  const uint64_t *start;        // per block: offset of its output in dst (second walk)
  uint64_t chunk_start;         // output offset that dist[0] stands for
  uint32_t *dist;               // one zeroed word per output byte of the chunk (second walk)
  uint32_t dict_len;
  uint32_t *out_len;            // per block: decoded size
  int32_t *status;              // per block: 0, or 1 = the block is not for this path
};
// plan = false: sizes only; true: literals into dst, match distances into dist
void launch_link_walk(const LinkArgs &a, bool plan, hipStream_t s);
// one doubling round over n words; done: one zeroed byte per tile (link_tiles); *moved is set when a word changed
void launch_link_jump(uint32_t *dist, uint64_t n, uint8_t *done, uint32_t *moved, hipStream_t s);
void launch_link_gather(const uint32_t *dist, uint64_t n, uint8_t *out, uint64_t chunk_start, const uint8_t *dict,
                        uint32_t dict_len, hipStream_t s);
uint64_t link_dist_words(uint64_t n);  // words to allocate (and zero) for n output bytes
uint64_t link_tiles(uint64_t n);
// counter: the queue head; order: TWO words per block (queue order, and scratch for building it), or null:
// index order; order_cnt: two words behind counter in the same array.  The launch zeroes the words it uses (on s).
// The grid is sized to the wavefronts that stay resident.  form: 0 by launch size, 1 one wavefront per block,
// 2 two wavefronts per block, negative: the process default (set_decode_form).  Returns the first error of
// the fills and launches it issued.
hipError_t launch_decode(const BatchArgs &a, bool strict, uint32_t *counter, uint32_t *order_cnt, uint32_t *order,
                         uint32_t cu_count, int form, hipStream_t s);
// process default of the decode form (0, 1, 2 as above); returns the previous setting
int set_decode_form(int form);
void launch_encode_fast(const BatchArgs &a, uint32_t accel_shift, const EncodeScratch &sc, uint32_t *counter,
                        uint32_t grid, hipStream_t s);
void launch_dict_prime(const uint8_t *dict, uint32_t dict_len, uint32_t *dict_table, hipStream_t s);
void launch_xxh32_batch(const BatchArgs &a, uint32_t *d_hash, hipStream_t s);
void launch_xxh32_serial(const uint8_t *d_data, uint64_t len, uint32_t seed, uint32_t *d_out, hipStream_t s);
void launch_xxh32_stripes(const uint8_t *d_data, uint64_t stripes, uint32_t *d_state, hipStream_t s);

}  // namespace lz4mi
