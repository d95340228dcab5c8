// Host-side frame helpers shared by the one-shot and the streaming frame entry points
// (internal to liblz4mi).
#pragma once

#include <vector>

#include "api_common.h"

namespace lz4mi {

constexpr uint32_t kFrameMagic = 0x184D2204u, kSkippableBase = 0x184D2A50u, kLegacyMagic = 0x184C2102u;
constexpr size_t kLegacyBlockMax = 8u << 20;  // lz4_frame_encoder.dart:13, lz4_frame_decoder.dart:88
constexpr size_t kHistoryWindow = 64u << 10;  // lz4_frame_encoder.dart:151

inline uint32_t rd32(const uint8_t *p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
inline void wr32(uint8_t *p, uint32_t v) {
  p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}
inline bool is_skippable_magic(uint32_t m) { return (m & 0xFFFFFFF0u) == kSkippableBase; }
inline bool is_frame_boundary(uint32_t m) {  // lz4_frame_decoder.dart:150-155
  return m == kFrameMagic || m == kLegacyMagic || is_skippable_magic(m);
}

uint32_t descriptor_hash(const uint8_t *p, size_t len);
int32_t validate_options(const lz4mi_frame_options *o);
size_t build_header(const lz4mi_frame_options *o, uint8_t *h);

// Compresses every block of a device-resident range and plans its frame record; the records
// are then written by shard_write at the given offsets.
int32_t shard_compress(lz4mi_ctx *c, const uint8_t *d_src, uint64_t n, const lz4mi_frame_options *o,
                       const uint8_t *d_dict, uint32_t dict_len, uint32_t *nblocks_out);
int32_t shard_write(lz4mi_ctx *c, const uint64_t *d_rec_off, uint8_t *d_dst);
// shard_compress + offsets + shard_write: block records of [d_src, d_src+n) packed from
// d_dst + base on; *end = offset behind the last record (host value, stream synchronised).
int32_t encode_records(lz4mi_ctx *c, const uint8_t *d_src, uint64_t n, const lz4mi_frame_options *o,
                       const uint8_t *d_dict, uint32_t dict_len, uint8_t *d_dst, uint64_t base, uint64_t *end);

struct Blk {
  uint64_t payload_off;  // relative to the first byte handed to the device
  uint32_t size;
  uint32_t raw;
  uint32_t has_ck;
  uint32_t ck;
};

struct Header {
  int indep, block_ck, has_size, content_ck, has_dict;
  size_t bmax;
  uint64_t content_size;
  uint32_t dict_id;
  size_t len;  // bytes after the magic, including HC
};

int32_t parse_header(const uint8_t *p, size_t rem, Header *h);
// Most bytes block k can add to the output (slot = the frame's blockMaxSize).
uint64_t planned_output(const Blk &k, size_t slot);

// Optional extras of decode_blocks for callers that deliver output block by block.
struct DecodeExtras {
  // in: when set, a failing block does not fail the call: the blocks in front of it are
  // delivered, *good_blocks says how many, *deferred holds the failing block's status.
  bool partial = false;
  uint32_t good_blocks = 0;
  int32_t deferred = 0;
  std::vector<uint32_t> sizes;  // out: decoded size of each delivered block
  const uint8_t *d_out = nullptr;  // out: device copy of the delivered bytes (valid until the next call)
};

int32_t decode_blocks(lz4mi_ctx *c, const uint8_t *region, size_t region_len, const std::vector<Blk> &blocks,
                      size_t slot, bool indep, const uint8_t *dict, size_t dict_len, bool have_dict,
                      int32_t walk_err, int64_t max_out, bool want_content_hash, bool legacy,
                      bool legacy_last_has_next, uint8_t *dst, size_t cap, size_t *out_len,
                      uint32_t *content_hash, DecodeExtras *extras = nullptr);

// Xxh32.update with the bytes available both on the host (carry bookkeeping) and, optionally,
// already on the device (d_data != nullptr: no upload).
int32_t xxh_state_update(lz4mi_ctx *c, lz4mi_xxh32_state *s, const uint8_t *data, const uint8_t *d_data,
                         size_t len);
// Xxh32.update / xxh32 entirely on the calling thread: content checksums of bytes that lie in host memory
// (lz4_frame_encoder.dart:148,163,266; lz4_frame_decoder.dart:306-315), run on a host thread beside the GPU work.
void xxh_state_update_host(lz4mi_xxh32_state *s, const uint8_t *data, size_t len);
uint32_t xxh32_host(const uint8_t *data, size_t len);

}  // namespace lz4mi
