// Launch interface of the frame assembly kernels (internal to liblz4mi).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lz4mi {

struct FramePlanArgs {
  const uint8_t *src;       // uncompressed input (device), blocks of block_max bytes
  uint64_t src_total;
  uint32_t block_max;
  const uint8_t *comp;      // compressed slots (device), comp_stride apart
  uint64_t comp_stride;
  const uint32_t *comp_len; // per-block compressed length
  uint32_t nblocks;
  int block_checksum;
  uint32_t *stored_word;    // out: size word (bit 31 = stored raw)
  uint64_t *payload_addr;   // out: device address of the stored payload
  uint32_t *payload_len;    // out: stored payload length
  uint32_t *record_size;    // out: 4 + payload (+4)
};

struct RecordReadArgs {
  const uint8_t *records;    // this rank's window of the frame (device)
  uint64_t records_len;
  const uint64_t *rec_off;   // per block: offset of its size word in `records`
  uint32_t nblocks;
  uint32_t block_max;
  int block_checksum;
  uint64_t *payload_addr;    // out: device address of the stored payload
  uint32_t *payload_len;     // out: stored payload length (0 when the record failed a check)
  uint32_t *dec_len;         // out: payload length for the decoder (0 for stored blocks)
  uint32_t *raw_len;         // out: bytes to copy for stored blocks
  uint64_t *slot_off;        // out: b * block_max
  uint32_t *expect;          // out: checksum word behind the payload
  int32_t *pre_status;       // out: failure met before decoding, or OK
};

struct BatchArgs;
void launch_record_read(const RecordReadArgs &r, hipStream_t s);
void launch_record_finish(const RecordReadArgs &r, const uint32_t *hashes, uint32_t *out_len, int32_t *status,
                          hipStream_t s);
void launch_slot_addr(const uint8_t *slots, uint64_t stride, uint32_t nblocks, uint64_t *addr, hipStream_t s);
void launch_block_conformance(const BatchArgs &a, uint32_t *flags, hipStream_t s);
void launch_frame_plan(const FramePlanArgs &p, hipStream_t s);
void launch_scan_sizes(const uint32_t *sizes, uint32_t n, uint64_t base, uint64_t *offsets, uint64_t *total,
                       hipStream_t s);
void launch_frame_words(uint8_t *frame, const uint64_t *rec_off, const uint32_t *stored_word,
                        const uint32_t *payload_len, const uint32_t *hashes, uint32_t nblocks, hipStream_t s);
void launch_seg_copy(const uint64_t *src_addr, uint8_t *dst_base, const uint64_t *dst_off, uint64_t dst_delta,
                     const uint32_t *len, uint32_t nseg, hipStream_t s);
void launch_record_bounds(const uint64_t *rec_off, const uint32_t *record_size, uint32_t nblocks, uint64_t cap,
                          uint32_t *flag, hipStream_t s);
void launch_put_words(uint8_t *dst, const uint32_t *words, uint32_t nwords, bool end_mark, hipStream_t s);

}  // namespace lz4mi
