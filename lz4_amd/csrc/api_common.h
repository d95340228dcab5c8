// Context and host-side helpers of liblz4mi (internal).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <future>
#include <string>
#include <vector>

#include "../../include/lz4mi.h"
#include "frame_kernels.h"
#include "kernels.h"

namespace lz4mi {

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = n + n / 8 + 4096;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <typename T>
  T *as() const { return (T *)p; }
};

}  // namespace lz4mi

struct lz4mi_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int cu_count = 256;
  std::string err;

  uint32_t *d_counters = nullptr;  // work-queue heads, one per launch kind
  uint32_t *static_scratch = nullptr;  // this device's address of the library's static scratch array
  int decode_form = -1;                // lz4mi_ctx_set_decode_form: 0, 1, 2, or negative = the process default

  // fast/HC encoder hash tables: one 65536-entry slot per resident wavefront
  lz4mi::DevBuf tables, slot_gen, dict_table, dict_copy;
  uint32_t nslots = 0;
  uint32_t max_slots = 0;

  // staging for the host-buffer entry points
  lz4mi::DevBuf in, out, aux, dictbuf;
  // per-block bookkeeping
  lz4mi::DevBuf comp, comp_len, status, stored_word, payload_addr, payload_len, record_size, rec_off,
      hashes, words, total;
  // HC encoder: per-slot position index (rank, sorted, sort scratch) and bucket floors
  lz4mi::DevBuf hc_chain, hc_floor;
  // decoder: queue position -> block index of a launch (long blocks first)
  lz4mi::DevBuf order;
  // linked frames: block offsets, the distance array of a chunk, its tile flags
  lz4mi::DevBuf link;

  // state kept between lz4mi_frame_shard_compress and lz4mi_frame_shard_write
  uint32_t shard_blocks = 0;
  int shard_block_checksum = 0;
  // set by lz4mi_frame_shard_compress, cleared by every other entry point (they reuse the scratch)
  bool shard_valid = false;
  const uint8_t *shard_src = nullptr;
  uint64_t shard_n = 0;
  int shard_block_size_id = 0;

  void *h_pin = nullptr;  // pinned host scratch for small read-backs
  size_t h_pin_cap = 0;

  // Second stream for work that depends only on a call's input and runs beside its block launches (the
  // single-stream content checksum of a device-resident source); created on first use.
  hipStream_t side_stream = nullptr;
  hipEvent_t side_fork = nullptr, side_join = nullptr;
  // two pinned pieces through which device memory reaches a host thread that hashes it (hash_on_host)
  void *hash_pin = nullptr;
  // streams of the host threads that share a large copy to or from pageable host memory (host_copy)
  hipStream_t copy_streams[4] = {nullptr, nullptr, nullptr, nullptr};
};

namespace lz4mi {

inline int32_t fail(lz4mi_ctx *c, hipError_t e, const char *what) {
  c->err = std::string(what) + ": " + hipGetErrorString(e);
  return LZ4MI_BACKEND;
}

// First HIP call of every entry point: makes the context's device current and drops an error code
// left behind by the caller's own HIP calls (hipGetLastError after a launch would report it as ours).
inline hipError_t enter_device(lz4mi_ctx *c) {
  c->shard_valid = false;
  (void)hipGetLastError();
  return hipSetDevice(c->device);
}

#define LZ4MI_HIP(c, call)                                   \
  do {                                                       \
    hipError_t e_ = (call);                                  \
    if (e_ != hipSuccess) return ::lz4mi::fail((c), e_, #call); \
  } while (0)

inline size_t block_max_bytes(int id) {  // lib/src/frame/lz4_frame_options.dart:117-128
  switch (id) {
    case 4: return 64u << 10;
    case 5: return 256u << 10;
    case 6: return 1u << 20;
    case 7: return 4u << 20;
    default: return 0;
  }
}

inline size_t bound(size_t n) { return n + n / 255 + 16; }
inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// batch entry points used by both the public batch API and the frame layer
int32_t run_compress_fast(lz4mi_ctx *c, const BatchArgs &a, int64_t acceleration);
int32_t run_compress_hc(lz4mi_ctx *c, const BatchArgs &a, int64_t depth);
int32_t run_decompress(lz4mi_ctx *c, const BatchArgs &a, bool strict);
int32_t pin_reserve(lz4mi_ctx *c, size_t n);
// makes the side stream and its two events
int32_t side_stream_ready(lz4mi_ctx *c);

// f() on a thread of its own, beside the caller's GPU work; on the calling thread when no thread is to be had
// (the future then holds the finished result).  Nothing is thrown across the C ABI.
template <class F>
auto run_beside(F f) -> std::future<decltype(f())> {
  try {
    return std::async(std::launch::async, f);
  } catch (...) {
    std::packaged_task<decltype(f())()> task(f);
    auto fut = task.get_future();
    task();
    return fut;
  }
}

// Bulk copy between the caller's host memory and the device: a device-to-host copy first waits for the context's
// stream and returns when the bytes have arrived; a host-to-device copy returns when src may be reused and is
// ordered in front of whatever the caller queues on the context's stream next.  A copy of pageable memory is staged
// by the calling thread, which is what bounds it (17.6 GB/s device to host, 29 GB/s host to device on the GPU box);
// from 8 MiB on the range is cut into four slices that four host threads copy at once on streams of their own
// (43.5 and 42 GB/s).
int32_t host_copy(lz4mi_ctx *c, void *dst, const void *src, size_t n, hipMemcpyKind kind);

// The content checksum of a frame is ONE xxh32 stream: rounds that each wait for the one before (no way to split
// it over wavefronts).  One GPU lane-chain runs it at under 2 GB/s; a host core several times faster.  So device
// memory of this size or more is hashed by a host thread: pieces come over on the side stream into two pinned
// buffers, a piece is hashed while the next one is on its way.  LZ4MI_HASH_ENGINE=gpu keeps every hash on the
// device, =host sends every one to the host.
bool hash_on_host(uint64_t n);
constexpr size_t kHashPiece = 16u << 20;
// allocates the pinned pieces and the side stream (calling thread); then, possibly on another thread:
int32_t hash_host_ready(lz4mi_ctx *c);
// hashes d[0, n) on the calling thread; the side stream must already wait for whatever produces d
hipError_t xxh32_device_on_host(lz4mi_ctx *c, const uint8_t *d, uint64_t n, uint32_t seed, uint32_t *out);

}  // namespace lz4mi
