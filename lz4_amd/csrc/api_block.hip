// liblz4mi: context, block-level entry points (single block on host buffers and
// device-resident batches) and xxHash32.  See include/lz4mi.h for the contract.
#include "api_common.h"
#include "frame_host.h"

#include <mutex>
#include <thread>

using namespace lz4mi;

// ------------------------------------------------------------------ status

extern "C" const char *lz4mi_status_type(int32_t s) {
  if (s == LZ4MI_OK) return "OK";
  if (s >= 1 && s < 10) return "Lz4FormatException";
  if (s >= 10 && s < 30) return "Lz4CorruptDataException";
  if (s == LZ4MI_LIMIT) return "Lz4OutputLimitException";
  if (s >= 40 && s < 50) return "Lz4UnsupportedFeatureException";
  if (s == LZ4MI_DICT_NOT_FOUND) return "Lz4Exception";
  if (s == LZ4MI_RANGE) return "RangeError";
  return "Lz4BackendError";
}

extern "C" const char *lz4mi_status_message(int32_t s) {
  switch (s) {
    case LZ4MI_OK: return "";
    case LZ4MI_FORMAT_EOF: return "Unexpected end of input";
    case LZ4MI_FORMAT_MAGIC: return "Invalid LZ4 frame magic number";
    case LZ4MI_FORMAT_FLG_RESERVED: return "Reserved FLG bit is set";
    case LZ4MI_FORMAT_BD_RESERVED: return "Reserved BD bits are set";
    case LZ4MI_FORMAT_BLOCK_MAX: return "Invalid block maximum size";
    case LZ4MI_FORMAT_INT_OVERFLOW: return "Integer overflow";
    case LZ4MI_FORMAT_CONTENT_SIZE_ARG: return "contentSize does not match src length";
    case LZ4MI_FORMAT_CONTENT_SIZE_STREAM: return "contentSize does not match stream length";
    case LZ4MI_FORMAT_SIZE_HEADER: return "Input too short for size header";
    case LZ4MI_CORRUPT_LITERAL: return "Literal length exceeds output size";
    case LZ4MI_CORRUPT_MATCH: return "Match length exceeds output size";
    case LZ4MI_CORRUPT_TRAILING: return "Trailing bytes after end of block";
    case LZ4MI_CORRUPT_DISTANCE: return "Invalid match distance";
    case LZ4MI_CORRUPT_HEADER_CK: return "Header checksum mismatch";
    case LZ4MI_CORRUPT_BLOCK_SIZE: return "Block size exceeds maximum";
    case LZ4MI_CORRUPT_BLOCK_CK: return "Block checksum mismatch";
    case LZ4MI_CORRUPT_CONTENT_CK: return "Content checksum mismatch";
    case LZ4MI_CORRUPT_CONTENT_SIZE: return "Content size mismatch";
    case LZ4MI_CORRUPT_LEGACY_SIZE: return "Invalid legacy block size";
    case LZ4MI_CORRUPT_LEGACY_NOT_FULL: return "Legacy block is not full";
    case LZ4MI_LIMIT: return "Output limit exceeded";
    case LZ4MI_UNSUPPORTED_VERSION: return "Unsupported LZ4 frame version";
    case LZ4MI_UNSUPPORTED_DICT: return "Dictionary ID present but no dictionary resolver provided";
    case LZ4MI_UNSUPPORTED_CONTENT_SIZE: return "Content size exceeds supported integer range";
    case LZ4MI_DICT_NOT_FOUND: return "Dictionary not found for ID: ";
    case LZ4MI_RANGE: return "Invalid argument";
    case LZ4MI_BACKEND: return "HIP backend failure";
    case LZ4MI_DST_TOO_SMALL: return "destination buffer too small";
    case LZ4MI_NEED_DICT: return "dictionary required";
    case LZ4MI_STREAM_NEED_INPUT: return "more input required";
    case LZ4MI_UNSUPPORTED_HERE: return "not supported by the device path";
    default: return "unknown status";
  }
}

extern "C" int32_t lz4mi_version(void) { return 10000; }
extern "C" size_t lz4mi_compress_bound(size_t n) { return bound(n); }

// ----------------------------------------------------------------- context

enum { kCounterDecode = 0 /* three words */, kCounterEncode = 4, kCounterHc = 5, kCounterOrder = 8 /* two words */, kNumCounters = 16 };

// reason of the last failed lz4mi_ctx_create on the calling thread
static thread_local std::string g_create_error;

// Device memory of lz4mi_decompress_batch_static: the decoder's queue words and its block order list, which
// lz4mi_decompress_batch keeps in buffers the context allocates.  Every GPU the library runs on holds its own
// copy of this array (8 MiB + 256 B) until the process ends.  `scratch` is a host-side stand-in for the symbol:
// host code uses the per-device address that lz4mi_ctx_create looks up, never the name.  (No `used`
// attribute: the compiler exports the symbol to the runtime because host code names it; with `used`
// it stays local to the code object and the runtime cannot find it.)
constexpr uint32_t kStaticMaxBlocks = LZ4MI_STATIC_MAX_BLOCKS;
constexpr uint32_t kStaticHead = 64;  // words in front of the order list (queue heads)
static __device__ __align__(256) uint32_t scratch[kStaticHead + 2 * kStaticMaxBlocks];  // two order words per block

// Every context on a GPU shares that GPU's scratch array, so launches that use it are put in order here: the
// last one's stream and an event recorded behind it are kept per device, and a call that arrives on another
// stream first makes its stream wait for that event (no allocation: the event is created with the device's
// first context).  Calls from several threads take the device's lock around wait + launch + record.
namespace {
struct StaticGuard {
  std::mutex mu;
  hipEvent_t done = nullptr;
  hipStream_t last = nullptr;
  bool used = false;
};
constexpr int kMaxDevices = 64;
StaticGuard g_static_guard[kMaxDevices];
}  // namespace

static int32_t create_fail(lz4mi_ctx *c, hipError_t e, const char *what) {
  g_create_error = std::string(what) + ": " + hipGetErrorString(e);
  if (c) lz4mi_ctx_destroy(c);
  return LZ4MI_BACKEND;
}

extern "C" int32_t lz4mi_ctx_create(int32_t device, void *stream, lz4mi_ctx **out) {
  if (!out) return LZ4MI_RANGE;
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess) return create_fail(nullptr, e, "hipGetDeviceCount");
  if (device < 0 || device >= count) {
    g_create_error = "device ordinal out of range (" + std::to_string(count) + " devices visible)";
    return LZ4MI_BACKEND;
  }
  lz4mi_ctx *c = new lz4mi_ctx();
  c->device = device;
  if ((e = hipSetDevice(device)) != hipSuccess) return create_fail(c, e, "hipSetDevice");
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
    c->cu_count = cus;
  if (stream) {
    c->stream = (hipStream_t)stream;
  } else {
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess)
      return create_fail(c, e, "hipStreamCreateWithFlags");
    c->own_stream = true;
  }
  if ((e = hipMalloc((void **)&c->d_counters, kNumCounters * sizeof(uint32_t))) != hipSuccess)
    return create_fail(c, e, "hipMalloc");
  // this device's copy of the static scratch array (the context's device is current here)
  void *sp = nullptr;
  if ((e = hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch))) != hipSuccess || !sp)
    return create_fail(c, e, "hipGetSymbolAddress(scratch)");
  c->static_scratch = (uint32_t *)sp;
  if (device < kMaxDevices) {
    StaticGuard &g = g_static_guard[device];
    std::lock_guard<std::mutex> lock(g.mu);
    if (!g.done && (e = hipEventCreateWithFlags(&g.done, hipEventDisableTiming)) != hipSuccess)
      return create_fail(c, e, "hipEventCreateWithFlags");
  }
  uint32_t wpc = 16;  // resident encoder wavefronts per CU (one 256 KiB table each)
  if (const char *v = getenv("LZ4MI_ENC_WAVES_PER_CU")) {
    int n = atoi(v);
    if (n >= 1 && n <= 32) wpc = (uint32_t)n;
  }
  c->max_slots = (uint32_t)c->cu_count * wpc;
  *out = c;
  return LZ4MI_OK;
}

static std::vector<DevBuf *> ctx_buffers(lz4mi_ctx *c);

extern "C" void lz4mi_ctx_destroy(lz4mi_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (DevBuf *b : ctx_buffers(c)) b->release();
  if (c->d_counters) (void)hipFree(c->d_counters);
  if (c->h_pin) (void)hipHostFree(c->h_pin);
  if (c->hash_pin) (void)hipHostFree(c->hash_pin);
  for (hipStream_t &cs : c->copy_streams)
    if (cs) (void)hipStreamDestroy(cs);
  if (c->side_stream) {
    (void)hipStreamSynchronize(c->side_stream);
    (void)hipStreamDestroy(c->side_stream);
  }
  if (c->side_fork) (void)hipEventDestroy(c->side_fork);
  if (c->side_join) (void)hipEventDestroy(c->side_join);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

static std::vector<DevBuf *> ctx_buffers(lz4mi_ctx *c) {
  return {&c->tables, &c->slot_gen, &c->dict_table, &c->dict_copy, &c->in, &c->out, &c->aux,
          &c->dictbuf, &c->comp, &c->comp_len, &c->status, &c->stored_word, &c->payload_addr,
          &c->payload_len, &c->record_size, &c->rec_off, &c->hashes, &c->words, &c->total,
          &c->hc_chain, &c->hc_floor, &c->order, &c->link};
}

extern "C" uint64_t lz4mi_ctx_device_bytes(lz4mi_ctx *c) {
  if (!c) return 0;
  uint64_t n = c->d_counters ? kNumCounters * sizeof(uint32_t) : 0;
  for (DevBuf *b : ctx_buffers(c)) n += b->cap;
  return n;
}

extern "C" int32_t lz4mi_ctx_synchronize(lz4mi_ctx *c) {
  LZ4MI_HIP(c, hipStreamSynchronize(c->stream));
  return LZ4MI_OK;
}

extern "C" const char *lz4mi_last_backend_error(lz4mi_ctx *c) { return c ? c->err.c_str() : g_create_error.c_str(); }

namespace lz4mi {

int32_t side_stream_ready(lz4mi_ctx *c) {
  if (!c->side_stream) LZ4MI_HIP(c, hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
  if (!c->side_fork) LZ4MI_HIP(c, hipEventCreateWithFlags(&c->side_fork, hipEventDisableTiming));
  if (!c->side_join) LZ4MI_HIP(c, hipEventCreateWithFlags(&c->side_join, hipEventDisableTiming));
  return LZ4MI_OK;
}

int32_t pin_reserve(lz4mi_ctx *c, size_t n) {
  if (n <= c->h_pin_cap) return LZ4MI_OK;
  if (c->h_pin) (void)hipHostFree(c->h_pin);
  c->h_pin = nullptr;
  c->h_pin_cap = 0;
  size_t want = round_up(n + n / 4, 4096);
  LZ4MI_HIP(c, hipHostMalloc(&c->h_pin, want, hipHostMallocDefault));
  c->h_pin_cap = want;
  return LZ4MI_OK;
}

static int32_t ensure_slots(lz4mi_ctx *c, uint32_t want) {
  if (want > c->max_slots) want = c->max_slots;
  want = (uint32_t)round_up(want, 4);
  if (want <= c->nslots) return LZ4MI_OK;
  // growing restarts every slot at generation 0 over a zeroed table
  LZ4MI_HIP(c, hipStreamSynchronize(c->stream));
  LZ4MI_HIP(c, c->tables.ensure((size_t)want * kHashEntries * sizeof(uint32_t)));
  LZ4MI_HIP(c, c->slot_gen.ensure((size_t)want * sizeof(uint32_t)));
  LZ4MI_HIP(c, hipMemsetAsync(c->tables.p, 0, (size_t)want * kHashEntries * sizeof(uint32_t), c->stream));
  LZ4MI_HIP(c, hipMemsetAsync(c->slot_gen.p, 0, (size_t)want * sizeof(uint32_t), c->stream));
  c->nslots = want;
  return LZ4MI_OK;
}

int32_t run_compress_fast(lz4mi_ctx *c, const BatchArgs &a_in, int64_t acceleration) {
  if (acceleration < 1) return LZ4MI_RANGE;  // lz4_block_encoder.dart:41-43
  if (a_in.nblocks == 0) return LZ4MI_OK;
  BatchArgs a = a_in;
  if (a.dict == nullptr) a.dict_len = 0;
  if (a.dict_len > 65535) {  // encoder keeps the last 65535 dictionary bytes (:50-61)
    a.dict += a.dict_len - 65535;
    a.dict_len = 65535;
  }
  int32_t st = ensure_slots(c, a.nblocks);
  if (st) return st;
  EncodeScratch sc;
  sc.tables = c->tables.as<uint32_t>();
  sc.slot_gen = c->slot_gen.as<uint32_t>();
  sc.nslots = c->nslots;
  sc.dict_table = nullptr;
  if (a.dict_len && !a.linked) {
    LZ4MI_HIP(c, c->dict_table.ensure(kHashEntries * sizeof(uint32_t)));
    sc.dict_table = c->dict_table.as<uint32_t>();
    launch_dict_prime(a.dict, a.dict_len, sc.dict_table, c->stream);
  }
  uint32_t sh = acceleration > 31 ? 31u : (uint32_t)acceleration;  // offsets stay below 2^31
  uint32_t *counter = c->d_counters + kCounterEncode;
  LZ4MI_HIP(c, hipMemsetAsync(counter, 0, sizeof(uint32_t), c->stream));
  uint32_t grid = (a.nblocks + 3) / 4;
  if (grid > c->nslots / 4) grid = c->nslots / 4;
  launch_encode_fast(a, sh, sc, counter, grid, c->stream);
  LZ4MI_HIP(c, hipGetLastError());
  return LZ4MI_OK;
}

// counters: kNumCounters queue words; order: two words per block.  Both are device memory that nothing
// else on the stream uses while the launch runs.
static int32_t run_decompress_with(lz4mi_ctx *c, const BatchArgs &a_in, bool strict, uint32_t *counters, uint32_t *order) {
  if (a_in.nblocks == 0) return LZ4MI_OK;
  BatchArgs a = a_in;
  if (a.dict == nullptr) a.dict_len = 0;
  if (a.dict_len > 65536) {  // decoder keeps the last 65536 bytes (lz4_frame_decoder.dart:278-282)
    a.dict += a.dict_len - 65536;
    a.dict_len = 65536;
  }
  uint32_t *counter = counters + kCounterDecode;
  uint32_t *order_cnt = counters + kCounterOrder;
  // the launch zeroes what it uses: the decode word alone, or in one fill the decode word and the order words
  // behind it (the encoders' words in between are zeroed by their own launches before use; everything runs on
  // the context's one stream)
  static_assert(kCounterDecode == 0 && kCounterOrder + 2 <= kNumCounters, "counter layout");
  LZ4MI_HIP(c, launch_decode(a, strict, counter, order_cnt, order, (uint32_t)c->cu_count, c->decode_form, c->stream));
  return LZ4MI_OK;
}

int32_t run_decompress(lz4mi_ctx *c, const BatchArgs &a, bool strict) {
  if (a.nblocks == 0) return LZ4MI_OK;
  LZ4MI_HIP(c, c->order.ensure((size_t)a.nblocks * 2 * sizeof(uint32_t)));
  return run_decompress_with(c, a, strict, c->d_counters, c->order.as<uint32_t>());
}

// Same launch, with the queue words and the order list in the static scratch array: no allocation.
int32_t run_decompress_static(lz4mi_ctx *c, const BatchArgs &a, bool strict) {
  static_assert(kNumCounters <= kStaticHead, "scratch layout");
  if (a.nblocks > kStaticMaxBlocks) {
    c->err = "lz4mi_decompress_batch_static: " + std::to_string(a.nblocks) + " blocks, the static scratch holds " +
             std::to_string(kStaticMaxBlocks);
    return LZ4MI_UNSUPPORTED_HERE;
  }
  if (a.nblocks == 0) return LZ4MI_OK;
  if (c->device >= kMaxDevices) {
    c->err = "lz4mi_decompress_batch_static: device ordinal beyond the per-device guard table";
    return LZ4MI_UNSUPPORTED_HERE;
  }
  StaticGuard &g = g_static_guard[c->device];
  std::lock_guard<std::mutex> lock(g.mu);
  // the scratch array's last user ran on another stream: this launch waits for it instead of racing with it
  if (g.used && g.last != c->stream) LZ4MI_HIP(c, hipStreamWaitEvent(c->stream, g.done, 0));
  const int32_t st = run_decompress_with(c, a, strict, c->static_scratch, c->static_scratch + kStaticHead);
  // Whatever was enqueued (also by a launch sequence that failed half-way) is what the next user has to wait for.
  const hipError_t e = hipEventRecord(g.done, c->stream);
  g.last = c->stream;
  g.used = true;
  if (st) return st;
  LZ4MI_HIP(c, e);
  return LZ4MI_OK;
}

static void to_args(const lz4mi_batch *b, BatchArgs &a) {
  a.src = b->d_src;
  a.src_off = b->d_src_off;
  a.src_len = b->d_src_len;
  a.src_stride = b->src_stride;
  a.src_len_uniform = b->src_len_uniform;
  a.src_total = b->src_total;
  a.dst = b->d_dst;
  a.dst_off = b->d_dst_off;
  a.dst_stride = b->dst_stride;
  a.dst_cap = b->d_dst_cap;
  a.dst_cap_uniform = b->dst_cap_uniform;
  a.dict = b->d_dict;
  a.dict_len = b->d_dict ? b->dict_len : 0;
  a.nblocks = b->nblocks;
  a.out_len = b->d_out_len;
  a.status = b->d_status;
  a.linked = 0;
  a.conformant = 0;
}

}  // namespace lz4mi

// ------------------------------------------------------------ batch entry points

extern "C" int32_t lz4mi_compress_fast_batch(lz4mi_ctx *c, const lz4mi_batch *b, int64_t acceleration) {
  if (!c || !b) return LZ4MI_RANGE;
  LZ4MI_HIP(c, ::lz4mi::enter_device(c));
  BatchArgs a;
  to_args(b, a);
  return run_compress_fast(c, a, acceleration);
}

extern "C" int32_t lz4mi_compress_hc_batch(lz4mi_ctx *c, const lz4mi_batch *b, int64_t depth) {
  if (!c || !b) return LZ4MI_RANGE;
  LZ4MI_HIP(c, ::lz4mi::enter_device(c));
  BatchArgs a;
  to_args(b, a);
  return run_compress_hc(c, a, depth);
}

extern "C" int32_t lz4mi_decompress_batch(lz4mi_ctx *c, const lz4mi_batch *b, int32_t strict) {
  if (!c || !b) return LZ4MI_RANGE;
  LZ4MI_HIP(c, ::lz4mi::enter_device(c));
  BatchArgs a;
  to_args(b, a);
  return run_decompress(c, a, strict != 0);
}

extern "C" int32_t lz4mi_set_decode_form(int32_t form) {
  if (form < 0 || form > 2) return -1;
  return set_decode_form(form);
}

extern "C" int32_t lz4mi_ctx_set_decode_form(lz4mi_ctx *c, int32_t form) {
  if (!c || form < -1 || form > 2) return -2;
  const int32_t prev = c->decode_form;
  c->decode_form = form;
  return prev;
}

extern "C" int32_t lz4mi_decompress_batch_static(lz4mi_ctx *c, const lz4mi_batch *b, int32_t strict) {
  if (!c || !b) return LZ4MI_RANGE;
  LZ4MI_HIP(c, ::lz4mi::enter_device(c));
  BatchArgs a;
  to_args(b, a);
  return run_decompress_static(c, a, strict != 0);
}

extern "C" int32_t lz4mi_xxh32_batch(lz4mi_ctx *c, const lz4mi_batch *b, uint32_t *d_hash) {
  if (!c || !b || !d_hash) return LZ4MI_RANGE;
  LZ4MI_HIP(c, ::lz4mi::enter_device(c));
  BatchArgs a;
  to_args(b, a);
  launch_xxh32_batch(a, d_hash, c->stream);
  LZ4MI_HIP(c, hipGetLastError());
  return LZ4MI_OK;
}

namespace lz4mi {

int32_t host_copy(lz4mi_ctx *c, void *dst, const void *src, size_t n, hipMemcpyKind kind) {
  if (!n) return LZ4MI_OK;
  if (kind == hipMemcpyDeviceToHost) LZ4MI_HIP(c, hipStreamSynchronize(c->stream));
  constexpr size_t kParallelMin = 8u << 20;
  constexpr unsigned kThreads = 4;
  if (n < kParallelMin) {  // in stream order; a host-to-device copy is not waited for (the runtime has read src)
    LZ4MI_HIP(c, hipMemcpyAsync(dst, src, n, kind, c->stream));
    if (kind == hipMemcpyDeviceToHost) LZ4MI_HIP(c, hipStreamSynchronize(c->stream));
    return LZ4MI_OK;
  }
  for (hipStream_t &cs : c->copy_streams)
    if (!cs) LZ4MI_HIP(c, hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
  // the slices go over streams of their own: nothing queued on the context's stream may still touch either side
  if (kind != hipMemcpyDeviceToHost) LZ4MI_HIP(c, hipStreamSynchronize(c->stream));
  const size_t slice = round_up((n + kThreads - 1) / kThreads, 4096);
  hipError_t err[kThreads];
  auto part = [&](unsigned k) {
    const size_t lo = (size_t)k * slice;
    if (lo >= n) { err[k] = hipSuccess; return; }
    const size_t m = n - lo < slice ? n - lo : slice;
    hipError_t e = hipSetDevice(c->device);
    if (e == hipSuccess) e = hipMemcpyAsync((uint8_t *)dst + lo, (const uint8_t *)src + lo, m, kind, c->copy_streams[k]);
    if (e == hipSuccess) e = hipStreamSynchronize(c->copy_streams[k]);
    err[k] = e;
  };
  std::thread helpers[kThreads - 1];
  for (unsigned k = 1; k < kThreads; k++) {
    try {
      helpers[k - 1] = std::thread(part, k);
    } catch (...) {  // no thread to be had: this slice goes with the calling thread's
      part(k);
    }
  }
  part(0);
  for (std::thread &t : helpers)
    if (t.joinable()) t.join();
  for (unsigned k = 0; k < kThreads; k++) LZ4MI_HIP(c, err[k]);
  return LZ4MI_OK;
}

bool hash_on_host(uint64_t n) {
  const char *v = getenv("LZ4MI_HASH_ENGINE");
  if (v && v[0] == 'g') return false;
  if (v && v[0] == 'h') return n != 0;
  return n >= (4u << 20);
}

int32_t hash_host_ready(lz4mi_ctx *c) {
  int32_t st = side_stream_ready(c);
  if (st) return st;
  if (!c->hash_pin) LZ4MI_HIP(c, hipHostMalloc(&c->hash_pin, 2 * kHashPiece, hipHostMallocDefault));
  return LZ4MI_OK;
}

hipError_t xxh32_device_on_host(lz4mi_ctx *c, const uint8_t *d, uint64_t n, uint32_t seed, uint32_t *out) {
  hipError_t e = hipSetDevice(c->device);  // this may be a thread of its own
  if (e != hipSuccess) return e;
  uint8_t *pin[2] = {(uint8_t *)c->hash_pin, (uint8_t *)c->hash_pin + kHashPiece};
  lz4mi_xxh32_state st;
  lz4mi_xxh32_state_init(&st, seed);
  uint64_t off = 0;
  uint32_t k = 0;
  if (n) e = hipMemcpyAsync(pin[0], d, n < kHashPiece ? (size_t)n : kHashPiece, hipMemcpyDeviceToHost, c->side_stream);
  while (off < n && e == hipSuccess) {
    const size_t m = n - off < kHashPiece ? (size_t)(n - off) : kHashPiece;
    e = hipStreamSynchronize(c->side_stream);  // piece k is here
    if (e != hipSuccess) break;
    const uint64_t next = off + m;
    if (next < n)
      e = hipMemcpyAsync(pin[(k + 1) & 1], d + next, n - next < kHashPiece ? (size_t)(n - next) : kHashPiece,
                         hipMemcpyDeviceToHost, c->side_stream);
    xxh_state_update_host(&st, pin[k & 1], m);
    off = next;
    k++;
  }
  if (e != hipSuccess) {
    (void)hipStreamSynchronize(c->side_stream);
    return e;
  }
  *out = lz4mi_xxh32_state_digest(&st);
  return hipSuccess;
}

}  // namespace lz4mi

extern "C" int32_t lz4mi_xxh32_device(lz4mi_ctx *c, const uint8_t *d_data, uint64_t len, uint32_t seed,
                                      uint32_t *d_out) {
  if (!c || !d_out) return LZ4MI_RANGE;
  LZ4MI_HIP(c, ::lz4mi::enter_device(c));
  if (::lz4mi::hash_on_host(len)) {
    // the calling thread hashes; the word reaches d_out in stream order like the kernel's would
    int32_t st = ::lz4mi::hash_host_ready(c);
    if (st) return st;
    st = pin_reserve(c, 256);
    if (st) return st;
    LZ4MI_HIP(c, hipEventRecord(c->side_fork, c->stream));
    LZ4MI_HIP(c, hipStreamWaitEvent(c->side_stream, c->side_fork, 0));
    uint32_t *h = (uint32_t *)((uint8_t *)c->h_pin + 128);
    LZ4MI_HIP(c, ::lz4mi::xxh32_device_on_host(c, d_data, len, seed, h));
    LZ4MI_HIP(c, hipMemcpyAsync(d_out, h, 4, hipMemcpyHostToDevice, c->stream));
    LZ4MI_HIP(c, hipStreamSynchronize(c->stream));  // the pinned word is free again
    return LZ4MI_OK;
  }
  launch_xxh32_serial(d_data, len, seed, d_out, c->stream);
  LZ4MI_HIP(c, hipGetLastError());
  return LZ4MI_OK;
}

// ------------------------------------------------- single block, host buffers

namespace {

struct OneResult {
  uint32_t out_len;
  int32_t status;
};

// Runs one block through a batch kernel: src (+dict) up, result down.
int32_t one_block(lz4mi_ctx *c, const uint8_t *src, size_t n, const uint8_t *dict, size_t dict_len,
                  size_t dev_out_bytes, uint32_t cap_field, int kind, int64_t param, OneResult *res) {
  LZ4MI_HIP(c, ::lz4mi::enter_device(c));
  LZ4MI_HIP(c, c->in.ensure(n + 16));
  LZ4MI_HIP(c, c->out.ensure(dev_out_bytes + 16));
  LZ4MI_HIP(c, c->comp_len.ensure(16));
  LZ4MI_HIP(c, c->status.ensure(16));
  int32_t st = pin_reserve(c, 64);
  if (st) return st;
  st = host_copy(c, c->in.p, src, n, hipMemcpyHostToDevice);
  if (st) return st;
  if (dict_len) {
    LZ4MI_HIP(c, c->dictbuf.ensure(dict_len));
    LZ4MI_HIP(c, hipMemcpyAsync(c->dictbuf.p, dict, dict_len, hipMemcpyHostToDevice, c->stream));
  }
  BatchArgs a;
  memset(&a, 0, sizeof a);
  a.src = c->in.as<uint8_t>();
  a.src_len_uniform = (uint32_t)n;
  a.src_total = n;
  a.dst = c->out.as<uint8_t>();
  a.dst_cap_uniform = cap_field;
  a.dict = dict_len ? c->dictbuf.as<uint8_t>() : nullptr;
  a.dict_len = (uint32_t)dict_len;
  a.nblocks = 1;
  a.out_len = c->comp_len.as<uint32_t>();
  a.status = c->status.as<int32_t>();
  if (kind == 0) st = run_compress_fast(c, a, param);
  else if (kind == 1) st = run_compress_hc(c, a, param);
  else st = run_decompress(c, a, kind == 2);
  if (st) return st;
  uint32_t *h = (uint32_t *)c->h_pin;
  LZ4MI_HIP(c, hipMemcpyAsync(h, a.out_len, 4, hipMemcpyDeviceToHost, c->stream));
  LZ4MI_HIP(c, hipMemcpyAsync(h + 1, a.status, 4, hipMemcpyDeviceToHost, c->stream));
  LZ4MI_HIP(c, hipStreamSynchronize(c->stream));
  res->out_len = h[0];
  res->status = (int32_t)h[1];
  return LZ4MI_OK;
}

int32_t compress_one(lz4mi_ctx *c, int kind, const uint8_t *src, size_t n, const uint8_t *dict, size_t dict_len,
                     int64_t param, uint8_t *dst, size_t cap, size_t *out_len) {
  if (!c) return LZ4MI_RANGE;
  if (out_len) *out_len = 0;
  if (param < 1) return LZ4MI_RANGE;
  if (n == 0) return LZ4MI_OK;  // lz4_block_encoder.dart:45-48
  if (n > 0x7FFFFFF0u - 65536u) return LZ4MI_UNSUPPORTED_HERE;  // positions are 31-bit on the device
  if (!dict) dict_len = 0;
  OneResult r;
  size_t b = bound(n);
  int32_t st = one_block(c, src, n, dict, dict_len, b, (uint32_t)b, kind, param, &r);
  if (st) return st;
  if (r.status) return r.status;
  if (r.out_len > cap) return LZ4MI_DST_TOO_SMALL;
  LZ4MI_HIP(c, hipMemcpy(dst, c->out.p, r.out_len, hipMemcpyDeviceToHost));
  if (out_len) *out_len = r.out_len;
  return LZ4MI_OK;
}

}  // namespace

extern "C" int32_t lz4mi_block_compress_fast(lz4mi_ctx *c, const uint8_t *src, size_t n, const uint8_t *dict,
                                             size_t dict_len, int64_t acceleration, uint8_t *dst, size_t cap,
                                             size_t *out_len) {
  return compress_one(c, 0, src, n, dict, dict_len, acceleration, dst, cap, out_len);
}

extern "C" int32_t lz4mi_block_compress_hc(lz4mi_ctx *c, const uint8_t *src, size_t n, const uint8_t *dict,
                                           size_t dict_len, int64_t depth, uint8_t *dst, size_t cap,
                                           size_t *out_len) {
  return compress_one(c, 1, src, n, dict, dict_len, depth, dst, cap, out_len);
}

extern "C" int32_t lz4mi_block_decompress(lz4mi_ctx *c, const uint8_t *src, size_t n, uint8_t *dst,
                                          int64_t decompressed_size) {
  if (!c) return LZ4MI_RANGE;
  if (decompressed_size < 0) return LZ4MI_RANGE;  // lz4_block_decoder.dart:11-13
  if (decompressed_size > 0x7fffffff || n > 0x7fffffff) return LZ4MI_UNSUPPORTED_HERE;
  OneResult r;
  int32_t st = one_block(c, src, n, nullptr, 0, (size_t)decompressed_size, (uint32_t)decompressed_size, 2, 0, &r);
  if (st) return st;
  if (r.status) return r.status;
  if (r.out_len) LZ4MI_HIP(c, hipMemcpy(dst, c->out.p, r.out_len, hipMemcpyDeviceToHost));
  return LZ4MI_OK;
}

extern "C" int32_t lz4mi_block_decompress_into(lz4mi_ctx *c, const uint8_t *src, size_t n, uint8_t *dst,
                                               size_t prefix_len, size_t max_len, size_t *out_len) {
  if (!c) return LZ4MI_RANGE;
  if (out_len) *out_len = prefix_len;
  if (max_len < prefix_len) return LZ4MI_LIMIT;
  if (max_len - prefix_len > 0x7fffffff || n > 0x7fffffff || prefix_len > 0x7fffffff) return LZ4MI_UNSUPPORTED_HERE;
  OneResult r;
  size_t room = max_len - prefix_len;
  // a match reaches back 65535 bytes at most: only the tail of a long prefix goes to the device
  const size_t keep = prefix_len < 65536 ? prefix_len : 65536;
  int32_t st = one_block(c, src, n, dst + (prefix_len - keep), keep, room, (uint32_t)room, 3, 0, &r);
  if (st) return st;
  if (r.out_len) LZ4MI_HIP(c, hipMemcpy(dst + prefix_len, c->out.p, r.out_len, hipMemcpyDeviceToHost));
  if (out_len) *out_len = prefix_len + r.out_len;
  return r.status;
}

extern "C" int32_t lz4mi_xxh32(lz4mi_ctx *c, const uint8_t *data, size_t len, uint32_t seed, uint32_t *out) {
  if (!c || !out) return LZ4MI_RANGE;
  LZ4MI_HIP(c, ::lz4mi::enter_device(c));
  if (::lz4mi::hash_on_host(len)) {  // one serial stream over bytes that are in host memory already (api_common.h)
    lz4mi_xxh32_state hs;
    lz4mi_xxh32_state_init(&hs, seed);
    ::lz4mi::xxh_state_update_host(&hs, data, len);
    *out = lz4mi_xxh32_state_digest(&hs);
    return LZ4MI_OK;
  }
  LZ4MI_HIP(c, c->in.ensure(len + 16));
  LZ4MI_HIP(c, c->words.ensure(64));
  int32_t st = pin_reserve(c, 64);
  if (st) return st;
  if (len) LZ4MI_HIP(c, hipMemcpyAsync(c->in.p, data, len, hipMemcpyHostToDevice, c->stream));
  launch_xxh32_serial(c->in.as<uint8_t>(), len, seed, c->words.as<uint32_t>(), c->stream);
  LZ4MI_HIP(c, hipGetLastError());
  LZ4MI_HIP(c, hipMemcpyAsync(c->h_pin, c->words.p, 4, hipMemcpyDeviceToHost, c->stream));
  LZ4MI_HIP(c, hipStreamSynchronize(c->stream));
  *out = *(uint32_t *)c->h_pin;
  return LZ4MI_OK;
}
