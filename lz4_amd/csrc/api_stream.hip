// liblz4mi: streaming xxHash32 state, the host-only frame formats (frame info, skippable and
// legacy frames) and the chunk-at-a-time frame encoder / decoder.  The per-block work is the
// same batched kernels the one-shot frame calls use; this file is the buffering around them.
#include <algorithm>
#include <future>
#include <new>

#include "api_common.h"
#include "frame_host.h"

using namespace lz4mi;

// ------------------------------------------------------- Xxh32 (streaming state)

namespace {

constexpr uint32_t P1 = 0x9E3779B1u, P2 = 0x85EBCA77u, P3 = 0xC2B2AE3Du, P4 = 0x27D4EB2Fu, P5 = 0x165667B1u;
inline uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

}  // namespace

extern "C" void lz4mi_xxh32_state_init(lz4mi_xxh32_state *s, uint32_t seed) {  // xxh32.dart:33-41
  memset(s, 0, sizeof *s);
  s->seed = seed;
  s->v[0] = seed + P1 + P2;
  s->v[1] = seed + P2;
  s->v[2] = seed;
  s->v[3] = seed - P1;
}

namespace lz4mi {

int32_t xxh_state_update(lz4mi_ctx *c, lz4mi_xxh32_state *s, const uint8_t *data, const uint8_t *d_data,
                         size_t len) {  // xxh32.dart:43-125
  if (len == 0) return LZ4MI_OK;
  s->total_len += len;
  size_t p = 0;
  if (s->mem_size + len < 16) {  // :60-64
    memcpy(s->mem + s->mem_size, data, len);
    s->mem_size += (uint32_t)len;
    return LZ4MI_OK;
  }
  if (s->mem_size) {  // :66-81: the stripe completed by the carry
    const size_t fill = 16 - s->mem_size;
    memcpy(s->mem + s->mem_size, data, fill);
    p += fill;
    len -= fill;
    s->mem_size = 0;
    for (int k = 0; k < 4; k++) s->v[k] = rotl(s->v[k] + rd32(s->mem + 4 * k) * P2, 13) * P1;
  }
  const size_t stripes = len / 16;
  if (stripes) {  // :83-118 on the device
    if (!c) return LZ4MI_RANGE;
    LZ4MI_HIP(c, ::lz4mi::enter_device(c));
    const uint8_t *dp = nullptr;
    if (d_data) {
      dp = d_data + p;
    } else {
      LZ4MI_HIP(c, c->in.ensure(stripes * 16 + 16));
      LZ4MI_HIP(c, hipMemcpyAsync(c->in.p, data + p, stripes * 16, hipMemcpyHostToDevice, c->stream));
      dp = c->in.as<uint8_t>();
    }
    LZ4MI_HIP(c, c->words.ensure(64));
    int32_t st = pin_reserve(c, 256);
    if (st) return st;
    uint32_t *h = (uint32_t *)((uint8_t *)c->h_pin + 192);
    uint32_t *d_state = c->words.as<uint32_t>() + 8;
    memcpy(h, s->v, 16);
    LZ4MI_HIP(c, hipMemcpyAsync(d_state, h, 16, hipMemcpyHostToDevice, c->stream));
    launch_xxh32_stripes(dp, stripes, d_state, c->stream);
    LZ4MI_HIP(c, hipGetLastError());
    LZ4MI_HIP(c, hipMemcpyAsync(h, d_state, 16, hipMemcpyDeviceToHost, c->stream));
    LZ4MI_HIP(c, hipStreamSynchronize(c->stream));
    memcpy(s->v, h, 16);
    p += stripes * 16;
    len -= stripes * 16;
  }
  if (len) {  // :120-124
    memcpy(s->mem, data + p, len);
    s->mem_size = (uint32_t)len;
  }
  return LZ4MI_OK;
}

// The same update with the whole-stripe rounds (:83-118) on the calling thread.  For content checksums of
// bytes that are in host memory anyway (host-buffer frame calls, the stream transformers): one serial stream
// of dependent rounds is what a CPU core does several times faster than one GPU wavefront, and on a host
// thread it runs beside the GPU's block work instead of in line with it.
void xxh_state_update_host(lz4mi_xxh32_state *s, const uint8_t *data, size_t len) {
  if (len == 0) return;
  s->total_len += len;
  size_t p = 0;
  if (s->mem_size + len < 16) {
    memcpy(s->mem + s->mem_size, data, len);
    s->mem_size += (uint32_t)len;
    return;
  }
  uint32_t v0 = s->v[0], v1 = s->v[1], v2 = s->v[2], v3 = s->v[3];
  if (s->mem_size) {
    const size_t fill = 16 - s->mem_size;
    memcpy(s->mem + s->mem_size, data, fill);
    p += fill;
    len -= fill;
    s->mem_size = 0;
    v0 = rotl(v0 + rd32(s->mem) * P2, 13) * P1;
    v1 = rotl(v1 + rd32(s->mem + 4) * P2, 13) * P1;
    v2 = rotl(v2 + rd32(s->mem + 8) * P2, 13) * P1;
    v3 = rotl(v3 + rd32(s->mem + 12) * P2, 13) * P1;
  }
  const uint8_t *q = data + p, *const end = q + (len & ~(size_t)15);
  for (; q < end; q += 16) {
    uint32_t w[4];
    memcpy(w, q, 16);  // little-endian host
    v0 = rotl(v0 + w[0] * P2, 13) * P1;
    v1 = rotl(v1 + w[1] * P2, 13) * P1;
    v2 = rotl(v2 + w[2] * P2, 13) * P1;
    v3 = rotl(v3 + w[3] * P2, 13) * P1;
  }
  s->v[0] = v0; s->v[1] = v1; s->v[2] = v2; s->v[3] = v3;
  len &= 15;
  if (len) {
    memcpy(s->mem, q, len);
    s->mem_size = (uint32_t)len;
  }
}

uint32_t xxh32_host(const uint8_t *data, size_t len) {
  lz4mi_xxh32_state s;
  lz4mi_xxh32_state_init(&s, 0);
  xxh_state_update_host(&s, data, len);
  return lz4mi_xxh32_state_digest(&s);
}

}  // namespace lz4mi

extern "C" int32_t lz4mi_xxh32_state_update(lz4mi_ctx *c, lz4mi_xxh32_state *s, const uint8_t *data, size_t len) {
  if (!s || (!data && len)) return LZ4MI_RANGE;
  return xxh_state_update(c, s, data, nullptr, len);
}

extern "C" uint32_t lz4mi_xxh32_state_digest(const lz4mi_xxh32_state *s) {  // xxh32.dart:127-163
  uint32_t h = s->total_len >= 16 ? rotl(s->v[0], 1) + rotl(s->v[1], 7) + rotl(s->v[2], 12) + rotl(s->v[3], 18)
                                  : s->seed + P5;
  h += (uint32_t)s->total_len;
  uint32_t p = 0;
  for (; p + 4 <= s->mem_size; p += 4) h = rotl(h + rd32(s->mem + p) * P3, 17) * P4;
  for (; p < s->mem_size; p++) h = rotl(h + s->mem[p] * P5, 11) * P1;
  h ^= h >> 15;
  h *= P2;
  h ^= h >> 13;
  h *= P3;
  h ^= h >> 16;
  return h;
}

// ------------------------------------------------------------ host-only formats

extern "C" int32_t lz4mi_frame_info_parse(const uint8_t *src, size_t len, lz4mi_frame_info *out) {
  if (!out || (!src && len)) return LZ4MI_RANGE;  // lz4_frame_info.dart:89-199
  memset(out, 0, sizeof *out);
  out->block_independence = 1;  // constructor defaults (:56-61)
  if (len < 4) return LZ4MI_FORMAT_EOF;
  const uint32_t magic = rd32(src);
  out->magic = magic;
  if (is_skippable_magic(magic)) {  // :99-111
    if (len - 4 < 4) return LZ4MI_FORMAT_EOF;
    out->is_skippable = 1;
    out->skippable_size = rd32(src + 4);
    out->header_size = 8;
    return LZ4MI_OK;
  }
  if (magic == kLegacyMagic) {  // :114-121
    out->is_legacy = 1;
    out->header_size = 4;
    return LZ4MI_OK;
  }
  if (magic != kFrameMagic) return LZ4MI_FORMAT_MAGIC;
  Header h;
  int32_t st = parse_header(src + 4, len - 4, &h);  // same checks in the same order as :128-185
  if (st) return st;
  out->block_independence = h.indep;
  out->block_checksum = h.block_ck;
  out->content_checksum = h.content_ck;
  out->has_content_size = h.has_size;
  out->content_size = h.content_size;
  out->has_dict_id = h.has_dict;
  out->dict_id = h.dict_id;
  out->block_max_size = (uint32_t)h.bmax;
  out->header_size = (uint32_t)(4 + h.len);
  return LZ4MI_OK;
}

extern "C" int32_t lz4mi_skippable_frame_encode(const uint8_t *data, size_t n, int32_t index, uint8_t *dst,
                                                size_t cap, size_t *out_len) {  // lz4_frame_encoder.dart:27-46
  if (out_len) *out_len = 0;
  if (index < 0 || index > 15) return LZ4MI_RANGE;
  if ((uint64_t)n > 0xFFFFFFFFull) return LZ4MI_RANGE;
  if ((!data && n) || !dst) return LZ4MI_RANGE;
  if (cap < 8 || cap - 8 < n) return LZ4MI_DST_TOO_SMALL;
  wr32(dst, kSkippableBase + (uint32_t)index);
  wr32(dst + 4, (uint32_t)n);
  if (n) memcpy(dst + 8, data, n);
  if (out_len) *out_len = 8 + n;
  return LZ4MI_OK;
}

extern "C" size_t lz4mi_legacy_frame_bound(size_t n) {
  const size_t nb = (n + kLegacyBlockMax - 1) / kLegacyBlockMax;
  return 4 + nb * (4 + 16 + kLegacyBlockMax / 255 + 1) + n;
}

extern "C" int32_t lz4mi_legacy_frame_encode(lz4mi_ctx *c, const uint8_t *src, size_t n, int64_t acceleration,
                                             uint8_t *dst, size_t cap, size_t *out_len) {
  if (!c || !dst || (!src && n)) return LZ4MI_RANGE;  // lz4_frame_encoder.dart:55-89
  if (out_len) *out_len = 0;
  if (cap < 4) return LZ4MI_DST_TOO_SMALL;
  wr32(dst, kLegacyMagic);
  size_t w = 4;
  if (n == 0) {  // the block encoder (and its acceleration check) is never reached (:62)
    if (out_len) *out_len = w;
    return LZ4MI_OK;
  }
  if (acceleration < 1) return LZ4MI_RANGE;
  if (n / kLegacyBlockMax > 0x7ffffff0u) return LZ4MI_RANGE;
  LZ4MI_HIP(c, ::lz4mi::enter_device(c));
  const uint32_t nb = (uint32_t)((n + kLegacyBlockMax - 1) / kLegacyBlockMax);
  const size_t blk = n < kLegacyBlockMax ? n : kLegacyBlockMax;
  const size_t stride = round_up(bound(blk), 64);
  LZ4MI_HIP(c, c->in.ensure(n + 16));
  LZ4MI_HIP(c, c->comp.ensure((size_t)nb * stride));
  LZ4MI_HIP(c, c->comp_len.ensure((size_t)nb * 4));
  LZ4MI_HIP(c, c->status.ensure((size_t)nb * 4));
  int32_t st = pin_reserve(c, (size_t)nb * 8 + 64);
  if (st) return st;
  st = host_copy(c, c->in.p, src, n, hipMemcpyHostToDevice);
  if (st) return st;
  BatchArgs a;
  memset(&a, 0, sizeof a);
  a.src = c->in.as<uint8_t>();
  a.src_stride = kLegacyBlockMax;
  a.src_len_uniform = (uint32_t)kLegacyBlockMax;
  a.src_total = n;
  a.dst = c->comp.as<uint8_t>();
  a.dst_stride = stride;
  a.dst_cap_uniform = (uint32_t)stride;
  a.nblocks = nb;
  a.out_len = c->comp_len.as<uint32_t>();
  a.status = c->status.as<int32_t>();
  st = run_compress_fast(c, a, acceleration);
  if (st) return st;
  uint32_t *h_len = (uint32_t *)c->h_pin;
  int32_t *h_st = (int32_t *)(h_len + nb);
  LZ4MI_HIP(c, hipMemcpyAsync(h_len, a.out_len, (size_t)nb * 4, hipMemcpyDeviceToHost, c->stream));
  LZ4MI_HIP(c, hipMemcpyAsync(h_st, a.status, (size_t)nb * 4, hipMemcpyDeviceToHost, c->stream));
  LZ4MI_HIP(c, hipStreamSynchronize(c->stream));
  for (uint32_t b = 0; b < nb; b++) {  // always the compressed form, even when larger (:75-83)
    if (h_st[b]) return h_st[b];
    if (cap - w < 4 || cap - w - 4 < h_len[b]) return LZ4MI_DST_TOO_SMALL;
    wr32(dst + w, h_len[b]);
    w += 4;
    LZ4MI_HIP(c, hipMemcpyAsync(dst + w, c->comp.as<uint8_t>() + (size_t)b * stride, h_len[b],
                                hipMemcpyDeviceToHost, c->stream));
    w += h_len[b];
  }
  LZ4MI_HIP(c, hipStreamSynchronize(c->stream));
  if (out_len) *out_len = w;
  return LZ4MI_OK;
}

// ------------------------------------------------------------- stream encoder
// lz4FrameEncoderTransformerWithOptions, lib/src/frame/lz4_frame_stream_encoder.dart:25-236.
// The reference buffers chunks up to blockMaxSize and encodes one block at a time with a
// 64 KiB history ring; here the buffer lives in device memory behind a 64 KiB history window
// and every update encodes all the full blocks it completes in one launch.

struct lz4mi_frame_encoder {
  lz4mi_ctx *c = nullptr;
  lz4mi_frame_options o;
  size_t bmax = 0;
  uint8_t *buf = nullptr;  // [history window: kHistoryWindow bytes][buffered input]
  size_t buf_cap = 0;
  size_t hist_len = 0;     // valid history bytes, ending at buf + kHistoryWindow (linked blocks)
  size_t pending = 0;      // buffered input bytes
  DevBuf dict;             // the external dictionary (independent blocks: used by every block)
  size_t dict_len = 0;
  uint64_t total_in = 0;
  lz4mi_xxh32_state hasher;
  bool finished = false;
};

namespace {

int32_t enc_reserve(lz4mi_frame_encoder *e, size_t pending_cap) {  // keeps history + buffered input
  lz4mi_ctx *c = e->c;
  const size_t want = kHistoryWindow + pending_cap + 64;
  if (want <= e->buf_cap) return LZ4MI_OK;
  const size_t grown = want + want / 4;
  uint8_t *nb = nullptr;
  LZ4MI_HIP(c, hipMalloc((void **)&nb, grown));
  if (e->buf) {
    const size_t keep_from = kHistoryWindow - e->hist_len;
    hipError_t er = hipMemcpyAsync(nb + keep_from, e->buf + keep_from, e->hist_len + e->pending,
                                   hipMemcpyDeviceToDevice, c->stream);
    if (er == hipSuccess) er = hipStreamSynchronize(c->stream);
    if (er != hipSuccess) {
      (void)hipFree(nb);
      return fail(c, er, "stream encoder buffer growth");
    }
    (void)hipFree(e->buf);
  }
  e->buf = nb;
  e->buf_cap = grown;
  return LZ4MI_OK;
}

// Encodes the first n buffered bytes as block records into dst.
int32_t enc_emit(lz4mi_frame_encoder *e, size_t n, uint8_t *dst, size_t cap, size_t *out_len) {
  lz4mi_ctx *c = e->c;
  const uint8_t *d_dict = nullptr;
  uint32_t dlen = 0;
  if (e->o.block_independence) {  // :187-188
    if (e->dict_len) { d_dict = e->dict.as<uint8_t>(); dlen = (uint32_t)e->dict_len; }
  } else if (e->hist_len) {       // :190-192
    d_dict = e->buf + kHistoryWindow - e->hist_len;
    dlen = (uint32_t)e->hist_len;
  }
  const size_t fb = lz4mi_frame_bound(n, &e->o);
  LZ4MI_HIP(c, c->out.ensure(fb + 16));
  uint64_t end = 0;
  int32_t st = encode_records(c, e->buf + kHistoryWindow, n, &e->o, d_dict, dlen, c->out.as<uint8_t>(), 0, &end);
  if (st) return st;
  if (end > cap) return LZ4MI_DST_TOO_SMALL;
  st = host_copy(c, dst, c->out.p, end, hipMemcpyDeviceToHost);
  if (st) return st;
  LZ4MI_HIP(c, hipStreamSynchronize(c->stream));
  *out_len = end;
  return LZ4MI_OK;
}

}  // namespace

extern "C" int32_t lz4mi_frame_encoder_create(lz4mi_ctx *c, const lz4mi_frame_options *o, const uint8_t *dict,
                                              size_t dict_len, lz4mi_frame_encoder **out) {
  if (!c || !out) return LZ4MI_RANGE;
  *out = nullptr;
  int32_t st = validate_options(o);
  if (st) return st;
  LZ4MI_HIP(c, ::lz4mi::enter_device(c));
  lz4mi_frame_encoder *e = new (std::nothrow) lz4mi_frame_encoder();
  if (!e) return LZ4MI_BACKEND;
  e->c = c;
  e->o = *o;
  e->bmax = block_max_bytes(o->block_size_id);
  lz4mi_xxh32_state_init(&e->hasher, 0);
  *out = e;  // from here on the caller destroys it, also after a failure below
  st = enc_reserve(e, e->bmax);
  if (st) return st;
  if (!dict) dict_len = 0;
  if (dict_len && o->block_independence) {
    LZ4MI_HIP(c, e->dict.ensure(dict_len));
    LZ4MI_HIP(c, hipMemcpyAsync(e->dict.p, dict, dict_len, hipMemcpyHostToDevice, c->stream));
    e->dict_len = dict_len;
  } else if (dict_len) {  // seeds the history ring (:64-73)
    const size_t keep = dict_len > kHistoryWindow ? kHistoryWindow : dict_len;
    LZ4MI_HIP(c, hipMemcpyAsync(e->buf + kHistoryWindow - keep, dict + dict_len - keep, keep,
                                hipMemcpyHostToDevice, c->stream));
    e->hist_len = keep;
  }
  LZ4MI_HIP(c, hipStreamSynchronize(c->stream));
  return LZ4MI_OK;
}

extern "C" void lz4mi_frame_encoder_destroy(lz4mi_frame_encoder *e) {
  if (!e) return;
  (void)hipSetDevice(e->c->device);
  (void)hipStreamSynchronize(e->c->stream);
  if (e->buf) (void)hipFree(e->buf);
  e->dict.release();
  delete e;
}

extern "C" size_t lz4mi_frame_encoder_bound(const lz4mi_frame_encoder *e, size_t n) {
  if (!e) return 0;
  const size_t blocks = (e->pending + n) / e->bmax + 1;  // + the short block finish may emit
  return blocks * (e->bmax + 8) + 8;
}

extern "C" int32_t lz4mi_frame_encoder_update(lz4mi_frame_encoder *e, const uint8_t *chunk, size_t n,
                                              uint8_t *dst, size_t cap, size_t *out_len) {
  if (!e || !out_len || (!chunk && n)) return LZ4MI_RANGE;
  *out_len = 0;
  if (e->finished) return LZ4MI_RANGE;
  if (n == 0) return LZ4MI_OK;  // :160-162
  lz4mi_ctx *c = e->c;
  LZ4MI_HIP(c, ::lz4mi::enter_device(c));
  e->total_in += n;
  if (e->o.has_content_size && e->total_in > e->o.content_size)  // :165-167
    return LZ4MI_FORMAT_CONTENT_SIZE_STREAM;
  int32_t st = enc_reserve(e, e->pending + n);
  if (st) return st;
  uint8_t *d_chunk = e->buf + kHistoryWindow + e->pending;
  st = host_copy(c, d_chunk, chunk, n, hipMemcpyHostToDevice);
  if (st) return st;
  // :168-170: the chunk is host memory; its content-checksum rounds run on a host thread while the GPU encodes
  // the blocks this chunk completes (joined before the call returns, on every path)
  struct Join {
    std::future<void> f;
    ~Join() { if (f.valid()) f.wait(); }
  } hashing;
  if (e->o.content_checksum) {
    lz4mi_xxh32_state *hs = &e->hasher;
    if (n >= (256u << 10)) hashing.f = run_beside([hs, chunk, n] { xxh_state_update_host(hs, chunk, n); });
    else xxh_state_update_host(hs, chunk, n);
  }
  e->pending += n;
  const size_t full = e->pending / e->bmax * e->bmax;  // :175-200, all completed blocks at once
  if (!full) {
    LZ4MI_HIP(c, hipStreamSynchronize(c->stream));  // the caller may reuse `chunk`
    return LZ4MI_OK;
  }
  st = enc_emit(e, full, dst, cap, out_len);
  if (st) return st;
  uint8_t *base = e->buf + kHistoryWindow;
  if (!e->o.block_independence) {  // appendHistory (:75-96): a full block refills the whole window
    LZ4MI_HIP(c, hipMemcpyAsync(e->buf, base + full - kHistoryWindow, kHistoryWindow, hipMemcpyDeviceToDevice,
                                c->stream));
    e->hist_len = kHistoryWindow;
  }
  const size_t rest = e->pending - full;
  if (rest) LZ4MI_HIP(c, hipMemcpyAsync(base, base + full, rest, hipMemcpyDeviceToDevice, c->stream));
  LZ4MI_HIP(c, hipStreamSynchronize(c->stream));
  e->pending = rest;
  return LZ4MI_OK;
}

extern "C" int32_t lz4mi_frame_encoder_finish(lz4mi_frame_encoder *e, uint8_t *dst, size_t cap, size_t *out_len) {
  if (!e || !out_len || !dst) return LZ4MI_RANGE;
  *out_len = 0;
  if (e->finished) return LZ4MI_RANGE;
  lz4mi_ctx *c = e->c;
  LZ4MI_HIP(c, ::lz4mi::enter_device(c));
  size_t w = 0;
  if (e->pending) {  // the short last block (:203-222)
    int32_t st = enc_emit(e, e->pending, dst, cap, &w);
    if (st) return st;
    e->pending = 0;
  }
  e->finished = true;
  *out_len = w;  // what was yielded before a contentSize error (:224-226) stays with the caller
  if (e->o.has_content_size && e->total_in != e->o.content_size) return LZ4MI_FORMAT_CONTENT_SIZE_STREAM;
  const size_t tail = 4 + (e->o.content_checksum ? 4 : 0);
  if (cap - w < tail) return LZ4MI_DST_TOO_SMALL;
  wr32(dst + w, 0);  // end mark (:228)
  w += 4;
  if (e->o.content_checksum) {  // :230-234
    wr32(dst + w, lz4mi_xxh32_state_digest(&e->hasher));
    w += 4;
  }
  *out_len = w;
  return LZ4MI_OK;
}

// ------------------------------------------------------------- stream decoder
// _Lz4FrameStreamDecoder, lib/src/frame/lz4_frame_stream_decoder.dart:62-599: a resumable
// state machine over the bytes fed so far.  The reference decodes one block per step; here
// every complete block record that is buffered goes into one launch, and what the blocks in
// front of a failing element produced is delivered before the failure is reported, which is
// the order a consumer of the reference's stream sees.

struct lz4mi_frame_decoder {
  enum State {
    kMagic, kSkippableSize, kSkippableSkip, kDescriptor, kHeaderChecksum, kResolveDict, kBlocks,
    kContentChecksum, kLegacyBlockSize, kLegacyBlockData, kLegacyAfterBlock, kFailed
  };
  lz4mi_ctx *c = nullptr;
  int64_t max_out = -1;
  std::vector<uint8_t> buf;  // _ChunkBuffer
  size_t start = 0;
  bool finished = false;
  State state = kMagic;
  uint64_t skip_left = 0;
  std::vector<uint8_t> descriptor;
  bool indep = true, block_ck = false, has_size = false, content_ck = false, has_dict_id = false;
  size_t bmax = 0;
  uint64_t content_size = 0;
  uint32_t dict_id = 0;
  bool have_dict = false;
  std::vector<uint8_t> dict;
  int dict_answer = 0;             // 0 none yet, 1 resolved, 2 no resolver, 3 resolver returned null
  std::vector<uint8_t> history;    // linked blocks: last 64 KiB of dictionary || frame output
  lz4mi_xxh32_state hasher;
  uint64_t frame_produced = 0, total_produced = 0;
  uint32_t legacy_csize = 0;
  bool legacy_has_pending = false;
  std::vector<uint8_t> legacy_pending;
  int32_t deferred = 0;            // failure to report once the output in front of it is delivered

  size_t avail() const { return buf.size() - start; }
  const uint8_t *at(size_t off = 0) const { return buf.data() + start + off; }
};

namespace {

typedef lz4mi_frame_decoder Dec;

void dec_reset_frame(Dec *d) {  // _resetFrameState :568-598
  d->skip_left = 0;
  d->descriptor.clear();
  d->indep = true;
  d->block_ck = d->has_size = d->content_ck = d->has_dict_id = false;
  d->bmax = 0;
  d->content_size = 0;
  d->dict_id = 0;
  d->have_dict = false;
  d->dict.clear();
  d->dict_answer = 0;
  d->history.clear();
  d->frame_produced = 0;
  d->legacy_csize = 0;
  d->legacy_has_pending = false;
  d->legacy_pending.clear();
}

void dec_append_history(Dec *d, const uint8_t *p, size_t n) {  // _appendHistory :534-559
  if (n >= kHistoryWindow) {
    d->history.assign(p + n - kHistoryWindow, p + n);
    return;
  }
  const size_t need = d->history.size() + n;
  if (need > kHistoryWindow) d->history.erase(d->history.begin(), d->history.begin() + (need - kHistoryWindow));
  d->history.insert(d->history.end(), p, p + n);
}

int32_t dec_fail(Dec *d, int32_t st) {
  d->state = Dec::kFailed;
  d->deferred = st;
  return st;
}

// Blocks of the current standard frame: everything complete in the buffer, in one launch.
// Returns OK with *out_len > 0 (delivered), OK with *out_len == 0 (state advanced, run again),
// LZ4MI_STREAM_NEED_INPUT, or a failure.
int32_t dec_blocks(Dec *d, uint8_t *dst, size_t cap, size_t *out_len) {
  std::vector<Blk> blocks;
  size_t pos = 0;
  int32_t walk_err = 0;
  bool end_mark = false;
  for (;;) {  // blockSize / blockData / blockChecksum :255-312, structure only
    if (d->avail() - pos < 4) break;
    const uint32_t raw = rd32(d->at(pos));
    if (raw == 0) { end_mark = true; break; }
    const uint32_t size = raw & 0x7FFFFFFFu;
    if (size > d->bmax) { walk_err = LZ4MI_CORRUPT_BLOCK_SIZE; break; }  // :273-275
    const size_t rec = 4 + (size_t)size + (d->block_ck ? 4 : 0);
    if (d->avail() - pos < rec) break;
    if ((blocks.size() + 1) * d->bmax > cap) break;
    Blk k = {pos + 4, size, (raw >> 31) & 1u, 0, 0};
    if (d->block_ck) { k.has_ck = 1; k.ck = rd32(d->at(pos + 4 + size)); }
    blocks.push_back(k);
    pos += rec;
  }
  if (blocks.empty()) {
    if (walk_err) return dec_fail(d, walk_err);
    if (end_mark) {  // :260-268
      d->start += 4;
      if (d->content_ck) { d->state = Dec::kContentChecksum; return LZ4MI_OK; }
      if (d->has_size && d->frame_produced != d->content_size) return dec_fail(d, LZ4MI_CORRUPT_CONTENT_SIZE);
      d->state = Dec::kMagic;
      return LZ4MI_OK;
    }
    if (d->avail() >= 4) {  // a complete record is waiting but dst cannot take one block
      const uint32_t size = rd32(d->at()) & 0x7FFFFFFFu;
      if (d->avail() >= 4 + (size_t)size + (d->block_ck ? 4 : 0) && cap < d->bmax) {
        *out_len = d->bmax;
        return LZ4MI_DST_TOO_SMALL;
      }
    }
    return LZ4MI_STREAM_NEED_INPUT;
  }
  const uint8_t *prefix = nullptr;
  size_t prefix_len = 0;
  if (d->indep) {  // :435-457
    if (d->have_dict) { prefix = d->dict.data(); prefix_len = d->dict.size(); }
  } else {         // :458-506: last 64 KiB of dictionary || output so far
    prefix = d->history.data();
    prefix_len = d->history.size();
  }
  DecodeExtras ex;
  ex.partial = true;
  const int64_t budget = d->max_out < 0 ? -1 : d->max_out - (int64_t)d->total_produced;  // :517-520
  uint32_t unused = 0;
  size_t produced = 0;
  int32_t st = decode_blocks(d->c, d->at(), pos, blocks, d->bmax, d->indep, prefix, prefix_len,
                             prefix_len != 0 || (d->indep && d->have_dict), 0, budget, false, false, false, dst,
                             cap, &produced, &unused, &ex);
  if (st) return dec_fail(d, st);
  if (produced) {
    if (d->content_ck) {  // _contentHasher.update(produced) :525
      xxh_state_update_host(&d->hasher, dst, produced);  // the bytes were just delivered into host memory
      st = LZ4MI_OK;
      if (st) return dec_fail(d, st);
    }
    if (!d->indep) dec_append_history(d, dst, produced);
    d->total_produced += produced;
    d->frame_produced += produced;
  }
  size_t used = 0;
  for (uint32_t b = 0; b < ex.good_blocks; b++)
    used = (size_t)blocks[b].payload_off + blocks[b].size + (blocks[b].has_ck ? 4 : 0);
  d->start += used;
  if (ex.deferred) {
    d->state = Dec::kFailed;
    d->deferred = ex.deferred;
    if (!produced) return ex.deferred;
  }
  *out_len = produced;
  return LZ4MI_OK;
}

int32_t dec_legacy_block(Dec *d) {  // legacyBlockData :356-376
  std::vector<Blk> one(1);
  one[0] = Blk{0, d->legacy_csize, 0, 0, 0};
  d->legacy_pending.resize(kLegacyBlockMax);
  size_t produced = 0;
  uint32_t unused = 0;
  // tmp ByteWriter(maxLength: 8 MiB); the running output limit is checked afterwards (:367-370)
  int32_t st = decode_blocks(d->c, d->at(), d->legacy_csize, one, kLegacyBlockMax, true, nullptr, 0, false, 0, -1,
                             false, false, false, d->legacy_pending.data(), kLegacyBlockMax, &produced, &unused);
  if (st) return st;
  if (d->max_out >= 0 && d->total_produced + produced > (uint64_t)d->max_out) return LZ4MI_LIMIT;
  d->total_produced += produced;
  d->legacy_pending.resize(produced);
  d->legacy_has_pending = true;
  d->start += d->legacy_csize;
  return LZ4MI_OK;
}

// Hands the pending legacy block to the caller; false (and *need set) when dst is too small.
bool dec_yield_legacy(Dec *d, uint8_t *dst, size_t cap, size_t *out_len) {
  const size_t n = d->legacy_pending.size();
  if (n > cap) { *out_len = n; return false; }
  if (n) memcpy(dst, d->legacy_pending.data(), n);
  *out_len = n;
  d->legacy_has_pending = false;
  d->legacy_pending.clear();
  return true;
}

}  // namespace

extern "C" int32_t lz4mi_frame_decoder_create(lz4mi_ctx *c, int64_t max_output_bytes, lz4mi_frame_decoder **out) {
  if (!c || !out) return LZ4MI_RANGE;
  Dec *d = new (std::nothrow) Dec();
  *out = d;
  if (!d) return LZ4MI_BACKEND;
  d->c = c;
  d->max_out = max_output_bytes < 0 ? -1 : max_output_bytes;
  lz4mi_xxh32_state_init(&d->hasher, 0);
  return LZ4MI_OK;
}

extern "C" void lz4mi_frame_decoder_destroy(lz4mi_frame_decoder *d) { delete d; }

extern "C" int32_t lz4mi_frame_decoder_feed(lz4mi_frame_decoder *d, const uint8_t *chunk, size_t n) {
  if (!d || (!chunk && n)) return LZ4MI_RANGE;  // _ChunkBuffer.add :615-622
  if (d->start && d->start == d->buf.size()) {
    d->buf.clear();
    d->start = 0;
  } else if (d->start > (1u << 20) && d->start > d->buf.size() / 2) {
    d->buf.erase(d->buf.begin(), d->buf.begin() + d->start);
    d->start = 0;
  }
  d->buf.insert(d->buf.end(), chunk, chunk + n);
  return LZ4MI_OK;
}

extern "C" int32_t lz4mi_frame_decoder_finish(lz4mi_frame_decoder *d) {  // finish :415-426
  if (!d) return LZ4MI_RANGE;
  d->finished = true;
  if (d->state == Dec::kFailed) return d->deferred;
  if (d->avail() == 0 && d->state == Dec::kMagic) return LZ4MI_OK;
  if (d->state == Dec::kLegacyAfterBlock && d->legacy_has_pending && d->avail() == 0) return LZ4MI_OK;
  return dec_fail(d, LZ4MI_FORMAT_EOF);
}

extern "C" int32_t lz4mi_frame_decoder_set_dictionary(lz4mi_frame_decoder *d, const uint8_t *dict, size_t n,
                                                      int32_t answer) {
  if (!d || answer < 1 || answer > 3 || (!dict && n)) return LZ4MI_RANGE;
  d->dict_answer = answer;
  if (answer == 1) d->dict.assign(dict, dict + n);
  return LZ4MI_OK;
}

extern "C" int32_t lz4mi_frame_decoder_next(lz4mi_frame_decoder *d, uint8_t *dst, size_t cap, size_t *out_len,
                                            uint32_t *dict_id_needed) {
  if (!d || !out_len || (!dst && cap)) return LZ4MI_RANGE;
  *out_len = 0;
  for (;;) {  // decodeNext :113-413
    switch (d->state) {
      case Dec::kFailed:
        return d->deferred;
      case Dec::kMagic: {  // :116-135
        if (d->avail() < 4) return LZ4MI_STREAM_NEED_INPUT;
        const uint32_t magic = rd32(d->at());
        d->start += 4;
        if (is_skippable_magic(magic)) { d->state = Dec::kSkippableSize; break; }
        if (magic == kLegacyMagic) { dec_reset_frame(d); d->state = Dec::kLegacyBlockSize; break; }
        if (magic != kFrameMagic) return dec_fail(d, LZ4MI_FORMAT_MAGIC);
        dec_reset_frame(d);
        d->state = Dec::kDescriptor;
        break;
      }
      case Dec::kSkippableSize:  // :137-143
        if (d->avail() < 4) return LZ4MI_STREAM_NEED_INPUT;
        d->skip_left = rd32(d->at());
        d->start += 4;
        d->state = Dec::kSkippableSkip;
        break;
      case Dec::kSkippableSkip: {  // :145-156
        const size_t take = (size_t)std::min<uint64_t>(d->skip_left, d->avail());
        d->start += take;
        d->skip_left -= take;
        if (d->skip_left) return LZ4MI_STREAM_NEED_INPUT;
        d->state = Dec::kMagic;
        break;
      }
      case Dec::kDescriptor: {  // :158-223
        if (d->avail() < 2) return LZ4MI_STREAM_NEED_INPUT;
        const uint32_t flg = d->at()[0], bd = d->at()[1];
        const size_t required = 2 + ((flg >> 3) & 1 ? 8 : 0) + ((flg & 1) ? 4 : 0);
        if (d->avail() < required) return LZ4MI_STREAM_NEED_INPUT;
        d->descriptor.assign(d->at(), d->at() + required);
        const uint8_t *p = d->at();
        d->start += required;
        if (((flg >> 6) & 3) != 1) return dec_fail(d, LZ4MI_UNSUPPORTED_VERSION);
        d->indep = ((flg >> 5) & 1) != 0;
        d->block_ck = ((flg >> 4) & 1) != 0;
        d->has_size = ((flg >> 3) & 1) != 0;
        d->content_ck = ((flg >> 2) & 1) != 0;
        d->has_dict_id = (flg & 1) != 0;
        if ((flg >> 1) & 1) return dec_fail(d, LZ4MI_FORMAT_FLG_RESERVED);
        if (bd & 0x8F) return dec_fail(d, LZ4MI_FORMAT_BD_RESERVED);
        d->bmax = block_max_bytes((bd >> 4) & 7);
        if (!d->bmax) return dec_fail(d, LZ4MI_FORMAT_BLOCK_MAX);
        size_t off = 2;
        if (d->has_size) {
          d->content_size = (uint64_t)rd32(p + off) | ((uint64_t)rd32(p + off + 4) << 32);
          if (d->content_size >> 63) return dec_fail(d, LZ4MI_UNSUPPORTED_CONTENT_SIZE);  // :208-212
          off += 8;
        }
        if (d->has_dict_id) d->dict_id = rd32(p + off);
        d->state = Dec::kHeaderChecksum;
        break;
      }
      case Dec::kHeaderChecksum: {  // :225-235
        if (d->avail() < 1) return LZ4MI_STREAM_NEED_INPUT;
        const uint32_t hc = d->at()[0];
        d->start += 1;
        if (hc != ((descriptor_hash(d->descriptor.data(), d->descriptor.size()) >> 8) & 0xff))
          return dec_fail(d, LZ4MI_CORRUPT_HEADER_CK);
        d->state = Dec::kResolveDict;
        break;
      }
      case Dec::kResolveDict:  // :237-253
        if (d->has_dict_id) {
          if (d->dict_answer == 0) {
            if (dict_id_needed) *dict_id_needed = d->dict_id;
            return LZ4MI_NEED_DICT;
          }
          if (d->dict_answer == 2) return dec_fail(d, LZ4MI_UNSUPPORTED_DICT);
          if (d->dict_answer == 3) {
            if (dict_id_needed) *dict_id_needed = d->dict_id;
            return dec_fail(d, LZ4MI_DICT_NOT_FOUND);
          }
          d->have_dict = true;
          if (!d->indep) dec_append_history(d, d->dict.data(), d->dict.size());
        }
        lz4mi_xxh32_state_init(&d->hasher, 0);
        d->state = Dec::kBlocks;
        break;
      case Dec::kBlocks: {
        int32_t st = dec_blocks(d, dst, cap, out_len);
        if (st) return st;
        if (*out_len) return LZ4MI_OK;
        break;
      }
      case Dec::kContentChecksum: {  // :314-326
        if (d->avail() < 4) return LZ4MI_STREAM_NEED_INPUT;
        const uint32_t expected = rd32(d->at());
        d->start += 4;
        if (lz4mi_xxh32_state_digest(&d->hasher) != expected) return dec_fail(d, LZ4MI_CORRUPT_CONTENT_CK);
        if (d->has_size && d->frame_produced != d->content_size) return dec_fail(d, LZ4MI_CORRUPT_CONTENT_SIZE);
        d->state = Dec::kMagic;
        break;
      }
      case Dec::kLegacyBlockSize: {  // :328-354
        if (d->legacy_has_pending) { d->state = Dec::kLegacyAfterBlock; break; }
        if (d->avail() < 4) {
          if (d->finished) { d->state = Dec::kMagic; break; }
          return LZ4MI_STREAM_NEED_INPUT;
        }
        const uint32_t next = rd32(d->at());
        if (is_frame_boundary(next)) { d->state = Dec::kMagic; break; }
        d->start += 4;
        if (next == 0) return dec_fail(d, LZ4MI_CORRUPT_LEGACY_SIZE);
        d->legacy_csize = next;
        d->state = Dec::kLegacyBlockData;
        break;
      }
      case Dec::kLegacyBlockData: {  // :356-376
        if (d->avail() < d->legacy_csize) return LZ4MI_STREAM_NEED_INPUT;
        int32_t st = dec_legacy_block(d);
        if (st) return dec_fail(d, st);
        d->state = Dec::kLegacyAfterBlock;
        break;
      }
      case Dec::kLegacyAfterBlock: {  // :378-410
        if (!d->legacy_has_pending) { d->state = Dec::kLegacyBlockSize; break; }
        Dec::State after;
        if (d->avail() < 4) {
          if (!d->finished) return LZ4MI_STREAM_NEED_INPUT;
          after = Dec::kMagic;
        } else if (is_frame_boundary(rd32(d->at()))) {
          after = Dec::kMagic;
        } else {
          if (d->legacy_pending.size() != kLegacyBlockMax) return dec_fail(d, LZ4MI_CORRUPT_LEGACY_NOT_FULL);
          after = Dec::kLegacyBlockSize;
        }
        if (!dec_yield_legacy(d, dst, cap, out_len)) return LZ4MI_DST_TOO_SMALL;
        d->state = after;
        if (*out_len) return LZ4MI_OK;
        break;
      }
    }
  }
}
