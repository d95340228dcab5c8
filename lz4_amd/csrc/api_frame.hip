// liblz4mi: frame layer.  Host code parses/validates headers and walks the block
// size words; every per-block loop body of the reference's dispatchers
// (lib/src/frame/lz4_frame_encoder.dart:154-261, lz4_frame_decoder.dart:238-304)
// runs as batched kernels over all blocks of the frame.
#include <future>

#include "api_common.h"
#include "frame_host.h"

using namespace lz4mi;

namespace lz4mi {

// xxHash32 of a frame descriptor (2..14 bytes, always < 16 so only the short tail
// path of lib/src/xxhash/xxh32.dart:215-239 applies).  Host-side because the header is
// built and validated on the host; bulk hashing runs on the GPU (xxh32_kernel.hip).
uint32_t descriptor_hash(const uint8_t *p, size_t len) {
  auto rotl = [](uint32_t x, int r) { return (x << r) | (x >> (32 - r)); };
  uint32_t h = 0x165667B1u + (uint32_t)len;
  size_t k = 0;
  for (; k + 4 <= len; k += 4) {
    uint32_t w = (uint32_t)p[k] | ((uint32_t)p[k + 1] << 8) | ((uint32_t)p[k + 2] << 16) | ((uint32_t)p[k + 3] << 24);
    h = rotl(h + w * 0xC2B2AE3Du, 17) * 0x27D4EB2Fu;
  }
  for (; k < len; k++) h = rotl(h + p[k] * 0x165667B1u, 11) * 0x9E3779B1u;
  h ^= h >> 15;
  h *= 0x85EBCA77u;
  h ^= h >> 13;
  h *= 0xC2B2AE3Du;
  h ^= h >> 16;
  return h;
}

int32_t validate_options(const lz4mi_frame_options *o) {
  if (!o) return LZ4MI_RANGE;
  if (o->acceleration < 1) return LZ4MI_RANGE;        // lz4_frame_options.dart:89-91
  if (!block_max_bytes(o->block_size_id)) return LZ4MI_RANGE;
  if (o->compression == 1 && o->hc_max_search_depth < 1) return LZ4MI_RANGE;
  if (o->has_content_size && (o->content_size >> 63)) return LZ4MI_RANGE;  // :92-95
  return LZ4MI_OK;
}

size_t build_header(const lz4mi_frame_options *o, uint8_t *h) {  // lz4_frame_encoder.dart:110-144
  wr32(h, kFrameMagic);
  h[4] = (uint8_t)((1u << 6) | ((o->block_independence ? 1u : 0u) << 5) | ((o->block_checksum ? 1u : 0u) << 4) |
                   ((o->has_content_size ? 1u : 0u) << 3) | ((o->content_checksum ? 1u : 0u) << 2) |
                   (o->has_dict_id ? 1u : 0u));
  h[5] = (uint8_t)(((uint32_t)o->block_size_id & 7u) << 4);
  size_t n = 6;
  if (o->has_content_size) {
    wr32(h + n, (uint32_t)(o->content_size & 0xFFFFFFFFu));
    wr32(h + n + 4, (uint32_t)(o->content_size >> 32));
    n += 8;
  }
  if (o->has_dict_id) {
    wr32(h + n, o->dict_id);
    n += 4;
  }
  h[n] = (uint8_t)((descriptor_hash(h + 4, n - 4) >> 8) & 0xff);
  return n + 1;
}

// Compresses every block of a device-resident range and plans its frame record.
int32_t shard_compress(lz4mi_ctx *c, const uint8_t *d_src, uint64_t n, const lz4mi_frame_options *o,
                       const uint8_t *d_dict, uint32_t dict_len, uint32_t *nblocks_out) {
  const size_t bmax = block_max_bytes(o->block_size_id);
  if (n / bmax > 0x7ffffff0u) return LZ4MI_RANGE;
  const uint32_t nb = (uint32_t)((n + bmax - 1) / bmax);
  *nblocks_out = nb;
  c->shard_blocks = nb;
  c->shard_block_checksum = o->block_checksum;
  if (nb == 0) return LZ4MI_OK;
  const size_t blk = n < bmax ? (size_t)n : bmax;
  const size_t stride = round_up(bound(blk), 64);
  LZ4MI_HIP(c, c->comp.ensure((size_t)nb * stride));
  LZ4MI_HIP(c, c->comp_len.ensure((size_t)nb * 4));
  LZ4MI_HIP(c, c->status.ensure((size_t)nb * 4));
  LZ4MI_HIP(c, c->stored_word.ensure((size_t)nb * 4));
  LZ4MI_HIP(c, c->payload_addr.ensure((size_t)nb * 8));
  LZ4MI_HIP(c, c->payload_len.ensure((size_t)nb * 4));
  LZ4MI_HIP(c, c->record_size.ensure((size_t)nb * 4));
  LZ4MI_HIP(c, c->hashes.ensure((size_t)nb * 4));

  BatchArgs a;
  memset(&a, 0, sizeof a);
  a.src = d_src;
  a.src_stride = bmax;
  a.src_len_uniform = (uint32_t)bmax;
  a.src_total = n;
  a.dst = c->comp.as<uint8_t>();
  a.dst_stride = stride;
  a.dst_cap_uniform = (uint32_t)stride;
  a.dict = d_dict;
  a.dict_len = d_dict ? dict_len : 0;
  a.linked = o->block_independence ? 0u : 1u;
  a.conformant = o->spec_conformant ? 1u : 0u;
  a.nblocks = nb;
  a.out_len = c->comp_len.as<uint32_t>();
  a.status = c->status.as<int32_t>();
  int32_t st = o->compression == 0 ? run_compress_fast(c, a, o->acceleration)
                                   : run_compress_hc(c, a, o->hc_max_search_depth);
  if (st) return st;

  FramePlanArgs p;
  p.src = d_src;
  p.src_total = n;
  p.block_max = (uint32_t)bmax;
  p.comp = c->comp.as<uint8_t>();
  p.comp_stride = stride;
  p.comp_len = c->comp_len.as<uint32_t>();
  p.nblocks = nb;
  p.block_checksum = o->block_checksum;
  p.stored_word = c->stored_word.as<uint32_t>();
  p.payload_addr = c->payload_addr.as<uint64_t>();
  p.payload_len = c->payload_len.as<uint32_t>();
  p.record_size = c->record_size.as<uint32_t>();
  launch_frame_plan(p, c->stream);
  if (o->block_checksum) {  // checksum of the payload as stored (:242-257)
    BatchArgs h;
    memset(&h, 0, sizeof h);
    h.src = nullptr;  // absolute addresses
    h.src_off = c->payload_addr.as<uint64_t>();
    h.src_len = c->payload_len.as<uint32_t>();
    h.nblocks = nb;
    launch_xxh32_batch(h, c->hashes.as<uint32_t>(), c->stream);
  }
  LZ4MI_HIP(c, hipGetLastError());
  return LZ4MI_OK;
}

int32_t shard_write(lz4mi_ctx *c, const uint64_t *d_rec_off, uint8_t *d_dst) {
  const uint32_t nb = c->shard_blocks;
  if (!nb) return LZ4MI_OK;
  launch_frame_words(d_dst, d_rec_off, c->stored_word.as<uint32_t>(), c->payload_len.as<uint32_t>(),
                     c->shard_block_checksum ? c->hashes.as<uint32_t>() : nullptr, nb, c->stream);
  launch_seg_copy(c->payload_addr.as<uint64_t>(), d_dst, d_rec_off, 4, c->payload_len.as<uint32_t>(), nb,
                  c->stream);
  LZ4MI_HIP(c, hipGetLastError());
  return LZ4MI_OK;
}

int32_t encode_records(lz4mi_ctx *c, const uint8_t *d_src, uint64_t n, const lz4mi_frame_options *o,
                       const uint8_t *d_dict, uint32_t dict_len, uint8_t *d_dst, uint64_t base, uint64_t *end) {
  uint32_t nb = 0;
  int32_t st = shard_compress(c, d_src, n, o, d_dict, dict_len, &nb);
  if (st) return st;
  LZ4MI_HIP(c, c->rec_off.ensure((size_t)(nb + 1) * 8));
  LZ4MI_HIP(c, c->total.ensure(64));
  st = pin_reserve(c, 256);
  if (st) return st;
  launch_scan_sizes(c->record_size.as<uint32_t>(), nb, base, c->rec_off.as<uint64_t>(), c->total.as<uint64_t>(),
                    c->stream);
  st = shard_write(c, c->rec_off.as<uint64_t>(), d_dst);
  if (st) return st;
  uint64_t *h_total = (uint64_t *)((uint8_t *)c->h_pin + 128);
  LZ4MI_HIP(c, hipMemcpyAsync(h_total, c->total.p, 8, hipMemcpyDeviceToHost, c->stream));
  LZ4MI_HIP(c, hipStreamSynchronize(c->stream));
  *end = *h_total;
  return LZ4MI_OK;
}

}  // namespace lz4mi

extern "C" void lz4mi_frame_options_default(lz4mi_frame_options *o) {  // lz4_frame_options.dart:78-88
  memset(o, 0, sizeof *o);
  o->block_size_id = 7;
  o->block_independence = 1;
  o->acceleration = 1;
  o->hc_max_search_depth = 64;
}

extern "C" size_t lz4mi_frame_bound(size_t n, const lz4mi_frame_options *o) {
  size_t bmax = block_max_bytes(o ? o->block_size_id : 7);
  if (!bmax) bmax = 4u << 20;
  size_t nb = (n + bmax - 1) / bmax;
  return 19 + nb * 8 + n + 8;  // a block is never stored larger than its source (:235)
}

extern "C" int32_t lz4mi_frame_header(const lz4mi_frame_options *o, uint8_t *dst, size_t cap, size_t *out_len) {
  int32_t st = validate_options(o);
  if (st) return st;
  uint8_t h[32];
  size_t n = build_header(o, h);
  if (n > cap) return LZ4MI_DST_TOO_SMALL;
  memcpy(dst, h, n);
  if (out_len) *out_len = n;
  return LZ4MI_OK;
}

extern "C" int32_t lz4mi_frame_shard_compress(lz4mi_ctx *c, const uint8_t *d_src, uint64_t n,
                                              const lz4mi_frame_options *o, const uint8_t *d_dict,
                                              uint32_t dict_len, uint32_t *d_record_sizes, uint32_t *nblocks_out) {
  if (!c || !nblocks_out) return LZ4MI_RANGE;
  int32_t st = validate_options(o);
  if (st) return st;
  if (!o->block_independence) return LZ4MI_UNSUPPORTED_HERE;  // a rank's first block would need the previous rank's bytes
  LZ4MI_HIP(c, ::lz4mi::enter_device(c));
  st = shard_compress(c, d_src, n, o, d_dict, dict_len, nblocks_out);
  if (st) return st;
  if (d_record_sizes && *nblocks_out)
    LZ4MI_HIP(c, hipMemcpyAsync(d_record_sizes, c->record_size.p, (size_t)*nblocks_out * 4,
                                hipMemcpyDeviceToDevice, c->stream));
  c->shard_valid = true;
  c->shard_src = d_src;
  c->shard_n = n;
  c->shard_block_size_id = o->block_size_id;
  return LZ4MI_OK;
}

extern "C" int32_t lz4mi_frame_shard_write(lz4mi_ctx *c, const uint8_t *d_src, uint64_t n,
                                           const lz4mi_frame_options *o, const uint64_t *d_record_offsets,
                                           uint8_t *d_dst, uint64_t cap) {
  if (!c || !o || !d_record_offsets || !d_dst) return LZ4MI_RANGE;
  // the records come out of the scratch step 1 left behind: refuse when anything ran in between, or
  // when this is not step 1's shard
  const bool valid = c->shard_valid && c->shard_src == d_src && c->shard_n == n &&
                     c->shard_block_size_id == o->block_size_id &&
                     (c->shard_block_checksum != 0) == (o->block_checksum != 0);
  LZ4MI_HIP(c, ::lz4mi::enter_device(c));
  if (!valid) {
    c->err = "lz4mi_frame_shard_write: no matching lz4mi_frame_shard_compress directly before this call";
    return LZ4MI_RANGE;
  }
  const uint32_t nb = c->shard_blocks;
  if (nb) {  // every record inside [0, cap) of the window, before anything is written
    int32_t st = pin_reserve(c, 64);
    if (st) return st;
    LZ4MI_HIP(c, c->total.ensure(64));
    uint32_t *flag = c->total.as<uint32_t>();
    LZ4MI_HIP(c, hipMemsetAsync(flag, 0, 4, c->stream));
    launch_record_bounds(d_record_offsets, c->record_size.as<uint32_t>(), nb, cap, flag, c->stream);
    LZ4MI_HIP(c, hipGetLastError());
    LZ4MI_HIP(c, hipMemcpyAsync(c->h_pin, flag, 4, hipMemcpyDeviceToHost, c->stream));
    LZ4MI_HIP(c, hipStreamSynchronize(c->stream));
    if (*(uint32_t *)c->h_pin) {
      c->shard_valid = true;  // nothing was touched: the caller may try again with a larger window
      return LZ4MI_DST_TOO_SMALL;
    }
  }
  int32_t st = shard_write(c, d_record_offsets, d_dst);
  c->shard_valid = st == LZ4MI_OK;  // writing again (another window) is allowed
  return st;
}

namespace lz4mi {

// host_hash: the content checksum when the caller computes it itself from a host copy of the bytes (on a host
// thread, beside this call's GPU work); null: the source is only on the device and the single-stream kernel
// hashes it on the context's side stream, beside the block launches (it depends on nothing but d_src).
static int32_t frame_encode_device_impl(lz4mi_ctx *c, const uint8_t *d_src, uint64_t n, const lz4mi_frame_options *o,
                                        const uint8_t *d_dict, uint32_t dict_len, uint8_t *d_dst, uint64_t cap,
                                        uint64_t *out_len, std::future<uint32_t> *host_hash,
                                        bool patch_later = false) {
  if (cap < lz4mi_frame_bound((size_t)n, o)) return LZ4MI_DST_TOO_SMALL;
  uint8_t hdr[32];
  const size_t hlen = build_header(o, hdr);
  int32_t st = pin_reserve(c, 256);
  if (st) return st;
  memcpy(c->h_pin, hdr, hlen);
  LZ4MI_HIP(c, hipMemcpyAsync(d_dst, c->h_pin, hlen, hipMemcpyHostToDevice, c->stream));
  LZ4MI_HIP(c, c->words.ensure(64));
  // device-resident source of some size: a host thread hashes it from pinned pieces (api_common.h: hash_on_host)
  hipError_t via_host_err = hipSuccess;
  std::future<uint32_t> via_host;  // joined by its destructor on every return path
  if (o->content_checksum && !host_hash && hash_on_host(n)) {
    st = hash_host_ready(c);
    if (st) return st;
    LZ4MI_HIP(c, hipEventRecord(c->side_fork, c->stream));  // whatever produced d_src on this stream is done
    LZ4MI_HIP(c, hipStreamWaitEvent(c->side_stream, c->side_fork, 0));
    hipError_t *ep = &via_host_err;
    via_host = run_beside([c, d_src, n, ep] {
      uint32_t h = 0;
      *ep = xxh32_device_on_host(c, d_src, n, 0, &h);
      return h;
    });
    host_hash = &via_host;
  }
  const bool side = o->content_checksum && !host_hash;
  if (side) {  // Xxh32 over all uncompressed content (:148,163,266)
    st = side_stream_ready(c);
    if (st) return st;
    LZ4MI_HIP(c, hipEventRecord(c->side_fork, c->stream));  // whatever produced d_src on this stream is done
    LZ4MI_HIP(c, hipStreamWaitEvent(c->side_stream, c->side_fork, 0));
    launch_xxh32_serial(d_src, n, 0, c->words.as<uint32_t>(), c->side_stream);
    LZ4MI_HIP(c, hipGetLastError());
    LZ4MI_HIP(c, hipEventRecord(c->side_join, c->side_stream));
  }
  uint64_t body_end = 0;
  st = encode_records(c, d_src, n, o, d_dict, dict_len, d_dst, hlen, &body_end);
  if (side) {  // also on a failure: nothing of this call stays behind on the side stream
    const hipError_t e = hipStreamWaitEvent(c->stream, c->side_join, 0);
    if (!st) LZ4MI_HIP(c, e);
  }
  if (st) return st;
  if (o->content_checksum && host_hash && patch_later) {
    // the caller downloads the frame while the hash is still running and writes the word into its copy
    LZ4MI_HIP(c, hipMemsetAsync(c->words.p, 0, 4, c->stream));
  } else if (o->content_checksum && host_hash) {
    uint32_t *h = (uint32_t *)((uint8_t *)c->h_pin + 64);
    *h = host_hash->get();
    LZ4MI_HIP(c, via_host_err);
    LZ4MI_HIP(c, hipMemcpyAsync(c->words.p, h, 4, hipMemcpyHostToDevice, c->stream));
  }
  // end mark (:263) [+ content checksum] behind the last record
  launch_put_words(d_dst + body_end, c->words.as<uint32_t>(), o->content_checksum ? 1 : 0, true, c->stream);
  LZ4MI_HIP(c, hipGetLastError());
  if (out_len) *out_len = body_end + 4 + (o->content_checksum ? 4 : 0);
  // raised after all the work, like the reference (:269-271): the frame is in d_dst, *out_len is set
  if (o->has_content_size && n != o->content_size) return LZ4MI_FORMAT_CONTENT_SIZE_ARG;
  return LZ4MI_OK;
}

}  // namespace lz4mi

extern "C" int32_t lz4mi_frame_encode_device(lz4mi_ctx *c, const uint8_t *d_src, uint64_t n,
                                             const lz4mi_frame_options *o, const uint8_t *d_dict,
                                             uint32_t dict_len, uint8_t *d_dst, uint64_t cap, uint64_t *out_len) {
  if (!c) return LZ4MI_RANGE;
  int32_t st = validate_options(o);
  if (st) return st;
  LZ4MI_HIP(c, ::lz4mi::enter_device(c));
  return frame_encode_device_impl(c, d_src, n, o, d_dict, dict_len, d_dst, cap, out_len, nullptr);
}

extern "C" int32_t lz4mi_frame_encode(lz4mi_ctx *c, const uint8_t *src, size_t n, const lz4mi_frame_options *o,
                                      const uint8_t *dict, size_t dict_len, uint8_t *dst, size_t cap,
                                      size_t *out_len) {
  if (!c) return LZ4MI_RANGE;
  if (out_len) *out_len = 0;
  int32_t st = validate_options(o);
  if (st) return st;
  LZ4MI_HIP(c, ::lz4mi::enter_device(c));
  // The source is host memory: its content checksum (one serial chain of rounds, :148,163,266) runs on a host
  // thread while the GPU compresses the blocks.  The future's destructor joins the thread on every return path.
  std::future<uint32_t> content_hash;
  if (o->content_checksum) content_hash = run_beside([src, n] { return xxh32_host(src, n); });
  const size_t fb = lz4mi_frame_bound(n, o);
  LZ4MI_HIP(c, c->in.ensure(n + 16));
  LZ4MI_HIP(c, c->out.ensure(fb + 16));
  st = host_copy(c, c->in.p, src, n, hipMemcpyHostToDevice);
  if (st) return st;
  const uint8_t *d_dict = nullptr;
  if (dict && dict_len) {
    LZ4MI_HIP(c, c->dictbuf.ensure(dict_len));
    LZ4MI_HIP(c, hipMemcpyAsync(c->dictbuf.p, dict, dict_len, hipMemcpyHostToDevice, c->stream));
    d_dict = c->dictbuf.as<uint8_t>();
  } else {
    dict_len = 0;
  }
  uint64_t total = 0;
  st = frame_encode_device_impl(c, c->in.as<uint8_t>(), n, o, d_dict, (uint32_t)dict_len, c->out.as<uint8_t>(), fb,
                                &total, o->content_checksum ? &content_hash : nullptr, true);
  if (st && st != LZ4MI_FORMAT_CONTENT_SIZE_ARG) return st;
  if (total > cap) return LZ4MI_DST_TOO_SMALL;
  st = host_copy(c, dst, c->out.p, total, hipMemcpyDeviceToHost);
  if (st) return st;
  if (o->content_checksum) {  // the frame's last word (:266), from the host thread that hashed src meanwhile
    const uint32_t h = content_hash.get();
    for (int k = 0; k < 4; k++) dst[total - 4 + k] = (uint8_t)(h >> (8 * k));
  }
  if (out_len) *out_len = total;
  // raised after all the work, like the reference (:269-271)
  if (o->has_content_size && (uint64_t)n != o->content_size) return LZ4MI_FORMAT_CONTENT_SIZE_ARG;
  return LZ4MI_OK;
}

// --------------------------------------------------------------- frame decode

namespace lz4mi {

// Header parse + validation in the reference's order (lz4_frame_decoder.dart:166-223).
int32_t parse_header(const uint8_t *p, size_t rem, Header *h) {
  if (rem < 3) return LZ4MI_FORMAT_EOF;
  uint32_t flg = p[0], bd = p[1];
  if (((flg >> 6) & 3) != 1) return LZ4MI_UNSUPPORTED_VERSION;
  h->indep = (flg >> 5) & 1;
  h->block_ck = (flg >> 4) & 1;
  h->has_size = (flg >> 3) & 1;
  h->content_ck = (flg >> 2) & 1;
  h->has_dict = flg & 1;
  if ((flg >> 1) & 1) return LZ4MI_FORMAT_FLG_RESERVED;
  if (bd & 0x8F) return LZ4MI_FORMAT_BD_RESERVED;
  h->bmax = block_max_bytes((bd >> 4) & 7);
  if (!h->bmax) return LZ4MI_FORMAT_BLOCK_MAX;
  size_t off = 2;
  h->content_size = 0;
  if (h->has_size) {
    if (rem - off < 8) return LZ4MI_FORMAT_EOF;
    h->content_size = (uint64_t)rd32(p + off) | ((uint64_t)rd32(p + off + 4) << 32);
    if (h->content_size >> 63) return LZ4MI_FORMAT_INT_OVERFLOW;  // byte_reader.dart:74-76
    off += 8;
  }
  h->dict_id = 0;
  if (h->has_dict) {
    if (rem - off < 4) return LZ4MI_FORMAT_EOF;
    h->dict_id = rd32(p + off);
    off += 4;
  }
  if (rem - off < 1) return LZ4MI_FORMAT_EOF;
  if (p[off] != ((descriptor_hash(p, off) >> 8) & 0xff)) return LZ4MI_CORRUPT_HEADER_CK;
  h->len = off + 1;
  return LZ4MI_OK;
}

// Most bytes a compressed block of `size` bytes can decode to: every input byte is at best a length
// extension byte worth 255 output bytes (a token's 15 + 19 come with three bytes of input).
static inline uint64_t decoded_bound(uint32_t size, size_t slot) {
  const uint64_t m = 255ull * size + 32u;
  return m < slot ? m : slot;
}

uint64_t planned_output(const Blk &k, size_t slot) { return k.raw ? k.size : decoded_bound(k.size, slot); }

// Device bytes the block slots of one launch may take; more blocks go through further launches.
static size_t decode_scratch_budget() {
  size_t mb = 2048;
  if (const char *e = getenv("LZ4MI_DECODE_SCRATCH_MB")) {  // read per call: tests shrink it to force several launches
    long n = atol(e);
    if (n >= 1 && n <= (1l << 20)) mb = (size_t)n;
  }
  return mb << 20;
}

// A linked frame of at least this many blocks is decoded all blocks at once (fewer: the chain's one launch is
// quicker than the fixed cost of the passes).
constexpr uint32_t kLinkMinBlocks = 4;
// Output bytes resolved together (four bytes of scratch each); LZ4MI_LINK_CHUNK sets another size (tests: several
// chunks from a small frame).
static uint64_t link_chunk_bytes() {
  const char *v = getenv("LZ4MI_LINK_CHUNK");
  const unsigned long long n = v ? strtoull(v, nullptr, 10) : 0;
  return n ? n : (256ull << 20);
}

// LZ4MI_LINK_RESOLVE: 0 = the chain for every linked frame, 2 = a frame that ends up on the chain is an error
// (tests: the path under test is the one that ran); read per call.
static int link_resolve_mode() {
  const char *v = getenv("LZ4MI_LINK_RESOLVE");
  return v && v[0] >= '0' && v[0] <= '2' ? v[0] - '0' : 1;
}

// Sizes of all blocks (h_olen, h_stat), then chunk by chunk: literals and distances, doubling rounds until no
// distance moves, the copied bytes.  limit_total: what the caller's buffer and output limit allow.  h_word: a pinned word for the "something moved" read-back.  *resolved = false when a block is not for this path (the output buffer may
// hold partial data then; the chain writes all of it again).
static int32_t link_resolve(lz4mi_ctx *c, LinkArgs la, uint32_t nb, const uint8_t *d_dict, uint64_t limit_total,
                            uint32_t *h_olen, int32_t *h_stat, uint32_t *h_word, bool *resolved) {
  *resolved = false;
  const uint64_t slot = la.block_max;
  la.first = 0;
  la.nblocks = nb;
  launch_link_walk(la, false, c->stream);
  LZ4MI_HIP(c, hipGetLastError());
  LZ4MI_HIP(c, hipMemcpyAsync(h_olen, la.out_len, (size_t)nb * 12, hipMemcpyDeviceToHost, c->stream));  // sizes, statuses, block hashes
  LZ4MI_HIP(c, hipStreamSynchronize(c->stream));
  uint64_t total = 0;
  for (uint32_t b = 0; b < nb; b++) {
    if (h_stat[b] != 0 || h_olen[b] > slot) return LZ4MI_OK;
    total += h_olen[b];
  }
  if (total > limit_total) return LZ4MI_OK;  // the caller's buffer or output limit ends inside the frame: the chain reports it
  // scratch: block offsets, tile flags and the moved word, the distance array of the largest chunk
  const uint64_t chunk_bytes = link_chunk_bytes();
  uint64_t chunk_cap = 0, run = 0;
  for (uint32_t b = 0; b < nb; b++) {  // chunks as the loop below cuts them
    if (run && run + h_olen[b] > chunk_bytes) {
      chunk_cap = run > chunk_cap ? run : chunk_cap;
      run = 0;
    }
    run += h_olen[b];
  }
  chunk_cap = run > chunk_cap ? run : chunk_cap;
  const size_t starts_bytes = round_up((size_t)nb * 8, 256);
  const size_t flag_bytes = round_up((size_t)link_tiles(chunk_cap) + 16, 256);
  const size_t dist_bytes = (size_t)link_dist_words(chunk_cap) * 4;
  LZ4MI_HIP(c, c->link.ensure(starts_bytes + flag_bytes + dist_bytes));
  LZ4MI_HIP(c, c->out.ensure((size_t)total + 16));
  uint64_t *d_start = c->link.as<uint64_t>();
  uint8_t *d_flags = c->link.as<uint8_t>() + starts_bytes;  // [moved u32][pad][one byte per tile]
  uint32_t *d_moved = (uint32_t *)d_flags;
  uint8_t *d_done = d_flags + 16;
  uint32_t *d_dist = (uint32_t *)(c->link.as<uint8_t>() + starts_bytes + flag_bytes);
  std::vector<uint64_t> starts(nb);
  uint64_t at = 0;
  for (uint32_t b = 0; b < nb; b++) {
    starts[b] = at;
    at += h_olen[b];
  }
  LZ4MI_HIP(c, hipMemcpyAsync(d_start, starts.data(), (size_t)nb * 8, hipMemcpyHostToDevice, c->stream));
  la.dst = c->out.as<uint8_t>();
  la.start = d_start;
  la.dist = d_dist;
  uint32_t b0 = 0;
  while (b0 < nb) {
    uint32_t b1 = b0;
    uint64_t n = 0;
    while (b1 < nb && (b1 == b0 || n + h_olen[b1] <= chunk_bytes)) n += h_olen[b1++];
    la.first = b0;
    la.nblocks = b1 - b0;
    la.chunk_start = starts[b0];
    if (n) {
      LZ4MI_HIP(c, hipMemsetAsync(d_dist, 0, (size_t)link_dist_words(n) * 4, c->stream));
      LZ4MI_HIP(c, hipMemsetAsync(d_flags, 0, (size_t)link_tiles(n) + 16, c->stream));
      launch_link_walk(la, true, c->stream);
      LZ4MI_HIP(c, hipGetLastError());
      // the second walk checks the distances against the block's place in the frame
      LZ4MI_HIP(c, hipMemcpyAsync(h_stat + b0, la.status + b0, (size_t)(b1 - b0) * 4, hipMemcpyDeviceToHost, c->stream));
      bool first_round = true;
      for (;;) {
        LZ4MI_HIP(c, hipMemsetAsync(d_moved, 0, 4, c->stream));
        launch_link_jump(d_dist, n, d_done, d_moved, c->stream);
        launch_link_jump(d_dist, n, d_done, d_moved, c->stream);
        LZ4MI_HIP(c, hipGetLastError());
        LZ4MI_HIP(c, hipMemcpyAsync(h_word, d_moved, 4, hipMemcpyDeviceToHost, c->stream));
        LZ4MI_HIP(c, hipStreamSynchronize(c->stream));
        if (first_round) {
          for (uint32_t b = b0; b < b1; b++)
            if (h_stat[b] != 0) return LZ4MI_OK;
          first_round = false;
        }
        if (!*h_word) break;
      }
      launch_link_gather(d_dist, n, c->out.as<uint8_t>(), la.chunk_start, d_dict, la.dict_len, c->stream);
      LZ4MI_HIP(c, hipGetLastError());
    }
    b0 = b1;
  }
  *resolved = true;
  return LZ4MI_OK;
}

// Decodes the collected blocks on the device and folds the per-block results in stream
// order.  `region` = host bytes that contain every payload, uploaded once.
// walk_err: structural error met while reading block blocks.size() (0 = none).
//
// Device memory follows the data, not blocks x blockMaxSize: a block's slot is what its payload can
// decode to at most (decoded_bound), the slots of one launch stay within decode_scratch_budget() and
// within what the output limits still allow plus one block, and a frame that takes several launches, or
// whose blocks do not fill their slots, is compacted launch by launch into one buffer sized by those
// same limits.
int32_t decode_blocks(lz4mi_ctx *c, const uint8_t *region, size_t region_len, const std::vector<Blk> &blocks,
                      size_t slot, bool indep, const uint8_t *dict, size_t dict_len, bool have_dict,
                      int32_t walk_err, int64_t max_out, bool want_content_hash, bool legacy,
                      bool legacy_last_has_next, uint8_t *dst,
                      size_t cap, size_t *out_len, uint32_t *content_hash, DecodeExtras *extras) {
  const uint32_t nb = (uint32_t)blocks.size();
  *out_len = 0;
  if (nb == 0) {
    if (walk_err) return walk_err;
    if (want_content_hash) *content_hash = xxh32_host(nullptr, 0);
    return LZ4MI_OK;
  }
  LZ4MI_HIP(c, c->in.ensure(region_len + 16));
  {
    const int32_t up = host_copy(c, c->in.p, region, region_len, hipMemcpyHostToDevice);
    if (up) return up;
  }
  const uint8_t *d_dict = nullptr;
  if (have_dict && dict_len) {
    if (dict_len > 65536) {  // the decoder keeps the last 65536 bytes (lz4_frame_decoder.dart:278-282)
      dict += dict_len - 65536;
      dict_len = 65536;
    }
    LZ4MI_HIP(c, c->dictbuf.ensure(dict_len));
    LZ4MI_HIP(c, hipMemcpyAsync(c->dictbuf.p, dict, dict_len, hipMemcpyHostToDevice, c->stream));
    d_dict = c->dictbuf.as<uint8_t>();
  }
  // what the caller's buffer and the output limit still allow; the block that crosses it may
  // overshoot by one slot before the fold below reports it
  uint64_t limit_total = cap;
  if (max_out >= 0 && (uint64_t)max_out < limit_total) limit_total = (uint64_t)max_out;
  uint64_t plan_total = 0;
  for (uint32_t b = 0; b < nb; b++) plan_total += planned_output(blocks[b], slot);
  const uint64_t final_cap = plan_total < limit_total + slot ? plan_total : limit_total + slot;

  // per-block arrays, host (pinned) and device, same layout:
  // fixed:    [payload_off u64][payload_len u32][raw u32][dec_len u32][cap u32][stored_len u32]
  // per launch: [dst_off u64][copy_src u64][copy_dst u64][copy_len u32]
  // results:  [out_len u32][status i32][hash u32]
  const size_t fixed_bytes = (size_t)nb * 28, hb = (size_t)nb * 72;  // 68 bytes per block, 8-byte aligned arrays
  int32_t st = pin_reserve(c, hb + 64);
  if (st) return st;
  uint64_t *h_poff = (uint64_t *)c->h_pin;
  uint32_t *h_plen = (uint32_t *)(h_poff + nb);
  uint32_t *h_raw = h_plen + nb;
  uint32_t *h_dlen = h_raw + nb;
  uint32_t *h_cap = h_dlen + nb;
  uint32_t *h_slen = h_cap + nb;
  uint64_t *h_doff = (uint64_t *)(((uintptr_t)(h_slen + nb) + 7) & ~(uintptr_t)7);
  uint64_t *h_csrc = h_doff + nb;
  uint64_t *h_cdst = h_csrc + nb;
  uint32_t *h_clen = (uint32_t *)(h_cdst + nb);
  uint32_t *h_olen = h_clen + nb;
  int32_t *h_stat = (int32_t *)(h_olen + nb);
  uint32_t *h_hash = (uint32_t *)(h_stat + nb);
  LZ4MI_HIP(c, c->aux.ensure(hb + 64));
  uint64_t *d_poff = c->aux.as<uint64_t>();
  uint32_t *d_plen = (uint32_t *)(d_poff + nb);
  uint32_t *d_raw = d_plen + nb;
  uint32_t *d_dlen = d_raw + nb;
  uint32_t *d_cap = d_dlen + nb;
  uint32_t *d_slen = d_cap + nb;
  uint64_t *d_doff = (uint64_t *)(((uintptr_t)(d_slen + nb) + 7) & ~(uintptr_t)7);
  uint64_t *d_csrc = d_doff + nb;
  uint64_t *d_cdst = d_csrc + nb;
  uint32_t *d_clen = (uint32_t *)(d_cdst + nb);
  uint32_t *d_olen = d_clen + nb;
  int32_t *d_stat = (int32_t *)(d_olen + nb);
  uint32_t *d_hash = (uint32_t *)(d_stat + nb);

  bool any_ck = false;
  for (uint32_t b = 0; b < nb; b++) {
    h_poff[b] = blocks[b].payload_off;
    h_plen[b] = blocks[b].size;
    h_raw[b] = blocks[b].raw;
    h_dlen[b] = (blocks[b].raw && indep) ? 0u : blocks[b].size;  // a stored block decodes to nothing: it is copied
    h_cap[b] = (uint32_t)decoded_bound(blocks[b].size, slot);    // tmp ByteWriter(maxLength: blockMaxSize) :292
    h_slen[b] = (blocks[b].raw && indep) ? blocks[b].size : 0u;    // copied by the decode launch
    any_ck |= blocks[b].has_ck != 0;
  }
  LZ4MI_HIP(c, hipMemcpyAsync(d_poff, h_poff, fixed_bytes, hipMemcpyHostToDevice, c->stream));
  LZ4MI_HIP(c, hipMemsetAsync(d_olen, 0, (size_t)nb * 8, c->stream));
  if (any_ck) {  // block checksum of the payload as stored (:257-266)
    BatchArgs h;
    memset(&h, 0, sizeof h);
    h.src = c->in.as<uint8_t>();
    h.src_off = d_poff;
    h.src_len = d_plen;
    h.nblocks = nb;
    launch_xxh32_batch(h, d_hash, c->stream);
  }

  uint64_t produced = 0;
  int32_t err = LZ4MI_OK;
  uint32_t good = 0;
  const uint8_t *d_frame_out = nullptr;

  // one block's results in stream order: the first failing block decides
  auto fold = [&](uint32_t b) -> bool {
    const Blk &k = blocks[b];
    if (k.has_ck && h_hash[b] != k.ck) { err = LZ4MI_CORRUPT_BLOCK_CK; return false; }
    const uint32_t dsize = (k.raw && indep) ? k.size : h_olen[b];
    if (!(k.raw && indep) && h_stat[b] != LZ4MI_OK) { err = h_stat[b]; return false; }
    if (!indep && !k.raw && dsize > slot) { err = LZ4MI_CORRUPT_BLOCK_SIZE; return false; }  // :376-378
    if (max_out >= 0 && produced + dsize > (uint64_t)max_out) { err = LZ4MI_LIMIT; return false; }
    if (produced + dsize > cap) { err = LZ4MI_DST_TOO_SMALL; return false; }
    h_olen[b] = dsize;
    produced += dsize;
    good = b + 1;
    if (legacy && (b + 1 < nb || legacy_last_has_next) && dsize != slot) {  // :128-132
      err = LZ4MI_CORRUPT_LEGACY_NOT_FULL;
      return false;
    }
    return true;
  };

  bool resolved = false;
  if (!indep && nb >= kLinkMinBlocks && link_resolve_mode() != 0) {
    // linked blocks, every block at once (decode_linked_kernel.hip); anything but a well-formed frame that fits
    // the limits is left to the chain below
    LinkArgs la;
    memset(&la, 0, sizeof la);
    la.src = c->in.as<uint8_t>();
    la.payload_off = d_poff;
    la.payload_len = d_plen;
    la.raw = d_raw;
    la.block_max = (uint32_t)slot;
    la.dict_len = (uint32_t)(d_dict ? dict_len : 0);
    la.out_len = d_olen;
    la.status = d_stat;
    st = link_resolve(c, la, nb, d_dict, limit_total, h_olen, h_stat, (uint32_t *)((uint8_t *)c->h_pin + hb + 32), &resolved);
    if (st) return st;
    if (resolved) {
      for (uint32_t b = 0; b < nb; b++)
        if (!fold(b)) break;
      d_frame_out = c->out.as<uint8_t>();
    } else {
      if (link_resolve_mode() == 2) {
        c->err = "linked frame left to the chain";
        return LZ4MI_BACKEND;
      }
      LZ4MI_HIP(c, hipMemsetAsync(d_olen, 0, (size_t)nb * 8, c->stream));
    }
  }
  if (resolved) {
  } else if (!indep) {
    // linked blocks: one wavefront walks the chain over one contiguous output
    LZ4MI_HIP(c, c->out.ensure((size_t)final_cap + 16));
    ChainArgs ch;
    ch.src = c->in.as<uint8_t>();
    ch.payload_off = d_poff;
    ch.payload_len = d_plen;
    ch.raw = d_raw;
    ch.nblocks = nb;
    ch.block_max = (uint32_t)slot;
    ch.dst = c->out.as<uint8_t>();
    ch.dst_limit = limit_total;  // the buffer holds one more slot than this
    ch.dict = d_dict;
    ch.dict_len = (uint32_t)(d_dict ? dict_len : 0);
    ch.out_len = d_olen;
    ch.status = d_stat;
    launch_decode_chain(ch, c->stream);
    LZ4MI_HIP(c, hipGetLastError());
    LZ4MI_HIP(c, hipMemcpyAsync(h_olen, d_olen, (size_t)nb * 12, hipMemcpyDeviceToHost, c->stream));
    LZ4MI_HIP(c, hipStreamSynchronize(c->stream));
    for (uint32_t b = 0; b < nb; b++)
      if (!fold(b)) break;
    d_frame_out = c->out.as<uint8_t>();
  } else {
    const size_t budget = decode_scratch_budget();
    bool gathered = false;  // earlier launches' output sits in c->comp
    uint32_t b0 = 0;
    while (b0 < nb && !err) {
      // ---- the blocks of this launch, their slots packed at 16-byte steps
      const uint64_t left = limit_total > produced ? limit_total - produced : 0;
      const uint64_t room = left + slot < budget ? left + slot : budget;
      uint64_t run = 0;
      uint32_t b1 = b0;
      while (b1 < nb) {
        const uint64_t need = round_up((size_t)planned_output(blocks[b1], slot), 16);
        if (b1 > b0 && run + need > room) break;
        h_doff[b1] = run;
        run += need;
        b1++;
      }
      const uint32_t n = b1 - b0;
      LZ4MI_HIP(c, c->out.ensure((size_t)run + 16));
      uint8_t *d_slots = c->out.as<uint8_t>();
      LZ4MI_HIP(c, hipMemcpyAsync(d_doff + b0, h_doff + b0, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
      BatchArgs a;
      memset(&a, 0, sizeof a);
      a.src = c->in.as<uint8_t>();
      a.src_off = d_poff + b0;
      a.src_len = d_dlen + b0;
      a.dst = d_slots;
      a.dst_off = d_doff + b0;
      a.dst_cap = d_cap + b0;
      a.dict = d_dict;
      a.dict_len = (uint32_t)(d_dict ? dict_len : 0);
      a.nblocks = n;
      a.out_len = d_olen + b0;
      a.status = d_stat + b0;
      a.stored_len = d_slen + b0;  // stored blocks are plain copies into their slots, in the same launch
      st = run_decompress(c, a, false);
      if (st) return st;
      LZ4MI_HIP(c, hipGetLastError());
      LZ4MI_HIP(c, hipMemcpyAsync(h_olen + b0, d_olen + b0, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
      LZ4MI_HIP(c, hipMemcpyAsync(h_stat + b0, d_stat + b0, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
      if (any_ck) LZ4MI_HIP(c, hipMemcpyAsync(h_hash + b0, d_hash + b0, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
      LZ4MI_HIP(c, hipStreamSynchronize(c->stream));

      // ---- fold, and see whether the good blocks already lie back to back in their slots
      const uint64_t before = produced;
      bool packed = true;
      uint32_t g1 = b0;
      for (uint32_t b = b0; b < b1; b++) {
        if (produced - before != h_doff[b]) packed = false;
        const bool ok = fold(b);
        if (good == b + 1) g1 = b + 1;
        if (!ok) break;
      }
      if (b0 == 0 && !gathered && packed && (b1 == nb || err)) {
        d_frame_out = d_slots;  // the whole output, in place
      } else if (g1 > b0) {
        if (!gathered) {
          LZ4MI_HIP(c, c->comp.ensure((size_t)final_cap + 16));
          gathered = true;
        }
        for (uint32_t b = b0; b < g1; b++) {
          h_csrc[b - b0] = (uint64_t)(uintptr_t)(d_slots + h_doff[b]);
          h_cdst[b - b0] = b == b0 ? before : h_cdst[b - b0 - 1] + h_olen[b - 1];
          h_clen[b - b0] = h_olen[b];
        }
        const uint32_t m = g1 - b0;
        LZ4MI_HIP(c, hipMemcpyAsync(d_csrc, h_csrc, (size_t)m * 8, hipMemcpyHostToDevice, c->stream));
        LZ4MI_HIP(c, hipMemcpyAsync(d_cdst, h_cdst, (size_t)m * 8, hipMemcpyHostToDevice, c->stream));
        LZ4MI_HIP(c, hipMemcpyAsync(d_clen, h_clen, (size_t)m * 4, hipMemcpyHostToDevice, c->stream));
        launch_seg_copy(d_csrc, c->comp.as<uint8_t>(), d_cdst, 0, d_clen, m, c->stream);
        LZ4MI_HIP(c, hipGetLastError());
        LZ4MI_HIP(c, hipStreamSynchronize(c->stream));  // the lists and the slots are reused by the next launch
        d_frame_out = c->comp.as<uint8_t>();
      }
      b0 = b1;
    }
    if (!d_frame_out) d_frame_out = c->out.as<uint8_t>();
  }

  if (!err && walk_err) err = walk_err;
  if (err && extras && extras->partial) {  // deliver the blocks in front of the failing one
    extras->deferred = err;
    err = LZ4MI_OK;
  }
  if (err) return err;
  if (extras) {
    extras->good_blocks = good;
    extras->sizes.assign(h_olen, h_olen + good);
  }
  if (want_content_hash) {
    // xxh32(frameOut) :306-315.  The output goes to host memory anyway: it is downloaded in pieces and a host
    // thread hashes piece k while piece k + 1 is on its way (one serial chain of rounds; a CPU core runs it
    // several times faster than one GPU wavefront, and beside the copy instead of in front of it).
    lz4mi_xxh32_state hs;
    lz4mi_xxh32_state_init(&hs, 0);
    std::future<void> hashing;
    const size_t piece = 32u << 20;
    hipError_t e = hipSuccess;
    for (size_t off = 0; off < produced && e == hipSuccess; off += piece) {
      const size_t m = produced - off < piece ? (size_t)(produced - off) : piece;
      const int32_t down = host_copy(c, dst + off, d_frame_out + off, m, hipMemcpyDeviceToHost);
      if (hashing.valid()) hashing.get();
      if (down) return down;
      if (e == hipSuccess) {
        lz4mi_xxh32_state *hp = &hs;
        const uint8_t *q = dst + off;
        hashing = run_beside([hp, q, m] { xxh_state_update_host(hp, q, m); });
      }
    }
    if (hashing.valid()) hashing.get();
    LZ4MI_HIP(c, e);
    *content_hash = lz4mi_xxh32_state_digest(&hs);
  } else {
    const int32_t down = host_copy(c, dst, d_frame_out, produced, hipMemcpyDeviceToHost);
    if (down) return down;
    LZ4MI_HIP(c, hipStreamSynchronize(c->stream));
  }
  if (extras) extras->d_out = d_frame_out;
  *out_len = produced;
  return LZ4MI_OK;
}

}  // namespace lz4mi

extern "C" int32_t lz4mi_frame_decode_next(lz4mi_ctx *c, const uint8_t *src, size_t len, size_t *consumed,
                                           const uint8_t *dict, size_t dict_len, int32_t dict_state,
                                           int32_t have_prev_dict, int64_t max_out, uint8_t *dst, size_t cap,
                                           size_t *out_len, uint32_t *dict_id_needed) {
  if (!c || !consumed || !out_len) return LZ4MI_RANGE;
  *consumed = 0;
  *out_len = 0;
  LZ4MI_HIP(c, ::lz4mi::enter_device(c));
  if (len < 4) return LZ4MI_FORMAT_EOF;  // lz4_frame_decoder.dart:53-55
  const uint32_t magic = rd32(src);
  size_t off = 4;
  if ((magic & 0xFFFFFFF0u) == kSkippableBase) {  // :59-62, :78-84
    if (len - off < 4) return LZ4MI_FORMAT_EOF;
    uint32_t size = rd32(src + off);
    off += 4;
    if (size > len - off) return LZ4MI_FORMAT_EOF;
    *consumed = off + size;
    return LZ4MI_OK;
  }

  if (magic == kLegacyMagic) {  // :86-136
    const size_t legacy_max = 8u << 20;
    std::vector<Blk> blocks;
    int32_t walk_err = 0;
    bool last_has_next = false;  // a non-boundary word follows the last collected block
    auto boundary = [](uint32_t m) {
      return m == kFrameMagic || m == kLegacyMagic || (m & 0xFFFFFFF0u) == kSkippableBase;
    };
    for (;;) {
      if (len - off == 0) break;
      if (len - off < 4) { walk_err = LZ4MI_FORMAT_EOF; break; }
      uint32_t w = rd32(src + off);
      if (boundary(w)) break;
      off += 4;
      if (w == 0) { walk_err = LZ4MI_CORRUPT_LEGACY_SIZE; last_has_next = true; break; }
      if (w > len - off) { walk_err = LZ4MI_FORMAT_EOF; last_has_next = true; break; }
      Blk k = {off, w, 0, 0, 0};
      blocks.push_back(k);
      off += w;
      if (len - off != 0 && len - off < 4) {  // :124-126, raised after this block has been decoded
        walk_err = LZ4MI_FORMAT_EOF;
        break;
      }
    }
    uint32_t unused = 0;
    int32_t st = decode_blocks(c, src, off, blocks, legacy_max, true, nullptr, 0, false, walk_err, max_out, false,
                               true, last_has_next, dst, cap, out_len, &unused);
    if (st) return st;
    *consumed = off;
    return LZ4MI_OK;
  }
  if (magic != kFrameMagic) return LZ4MI_FORMAT_MAGIC;  // :70-72

  Header h;
  int32_t st = parse_header(src + off, len - off, &h);
  if (st) return st;
  bool have_dict = have_prev_dict != 0;  // _dict persists across frames (:34,235)
  if (h.has_dict) {                      // :225-236
    if (dict_state == 0) {
      if (dict_id_needed) *dict_id_needed = h.dict_id;
      return LZ4MI_NEED_DICT;
    }
    if (dict_state == 2) return LZ4MI_UNSUPPORTED_DICT;
    if (dict_state == 3) {
      if (dict_id_needed) *dict_id_needed = h.dict_id;
      return LZ4MI_DICT_NOT_FOUND;
    }
    have_dict = true;
  }
  if (!have_dict) { dict = nullptr; dict_len = 0; }
  off += h.len;

  std::vector<Blk> blocks;
  int32_t walk_err = 0;
  bool ended = false;
  for (;;) {  // :238-304, structure only
    if (len - off < 4) { walk_err = LZ4MI_FORMAT_EOF; break; }
    uint32_t raw = rd32(src + off);
    off += 4;
    if (raw == 0) { ended = true; break; }
    uint32_t size = raw & 0x7FFFFFFFu;
    if (size > h.bmax) { walk_err = LZ4MI_CORRUPT_BLOCK_SIZE; break; }
    if (size > len - off) { walk_err = LZ4MI_FORMAT_EOF; break; }
    Blk k = {off, size, (raw >> 31) & 1u, 0, 0};
    off += size;
    if (h.block_ck) {
      if (len - off < 4) { walk_err = LZ4MI_FORMAT_EOF; break; }
      k.has_ck = 1;
      k.ck = rd32(src + off);
      off += 4;
    }
    blocks.push_back(k);
  }
  uint32_t expected_content = 0;
  int32_t tail_err = 0;
  if (ended && h.content_ck) {  // :306-310
    if (len - off < 4) tail_err = LZ4MI_FORMAT_EOF;
    else { expected_content = rd32(src + off); off += 4; }
  }
  uint32_t content_hash = 0;
  st = decode_blocks(c, src, off, blocks, h.bmax, h.indep != 0, dict, dict_len, have_dict, walk_err, max_out,
                     ended && h.content_ck && !tail_err, false, false, dst, cap, out_len, &content_hash);
  if (st) return st;
  if (tail_err) return tail_err;
  if (h.content_ck && content_hash != expected_content) return LZ4MI_CORRUPT_CONTENT_CK;  // :311-314
  if (h.has_size && (uint64_t)*out_len != h.content_size) return LZ4MI_CORRUPT_CONTENT_SIZE;  // :317-321
  *consumed = off;
  return LZ4MI_OK;
}

// ------------------------------------------------- sharded decode (device resident)

extern "C" int32_t lz4mi_frame_block_table(const uint8_t *src, size_t len, lz4mi_frame_info *info,
                                           uint64_t *record_off, size_t max_records, size_t *nrecords,
                                           size_t *end_off) {
  if (!src || !nrecords) return LZ4MI_RANGE;
  *nrecords = 0;
  if (end_off) *end_off = 0;
  if (len < 4) return LZ4MI_FORMAT_EOF;
  if (rd32(src) != kFrameMagic) return LZ4MI_FORMAT_MAGIC;
  Header h;
  int32_t st = parse_header(src + 4, len - 4, &h);
  if (st) return st;
  if (info) {
    memset(info, 0, sizeof *info);
    info->magic = kFrameMagic;
    info->block_independence = h.indep;
    info->block_checksum = h.block_ck;
    info->content_checksum = h.content_ck;
    info->has_content_size = h.has_size;
    info->has_dict_id = h.has_dict;
    info->dict_id = h.dict_id;
    info->block_max_size = (uint32_t)h.bmax;
    info->header_size = (uint32_t)(4 + h.len);
    info->content_size = h.content_size;
  }
  size_t off = 4 + h.len, n = 0;
  for (;;) {  // lz4_frame_decoder.dart:243-255, structure only
    if (len - off < 4) return LZ4MI_FORMAT_EOF;
    const uint32_t word = rd32(src + off);
    if (word == 0) { off += 4; break; }
    const uint32_t size = word & 0x7FFFFFFFu;
    if (size > h.bmax) return LZ4MI_CORRUPT_BLOCK_SIZE;
    const size_t rec = 4 + (size_t)size + (h.block_ck ? 4 : 0);
    if (rec > len - off) return LZ4MI_FORMAT_EOF;
    if (record_off && n < max_records) record_off[n] = off;
    n++;
    off += rec;
  }
  if (h.content_ck) {
    if (len - off < 4) return LZ4MI_FORMAT_EOF;
    off += 4;
  }
  *nrecords = n;
  if (end_off) *end_off = off;
  return record_off && n > max_records ? LZ4MI_DST_TOO_SMALL : LZ4MI_OK;
}

extern "C" int32_t lz4mi_frame_shard_decode(lz4mi_ctx *c, const uint8_t *d_records, uint64_t records_len,
                                            const uint64_t *d_record_off, uint32_t nblocks,
                                            const lz4mi_frame_options *o, const uint8_t *d_dict, uint32_t dict_len,
                                            uint8_t *d_dst, uint32_t *d_out_len, int32_t *d_status) {
  if (!c || !o || !d_out_len || !d_status) return LZ4MI_RANGE;
  const size_t bmax = block_max_bytes(o->block_size_id);
  if (!bmax) return LZ4MI_RANGE;
  if (!o->block_independence) return LZ4MI_UNSUPPORTED_HERE;  // a linked frame is one serial chain
  LZ4MI_HIP(c, ::lz4mi::enter_device(c));
  if (nblocks == 0) return LZ4MI_OK;
  if (!d_records || !d_record_off || !d_dst) return LZ4MI_RANGE;
  LZ4MI_HIP(c, c->payload_addr.ensure((size_t)nblocks * 8));
  LZ4MI_HIP(c, c->payload_len.ensure((size_t)nblocks * 4));
  LZ4MI_HIP(c, c->comp_len.ensure((size_t)nblocks * 4));
  LZ4MI_HIP(c, c->record_size.ensure((size_t)nblocks * 4));
  LZ4MI_HIP(c, c->rec_off.ensure((size_t)nblocks * 8));
  LZ4MI_HIP(c, c->stored_word.ensure((size_t)nblocks * 4));
  LZ4MI_HIP(c, c->status.ensure((size_t)nblocks * 4));
  LZ4MI_HIP(c, c->hashes.ensure((size_t)nblocks * 4));
  RecordReadArgs r;
  r.records = d_records;
  r.records_len = records_len;
  r.rec_off = d_record_off;
  r.nblocks = nblocks;
  r.block_max = (uint32_t)bmax;
  r.block_checksum = o->block_checksum;
  r.payload_addr = c->payload_addr.as<uint64_t>();
  r.payload_len = c->payload_len.as<uint32_t>();
  r.dec_len = c->comp_len.as<uint32_t>();
  r.raw_len = c->record_size.as<uint32_t>();
  r.slot_off = c->rec_off.as<uint64_t>();
  r.expect = c->stored_word.as<uint32_t>();
  r.pre_status = c->status.as<int32_t>();
  launch_record_read(r, c->stream);
  BatchArgs a;
  memset(&a, 0, sizeof a);
  a.src = nullptr;  // absolute addresses
  a.src_off = r.payload_addr;
  a.src_len = r.dec_len;
  a.dst = d_dst;
  a.dst_stride = bmax;
  a.dst_cap_uniform = (uint32_t)bmax;  // tmp ByteWriter(maxLength: blockMaxSize) lz4_frame_decoder.dart:292
  a.dict = d_dict;
  a.dict_len = d_dict ? dict_len : 0;
  a.nblocks = nblocks;
  a.out_len = d_out_len;
  a.status = d_status;
  a.stored_len = r.raw_len;  // stored blocks (:268-271) are copied by the same launch
  int32_t st = run_decompress(c, a, false);
  if (st) return st;
  if (o->block_checksum) {
    BatchArgs h;
    memset(&h, 0, sizeof h);
    h.src = nullptr;
    h.src_off = r.payload_addr;
    h.src_len = r.payload_len;
    h.nblocks = nblocks;
    launch_xxh32_batch(h, c->hashes.as<uint32_t>(), c->stream);
  }
  launch_record_finish(r, c->hashes.as<uint32_t>(), d_out_len, d_status, c->stream);
  LZ4MI_HIP(c, hipGetLastError());
  return LZ4MI_OK;
}

extern "C" int32_t lz4mi_blocks_gather(lz4mi_ctx *c, const uint8_t *d_slots, uint64_t slot_stride,
                                       const uint32_t *d_len, const uint64_t *d_dst_off, uint32_t nblocks,
                                       uint8_t *d_dst, uint64_t cap) {
  if (!c) return LZ4MI_RANGE;
  LZ4MI_HIP(c, ::lz4mi::enter_device(c));
  if (nblocks == 0) return LZ4MI_OK;
  if (!d_slots || !d_len || !d_dst_off || !d_dst) return LZ4MI_RANGE;
  int32_t st = pin_reserve(c, 64);
  if (st) return st;
  LZ4MI_HIP(c, c->total.ensure(64));
  LZ4MI_HIP(c, c->payload_addr.ensure((size_t)nblocks * 8));
  uint32_t *flag = c->total.as<uint32_t>();
  LZ4MI_HIP(c, hipMemsetAsync(flag, 0, 4, c->stream));
  launch_record_bounds(d_dst_off, d_len, nblocks, cap, flag, c->stream);
  LZ4MI_HIP(c, hipMemcpyAsync(c->h_pin, flag, 4, hipMemcpyDeviceToHost, c->stream));
  launch_slot_addr(d_slots, slot_stride, nblocks, c->payload_addr.as<uint64_t>(), c->stream);
  LZ4MI_HIP(c, hipGetLastError());
  LZ4MI_HIP(c, hipStreamSynchronize(c->stream));
  if (*(uint32_t *)c->h_pin) return LZ4MI_DST_TOO_SMALL;  // nothing written
  launch_seg_copy(c->payload_addr.as<uint64_t>(), d_dst, d_dst_off, 0, d_len, nblocks, c->stream);
  LZ4MI_HIP(c, hipGetLastError());
  return LZ4MI_OK;
}

extern "C" int32_t lz4mi_blocks_conformance(lz4mi_ctx *c, const lz4mi_batch *b, uint32_t *d_flags) {
  if (!c || !b || !d_flags) return LZ4MI_RANGE;
  LZ4MI_HIP(c, ::lz4mi::enter_device(c));
  BatchArgs a;
  memset(&a, 0, sizeof a);
  a.src = b->d_src;
  a.src_off = b->d_src_off;
  a.src_len = b->d_src_len;
  a.src_stride = b->src_stride;
  a.src_len_uniform = b->src_len_uniform;
  a.src_total = b->src_total;
  a.nblocks = b->nblocks;
  launch_block_conformance(a, d_flags, c->stream);
  LZ4MI_HIP(c, hipGetLastError());
  return LZ4MI_OK;
}

extern "C" int32_t lz4mi_frame_decode_bound(const uint8_t *src, size_t len, size_t *bound_out) {
  if (!bound_out) return LZ4MI_RANGE;
  *bound_out = 0;
  if (len < 4) return LZ4MI_OK;
  const uint32_t magic = rd32(src);
  size_t off = 4;
  if ((magic & 0xFFFFFFF0u) == kSkippableBase) return LZ4MI_OK;
  size_t bmax = 8u << 20;
  bool ck = false;
  if (magic == kFrameMagic) {
    Header h;
    if (parse_header(src + off, len - off, &h)) return LZ4MI_OK;
    off += h.len;
    bmax = h.bmax;
    ck = h.block_ck != 0;
  } else if (magic != kLegacyMagic) {
    return LZ4MI_OK;
  }
  uint64_t total = 0;
  while (len - off >= 4) {
    uint32_t raw = rd32(src + off);
    if (magic == kLegacyMagic &&
        (raw == kFrameMagic || raw == kLegacyMagic || (raw & 0xFFFFFFF0u) == kSkippableBase))
      break;
    off += 4;
    if (raw == 0) break;
    uint32_t size = magic == kLegacyMagic ? raw : raw & 0x7FFFFFFFu;
    if (size > len - off) break;
    off += size + (ck ? 4 : 0);
    if (off > len) break;
    Blk k = {0, size, magic == kLegacyMagic ? 0u : (raw >> 31) & 1u, 0, 0};
    total += planned_output(k, bmax);
  }
  *bound_out = (size_t)total;
  return LZ4MI_OK;
}
