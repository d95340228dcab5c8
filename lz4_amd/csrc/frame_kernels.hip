// Frame assembly kernels: the bookkeeping half of the frame block dispatcher
// (lib/src/frame/lz4_frame_encoder.dart:209-258): stored-vs-compressed choice,
// record offsets by exclusive prefix sum, size/checksum words, payload gather.
#include "device_common.h"
#include "kernels.h"
#include "frame_kernels.h"

namespace lz4mi {

// Per block: stored length, raw flag, record size, payload source address.
// Stored compressed iff c < n, strictly (lz4_frame_encoder.dart:235).
__global__ void frame_plan_kernel(FramePlanArgs p) {
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= p.nblocks) return;
  uint64_t start = (uint64_t)b * p.block_max;
  uint32_t n = (uint32_t)(p.src_total - start < p.block_max ? p.src_total - start : p.block_max);
  uint32_t c = p.comp_len[b];
  bool raw = !(c < n);
  uint32_t stored = raw ? n : c;
  p.stored_word[b] = raw ? (0x80000000u | n) : c;
  p.payload_addr[b] = raw ? (uint64_t)(uintptr_t)(p.src + start)
                          : (uint64_t)(uintptr_t)(p.comp + (uint64_t)b * p.comp_stride);
  p.payload_len[b] = stored;
  p.record_size[b] = 4u + stored + (p.block_checksum ? 4u : 0u);
}

// Exclusive prefix sum of u32 sizes into u64 offsets (+ base); single workgroup.
// total[0] receives base + sum.
__global__ __launch_bounds__(1024) void scan_sizes_kernel(const uint32_t *sizes, uint32_t n, uint64_t base,
                                                          uint64_t *offsets, uint64_t *total) {
  __shared__ uint64_t part[1024];
  const uint32_t t = threadIdx.x;
  const uint32_t per = (n + 1023u) / 1024u;
  const uint32_t lo = t * per < n ? t * per : n;
  const uint32_t hi = lo + per < n ? lo + per : n;
  uint64_t s = 0;
  for (uint32_t k = lo; k < hi; k++) s += sizes[k];
  part[t] = s;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan
    uint64_t v = t >= d ? part[t - d] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  uint64_t run = base + (t ? part[t - 1] : 0);
  for (uint32_t k = lo; k < hi; k++) {
    offsets[k] = run;
    run += sizes[k];
  }
  if (t == 1023) total[0] = base + part[1023];
}

__device__ __forceinline__ void put_u32le(uint8_t *p, uint32_t v) { ((U32U *)p)->v = v; }

// Size word in front of each record and the optional xxh32 word behind the payload.
__global__ void frame_words_kernel(uint8_t *frame, const uint64_t *rec_off, const uint32_t *stored_word,
                                   const uint32_t *payload_len, const uint32_t *hashes, uint32_t nblocks) {
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblocks) return;
  uint8_t *r = frame + rec_off[b];
  put_u32le(r, stored_word[b]);
  if (hashes) put_u32le(r + 4 + payload_len[b], hashes[b]);
}

// One workgroup per segment: dst_base + dst_off[k] + dst_delta  <-  (uint8_t*)src_addr[k], len[k] bytes.
__global__ __launch_bounds__(256) void seg_copy_kernel(const uint64_t *src_addr, uint8_t *dst_base,
                                                       const uint64_t *dst_off, uint64_t dst_delta,
                                                       const uint32_t *len, uint32_t nseg) {
  for (uint32_t k = blockIdx.x; k < nseg; k += gridDim.x) {
    const uint8_t *src = (const uint8_t *)(uintptr_t)src_addr[k];
    uint8_t *dst = dst_base + dst_off[k] + dst_delta;
    uint32_t n = len[k];
    const uint32_t t = threadIdx.x;
    uint32_t head = (uint32_t)(-(uintptr_t)dst) & 15u;
    if (head > n) head = n;
    if (t < head) dst[t] = src[t];
    dst += head;
    src += head;
    n -= head;
    uint32_t body = n & ~15u;
    for (uint32_t off = t * 16; off < body; off += 256 * 16) st128(dst + off, ld128(src + off));
    uint32_t tail = n - body;
    if (t < tail) dst[body + t] = src[body + t];
  }
}

// flag[0] = 1 when a record would end behind `cap` (offsets are relative to the window).
__global__ void record_bounds_kernel(const uint64_t *rec_off, const uint32_t *record_size, uint32_t nblocks,
                                     uint64_t cap, uint32_t *flag) {
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblocks) return;
  const uint64_t o = rec_off[b];
  if (o > cap || cap - o < record_size[b]) flag[0] = 1u;
}

// Decode side of the dispatcher (lib/src/frame/lz4_frame_decoder.dart:243-271): reads each record's size
// word, checks it the way the reference does before touching the payload, and splits the blocks into
// "decode" (dec_len), "copy" (raw_len) and "failed before decoding" (pre_status).
__global__ void record_read_kernel(RecordReadArgs r) {
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= r.nblocks) return;
  const uint64_t off = r.rec_off[b];
  int32_t pre = LZ4MI_OK;
  uint32_t size = 0, raw = 0, expect = 0;
  if (off > r.records_len || r.records_len - off < 4) {
    pre = LZ4MI_FORMAT_EOF;  // :243-249
  } else {
    const uint32_t word = ld32(r.records + off);
    size = word & 0x7FFFFFFFu;
    raw = word >> 31;
    const uint64_t left = r.records_len - off - 4;
    if (size > r.block_max) pre = LZ4MI_CORRUPT_BLOCK_SIZE;                    // :251-253
    else if (size > left) pre = LZ4MI_FORMAT_EOF;                               // :255
    else if (r.block_checksum && left - size < 4) pre = LZ4MI_FORMAT_EOF;       // :258
    else if (r.block_checksum) expect = ld32(r.records + off + 4 + size);
  }
  const bool ok = pre == LZ4MI_OK;
  r.payload_addr[b] = (uint64_t)(uintptr_t)(r.records + (ok ? off + 4 : 0));
  r.payload_len[b] = ok ? size : 0u;
  r.dec_len[b] = ok && !raw ? size : 0u;
  r.raw_len[b] = ok && raw ? size : 0u;
  r.slot_off[b] = (uint64_t)b * r.block_max;
  r.expect[b] = expect;
  r.pre_status[b] = pre;
}

// Folds what came before the decode into the per-block results, in the reference's order: size and
// truncation checks, block checksum (:257-266), then the decoder's own status.
__global__ void record_finish_kernel(RecordReadArgs r, const uint32_t *hashes, uint32_t *out_len, int32_t *status) {
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= r.nblocks) return;
  const int32_t pre = r.pre_status[b];
  if (pre != LZ4MI_OK) {
    status[b] = pre;
    out_len[b] = 0;
  } else if (r.block_checksum && hashes[b] != r.expect[b]) {
    status[b] = LZ4MI_CORRUPT_BLOCK_CK;
    out_len[b] = 0;
  } else if (r.raw_len[b] || !r.dec_len[b]) {  // stored block (possibly empty): copied, not decoded
    status[b] = LZ4MI_OK;
    out_len[b] = r.raw_len[b];
  }
}

// Absolute source addresses of block slots for a gather.
__global__ void slot_addr_kernel(const uint8_t *slots, uint64_t stride, uint32_t nblocks, uint64_t *addr) {
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nblocks) addr[b] = (uint64_t)(uintptr_t)(slots + (uint64_t)b * stride);
}

// flags[b] = 1 when compressed block b breaks the LZ4 block format's end-of-block rules: the last five bytes
// are literals and the last match starts at least twelve bytes before the end of the block.  (The reference's
// encoders search to the last four bytes and let matches run to the end: lz4_block_encoder.dart:95,118.)
// One lane per block walks the token chain; the blocks are trusted only as far as their length.
__global__ void block_conformance_kernel(BatchArgs a, uint32_t *flags) {
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.nblocks) return;
  const uint8_t *p = a.src + (a.src_off ? a.src_off[b] : (uint64_t)b * a.src_stride);
  const uint32_t c = block_src_len(a, b);
  uint32_t ip = 0, nmatch = 0, trailing = 0;
  uint64_t op = 0, last_start = 0;
  while (ip < c) {
    const uint32_t tok = p[ip++];
    uint32_t ll = tok >> 4;
    if (ll == 15)
      while (ip < c) {
        const uint32_t x = p[ip++];
        ll += x;
        if (x != 255) break;
      }
    ip += ll;
    op += ll;
    trailing = ll;
    if (ip + 2 > c) break;  // last literals
    ip += 2;
    uint32_t ml = (tok & 15u) + 4u;
    if ((tok & 15u) == 15u)
      while (ip < c) {
        const uint32_t x = p[ip++];
        ml += x;
        if (x != 255) break;
      }
    last_start = op;
    op += ml;
    nmatch++;
    trailing = 0;
  }
  flags[b] = nmatch && (trailing < 5u || last_start + 12u > op) ? 1u : 0u;
}

__global__ void put_words_kernel(uint8_t *dst, const uint32_t *words, uint32_t nwords, uint32_t literal0,
                                 int use_literal0) {
  // trailer: end mark (literal 0) followed by device-computed words
  uint32_t t = threadIdx.x;
  if (use_literal0 && t == 0) put_u32le(dst, literal0);
  if (t < nwords) put_u32le(dst + (use_literal0 ? 4 : 0) + 4 * t, words[t]);
}

void launch_frame_plan(const FramePlanArgs &p, hipStream_t s) {
  if (!p.nblocks) return;
  hipLaunchKernelGGL(frame_plan_kernel, dim3((p.nblocks + 255) / 256), dim3(256), 0, s, p);
}
void launch_scan_sizes(const uint32_t *sizes, uint32_t n, uint64_t base, uint64_t *offsets, uint64_t *total,
                       hipStream_t s) {
  hipLaunchKernelGGL(scan_sizes_kernel, dim3(1), dim3(1024), 0, s, sizes, n, base, offsets, total);
}
void launch_frame_words(uint8_t *frame, const uint64_t *rec_off, const uint32_t *stored_word,
                        const uint32_t *payload_len, const uint32_t *hashes, uint32_t nblocks, hipStream_t s) {
  if (!nblocks) return;
  hipLaunchKernelGGL(frame_words_kernel, dim3((nblocks + 255) / 256), dim3(256), 0, s, frame, rec_off,
                     stored_word, payload_len, hashes, nblocks);
}
void launch_seg_copy(const uint64_t *src_addr, uint8_t *dst_base, const uint64_t *dst_off, uint64_t dst_delta,
                     const uint32_t *len, uint32_t nseg, hipStream_t s) {
  if (!nseg) return;
  uint32_t grid = nseg < 65536u ? nseg : 65536u;
  hipLaunchKernelGGL(seg_copy_kernel, dim3(grid), dim3(256), 0, s, src_addr, dst_base, dst_off, dst_delta, len,
                     nseg);
}
void launch_record_bounds(const uint64_t *rec_off, const uint32_t *record_size, uint32_t nblocks, uint64_t cap,
                          uint32_t *flag, hipStream_t s) {
  if (!nblocks) return;
  hipLaunchKernelGGL(record_bounds_kernel, dim3((nblocks + 255) / 256), dim3(256), 0, s, rec_off, record_size,
                     nblocks, cap, flag);
}
void launch_record_read(const RecordReadArgs &r, hipStream_t s) {
  if (!r.nblocks) return;
  hipLaunchKernelGGL(record_read_kernel, dim3((r.nblocks + 255) / 256), dim3(256), 0, s, r);
}
void launch_record_finish(const RecordReadArgs &r, const uint32_t *hashes, uint32_t *out_len, int32_t *status,
                          hipStream_t s) {
  if (!r.nblocks) return;
  hipLaunchKernelGGL(record_finish_kernel, dim3((r.nblocks + 255) / 256), dim3(256), 0, s, r, hashes, out_len,
                     status);
}
void launch_slot_addr(const uint8_t *slots, uint64_t stride, uint32_t nblocks, uint64_t *addr, hipStream_t s) {
  if (!nblocks) return;
  hipLaunchKernelGGL(slot_addr_kernel, dim3((nblocks + 255) / 256), dim3(256), 0, s, slots, stride, nblocks, addr);
}
void launch_block_conformance(const BatchArgs &a, uint32_t *flags, hipStream_t s) {
  if (!a.nblocks) return;
  hipLaunchKernelGGL(block_conformance_kernel, dim3((a.nblocks + 63) / 64), dim3(64), 0, s, a, flags);
}
void launch_put_words(uint8_t *dst, const uint32_t *words, uint32_t nwords, bool end_mark, hipStream_t s) {
  hipLaunchKernelGGL(put_words_kernel, dim3(1), dim3(64), 0, s, dst, words, nwords, 0u, end_mark ? 1 : 0);
}

}  // namespace lz4mi
