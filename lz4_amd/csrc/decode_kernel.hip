// LZ4 block decoder for gfx950: one or two wavefronts per block.
//
// Follows lz4BlockDecompress (strict) and lz4BlockDecompressInto (lenient) of
// lib/src/block/lz4_block_decoder.dart:7-119 with ByteWriter.copyMatch
// (lib/src/internal/byte_writer.dart:93-121) check for check, so the status of
// a malformed block equals the exception the reference raises.
//
// Two loops share a block.
//
// The batch loop takes up to 64 sequences at a time.  The compressed bytes sit in an LDS window;
// every lane parses the byte it stands on as if it were a token, and the chain of real token
// starts is followed with one cross-lane read per sequence.  After that lane j owns sequence j: a
// prefix sum places all sequences in the output, the reference's checks are evaluated for all of
// them at once, and the copies run out of and into an LDS buffer that mirrors the most recent
// output: literals lane by lane, matches one after the other with the whole wavefront (LDS
// operations of a wavefront execute in order, so a match sees everything written before it),
// matches older than the buffer four at a time out of memory.  The buffer is written to memory in
// aligned 16-byte pieces once per batch.  A sequence that does not fit this scheme (eight or more
// length-extension bytes, the last sequence of a block, anything that fails a check) ends the
// batch in front of it.
//
// The general loop (decode_run) takes such a sequence on its own, straight from and to memory,
// in the reference's order of checks, and is where every block ends: all status codes come from it.
//
// The batch is made of pieces (parse_batch, copy_literals, copy_far, copy_matches, stg_flush) that
// one wavefront runs in turn (decode_block) or two share (decode_block_pair: one parses the next
// batch and places its literals while the other copies the matches of the current one).
#include <atomic>
#include <cstdlib>

#include "device_common.h"
#include "kernels.h"
#include "profile.h"

namespace lz4mi {

#ifdef LZ4MI_PROFILE
__device__ unsigned long long g_dec_phase[2 * PH_SLOTS];  // parsing wavefront, copying wavefront
#endif
#if defined(LZ4MI_PROFILE) || defined(LZ4MI_SPANS)
#define LZ4MI_DEC_SPANS 1
__device__ unsigned long long g_dec_span[2 * 65536];  // per block: start and end (100 MHz ticks)
#endif

// 1 KiB of the compressed block in registers: lane l holds the dwords at v-space offsets
// base + 4*l + 256*k (k = 0..3), v-space being byte offsets from the 4-byte aligned address
// at or below src.  A refill is the only point where the parse waits on memory.  (Requesting
// the next window ahead of time was measured slower: with loads in flight the compiler guards
// later register writes with full drains of the output stores.)
struct InWindow {
  const uint8_t *al;   // 4-byte aligned base (src - mis)
  uint32_t mis;        // src - al, 0..3
  uint32_t vend;       // mis + src_len rounded up to 4
  uint32_t base;       // window start in v-space (multiple of 4), covers [base, base + 1024)
  uint32_t r0, r1, r2, r3;
};
constexpr uint32_t kInWin = 1024;

// true when the 64 bytes at v-space offset v are inside the window
__device__ __forceinline__ bool win_covers(const InWindow &w, uint32_t v) { return v - w.base <= kInWin - 64; }

__device__ __forceinline__ void win_seek(InWindow &w, uint32_t vpos, uint32_t lane) {
  w.base = vpos & ~3u;
  const uint32_t p = w.base + 4 * lane;
  w.r0 = p < w.vend ? *(const uint32_t *)(w.al + p) : 0u;
  w.r1 = p + 256 < w.vend ? *(const uint32_t *)(w.al + p + 256) : 0u;
  w.r2 = p + 512 < w.vend ? *(const uint32_t *)(w.al + p + 512) : 0u;
  w.r3 = p + 768 < w.vend ? *(const uint32_t *)(w.al + p + 768) : 0u;
  vm_quiesce();
}

// Byte at compressed offset pos + lane, straight from the register window.  Caller guarantees
// that the bytes it goes on to use lie inside the window.
__device__ __forceinline__ uint32_t win_lane_byte(const InWindow &w, uint32_t pos, uint32_t lane) {
  const uint32_t rel = pos + lane + w.mis - w.base;
  const uint32_t d = rel >> 2;
  const int sel = (int)((d & 63u) << 2);
  const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)w.r0);
  const uint32_t b = (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)w.r1);
  const uint32_t c = (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)w.r2);
  const uint32_t e = (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)w.r3);
  const uint32_t lo = (d & 64u) ? b : a, hi = (d & 64u) ? e : c;
  const uint32_t word = (d & 128u) ? hi : lo;
  return (word >> ((rel & 3u) * 8u)) & 0xffu;
}

// The parse reads single bytes (token, length extensions, offset).  One cross-lane gather puts
// the 64 compressed bytes that follow a position into one register, a byte per lane; after that
// every byte of the sequence header costs one v_readlane, and a literal run that fits is stored
// straight from that register.
struct Stream {
  uint32_t sp;  // compressed offset of lane 0
  uint32_t S;   // lane l: byte at sp + l (meaningless at or past the end of the block)
};

__device__ __forceinline__ void stream_load(InWindow &w, Stream &s, uint32_t pos, uint32_t lane) {
  const uint32_t v = pos + w.mis;
  if (!win_covers(w, v)) win_seek(w, v, lane);
  s.sp = pos;
  s.S = win_lane_byte(w, pos, lane);
}

// byte at real offset `pos` of the compressed block (caller guarantees pos < src_len)
__device__ __forceinline__ uint32_t sbyte(InWindow &w, Stream &s, uint32_t pos, uint32_t lane) {
  if (pos - s.sp >= kWave) stream_load(w, s, pos, lane);
  return rdlane(s.S, pos - s.sp);
}

// Sum of length-extension bytes (lz4_block_decoder.dart:121-130): every byte is added, a byte
// other than 255 ends the run.  Returns false on running off the end of the input ('Unexpected
// end of input', byte_reader.dart:102-106).  One vote finds the end of the run among the
// streamed bytes.
__device__ __forceinline__ bool read_ext(InWindow &w, Stream &s, uint32_t &ip, uint32_t in_end, uint64_t &len,
                                         uint32_t lane) {
  for (;;) {
    if (ip >= in_end) return false;
    if (ip - s.sp >= kWave) stream_load(w, s, ip, lane);
    const uint32_t rel = ip - s.sp;
    const uint64_t m = ballot(lane >= rel && (s.sp + lane >= in_end || s.S != 255u));
    if (m == 0) {
      len += 255u * (kWave - rel);
      ip += kWave - rel;
      continue;
    }
    const uint32_t f = (uint32_t)__builtin_ctzll(m);
    len += 255u * (f - rel);
    if (s.sp + f >= in_end) return false;
    len += rdlane(s.S, f);
    ip = s.sp + f + 1;
    return true;
  }
}

// A long match with a short distance (a run of one byte, a short period repeated for kilobytes) is a pattern fill:
// output byte k is byte k mod dist of the period in front of the match (byte_writer.dart:116-118).  The period goes
// into LDS once, repeated far enough that any 16 bytes of the pattern lie in one piece, and every lane then builds
// its 16-byte stores from there: stores only, no store of the match waits for a load of what it wrote before.
// `pat` holds kPatPeriod + 32 bytes; the period [src_period, src_period + dist) is in memory.
constexpr uint32_t kPatPeriod = 2048;  // longest distance filled this way
constexpr uint32_t kPatMin = 1024;     // shortest match worth the set-up
__device__ __forceinline__ void pattern_fill(uint8_t *dst, const uint8_t *src_period, uint32_t dist, uint32_t ml,
                                             lds_u8 *pat, uint32_t lane) {
  // pattern bytes [0, dist + 20): byte j is period byte j mod dist
  {
    uint32_t r = lane % dist;
    const uint32_t step = kWave % dist;
    for (uint32_t j = lane; j < dist + 20u; j += kWave) {
      pat[j] = src_period[r];
      r += step;
      if (r >= dist) r -= dist;
    }
  }
  const uint32_t head = (uint32_t)(-(uintptr_t)dst) & 15u;  // ml >= kPatMin > head
  if (lane < head) dst[lane] = src_period[lane % dist];
  vm_quiesce();
  const uint32_t body = (ml - head) & ~15u;
  uint8_t *d = dst + head;
  uint32_t r = (head + 16u * lane) % dist;  // where this lane's piece starts in the pattern
  const uint32_t step = 1024u % dist;
  for (uint32_t off = 16u * lane; off < body; off += 1024u) {
    const lds_u32 *q = (const lds_u32 *)(pat + (r & ~3u));
    const uint32_t w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3], w4 = q[4];
    const uint32_t sh = r & 3u;
    U128U v;
    v.a = __builtin_amdgcn_alignbyte(w1, w0, sh);
    v.b = __builtin_amdgcn_alignbyte(w2, w1, sh);
    v.c = __builtin_amdgcn_alignbyte(w3, w2, sh);
    v.d = __builtin_amdgcn_alignbyte(w4, w3, sh);
    st128(d + off, v);
    r += step;
    if (r >= dist) r -= dist;
  }
  const uint32_t tail = ml - head - body;
  if (lane < tail) d[body + lane] = pat[(head + body + lane) % dist];
  vm_quiesce();
}

template <bool HAS_PREFIX>
__device__ __forceinline__ void copy_match(uint8_t *dst, uint64_t op, uint32_t dist, uint32_t ml,
                                           const uint8_t *__restrict__ prefix, uint32_t prefix_len,
                                           uint32_t lane) {
  int64_t s = (int64_t)op - (int64_t)dist;  // source start relative to dst; < 0 reaches into the prefix
  if (dist >= ml) {
    if (!HAS_PREFIX || s >= 0) {
      wave_copy(dst + op, dst + s, ml, lane);
    } else {
      uint32_t npre = (uint32_t)(-s) < ml ? (uint32_t)(-s) : ml;
      wave_copy(dst + op, prefix + prefix_len + s, npre, lane);
      if (ml > npre) wave_copy(dst + op + npre, dst, ml - npre, lane);
    }
    vm_quiesce();
    return;
  }
  // Overlapping match: byte-serial forward copy semantics (byte_writer.dart:116-118)
  // == replication of the period [op-dist, op), which is complete before the match starts.
  // The first stretch is put down from the period itself (a plain copy of one period when the period
  // is at least a wavefront wide, else 256 bytes through a modulo gather); after that the match's own
  // output holds whole periods, and the rest is copied from it in spans that double each time
  // (a span is a multiple of the period and never longer than what has been written, so source and
  // destination of one copy do not overlap and it moves 16 bytes per lane).
  uint32_t done;
  if (dist >= kWave) {
    if (!HAS_PREFIX || s >= 0) {
      wave_copy(dst + op, dst + s, dist, lane);
    } else {
      uint32_t npre = (uint32_t)(-s) < dist ? (uint32_t)(-s) : dist;
      wave_copy(dst + op, prefix + prefix_len + s, npre, lane);
      if (dist > npre) wave_copy(dst + op + npre, dst, dist - npre, lane);
    }
    done = dist;
  } else {
    const uint32_t first = ml < 4 * kWave ? ml : 4 * kWave;
    uint32_t r = lane % dist;
    const uint32_t step = kWave % dist;
    for (uint32_t k = lane; k < first; k += kWave) {
      int64_t sp = s + r;
      uint8_t v;
      if (HAS_PREFIX && sp < 0)
        v = prefix[(int64_t)prefix_len + sp];
      else
        v = dst[sp];
      dst[op + k] = v;
      r += step;
      if (r >= dist) r -= dist;
    }
    done = first;
  }
  for (uint32_t span = done / dist * dist; done < ml; span *= 2) {
    const uint32_t nb = ml - done < span ? ml - done : span;
    wave_copy(dst + op + done, dst + op + done - span, nb, lane);
    done += nb;
  }
  vm_quiesce();
}

// State the two wavefronts of a block share (decode_block_pair).
struct PairShared {
  uint32_t hdr[2][4];        // per descriptor slot: sequences, output position behind the batch, compressed position, general-loop flag
  uint32_t desc[2][4][kWave];  // per slot and lane: match source position, match output position, length | far << 31, distance
  uint32_t ctl[8];           // behind the general loop: ip, op, fl, sneg, vlo, status, done
  uint32_t req[6];           // general loop: a copy for both wavefronts (kind, destination, source, length) or its end
  uint32_t next;             // next block of the queue
};
typedef __attribute__((address_space(3))) PairShared lds_pair;
constexpr uint32_t kReqCopy = 1, kReqEnd = 2;
constexpr uint32_t kPairCopyMin = 1024;  // literal runs from this length on are copied by both wavefronts

// The copying wavefront, inside the general loop, has a long copy: the parsing wavefront (waiting in
// pair_serve) takes the first part.  Both have left the copy behind when this returns.
__device__ __forceinline__ void pair_copy(lds_pair *sh, uint8_t *dst, const uint8_t *src, uint32_t n, uint32_t lane) {
  uint32_t h = n / 2u;
  h -= (uint32_t)((uintptr_t)dst + h) & 15u;  // the second part starts on a 16-byte boundary
  if (lane < 6) {
    const uint64_t d = (uint64_t)(uintptr_t)dst, g = (uint64_t)(uintptr_t)src;
    sh->req[lane] = lane == 0 ? kReqCopy : lane == 1 ? (uint32_t)d : lane == 2 ? (uint32_t)(d >> 32) : lane == 3 ? (uint32_t)g
                    : lane == 4 ? (uint32_t)(g >> 32) : h;
  }
  __syncthreads();
  if (n - h >= kCopyLong) wave_copy_long(dst + h, src + h, n - h, lane);
  else wave_copy(dst + h, src + h, n - h, lane);
  vm_quiesce();
  __syncthreads();
}
// The parsing wavefront while the general loop runs on the other one: serves its copies until the end is signalled.
__device__ __forceinline__ void pair_serve(lds_pair *sh, uint32_t lane) {
  for (;;) {
    __syncthreads();
    if (uni(sh->req[0]) == kReqEnd) break;
    uint8_t *dst = (uint8_t *)(uintptr_t)(((uint64_t)uni(sh->req[2]) << 32) | uni(sh->req[1]));
    const uint8_t *src = (const uint8_t *)(uintptr_t)(((uint64_t)uni(sh->req[4]) << 32) | uni(sh->req[3]));
    const uint32_t n = uni(sh->req[5]);
    if (n >= kCopyLong) wave_copy_long(dst, src, n, lane);
    else wave_copy(dst, src, n, lane);
    vm_quiesce();
    __syncthreads();
  }
}

constexpr uint32_t kRing = 4096;   // bytes of recent output the general loop mirrors in LDS
constexpr uint32_t kShort = 256;   // copies up to this length go through its ring path

// ---------------------------------------------------------------------------------------------
// General loop: one sequence per iteration, every check in the reference's order.  Decodes into
// dst[op_start ...).  `cap` is decompressedSize (STRICT) or the ByteWriter maxLength minus the
// pre-filled prefix (lenient).  A match may reach `back_limit` bytes before op_start: bytes at
// non-negative positions come from dst itself (earlier output), negative positions from the tail
// of `prefix` (dictionary).  Starts from fresh window/ring state, as if the rest of the block
// were a block of its own.  With ONE it returns kSwitchRun after one complete sequence (literals
// and match); in_used / out_len say how far it got.
constexpr int32_t kSwitchRun = -1;  // not a status: the block goes on at in_used / out_len
template <bool STRICT, bool HAS_PREFIX, bool ONE, bool PAIR = false>
__device__ __forceinline__ int32_t decode_run(const uint8_t *__restrict__ src, uint32_t src_len, uint8_t *dst,
                                              const uint64_t op_start, uint32_t cap, uint32_t back_limit,
                                              const uint8_t *__restrict__ prefix, uint32_t prefix_len,
                                              lds_u8 *ring, uint32_t lane, uint32_t &in_used, uint32_t &out_len,
                                              lds_pair *sh = nullptr) {
  InWindow w;
  w.mis = (uint32_t)((uintptr_t)src & 3u);
  w.al = src - w.mis;
  w.vend = (w.mis + src_len + 3u) & ~3u;
  w.base = 0xfffff000u;  // nothing loaded yet
  w.r0 = w.r1 = w.r2 = w.r3 = 0;
  Stream sm;
  sm.sp = 0xffffff00u;
  sm.S = 0;

  uint32_t ip = 0, op = 0;  // op counts bytes written by this call
  uint32_t rlo = 0;         // the ring mirrors output positions [rlo, op) of this call
  const uint32_t in_end = src_len;
  int32_t st = LZ4MI_OK;
  uint8_t *const out = dst + op_start;

  for (;;) {
    if (STRICT) {
      if (op >= cap) {  // loop condition :21, then the trailing check :72-74
        if (ip != in_end) st = LZ4MI_CORRUPT_TRAILING;
        break;
      }
      if (ip >= in_end) { st = LZ4MI_FORMAT_EOF; break; }  // :22-24
    } else {
      if (ip >= in_end) break;  // :86-88
    }
    uint32_t token = sbyte(w, sm, ip, lane);
    ip++;
    uint64_t lit64 = token >> 4;
    if (lit64 == 15) {
      if (!read_ext(w, sm, ip, in_end, lit64, lane)) { st = LZ4MI_FORMAT_EOF; break; }
    }
    if (STRICT && lit64 > (uint64_t)(cap - op)) { st = LZ4MI_CORRUPT_LITERAL; break; }  // :33-35
    if (lit64 != 0) {
      if (lit64 > (uint64_t)(in_end - ip)) { st = LZ4MI_FORMAT_EOF; break; }  // readBytesView
      if (!STRICT && lit64 > (uint64_t)(cap - op)) { st = LZ4MI_LIMIT; break; } // writer maxLength
      const uint32_t lit = (uint32_t)lit64;
      if (ip - sm.sp + lit <= kWave) {
        // the whole run is in the stream register: lanes [rel, rel + lit) store their byte
        const uint32_t rel = ip - sm.sp;
        const uint32_t k = lane - rel;
        if (k < lit) {
          out[op + k] = (uint8_t)sm.S;
          ring[(op + k) & (kRing - 1)] = (uint8_t)sm.S;
        }
        if (op + lit > rlo + kRing) rlo = op + lit - kRing;
      } else if (lit <= kShort) {
        // short run: 64 bytes per step out of the register window; the last gather also holds the
        // bytes that follow the run and becomes the stream register
        uint32_t v = 0, k0 = 0;
        for (; k0 < lit; k0 += kWave) {
          if (!win_covers(w, ip + k0 + w.mis)) win_seek(w, ip + k0 + w.mis, lane);
          const uint32_t k = k0 + lane;
          v = win_lane_byte(w, ip + k0, lane);
          if (k < lit) {
            out[op + k] = (uint8_t)v;
            ring[(op + k) & (kRing - 1)] = (uint8_t)v;
          }
        }
        sm.sp = ip + k0 - kWave;
        sm.S = v;
        if (op + lit > rlo + kRing) rlo = op + lit - kRing;
      } else {
        if (PAIR && lit >= kPairCopyMin) {
          pair_copy(sh, out + op, src + ip, lit, lane);
        } else {
          if (lit >= kCopyLong) wave_copy_long(out + op, src + ip, lit, lane);
          else wave_copy(out + op, src + ip, lit, lane);
          vm_quiesce();
        }
        rlo = op + lit;  // ring holds nothing older than this
      }
      ip += lit;
      op += lit;
    }
    if (STRICT) {
      if (op == cap) {  // :42-48
        if (ip != in_end) st = LZ4MI_CORRUPT_TRAILING;
        break;
      }
    } else {
      if (ip >= in_end) break;  // :102-104
    }
    if (in_end - ip < 2) { st = LZ4MI_FORMAT_EOF; break; }  // :50-52 / :106-108
    const uint32_t dist = sbyte(w, sm, ip, lane) | (sbyte(w, sm, ip + 1, lane) << 8);
    ip += 2;
    uint64_t ml64 = (token & 15u) + 4u;
    if ((token & 15u) == 15u) {
      if (!read_ext(w, sm, ip, in_end, ml64, lane)) { st = LZ4MI_FORMAT_EOF; break; }
    }
    if (STRICT && ml64 > (uint64_t)(cap - op)) { st = LZ4MI_CORRUPT_MATCH; break; }  // :61-63
    // copyMatch: distance first (byte_writer.dart:97-99), then capacity (:104)
    if (dist == 0 || (uint64_t)dist > (uint64_t)op + back_limit) { st = LZ4MI_CORRUPT_DISTANCE; break; }
    if (!STRICT && ml64 > (uint64_t)(cap - op)) { st = LZ4MI_LIMIT; break; }
    const uint32_t ml = (uint32_t)ml64;
    if (ml <= kShort) {
      // byte-serial copy semantics (byte_writer.dart:116-118): source byte of output k is
      // k mod dist bytes into the period that ends at op
      const int32_t s = (int32_t)op - (int32_t)dist;
      const int32_t ring_from = (int32_t)(op + ml > rlo + kRing ? op + ml - kRing : rlo);
      const bool overlap = dist < ml;
      uint32_t r = lane, step = kWave;
      if (overlap) {
        r = dist >= kWave ? lane : lane % dist;
        step = dist > kWave ? kWave : kWave % dist;
      }
      if (s >= ring_from) {
        // the whole source is in the LDS ring: no memory read, nothing to wait for
        for (uint32_t k = lane; k < ml; k += kWave) {
          const uint32_t v = ring[(uint32_t)(s + (int32_t)r) & (kRing - 1)];
          out[op + k] = (uint8_t)v;
          ring[(op + k) & (kRing - 1)] = (uint8_t)v;
          r += step;
          if (overlap && r >= dist) r -= dist;
        }
      } else {
        for (uint32_t k = lane; k < ml; k += kWave) {
          int32_t p = s + (int32_t)r;
          uint32_t v;
          if (p >= ring_from) {
            v = ring[(uint32_t)p & (kRing - 1)];
          } else {
            int64_t ap = (int64_t)op_start + p;
            v = (HAS_PREFIX && ap < 0) ? prefix[(int64_t)prefix_len + ap] : dst[ap];
          }
          out[op + k] = (uint8_t)v;
          ring[(op + k) & (kRing - 1)] = (uint8_t)v;
          r += step;
          if (overlap && r >= dist) r -= dist;
        }
        vm_quiesce();
      }
      rlo = (uint32_t)ring_from;
    } else if (dist <= kPatPeriod && ml >= kPatMin && (int64_t)(op_start + op) >= (int64_t)dist) {
      // (the period lies in the output, not in the prefix; the ring holds nothing the general loop needs past this match)
      pattern_fill(dst + op_start + op, dst + op_start + op - dist, dist, ml, ring, lane);
      rlo = op + ml;
    } else {
      copy_match<HAS_PREFIX>(dst, op_start + op, dist, ml, prefix, prefix_len, lane);
      rlo = op + ml;
    }
    op += ml;
    if (ONE) {
      // Back to the batch loop, unless the next sequence would come straight back: a literal run with eight or
      // more length bytes (token, then seven bytes of 255) is not for a batch.
      bool again = false;
      if (in_end - ip >= 9u) {
        if (ip - sm.sp + 8u > kWave) stream_load(w, sm, ip, lane);
        const uint32_t rel = ip - sm.sp;
        const uint64_t ff = ballot(sm.S == 255u) >> (rel + 1u);
        again = (rdlane(sm.S, rel) >> 4) == 15u && (ff & 0x7full) == 0x7full;
      }
      if (!again) { st = kSwitchRun; break; }
    }
  }
  in_used = ip;
  out_len = op;
  return st;
}

// ---------------------------------------------------------------------------------------------
// Batch loop.
#ifndef LZ4MI_HIST
#define LZ4MI_HIST 4096
#endif
#ifndef LZ4MI_BATCH
#define LZ4MI_BATCH 3584
#endif
#ifndef LZ4MI_DEC_WPE
#define LZ4MI_DEC_WPE 4  // resident wavefronts per SIMD the decode kernels are compiled for (register budget 512 / this)
#endif
#ifndef LZ4MI_PAIR_WPE
#define LZ4MI_PAIR_WPE 4
#endif
#ifndef LZ4MI_DEC_ATTR
#define LZ4MI_DEC_ATTR
#endif
constexpr uint32_t kH = LZ4MI_HIST;         // output bytes kept in LDS in front of a batch (multiple of 1024)
constexpr uint32_t kBatchOut = LZ4MI_BATCH; // most output bytes of one batch
constexpr uint32_t kStgBytes = kH + 16 + kBatchOut + 16;  // staging buffer of a wavefront on its own: history, alignment slack, batch, store slack
constexpr uint32_t kStgPair = kStgBytes + kBatchOut;      // of a pair of wavefronts: the batch being copied and the one being placed
// 64 + 8 bytes behind the buffer (offset D = its size in the copy functions) take what lanes with nothing to write write
constexpr uint32_t kStgAlloc = kStgBytes + 80, kStgPairAlloc = kStgPair + 80;
constexpr uint32_t kCw = 2048;              // compressed window in LDS
constexpr uint32_t kCwPad = 16;
constexpr uint32_t kCwTop = kCw + kCwPad - 12;  // highest offset an 8-byte read may start at (three aligned dwords from the one below it)
constexpr uint32_t kLitMax = 15 + 6 * 255 + 254;  // longest literal run of a batched sequence (up to seven extension bytes)
constexpr uint32_t kMatMax = 19 + 5 * 255 + 254;  // longest match (up to six extension bytes)
constexpr uint32_t kSeqIn = 8 + kLitMax + 8 + 8;  // compressed bytes a batched sequence may span, plus slack
constexpr uint32_t kSeqTyp = 320;           // a window with fewer bytes than this behind ip is not worth a parse
#ifndef LZ4MI_PRIO_T
#define LZ4MI_PRIO_T 12288, 8192, 4096
#endif
constexpr uint32_t kPrioT[3] = {LZ4MI_PRIO_T};  // compressed bytes left from which a block runs at issue priority 3, 2, 1
constexpr uint32_t kPrio3 = kPrioT[0], kPrio2 = kPrioT[1], kPrio1 = kPrioT[2];
constexpr uint32_t kHelpMin = 1024;         // compressed bytes left from which a block is worth a second wavefront
#ifndef LZ4MI_PACKED_MIN
#define LZ4MI_PACKED_MIN 10
#endif
#ifndef LZ4MI_WALK_CHUNKS
#define LZ4MI_WALK_CHUNKS 2
#endif
constexpr uint32_t kWalk = LZ4MI_WALK_CHUNKS;  // 64-byte chunks per step of the token walk
#ifndef LZ4MI_PACKED_RUN
#define LZ4MI_PACKED_RUN 2
#endif
constexpr uint32_t kPackedRun = LZ4MI_PACKED_RUN;  // ... in so many chunk steps in a row
constexpr uint32_t kPackedMin = LZ4MI_PACKED_MIN;  // tokens per 64-byte chunk from which the four-chunk step is taken
constexpr uint32_t kTl = 136;               // token list entries (two bytes each): 64 tokens, a spare, the dummy slot at 128
static_assert(kLitMax + kMatMax <= kBatchOut && kSeqIn <= kCw, "a batch holds at least one sequence");
static_assert(kStgBytes >= kRing, "the general loop uses the first kRing bytes of the staging buffer");
static_assert(kH % 1024 == 0, "layout");

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x2 lds_u64v;
typedef __attribute__((address_space(3))) uint16_t lds_u16;

// bytes at window offset a (clamped: a read past the window returns bytes that no caller uses).
// An LDS access whose address is not a multiple of its size stalls the CU's LDS pipe (shared by its
// sixteen wavefronts) for about twenty cycles, so the bytes are put together from aligned dwords
// (scripts/micro/lds_unaligned_tp.hip: 304 against 46 cycles per read and wavefront).
__device__ __forceinline__ uint32_t cw32(const lds_u8 *cw, uint32_t a) {
  a = a < kCwTop ? a : kCwTop;
  const lds_u32 *q = (const lds_u32 *)(cw + (a & ~3u));
  return __builtin_amdgcn_alignbyte(q[1], q[0], a & 3u);
}
__device__ __forceinline__ uint64_t cw64(const lds_u8 *cw, uint32_t a) {
  a = a < kCwTop ? a : kCwTop;
  const lds_u32 *q = (const lds_u32 *)(cw + (a & ~3u));
  const uint32_t w0 = q[0], w1 = q[1], w2 = q[2];
  return ((uint64_t)__builtin_amdgcn_alignbyte(w2, w1, a & 3u) << 32) | __builtin_amdgcn_alignbyte(w1, w0, a & 3u);
}

// Parses the byte at window offset a as a token, common form: each length with at most one
// extension byte.  nx1 = where the next token starts in that form, which is all the chain of token
// starts needs.  `lng`: a length has a second extension byte; seq_long reads such a sequence.
struct SeqFields {
  uint32_t ll, ls, ml, dist, nx;  // literal length, window offset of the literals, match length, distance, next token
  bool ok, lng;
};
__device__ __forceinline__ SeqFields spec_parse(const lds_u8 *cw, uint32_t a, uint32_t wlim) {
  SeqFields f;
  const uint32_t t4 = cw32(cw, a);
  const uint32_t ll0 = (t4 >> 4) & 15u, mc = t4 & 15u, e1 = (t4 >> 8) & 255u;
  const bool isl = ll0 == 15u, ism = mc == 15u;
  f.ll = ll0 + (isl ? e1 : 0u);
  f.ls = a + 1u + (isl ? 1u : 0u);
  const uint32_t q = f.ls + f.ll;
  const uint32_t o4 = cw32(cw, q);
  const uint32_t f1 = (o4 >> 16) & 255u;
  f.dist = o4 & 0xffffu;
  f.ml = mc + 4u + (ism ? f1 : 0u);
  f.nx = q + 2u + (ism ? 1u : 0u);
  const bool llong = isl && e1 == 255u;
  f.lng = llong || (ism && f1 == 255u);
  // the header as far as it was read lies inside the window (a long literal run: its first extension byte does)
  f.ok = (llong ? a + 2u : f.nx) <= wlim;
  return f;
}
__device__ __forceinline__ uint32_t ff_bytes(uint64_t x) {  // bytes of 0xFF at the low end of x (x has a byte that is not)
  return (uint32_t)__builtin_ctzll(~x) >> 3;
}
// The same for any sequence the batch takes: up to seven extension bytes of the literal length, up
// to six of the match length.  ok: within those limits and inside the window.
__device__ __forceinline__ SeqFields seq_long(const lds_u8 *cw, uint32_t a, uint32_t wlim) {
  SeqFields f;
  const uint64_t t8 = cw64(cw, a);
  const uint32_t tok = (uint32_t)t8 & 255u;
  const uint32_t ll0 = tok >> 4, mc = tok & 15u;
  const bool isl = ll0 == 15u, ism = mc == 15u;
  const uint64_t ext = t8 >> 8;                            // seven extension bytes, then a zero byte
  const uint32_t nff = ff_bytes(ext);                      // 0..7 bytes of 255 in front of the closing byte
  const uint32_t eb = (uint32_t)(ext >> (8u * (nff & 7u))) & 255u;
  f.ll = isl ? 15u + 255u * nff + eb : ll0;
  f.ls = a + 1u + (isl ? nff + 1u : 0u);
  const uint32_t q = f.ls + f.ll;
  const uint64_t o8 = cw64(cw, q);
  f.dist = (uint32_t)o8 & 0xffffu;
  const uint64_t mext = o8 >> 16;                          // six extension bytes, then zero bytes
  const uint32_t mff = ff_bytes(mext);
  const uint32_t mb = (uint32_t)(mext >> (8u * (mff & 7u))) & 255u;
  f.ml = ism ? 19u + 255u * mff + mb : mc + 4u;
  f.nx = q + 2u + (ism ? mff + 1u : 0u);
  f.lng = false;
  f.ok = !(isl && nff == 7u) && !(ism && mff == 6u) && f.nx <= wlim;
  return f;
}

// Loads kCw bytes of the block, starting at the 16-byte aligned address at or below src + ip, into the
// window.  wb = compressed offset of window byte 0 (may be negative by up to 15), wlim = valid bytes.
__device__ __forceinline__ void cw_load(lds_u8 *cw, const uint8_t *__restrict__ src, uint32_t in_end, uint32_t ip,
                                        uint32_t lane, int32_t &wb, uint32_t &wlim) {
  const uint32_t mis = (uint32_t)((uintptr_t)(src + ip) & 15u);
  wb = (int32_t)ip - (int32_t)mis;
  const uint8_t *g = src + wb;
  const uint32_t avail = in_end - (uint32_t)wb;
  wlim = avail < kCw ? avail : kCw;
  // every lane loads (a lane past the end re-reads the window's first 16 bytes): no branch around the loads
  const uint32_t o0 = 16u * lane, o1 = 1024u + 16u * lane;
  const u32x4 v0 = *(const u32x4 *)(g + (o0 < wlim ? o0 : 0u));
  const u32x4 v1 = *(const u32x4 *)(g + (o1 < wlim ? o1 : 0u));
  *(lds_u128 *)(cw + o0) = v0;
  *(lds_u128 *)(cw + o1) = v1;
  vm_quiesce();
}

__device__ __forceinline__ uint32_t mbcnt64(uint64_t m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Staging buffer: a linear image of output positions [sbase, sbase + kStgBytes): position p sits at offset
// p - sbase, and sbase + (low four address bits of position 0) is a multiple of 16, so a 16-byte aligned
// piece of memory is a 16-byte aligned piece of the buffer.
//
// Writes positions [from, to) of the buffer to memory; with `all` to the last byte, otherwise up to the
// last 16-byte boundary (the rest follows with the next batch).  Returns the new flushed-up-to position.
__device__ __forceinline__ uint32_t stg_flush(const lds_u8 *stg, uint8_t *out, uint32_t sneg, uint32_t from, uint32_t to,
                                              bool all, uint32_t lane) {
  if (from >= to) return from;
  const uint32_t head = (0u - (from + sneg)) & 15u;
  if (head) {
    const uint32_t n = head < to - from ? head : to - from;
    if (lane < n) out[from + lane] = stg[from + sneg + lane];
    from += n;
  }
  const uint32_t pieces = (to - from) >> 4;
  for (uint32_t c0 = 0; c0 < pieces; c0 += 4 * kWave) {
    u32x4 v[4];
#pragma unroll
    for (uint32_t u = 0; u < 4; u++) {  // lanes past the end store the last piece again
      const uint32_t c = c0 + u * kWave + lane;
      v[u] = *(const lds_u128 *)(stg + from + sneg + 16u * (c < pieces ? c : pieces - 1u));
    }
#pragma unroll
    for (uint32_t u = 0; u < 4; u++) {
      const uint32_t c = c0 + u * kWave + lane;
      *(u32x4 *)(out + from + 16u * (c < pieces ? c : pieces - 1u)) = v[u];
    }
  }
  from += 16u * pieces;
  if (all && from < to) {
    if (lane < to - from) out[from + lane] = stg[from + sneg + lane];
    from = to;
  }
  return from;
}

// Where stg_flush(.., all = false) stops: the flushed-up-to position after positions [from, to) were offered.
__device__ __forceinline__ uint32_t flush_end(uint32_t sneg, uint32_t from, uint32_t to) {
  if (from >= to) return from;
  const uint32_t head = (0u - (from + sneg)) & 15u;
  if (head) from += head < to - from ? head : to - from;
  return from + ((to - from) & ~15u);
}

// ---------------------------------------------------------------------------------------------
// A batch, in pieces.  parse_batch finds it (token chain, lane j reads sequence j, the reference's checks);
// copy_literals, copy_far, copy_matches and stg_flush move its bytes.  One wavefront may run all of them in
// turn (decode_block) or two may share them (decode_block_pair).
struct ParseCtx {      // what the parsing side keeps between batches
  int32_t wb;          // compressed offset of window byte 0 (may be negative by up to 15)
  uint32_t wlim;       // valid bytes in the window
  bool have_win;
  bool dense;          // token chain: tokens are close together (chunk steps) or far apart (scalar steps)
  uint32_t shortrun;   // scalar steps in a row that were short
  uint32_t packed;     // chunk steps in a row with so many tokens per chunk that four chunks at once pay (kPackedMin)
};
struct BatchPlan {     // wave-uniform
  uint32_t nb;         // sequences the batch takes
  uint32_t b1;         // output position behind them
  uint32_t ip_stop;    // compressed position behind them
  bool general;        // what follows goes through the general loop
};
struct SeqLane {       // lane j: sequence j
  uint32_t ls, ll, o_lit;     // literals: window offset, length (0 at or past nb), output position
  int32_t sp;                 // match: source position (may lie in front of the block)
  uint32_t o_mat, ml, dist;   //        output position, length, distance
  bool far;                   //        source older than the buffer: fetched from memory
};

// The token chain over four chunks (256 bytes) at once, without a walk from token to token: lane p of chunk c parses
// byte a + 64 c + p as a token; its link says where the token behind it starts: four times the lane, if that is in
// the same chunk, else the distance in lanes << 8 (the low byte is 0); a token that is not for the chain, or not
// inside the window, carries (0x8000 | lane) << 8.  Pointer jumping finds every lane's way out of its chunk in five
// rounds (a chunk holds at most 22 tokens), the four chunks side by side (every round issues its four cross-lane
// reads before it uses the first: left to itself the scheduler keeps two in flight); the ways out give the lane at
// which the chain enters each chunk, and five more rounds push a mark from there along the links of 1, 2, 4, 8 and
// 16 tokens.  A lane without a mark, or whose link leads out, sends to lane 0, which no link names.  The list may
// get more entries than a batch takes (they are found again by the next parse) and drops what does not fit.
// Returns tokens found | bytes the chain moved on << 8 | (it ends in front of a token that is not for it) << 31.
// A function of its own, not inlined: its twenty link registers stay out of the register allocation of the kernels'
// other paths, and the kernels stay smaller by one copy per instantiation.
__device__ __attribute__((noinline)) uint32_t chain_four_chunks(const lds_u8 *cw, lds_u16 *tl, uint32_t a, uint32_t wlim,
                                                                  uint32_t count, uint32_t lane) {
  constexpr uint32_t G = 4;
  uint32_t Jk[5][G], J[G];
#pragma unroll
  for (uint32_t c = 0; c < G; c++) {
    const uint32_t t4 = cw32(cw, a + 64u * c + lane);
    const uint32_t ll0 = (t4 >> 4) & 15u, e1 = (t4 >> 8) & 255u;
    const bool isl = ll0 == 15u;
    const uint32_t nx = lane + 3u + (isl ? 1u + e1 : 0u) + ll0 + ((t4 & 15u) == 15u ? 1u : 0u);
    const bool ok = a + 64u * c + nx <= wlim && !(isl && e1 == 255u);
    J[c] = ok ? (nx < kWave ? 4u * nx : nx << 8) : (0x8000u | lane) << 8;
  }
#pragma unroll
  for (uint32_t k = 0; k < 5; k++) {
    uint32_t j2[G];
#pragma unroll
    for (uint32_t c = 0; c < G; c++) {
      Jk[k][c] = J[c];
      j2[c] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)J[c], (int)J[c]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (uint32_t c = 0; c < G; c++) J[c] = J[c] < 256u ? j2[c] : J[c];
    __builtin_amdgcn_sched_barrier(0);
  }
  // the lane at which the chain enters each chunk (64: it passes over the chunk, or has ended), and where it ends
  uint32_t ent[G], r_end = 0;
  bool notok = false;
  {
    bool open = true;
#pragma unroll
    for (uint32_t c = 0; c < G; c++) {
      ent[c] = kWave;
      if (open && r_end < 64u * (c + 1u)) {
        ent[c] = r_end - 64u * c;
        const uint32_t x = rdlane(J[c], ent[c]) >> 8;
        notok = (x & 0x8000u) != 0u;  // the chain ends in front of that token
        r_end = 64u * c + (x & 0x7fffu);
        open = !notok;
      }
    }
  }
  const uint32_t not0 = lane != 0u ? ~0u : 0u;
  uint32_t R[G];
#pragma unroll
  for (uint32_t c = 0; c < G; c++) R[c] = lane == ent[c] ? ~0u : 0u;
#pragma unroll
  for (uint32_t k = 0; k < 5; k++) {
    uint32_t got[G];
#pragma unroll
    for (uint32_t c = 0; c < G; c++) got[c] = (uint32_t)__builtin_amdgcn_ds_permute((int)(R[c] & Jk[k][c]), -1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (uint32_t c = 0; c < G; c++) R[c] |= got[c] & not0;
    __builtin_amdgcn_sched_barrier(0);
  }
  uint32_t n = 0;
#pragma unroll
  for (uint32_t c = 0; c < G; c++) {
    const uint64_t starts = ballot(R[c] != 0u && !(Jk[0][c] & 0x800000u));
    const uint32_t at = count + n + mbcnt64(starts);
    tl[((starts >> lane) & 1ull) && at < 2u * kWave ? at : 2u * kWave] = (uint16_t)(a + 64u * c + lane);
    n += (uint32_t)__builtin_popcountll(starts);
  }
  return n | r_end << 8 | (notok ? 0x80000000u : 0u);
}

// lo = lowest output position the buffer holds when the batch is copied, fl = output in front of this
// position is in memory by then.  BATCH = most output bytes of the batch.
template <uint32_t BATCH>
__device__ __forceinline__ BatchPlan parse_batch(const uint8_t *__restrict__ src, uint32_t in_end, uint32_t ip, uint32_t op,
                                                 uint32_t cap, uint32_t back_limit, int32_t lo, uint32_t fl, lds_u8 *cw,
                                                 lds_u16 *tl, uint32_t lane, ParseCtx &pc, SeqLane &q) {
#ifdef LZ4MI_PROFILE
  PhState pps{ph_column(), __builtin_readcyclecounter()};  // slots 16..: where a parse spends its cycles
#define PP(i) PHS(pps, i)
#define PPC(i) do { pps.c[(i) * 64] += 1; } while (0)
#else
#define PP(i) do {} while (0)
#define PPC(i) do {} while (0)
#endif
  // ---- window: keep it while a typical sequence still fits behind ip, or it holds the block's end
  if (!pc.have_win || !((ip - (uint32_t)pc.wb) + kSeqTyp <= kCw || (uint32_t)(pc.wb + (int32_t)kCw) >= in_end)) {
    cw_load(cw, src, in_end, ip, lane, pc.wb, pc.wlim);
    pc.have_win = true;
  }
  const int32_t wb = pc.wb;
  const uint32_t wlim = pc.wlim;
  PP(16);
  // ---- chain of token starts.  Only where the tokens are is found here, under the assumption that no
  // match length has a second extension byte; the positions go into a list, lane j then reads sequence j
  // in full, and a sequence whose real successor is not the next position of the list ends the batch.
  // Three ways to step: where tokens are dense, lane p parses the bytes a + p and a + 64 + p of two 64-byte
  // chunks as tokens and the real starts are followed with one cross-lane read each; where they are packed
  // (kPackedMin per chunk, kPackedRun steps in a row) four chunks are taken at once by pointer jumping
  // (chain_four_chunks); where literal runs are long (one token per chunk), the token is read into scalar
  // registers and parsed there.
  uint32_t a = ip - (uint32_t)wb, count = 0, tot = 0;
  bool stopped = false;  // the chain ends at a token (at a) that is not one for the batch, or not inside the window
  bool dense = pc.dense;
  uint32_t packed = pc.packed;
  uint32_t shortrun = pc.shortrun;
  while (count < kWave && tot <= BATCH - 512u) {
    if (dense && packed >= kPackedRun) {  // four chunks at once
      const uint32_t res = uni(chain_four_chunks(cw, tl, a, wlim, count, lane));
      const uint32_t n = res & 0xffu, r_end = (res >> 8) & 0x7fffffu;
      const bool notok = (res >> 31) != 0u;
      count += n;
      count = count < 2u * kWave - 1u ? count : 2u * kWave - 1u;
      tot += r_end + 64u * n;  // compressed bytes stand for the literals, a guess for the matches
      a += r_end;
      dense = n >= 2u && !notok;
      packed = n >= kPackedMin * 4u ? packed : 0u;
      PP(17);
      PPC(21);
      continue;
    }
    if (dense && count <= kWave - 22u) {
      // kWalk chunks per step: lane p parses bytes a + 64 c + p as tokens, and the real starts are followed
      // from token to token through all of them (one LDS round trip and one set-up for that many bytes; where
      // tokens are far apart that is most of a step).  link = where the token behind this one starts, counted from
      // a; a token that is not for the chain (or not inside the window) points past every lane and carries its own
      // place, so the walk is one cross-lane read and one test per token.
      uint32_t link[kWalk];
#pragma unroll
      for (uint32_t c = 0; c < kWalk; c++) {
        const uint32_t t4 = cw32(cw, a + 64u * c + lane);
        const uint32_t ll0 = (t4 >> 4) & 15u, e1 = (t4 >> 8) & 255u;
        const bool isl = ll0 == 15u;
        const uint32_t nx = 64u * c + lane + 3u + (isl ? 1u + e1 : 0u) + ll0 + ((t4 & 15u) == 15u ? 1u : 0u);
        link[c] = (a + nx <= wlim && !(isl && e1 == 255u)) ? nx : 0x8000u | (64u * c + lane);
      }
      uint64_t starts[kWalk];
      uint32_t r = 0;
#pragma unroll
      for (uint32_t c = 0; c < kWalk; c++) {
        starts[c] = 0;
        while (r < 64u * (c + 1u)) {  // (the first token is at lane 0 of chunk 0)
          starts[c] |= 1ull << (r - 64u * c);
          r = rdlane(link[c], r - 64u * c);
        }
      }
      const bool notok = (r & 0x8000u) != 0u;
      if (notok) r &= 0x7fffu;  // the chain ends in front of that token
      uint32_t n = 0;
#pragma unroll
      for (uint32_t c = 0; c < kWalk; c++) {
        if (notok && r - 64u * c < kWave) starts[c] &= ~(1ull << (r - 64u * c));
        const uint32_t at = count + n + mbcnt64(starts[c]);
        tl[((starts[c] >> lane) & 1ull) && at < 2u * kWave ? at : 2u * kWave] = (uint16_t)(a + 64u * c + lane);
        n += (uint32_t)__builtin_popcountll(starts[c]);
      }
      count += n;
      count = count < 2u * kWave - 1u ? count : 2u * kWave - 1u;
      tot += r + 64u * n;  // compressed bytes stand for the literals, a guess for the matches
      a += r;
      dense = n >= 2u && !notok;
      packed = n >= kWalk * kPackedMin ? packed + 1u : 0u;
      PP(24);
      PPC(25);
      continue;
    }
    // one token, in scalar registers
    if (a + 4u > wlim) { stopped = true; break; }
    const uint64_t t8v = cw64(cw, a);
    const uint32_t lo32 = uni((uint32_t)t8v), hi32 = uni((uint32_t)(t8v >> 32));
    const uint32_t ll0 = (lo32 >> 4) & 15u;
    uint32_t ll = ll0, xl = 0;
    if (ll0 == 15u) {
      const uint32_t e1 = (lo32 >> 8) & 255u;
      ll = 15u + e1;
      xl = 1;
      if (e1 == 255u) {  // up to seven extension bytes
        const uint64_t ext = (((uint64_t)hi32 << 32) | lo32) >> 8;
        const uint32_t nff = ff_bytes(ext);
        if (nff == 7u) { stopped = true; break; }
        ll = 15u + 255u * nff + ((uint32_t)(ext >> (8u * nff)) & 255u);
        xl = nff + 1u;
      }
    }
    const uint32_t nx = a + 1u + xl + ll + 2u + ((lo32 & 15u) == 15u ? 1u : 0u);
    if (nx > wlim) { stopped = true; break; }
    tl[count] = (uint16_t)a;
    count++;
    tot += ll + 64u;
    shortrun = nx - a < 24u ? shortrun + 1u : 0u;
    dense = shortrun >= 2u;
    a = nx;
    PP(18);
    PPC(22);
  }
  PP(18);
  PPC(23);
  pc.dense = dense;
  pc.packed = packed;
  pc.shortrun = shortrun;
  BatchPlan bp;
  bp.nb = 0;
  bp.b1 = op;
  bp.ip_stop = (uint32_t)wb + a;
  // A token that ends the chain: too many extension bytes, the end of the block, or the end of the window.
  // A window that was loaded at this token holds any sequence the batch takes, so there it is not the window.
  bp.general = stopped && ((uint32_t)(wb + (int32_t)kCw) >= in_end || a < 16u);
  if (stopped && !bp.general) pc.have_win = false;
  q.ls = 0; q.ll = 0; q.o_lit = op; q.sp = 0; q.o_mat = op; q.ml = 0; q.dist = 1; q.far = false;
  if (!count) return bp;

  // ---- lane j owns sequence j
  const bool mine = lane < count;
  const uint32_t ta = tl[lane], tn = tl[lane + 1u];
  const SeqFields g = seq_long(cw, ta, wlim);
  const uint32_t ll = mine ? g.ll : 0u, ml = mine ? g.ml : 0u, dist = g.dist;
  PP(19);
  // the sequences in front of the first one that is not for the batch, or whose successor is not the
  // next token of the list, are real
  const uint64_t nokm = ballot(mine && !g.ok), brkm = ballot(mine && lane + 1u < count && g.nx != tn);
  const uint32_t nnok = nokm ? (uint32_t)__builtin_ctzll(nokm) : kWave;
  const uint32_t nbrk = brkm ? (uint32_t)__builtin_ctzll(brkm) + 1u : kWave;
  const uint32_t nreal = nnok < nbrk ? (nnok < count ? nnok : count) : (nbrk < count ? nbrk : count);
  const uint32_t incl = wave_incl_scan(ll + ml);
  const uint32_t o_lit = op + incl - (ll + ml);  // where the literals go
  const uint32_t o_mat = o_lit + ll;             // where the match goes
  // lz4_block_decoder.dart:33-35,61-63 and byte_writer.dart:97-104: a sequence that would fail
  // any of them is left to the general loop, which names the failure
  const bool good = incl <= cap - op && dist != 0u && (uint64_t)dist <= (uint64_t)o_mat + back_limit;
  const int32_t sp = (int32_t)o_mat - (int32_t)dist;  // match source (may lie in front of the block)
  bool far = mine && sp < lo;
  // an out-of-buffer source that reaches into output not yet in memory (possible right after the general
  // loop): general loop
  const bool bad = mine && (!good || (far && sp + (int32_t)ml > (int32_t)fl));
  const uint64_t realm = nreal < kWave ? (1ull << nreal) - 1ull : ~0ull;
  const uint64_t badm = ballot(bad) & realm, overm = ballot(incl > BATCH) & realm;
  const uint32_t nbad = badm ? (uint32_t)__builtin_ctzll(badm) : kWave;
  const uint32_t nover = overm ? (uint32_t)__builtin_ctzll(overm) : kWave;
  uint32_t nb = nreal;
  if (nbad < nb || nover < nb) {  // a sequence that fails a check, or the batch is full
    nb = nbad < nover ? nbad : nover;
    bp.general = nbad <= nover || nb == 0u;  // a first sequence that is a batch of its own and more: general loop
    pc.have_win = true;
    bp.ip_stop = (uint32_t)wb + rdlane(ta, nb);
  } else if (nreal < count && nnok < nbrk) {  // a token that is not for the batch
    // (strictly in front of the place where the list leaves the real chain: with nnok == nbrk the list entry at
    // nnok is not a token at all, it only parsed as one that a batch cannot take, and the block goes on at the
    // real successor of the sequence in front of it, below)
    const uint32_t at = rdlane(ta, nreal);
    bp.ip_stop = (uint32_t)wb + at;
    bp.general = (uint32_t)(wb + (int32_t)kCw) >= in_end || at < 16u;
    pc.have_win = bp.general;
  } else {
    // the chain went on behind a match with more extension bytes than it assumed (then the rest of the list
    // is void), or the list is used up: go on at the real successor of the last sequence taken
    const uint32_t nxl = rdlane(g.nx, nb - 1u);
    if (nreal < count || nxl != a) {
      bp.general = false;
      pc.have_win = true;
    }
    bp.ip_stop = (uint32_t)wb + nxl;
  }
  bp.nb = nb;
  bp.b1 = op + (nb ? rdlane(incl, nb - 1u) : 0u);
  const bool in = lane < nb;
  q.ls = in ? g.ls : 0u;
  q.ll = in ? ll : 0u;
  q.o_lit = o_lit;
  q.sp = sp;
  q.o_mat = o_mat;
  q.ml = in ? ml : 0u;
  q.dist = dist ? dist : 1u;
  q.far = far && in;
  PP(20);
  return bp;
}

template <uint32_t N>
struct UConst {
  static constexpr uint32_t value = N;
};

// ---- literals: lane j copies the first 64 bytes of its run, window -> buffer.  No lane is ever masked
// off (that costs more than the copies) and every LDS access is aligned (an unaligned one stalls the
// LDS pipe of the CU): the up to three bytes in front of the first aligned dword of the destination
// and the up to three behind the last go byte by byte, as does a run too short to hold an aligned
// dword; the dwords in between are put together from the aligned dwords of the window they straddle.
// What does not apply writes to a dummy slot.
template <uint32_t D>
__device__ __forceinline__ void copy_literals(const lds_u8 *cw, lds_u8 *stg, uint32_t sneg, const SeqLane &q, uint32_t lane) {
  const uint32_t n = q.ll;
  const uint32_t lsx = q.ls;
  const uint32_t d0 = q.o_lit + sneg;
  const uint32_t n1 = n < 64u ? n : 64u;
  const uint32_t head = (0u - d0) & 3u;
  const bool mid = n1 >= head + 4u;
  const uint32_t body = mid ? (n1 - head) >> 2 : 0u;  // aligned dwords, at most 16
  const uint32_t tail = mid ? (n1 - head) & 3u : 0u;
  {
    // as many dwords per lane as the longest run of the batch needs, in steps (text-like data through a strong
    // encoder has a byte or two of literals per sequence: most batches need none or two)
    const uint32_t sb = lsx + head, sh = sb & 3u;
    const lds_u32 *p = (const lds_u32 *)(cw + ((sb < kCwTop ? sb : kCwTop) & ~3u));
    lds_u32 *d = (lds_u32 *)(stg + d0 + head);
    lds_u32 *dummy = (lds_u32 *)(stg + D + 64u);
    const auto dwords = [&](auto kc) __attribute__((always_inline)) {
      constexpr uint32_t K = decltype(kc)::value;
      uint32_t wd[K + 1];
#pragma unroll
      for (uint32_t k = 0; k <= K; k++) wd[k] = p[k];
#pragma unroll
      for (uint32_t k = 0; k < K; k++) *(k < body ? d + k : dummy) = __builtin_amdgcn_alignbyte(wd[k + 1], wd[k], sh);
    };
    if (!ballot(body > 2u)) {
      if (ballot(body > 0u)) dwords(UConst<2>{});
    } else if (!ballot(body > 4u)) {
      dwords(UConst<4>{});
    } else if (!ballot(body > 8u)) {
      dwords(UConst<8>{});
    } else {
      dwords(UConst<16>{});
    }
  }
  {
    // six single bytes: head (or the whole of a short run) from the run's first bytes, tail from its last
    const uint64_t f8 = cw64(cw, lsx);
    const uint32_t toff = head + 4u * body;
    const uint32_t t4 = cw32(cw, lsx + toff);
    const uint32_t nfirst = mid ? head : n1;  // bytes taken from the front (at most six)
    lds_u8 *const at_front = stg + d0, *const at_back = stg + (d0 + toff - 3u), *const dummy = stg + D + lane;
#pragma unroll
    for (uint32_t i = 0; i < 6; i++) {  // (the byte's place is the store's constant offset)
      const bool front = i < nfirst;
      const bool back = mid && i >= 3u && i - 3u < tail;
      const uint32_t v = front ? (uint32_t)(f8 >> (8u * i)) : t4 >> (8u * (i - 3u < 3u ? i - 3u : 0u));
      (front ? at_front : back ? at_back : dummy)[i] = (uint8_t)v;
    }
  }
  // runs longer than 64 bytes, one after the other: bytes 56.. of the run, 256 bytes per step, the same way
  // with the whole wavefront on one run (at least 9 bytes are left)
  for (uint64_t lm = ballot(n > 64u); lm; lm &= lm - 1ull) {
    const uint32_t j = (uint32_t)__builtin_ctzll(lm);
    const uint32_t s0 = rdlane(lsx, j) + 56u, e0 = rdlane(d0, j) + 56u, rem = rdlane(n, j) - 56u;
    const uint32_t hb = (0u - e0) & 3u;
    const uint32_t bd = (rem - hb) >> 2, tl3 = (rem - hb) & 3u;
    {
      const bool edge = lane < hb || (lane >= 4u && lane - 4u < tl3);
      const uint32_t eo = lane < hb ? lane : hb + 4u * bd + (lane - 4u);
      const uint32_t v = cw[s0 + (edge ? eo : 0u)];
      stg[edge ? e0 + eo : D + lane] = (uint8_t)v;
    }
    const uint32_t sb = s0 + hb, sh = sb & 3u;
    const lds_u32 *p = (const lds_u32 *)(cw + (sb & ~3u));
    lds_u32 *d = (lds_u32 *)(stg + e0 + hb);
    for (uint32_t k0 = 0; k0 < bd; k0 += kWave) {
      const uint32_t k = k0 + lane;
      const uint32_t kk = k < bd ? k : 0u;
      const uint32_t v = __builtin_amdgcn_alignbyte(p[kk + 1u], p[kk], sh);
      *(k < bd ? d + k : (lds_u32 *)(stg + D + 64u)) = v;
    }
  }
}

// ---- matches whose source is older than the buffer: fetched from memory in pieces of up to 64 bytes (a byte per
// lane), sixteen pieces per round trip whichever matches they belong to.  (Data with long distances, and every
// linked block near its start, has most of its matches here: the round trips are what that costs.)
constexpr uint32_t kFarPieces = 16;
#ifndef LZ4MI_FAR_EARLY
#define LZ4MI_FAR_EARLY 8  // pieces of longer out-of-buffer matches loaded in front of the caller's work
#endif
#ifndef LZ4MI_FAR_SHORT
#define LZ4MI_FAR_SHORT 1  // 0: every out-of-buffer match goes through the pieces (A/B builds)
#endif
template <bool HAS_PREFIX, uint32_t D, uint32_t E, typename Between>
__device__ __forceinline__ void copy_far(lds_u8 *stg, const uint8_t *dst, uint64_t op_start, const uint8_t *__restrict__ prefix,
                                         uint32_t prefix_len, uint32_t sneg, const SeqLane &q, uint32_t lane,
                                         Between between) {
  uint64_t fm = ballot(q.far);
  // Short matches (text-like data through a strong encoder: eight bytes per match at distances all over the
  // block, so four of five are out of the buffer) would each take a piece of 64 lanes below.
  // One lane per match instead: a single 16-byte load (any alignment) per lane brings every short match of the
  // batch in one instruction and one round trip; the bytes behind a match's end are read and dropped (its source
  // lies more than the history in front of the output, so they are inside the output buffer).  A source that
  // touches the prefix's last 15 bytes would read past that buffer: it stays with the pieces below.
  // `between` is work of the caller's that neither reads nor writes what the matches here write (a wavefront on its
  // own: the batch's literal copies); it runs while the short matches' load and a first round of E pieces of the
  // longer ones are in flight, so most batches wait for memory once, behind other work.
  int32_t cs = 0;            // the match being cut into pieces: source position, buffer offset, bytes left
  uint32_t cd = 0, cn = 0;
  const auto next_piece = [&](int32_t &s_, uint32_t &d_, uint32_t &n_) __attribute__((always_inline)) {
    if (cn == 0 && fm) {
      const uint32_t j = (uint32_t)__builtin_ctzll(fm);
      fm &= fm - 1ull;
      cs = (int32_t)rdlane((uint32_t)q.sp, j);
      cd = rdlane(q.o_mat, j) + sneg;
      cn = rdlane(q.ml, j);
    }
    const uint32_t n = cn < kWave ? cn : kWave;
    s_ = cs; d_ = cd; n_ = n;
    cs += (int32_t)n; cd += n; cn -= n;
  };
  const auto piece_load = [&](int32_t s_, uint32_t n_) __attribute__((always_inline)) -> uint32_t {
    if (!n_) return 0u;  // lanes past the end read the last byte again
    const int64_t ap = (int64_t)op_start + s_ + (int64_t)(lane < n_ ? lane : n_ - 1u);
    return (HAS_PREFIX && ap < 0) ? prefix[(int64_t)prefix_len + ap] : dst[ap];
  };
  {
    const int64_t ap = (int64_t)op_start + q.sp;
    const bool fs = LZ4MI_FAR_SHORT && q.far && q.ml <= 16u && (!HAS_PREFIX || ap >= 0 || ap + 16 <= 0);
    const uint64_t sm = ballot(fs);
    const bool shorts = (uint32_t)__builtin_popcountll(sm) >= 4u;
    typedef u32x4 u32x4_any __attribute__((aligned(1)));
    typedef __attribute__((address_space(1))) u32x4_any g_v4any;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (shorts && fs) v = *(const g_v4any *)((HAS_PREFIX && ap < 0) ? prefix + ((int64_t)prefix_len + ap) : dst + ap);
    if (shorts) fm &= ~sm;
    constexpr uint32_t EA = E ? E : 1u;  // (no zero-length arrays)
    int32_t es[EA];
    uint32_t ed[EA], en[EA], ev[EA];
#pragma unroll
    for (uint32_t u = 0; u < EA; u++) es[u] = 0, ed[u] = 0, en[u] = 0, ev[u] = 0;
    if (E && fm) {
#pragma unroll
      for (uint32_t u = 0; u < E; u++) next_piece(es[u], ed[u], en[u]);
#pragma unroll
      for (uint32_t u = 0; u < E; u++) ev[u] = piece_load(es[u], en[u]);
    }
    between();
    if (shorts) {
      // (one base per lane and the byte's place as the store's constant offset: the dummy slot has room for it)
      const uint32_t n = fs ? q.ml : 0u;
      lds_u8 *const to = stg + (q.o_mat + sneg), *const dummy = stg + D + lane;
#pragma unroll
      for (uint32_t t = 0; t < 16u; t++) {
        if ((t == 8u || t == 12u) && !ballot(t < n)) break;
        (t < n ? to : dummy)[t] = (uint8_t)(v[t >> 2] >> (8u * (t & 3u)));
      }
    }
#pragma unroll
    for (uint32_t u = 0; u < E; u++)
      if (en[u]) *(lane < en[u] ? stg + lane + ed[u] : stg + D + lane) = (uint8_t)ev[u];
    if (shorts || (E && en[0])) vm_quiesce();
  }
  if (cn == 0 && !fm) return;
  for (;;) {
    int32_t ps[kFarPieces];
    uint32_t pd[kFarPieces], pn[kFarPieces];
#pragma unroll
    for (uint32_t u = 0; u < kFarPieces; u++) next_piece(ps[u], pd[u], pn[u]);
    if (pn[0] == 0) break;
    uint32_t v[kFarPieces];
#pragma unroll
    for (uint32_t u = 0; u < kFarPieces; u++) v[u] = piece_load(ps[u], pn[u]);
#pragma unroll
    for (uint32_t u = 0; u < kFarPieces; u++)
      if (pn[u]) *(lane < pn[u] ? stg + lane + pd[u] : stg + D + lane) = (uint8_t)v[u];
    vm_quiesce();
    if (cn == 0 && !fm) break;
  }
}

// ---- matches in order, buffer -> buffer.  LDS operations of a wavefront execute in issue order, so a
// match reads whatever earlier matches and the literals wrote.  byte_writer.dart:116-118 copies
// byte by byte: output byte k is byte k mod dist of the period in front of the match.
// A match is copied in pieces of up to 64 bytes, a byte per lane; lanes past the end write to a dummy slot.
#ifndef LZ4MI_NEAR_PACK
#define LZ4MI_NEAR_PACK 1
#endif
template <uint32_t D>
__device__ __forceinline__ void copy_matches(lds_u8 *stg, uint32_t sneg, uint32_t nb, const SeqLane &q, uint32_t lane) {
  // Every value the copy loop needs sits in a register of its own, ready for one cross-lane read: the loop
  // runs once per match on the scalar unit, which the wavefronts of a CU share, and unpacking there costs
  // more than the extra reads.  An overlapping match repeats its period: lane mod dist without a division
  // is lane - ((lane * mg) >> 16) * dist with mg = 65536 / dist + 1 (off by less than 1 / dist for lane < 64),
  // and a piece of 64 bytes moves on by 64 mod dist.  A plain copy has mg = 0 (lane mod dist = lane where
  // it matters), step 64 and a period it never reaches.
  const bool ovl = q.dist < q.ml;
  const uint32_t m_src = ((uint32_t)q.sp + sneg) & 0xffffu, m_dst = q.o_mat + sneg;
  const uint32_t m_len = q.far ? 0u : q.ml, m_per = ovl ? q.dist : 0xffffu;
  uint32_t m_mg = 0u, m_step = kWave;
  if (ovl) {  // the two divisions only where a lane has such a match (text-like data: hardly ever)
    m_mg = 65536u / q.dist + 1u;
    m_step = 64u % q.dist;
  }
  // Short matches out of the history: a match of at most sixteen bytes whose source ends in front of the batch's
  // first output byte reads nothing that this batch writes, so all of them are copied first, packed four to an
  // LDS operation (sixteen lanes per match, as the out-of-buffer loads are), every read of a round queued before
  // its writes.  What is left for the loop below keeps its order, and none of it reads a byte that such a match
  // writes later than the loop expects: a later match's output lies behind an earlier match's source.
  uint32_t m_todo = m_len;
  if (LZ4MI_NEAR_PACK) {
    const bool pk = lane < nb && m_len != 0 && m_len <= 16u && !ovl &&
                    q.sp + (int32_t)q.ml <= (int32_t)rdlane(q.o_lit, 0);
    const uint64_t pm = ballot(pk);
    const uint32_t np = (uint32_t)__builtin_popcountll(pm);
    if (np >= 4u) {
      const int to = (int)(4u * (pk ? mbcnt64(pm) : 63u));  // lane k: the k-th of them (lane 63 takes the rest)
      const uint32_t c_sl = (uint32_t)__builtin_amdgcn_ds_permute(to, (int)(m_src | (m_len << 16)));
      const uint32_t c_d = (uint32_t)__builtin_amdgcn_ds_permute(to, (int)m_dst);
      const uint32_t g = lane >> 4, t = lane & 15u;
      for (uint32_t k0 = 0; k0 < np; k0 += 32u) {
        uint32_t v[8];
#pragma unroll
        for (uint32_t r = 0; r < 8; r++) {
          v[r] = 0;
          if (k0 + 4u * r < np) {
            const uint32_t k = k0 + 4u * r + g;
            const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(4u * k), (int)c_sl);
            v[r] = stg[k < np ? (a & 0xffffu) + t : D + lane];
          }
        }
#pragma unroll
        for (uint32_t r = 0; r < 8; r++) {
          if (k0 + 4u * r < np) {
            const uint32_t k = k0 + 4u * r + g;
            const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(4u * k), (int)c_sl);
            const uint32_t dd = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(4u * k), (int)c_d);
            stg[k < np && t < (a >> 16) ? dd + t : D + lane] = (uint8_t)v[r];
          }
        }
      }
      if (pk) m_todo = 0;
    }
  }
  // A match of at most 64 bytes that does not overlap its source (nearly every match of text-like data) is one read
  // and one write; bit 31 of the length marks the others (a period to repeat, or more than one piece).
  lds_u8 *const at_lane = stg + lane, *const dummy = stg + D + lane;
  const auto one = [&](uint32_t j, uint32_t mf) __attribute__((always_inline)) {
    const uint32_t ms = rdlane(m_src, j), md = rdlane(m_dst, j);
    if (__builtin_expect(!(mf >> 31), 1)) {
      const uint32_t v = at_lane[ms];
      *(lane < mf ? at_lane + md : dummy) = (uint8_t)v;
    } else {
      const uint32_t m = mf & 0x7fffffffu;
      const uint32_t per = rdlane(m_per, j), step = rdlane(m_step, j);
      uint32_t r = lane - __umul24(__umul24(lane, rdlane(m_mg, j)) >> 16, per);  // lane mod dist
      for (uint32_t k0 = 0; k0 < m; k0 += kWave) {
        const uint32_t v = stg[ms + r];
        *(k0 + lane < m ? at_lane + md + k0 : dummy) = (uint8_t)v;
        r += step;
        r = r >= per ? r - per : r;
      }
    }
  };
  m_todo |= (ovl || m_todo > kWave) ? 0x80000000u : 0u;
  // Most of the batch still to copy: count through it.  Few left (the rest was out of the buffer or went packed):
  // step from one to the next over a bit mask, which costs a little more per match and nothing per skipped one.
  const uint64_t left0 = ballot(lane < nb && (m_todo & 0x7fffffffu) != 0u);
  if (2u * (uint32_t)__builtin_popcountll(left0) > nb) {
    for (uint32_t j = 0; j < nb; j++) {
      const uint32_t m = rdlane(m_todo, j);
      if (m & 0x7fffffffu) one(j, m);
    }
  } else {
    for (uint64_t left = left0; left; left &= left - 1ull) {
      const uint32_t j = (uint32_t)__builtin_ctzll(left);
      one(j, rdlane(m_todo, j));
    }
  }
}

// Moves the buffer down by `shift` (a multiple of 16): `bytes` bytes from offset shift on.
__device__ __forceinline__ void stg_slide(lds_u8 *stg, uint32_t shift, uint32_t bytes, uint32_t lane) {
  const uint32_t pieces = (bytes + 15u) >> 4;
  for (uint32_t c0 = 0; c0 < pieces; c0 += 4 * kWave) {
    u32x4 v[4];
#pragma unroll
    for (uint32_t u = 0; u < 4; u++) {  // lanes past the end move the last piece again
      const uint32_t c = c0 + u * kWave + lane;
      v[u] = *(const lds_u128 *)(stg + shift + 16u * (c < pieces ? c : pieces - 1u));
    }
#pragma unroll
    for (uint32_t u = 0; u < 4; u++) {
      const uint32_t c = c0 + u * kWave + lane;
      *(lds_u128 *)(stg + 16u * (c < pieces ? c : pieces - 1u)) = v[u];
    }
  }
}

// Behind the general loop (which used the buffer for itself), or at the start of a linked block: brings back the
// history, up to H bytes of output in front of position op in whole aligned 16-byte pieces, under the layout sneg.
// Nothing in front of position min_pos is read (0: the block's own output; a linked block's history goes on into
// the blocks in front of it).  Returns the lowest position held.
template <uint32_t H>
__device__ __forceinline__ int32_t stg_restore(lds_u8 *stg, uint8_t *out, uint32_t op, uint32_t sneg, int32_t min_pos, uint32_t lane) {
  // piece c holds positions [16c - sneg, 16c - sneg + 16); the last piece that holds a byte of output is
  // the one of position op - 1
  const uint32_t last = (op - 1u + sneg) >> 4;  // <= H / 16
  // lowest piece that lies inside the output that exists
  const int32_t sn = (int32_t)sneg + min_pos;  // buffer offset of position min_pos
  const uint32_t cmin = sn > 0 ? (uint32_t)(sn + 15) >> 4 : 0u;
  const u32x4 z = {0, 0, 0, 0};
  const uint8_t *g = out - (int32_t)sneg;
  for (uint32_t b0 = cmin; b0 <= last; b0 += 5 * kWave) {  // five pieces per lane and round trip (all of H = 4096)
    const uint32_t c0 = b0 + lane, c1 = c0 + kWave, c2 = c0 + 2 * kWave, c3 = c0 + 3 * kWave, c4 = c0 + 4 * kWave;
    const u32x4 v0 = c0 <= last ? *(const u32x4 *)(g + 16 * (int32_t)c0) : z;
    const u32x4 v1 = c1 <= last ? *(const u32x4 *)(g + 16 * (int32_t)c1) : z;
    const u32x4 v2 = c2 <= last ? *(const u32x4 *)(g + 16 * (int32_t)c2) : z;
    const u32x4 v3 = c3 <= last ? *(const u32x4 *)(g + 16 * (int32_t)c3) : z;
    const u32x4 v4 = c4 <= last ? *(const u32x4 *)(g + 16 * (int32_t)c4) : z;
    if (c0 <= last) *(lds_u128 *)(stg + 16u * c0) = v0;
    if (c1 <= last) *(lds_u128 *)(stg + 16u * c1) = v1;
    if (c2 <= last) *(lds_u128 *)(stg + 16u * c2) = v2;
    if (c3 <= last) *(lds_u128 *)(stg + 16u * c3) = v3;
    if (c4 <= last) *(lds_u128 *)(stg + 16u * c4) = v4;
  }
  vm_quiesce();
  return (int32_t)(16u * cmin) - (int32_t)sneg;
}

// Compressed to 15/16 of its output or more: (nearly) nothing but literals.
__device__ __forceinline__ bool literal_block(uint32_t src_len, uint32_t cap) {
  return (uint64_t)src_len * 16u >= (uint64_t)cap * 15u;
}

// Longest remaining work first: the wavefronts of a SIMD compete for its issue slots, and a launch
// ends with its slowest block.  What is left of the compressed block stands for the work left.
__device__ __forceinline__ void prio_by_left(uint32_t left) {
#ifndef LZ4MI_NO_PRIO
  if (left >= kPrio3) __builtin_amdgcn_s_setprio(3);
  else if (left >= kPrio2) __builtin_amdgcn_s_setprio(2);
  else if (left >= kPrio1) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
#endif
}

// Where a block stands between two batches: enough for a second wavefront to join in (decode_block_pair).
struct BlockState {
  uint32_t ip, op;   // compressed and output position of the next batch
  uint32_t fl;       // output in front of this position is in memory; [fl, op) is in the buffer only
  uint32_t sneg;     // buffer offset of position p is p + sneg
  int32_t vlo;       // the buffer holds output positions [max(vlo, -sneg), op)
};
constexpr int32_t kHandOver = -2;  // not a status: decode_block stopped between two batches for a helper to join

// The two wavefronts of a workgroup that decode a block each: the one that runs out of blocks offers its help,
// the other takes the offer between two batches and publishes where its block stands.
struct HelpShared {
  uint32_t offer[2];   // wavefront w has nothing left to do
  uint32_t accept;     // 1 + w: wavefront w took the other one's offer
  uint32_t block;      // its block
  uint32_t state[5];   // BlockState
};
typedef __attribute__((address_space(3))) HelpShared lds_help;

// One wavefront per block, batch after batch: the form for launches that fill the chip (sixteen blocks per CU).
// With `help`: looks for the other wavefront's offer between two batches and returns kHandOver (state in bs, pc).
template <bool STRICT, bool HAS_PREFIX>
__device__ __forceinline__ int32_t decode_block(const uint8_t *__restrict__ src, uint32_t src_len, uint8_t *dst,
                                                const uint64_t op_start, uint32_t cap, uint32_t back_limit,
                                                const uint8_t *__restrict__ prefix, uint32_t prefix_len,
                                                lds_u8 *stg, lds_u8 *cw, lds_u16 *tl, uint32_t lane, uint32_t &out_len,
                                                lds_help *help, uint32_t other, BlockState &bs, ParseCtx &pc) {
  const uint32_t in_end = src_len;
  uint8_t *const out = dst + op_start;
  const uint32_t goff = (uint32_t)((uintptr_t)out & 15u);
  uint32_t ip = 0, op = 0;
  uint32_t fl = 0;          // output in front of this position is in memory; [fl, op) is in the buffer only
  int32_t vlo = 0;          // the buffer holds output positions [max(vlo, sbase), op)
  uint32_t sneg = kH + goff;  // buffer offset of position p is p + sneg (sneg = -sbase)
  pc = ParseCtx{0, 0, false, false, 0};
  int32_t st = LZ4MI_OK;
  // A block that is hardly shorter than its output starts with a long literal run (often it is nothing else):
  // its first sequence goes straight to the general loop, without a window and a parse in front of the copy.
  bool lit_first = literal_block(src_len, cap);

  for (;;) {
    bool general = true;
    if (in_end - ip >= 8u && !lit_first) {
      prio_by_left(in_end - ip);
      // ---- slide the buffer so that the batch starts right behind kH bytes of history
      const uint32_t shift = (op + sneg - kH) & ~15u;  // op + sneg >= kH always
      if (shift) {
        stg_slide(stg, shift, kH + 16u, lane);
        sneg -= shift;
      }
      if (help && in_end - ip >= kHelpMin && uni(help->offer[other])) {
        bs = BlockState{ip, op, fl, sneg, vlo};
        out_len = op;
        return kHandOver;
      }
      SeqLane q;
      const int32_t lo = vlo > -(int32_t)sneg ? vlo : -(int32_t)sneg;  // lowest position the buffer holds
      const BatchPlan bp = parse_batch<kBatchOut>(src, in_end, ip, op, cap, back_limit, lo, fl, cw, tl, lane, pc, q);
      general = bp.general;
      if (bp.nb) {
        copy_far<HAS_PREFIX, kStgBytes, LZ4MI_FAR_EARLY>(stg, dst, op_start, prefix, prefix_len, sneg, q, lane,
                                        [&]() __attribute__((always_inline)) { copy_literals<kStgBytes>(cw, stg, sneg, q, lane); });
        copy_matches<kStgBytes>(stg, sneg, bp.nb, q, lane);
        // ---- the batch goes to memory in aligned 16-byte pieces
        fl = stg_flush(stg, out, sneg, fl, bp.b1, false, lane);
        op = bp.b1;
      }
      ip = bp.ip_stop;
    }
    lit_first = false;
    if (!general) continue;

    // ---- one sequence through the general loop (and every end of a block)
    fl = stg_flush(stg, out, sneg, fl, op, true, lane);
    uint32_t di = 0, dn = 0;
    st = decode_run<STRICT, HAS_PREFIX, true>(src + ip, in_end - ip, dst, op_start + op, cap - op, back_limit + op,
                                              prefix, prefix_len, stg, lane, di, dn);
    ip += di;
    op += dn;
    fl = op;
    if (st != kSwitchRun) break;
    sneg = kH + ((op + goff) & 15u) - op;
    vlo = (int32_t)op;
    if (in_end - ip >= 8u && op > 0) vlo = stg_restore<kH>(stg, out, op, sneg, 0, lane);
  }
#ifndef LZ4MI_NO_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif
  out_len = op;
  return st;
}

// ---------------------------------------------------------------------------------------------
// Two wavefronts per block.  A block on one wavefront is one chain of dependent steps (find the tokens, read the
// sequences, place the literals, copy match after match), each waiting on the LDS round trip of the one before, and a
// launch of few blocks per wavefront ends with its slowest block.  Here the parsing wavefront (0) finds batch k + 1
// and places its literals while the copying wavefront (1) fetches and copies the matches of batch k and writes it to
// memory; they meet at a barrier twice per batch.  Everything the parser needs from the copier's side it can work out
// itself: where the buffer will stand (the slide is a function of the batch ends) and how far the output will have
// been written to memory.  The general loop runs on the copier with the parser waiting, and hands the state back
// through `ctl`.

// BATCH = most output bytes of a batch, H = bytes of history the buffer keeps, D = size of the buffer
// (H + 16 + 2 * BATCH + 16).  The block starts at `start` (its first byte, or where a wavefront on its own had got to:
// then pc is that wavefront's parse state).  LINKED: the output in front of the block (back to position -hist_len)
// is history too, and the buffer starts with the last H bytes of it.
template <bool STRICT, bool HAS_PREFIX, uint32_t BATCH, uint32_t D, uint32_t H = kH, bool LINKED = false>
__device__ __forceinline__ int32_t decode_block_pair(const uint8_t *__restrict__ src, uint32_t src_len, uint8_t *dst,
                                                     const uint64_t op_start, uint32_t cap, uint32_t back_limit,
                                                     const uint8_t *__restrict__ prefix, uint32_t prefix_len,
                                                     lds_u8 *stg, lds_u8 *cw, lds_u16 *tl,
                                                     __attribute__((address_space(3))) PairShared *sh, uint32_t role,
                                                     uint32_t lane, uint32_t &out_len, const BlockState start, ParseCtx pc,
                                                     uint32_t hist_len = 0) {
  static_assert(D == H + 16 + 2 * BATCH + 16 && D + 80 <= 65536, "layout");
  const uint32_t in_end = src_len;
  uint8_t *const out = dst + op_start;
  const uint32_t goff = (uint32_t)((uintptr_t)out & 15u);
  // both wavefronts keep the same picture: the batch that is being copied starts at (ip, op), fl and the layout
  // (sneg, vlo) are the copier's
  uint32_t ip = start.ip, op = start.op, fl = start.fl;
  int32_t vlo = start.vlo;
  uint32_t sneg = start.sneg;
  int32_t st = LZ4MI_OK;
  bool lit_first = start.ip == 0u && literal_block(src_len, cap);  // the first sequence goes straight to the general loop
  const int32_t min_pos = LINKED ? -(int32_t)hist_len : 0;  // lowest position that is output of this frame
  PH_DECL
  if (LINKED && hist_len && !lit_first) {
    // the copying wavefront wrote that output (this block's predecessors ran on the same two wavefronts)
    int32_t lo = 0;
    if (role == 1) lo = stg_restore<H>(stg, out, 0u, sneg, min_pos, lane);
    const int32_t sn = (int32_t)sneg + min_pos;
    vlo = (int32_t)(16u * (sn > 0 ? (uint32_t)(sn + 15) >> 4 : 0u)) - (int32_t)sneg;  // what stg_restore returns
    (void)lo;
  }

  for (;;) {
    // ---- fill: behind the block's start or the general loop the pipeline starts with a batch of nothing, so the
    // parser makes the first real one with nobody copying yet.  The layout stands as that batch needs it (no slide).
    if (role == 0 && lane < 4) sh->hdr[0][lane] = lane == 1 ? op : lane == 2 ? ip : lane == 3 && lit_first ? 1u : 0u;
    lit_first = false;
    PH(0);
    __syncthreads();
    PH(1);
    uint32_t cur = 0;
    for (;;) {
      const uint32_t nb = uni(sh->hdr[cur][0]), b1 = uni(sh->hdr[cur][1]), ip_stop = uni(sh->hdr[cur][2]);
      const bool stop = uni(sh->hdr[cur][3]) != 0u;
      // where the buffer stands for the batch after this one, and how far memory is written by then
      const uint32_t shift = stop ? 0u : (b1 + sneg - H) & ~15u;
      const uint32_t fl_next = flush_end(sneg, fl, b1);
      if (role == 1) {
        if (nb) {
          SeqLane q;
          q.ls = 0; q.ll = 0; q.o_lit = 0;
          q.sp = (int32_t)sh->desc[cur][0][lane];
          q.o_mat = sh->desc[cur][1][lane];
          const uint32_t mf = sh->desc[cur][2][lane];
          q.ml = mf & 0x7fffffffu;
          q.far = (mf >> 31) != 0u;
          q.dist = sh->desc[cur][3][lane];
          if (lane >= nb) { q.ml = 0; q.far = false; q.dist = 1; }
          prio_by_left(in_end - ip);
          copy_far<HAS_PREFIX, D, LZ4MI_FAR_EARLY>(stg, dst, op_start, prefix, prefix_len, sneg, q, lane, []() {});  // (no work meanwhile, but both kinds of loads in one round trip)
          PH(2);
          copy_matches<D>(stg, sneg, nb, q, lane);
          PH(3);
        }
        (void)stg_flush(stg, out, sneg, fl, b1, false, lane);  // ends at fl_next
        PH(4);
      } else if (!stop) {
        BatchPlan bp = {0, b1, ip_stop, true};
        SeqLane q;
        if (in_end - ip_stop >= 8u) {
          prio_by_left(in_end - ip_stop);
          const uint32_t sneg_next = sneg - shift;
          const int32_t lo = vlo > -(int32_t)sneg_next ? vlo : -(int32_t)sneg_next;
          uint32_t at = ip_stop;
          do {
            bp = parse_batch<BATCH>(src, in_end, at, b1, cap, back_limit, lo, fl_next, cw, tl, lane, pc, q);
            at = bp.ip_stop;
          } while (!bp.nb && !bp.general);
          PH(5);
          if (bp.nb) copy_literals<D>(cw, stg, sneg, q, lane);  // the buffer stands as the batch being copied needs it
          PH(6);
        }
        if (bp.nb) {
          sh->desc[cur ^ 1u][0][lane] = (uint32_t)q.sp;
          sh->desc[cur ^ 1u][1][lane] = q.o_mat;
          sh->desc[cur ^ 1u][2][lane] = q.ml | (q.far ? 0x80000000u : 0u);
          sh->desc[cur ^ 1u][3][lane] = q.dist;
        }
        if (lane < 4)
          sh->hdr[cur ^ 1u][lane] = lane == 0 ? bp.nb : lane == 1 ? bp.b1 : lane == 2 ? bp.ip_stop : (bp.general ? 1u : 0u);
      }
      __syncthreads();
      PH(7);
      op = b1;
      ip = ip_stop;
      fl = fl_next;
      if (stop) break;
      // ---- slide: the next batch starts right behind kH bytes of history; its literals are already in the buffer
      if (shift) {
        if (role == 1) stg_slide(stg, shift, H + 16u + (uni(sh->hdr[cur ^ 1u][1]) - b1) + 16u, lane);
        sneg -= shift;
      }
      PH(8);
      __syncthreads();
      PH(9);
      cur ^= 1u;
    }
    // ---- one sequence through the general loop (and every end of a block), on the copying wavefront
    if (role == 0) {
      pair_serve(sh, lane);
    } else {
      fl = stg_flush(stg, out, sneg, fl, op, true, lane);
      uint32_t di = 0, dn = 0;
      st = decode_run<STRICT, HAS_PREFIX, true, true>(src + ip, in_end - ip, dst, op_start + op, cap - op, back_limit + op,
                                                      prefix, prefix_len, stg, lane, di, dn, sh);
      ip += di;
      op += dn;
      fl = op;
      if (st == kSwitchRun) {
        sneg = H + ((op + goff) & 15u) - op;
        vlo = (int32_t)op;
        if (in_end - ip >= 8u && (op > 0 || min_pos < 0)) vlo = stg_restore<H>(stg, out, op, sneg, min_pos, lane);
      }
      if (lane < 7)
        sh->ctl[lane] = lane == 0 ? ip : lane == 1 ? op : lane == 2 ? fl : lane == 3 ? sneg : lane == 4 ? (uint32_t)vlo
                        : lane == 5 ? (uint32_t)st : (st != kSwitchRun ? 1u : 0u);
      if (lane == 0) sh->req[0] = kReqEnd;
      PH(10);
      __syncthreads();
    }
    PH(11);
    ip = uni(sh->ctl[0]);
    op = uni(sh->ctl[1]);
    fl = uni(sh->ctl[2]);
    sneg = uni(sh->ctl[3]);
    vlo = (int32_t)uni(sh->ctl[4]);
    st = (int32_t)uni(sh->ctl[5]);
    const bool done = uni(sh->ctl[6]) != 0u;
    __syncthreads();  // ctl is read before the next general loop writes it
    if (done) break;
  }
  PH_FLUSH(g_dec_phase + role * PH_SLOTS, lane);
#ifndef LZ4MI_NO_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif
  out_len = op;
  return st;
}

// What a block's decode needs, fetched in one memory round trip: lane k loads the k-th per-block word (the arrays a
// launch does not use cost nothing), one wait, cross-lane reads.  Literal-only blocks are short enough for a chain of
// dependent loads in front of each of them to show.
struct BlockParams {
  const uint8_t *src;
  uint8_t *dst;
  uint32_t src_len, cap, stored;
};
__device__ __forceinline__ BlockParams load_block_params(const BatchArgs &a, uint32_t b, uint32_t lane) {
  uint32_t lo = 0, hi = 0;
  if (lane == 0 && a.src_off) { const uint64_t v = a.src_off[b]; lo = (uint32_t)v; hi = (uint32_t)(v >> 32); }
  if (lane == 1 && a.dst_off) { const uint64_t v = a.dst_off[b]; lo = (uint32_t)v; hi = (uint32_t)(v >> 32); }
  if (lane == 2 && a.src_len) lo = a.src_len[b];
  if (lane == 3 && a.dst_cap) lo = a.dst_cap[b];
  if (lane == 4 && a.stored_len) lo = a.stored_len[b];
  BlockParams p;
  const uint64_t so = a.src_off ? ((uint64_t)rdlane(hi, 0) << 32) | rdlane(lo, 0) : (uint64_t)b * a.src_stride;
  const uint64_t dof = a.dst_off ? ((uint64_t)rdlane(hi, 1) << 32) | rdlane(lo, 1) : (uint64_t)b * a.dst_stride;
  p.src = a.src + so;
  p.dst = a.dst + dof;
  if (a.src_len) {
    p.src_len = rdlane(lo, 2);
  } else {
    const uint64_t start = (uint64_t)b * a.src_len_uniform;
    const uint64_t left = a.src_total > start ? a.src_total - start : 0;
    p.src_len = left < a.src_len_uniform ? (uint32_t)left : a.src_len_uniform;
  }
  p.cap = a.dst_cap ? rdlane(lo, 3) : a.dst_cap_uniform;
  p.stored = a.stored_len ? rdlane(lo, 4) : 0u;
  return p;
}

template <uint32_t H = kH>
__device__ __forceinline__ BlockState block_start(const uint8_t *out) {
  return BlockState{0, 0, 0, H + (uint32_t)((uintptr_t)out & 15u), 0};
}

// Two wavefronts per workgroup, a block each (a wavefront's buffers are what the hardware packs: sixteen wavefronts
// per CU).  A launch ends with its slowest block, so a wavefront that finds the queue empty does not leave: it
// offers its help, and the other one, between two batches of its last block, takes it.  From there on the two share
// that block as decode_block_pair does, in the two buffers taken together.
template <bool STRICT, bool HAS_PREFIX>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(LZ4MI_DEC_WPE, LZ4MI_DEC_WPE))) LZ4MI_DEC_ATTR void decode_kernel(BatchArgs a, uint32_t *counter, const uint32_t *order) {
  __shared__ __attribute__((aligned(16))) uint8_t s_stg[2][kStgAlloc];
  __shared__ __attribute__((aligned(16))) uint8_t s_cw[2][kCw + kCwPad];
  __shared__ __attribute__((aligned(16))) uint16_t s_tl[2][kTl];
  __shared__ HelpShared s_help;
  static_assert(kStgPairAlloc % 16 == 0 && kStgPairAlloc + sizeof(PairShared) <= 2 * kStgAlloc, "the pair's buffer and shared words fit the two buffers");
  const uint32_t lane = lane_id();
  const uint32_t wave = uni(threadIdx.x / kWave);
  lds_help *help = (lds_help *)&s_help;
  if (threadIdx.x < 3) ((lds_u32 *)help)[threadIdx.x] = 0;  // offer[0], offer[1], accept
  __syncthreads();
  lds_u8 *stg = (lds_u8 *)s_stg[wave];
  lds_u8 *cw = (lds_u8 *)s_cw[wave];
  lds_u16 *tl = (lds_u16 *)s_tl[wave];
  // the decoder keeps the last 65536 bytes of a dictionary (lz4_frame_decoder.dart:278-282)
  const uint8_t *prefix = a.dict;
  const uint32_t prefix_len = a.dict_len;
  // When the launch has a wavefront for every block, a wavefront's block is its own index; larger
  // batches are handed out through the queue, where a wavefront's first block is still its own
  // index (thousands of wavefronts popping one queue head at launch serialise on it).
  const uint32_t nwaves = gridDim.x * 2;
  const bool direct = a.nblocks <= nwaves;
  BlockState bs = {0, 0, 0, 0, 0};
  ParseCtx pc = {0, 0, false, false, 0};
  bool handed = false;  // this wavefront took the other one's offer: its block goes on below
  uint32_t b = uni(blockIdx.x * 2 + wave);
  for (;;) {
    if (b >= a.nblocks) break;
    if (order) b = uni(order[b]);  // queue position -> block (long blocks first)
    const BlockParams bp_ = load_block_params(a, b, lane);
    const uint8_t *src = bp_.src;
    const uint32_t src_len = bp_.src_len;
    uint8_t *dst = bp_.dst;
    const uint32_t cap = bp_.cap;
    uint32_t out_len = 0;
#ifdef LZ4MI_DEC_SPANS
    if (lane == 0 && b < 65536) g_dec_span[2 * b] = __builtin_amdgcn_s_memrealtime();
#endif
    int32_t st = LZ4MI_OK;
    const uint32_t stored = bp_.stored;
    if (stored) {  // a stored block rides in the same queue: its copy overlaps the issue-bound blocks
      if (stored >= kCopyLong) wave_copy_long(dst, src, stored, lane);
      else wave_copy(dst, src, stored, lane);
      vm_quiesce();
      out_len = stored;
    } else {
      st = decode_block<STRICT, HAS_PREFIX>(src, src_len, dst, 0, cap, prefix_len, prefix, prefix_len, stg, cw, tl, lane,
#ifdef LZ4MI_NO_HELP
                                            out_len, nullptr, wave ^ 1u, bs, pc);
#else
                                            out_len, help, wave ^ 1u, bs, pc);
#endif
      if (st == kHandOver) {
        handed = true;
        break;
      }
    }
#ifdef LZ4MI_DEC_SPANS
    if (lane == 0 && b < 65536) g_dec_span[2 * b + 1] = __builtin_amdgcn_s_memrealtime();
#endif
    if (lane == 0) {
      a.out_len[b] = out_len;
      a.status[b] = st;
    }
    if (direct) break;
    b = nwaves + next_block(counter, lane);
  }
  // ---- out of blocks, or holding one that the other wavefront will help with
  if (handed) {
    if (lane == 0) {
      help->accept = 1u + wave;
      help->block = b;
      help->state[0] = bs.ip; help->state[1] = bs.op; help->state[2] = bs.fl; help->state[3] = bs.sneg;
      help->state[4] = (uint32_t)bs.vlo;
    }
  } else if (lane == 0) {
    help->offer[wave] = 1u;
  }
  __syncthreads();
  const uint32_t acc = uni(help->accept);
  if (!acc) return;  // both are out of blocks
  const uint32_t owner = acc - 1u;  // the block is the owner's and lives in its buffers; the owner goes on parsing
  b = uni(help->block);
  bs = BlockState{uni(help->state[0]), uni(help->state[1]), uni(help->state[2]), uni(help->state[3]), (int32_t)uni(help->state[4])};
  {
    const BlockParams bp_ = load_block_params(a, b, lane);
    const uint8_t *src = bp_.src;
    const uint32_t src_len = bp_.src_len;
    uint8_t *dst = bp_.dst;
    const uint32_t cap = bp_.cap;
    const uint32_t role = wave == owner ? 0u : 1u;
    // The pair's buffer (two batches behind the history) is the two wavefronts' buffers taken together, from the
    // first one's start; the pair's shared words sit behind it.  A block of the second wavefront moves there first
    // (what a buffer holds between two batches is its history: less than kH + 32 bytes).
    lds_u8 *pstg = (lds_u8 *)s_stg[0];
    if (owner == 1u) {
      if (wave == 0u) {
        for (uint32_t c = lane; c < (kH + 32u) / 16u; c += kWave) *(lds_u128 *)(pstg + 16u * c) = *(const lds_u128 *)((lds_u8 *)s_stg[1] + 16u * c);
      }
      __syncthreads();
    }
    lds_pair *sh = (lds_pair *)(pstg + kStgPairAlloc);
    uint32_t out_len = 0;
    const int32_t st = decode_block_pair<STRICT, HAS_PREFIX, kBatchOut, kStgPair>(
        src, src_len, dst, 0, cap, prefix_len, prefix, prefix_len, pstg, (lds_u8 *)s_cw[owner],
        (lds_u16 *)s_tl[owner], sh, role, lane, out_len, bs, pc);
#ifdef LZ4MI_DEC_SPANS
    if (role == 1 && lane == 0 && b < 65536) g_dec_span[2 * b + 1] = __builtin_amdgcn_s_memrealtime();
#endif
    if (role == 1 && lane == 0) {
      a.out_len[b] = out_len;
      a.status[b] = st;
    }
  }
}

// One workgroup of two wavefronts per block: the form for launches that leave wavefront slots empty (a block is
// done in 0.6 of the time a wavefront on its own takes, but a CU holds eight such blocks instead of sixteen).
template <bool STRICT, bool HAS_PREFIX>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(LZ4MI_PAIR_WPE, LZ4MI_PAIR_WPE))) LZ4MI_DEC_ATTR void decode_pair_kernel(BatchArgs a, uint32_t *counter, const uint32_t *order) {
  __shared__ __attribute__((aligned(16))) uint8_t s_stg[kStgPairAlloc];
  __shared__ __attribute__((aligned(16))) uint8_t s_cw[kCw + kCwPad];
  __shared__ __attribute__((aligned(16))) uint16_t s_tl[kTl];
  __shared__ __attribute__((aligned(16))) PairShared s_sh;
  const uint32_t lane = lane_id();
  const uint32_t role = uni(threadIdx.x / kWave);
  lds_u8 *stg = (lds_u8 *)s_stg;
  lds_u8 *cw = (lds_u8 *)s_cw;
  lds_u16 *tl = (lds_u16 *)s_tl;
  __attribute__((address_space(3))) PairShared *sh = (__attribute__((address_space(3))) PairShared *)&s_sh;
  // the decoder keeps the last 65536 bytes of a dictionary (lz4_frame_decoder.dart:278-282)
  const uint8_t *prefix = a.dict;
  const uint32_t prefix_len = a.dict_len;
  // When the launch has a workgroup for every block, a workgroup's block is its own index; larger
  // batches are handed out through the queue, where a workgroup's first block is still its own
  // index (thousands of workgroups popping one queue head at launch serialise on it).
  const uint32_t ngroups = gridDim.x;
  const bool direct = a.nblocks <= ngroups;
  uint32_t b = uni(blockIdx.x);
  for (;;) {
    if (b >= a.nblocks) break;
    if (order) b = uni(order[b]);  // queue position -> block (long blocks first)
    const BlockParams bp_ = load_block_params(a, b, lane);
    const uint8_t *src = bp_.src;
    const uint32_t src_len = bp_.src_len;
    uint8_t *dst = bp_.dst;
    const uint32_t cap = bp_.cap;
    uint32_t out_len = 0;
#ifdef LZ4MI_DEC_SPANS
    if (role == 1 && lane == 0 && b < 65536) g_dec_span[2 * b] = __builtin_amdgcn_s_memrealtime();
#endif
    int32_t st = LZ4MI_OK;
    const uint32_t stored = bp_.stored;
    if (stored) {  // a stored block rides in the same queue: its copy overlaps the issue-bound blocks
      const uint32_t half = (stored / 2u) & ~15u;
      if (role == 0) wave_copy(dst, src, half, lane);
      else wave_copy(dst + half, src + half, stored - half, lane);
      vm_quiesce();
      out_len = stored;
    } else {
      st = decode_block_pair<STRICT, HAS_PREFIX, kBatchOut, kStgPair>(src, src_len, dst, 0, cap, prefix_len, prefix, prefix_len,
                                                                     stg, cw, tl, sh, role, lane, out_len, block_start(dst),
                                                                     ParseCtx{0, 0, false, false, 0});
    }
#ifdef LZ4MI_DEC_SPANS
    if (role == 1 && lane == 0 && b < 65536) g_dec_span[2 * b + 1] = __builtin_amdgcn_s_memrealtime();
#endif
    if (role == 1 && lane == 0) {
      a.out_len[b] = out_len;
      a.status[b] = st;
    }
    if (direct) break;
    if (role == 0) {
      const uint32_t nx = next_block(counter, lane);
      if (lane == 0) sh->next = nx;
    }
    __syncthreads();
    b = ngroups + uni(sh->next);
    __syncthreads();  // next is read before the parsing wavefront writes it again
  }
}

// One workgroup has the CU's LDS to itself here, so the buffer keeps 32 KiB of history instead of 4: a linked block
// made of short matches into the block in front of it (what priming the table with the whole history makes of text)
// finds nearly all of its sources in LDS; only distances beyond 32 KiB are fetched from memory.
constexpr uint32_t kHChain = 32768, kStgChain = kHChain + 16 + 2 * kBatchOut + 16;

// Linked ("dependent") blocks: block b may reference the 64 KiB that precede it in the
// frame output, topped up from the dictionary while the output is shorter than that
// (_decodeDependentBlock, lib/src/frame/lz4_frame_decoder.dart:326-382).  The chain is
// serial, so one workgroup walks the blocks in order over one contiguous output.
__global__ __launch_bounds__(128) void decode_chain_kernel(ChainArgs c) {
  __shared__ __attribute__((aligned(16))) uint8_t s_stg[kStgChain + 80];
  __shared__ __attribute__((aligned(16))) uint8_t s_cw[kCw + kCwPad];
  __shared__ __attribute__((aligned(16))) uint16_t s_tl[kTl];
  __shared__ __attribute__((aligned(16))) PairShared s_sh;
  lds_u8 *stg = (lds_u8 *)s_stg;
  lds_u8 *cw = (lds_u8 *)s_cw;
  lds_u16 *tl = (lds_u16 *)s_tl;
  __attribute__((address_space(3))) PairShared *sh = (__attribute__((address_space(3))) PairShared *)&s_sh;
  const uint32_t lane = lane_id();
  const uint32_t role = uni(threadIdx.x / kWave);
  uint64_t start = 0;
  uint32_t b = 0;
  for (; b < c.nblocks; b++) {
    if (start > c.dst_limit) break;  // the caller's output limit is already crossed: the host reports it
    const uint8_t *payload = c.src + c.payload_off[b];
    const uint32_t len = c.payload_len[b];
    uint32_t produced = 0;
    int32_t st = LZ4MI_OK;
    if (c.raw[b]) {
      if (role == 1) {
        wave_copy(c.dst + start, payload, len, lane);
        vm_quiesce();
      }
      produced = len;
    } else {
      uint32_t back = start < 65536u ? (uint32_t)start : 65536u;
      if (back < 65536u && c.dict_len) back += c.dict_len < 65536u - back ? c.dict_len : 65536u - back;
      st = decode_block_pair<false, true, kBatchOut, kStgChain, kHChain, true>(
          payload, len, c.dst, start, c.block_max, back, c.dict, c.dict_len, stg, cw, tl, sh, role, lane, produced,
          block_start<kHChain>(c.dst + start), ParseCtx{0, 0, false, false, 0},
          start < kHChain ? (uint32_t)start : kHChain);
    }
    if (role == 1 && lane == 0) {
      c.out_len[b] = produced;
      c.status[b] = st;
    }
    if (st != LZ4MI_OK) { b++; break; }
    start += produced;
  }
  if (role == 1)
    for (uint32_t k = b + lane; k < c.nblocks; k += kWave) {  // blocks after an error never ran
      c.out_len[k] = 0;
      c.status[k] = LZ4MI_OK;
    }
}

#ifdef LZ4MI_DEC_SPANS
extern "C" void lz4mi_debug_dec_spans(unsigned long long *out, int n) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dec_span), sizeof(unsigned long long) * 2 * n);
}
#endif
#ifdef LZ4MI_PROFILE
extern "C" void lz4mi_debug_dec_phases(unsigned long long *out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dec_phase), sizeof(unsigned long long) * 2 * PH_SLOTS);
  if (reset) {
    unsigned long long z[2 * PH_SLOTS] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dec_phase), z, sizeof z);
  }
}
#endif

void launch_decode_chain(const ChainArgs &c, hipStream_t s) {
  if (!c.nblocks) return;
  hipLaunchKernelGGL(decode_chain_kernel, dim3(1), dim3(128), 0, s, c);
}

// A launch ends when its longest block does, and the queue hands blocks out in order: a long block
// far down the queue starts late.  What a block costs the decoder shows in its size: literal-only
// blocks (compressed >= about the output size) and run-like blocks (a few bytes) are streamed, the
// ones in between are sequence after sequence on one wavefront ("heavy").  Heavy blocks are bound by
// instruction issue, streamed ones by memory bandwidth, so the queue mixes them: `mix` streamed
// blocks behind every heavy one while both kinds last.  mix = 0 puts every heavy block in front
// (right for a launch of few blocks per wavefront, which ends with its slowest block); a launch of
// many blocks per wavefront runs both kinds side by side from start to end.
// Step 1: kind and index within the kind of every block (tag[b] = heavy << 31 | index), totals in cnt[0..1].
__global__ __launch_bounds__(256) void decode_order_count_kernel(BatchArgs a, uint32_t *cnt, uint32_t *tag) {
  const uint32_t lane = lane_id();
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = b < a.nblocks;
  bool heavy = false;
  if (valid) {
    const uint64_t c = block_src_len(a, b), n = a.dst_cap ? a.dst_cap[b] : a.dst_cap_uniform;
    heavy = c * 16 < n * 15 && c * 32 > n && !(a.stored_len && a.stored_len[b]);
  }
  const uint64_t hm = ballot(heavy), lm = ballot(valid && !heavy);
  uint32_t bh = 0, bl = 0;
  if (lane == 0) {
    if (hm) bh = atomicAdd(cnt, (uint32_t)__builtin_popcountll(hm));
    if (lm) bl = atomicAdd(cnt + 1, (uint32_t)__builtin_popcountll(lm));
  }
  bh = uni(bh);
  bl = uni(bl);
  const uint64_t below = (1ull << lane) - 1ull;
  if (heavy) tag[b] = 0x80000000u | (bh + (uint32_t)__builtin_popcountll(hm & below));
  else if (valid) tag[b] = bl + (uint32_t)__builtin_popcountll(lm & below);
}

// Step 2: queue positions.  G = min(H, L / mix) groups of one heavy block and `mix` streamed ones, then
// the heavy blocks that are left, then the streamed ones that are left.
__global__ __launch_bounds__(256) void decode_order_place_kernel(uint32_t nblocks, uint32_t mix, const uint32_t *cnt,
                                                                 const uint32_t *tag, uint32_t *order) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblocks) return;
  const uint32_t H = cnt[0], L = cnt[1];
  const uint32_t t = tag[b], k = t & 0x7FFFFFFFu;
  uint32_t pos;
  if (mix == 0) {
    pos = (t >> 31) ? k : H + k;
  } else {
    const uint32_t G = H < L / mix ? H : L / mix;
    if (t >> 31) pos = k < G ? k * (mix + 1u) : G * (mix + 1u) + (k - G);
    else pos = k < G * mix ? (k / mix) * (mix + 1u) + 1u + k % mix : G * (mix + 1u) + (H - G) + (k - G * mix);
  }
  order[pos] = b;
}

// The order of a launch that ends with its slowest block (up to kOrderSmallMax blocks), in one launch of one
// workgroup.  Three kinds by what a block will cost: 0 heavy as above; 1 nearly incompressible but not
// literal-only (compressed to less than one literal run of the output size takes: a text-like block with a few hundred sequences between long
// literal runs takes 100-300 us on its wavefront, ten times a literal-only block, and as one of the last out
// of the queue it used to end the launch 50 us behind the last heavy block); 2 streamed (literal-only,
// run-like, stored).  Kinds in that order, index order within a kind.  The workgroup counts blocks per kind
// and group of 64 in LDS, scans the counts, places every block and zeroes the queue head on the way.
constexpr uint32_t kOrderSmallMax = 65536;
__global__ __launch_bounds__(1024) void decode_order_small_kernel(BatchArgs a, uint32_t *counter, uint32_t *order) {
  __shared__ uint32_t s_cnt[2][kOrderSmallMax / kWave];  // per kind 0/1: blocks per group, then blocks in front of the group
  __shared__ uint32_t s_wsum[2][16];
  const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const uint32_t nb = a.nblocks, ngroups = (nb + kWave - 1) / kWave;
  if (tid == 0) *counter = 0;
  // group g = wave + 16 r belongs to this wavefront; bits 2r, 2r+1 of kinds[r / 16]: kind of this lane's block
  // of that group.  Sixteen groups per step with their size words in flight together (a single workgroup has
  // nothing else to hide the round trips with).
  constexpr uint32_t kStep = 16;  // groups per step: 16384 blocks are one round trip
  uint32_t kinds[4] = {0, 0, 0, 0};
  const uint32_t *sl = a.src_len, *dc = a.dst_cap, *stl = a.stored_len;
#pragma unroll
  for (uint32_t r0 = 0; r0 < 64; r0 += kStep) {
    if (r0 * 16 >= ngroups) break;
    uint32_t bb[kStep], c[kStep], n[kStep], st[kStep];
#pragma unroll
    for (uint32_t u = 0; u < kStep; u++) {
      const uint32_t b = (wave + 16 * (r0 + u)) * kWave + lane;
      bb[u] = b < nb ? b : nb - 1;
    }
    if (sl) {
#pragma unroll
      for (uint32_t u = 0; u < kStep; u++) c[u] = sl[bb[u]];
    } else {
#pragma unroll
      for (uint32_t u = 0; u < kStep; u++) c[u] = block_src_len(a, bb[u]);
    }
#pragma unroll
    for (uint32_t u = 0; u < kStep; u++) n[u] = a.dst_cap_uniform;
    if (dc) {
#pragma unroll
      for (uint32_t u = 0; u < kStep; u++) n[u] = dc[bb[u]];
    }
#pragma unroll
    for (uint32_t u = 0; u < kStep; u++) st[u] = 0;
    if (stl) {
#pragma unroll
      for (uint32_t u = 0; u < kStep; u++) st[u] = stl[bb[u]];
    }
#pragma unroll
    for (uint32_t u = 0; u < kStep; u++) {
      const uint32_t g = wave + 16 * (r0 + u);
      const bool valid = g * kWave + lane < nb;
      // 32-bit arithmetic where it can be: one CU runs this for every block of the launch
      const uint32_t cu = c[u], nu = n[u];
      const uint64_t cc = cu, nn = nu;
      uint32_t kind = 2;
      const uint64_t lit_only = nn + 1 + (nu >= 15 ? (nu - 15) / 255u + 1u : 0u);  // token, length bytes, the literals
      if (st[u] == 0 && cu > (nu >> 5)) kind = (cc << 4) < nn * 15 ? 0u : (cc < lit_only ? 1u : 2u);
      const uint64_t m0 = ballot(valid && kind == 0), m1 = ballot(valid && kind == 1);
      if (lane == 0 && g < ngroups) {
        s_cnt[0][g] = (uint32_t)__builtin_popcountll(m0);
        s_cnt[1][g] = (uint32_t)__builtin_popcountll(m1);
      }
      kinds[(r0 + u) / 16] |= kind << (2 * ((r0 + u) % 16));
    }
  }
  __syncthreads();
  uint32_t tot[2];
#pragma unroll
  for (uint32_t k = 0; k < 2; k++) {
    const uint32_t v = tid < ngroups ? s_cnt[k][tid] : 0u;
    const uint32_t incl = wave_incl_scan(v);
    if (lane == kWave - 1) s_wsum[k][wave] = incl;
    __syncthreads();
    uint32_t wbase = 0, all = 0;
    for (uint32_t j = 0; j < 16; j++) {
      const uint32_t t = s_wsum[k][j];
      if (j < wave) wbase += t;
      all += t;
    }
    tot[k] = all;
    s_cnt[k][tid] = wbase + incl - v;  // every thread read its own entry above
  }
  __syncthreads();
  const uint64_t below = (1ull << lane) - 1ull;
#pragma unroll
  for (uint32_t q = 0; q < 4; q++) {
    for (uint32_t r = 16 * q; r < 16 * q + 16 && wave + 16 * r < ngroups; r++) {
      const uint32_t g = wave + 16 * r, b = g * kWave + lane;
      const bool valid = b < nb;
      const uint32_t kind = (kinds[q] >> (2 * (r % 16))) & 3u;
      const uint32_t b0 = s_cnt[0][g] + (uint32_t)__builtin_popcountll(ballot(valid && kind == 0) & below);  // kind-0 blocks below b
      const uint32_t b1 = s_cnt[1][g] + (uint32_t)__builtin_popcountll(ballot(valid && kind == 1) & below);
      if (valid) order[kind == 0 ? b0 : kind == 1 ? tot[0] + b1 : tot[0] + tot[1] + (b - b0 - b1)] = b;
    }
  }
}

constexpr uint32_t kMix = 3;            // streamed blocks per heavy block in the queue of a long launch
constexpr uint32_t kMixFromRounds = 8;  // a launch is long from this many blocks per resident wavefront

// 0: the form follows the launch size; 1: one wavefront per block; 2: two wavefronts per block.
// LZ4MI_DEC_FORM=single|pair sets the initial value, lz4mi_set_decode_form changes it.
static std::atomic<int> g_decode_form{[] {
  const char *e = getenv("LZ4MI_DEC_FORM");
  return !e ? 0 : e[0] == 'p' ? 2 : e[0] == 's' ? 1 : 0;
}()};
int set_decode_form(int form) { return g_decode_form.exchange(form, std::memory_order_relaxed); }

// Resident workgroups per CU of a decode kernel: bounded by its LDS and registers.
template <typename K>
static uint32_t resident_per_cu(K kernel, int threads) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, 0) != hipSuccess || n < 1) n = 8;
  return (uint32_t)n;
}

hipError_t launch_decode(const BatchArgs &a, bool strict, uint32_t *counter, uint32_t *order_cnt, uint32_t *order,
                         uint32_t cu_count, int form, hipStream_t s) {
  // as many workgroups as stay resident (one that is not would hold its first block back)
  static const uint32_t per_cu[4] = {resident_per_cu(decode_kernel<false, false>, 128), resident_per_cu(decode_kernel<false, true>, 128),
                                     resident_per_cu(decode_kernel<true, false>, 128), resident_per_cu(decode_kernel<true, true>, 128)};
  static const uint32_t pair_per_cu[4] = {
      resident_per_cu(decode_pair_kernel<false, false>, 128), resident_per_cu(decode_pair_kernel<false, true>, 128),
      resident_per_cu(decode_pair_kernel<true, false>, 128), resident_per_cu(decode_pair_kernel<true, true>, 128)};
  // streamed blocks behind every heavy one in the queue of a long launch (LZ4MI_DEC_MIX overrides; 0 = heavy first)
  static const int mix_env = [] {
    const char *e = getenv("LZ4MI_DEC_MIX");
    return e ? atoi(e) : -1;
  }();
  const int form_env = form >= 0 ? form : g_decode_form.load(std::memory_order_relaxed);
  const bool pre = a.dict_len != 0;
  const uint32_t v = (strict ? 2 : 0) + (pre ? 1 : 0);
  // Few blocks (every one gets a resident pair of wavefronts): two wavefronts per block, each block done sooner.
  // More blocks than that: one wavefront per block, twice as many blocks in flight, and a second one for the
  // blocks that are still running when the queue is empty.
  const bool pair = form_env ? form_env == 2 : a.nblocks <= cu_count * pair_per_cu[v];
  const uint32_t per_wg = pair ? 1u : 2u;  // blocks a workgroup of two wavefronts works on at a time
  const uint32_t slots = cu_count * (pair ? pair_per_cu[v] : per_cu[v]) * per_wg;
  const uint32_t inflight = a.nblocks < slots ? a.nblocks : slots;
  const uint32_t grid = (inflight + per_wg - 1u) / per_wg;
  if (order && a.nblocks > grid * per_wg) {  // only a queued launch has an order to choose
    // few blocks per wavefront: the launch ends with its slowest block, heavy blocks go first
    uint32_t mix = a.nblocks >= kMixFromRounds * grid * per_wg ? kMix : 0u;
    if (mix_env >= 0) mix = (uint32_t)mix_env;
    if (mix == 0 && a.nblocks <= kOrderSmallMax) {
      hipLaunchKernelGGL(decode_order_small_kernel, dim3(1), dim3(1024), 0, s, a, counter, order);
    } else {
      // a fill that is refused (a stream that cannot take it) must not be followed by a launch that pops a stale queue head
      if (hipError_t e = hipMemsetAsync(counter, 0, (size_t)((order_cnt + 2) - counter) * sizeof(uint32_t), s)) return e;
      uint32_t *tag = order + a.nblocks;
      const dim3 g((a.nblocks + 255) / 256), blk(256);
      hipLaunchKernelGGL(decode_order_count_kernel, g, blk, 0, s, a, order_cnt, tag);
      hipLaunchKernelGGL(decode_order_place_kernel, g, blk, 0, s, a.nblocks, mix, order_cnt, tag, order);
    }
  } else {
    order = nullptr;
    if (hipError_t e = hipMemsetAsync(counter, 0, sizeof(uint32_t), s)) return e;
  }
  if (hipError_t e = hipGetLastError()) return e;  // the order kernels
  dim3 g(grid), blk(128);
  if (pair) {
    if (strict) {
      if (pre) hipLaunchKernelGGL((decode_pair_kernel<true, true>), g, blk, 0, s, a, counter, order);
      else hipLaunchKernelGGL((decode_pair_kernel<true, false>), g, blk, 0, s, a, counter, order);
    } else {
      if (pre) hipLaunchKernelGGL((decode_pair_kernel<false, true>), g, blk, 0, s, a, counter, order);
      else hipLaunchKernelGGL((decode_pair_kernel<false, false>), g, blk, 0, s, a, counter, order);
    }
  } else if (strict) {
    if (pre) hipLaunchKernelGGL((decode_kernel<true, true>), g, blk, 0, s, a, counter, order);
    else hipLaunchKernelGGL((decode_kernel<true, false>), g, blk, 0, s, a, counter, order);
  } else {
    if (pre) hipLaunchKernelGGL((decode_kernel<false, true>), g, blk, 0, s, a, counter, order);
    else hipLaunchKernelGGL((decode_kernel<false, false>), g, blk, 0, s, a, counter, order);
  }
  return hipGetLastError();
}

}  // namespace lz4mi
