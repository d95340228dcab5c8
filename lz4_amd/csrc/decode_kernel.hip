// LZ4 block decoder for gfx950: one wavefront per block.
//
// Follows lz4BlockDecompress (strict) and lz4BlockDecompressInto (lenient) of
// lib/src/block/lz4_block_decoder.dart:7-119 with ByteWriter.copyMatch
// (lib/src/internal/byte_writer.dart:93-121) check for check, so the status of
// a malformed block equals the exception the reference raises.
//
// Layout: the token stream is parsed wave-uniformly out of a 1 KiB register window of the
// compressed block (four dwords per lane, fetched with one coalesced load and read back with
// v_readlane); a refill of that window is the only point where the parse waits on memory.
// Short literal runs are taken from the same registers (ds_bpermute), and the most recent
// 8 KiB of output are mirrored in an LDS ring per wavefront, so a short match whose source is
// recent is served from LDS: a typical sequence needs no memory round trip at all, only stores.
// That matters because loads, stores and atomics of a wavefront retire through one in-order
// counter on gfx950: any load issued behind output stores waits for all of them.  Paths that
// do load (far matches, long runs, window refills) end with an explicit vmcnt(0) so that the
// compiler's wait insertion never adds a drain of its own to the load-free paths.
// Long runs go memory to memory with 16-byte aligned stores (1 KiB per wave-instruction);
// matches that overlap their own output replicate the period with a modulo gather.
// Blocks that consist of short sequences (text-like data: ten bytes per sequence) are put down
// by the first kernel and finished by a second one that decodes about ten sequences per step.
#include "device_common.h"
#include "kernels.h"
#include "profile.h"

namespace lz4mi {

#ifdef LZ4MI_PROFILE
__device__ unsigned long long g_dec_phase[PH_SLOTS];
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

__device__ __forceinline__ uint32_t lane_pull(uint32_t v, uint32_t from) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(from << 2), (int)v);
}

#ifndef LZ4MI_MS_ENTER
#define LZ4MI_MS_ENTER 64
#endif
#ifndef LZ4MI_MS_MIN
#define LZ4MI_MS_MIN 3
#endif
constexpr uint32_t kMsEnter = LZ4MI_MS_ENTER;  // short sequences in a row before the first kernel puts a block down
constexpr uint32_t kMsBytes = 20;              // output bytes per sequence below which a block counts as short-sequence
constexpr uint32_t kMsPenalty = 8;             // what a sequence with a length extension takes off the score
constexpr uint32_t kMsReenter = 8;             // the same inside the second kernel, where changing loops is cheap
constexpr uint32_t kMsTry = 1;                 // GROUPS: short sequences in a row before the next try
constexpr uint32_t kMsLeave = 16;              // GROUPS: single sequences in a row before handing back
constexpr uint32_t kMsMin = LZ4MI_MS_MIN;      // smallest group worth giving up the ring for

#ifndef LZ4MI_RING
#define LZ4MI_RING 8192
#endif
constexpr uint32_t kRing = LZ4MI_RING;   // bytes of recent output mirrored in LDS per wavefront
constexpr uint32_t kShort = 256;   // copies up to this length go through the ring path

// Decodes one block with the whole wave into dst[op_start ...).  `cap` is
// decompressedSize (STRICT) or the ByteWriter maxLength minus the pre-filled prefix
// (lenient).  A match may reach `back_limit` bytes before op_start: bytes at
// non-negative positions come from dst itself (earlier blocks of a linked frame),
// negative positions from the tail of `prefix` (dictionary).  out_len = bytes written.
//
// The loop exists in two variants that hand a block back and forth at sequence boundaries, each
// starting from fresh window/ring state as if the rest of the block were a block of its own:
// GROUPS == false takes one sequence per iteration and, when HANDOVER is nonzero, returns
// kSwitchRun once its short-sequence score (+1 for a sequence without length extensions,
// -kMsPenalty for any other) reaches HANDOVER; GROUPS == true tries whole groups of short
// sequences first (see below) and returns kSwitchRun once kMsLeave sequences in a row had to go
// one by one.  (With the group code inside the only loop of the only kernel, long-match data
// decoded 5-9 % slower; the first kernel of a launch therefore contains the plain loop alone.)
constexpr int32_t kSwitchRun = -1;  // not a status: continue with the other variant at in_used / out_len
template <bool STRICT, bool HAS_PREFIX, bool GROUPS, uint32_t HANDOVER>
__device__ __forceinline__ int32_t decode_run(const uint8_t *__restrict__ src, uint32_t src_len, uint8_t *dst,
                                              const uint64_t op_start, uint32_t cap, uint32_t back_limit,
                                              const uint8_t *__restrict__ prefix, uint32_t prefix_len,
                                              lds_u8 *ring, uint32_t lane, uint32_t &in_used, uint32_t &out_len) {
  InWindow w;
  w.mis = (uint32_t)((uintptr_t)src & 3u);
  w.al = src - w.mis;
  w.vend = (w.mis + src_len + 3u) & ~3u;
  w.base = 0xfffff000u;  // nothing loaded yet
  w.r0 = w.r1 = w.r2 = w.r3 = 0;
  Stream sm;
  sm.sp = 0xffffff00u;
  sm.S = 0;

  uint32_t ip = 0, op = 0;  // op counts bytes written by this block
  uint32_t rlo = 0;         // the ring mirrors output positions [rlo, op) of this block
  uint32_t nseq = 0, mark = 0;  // plain loop with HANDOVER: sequences seen, op at the last check
  uint32_t shorts = 0;      // score: +1 per sequence without length extensions, -kMsPenalty per other one
  uint32_t idle = 0;        // GROUPS: sequences taken one by one since the last group
  const uint32_t in_end = src_len;
  int32_t st = LZ4MI_OK;
  uint8_t *const out = dst + op_start;
  PH_DECL

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
    if (!GROUPS && HANDOVER) {
      // every HANDOVER (a power of two) sequences: fewer than kMsBytes output bytes per sequence
      // over that stretch means short sequences
      if ((++nseq & (HANDOVER - 1u)) == 0u) {
        if (op - mark < HANDOVER * kMsBytes) { st = kSwitchRun; break; }
        mark = op;
        // a long block decides when the launch ends: it takes issue priority over its neighbours
        if (nseq == 64) __builtin_amdgcn_s_setprio(1);
        if (nseq == 192) __builtin_amdgcn_s_setprio(2);
        if (nseq == 448) __builtin_amdgcn_s_setprio(3);
      }
    }
    uint32_t token = sbyte(w, sm, ip, lane);
    ip++;
    const bool shortseq = (token & 0xF0u) != 0xF0u && (token & 15u) != 15u;  // no length extensions
    uint64_t lit64 = token >> 4;
    if (lit64 == 15) {
      if (!read_ext(w, sm, ip, in_end, lit64, lane)) { st = LZ4MI_FORMAT_EOF; break; }
    }
    if (STRICT && lit64 > (uint64_t)(cap - op)) { st = LZ4MI_CORRUPT_LITERAL; break; }  // :33-35
    if (lit64 != 0) {
      if (lit64 > (uint64_t)(in_end - ip)) { st = LZ4MI_FORMAT_EOF; break; }  // readBytesView
      if (!STRICT && lit64 > (uint64_t)(cap - op)) { st = LZ4MI_LIMIT; break; } // writer maxLength
      const uint32_t lit = (uint32_t)lit64;
      PH(0);
      if (ip - sm.sp + lit <= kWave) {
        // the whole run is in the stream register: lanes [rel, rel + lit) store their byte
        const uint32_t rel = ip - sm.sp;
        const uint32_t k = lane - rel;
        if (k < lit) {
          out[op + k] = (uint8_t)sm.S;
          ring[(op + k) & (kRing - 1)] = (uint8_t)sm.S;
        }
        if (op + lit > rlo + kRing) rlo = op + lit - kRing;
        PH(1); PH_CNT(8);
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
        PH(1); PH_CNT(14);
      } else {
        wave_copy(out + op, src + ip, lit, lane);
        vm_quiesce();
        rlo = op + lit;  // ring holds nothing older than this
        PH(2); PH_CNT(9);
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
    PH(6);
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
      if (s >= ring_from) { PH(3); PH_CNT(10); } else { PH(4); PH_CNT(11); }
    } else {
      copy_match<HAS_PREFIX>(dst, op_start + op, dist, ml, prefix, prefix_len, lane);
      rlo = op + ml;
      PH(5); PH_CNT(13);
    }
    op += ml;
    if (!GROUPS) continue;
    shorts = shortseq ? shorts + 1u : (shorts > kMsPenalty ? shorts - kMsPenalty : 0u);
    if (++idle > kMsLeave && ip < in_end && op < cap) { st = kSwitchRun; break; }
    while (shorts >= kMsTry && ip < in_end && op < cap) {  // else the checks at the top decide
      // ---- several short sequences at once.  Data made of short sequences (text) spends its time
      // on per-sequence latency, above all the load of a match source that is older than the ring.
      // Here every lane parses the byte it holds as if it were a token; following the chain of
      // real token starts from lane 0 is a handful of scalar steps; a prefix sum places every
      // sequence in the output; then all literal bytes of the group go out in one store, all match
      // sources are fetched with one round trip and all matches are written together.  A sequence
      // joins the group only if it has no length extension, is not the last of the block, passes
      // the reference's checks, and its source lies wholly in output written before the group
      // (18 bytes from the source start, so the fixed-size fetch never reads ahead of the data);
      // the first sequence that does not qualify ends the group and goes through the general path.
      if (sm.sp != ip) stream_load(w, sm, ip, lane);
      const uint32_t avail_in = in_end - ip;
      const uint32_t ll = sm.S >> 4, ml = (sm.S & 15u) + 4u;
      const uint32_t nx = lane + 3u + ll;  // where the next token would start
      const uint32_t off = lane_pull(sm.S, lane + 1u + ll) | (lane_pull(sm.S, lane + 2u + ll) << 8);
      const bool ok = ll < 15u && ml < 19u && nx <= kWave && nx < avail_in && off != 0u;
      const uint64_t okm = ballot(ok);
      uint64_t starts = 0;
      uint32_t r = 0;
      while (r < kWave && ((okm >> r) & 1ull)) {
        starts |= 1ull << r;
        r = rdlane(nx, r);
      }
      if (starts) {
        bool is_start = (starts >> lane) & 1ull;
        const uint32_t mine = is_start ? ll + ml : 0u;
        const uint32_t incl = wave_incl_scan(mine);
        const uint32_t excl = incl - mine;
        const uint32_t mrel = excl + ll;  // match position relative to op
        const int64_t srcpos = (int64_t)op_start + op + mrel - off;
        const bool bad = is_start && ((uint64_t)off > (uint64_t)op + mrel + back_limit ||  // copyMatch's distance check
                                      off < mrel + 18u || srcpos < 0);
        const uint64_t bm = ballot(bad);
        uint32_t total = rdlane(incl, kWave - 1);
        if (bm) {
          const uint32_t v = (uint32_t)__builtin_ctzll(bm);
          starts &= (1ull << v) - 1ull;
          r = v;
          total = rdlane(excl, v);
          is_start = (starts >> lane) & 1ull;
        }
        if ((uint32_t)__builtin_popcountll(starts) >= kMsMin && total <= cap - op) {
          uint64_t v0 = 0, v1 = 0;
          uint32_t v2 = 0;
          if (is_start) {  // sources first: they queue behind the previous group's stores only
            const uint8_t *sp8 = dst + srcpos;
            v0 = ld64(sp8);
            v1 = ld64(sp8 + 8);
            v2 = ld16(sp8 + 16);
          }
          {  // every literal byte of the group: lane j belongs to the nearest start at or below it
            const uint64_t below = starts & (~0ull >> (63u - lane));
            const uint32_t k = 63u - (uint32_t)__builtin_clzll(below | 1ull);
            const uint32_t llk = lane_pull(ll, k), exk = lane_pull(excl, k);
            const uint32_t li = lane - k - 1u;
            if (lane < r && li < llk) out[op + exk + li] = (uint8_t)sm.S;
          }
          if (is_start) {
            uint8_t *d = out + op + mrel;
            if (ml >= 16u) {
              st64(d, v0);
              st64(d + 8, v1);
              if (ml == 18u) st16(d + 16, v2);
              if (ml == 17u) d[16] = (uint8_t)v2;
            } else {
              uint64_t cur = v0;
              if (ml & 8u) { st64(d, v0); d += 8; cur = v1; }
              if (ml & 4u) { st32(d, (uint32_t)cur); d += 4; cur >>= 32; }
              if (ml & 2u) { st16(d, (uint32_t)cur); d += 2; cur >>= 16; }
              if (ml & 1u) d[0] = (uint8_t)cur;
            }
          }
          ip += r;
          op += total;
          rlo = op;  // the ring holds nothing of this group
          idle = 0;
          continue;
        }
      }
      shorts = 0;
    }
  }
  PH(0);
  PH_FLUSH(g_dec_phase, lane);
  in_used = ip;
  out_len = op;
  return st;
}

template <bool STRICT, bool HAS_PREFIX, bool ALLOW_GROUPS>
__device__ __forceinline__ int32_t decode_block(const uint8_t *__restrict__ src, uint32_t src_len, uint8_t *dst,
                                                const uint64_t op_start, uint32_t cap, uint32_t back_limit,
                                                const uint8_t *__restrict__ prefix, uint32_t prefix_len,
                                                lds_u8 *ring, uint32_t lane, uint32_t &out_len) {
  if (!ALLOW_GROUPS) {  // the plain loop alone, start to end
    uint32_t used = 0;
    return decode_run<STRICT, HAS_PREFIX, false, 0>(src, src_len, dst, op_start, cap, back_limit, prefix,
                                                        prefix_len, ring, lane, used, out_len);
  }
  uint32_t ip = 0, op = 0;
  int32_t st;
  for (bool groups = true;; groups = !groups) {  // callers come here because short sequences were seen
    uint32_t di = 0, dn = 0;
    st = groups ? decode_run<STRICT, HAS_PREFIX, true, 1>(src + ip, src_len - ip, dst, op_start + op, cap - op,
                                                              back_limit + op, prefix, prefix_len, ring, lane, di, dn)
                : decode_run<STRICT, HAS_PREFIX, false, kMsReenter>(src + ip, src_len - ip, dst, op_start + op, cap - op,
                                                               back_limit + op, prefix, prefix_len, ring, lane, di,
                                                               dn);
    ip += di;
    op += dn;
    if (st != kSwitchRun) break;
  }
  out_len = op;
  return st;
}

// A launch is two kernels on the stream.  decode_kernel takes every block through the plain
// loop; a block that turns out to consist of short sequences is put down at the sequence boundary
// it has reached (block index, bytes consumed, bytes produced go into the redo list) and the
// wavefront moves on.  decode_redo_kernel then finishes the listed blocks with the group variant.
// Long-match data never pays for the group code (it is not in the first kernel at all); when the
// list stays empty the second kernel returns at once.
struct RedoEntry { uint32_t block, in_used, out_len, pad; };

// In the second kernel the launch arguments are not held in registers across a block: its loops
// keep their control state in scalar registers and need nearly all of them (holding the
// arguments too made the compiler spill scalars in the sequence loop).  They are read again
// from the kernel argument segment (BatchArgs first, then the other arguments in order).
struct DecodeKArgs {
  BatchArgs a;
  uint32_t *counter;
  uint32_t *redo_count;
  RedoEntry *redo;
  const uint32_t *order;
};
typedef const volatile __attribute__((address_space(4))) DecodeKArgs *DecodeKArgsPtr;

struct BlockIo {
  const uint8_t *src;
  uint32_t src_len;
  uint8_t *dst;
  uint32_t cap;
};
__device__ __forceinline__ BlockIo block_io(DecodeKArgsPtr ka, uint32_t b) {
  BlockIo io;
  const uint64_t *src_off = ka->a.src_off, *dst_off = ka->a.dst_off;
  const uint32_t *src_len_v = ka->a.src_len, *dst_cap = ka->a.dst_cap;
  io.src = ka->a.src + uni64(src_off ? src_off[b] : (uint64_t)b * ka->a.src_stride);
  if (src_len_v) {
    io.src_len = uni(src_len_v[b]);
  } else {
    const uint32_t u = ka->a.src_len_uniform;
    const uint64_t start = (uint64_t)b * u, tot = ka->a.src_total;
    const uint64_t left = tot > start ? tot - start : 0;
    io.src_len = left < u ? (uint32_t)left : u;
  }
  io.dst = ka->a.dst + uni64(dst_off ? dst_off[b] : (uint64_t)b * ka->a.dst_stride);
  io.cap = uni(dst_cap ? dst_cap[b] : ka->a.dst_cap_uniform);
  return io;
}

template <bool STRICT, bool HAS_PREFIX>
__global__ __launch_bounds__(256) void decode_kernel(BatchArgs a, uint32_t *counter, uint32_t *redo_count,
                                                     RedoEntry *redo, const uint32_t *order) {
  __shared__ __attribute__((aligned(16))) uint8_t s_ring[4][kRing];
  const uint32_t lane = lane_id();
  lds_u8 *ring = (lds_u8 *)s_ring[uni(threadIdx.x / kWave)];
  // the decoder keeps the last 65536 bytes of a dictionary (lz4_frame_decoder.dart:278-282)
  const uint8_t *prefix = a.dict;
  const uint32_t prefix_len = a.dict_len;
  // When the launch has a wavefront for every block, a wavefront's block is its own index (thousands
  // of wavefronts starting at once would otherwise serialise on the queue head); larger batches
  // are handed out through the queue.
  const uint32_t nwaves = gridDim.x * (blockDim.x / kWave);
  const bool direct = a.nblocks <= nwaves;
  // a wavefront's first block is its own index (thousands of wavefronts popping one queue head at
  // launch serialise on it); the queue hands out the blocks behind those
  uint32_t b = uni(blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave);
  for (;; b = direct ? 0xffffffffu : nwaves + next_block(counter, lane)) {
    if (b >= a.nblocks) break;
    if (order) b = uni(order[b]);  // queue position -> block (long blocks first)
    const uint8_t *src = a.src + uni64(a.src_off ? a.src_off[b] : (uint64_t)b * a.src_stride);
    const uint32_t src_len = uni(block_src_len(a, b));
    uint8_t *dst = a.dst + uni64(a.dst_off ? a.dst_off[b] : (uint64_t)b * a.dst_stride);
    const uint32_t cap = uni(a.dst_cap ? a.dst_cap[b] : a.dst_cap_uniform);
    uint32_t used = 0, out_len = 0;
    __builtin_amdgcn_s_setprio(0);
#ifdef LZ4MI_PROFILE
    if (lane == 0 && b < 65536) g_dec_span[2 * b] = __builtin_amdgcn_s_memrealtime();
#endif
    const int32_t st = decode_run<STRICT, HAS_PREFIX, false, kMsEnter>(src, src_len, dst, 0, cap, prefix_len, prefix,
                                                                   prefix_len, ring, lane, used, out_len);
#ifdef LZ4MI_PROFILE
    if (lane == 0 && b < 65536) g_dec_span[2 * b + 1] = __builtin_amdgcn_s_memrealtime();
#endif
    if (lane == 0) {
      if (st == kSwitchRun) {
        const uint32_t k = atomicAdd(redo_count, 1u);
        RedoEntry e = {b, used, out_len, 0u};
        redo[k] = e;
      } else {
        a.out_len[b] = out_len;
        a.status[b] = st;
      }
    }
  }
}

template <bool STRICT, bool HAS_PREFIX>
__global__ __launch_bounds__(256) void decode_redo_kernel(BatchArgs, uint32_t *, uint32_t *, RedoEntry *, const uint32_t *) {
  __shared__ __attribute__((aligned(16))) uint8_t s_ring[4][kRing];
  const uint32_t lane = lane_id();
  lds_u8 *ring = (lds_u8 *)s_ring[uni(threadIdx.x / kWave)];
  DecodeKArgsPtr ka = (DecodeKArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
  const uint32_t nredo = *ka->redo_count;
  if (nredo == 0) return;
  const uint32_t nwaves = gridDim.x * (blockDim.x / kWave);
  const bool direct = nredo <= nwaves;
  uint32_t k = direct ? uni(blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave)
                      : next_block(ka->counter + 1, lane);  // the second kernel's own queue head
  for (;; k = direct ? 0xffffffffu : next_block(ka->counter + 1, lane)) {
    if (k >= nredo) break;
    const uint32_t b = uni(ka->redo[k].block), ip = uni(ka->redo[k].in_used), op = uni(ka->redo[k].out_len);
    const BlockIo io = block_io(ka, b);
    const uint8_t *prefix = ka->a.dict;
    const uint32_t prefix_len = ka->a.dict_len;
    uint32_t more = 0;
    const int32_t st = decode_block<STRICT, HAS_PREFIX, true>(io.src + ip, io.src_len - ip, io.dst, op, io.cap - op,
                                                              prefix_len + op, prefix, prefix_len, ring, lane, more);
    if (lane == 0) {
      ka->a.out_len[b] = op + more;
      ka->a.status[b] = st;
    }
  }
}

// Linked ("dependent") blocks: block b may reference the 64 KiB that precede it in the
// frame output, topped up from the dictionary while the output is shorter than that
// (_decodeDependentBlock, lib/src/frame/lz4_frame_decoder.dart:326-382).  The chain is
// serial, so one wavefront walks the blocks in order over one contiguous output.
__global__ __launch_bounds__(64) void decode_chain_kernel(ChainArgs c) {
  __shared__ __attribute__((aligned(16))) uint8_t s_ring[kRing];
  lds_u8 *ring = (lds_u8 *)s_ring;
  const uint32_t lane = lane_id();
  uint64_t start = 0;
  uint32_t b = 0;
  for (; b < c.nblocks; b++) {
    const uint8_t *payload = c.src + c.payload_off[b];
    const uint32_t len = c.payload_len[b];
    uint32_t produced = 0;
    int32_t st = LZ4MI_OK;
    if (c.raw[b]) {
      wave_copy(c.dst + start, payload, len, lane);
      produced = len;
    } else {
      uint32_t back = start < 65536u ? (uint32_t)start : 65536u;
      if (back < 65536u && c.dict_len) back += c.dict_len < 65536u - back ? c.dict_len : 65536u - back;
      st = decode_block<false, true, true>(payload, len, c.dst, start, c.block_max, back, c.dict, c.dict_len, ring, lane,
                                     produced);
    }
    if (lane == 0) {
      c.out_len[b] = produced;
      c.status[b] = st;
    }
    if (st != LZ4MI_OK) { b++; break; }
    start += produced;
  }
  for (uint32_t k = b + lane; k < c.nblocks; k += kWave) {  // blocks after an error never ran
    c.out_len[k] = 0;
    c.status[k] = LZ4MI_OK;
  }
}

#ifdef LZ4MI_PROFILE
extern "C" void lz4mi_debug_dec_spans(unsigned long long *out, int n) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dec_span), sizeof(unsigned long long) * 2 * n);
}
extern "C" void lz4mi_debug_dec_phases(unsigned long long *out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dec_phase), sizeof(unsigned long long) * PH_SLOTS);
  if (reset) {
    unsigned long long z[PH_SLOTS] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dec_phase), z, sizeof z);
  }
}
#endif

void launch_decode_chain(const ChainArgs &c, hipStream_t s) {
  if (!c.nblocks) return;
  hipLaunchKernelGGL(decode_chain_kernel, dim3(1), dim3(64), 0, s, c);
}

// A launch ends when its longest block does, and the queue hands blocks out in order: a long block
// far down the queue starts late.  What a block costs the decoder shows in its size: literal-only
// blocks (compressed >= about the output size) and run-like blocks (a few bytes) are streamed, the
// ones in between are sequence after sequence on one wavefront.  Those go to the front of the queue.
__global__ __launch_bounds__(256) void decode_order_kernel(BatchArgs a, uint32_t *cnt, uint32_t *order) {
  const uint32_t lane = lane_id();
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = b < a.nblocks;
  bool heavy = false;
  if (valid) {
    const uint64_t c = block_src_len(a, b), n = a.dst_cap ? a.dst_cap[b] : a.dst_cap_uniform;
    heavy = c * 16 < n * 15 && c * 32 > n;
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
  if (heavy) order[bh + (uint32_t)__builtin_popcountll(hm & below)] = b;
  else if (valid) order[a.nblocks - 1u - (bl + (uint32_t)__builtin_popcountll(lm & below))] = b;
}

void launch_decode(const BatchArgs &a, bool strict, uint32_t *counter, uint32_t *redo_count, void *redo,
                   uint32_t *order_cnt, uint32_t *order, uint32_t grid, hipStream_t s) {
  if (order && a.nblocks > grid * 4u)  // only a queued launch has an order to choose
    hipLaunchKernelGGL(decode_order_kernel, dim3((a.nblocks + 255) / 256), dim3(256), 0, s, a, order_cnt, order);
  else
    order = nullptr;
  dim3 g(grid), blk(256);
  RedoEntry *r = (RedoEntry *)redo;
  const bool pre = a.dict_len != 0;
  if (strict) {
    if (pre) {
      hipLaunchKernelGGL((decode_kernel<true, true>), g, blk, 0, s, a, counter, redo_count, r, order);
      hipLaunchKernelGGL((decode_redo_kernel<true, true>), g, blk, 0, s, a, counter, redo_count, r, order);
    } else {
      hipLaunchKernelGGL((decode_kernel<true, false>), g, blk, 0, s, a, counter, redo_count, r, order);
      hipLaunchKernelGGL((decode_redo_kernel<true, false>), g, blk, 0, s, a, counter, redo_count, r, order);
    }
  } else {
    if (pre) {
      hipLaunchKernelGGL((decode_kernel<false, true>), g, blk, 0, s, a, counter, redo_count, r, order);
      hipLaunchKernelGGL((decode_redo_kernel<false, true>), g, blk, 0, s, a, counter, redo_count, r, order);
    } else {
      hipLaunchKernelGGL((decode_kernel<false, false>), g, blk, 0, s, a, counter, redo_count, r, order);
      hipLaunchKernelGGL((decode_redo_kernel<false, false>), g, blk, 0, s, a, counter, redo_count, r, order);
    }
  }
}

}  // namespace lz4mi
