// Fast LZ4 block encoder for gfx950: one wavefront per block, byte-identical to
// lz4BlockCompressToWriter (lib/src/block/lz4_block_encoder.dart:35-157).
//
// The reference parse is a serial greedy scan, but between two matches its
// probe positions are a fixed schedule  i_{k+1} = i_k + 1 + ((i_k - anchor) >> accel)
// (:150).  A wavefront therefore evaluates a whole run of probes at once:
// lane k gathers the 4 bytes at i_k, hashes them ((seq * 2654435761) >> 16, :167-170),
// gathers the table entry, gathers the candidate's 4 bytes and votes; the first
// hitting lane is the reference's next match.  Two facts keep this exact:
//   * the table only ever holds the LARGEST position inserted per bucket (every
//     insert of :100 and :139-145 is at a position larger than what the bucket
//     held), so inserts are unordered atomic-max operations;
//   * probes of the same run that share a bucket see each other in lane order
//     (resolved in registers before the vote).
// Inside a match of distance d only the last d positions can survive in the
// table (position j and j+d hash the same bytes and j+d is inserted later), so
// the interior re-hash (:139-145) touches min(length-4, d) positions.
//
// Table: 65536 x u32 per wavefront slot in HBM/L2, entry = generation<<24 | (pos+1);
// a stale generation reads as empty, so a slot is reused without clearing.
// A shared dictionary is primed once into a read-only table consulted when the
// slot entry is empty (dictionary positions are always smaller than block positions).
#include "encode_common.h"
#include "profile.h"

namespace lz4mi {

#ifdef LZ4MI_PROFILE
__device__ unsigned long long g_enc_phase[PH_SLOTS];
#endif

// ---------------------------------------------------------------- input window
// A 4 KiB ring of the block's bytes around the current anchor lives in LDS (one ring per
// wavefront).  Probe reads near the anchor, the anchor side of match extension, literal
// copies and the interior re-hash are served from it (LDS latency) instead of L2/HBM round
// trips; the next 1 KiB segment is fetched into registers one run ahead and written to the
// ring at the start of the following run.
constexpr uint32_t kWinBytes = 4096, kSeg = 1024;

struct Win {
  lds_u8 *lds;         // this wavefront's ring (16-byte aligned)
  const uint8_t *src;  // block bytes
  uint32_t n;          // block length
  uint32_t lo, hi;     // ring holds block offsets [lo, hi); hi may overshoot n (zero padded)
  uint32_t pend_pos;
  bool pending;
  U128U pend;
};

__device__ __forceinline__ void win_issue(Win &w, uint32_t pos, uint32_t lane) {
  uint32_t o = pos + 16 * lane;
  if (o + 16 <= w.n) {
    w.pend = ld128(w.src + o);
  } else {  // last segment of the block: byte-wise, zero padded
    uint32_t v[4] = {0, 0, 0, 0};
    for (uint32_t k = 0; k < 16; k++)
      if (o + k < w.n) v[k >> 2] |= (uint32_t)w.src[o + k] << (8 * (k & 3));
    w.pend = U128U{v[0], v[1], v[2], v[3]};
  }
  w.pend_pos = pos;
  w.pending = true;
}

__device__ __forceinline__ void win_commit(Win &w, uint32_t lane) {
  *(lds_u128 *)(w.lds + ((w.pend_pos + 16 * lane) & (kWinBytes - 1))) =
      u32x4{w.pend.a, w.pend.b, w.pend.c, w.pend.d};
  w.hi = w.pend_pos + kSeg;
  if (w.hi - w.lo > kWinBytes) w.lo = w.hi - kWinBytes;
  w.pending = false;
}

// Run start: makes [s, s + 256) resident (clipped to the block).  Normally the ring is already
// further ahead and nothing happens here; the in-flight segment is only waited for when needed.
__device__ __forceinline__ void win_prepare(Win &w, uint32_t s, uint32_t lane) {
  if (s >= w.lo && s <= w.hi && (w.hi >= s + 256 || w.hi >= w.n)) return;
  if (w.pending) win_commit(w, lane);
  if (s < w.lo || s > w.hi) w.lo = w.hi = s & ~(kSeg - 1);  // jumped out of the ring
  while (w.hi < s + 256 && w.hi < w.n) {
    win_issue(w, w.hi, lane);
    win_commit(w, lane);
  }
}

// Called right after a point where the wavefront has waited on memory anyway: lands the
// segment that was in flight and requests the next one, keeping the ring ~2 KiB ahead of s.
__device__ __forceinline__ void win_service(Win &w, uint32_t s, uint32_t lane) {
  if (w.pending) win_commit(w, lane);
  if (w.hi < w.n && w.hi < s + 2 * kSeg) win_issue(w, w.hi, lane);  // keeps lo <= s - 1 KiB after it lands
}

__device__ __forceinline__ bool win_has(const Win &w, uint32_t s, uint32_t len) {
  uint32_t top = w.hi < w.n ? w.hi : w.n;
  return s >= w.lo && s + len <= top;
}
__device__ __forceinline__ uint32_t win_ld32(const Win &w, uint32_t s) {
  uint32_t r = s & (kWinBytes - 1), a0 = r & ~3u, a1 = (a0 + 4) & (kWinBytes - 1);
  uint32_t lo = *(const lds_u32 *)(w.lds + a0), hi = *(const lds_u32 *)(w.lds + a1);
  return __builtin_amdgcn_alignbyte(hi, lo, r & 3u);
}
__device__ __forceinline__ uint32_t win_u8(const Win &w, uint32_t s) { return w.lds[s & (kWinBytes - 1)]; }

// position-based reads of dict||src: ring when resident, memory otherwise
template <bool HAS_DICT>
__device__ __forceinline__ uint32_t rd32(const EncIn &in, const Win &w, uint32_t pos) {
  const uint32_t D = HAS_DICT ? in.D : 0u;
  if (pos >= D && win_has(w, pos - D, 4)) return win_ld32(w, pos - D);
  return in_ld32<HAS_DICT>(in, pos);
}
template <bool HAS_DICT>
__device__ __forceinline__ uint32_t rd8(const EncIn &in, const Win &w, uint32_t pos) {
  const uint32_t D = HAS_DICT ? in.D : 0u;
  if (pos >= D && win_has(w, pos - D, 1)) return win_u8(w, pos - D);
  return in_byte<HAS_DICT>(in, pos);
}

// Number of equal leading bytes (0..8) of the 8 bytes at p and q, never reading at or past `total`.
template <bool HAS_DICT>
__device__ __forceinline__ uint32_t eq8(const EncIn &in, const Win &w, uint32_t p, uint32_t q, uint32_t total) {
  uint32_t avail = p < total ? (total - p < 8 ? total - p : 8u) : 0u;
  if (avail == 8) {  // all four reads are issued before the first compare: one round trip
    uint32_t a0 = rd32<HAS_DICT>(in, w, p), b0 = rd32<HAS_DICT>(in, w, q);
    uint32_t a1 = rd32<HAS_DICT>(in, w, p + 4), b1 = rd32<HAS_DICT>(in, w, q + 4);
    uint32_t x0 = a0 ^ b0, x1 = a1 ^ b1;
    return x0 ? ((uint32_t)__builtin_ctz(x0) >> 3) : (x1 ? 4u + ((uint32_t)__builtin_ctz(x1) >> 3) : 8u);
  }
  uint32_t eq = 0;
  for (uint32_t k = 0; k < avail; k++) {
    if (rd8<HAS_DICT>(in, w, p + k) != rd8<HAS_DICT>(in, w, q + k)) break;
    eq++;
  }
  return eq;
}

template <bool HAS_DICT>
__device__ __forceinline__ int32_t encode_block_fast(const EncIn in, uint8_t *out, uint32_t cap, uint32_t sh,
                                                     uint32_t *T, const uint32_t *__restrict__ dictT,
                                                     uint32_t gen, uint32_t offA, uint32_t offB, lds_u8 *lds,
                                                     uint32_t lane, uint32_t &out_len) {
  const uint32_t D = HAS_DICT ? in.D : 0u;
  const uint32_t total = in.total;
  const uint32_t n = total - D;
  out_len = 0;
  if (n == 0) return LZ4MI_OK;  // :45-48
  if ((uint64_t)cap < (uint64_t)n + n / 255u + 16u) return LZ4MI_DST_TOO_SMALL;
  if (total > kPosMask - 1) return LZ4MI_UNSUPPORTED_HERE;  // table positions are 24-bit
  const uint32_t gtag = gen << kPosBits;
  uint32_t anchor = D, op = 0;
  Win w;
  w.lds = lds;
  w.src = in.src;
  w.n = n;
  w.lo = w.hi = 0;
  w.pending = false;
  w.pend_pos = 0;
  PH_DECL

  if (n >= 4) {  // :74-78
    while (anchor + 4 <= total) {
      bool matched = false;
      uint32_t o_last = 0;
      win_prepare(w, anchor - D, lane);
      for (uint32_t chunk = 0;; chunk++) {
        uint32_t o, nl;
        if (chunk == 0) {
          o = offA; nl = 8;
        } else if (chunk == 1) {
          o = offB; nl = kWave;
        } else {
          o = o_last; nl = kWave;
          for (uint32_t t = 0; t < kWave; t++)
            if (t <= lane) o = sched_next(o, sh);
        }
        uint32_t ik = anchor + o;
        bool valid = lane < nl && ik + 4 <= total;
        const uint64_t vmask = ballot(valid);  // valid lanes form a prefix
        uint32_t nvalid = (uint32_t)__builtin_popcountll(vmask);
        if (nvalid == 0) break;
        PH(0); PH_CNT(8);
        uint32_t seq = 0, h = 0;
        int32_t ref = -1;
        if (valid) {
          seq = rd32<HAS_DICT>(in, w, ik);
          h = hash16(seq);
          uint32_t e = tbl_load(T + h);
          if ((e >> kPosBits) == gen) ref = (int32_t)(e & kPosMask) - 1;
          else if (HAS_DICT) ref = (int32_t)dictT[h] - 1;
        }
#ifdef LZ4MI_PROFILE
        { uint64_t w_ = ballot(ref >= 0); asm volatile("" ::"s"(w_)); }
#endif
        PH(1);
        // Earlier probes of this chunk that land in the same bucket act as the table entry
        // (:99-100 in program order): the nearest earlier lane with an equal hash wins.
#ifdef LZ4MI_ABL_NOCOLL
        if (false) {
#else
        if (nvalid <= 12) {
#endif
          for (uint32_t j = 0; j + 1 < nvalid; j++) {
            uint32_t hj = rdlane(h, j);
            if (ballot(valid && lane > j && h == hj)) {
              uint32_t ij = rdlane(ik, j);
              if (valid && lane > j && h == hj) ref = (int32_t)ij;
            }
          }
#ifdef LZ4MI_ABL_NOCOLL
        } else if (false) {
#else
        } else {
#endif
          uint64_t peers = vmask & ((1ull << lane) - 1ull);
          for (uint32_t bit = 0; bit < 16; bit++) {
            bool set = (h >> bit) & 1u;
            uint64_t m = ballot(set);
            peers &= set ? m : ~m;
          }
          if (!valid) peers = 0;
          uint32_t j = peers ? 63u - (uint32_t)__builtin_clzll(peers) : lane;
          uint32_t ij = (uint32_t)__shfl((int)ik, (int)j);
          if (peers) ref = (int32_t)ij;
        }
        PH(2);
        bool hit = false;
        if (valid && ref >= 0 && ik - (uint32_t)ref <= 0xFFFFu)  // :102-103
          hit = rd32<HAS_DICT>(in, w, (uint32_t)ref) == seq;
        uint64_t hm = ballot(hit);
        win_service(w, anchor - D, lane);
        uint32_t K = hm ? (uint32_t)__builtin_ctzll(hm) : nvalid - 1;
        if (valid && lane <= K) tbl_max(T + h, gtag | (ik + 1));  // :100 for every probe up to the hit
        PH(3);
        if (hm) {
          PH_CNT(9);
          const uint32_t i = rdlane(ik, K);
          const uint32_t r0 = rdlane((uint32_t)ref, K);
          const uint32_t dist = i - r0;
          // One round trip for both directions: lanes 0-31 compare one byte each going
          // backwards (:107-113), lanes 32-63 compare 8 bytes each going forwards (:117-121).
          const uint32_t maxb = i - anchor < r0 ? i - anchor : r0;
          bool stop;
          uint32_t eq = 0;
          if (lane < 32) {
            stop = !(lane < maxb && rd8<HAS_DICT>(in, w, i - 1 - lane) == rd8<HAS_DICT>(in, w, r0 - 1 - lane));
          } else {
            uint32_t pp = i + 4 + 8 * (lane - 32);
            eq = eq8<HAS_DICT>(in, w, pp, pp - dist, total);
            stop = eq < 8;
          }
          const uint64_t sm = ballot(stop);
          uint32_t b = (uint32_t)sm ? (uint32_t)__builtin_ctz((uint32_t)sm) : 32u;
          while (b >= 32 && b < maxb) {  // long backward run: 64 bytes per step
            uint32_t t = b + lane;
            bool same = t < maxb && rd8<HAS_DICT>(in, w, i - 1 - t) == rd8<HAS_DICT>(in, w, r0 - 1 - t);
            uint64_t m = ballot(!same);
            if (m) { b += (uint32_t)__builtin_ctzll(m); break; }
            b += kWave;
          }
          const uint32_t ms = i - b;
          PH(4);
          uint32_t fwd;
          if ((uint32_t)(sm >> 32)) {
            uint32_t L = (uint32_t)__builtin_ctz((uint32_t)(sm >> 32));
            fwd = 8 * L + rdlane(eq, 32 + L);
          } else {  // longer than 256 bytes: 512 bytes per step
            fwd = 256;
            for (;;) {
              uint32_t pp = i + 4 + fwd + 8 * lane;
              uint32_t e8 = eq8<HAS_DICT>(in, w, pp, pp - dist, total);
              uint64_t m = ballot(e8 < 8);
              if (m) {
                uint32_t L = (uint32_t)__builtin_ctzll(m);
                fwd += 8 * L + rdlane(e8, L);
                break;
              }
              fwd += 8 * kWave;
            }
          }
          const uint32_t ml = b + 4 + fwd;
          const uint32_t lit = ms - anchor;
          PH(5);
          // _writeSequence :172-200
          const uint32_t m4 = ml - 4;
#ifndef LZ4MI_ABL_NOEMIT
          if (lane == 0) out[op] = (uint8_t)(((lit < 15 ? lit : 15u) << 4) | (m4 < 15 ? m4 : 15u));
          op++;
          if (lit >= 15) op = emit_len(out, op, lit - 15, lane);
          if (lit) {
            if (lit <= 256 && win_has(w, anchor - D, lit)) {
              for (uint32_t k = lane; k < lit; k += kWave) out[op + k] = (uint8_t)win_u8(w, anchor - D + k);
            } else {
              wave_copy(out + op, in.src + (anchor - D), lit, lane);
            }
          }
          op += lit;
          if (lane == 0) {
            out[op] = (uint8_t)dist;
            out[op + 1] = (uint8_t)(dist >> 8);
          }
          op += 2;
          if (m4 >= 15) op = emit_len(out, op, m4 - 15, lane);
#endif
          anchor = ms + ml;  // :132-133
          matched = true;
          PH(6);
#ifndef LZ4MI_ABL_NOREHASH
          if (anchor + 4 <= total) {  // :135-145 interior re-hash, last `dist` positions only
            uint32_t hi = anchor - 4;
            uint32_t lo = ms + 1;
            if (anchor >= 3 + dist && anchor - 3 - dist > lo) lo = anchor - 3 - dist;
            for (uint32_t j0 = lo; j0 <= hi; j0 += kWave) {
              const uint32_t j = j0 + lane;
              const bool live = j <= hi;
              uint32_t hj = live ? hash16(rd32<HAS_DICT>(in, w, j)) : 0u;
              if (live) tbl_max(T + hj, gtag | (j + 1));
            }
          }
#endif
          PH(7);
          break;
        }
        if (nvalid < nl) break;  // schedule ran past total-4 without a hit
        o_last = rdlane(o, nl - 1);
      }
      if (!matched) break;
    }
  }
  // _writeLastLiterals :153-156, :202-214
  const uint32_t last = total - anchor;
  if (last != 0) {
    if (lane == 0) out[op] = (uint8_t)((last < 15 ? last : 15u) << 4);
    op++;
    if (last >= 15) op = emit_len(out, op, last - 15, lane);
    wave_copy(out + op, in.src + (anchor - D), last, lane);
    op += last;
  }
  PH(10);
  PH_FLUSH(g_enc_phase, lane);
  out_len = op;
  return LZ4MI_OK;
}

template <bool HAS_DICT>
__global__ __launch_bounds__(256) void encode_fast_kernel(BatchArgs a, uint32_t sh, EncodeScratch sc,
                                                          uint32_t *counter) {
  __shared__ __attribute__((aligned(16))) uint8_t s_win[4][kWinBytes];
  const uint32_t lane = lane_id();
  const uint32_t wave = uni(threadIdx.x / kWave);
  const uint32_t slot = uni(blockIdx.x * (blockDim.x / kWave) + wave);
  uint32_t *T = sc.tables + (size_t)slot * kHashEntries;
  uint32_t gen = uni(sc.slot_gen[slot]);

  // probe schedule: offA = off(lane), offB = off(lane + 8)
  uint32_t offA = 0, offB = 0;
  {
    uint32_t o = 0;
    for (uint32_t t = 0; t < kWave + 8; t++) {
      if (t == lane) offA = o;
      if (t == lane + 8) offB = o;
      o = sched_next(o, sh);
    }
  }

  EncIn in;
  in.dict = a.dict;
  in.D = a.dict_len;
  for (;;) {
    const uint32_t b = next_block(counter, lane);
    if (b >= a.nblocks) break;
    if (++gen > kGenMax) {  // generation wrapped: clear the slot
      for (uint32_t k = lane * 4; k < kHashEntries; k += kWave * 4) *(uint4 *)(T + k) = make_uint4(0, 0, 0, 0);
      gen = 1;
    }
    in.src = a.src + uni64(a.src_off ? a.src_off[b] : (uint64_t)b * a.src_stride);
    in.total = in.D + uni(block_src_len(a, b));
    uint8_t *out = a.dst + uni64(a.dst_off ? a.dst_off[b] : (uint64_t)b * a.dst_stride);
    uint32_t cap = uni(a.dst_cap ? a.dst_cap[b] : a.dst_cap_uniform);
    uint32_t out_len = 0;
    int32_t st = encode_block_fast<HAS_DICT>(in, out, cap, sh, T, sc.dict_table, gen, offA, offB, (lds_u8 *)s_win[wave], lane,
                                             out_len);
    if (lane == 0) {
      a.out_len[b] = out_len;
      a.status[b] = st;
    }
  }
  if (lane == 0) sc.slot_gen[slot] = gen;
}

// Dictionary priming (:83-88): bucket <- largest dictionary position + 1.
__global__ void dict_prime_kernel(const uint8_t *__restrict__ dict, uint32_t dict_len, uint32_t *table) {
  uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos + 4 <= dict_len) atomicMax(table + hash16(ld32(dict + pos)), pos + 1);
}

void launch_dict_prime(const uint8_t *dict, uint32_t dict_len, uint32_t *dict_table, hipStream_t s) {
  (void)hipMemsetAsync(dict_table, 0, kHashEntries * sizeof(uint32_t), s);
  if (dict_len >= 4) {
    uint32_t n = dict_len - 3;
    hipLaunchKernelGGL(dict_prime_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dict, dict_len, dict_table);
  }
}

#ifdef LZ4MI_PROFILE
extern "C" void lz4mi_debug_enc_phases(unsigned long long *out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_enc_phase), sizeof(unsigned long long) * PH_SLOTS);
  if (reset) {
    unsigned long long z[PH_SLOTS] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_enc_phase), z, sizeof z);
  }
}
#endif

void launch_encode_fast(const BatchArgs &a, uint32_t accel_shift, const EncodeScratch &sc, uint32_t *counter,
                        uint32_t grid, hipStream_t s) {
  if (a.dict_len)
    hipLaunchKernelGGL((encode_fast_kernel<true>), dim3(grid), dim3(256), 0, s, a, accel_shift, sc, counter);
  else
    hipLaunchKernelGGL((encode_fast_kernel<false>), dim3(grid), dim3(256), 0, s, a, accel_shift, sc, counter);
}

}  // namespace lz4mi
