// LZ4HC block encoder for gfx950, byte-identical to lz4HcBlockCompressToWriter
// (lib/src/hc/lz4_hc_block_encoder.dart:49-197): hash-chain search, depth-bounded, greedy.
//
// One wavefront per block works on the real head table (65536 x u32 per wavefront slot,
// generation-tagged like the fast encoder's) and the real chain array (u32 per position) in
// HBM/L2, so every quirk of the reference's chain (positions that are never inserted,
// re-inserted positions that turn the chain into a cycle, far candidates that still consume
// search depth, :125-129) is reproduced by construction.  Positions without a match are taken
// 64 at a time (see encode_block_hc); the lanes also share the work inside one step: long match
// lengths (512 bytes per step), the backward extension, sequence emission, and the interior
// inserts after a match (:180-185), which go 64 positions per step with the in-step order
// resolved in registers (a position links to the nearest earlier lane of its bucket, the last
// lane of a bucket becomes its head).
// A shared dictionary is inserted once per batch into a read-only head table and chain.
#include "api_common.h"
#include "encode_common.h"

namespace lz4mi {

struct HcTables {
  uint32_t *head;            // slot head table, entry = gen<<24 | (pos+1)
  uint32_t *chain;           // slot chain, entry = pos+1 of the previous head (0 = none), index = pos - D
  const uint32_t *dict_head; // shared: pos+1 (0 = none)
  const uint32_t *dict_chain;// shared: index = pos (< D)
  uint32_t gen;
  uint32_t D;
};

__device__ __forceinline__ int32_t hc_head_get(const HcTables &t, uint32_t h) {
  uint32_t e = __hip_atomic_load(t.head + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if ((e >> kPosBits) == t.gen) return (int32_t)(e & kPosMask) - 1;
  if (t.dict_head) return (int32_t)t.dict_head[h] - 1;
  return -1;
}
__device__ __forceinline__ void hc_head_set(const HcTables &t, uint32_t h, uint32_t pos) {
  __hip_atomic_store(t.head + h, (t.gen << kPosBits) | (pos + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int32_t hc_chain_get(const HcTables &t, uint32_t pos) {
  if (pos < t.D) return (int32_t)t.dict_chain[pos] - 1;
  return (int32_t)__hip_atomic_load(t.chain + (pos - t.D), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - 1;
}
__device__ __forceinline__ void hc_chain_set(const HcTables &t, uint32_t pos, int32_t prev) {
  __hip_atomic_store(t.chain + (pos - t.D), (uint32_t)(prev + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// insert(pos) for pos = lo..hi ascending (:96-103), 64 positions per step.
template <bool HAS_DICT>
__device__ __forceinline__ void hc_insert_range(const EncIn &in, const HcTables &t, uint32_t lo, uint32_t hi,
                                                uint32_t lane) {
  for (uint32_t base = lo; base <= hi; base += kWave) {
    const uint32_t p = base + lane;
    const bool valid = p <= hi;
    const uint64_t vmask = ballot(valid);
    uint32_t h = 0;
    if (valid) h = hash16(in_ld32<HAS_DICT>(in, p));
    uint64_t same = vmask;
    for (uint32_t bit = 0; bit < 16; bit++) {
      bool set = (h >> bit) & 1u;
      uint64_t m = ballot(set);
      same &= set ? m : ~m;
    }
    if (valid) {
      const uint64_t earlier = same & ((1ull << lane) - 1ull);
      const uint64_t later = same & ~((2ull << lane) - 1ull);
      int32_t link = earlier ? (int32_t)(base + 63u - (uint32_t)__builtin_clzll(earlier)) : hc_head_get(t, h);
      hc_chain_set(t, p, link);
      if (!later) hc_head_set(t, h, p);
    }
  }
}

// equal bytes from p / q onwards (q < p), never reading at or past total; 512 bytes per step
template <bool HAS_DICT>
__device__ __forceinline__ uint32_t hc_match_len(const EncIn &in, uint32_t p, uint32_t q, uint32_t total,
                                                 uint32_t lane) {
  uint32_t len = 0;
  for (;;) {
    uint32_t pp = p + len + 8 * lane, qq = q + len + 8 * lane;
    uint32_t avail = pp < total ? (total - pp < 8 ? total - pp : 8u) : 0u;
    uint32_t eq = 0;
    if (avail == 8) {
      uint32_t x0 = in_ld32<HAS_DICT>(in, pp) ^ in_ld32<HAS_DICT>(in, qq);
      uint32_t x1 = in_ld32<HAS_DICT>(in, pp + 4) ^ in_ld32<HAS_DICT>(in, qq + 4);
      eq = x0 ? ((uint32_t)__builtin_ctz(x0) >> 3) : (x1 ? 4u + ((uint32_t)__builtin_ctz(x1) >> 3) : 8u);
    } else {
      for (uint32_t k = 0; k < avail; k++) {
        if (in_byte<HAS_DICT>(in, pp + k) != in_byte<HAS_DICT>(in, qq + k)) break;
        eq++;
      }
    }
    uint64_t m = ballot(eq < 8);
    if (m) {
      uint32_t L = (uint32_t)__builtin_ctzll(m);
      return len + 8 * L + rdlane(eq, L);
    }
    len += 8 * kWave;
  }
}

// Equal bytes (>= 4, the caller has compared the first four) of p.. and c.. (c < p), measured by
// one lane up to kLaneCmp bytes; `more` says the run was still going at that point.
constexpr uint32_t kLaneCmp = 64;
template <bool HAS_DICT>
__device__ __forceinline__ uint32_t hc_lane_len(const EncIn &in, uint32_t p, uint32_t c, uint32_t total, bool &more) {
  uint32_t len = 4;
  more = false;
  for (;;) {
    if (p + len + 4 <= total) {
      const uint32_t x = in_ld32<HAS_DICT>(in, p + len) ^ in_ld32<HAS_DICT>(in, c + len);
      if (x) return len + ((uint32_t)__builtin_ctz(x) >> 3);
      len += 4;
      if (len >= kLaneCmp) { more = p + len < total; return len; }
    } else {
      while (p + len < total && in_byte<HAS_DICT>(in, p + len) == in_byte<HAS_DICT>(in, c + len)) len++;
      return len;
    }
  }
}

// The reference visits positions one by one (:116-190), but as long as no match is taken the
// work at consecutive positions is independent: position p links to the latest earlier position
// of its bucket and walks that chain.  A wavefront therefore takes 64 consecutive positions at a
// time, one per lane: links inside the window are resolved in registers (nearest earlier lane of
// the same bucket, kept in a 64-entry LDS array for the walks), every lane walks its chain in
// lockstep, and the first lane whose walk found a match is the position where the reference
// takes its next match.  Only the lanes up to that one are then committed to the head table and
// chain; the later lanes' work is discarded, so the tables never hold an insert the reference
// would not have made at that point.
template <bool HAS_DICT>
__device__ __forceinline__ int32_t encode_block_hc(const EncIn in, uint8_t *out, uint32_t cap, uint32_t max_depth,
                                                   HcTables t, lds_u32 *wchain, uint32_t lane, uint32_t &out_len) {
  const uint32_t D = HAS_DICT ? in.D : 0u;
  const uint32_t total = in.total;
  const uint32_t n = total - D;
  out_len = 0;
  if (n == 0) return LZ4MI_OK;  // :57-60
  if ((uint64_t)cap < (uint64_t)n + n / 255u + 16u) return LZ4MI_DST_TOO_SMALL;
  if (total > kPosMask - 1) return LZ4MI_UNSUPPORTED_HERE;
  uint32_t anchor = D, i = D, op = 0;

  if (n >= 4) {  // :86-90
    while (i + 4 <= total) {  // :116
      const uint32_t p = i + lane;
      const bool valid = p + 4 <= total;  // a prefix of the lanes
      const uint32_t nvalid = (uint32_t)__builtin_popcountll(ballot(valid));
      uint32_t seq = 0, h = 0;
      if (valid) {
        seq = in_ld32<HAS_DICT>(in, p);
        h = hash16(seq);
      }
      uint64_t same = ballot(valid);
      for (uint32_t bit = 0; bit < 16; bit++) {
        const bool set = (h >> bit) & 1u;
        const uint64_t m = ballot(set);
        same &= set ? m : ~m;
      }
      if (!valid) same = 0;
      const uint64_t earlier = same & ((1ull << lane) - 1ull);
      const uint64_t later = same & ~((2ull << lane) - 1ull);
      int32_t pred = -1;  // insert(p): the bucket's head as position p sees it (:96-103)
      if (valid) pred = earlier ? (int32_t)(i + 63u - (uint32_t)__builtin_clzll(earlier)) : hc_head_get(t, h);
      wchain[lane] = (uint32_t)(pred + 1);

      // ---- chain walks, all lanes in lockstep (:123-147)
      int32_t cand = pred;
      uint32_t depth = 0, best_len = 0, best_ref = 0;
      bool done = !valid;
      uint32_t W = kWave;  // lowest lane known to hold a match
      for (;;) {
        const bool act = !done && lane <= W && cand >= 0 && depth < max_depth;
        if (!act) done = true;
        if (ballot(act) == 0) break;
        const uint32_t c = (uint32_t)cand;
        int32_t next = -1;
        uint32_t cseq = 0;
        bool near = false;
        if (act) {
          next = c >= i ? (int32_t)wchain[c - i] - 1 : hc_chain_get(t, c);
          near = p - c <= 0xFFFFu;  // far candidates only consume depth (:125-129)
          if (near) cseq = in_ld32<HAS_DICT>(in, c);
        }
        bool m4 = act && near && cseq == seq;
        // a lane with a 4-byte hit holds a match: lanes above the lowest such lane no longer matter
        const uint64_t hitm = ballot(m4);
        if (hitm) {
          const uint32_t Wc = (uint32_t)__builtin_ctzll(hitm);
          if (Wc < W) W = Wc;
        }
        m4 = m4 && lane <= W;
        uint32_t ml = 0;
        bool more = false;
        if (m4) ml = hc_lane_len<HAS_DICT>(in, p, c, total, more);
        for (uint64_t lm = ballot(more); lm; lm &= lm - 1) {  // long runs: the whole wavefront measures, 512 bytes per step
          const uint32_t L = (uint32_t)__builtin_ctzll(lm);
          const uint32_t cL = rdlane(c, L), lenL = rdlane(ml, L);
          const uint32_t full = lenL + hc_match_len<HAS_DICT>(in, i + L + lenL, cL + lenL, total, lane);
          if (lane == L) ml = full;
        }
        if (m4 && ml > best_len) {  // strictly longer: the nearest candidate wins ties (:137)
          best_len = ml;
          best_ref = c;
          if (p + best_len == total) done = true;  // :140-142
        }
        if (act) {
          cand = next;
          depth++;
        }
      }

      // ---- commit the inserts the reference has made by now: every lane up to the match
      const uint32_t ncommit = W < kWave ? W + 1 : nvalid;
      if (valid && lane < ncommit) {
        hc_chain_set(t, p, pred);
        const uint64_t inrange = ncommit < kWave ? (1ull << ncommit) - 1ull : ~0ull;
        if (!(later & inrange)) hc_head_set(t, h, p);
      }
      if (W == kWave) {  // no match in this window (:190 for each position)
        i += nvalid;
        continue;
      }
      // ---- match at lane W (:150-187)
      const uint32_t im = i + W;
      const uint32_t blen = rdlane(best_len, W), bref = rdlane(best_ref, W);
      const uint32_t maxb = im - anchor < bref ? im - anchor : bref;
      uint32_t b = 0;
      while (b < maxb) {  // backward extension :155-161
        const uint32_t tt = b + lane;
        const bool eq = tt < maxb && in_byte<HAS_DICT>(in, im - 1 - tt) == in_byte<HAS_DICT>(in, bref - 1 - tt);
        const uint64_t m = ballot(!eq);
        if (m) { b += (uint32_t)__builtin_ctzll(m); break; }
        b += kWave;
      }
      const uint32_t ms = im - b, ml = blen + b;
      op = emit_sequence(out, op, in.src + (anchor - D), ms - anchor, im - bref, ml, lane);
      i = ms + ml;  // :173-174
      anchor = i;
      if (i + 4 > total) break;  // :176-178
      hc_insert_range<HAS_DICT>(in, t, ms + 1, i - 4, lane);  // :180-185 (re-inserts included)
    }
  }
  const uint32_t last = total - anchor;  // :193-196
  if (last != 0) op = emit_last_literals(out, op, in.src + (anchor - D), last, lane);
  out_len = op;
  return LZ4MI_OK;
}

struct HcScratch {
  uint32_t *tables;      // nslots * 65536
  uint32_t *slot_gen;
  uint32_t *chains;      // nslots * chain_stride
  uint64_t chain_stride;
  const uint32_t *dict_head;
  const uint32_t *dict_chain;
};

template <bool HAS_DICT>
__global__ __launch_bounds__(256) void encode_hc_kernel(BatchArgs a, uint32_t max_depth, HcScratch sc,
                                                        uint32_t *counter) {
  __shared__ uint32_t s_wchain[4][kWave];
  const uint32_t lane = lane_id();
  const uint32_t slot = uni(blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave);
  lds_u32 *wchain = (lds_u32 *)s_wchain[uni(threadIdx.x / kWave)];
  HcTables t;
  t.head = sc.tables + (size_t)slot * kHashEntries;
  t.chain = sc.chains + (size_t)slot * sc.chain_stride;
  t.dict_head = HAS_DICT ? sc.dict_head : nullptr;
  t.dict_chain = sc.dict_chain;
  t.D = HAS_DICT ? a.dict_len : 0u;
  uint32_t gen = uni(sc.slot_gen[slot]);
  EncIn in;
  in.dict = a.dict;
  in.D = a.dict_len;
  for (;;) {
    const uint32_t b = next_block(counter, lane);
    if (b >= a.nblocks) break;
    if (++gen > kGenMax) {
      for (uint32_t k = lane * 4; k < kHashEntries; k += kWave * 4) *(uint4 *)(t.head + k) = make_uint4(0, 0, 0, 0);
      gen = 1;
    }
    t.gen = gen;
    in.src = a.src + uni64(a.src_off ? a.src_off[b] : (uint64_t)b * a.src_stride);
    in.total = in.D + uni(block_src_len(a, b));
    uint8_t *out = a.dst + uni64(a.dst_off ? a.dst_off[b] : (uint64_t)b * a.dst_stride);
    uint32_t cap = uni(a.dst_cap ? a.dst_cap[b] : a.dst_cap_uniform);
    uint32_t out_len = 0;
    int32_t st = (uint64_t)(in.total - in.D) > sc.chain_stride
                     ? (int32_t)LZ4MI_UNSUPPORTED_HERE
                     : encode_block_hc<HAS_DICT>(in, out, cap, max_depth, t, wchain, lane, out_len);
    if (lane == 0) {
      a.out_len[b] = out_len;
      a.status[b] = st;
    }
  }
  if (lane == 0) sc.slot_gen[slot] = gen;
}

// Dictionary priming (:105-109): one wavefront inserts every dictionary position in order
// into the shared head table / chain (entries are pos+1, 0 = none; no generation tag).
__global__ __launch_bounds__(64) void hc_dict_prime_kernel(const uint8_t *dict, uint32_t D, uint32_t *dict_head,
                                                           uint32_t *dict_chain) {
  const uint32_t lane = lane_id();
  if (D < 4) return;
  const uint32_t hi = D - 4;
  for (uint32_t base = 0; base <= hi; base += kWave) {
    const uint32_t p = base + lane;
    const bool valid = p <= hi;
    const uint64_t vmask = ballot(valid);
    uint32_t h = 0;
    if (valid) h = hash16(ld32(dict + p));
    uint64_t same = vmask;
    for (uint32_t bit = 0; bit < 16; bit++) {
      bool set = (h >> bit) & 1u;
      uint64_t m = ballot(set);
      same &= set ? m : ~m;
    }
    if (valid) {
      const uint64_t earlier = same & ((1ull << lane) - 1ull);
      const uint64_t later = same & ~((2ull << lane) - 1ull);
      uint32_t link = earlier ? base + 63u - (uint32_t)__builtin_clzll(earlier) + 1u
                              : __hip_atomic_load(dict_head + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      dict_chain[p] = link;
      if (!later) __hip_atomic_store(dict_head + h, p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}

int32_t run_compress_hc(lz4mi_ctx *c, const BatchArgs &a_in, int64_t depth) {
  if (depth < 1) return LZ4MI_RANGE;  // lib/src/hc/lz4_hc_options.dart:13-15
  if (a_in.nblocks == 0) return LZ4MI_OK;
  BatchArgs a = a_in;
  if (a.dict == nullptr) a.dict_len = 0;
  if (a.dict_len > 65535) {  // :62-72
    a.dict += a.dict_len - 65535;
    a.dict_len = 65535;
  }
  // chain capacity: the largest block this batch can contain
  uint64_t max_len = a.src_len ? (a.dst_cap_uniform ? a.dst_cap_uniform : (4u << 20)) : a.src_len_uniform;
  if (max_len > kPosMask) max_len = kPosMask;
  const uint64_t stride = round_up(max_len + 64, 64);
  const uint64_t budget = 8ull << 30;  // bytes of chain scratch
  uint32_t slots = (uint32_t)round_up(a.nblocks < c->max_slots ? a.nblocks : c->max_slots, 4);
  while (slots > 4 && (uint64_t)slots * stride * 4 > budget) slots -= 4;
  // head tables are shared with the fast encoder's slots
  if (slots > c->nslots) {
    LZ4MI_HIP(c, hipStreamSynchronize(c->stream));
    LZ4MI_HIP(c, c->tables.ensure((size_t)slots * kHashEntries * sizeof(uint32_t)));
    LZ4MI_HIP(c, c->slot_gen.ensure((size_t)slots * sizeof(uint32_t)));
    LZ4MI_HIP(c, hipMemsetAsync(c->tables.p, 0, (size_t)slots * kHashEntries * sizeof(uint32_t), c->stream));
    LZ4MI_HIP(c, hipMemsetAsync(c->slot_gen.p, 0, (size_t)slots * sizeof(uint32_t), c->stream));
    c->nslots = slots;
  }
  LZ4MI_HIP(c, c->hc_chain.ensure((size_t)slots * stride * 4));
  HcScratch sc;
  sc.tables = c->tables.as<uint32_t>();
  sc.slot_gen = c->slot_gen.as<uint32_t>();
  sc.chains = c->hc_chain.as<uint32_t>();
  sc.chain_stride = stride;
  sc.dict_head = nullptr;
  sc.dict_chain = nullptr;
  if (a.dict_len) {
    LZ4MI_HIP(c, c->dict_table.ensure(kHashEntries * sizeof(uint32_t)));
    LZ4MI_HIP(c, c->dict_copy.ensure((size_t)65536 * sizeof(uint32_t)));
    LZ4MI_HIP(c, hipMemsetAsync(c->dict_table.p, 0, kHashEntries * sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(hc_dict_prime_kernel, dim3(1), dim3(64), 0, c->stream, a.dict, a.dict_len,
                       c->dict_table.as<uint32_t>(), c->dict_copy.as<uint32_t>());
    sc.dict_head = c->dict_table.as<uint32_t>();
    sc.dict_chain = c->dict_copy.as<uint32_t>();
  }
  uint32_t *counter = c->d_counters + 2;
  LZ4MI_HIP(c, hipMemsetAsync(counter, 0, sizeof(uint32_t), c->stream));
  uint32_t grid = (a.nblocks + 3) / 4;
  if (grid > slots / 4) grid = slots / 4;
  uint32_t d32 = depth > 0x7fffffff ? 0x7fffffffu : (uint32_t)depth;
  if (a.dict_len)
    hipLaunchKernelGGL((encode_hc_kernel<true>), dim3(grid), dim3(256), 0, c->stream, a, d32, sc, counter);
  else
    hipLaunchKernelGGL((encode_hc_kernel<false>), dim3(grid), dim3(256), 0, c->stream, a, d32, sc, counter);
  LZ4MI_HIP(c, hipGetLastError());
  return LZ4MI_OK;
}

}  // namespace lz4mi
