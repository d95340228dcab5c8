// LZ4HC block encoder for gfx950, byte-identical to lz4HcBlockCompressToWriter
// (lib/src/hc/lz4_hc_block_encoder.dart:49-197): hash-chain search, depth-bounded, greedy.
//
// The reference keeps a head table and a chain array and walks the chain of a position's bucket
// candidate by candidate.  A chain is nothing but "the earlier inserted positions of the same
// bucket, nearest first", so this kernel replaces pointer chasing by a per-block index built in
// parallel before the parse: every position of dict||src sorted by (bucket, position).  The
// candidates of position p are then the entries just below p's own rank: a contiguous slice that
// 64 lanes read and evaluate at once.  Three facts of the reference make that exact:
//   * every position below the current one has been inserted, except the last three positions
//     of each match (:180-185 stops at i-4; the parse resumes at i) and the last three of the
//     dictionary: those entries carry a flag and are skipped without consuming search depth;
//   * far candidates (> 65535 back) still consume depth (:125-129);
//   * after a match that was extended backwards the interior loop re-inserts the positions
//     ms+1..i that had already been inserted.  Working the head/chain updates through shows the
//     result for a bucket with re-inserted positions j1 < .. < jm: head = jm, jm -> .. -> j1 -> jm,
//     a cycle that cuts off everything older than j1.  Going round the cycle only repeats
//     candidates (a repeat is never strictly longer, :137), so the bucket simply gets a floor:
//     ranks below j1 are unreachable from then on.
// Positions without a match are taken 64 at a time, one per lane, all lanes stepping through
// their slices in lockstep; the first lane whose walk found a match is where the reference takes
// its next match.  When only one or two lanes are still walking (text: a match at nearly every
// position, long chains) the whole wavefront evaluates one position's slice, 64 candidates per
// step.
#include "api_common.h"
#include "encode_common.h"
#include "profile.h"

namespace lz4mi {

#ifdef LZ4MI_PROFILE
__device__ unsigned long long g_hc_phase[PH_SLOTS];  // 0 index build, 1 parse, 15 total (cycles)
#endif

constexpr uint32_t kNoIns = 0x80000000u;  // index entry: this position is never inserted
constexpr uint32_t kHcPosMask = 0x7FFFFFFFu;  // the position of an index entry

struct HcIndex {
  uint32_t *floor_;  // 65536: lowest rank of the bucket a walk can still reach
  uint32_t *rank;    // per position of dict||src
  uint2 *sorted;     // per rank: position (| kNoIns) and the four bytes at that position (what a walk compares first)
  uint2 *tmp;        // the entries after the first sort pass
};

// Arrays of a slot are rewritten for every block by the wavefront that owns the slot; workgroup
// scope keeps the accesses out of the per-CU L1 (no stale lines from the previous block).
__device__ __forceinline__ uint32_t xld(const uint32_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void xst(uint32_t *p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// index entry: both words in one access
__device__ __forceinline__ uint2 xld_entry(const uint2 *p) {
  const unsigned long long v = __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}
__device__ __forceinline__ void xst_entry(uint2 *p, uint32_t pos, uint32_t seq) {
  __hip_atomic_store((unsigned long long *)p, ((unsigned long long)seq << 32) | pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void xst_entry_pos(uint2 *p, uint32_t pos) { xst((uint32_t *)p, pos); }

// lanes (among the valid ones) whose 16-bit hash equals this lane's
__device__ __forceinline__ uint64_t same_bucket(uint32_t h, bool valid) {
  uint64_t same = ballot(valid);
  for (uint32_t bit = 0; bit < 16; bit++) {
    const bool set = (h >> bit) & 1u;
    const uint64_t m = ballot(set);
    same &= set ? m : ~m;
  }
  return valid ? same : 0ull;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_max(uint32_t v) {
  const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
  return o > v ? o : v;
}
// maximum over the 64 lanes (wave-uniform result): row shifts, then the two row broadcasts
__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
  v = dpp_max<0x111, 0xF>(v);  // row_shr:1
  v = dpp_max<0x112, 0xF>(v);  // row_shr:2
  v = dpp_max<0x114, 0xF>(v);  // row_shr:4
  v = dpp_max<0x118, 0xF>(v);  // row_shr:8 -> lane 15 of every row holds the row's maximum
  v = dpp_max<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
  v = dpp_max<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3
  return rdlane(v, kWave - 1);
}

// lanes (among the valid ones) whose 8-bit digit equals this lane's
__device__ __forceinline__ uint64_t same_digit(uint32_t dg, bool valid) {
  uint64_t same = ballot(valid);
#pragma unroll
  for (uint32_t bit = 0; bit < 8; bit++) {
    const bool set = (dg >> bit) & 1u;
    const uint64_t m = ballot(set);
    same &= set ? m : ~m;
  }
  return valid ? same : 0ull;
}

// One stable counting-sort step over a chunk of 64 elements in order: where this lane's element goes.
// next[dg] = next free place of digit dg (LDS, owned by this wavefront); lanes of one chunk that share a
// digit take consecutive places in lane order, the last of them moves the counter on.
__device__ __forceinline__ uint32_t radix_place(lds_u32 *next, uint32_t dg, bool valid, uint32_t lane) {
  const uint64_t same = same_digit(dg, valid);
  const uint32_t last = same ? 63u - (uint32_t)__builtin_clzll(same) : lane;
  const uint32_t before = (uint32_t)__builtin_popcountll(same & ((1ull << lane) - 1ull));
  uint32_t old = 0;
  if (valid && lane == last) {
    old = next[dg];
    next[dg] = old + (uint32_t)__builtin_popcountll(same);
  }
  return (uint32_t)__shfl((int)old, (int)last) + before;
}

// Index of one block: the positions 0 .. T-1 sorted by (bucket, position).  A two-pass radix sort on the
// 16-bit bucket, low byte first, each pass a stable counting sort whose 256 counters live in LDS: the
// scattered writes of a pass go to 256 append streams (lines fill up in L2 and leave as whole lines)
// instead of 65 536 buckets spread over the slot's memory.  hist: 512 LDS words of this wavefront.
template <bool HAS_DICT>
__device__ __forceinline__ void hc_build_index(const EncIn &in, const HcIndex &x, uint32_t T, lds_u32 *hist,
                                               uint32_t lane) {
  const uint32_t D = HAS_DICT ? in.D : 0u;
  constexpr uint32_t U = 4;  // chunks of 64 positions whose loads are in flight together
  // 1: histograms of both bytes of the bucket
  for (uint32_t k = lane; k < 512u; k += kWave) hist[k] = 0;
  for (uint32_t base = 0; base < T; base += U * kWave) {
    uint32_t h[U];
#pragma unroll
    for (uint32_t u = 0; u < U; u++) {
      const uint32_t p = base + u * kWave + lane;
      h[u] = p < T ? hash16(in_ld32<HAS_DICT>(in, p)) : 0x10000u;
    }
#pragma unroll
    for (uint32_t u = 0; u < U; u++) {
      if (h[u] < 0x10000u) {
        (void)__hip_atomic_fetch_add(hist + (h[u] & 255u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        (void)__hip_atomic_fetch_add(hist + 256u + (h[u] >> 8), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
    }
  }
  // 2: exclusive prefix sums: first place of every digit (four counters per lane and histogram)
#pragma unroll
  for (uint32_t t = 0; t < 2; t++) {
    lds_u32 *hh = hist + 256u * t + 4u * lane;
    const uint32_t v0 = hh[0], v1 = hh[1], v2 = hh[2], v3 = hh[3];
    const uint32_t sum = v0 + v1 + v2 + v3;
    const uint32_t ex = wave_incl_scan(sum) - sum;
    hh[0] = ex;
    hh[1] = ex + v0;
    hh[2] = ex + v0 + v1;
    hh[3] = ex + v0 + v1 + v2;
  }
  // 3: by the low byte, input order -> tmp
  for (uint32_t base = 0; base < T; base += U * kWave) {
    uint32_t sq[U];
#pragma unroll
    for (uint32_t u = 0; u < U; u++) {
      const uint32_t p = base + u * kWave + lane;
      sq[u] = p < T ? in_ld32<HAS_DICT>(in, p) : 0u;
    }
#pragma unroll
    for (uint32_t u = 0; u < U; u++) {
      const uint32_t p = base + u * kWave + lane;
      const uint32_t o = radix_place(hist, hash16(sq[u]) & 255u, p < T, lane);
      if (p < T) xst_entry(x.tmp + o, p, sq[u]);
    }
  }
  vm_quiesce();
  // 4: by the high byte, tmp order -> sorted; rank[p] = where p went
  for (uint32_t base = 0; base < T; base += U * kWave) {
    uint2 en[U];
#pragma unroll
    for (uint32_t u = 0; u < U; u++) {
      const uint32_t i = base + u * kWave + lane;
      en[u] = i < T ? xld_entry(x.tmp + i) : make_uint2(0u, 0u);
    }
#pragma unroll
    for (uint32_t u = 0; u < U; u++) {
      const uint32_t i = base + u * kWave + lane;
      const uint32_t o = radix_place(hist + 256u, hash16(en[u].y) >> 8, i < T, lane);
      if (i < T) {
        xst_entry(x.sorted + o, en[u].x, en[u].y);
        xst(x.rank + en[u].x, o);
      }
    }
  }
  vm_quiesce();
  // 5: first rank of every bucket that holds a position
  uint32_t prev_key = 0x10000u;  // bucket of the entry in front of the chunk
  for (uint32_t base = 0; base < T; base += kWave) {
    const uint32_t i = base + lane;
    const uint32_t key = i < T ? hash16(xld_entry(x.sorted + i).y) : 0x10001u;
    uint32_t before = (uint32_t)__shfl_up((int)key, 1);
    if (lane == 0) before = prev_key;
    if (i < T && key != before) xst(x.floor_ + key, i);
    prev_key = rdlane(key, kWave - 1);
  }
  // the last three dictionary positions are never inserted (:105-109 stops at dictLength - 4)
  if (HAS_DICT && D != 0) {
    const uint32_t q = D - 1 - lane;
    if (lane < 3 && lane < D && q < T) xst_entry_pos(x.sorted + xld(x.rank + q), q | kNoIns);
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
// one lane up to `limit` bytes; `more` says the run was still going at that point.
template <bool HAS_DICT>
__device__ __forceinline__ uint32_t hc_lane_len(const EncIn &in, uint32_t p, uint32_t c, uint32_t total,
                                                uint32_t limit, bool &more) {
  uint32_t len = 4;
  more = false;
  if (p + 16 <= total) {  // most runs end within 16 bytes: the next twelve of both sides in one round trip
    const uint32_t x1 = in_ld32<HAS_DICT>(in, p + 4) ^ in_ld32<HAS_DICT>(in, c + 4);
    const uint32_t x2 = in_ld32<HAS_DICT>(in, p + 8) ^ in_ld32<HAS_DICT>(in, c + 8);
    const uint32_t x3 = in_ld32<HAS_DICT>(in, p + 12) ^ in_ld32<HAS_DICT>(in, c + 12);
    if (x1) return 4 + ((uint32_t)__builtin_ctz(x1) >> 3);
    if (x2) return 8 + ((uint32_t)__builtin_ctz(x2) >> 3);
    if (x3) return 12 + ((uint32_t)__builtin_ctz(x3) >> 3);
    len = 16;
    if (len >= limit) { more = p + len < total; return len; }
  }
  for (;;) {
    if (p + len + 4 <= total) {
      const uint32_t x = in_ld32<HAS_DICT>(in, p + len) ^ in_ld32<HAS_DICT>(in, c + len);
      if (x) return len + ((uint32_t)__builtin_ctz(x) >> 3);
      len += 4;
      if (len >= limit) { more = p + len < total; return len; }
    } else {
      while (p + len < total && in_byte<HAS_DICT>(in, p + len) == in_byte<HAS_DICT>(in, c + len)) len++;
      return len;
    }
  }
}

// Lanes flagged `more` hold a run that was still going after their own compare: the whole
// wavefront finishes those, one at a time.
template <bool HAS_DICT>
__device__ __forceinline__ void hc_finish_long(const EncIn &in, uint32_t p_uniform_base, bool per_lane_p, uint32_t c,
                                               uint32_t &ml, bool more, uint32_t total, uint32_t lane) {
  for (uint64_t lm = ballot(more); lm; lm &= lm - 1) {
    const uint32_t L = (uint32_t)__builtin_ctzll(lm);
    const uint32_t pL = per_lane_p ? p_uniform_base + L : p_uniform_base;
    const uint32_t cL = rdlane(c, L), lenL = rdlane(ml, L);
    const uint32_t full = lenL + hc_match_len<HAS_DICT>(in, pL + lenL, cL + lenL, total, lane);
    if (lane == L) ml = full;
  }
}

// The rest of one position's walk with the whole wavefront: 64 index entries per step, lane d
// looks at rank idx - d.  All arguments are wave-uniform.
template <bool HAS_DICT>
__device__ __forceinline__ void hc_eval_slice(const EncIn &in, const HcIndex &x, uint32_t P, uint32_t seq, int32_t idx,
                                              int32_t lo, uint32_t depth, uint32_t max_depth, uint32_t total,
                                              uint32_t lane, uint32_t &best_len, uint32_t &best_ref) {
  while (idx >= lo && depth < max_depth) {
    const int32_t my = idx - (int32_t)lane;
    uint2 en = make_uint2(kNoIns, 0u);
    if (my >= lo) en = xld_entry(x.sorted + my);
    const uint32_t e = en.x;
    const bool ins = !(e & kNoIns);
    const uint64_t insm = ballot(ins);
    const uint32_t ordinal = (uint32_t)__builtin_popcountll(insm & ((1ull << lane) - 1ull));
    const bool take = ins && depth + ordinal < max_depth;  // :123 depth bound, flagged entries are not on the chain
    const uint32_t c = e & kHcPosMask;
    const bool near = take && P - c <= 0xFFFFu;  // :125-129
    const bool m4 = near && en.y == seq;
    uint32_t ml = 0;
    bool more = false;
    if (m4) ml = hc_lane_len<HAS_DICT>(in, P, c, total, 512u, more);
    hc_finish_long<HAS_DICT>(in, P, false, c, ml, more, total, lane);
    const uint32_t mx = wave_max(m4 ? ml : 0u);
    if (mx > best_len) {  // strictly longer; among equals the nearest (lowest lane) was first (:137)
      const uint32_t Lb = (uint32_t)__builtin_ctzll(ballot(m4 && ml == mx));
      best_len = mx;
      best_ref = rdlane(c, Lb);
      if (P + best_len == total) return;  // :140-142 (nothing later can be strictly longer)
    }
    depth += (uint32_t)__builtin_popcountll(insm);
    idx -= (int32_t)kWave;
  }
}

template <bool HAS_DICT>
__device__ __forceinline__ int32_t encode_block_hc(const EncIn in, uint8_t *out, uint32_t cap, uint32_t max_depth,
                                                   const HcIndex x, lds_u32 *hist, uint32_t lane, uint32_t &out_len) {
  const uint32_t D = HAS_DICT ? in.D : 0u;
  const uint32_t total = in.total;
  const uint32_t plimit = in.plimit, elimit = in.elimit;  // both == total unless spec-conformant
  const uint32_t n = total - D;
  out_len = 0;
  if (n == 0) return LZ4MI_OK;  // :57-60
  if ((uint64_t)cap < (uint64_t)n + n / 255u + 16u) return LZ4MI_DST_TOO_SMALL;
  if (total > 0x7FFFFFF0u) return LZ4MI_UNSUPPORTED_HERE;
  uint32_t anchor = D, i = D, op = 0;
  PH_DECL

  if (n >= 4) {  // :86-90
    hc_build_index<HAS_DICT>(in, x, total - 3, hist, lane);
    PH(0);
    // Positions searched per step.  Every position's search stands on its own, so any width gives the
    // reference's parse; what it changes is the work thrown away: all lanes above the first match searched
    // for nothing, and every one of them cost scattered index reads (a launch of text-like blocks, a match at
    // nearly every position, is bound by exactly that traffic).  The width follows the data: four times the
    // distance to the last match, twice as wide after a step without one.
    uint32_t width = kWave;
    while (i + 4 <= plimit) {  // :116
      const uint32_t p = i + lane;
      const bool valid = p + 4 <= plimit && lane < width;  // a prefix of the lanes
      const uint32_t nvalid = (uint32_t)__builtin_popcountll(ballot(valid));
      uint32_t seq = 0;
      int32_t idx = -1, lo = 0;
      if (valid) {
        seq = in_ld32<HAS_DICT>(in, p);
        lo = (int32_t)xld(x.floor_ + hash16(seq));
        idx = (int32_t)xld(x.rank + p) - 1;
      }
      uint32_t depth = 0, best_len = 0, best_ref = 0;
      bool done = !valid;
      uint32_t W = kWave;  // lowest lane known to hold a match
      for (;;) {
        const bool act = !done && lane <= W && idx >= lo && depth < max_depth;
        if (!act) done = true;
        const uint64_t am = ballot(act);
        if (am == 0) break;
        if (__builtin_popcountll(am) <= 2) {
          // few walkers left: the wavefront finishes the lowest one, 64 candidates per step
          const uint32_t L = (uint32_t)__builtin_ctzll(am);
          uint32_t bl = rdlane(best_len, L), br = rdlane(best_ref, L);
          hc_eval_slice<HAS_DICT>(in, x, i + L, rdlane(seq, L), (int32_t)rdlane((uint32_t)idx, L),
                                  (int32_t)rdlane((uint32_t)lo, L), rdlane(depth, L), max_depth, elimit, lane, bl, br);
          if (lane == L) {
            best_len = bl;
            best_ref = br;
            done = true;
          }
          if (bl >= 4 && L < W) W = L;
          continue;
        }
        // one candidate per walking lane
        uint2 en = make_uint2(kNoIns, 0u);
        if (act) {
          en = xld_entry(x.sorted + idx);
          idx--;
        }
        const uint32_t e = en.x;
        const bool cnd = act && !(e & kNoIns);  // flagged entries are not on the chain
        const uint32_t c = e & kHcPosMask;
        const bool near = cnd && p - c <= 0xFFFFu;  // far candidates only consume depth (:125-129)
        bool m4 = near && en.y == seq;
        // a lane with a 4-byte hit holds a match: lanes above the lowest such lane no longer matter
        const uint64_t hitm = ballot(m4);
        if (hitm) {
          const uint32_t Wc = (uint32_t)__builtin_ctzll(hitm);
          if (Wc < W) W = Wc;
        }
        m4 = m4 && lane <= W;
        uint32_t ml = 0;
        bool more = false;
        if (m4) ml = hc_lane_len<HAS_DICT>(in, p, c, elimit, 64u, more);
        hc_finish_long<HAS_DICT>(in, i, true, c, ml, more, elimit, lane);
        if (m4 && ml > best_len) {  // strictly longer: the nearest candidate wins ties (:137)
          best_len = ml;
          best_ref = c;
          if (p + best_len == elimit) done = true;  // :140-142
        }
        if (cnd) depth++;
      }
      if (W == kWave) {  // no match in this window (:190 for each position)
        i += nvalid;
        width = width * 2 < kWave ? width * 2 : kWave;
        continue;
      }
      width = 4 * (W + 1) < kWave ? 4 * (W + 1) : kWave;
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
      if (i + 4 > plimit) break;  // :176-178
      // :180-185.  Positions ms+1 .. im are inserted a second time: their buckets get a floor at
      // the lowest of them (highest positions first, so the lowest one writes last).
      for (uint32_t hi = im; hi > ms; hi -= (hi - ms < kWave ? hi - ms : kWave)) {
        const bool v = hi - ms > lane;  // position hi - lane is above ms
        const uint32_t j = hi - lane;
        const uint32_t hj = v ? hash16(in_ld32<HAS_DICT>(in, j)) : 0u;
        const uint64_t same = same_bucket(hj, v);
        if (v && !(same & ~((2ull << lane) - 1ull))) xst(x.floor_ + hj, xld(x.rank + j));
      }
      // the last three positions of the match are never inserted
      {
        const uint32_t q = i - 3 + lane;
        if (lane < 3 && q + 4 <= total) xst_entry_pos(x.sorted + xld(x.rank + q), q | kNoIns);
      }
    }
  }
  const uint32_t last = total - anchor;  // :193-196
  if (last != 0) op = emit_last_literals(out, op, in.src + (anchor - D), last, lane);
  PH(1);
  PH_FLUSH(g_hc_phase, lane);
  out_len = op;
  return LZ4MI_OK;
}

#ifdef LZ4MI_PROFILE
extern "C" void lz4mi_debug_hc_phases(unsigned long long *out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hc_phase), sizeof(unsigned long long) * PH_SLOTS);
  if (reset) {
    unsigned long long z[PH_SLOTS] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_hc_phase), z, sizeof z);
  }
}
#endif

struct HcScratch {
  uint32_t *floor_;    // nslots * 65536
  uint32_t *index;     // nslots * 5 * stride words: rank[stride], sorted[stride], tmp[stride] (two words per entry)
  uint64_t stride;     // positions per slot
};

// The launch arguments as they lie in the kernel argument segment; read from there where needed
// instead of being held in scalar registers across a block (the search loops need those).
struct HcKArgs {
  BatchArgs a;
  uint32_t max_depth;
  HcScratch sc;
  uint32_t *counter;
};
typedef const volatile __attribute__((address_space(4))) HcKArgs *HcKArgsPtr;

#ifndef LZ4MI_HC_WPE
#define LZ4MI_HC_WPE 4
#endif
template <bool HAS_DICT, bool LINKED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LZ4MI_HC_WPE, LZ4MI_HC_WPE))) void encode_hc_kernel(BatchArgs, uint32_t, HcScratch, uint32_t *) {
  __shared__ uint32_t s_hist[4][512];
  HcKArgsPtr ka = (HcKArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
  const uint32_t lane = lane_id();
  const uint32_t slot = uni(blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave);
  lds_u32 *hist = (lds_u32 *)s_hist[uni(threadIdx.x / kWave)];
  const uint64_t stride = ka->sc.stride;
  HcIndex x;
  x.floor_ = ka->sc.floor_ + (size_t)slot * kHashEntries;
  x.rank = ka->sc.index + (size_t)slot * 5 * stride;
  x.sorted = (uint2 *)(x.rank + stride);
  x.tmp = x.sorted + stride;
  const uint32_t max_depth = ka->max_depth;
  // with a wavefront per block the block index is the wavefront's own (no contended queue pops at start)
  const uint32_t nwaves = gridDim.x * (blockDim.x / kWave);
  const bool direct = ka->a.nblocks <= nwaves;
  // a wavefront's first block is its own index (thousands of wavefronts popping one queue head at
  // launch serialise on it); the queue hands out the blocks behind those
  uint32_t b = uni(blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave);
  for (;; b = direct ? 0xffffffffu : nwaves + next_block(ka->counter, lane)) {
    if (b >= ka->a.nblocks) break;
    EncIn in;
    in.dict = ka->a.dict;
    in.D = HAS_DICT ? ka->a.dict_len : 0u;
    const uint64_t *src_off = ka->a.src_off, *dst_off = ka->a.dst_off;
    const uint32_t *src_len_v = ka->a.src_len, *dst_cap = ka->a.dst_cap;
    in.src = ka->a.src + uni64(src_off ? src_off[b] : (uint64_t)b * ka->a.src_stride);
    if (LINKED) {  // lz4_frame_encoder.dart:176-180; the encoder keeps the last 65535 bytes (:62-72)
      in.dict = b == 0 ? in.dict : in.src - 65535;
      in.D = b == 0 ? ka->a.dict_len : 65535u;
    }
    uint32_t n;
    if (src_len_v) {
      n = uni(src_len_v[b]);
    } else {
      const uint32_t u = ka->a.src_len_uniform;
      const uint64_t start = (uint64_t)b * u, tot = ka->a.src_total;
      const uint64_t left = tot > start ? tot - start : 0;
      n = left < u ? (uint32_t)left : u;
    }
    in.total = in.D + n;
    set_end_limits(in, ka->a.conformant);
    uint8_t *out = ka->a.dst + uni64(dst_off ? dst_off[b] : (uint64_t)b * ka->a.dst_stride);
    const uint32_t cap = uni(dst_cap ? dst_cap[b] : ka->a.dst_cap_uniform);
    uint32_t out_len = 0;
    int32_t st = (uint64_t)in.total > stride ? (int32_t)LZ4MI_UNSUPPORTED_HERE
                                             : encode_block_hc<HAS_DICT>(in, out, cap, max_depth, x, hist, lane, out_len);
    if (lane == 0) {
      ka->a.out_len[b] = out_len;
      ka->a.status[b] = st;
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
  // index capacity: dictionary plus the largest block this batch can contain
  uint64_t max_len = a.src_len ? (a.dst_cap_uniform ? a.dst_cap_uniform : (4u << 20)) : a.src_len_uniform;
  if (max_len > 0x7FFFFFF0u) max_len = 0x7FFFFFF0u;
  const uint64_t stride = round_up(max_len + (a.linked ? 65535u : a.dict_len) + 64, 64);
  // Index scratch: 20 bytes per position and slot.  Up to 8 GiB in all for a batch; a call whose single
  // slot needs more (one block of 400 MiB and more) may take what the device has free, and one that does not
  // fit there either is refused with the reason instead of failing inside hipMalloc.
  const uint64_t per_slot = stride * 20;
  uint64_t budget = 8ull << 30;
  if (per_slot > budget) {
    size_t free_b = 0, total_b = 0;
    LZ4MI_HIP(c, hipMemGetInfo(&free_b, &total_b));
    const uint64_t have = (uint64_t)free_b + c->hc_chain.cap;  // what a regrown buffer could take
    if (per_slot + (1ull << 30) > have) {
      c->err = "LZ4HC index of one block of " + std::to_string(max_len) + " bytes needs " + std::to_string(per_slot) +
               " bytes of device scratch, " + std::to_string(have) + " are free";
      return LZ4MI_UNSUPPORTED_HERE;
    }
    budget = per_slot;
  }
  const uint32_t max_slots = (uint32_t)c->cu_count * 4u * LZ4MI_HC_WPE;  // the wavefronts that stay resident
  const uint32_t want = a.nblocks < max_slots ? a.nblocks : max_slots;
  const uint32_t fit = (uint32_t)(budget / per_slot < max_slots ? budget / per_slot : max_slots);  // >= 1
  // wavefronts per workgroup: four, fewer when fewer blocks or slots than that exist
  uint32_t wpw = 4, slots;
  if (want < 4 || fit < 4) {
    slots = wpw = want < fit ? want : fit;
  } else {
    slots = (uint32_t)round_up(want, 4);
    if (slots > fit) slots = fit & ~3u;
  }
  // a launch with a wavefront per block leaves the slots behind the last block untouched: no memory for them
  const uint32_t mem_slots = slots < a.nblocks ? slots : a.nblocks;
  const size_t tbl = (size_t)mem_slots * kHashEntries * sizeof(uint32_t);
  LZ4MI_HIP(c, c->hc_floor.ensure(tbl));
  LZ4MI_HIP(c, c->hc_chain.ensure((size_t)mem_slots * per_slot));
  HcScratch sc;
  sc.floor_ = c->hc_floor.as<uint32_t>();
  sc.index = c->hc_chain.as<uint32_t>();
  sc.stride = stride;
  uint32_t *counter = c->d_counters + 5;  // word 5: HC queue head (0-2 decode, 4 fast encode)
  LZ4MI_HIP(c, hipMemsetAsync(counter, 0, sizeof(uint32_t), c->stream));
  const uint32_t grid = slots / wpw;
  const dim3 wg(wpw * kWave);
  uint32_t d32 = depth > 0x7fffffff ? 0x7fffffffu : (uint32_t)depth;
  if (a.linked)
    hipLaunchKernelGGL((encode_hc_kernel<true, true>), dim3(grid), wg, 0, c->stream, a, d32, sc, counter);
  else if (a.dict_len)
    hipLaunchKernelGGL((encode_hc_kernel<true, false>), dim3(grid), wg, 0, c->stream, a, d32, sc, counter);
  else
    hipLaunchKernelGGL((encode_hc_kernel<false, false>), dim3(grid), wg, 0, c->stream, a, d32, sc, counter);
  LZ4MI_HIP(c, hipGetLastError());
  return LZ4MI_OK;
}

}  // namespace lz4mi
