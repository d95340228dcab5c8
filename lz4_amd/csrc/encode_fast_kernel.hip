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
// Table: 65536 x u32 per wavefront slot in HBM/L2.  Three entry formats, one per launch:
//   * generation<<24 | (pos+1)<<7 | tag   while every position of block and dictionary fits 17 bits (TAG): the
//     tag is seven bits of the hashed sequence that the bucket index does not cover, so a candidate whose tag
//     differs is dropped without fetching its bytes.  The tag sits below the position because inserts are
//     maxima: a larger position must compare larger whatever the tags are;
//   * generation<<24 | (pos+1)            up to 16 MiB;
//   * pos+1                               above that (WIDE: one lz4Compress call on a large buffer), the slot
//     cleared in front of the block.
// A stale generation reads as empty, so a slot is reused without clearing.
// A shared dictionary is primed once into a read-only table consulted when the
// slot entry is empty (dictionary positions are always smaller than block positions).
//
// Two code paths produce the same bytes.  The general path takes any sequence: ten probes in the first step,
// 64 at a time after that, dictionary and linked history, references anywhere.  In front of it stands a short
// path (fast_sequence) for the case that nine of ten sequences of match-rich data are: it checks every one of
// its assumptions before its first write, so a sequence it cannot take starts over in the general path with
// nothing to undo.
#include "encode_common.h"
#include "profile.h"

namespace lz4mi {

#ifdef LZ4MI_PROFILE
__device__ unsigned long long g_enc_phase[PH_SLOTS];
__device__ unsigned long long g_fast_exit[16];  // how often the short path left at each of its exits
#define FAST_EXIT(k) do { if (lane == 0) atomicAdd(&g_fast_exit[k], 1ull); } while (0)
#else
#define FAST_EXIT(k) do {} while (0)
#endif
#if defined(LZ4MI_PROFILE) || defined(LZ4MI_SPANS)
#define LZ4MI_ENC_SPANS 1
__device__ unsigned long long g_enc_span[2 * 65536];  // per block: start and end (100 MHz ticks)
#endif

// ---------------------------------------------------------------- input window
// An 8 KiB ring of the block's bytes around the current anchor lives in LDS (one ring per
// wavefront): probe reads, reference checks, match extension, literal bytes and the interior
// re-hash of the common case (reference closer than ~6 KiB, match shorter than ~256 bytes) are
// LDS reads.  That matters because vector-memory operations of a wavefront complete in issue
// order: a global load issued after table atomics or output stores cannot return before those
// have drained (~600-950 cycles each), whereas LDS reads are independent of them.  With the
// ring, one sequence costs a single vector-memory wait (the hash-table gather of its probes);
// everything written for a sequence (output bytes, table inserts) is issued after the last
// read of that sequence.
// The first 16 bytes of the ring are mirrored behind its end so that a 4- or 8-byte read (the
// aligned dwords it is put together from) never has to be split at the wrap.
#ifndef LZ4MI_WIN
#define LZ4MI_WIN 8192
#endif
constexpr uint32_t kWinBytes = LZ4MI_WIN, kWinMask = kWinBytes - 1, kSeg = 1024, kWinPad = 16;
constexpr uint32_t kClaimSlots = 256;  // LDS dwords per wavefront for the same-bucket vote of the re-hash
#ifndef LZ4MI_FIRST
#define LZ4MI_FIRST 10
#endif
constexpr uint32_t kFirst = LZ4MI_FIRST;  // probes of a run evaluated in the first step (bench encode launch: 8 -> 2.74-2.79 ms, 9-11 -> 2.69-2.73, 16 and 4 clearly slower)
constexpr uint32_t kLead = kFirst > 8 ? 1024 : 640;  // an iteration may read ring bytes up to anchor + kLead


struct Win {
  lds_u8 *lds;         // this wavefront's ring (16-byte aligned, kWinBytes + kWinPad bytes)
  const uint8_t *src;  // block bytes
  uint32_t n;          // block length (>= 4)
  uint32_t lo, hi;     // ring holds block offsets [lo, hi), both multiples of kSeg; bytes at or past n read as zero
};

// Ring reads by block offset (the caller guarantees residency).  An LDS read whose address is not a
// multiple of its size stalls the CU's LDS pipe for about twenty cycles (sixteen wavefronts share that
// pipe: measured 304 cycles per unaligned ds_read_b32 per wavefront against 46 for the form below,
// scripts/micro/lds_unaligned_tp.hip), so the bytes are put together from aligned dwords.
__device__ __forceinline__ uint32_t w32(const Win &w, uint32_t s) {
  const uint32_t a = s & kWinMask;
  const lds_u32 *q = (const lds_u32 *)(w.lds + (a & ~3u));  // q[1] may lie in the wrap mirror
  return __builtin_amdgcn_alignbyte(q[1], q[0], a & 3u);
}
__device__ __forceinline__ uint64_t w64(const Win &w, uint32_t s) {
  const uint32_t a = s & kWinMask;
  const lds_u32 *q = (const lds_u32 *)(w.lds + (a & ~3u));
  const uint32_t w0 = q[0], w1 = q[1], w2 = q[2];
  return ((uint64_t)__builtin_amdgcn_alignbyte(w2, w1, a & 3u) << 32) | __builtin_amdgcn_alignbyte(w1, w0, a & 3u);
}
__device__ __forceinline__ uint32_t w8(const Win &w, uint32_t s) { return w.lds[s & kWinMask]; }

// 16 bytes per lane of the 1 KiB segment at block offset pos (zero past the end of the block).
__device__ __forceinline__ u32x4 seg_load(const Win &w, uint32_t pos, uint32_t lane) {
  const uint32_t o = pos + 16 * lane;
  u32x4 v = {0, 0, 0, 0};
  if (o + 16 <= w.n) {
    U128U t = ld128(w.src + o);
    v = u32x4{t.a, t.b, t.c, t.d};
  } else if (o < w.n) {  // the block's last, ragged 16 bytes: whole dwords, then one shifted out of the final dword
    const uint32_t last = ld32(w.src + w.n - 4);
    uint32_t t[4];
    for (uint32_t k = 0; k < 4; k++) {
      const uint32_t q = o + 4 * k;
      t[k] = q + 4 <= w.n ? ld32(w.src + q) : (q < w.n ? last >> (8 * (q + 4 - w.n)) : 0u);
    }
    v = u32x4{t[0], t[1], t[2], t[3]};
  }
  return v;
}

__device__ __forceinline__ void seg_store(Win &w, uint32_t pos, u32x4 v, uint32_t lane) {
  *(lds_u128 *)(w.lds + ((pos + 16 * lane) & kWinMask)) = v;
  if ((pos & kWinMask) == 0 && lane == 0) *(lds_u128 *)(w.lds + kWinBytes) = v;  // wrap mirror
  w.hi = pos + kSeg;
  if (w.hi - w.lo > kWinBytes) w.lo = w.hi - kWinBytes;
}

// Synchronous fill: makes [s, need) resident (need <= s + kLead, clipped to the block by the caller).
__device__ __forceinline__ void win_fill(Win &w, uint32_t s, uint32_t need, uint32_t lane) {
  if (s < w.lo || s > w.hi) w.lo = w.hi = s & ~(kSeg - 1);  // the anchor jumped out of the ring
  while (w.hi < need) seg_store(w, w.hi, seg_load(w, w.hi, lane), lane);
}

__device__ __forceinline__ bool win_has(const Win &w, uint32_t s, uint32_t len) {
  return s >= w.lo && s + len <= w.hi && s + len <= w.n;
}

// position-based reads of dict||src: ring when resident, memory otherwise
template <bool HAS_DICT>
__device__ __forceinline__ uint32_t rd32(const EncIn &in, const Win &w, uint32_t pos) {
  const uint32_t D = HAS_DICT ? in.D : 0u;
  if (pos >= D && win_has(w, pos - D, 4)) return w32(w, pos - D);
  return in_ld32<HAS_DICT>(in, pos);
}
template <bool HAS_DICT>
__device__ __forceinline__ uint32_t rd8(const EncIn &in, const Win &w, uint32_t pos) {
  const uint32_t D = HAS_DICT ? in.D : 0u;
  if (pos >= D && win_has(w, pos - D, 1)) return w8(w, pos - D);
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

// Equal leading bytes (0..16) of the 16 bytes at p and q (q < p), straight from memory: both loads
// are issued before the compare.  Falls back to the generic path near the block end and for
// references that start inside the dictionary.
template <bool HAS_DICT>
__device__ __forceinline__ uint32_t eq16g(const EncIn &in, const Win &w, uint32_t p, uint32_t q, uint32_t total) {
  const uint32_t D = HAS_DICT ? in.D : 0u;
  if (p + 16 <= total && q >= D) {
    const U128U a = ld128(in.src + (p - D)), b = ld128(in.src + (q - D));
    const uint64_t x0 = (((uint64_t)(a.b ^ b.b)) << 32) | (a.a ^ b.a);
    const uint64_t x1 = (((uint64_t)(a.d ^ b.d)) << 32) | (a.c ^ b.c);
    return x0 ? (uint32_t)__builtin_ctzll(x0) >> 3 : (x1 ? 8u + ((uint32_t)__builtin_ctzll(x1) >> 3) : 16u);
  }
  const uint32_t e = eq8<HAS_DICT>(in, w, p, q, total);
  return e < 8 ? e : 8 + eq8<HAS_DICT>(in, w, p + 8, q + 8, total);
}

// v of lane (this - D) within a row of 16 lanes; lanes without such a source get `fill`.
template <int D>
__device__ __forceinline__ uint32_t row_shr(uint32_t v, uint32_t fill) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x110 + D, 0xF, 0xF, false);
}

// ---------------------------------------------------------------- table entries
// Plain: generation (8 bits) | position + 1 (24 bits).  WIDE: position + 1.
// TAG (blocks whose positions fit 17 bits: 64 KiB blocks, with or without a dictionary): generation | position + 1
// (17 bits) | tag (7 bits), the tag being seven bits of the hashed sequence that the bucket index does not hold
// (below the position, so that the larger entry of a bucket is still the larger position: inserts are maxima).
// An entry whose tag differs from the probe's has other bytes at its position than the probe (the tag is a
// function of the four bytes), so the reference's byte compare (:103) would fail: the candidate is dropped without
// fetching those bytes.  Most candidates are such stale bucket-mates, and most of them lie behind the input ring,
// where looking at them is a round trip to memory.
constexpr uint32_t kHashMul = 2654435761u;  // :167-170: hash = (seq * kHashMul) >> 16
constexpr uint32_t kTagPosBits = 17, kTagPosMask = (1u << kTagPosBits) - 1;
__device__ __forceinline__ uint32_t tag_of(uint32_t prod) { return (prod >> 9) & 0x7Fu; }

template <bool WIDE, bool TAG>
__device__ __forceinline__ uint32_t entry_value(uint32_t gen, uint32_t prod, uint32_t pos) {
  if (WIDE) return pos + 1;
  if (TAG) return (gen << kPosBits) | ((pos + 1) << 7) | tag_of(prod);
  return (gen << kPosBits) | (pos + 1);
}
// Position held by a table entry for a probe with hashed sequence `prod`, or -1.  own: the bucket holds a position
// of this block (so a shared dictionary's table is not asked, whatever the tag says).
template <bool WIDE, bool TAG>
__device__ __forceinline__ int32_t entry_ref(uint32_t e, uint32_t gen, uint32_t prod, bool &own) {
  if (WIDE) {
    own = e != 0;
    return (int32_t)e - 1;
  }
  own = (e >> kPosBits) == gen;
  if (TAG) return own && (e & 0x7Fu) == tag_of(prod) ? (int32_t)((e >> 7) & kTagPosMask) - 1 : -1;
  return own ? (int32_t)(e & kPosMask) - 1 : -1;
}

// Lane masks straight from a compare (a ballot of a bool costs a select and a second compare on the way).
__device__ __forceinline__ uint64_t m_eq(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, 32); }
__device__ __forceinline__ uint64_t m_ne(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, 33); }
__device__ __forceinline__ uint64_t m_uge(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, 35); }
__device__ __forceinline__ uint64_t m_ult(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, 36); }
__device__ __forceinline__ uint64_t m_ule(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, 37); }

// One re-hash step: table[hash] = position for up to 64 ascending positions.  Inserts go out as
// plain stores (a store to a line that is not cached needs no fetch, an atomic does).  Positions
// rise with the lane, so of the lanes of one instruction that share a bucket only the highest may
// store: each lane max-claims a slot of a small LDS array with (lane + 1, hash) and reads the winner
// back.  Same hash: the winner is the bucket's highest lane.  Different hash (two buckets met in one
// slot): undecided; after a second round the few lanes still undecided fall back to the atomic max,
// which is always right.
__device__ __forceinline__ void insert_step(uint32_t *T, lds_u32 *claim, uint32_t hj, uint32_t value, bool live,
                                            uint32_t lane) {
  const uint32_t mine = ((lane + 1) << 16) | hj;
  uint32_t top = mine;  // a lane without a position never stores: `live` is asked again at the store
  if (live) {
    volatile lds_u32 *slot = claim + (hj & (kClaimSlots - 1));
    (void)__hip_atomic_fetch_max((lds_u32 *)slot, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    top = *slot;
    *slot = 0;
  }
  bool open = (top & 0xFFFFu) != hj;
  if (m_ne(top & 0xFFFFu, hj)) {
    // Lanes that met another bucket in their slot vote again among themselves, on the other half of
    // the hash (lanes of one bucket share their fate in every round, so the vote stays complete).
    // An atomic in front of the next table gather costs that gather the atomic's whole round trip,
    // which under load is several times the gather's own; a second LDS round trip is cheap next to it.
    if (open) {
      volatile lds_u32 *slot2 = claim + ((hj >> 8) & (kClaimSlots - 1));
      (void)__hip_atomic_fetch_max((lds_u32 *)slot2, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      top = *slot2;
      *slot2 = 0;
    }
    open = (top & 0xFFFFu) != hj;
  }
#ifndef LZ4MI_TIMING_NO_REHASH_STORES
  uint32_t *p = T + hj;
  if (open) tbl_max(p, value);
  if (top == mine && live) tbl_store(p, value);
#endif
}

// smallest h ^ h(lane - k) over k = 1..K within a row of 16 lanes: zero iff an earlier lane holds the same value
template <int K>
__device__ __forceinline__ uint32_t shr_min(uint32_t h) {
  if constexpr (K == 0) {
    return ~0u;
  } else {
    const uint32_t x = h ^ row_shr<K>(h, ~0u);
    const uint32_t r = shr_min<K - 1>(h);
    return x < r ? x : r;
  }
}

// Writing a sequence (_writeSequence, :172-200) is not on the path to the next match, so it is taken
// off the serial chain: a found sequence is noted down (literal start, literal length, distance, match
// length: four registers, sequence j of the batch in lane j, written with v_writelane) and up to kBatch
// of them are written together.  Lane j sizes sequence j, a prefix
// sum places it, the lane writes its token, length bytes and offset; the literal runs follow one
// after the other, wave-wide, out of the input ring when it still holds them.
constexpr uint32_t kBatch = 32;
struct SeqBatch {  // lane j holds sequence j of the batch
  uint32_t s, lit, dist, ml;
};
// Lane l of the four registers takes a sequence's four (wave-uniform) fields: v_writelane_b32, no exec masking.
// The lane number travels in m0 (an instruction reads one scalar register besides m0), which is handed back
// as it was found.
__device__ __forceinline__ void put_seq(SeqBatch &q, uint32_t s, uint32_t lit, uint32_t dist, uint32_t ml, uint32_t l) {
  uint32_t keep;
  asm("s_mov_b32 %4, m0\n\t"
      "s_mov_b32 m0, %9\n\t"
      "s_nop 0\n\t"
      "v_writelane_b32 %0, %5, m0\n\t"
      "v_writelane_b32 %1, %6, m0\n\t"
      "v_writelane_b32 %2, %7, m0\n\t"
      "v_writelane_b32 %3, %8, m0\n\t"
      "s_mov_b32 m0, %4"
      : "+v"(q.s), "+v"(q.lit), "+v"(q.dist), "+v"(q.ml), "=&s"(keep)
      : "s"(uni(s)), "s"(uni(lit)), "s"(uni(dist)), "s"(uni(ml)), "s"(uni(l)));  // uni: free for values that are scalar already
}
__device__ __forceinline__ uint32_t write_sequences(const Win &w, const SeqBatch &q, uint32_t cnt, uint8_t *out, uint32_t op,
                                                    uint32_t lane) {
  const bool mine = lane < cnt;
  const uint32_t s = mine ? q.s : 0u, lit = mine ? q.lit : 0u, dist = mine ? q.dist : 0u, ml = mine ? q.ml : 4u;
  const uint32_t m4 = ml - 4;
  const uint32_t e1 = lit >= 15 ? (lit - 15) / 255u + 1u : 0u;  // _writeLength :216-223: v / 255 bytes of 255, then v % 255
  const uint32_t mx = m4 >= 15 ? (m4 - 15) / 255u + 1u : 0u;
  const uint32_t size = mine ? 1 + e1 + lit + 2 + mx : 0u;
  const uint32_t incl = wave_incl_scan(size);
  const uint32_t base = op + incl - size;
  const uint32_t lpos = base + 1 + e1;  // where the literals go
  if (mine) {
    out[base] = (uint8_t)(((lit < 15 ? lit : 15u) << 4) | (m4 < 15 ? m4 : 15u));
    for (uint32_t k = 0; k < e1; k++) out[base + 1 + k] = k + 1 < e1 ? (uint8_t)255 : (uint8_t)((lit - 15) % 255u);
    const uint32_t q = lpos + lit;
    out[q] = (uint8_t)dist;
    out[q + 1] = (uint8_t)(dist >> 8);
    for (uint32_t k = 0; k < mx; k++) out[q + 2 + k] = k + 1 < mx ? (uint8_t)255 : (uint8_t)((m4 - 15) % 255u);
  }
  // runs of up to 64 bytes that the ring still holds (nearly all): four per LDS round trip
  const uint64_t slow = ballot(mine && lit != 0 && !(lit <= kWave && s >= w.lo));  // s + lit <= anchor <= w.hi
  for (uint32_t j = 0; j < cnt; j += 4) {
    uint32_t lj[4], dj[4], v[4];
#pragma unroll
    for (uint32_t u = 0; u < 4; u++) {
      const uint32_t jj = j + u;  // lanes at or past cnt hold lit = 0
      lj[u] = ((slow >> jj) & 1ull) ? 0u : rdlane(lit, jj);
      dj[u] = rdlane(lpos, jj);
      v[u] = w8(w, rdlane(s, jj) + lane);
    }
#pragma unroll
    for (uint32_t u = 0; u < 4; u++)
      if (lane < lj[u]) out[dj[u] + lane] = (uint8_t)v[u];
  }
  for (uint64_t m = slow; m; m &= m - 1) {  // long runs, runs the ring has dropped
    const uint32_t j = (uint32_t)__builtin_ctzll(m);
    wave_copy(out + rdlane(lpos, j), w.src + rdlane(s, j), rdlane(lit, j), lane);
  }
  if (slow) vm_quiesce();  // rare path: leave no load pending
  return op + rdlane(incl, kWave - 1);
}

// Interior re-hash of a match (:135-145) that started at ms and ended at `anchor` (exclusive): of the positions
// ms+1 .. anchor-4 only the last `dist` can survive in the table (position j and j + dist hash the same bytes and
// j + dist is inserted later).
template <bool HAS_DICT, bool WIDE, bool TAG>
__device__ __forceinline__ void rehash_match(const EncIn &in, const Win &w, uint32_t *T, lds_u32 *claim, uint32_t gen,
                                             uint32_t ms, uint32_t dist, uint32_t anchor, uint32_t lane) {
  const uint32_t D = HAS_DICT ? in.D : 0u;
  if (anchor + 4 <= in.plimit) {
    const uint32_t hi = anchor - 4;
    uint32_t lo = ms + 1;
    if (anchor >= 3 + dist && anchor - 3 - dist > lo) lo = anchor - 3 - dist;
    if (anchor - D <= w.hi) {  // lo - D >= s >= w.lo
      for (uint32_t j0 = lo; j0 <= hi; j0 += kWave) {
        const uint32_t j = j0 + lane;
        const uint32_t prod = w32(w, j - D) * kHashMul;
        insert_step(T, claim, prod >> 16, entry_value<WIDE, TAG>(gen, prod, j), j <= hi, lane);
      }
    } else {
      for (uint32_t j0 = lo; j0 <= hi; j0 += kWave) {
        const uint32_t j = j0 + lane;
        if (j <= hi) {
          const uint32_t prod = rd32<HAS_DICT>(in, w, j) * kHashMul;
          tbl_max(T + (prod >> 16), entry_value<WIDE, TAG>(gen, prod, j));
        }
      }
      vm_quiesce();  // rare path: leave no load pending at the join
    }
  }
}

// One sequence the short way, for blocks without a dictionary: every assumption of the common case is checked
// on the way and nothing is written before the last of them holds, so a `false` leaves the state as it was and
// the general code below takes the sequence from its start.  The common case: the ring holds what the sequence
// reads and no segment is due; all kFirst probes lie in front of the last searched position; no two of them
// meet in a slot of the vote array (so no two share a bucket); one of them hits; the reference lies inside the
// ring with the margins of the one-round-trip extension; the match is shorter than 32 bytes backwards and 256
// forwards.  What it does is what the general code does (same reads, same table writes, same sequence).
enum { kSeqDone = 0, kSeqGeneral = 1 };
template <bool TAG>
__device__ __forceinline__ int fast_sequence(const EncIn &in, Win &w, uint32_t *T, lds_u32 *claim, uint32_t gen,
                                              uint32_t offP, uint32_t ext_off, uint32_t ext_need, uint32_t lane,
                                              uint32_t &anchor, SeqBatch &batch, uint32_t &nbatch, uint32_t &op,
                                              uint8_t *out, PhState &ph) {
  // the margin of 128 bytes in front of the last searched position covers the probes' reach (offset 61 + 4 bytes with
  // acceleration 1, less with more) and the row shifts below cover nine earlier probes
  static_assert(kFirst <= 10, "the short path is written for at most ten probes in the first step");
  const uint32_t n = in.total, plimit = in.plimit, elimit = in.elimit;
  PHS(ph, 16);
  const uint32_t s = anchor;
  {
    const uint32_t need = s + kLead < n ? s + kLead : n;
    if (s < w.lo || w.hi < need) { FAST_EXIT(1); return kSeqGeneral; }
    if (s + 128 > plimit) { FAST_EXIT(3); return kSeqGeneral; }                      // the block's end (every probe below is a valid one)
  }
  // the next segment is requested together with the table gather and lands after the same wait
  const bool pend = w.hi < n && w.hi < s + kLead + kSeg;
  u32x4 pv = {0, 0, 0, 0};
  if (pend) pv = seg_load(w, w.hi, lane);
  // Lanes behind the first kFirst repeat lane 0's probe (offP is zero there): same addresses, same answers.
  const uint32_t ik = s + offP;
  const uint32_t seq = w32(w, ik);
  const uint32_t prod = seq * kHashMul;
  const uint32_t h = prod >> 16;
  const uint32_t e = tbl_load(T + h);
  PHS(ph, 17);
  // probes that meet in a byte slot of the vote array (all zero between uses) see each other's lane number
  uint32_t seen = lane + 1;
  if (lane < kFirst) {
    volatile lds_u8 *mark = (volatile lds_u8 *)claim + (h & (4 * kClaimSlots - 1));
    *mark = (uint8_t)(lane + 1);
    seen = *mark;
    *mark = 0;
  }
#ifdef LZ4MI_PROFILE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PHS(ph, 18);
#endif
  // the entry's position when its generation (and tag) are the probe's; :102 (such an entry lies in front of ik)
  uint32_t ref = TAG ? ((e >> 7) & kTagPosMask) - 1 : e - ((gen << kPosBits) + 1);
  uint32_t distv = ik - ref;
  uint64_t cand = (TAG ? m_eq((e ^ tag_of(prod)) & 0xFF00007Fu, gen << kPosBits) : m_eq(e >> kPosBits, gen)) & m_ule(distv, 0xFFFFu);
  const uint64_t clash = m_ne(seen, lane + 1);
  if (clash) {
    // Probes that may share a bucket: an earlier probe of the run in the same bucket acts as the table entry
    // (:99-100 in program order), the nearest one.  Frequent where the data repeats at a short distance.
    uint32_t near = 0;
    near = row_shr<9>(h, ~0u) == h ? 9u : near;
    near = row_shr<8>(h, ~0u) == h ? 8u : near;
    near = row_shr<7>(h, ~0u) == h ? 7u : near;
    near = row_shr<6>(h, ~0u) == h ? 6u : near;
    near = row_shr<5>(h, ~0u) == h ? 5u : near;
    near = row_shr<4>(h, ~0u) == h ? 4u : near;
    near = row_shr<3>(h, ~0u) == h ? 3u : near;
    near = row_shr<2>(h, ~0u) == h ? 2u : near;
    near = row_shr<1>(h, ~0u) == h ? 1u : near;
    const uint32_t ikp = (uint32_t)__shfl((int)ik, (int)(lane - near));
    const uint64_t dupl = m_ne(near, 0u);
    cand |= dupl;
    ref = near ? ikp : ref;
    distv = ik - ref;
  }
  cand &= (1ull << kFirst) - 1ull;  // the lanes behind the probes copy lane 0: no hit of theirs counts
  if (cand == 0) {
    if (pend) seg_store(w, w.hi, pv, lane);
    vm_quiesce();
    FAST_EXIT(4);
    return kSeqGeneral;
  }
  uint32_t rb = w32(w, ref);
  const uint64_t far = cand & m_ult(ref, w.lo);
#ifndef LZ4MI_TIMING_NO_FAR
  if (far) {  // references the ring has dropped: their bytes come from memory
    if (__builtin_amdgcn_inverse_ballot_w64(far)) rb = ld32(in.src + ref);
  }
#endif
  const uint64_t hit = cand & m_eq(rb, seq);  // :103
  if (pend) seg_store(w, w.hi, pv, lane);
  vm_quiesce();
  if (hit == 0) { FAST_EXIT(4); return kSeqGeneral; }  // the run goes on behind its first probes: the general code

  PHS(ph, 19);
  // ---- match at lane K
  const uint32_t K = (uint32_t)__builtin_ctzll(hit);
  const uint32_t i = rdlane(ik, K), dist = rdlane(distv, K), r0 = i - dist;
  const uint32_t maxb = i - anchor < r0 ? i - anchor : r0;
  // lanes 0-31 compare one byte each going backwards (:107-113), lanes 32-63 eight bytes each going forwards (:117-121)
  const uint32_t pp = i + ext_off;
  uint64_t x;
  if (r0 >= w.lo + 32 && (i + 272 <= w.hi || w.hi >= n)) {
    x = w64(w, pp) ^ w64(w, pp - dist);
  } else if (r0 >= 32 && i + 272 <= elimit && i + 272 <= w.hi) {
    x = w64(w, pp) ^ ld64(in.src + (pp - dist));  // the reference lies behind the ring: its side comes from memory
    vm_quiesce();
  } else {
    FAST_EXIT(5);
    return kSeqGeneral;
  }
  const uint32_t e8 = x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8u;
  const uint32_t avail = pp < elimit ? elimit - pp : 0u;  // bytes past the match limit never count
  const uint32_t eq = e8 < avail ? e8 : avail;
  // a lane stops the run when it has fewer equal bytes than it looks at (1 backwards, 8 forwards); so does
  // every backward lane at or past maxb
  const uint32_t mb = maxb < 32 ? maxb : 32u;
  const uint64_t sm = m_ult(eq, ext_need) | (0xFFFFFFFFull & ~((1ull << mb) - 1ull));
  const uint32_t sm_lo = (uint32_t)sm, sm_hi = (uint32_t)(sm >> 32);
  if (sm_lo == 0 || sm_hi == 0) { FAST_EXIT(6); return kSeqGeneral; }  // longer than 32 backwards or 256 forwards
  const uint32_t b = (uint32_t)__builtin_ctz(sm_lo);
  const uint32_t L = (uint32_t)__builtin_ctz(sm_hi);
  const uint32_t fwd = 8 * L + rdlane(eq, 32 + L);
  const uint32_t ms = i - b, ml = b + 4 + fwd, lit = ms - anchor;

  PHS(ph, 20);
  // ---- nothing can fail from here on: _writeSequence (deferred), :100 for the probes up to the hit, :132-145
  put_seq(batch, s, lit, dist, ml, nbatch);
  if (++nbatch == kBatch) {
    op = write_sequences(w, batch, nbatch, out, op, lane);
    nbatch = 0;
  }
  anchor = ms + ml;
#ifndef LZ4MI_TIMING_NO_PROBE_INS
  if (lane <= K) {  // :100 for every probe of the step up to the hit; probes that may share a bucket take the maximum
    const uint32_t v = entry_value<false, TAG>(gen, prod, ik);
    if (clash) tbl_max(T + h, v);
    else tbl_store(T + h, v);
  }
#endif
  PHS(ph, 21);
#ifndef LZ4MI_TIMING_NO_REHASH
  rehash_match<false, false, TAG>(in, w, T, claim, gen, ms, dist, anchor, lane);
#endif
  PHS(ph, 22);
  FAST_EXIT(0);
  return kSeqDone;
}


template <bool HAS_DICT, bool LINKED, bool WIDE, bool TAG>
__device__ __forceinline__ int32_t encode_block_fast(const EncIn in, uint8_t *out, uint32_t cap, uint32_t sh,
                                                     uint32_t *T, const uint32_t *__restrict__ dictT,
                                                     uint32_t gen, uint32_t offA, uint32_t offB, lds_u8 *lds,
                                                     lds_u32 *claim, uint32_t lane, uint32_t &out_len) {
  const uint32_t D = HAS_DICT ? in.D : 0u;
  const uint32_t total = in.total;
  const uint32_t plimit = in.plimit, elimit = in.elimit;
  const uint32_t n = total - D;
  out_len = 0;
  if (n == 0) return LZ4MI_OK;  // :45-48
  if ((uint64_t)cap < (uint64_t)n + n / 255u + 16u) return LZ4MI_DST_TOO_SMALL;
  if (total > (WIDE ? 0x7FFFFFF0u : TAG ? kTagPosMask : kPosMask - 1)) return LZ4MI_UNSUPPORTED_HERE;  // position width of the table
  uint32_t anchor = D, op = 0;
  PH_DECL
#ifdef LZ4MI_PROFILE
  PhState phs = {ph_c_, ph_t_};
#else
  PhState phs;
#endif

  if (n >= 4) {  // :74-78
    if (LINKED && D >= 4) {  // :83-88, the block's own dictionary goes into its own table
      for (uint32_t j0 = 0; j0 + 4 <= D; j0 += kWave) {
        const uint32_t j = j0 + lane;
        const bool live = j + 4 <= D;
        const uint32_t prod = live ? ld32(in.dict + j) * kHashMul : 0u;
        insert_step(T, claim, prod >> 16, entry_value<WIDE, TAG>(gen, prod, j), live, lane);
      }
    }
    Win w = {lds, in.src, n, 0, 0};
    // lane constants of the one-round-trip match extension: where a lane looks relative to the hit, how many bytes
    const uint32_t ext_off = lane < 32 ? 0u - 1u - lane : 4u + 8u * (lane - 32);
    const uint32_t ext_need = lane < 32 ? 1u : 8u;
    const uint32_t offP = lane < kFirst ? offA : 0u;  // probe offsets of the first step; the other lanes repeat lane 0
    uint32_t nbatch = 0;  // sequences found and not yet written
    SeqBatch batch = {0, 0, 0, 4};
    while (anchor + 4 <= plimit) {  // one iteration per sequence
#ifndef LZ4MI_NO_FAST_SEQ
      if (!HAS_DICT && !LINKED && !WIDE) {
        if (fast_sequence<TAG>(in, w, T, claim, gen, offP, ext_off, ext_need, lane, anchor, batch, nbatch, op, out, phs) ==
            kSeqDone)
          continue;
      }
#endif
#ifdef LZ4MI_PROFILE
      PHS(phs, 23);  // what a short-path attempt that gave up has cost
      ph_t_ = phs.t;
#endif
      const uint32_t s = anchor - D;
      {
        const uint32_t need = s + kLead < n ? s + kLead : n;
        if (s < w.lo || w.hi < need) win_fill(w, s, need, lane);
      }
      // the next segment is requested together with the table gather and lands after the same wait
      const bool pend = w.hi < n && w.hi < s + kLead + kSeg;
      u32x4 pv = {0, 0, 0, 0};
      if (pend) pv = seg_load(w, w.hi, lane);
      PH(0); PH_CNT(8);

      // ---- first kFirst probes of the run (:95-103): bytes from the ring
      uint32_t ik = anchor + offA;
      bool valid = lane < kFirst && ik + 4 <= plimit;
      uint32_t seq = w32(w, ik - D);
      uint32_t prod = seq * kHashMul;
      uint32_t h = prod >> 16;
      int32_t ref = -1;
      {
        // Every lane loads (idle lanes re-read bucket 0): a gather without a branch around it
        // gives one unconditional wait, which also covers the segment requested above.
        const uint32_t hx = valid ? h : 0u;
        const uint32_t e = tbl_load(T + hx);
        const uint32_t ed = (HAS_DICT && !LINKED) ? dictT[hx] : 0u;
        bool own;
        ref = entry_ref<WIDE, TAG>(e, gen, prod, own);
        if (HAS_DICT && !LINKED && !own) ref = (int32_t)ed - 1;
      }
      uint32_t nvalid = (uint32_t)__builtin_popcountll(ballot(valid));  // >= 1, valid lanes form a prefix
#ifdef LZ4MI_PROFILE
      { uint64_t w_ = ballot(ref >= 0); asm volatile("" ::"s"(w_)); }
#endif
      PH(1);
      bool dup;  // probes of the current step share a bucket: their inserts have to be atomic maxima
      {
        // Earlier probes of the run that land in the same bucket act as the table entry (:99-100
        // in program order).  Equal hashes among the first probes are rare: detect with row shifts, resolve
        // out of line.
#ifdef LZ4MI_DUP_DPP
        const uint32_t mn = shr_min<kFirst - 1>(h);
        dup = ballot(valid && mn == 0) != 0;
#else
        // Every probe marks the byte slot of its hash (1024 slots in the vote array, which is all zero between
        // uses) with its lane number and reads the slot back: two probes in one slot leave one number there, so
        // the other lane sees it.  A shared slot is a shared bucket or two buckets that meet in the slot (one
        // sequence in twenty); the loop below then compares the hashes themselves.  The three LDS operations
        // ride in the shadow of the table gather.
        {
          volatile lds_u8 *mark = (volatile lds_u8 *)claim + (h & (4 * kClaimSlots - 1));  // volatile: what comes back is another lane's
          uint32_t seen = lane + 1;
          if (valid) {
            *mark = (uint8_t)(lane + 1);
            seen = *mark;
            *mark = 0;
          }
          dup = ballot(seen != lane + 1) != 0;
        }
#endif
        if (dup) {
          for (uint32_t j = 0; j + 1 < nvalid; j++) {  // nearest earlier lane with an equal hash wins
            const uint32_t hj = rdlane(h, j), ij = rdlane(ik, j);
            if (valid && lane > j && h == hj) ref = (int32_t)ij;
          }
        }
      }
      PH(2);
      uint64_t hm;
      {
        const bool cand = valid && ref >= 0 && ik - (uint32_t)ref <= 0xFFFFu;  // :102
        const bool rnear = cand && (uint32_t)ref >= D + w.lo;                   // ref + 4 <= ik + 4 <= w.hi
        uint32_t rb = w32(w, (uint32_t)ref - D);
        if (ballot(cand && !rnear)) {
          if (cand && !rnear) rb = in_ld32<HAS_DICT>(in, (uint32_t)ref);
        }
        hm = ballot(cand && rb == seq);  // :103
      }
      if (pend) seg_store(w, w.hi, pv, lane);
      vm_quiesce();
      uint32_t K = hm ? (uint32_t)__builtin_ctzll(hm) : nvalid - 1;
      PH(3);

      // ---- the rest of the run, 64 probes at a time
      if (!hm && nvalid == kFirst) {
        if (valid) {  // :100, all of them missed
          if (dup) tbl_max(T + h, entry_value<WIDE, TAG>(gen, prod, ik));
          else tbl_store(T + h, entry_value<WIDE, TAG>(gen, prod, ik));
        }
        valid = false;
        uint32_t o_last = 0;
        for (uint32_t chunk = 1;; chunk++) {
          uint32_t o = offB;
          if (chunk > 1) {
            o = o_last;
            for (uint32_t t = 0; t < kWave; t++)
              if (t <= lane) o = sched_next(o, sh);
          }
          ik = anchor + o;
          valid = ik + 4 <= plimit;  // a prefix of the lanes
          nvalid = (uint32_t)__builtin_popcountll(ballot(valid));
          if (nvalid == 0) break;
          PH_CNT(8);
          seq = 0; prod = 0; h = 0; ref = -1;
          if (valid) {
            seq = rd32<HAS_DICT>(in, w, ik);
            prod = seq * kHashMul;
            h = prod >> 16;
            const uint32_t e = tbl_load(T + h);
            const uint32_t ed = (HAS_DICT && !LINKED) ? dictT[h] : 0u;
            bool own;
            ref = entry_ref<WIDE, TAG>(e, gen, prod, own);
            if (HAS_DICT && !LINKED && !own) ref = (int32_t)ed - 1;
          }
          {  // same-bucket probes of this chunk, nearest earlier lane: 16 bit-plane votes
            uint64_t peers = valid ? ((1ull << lane) - 1ull) : 0ull;
            for (uint32_t bit = 0; bit < 16; bit++) {
              const bool set = (h >> bit) & 1u;
              const uint64_t m = ballot(set);
              peers &= set ? m : ~m;
            }
            const uint32_t j = peers ? 63u - (uint32_t)__builtin_clzll(peers) : lane;
            const uint32_t ij = (uint32_t)__shfl((int)ik, (int)j);
            if (peers) ref = (int32_t)ij;
            dup = ballot(peers != 0) != 0;
          }
          bool hit = false;
          if (valid && ref >= 0 && ik - (uint32_t)ref <= 0xFFFFu) hit = rd32<HAS_DICT>(in, w, (uint32_t)ref) == seq;
          hm = ballot(hit);
          K = hm ? (uint32_t)__builtin_ctzll(hm) : nvalid - 1;
          if (hm) break;  // the probes up to the hit are inserted once the match is known
          if (valid) {
            if (dup) tbl_max(T + h, entry_value<WIDE, TAG>(gen, prod, ik));
            else tbl_store(T + h, entry_value<WIDE, TAG>(gen, prod, ik));
          }
          valid = false;
          if (nvalid < kWave) break;
          o_last = rdlane(o, kWave - 1);
        }
        vm_quiesce();
      }
      if (!hm) break;  // the schedule ran past the last searched position without a hit
      PH(14);

      // ---- match at lane K
      PH_CNT(9);
      const uint32_t i = rdlane(ik, K);
      const uint32_t r0 = rdlane((uint32_t)ref, K);
      const uint32_t dist = i - r0;
      const uint32_t maxb = i - anchor < r0 ? i - anchor : r0;
      // One round trip for both directions: lanes 0-31 compare one byte each going backwards
      // (:107-113), lanes 32-63 compare 8 bytes each going forwards (:117-121).
      const bool enear = r0 >= D + w.lo + 32 && (i - D + 272 <= w.hi || w.hi >= n);
      bool stop;
      uint32_t eq = 0;
      if (enear) {
        const uint32_t pp = lane < 32 ? i - 1 - lane : i + 4 + 8 * (lane - 32);
        const uint64_t x = w64(w, pp - D) ^ w64(w, pp - dist - D);
        const uint32_t e8 = x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8u;
        const uint32_t avail = pp < elimit ? elimit - pp : 0u;  // bytes past the match limit never count
        eq = e8 < avail ? e8 : avail;
        stop = lane < 32 ? !(lane < maxb && e8 != 0) : eq < 8;
      } else if (r0 >= D + 32 && i + 272 <= elimit && i - D + 272 <= w.hi) {
        // reference outside the ring: its side comes from memory (one unconditional 8-byte load per lane)
        PH_CNT(11);
        const uint32_t pp = lane < 32 ? i - 1 - lane : i + 4 + 8 * (lane - 32);
        const uint64_t x = w64(w, pp - D) ^ ld64(in.src + (pp - dist - D));
        const uint32_t e8 = x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8u;
        eq = e8;
        stop = lane < 32 ? !(lane < maxb && e8 != 0) : eq < 8;
      } else {
        if (lane < 32) {
          stop = !(lane < maxb && rd8<HAS_DICT>(in, w, i - 1 - lane) == rd8<HAS_DICT>(in, w, r0 - 1 - lane));
        } else {
          const uint32_t pp = i + 4 + 8 * (lane - 32);
          eq = eq8<HAS_DICT>(in, w, pp, pp - dist, elimit);
          stop = eq < 8;
        }
      }
      const uint64_t sm = ballot(stop);
      uint32_t b = (uint32_t)sm ? (uint32_t)__builtin_ctz((uint32_t)sm) : 32u;
      while (b >= 32 && b < maxb) {  // long backward run: 64 bytes per step
        const uint32_t t = b + lane;
        const bool same = t < maxb && rd8<HAS_DICT>(in, w, i - 1 - t) == rd8<HAS_DICT>(in, w, r0 - 1 - t);
        const uint64_t m = ballot(!same);
        if (m) { b += (uint32_t)__builtin_ctzll(m); break; }
        b += kWave;
      }
      const uint32_t ms = i - b;
      PH(4);
      uint32_t fwd;
      if ((uint32_t)(sm >> 32)) {
        const uint32_t L = (uint32_t)__builtin_ctz((uint32_t)(sm >> 32));
        fwd = 8 * L + rdlane(eq, 32 + L);
      } else {  // longer than 256 bytes: 1 KiB per step
        fwd = 256;
        // a very long match (a run, a short period) is one dependent chain of round trips: 4 KiB per
        // trip while that much lies in front of the match limit and the reference side is block data
        while (i + 4 + fwd + 4096 <= elimit && i + 4 + fwd - dist >= D) {
          const uint8_t *pa = in.src + (i + 4 + fwd - D) + 16 * lane, *pb = pa - dist;
          U128U a[4], c[4];
#pragma unroll
          for (int k = 0; k < 4; k++) { a[k] = ld128(pa + 1024 * k); c[k] = ld128(pb + 1024 * k); }
          uint32_t e16[4];
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const uint64_t x0 = (((uint64_t)(a[k].b ^ c[k].b)) << 32) | (a[k].a ^ c[k].a);
            const uint64_t x1 = (((uint64_t)(a[k].d ^ c[k].d)) << 32) | (a[k].c ^ c[k].c);
            e16[k] = x0 ? (uint32_t)__builtin_ctzll(x0) >> 3 : (x1 ? 8u + ((uint32_t)__builtin_ctzll(x1) >> 3) : 16u);
          }
          const uint64_t m0 = ballot(e16[0] < 16), m1 = ballot(e16[1] < 16), m2 = ballot(e16[2] < 16), m3 = ballot(e16[3] < 16);
          if (m0 | m1 | m2 | m3) break;  // the end is inside these 4 KiB: the 1 KiB loop below finds it
          fwd += 4096;
        }
        for (;;) {
          const uint32_t pp = i + 4 + fwd + 16 * lane;
          const uint32_t e16 = eq16g<HAS_DICT>(in, w, pp, pp - dist, elimit);
          const uint64_t m = ballot(e16 < 16);
          if (m) {
            const uint32_t L = (uint32_t)__builtin_ctzll(m);
            fwd += 16 * L + rdlane(e16, L);
            break;
          }
          fwd += 16 * kWave;
        }
      }
      vm_quiesce();  // every read of this sequence is done; only writes follow
      const uint32_t ml = b + 4 + fwd;
      const uint32_t lit = ms - anchor;
      PH(5);

      // ---- _writeSequence :172-200, deferred: the sequence is noted down and written with its batch
      put_seq(batch, s, lit, dist, ml, nbatch);
      if (++nbatch == kBatch) {
        op = write_sequences(w, batch, nbatch, out, op, lane);
        nbatch = 0;
      }
      anchor = ms + ml;  // :132-133
      if (valid && lane <= K) {  // :100 for every probe of the last step up to the hit
        // Every insert is at a larger position than the bucket held, and a wavefront's stores to one
        // address land in issue order, so a plain store does as long as no two lanes of THIS
        // instruction share a bucket.
        if (dup) tbl_max(T + h, entry_value<WIDE, TAG>(gen, prod, ik));
        else tbl_store(T + h, entry_value<WIDE, TAG>(gen, prod, ik));
      }
      PH(6);

      // ---- :135-145 interior re-hash, last `dist` positions only
      rehash_match<HAS_DICT, WIDE, TAG>(in, w, T, claim, gen, ms, dist, anchor, lane);
      PH(7);
    }
    if (nbatch) op = write_sequences(w, batch, nbatch, out, op, lane);
  }
  // _writeLastLiterals :153-156, :202-214
  const uint32_t last = total - anchor;
  if (last != 0) op = emit_last_literals(out, op, in.src + (anchor - D), last, lane);
  PH(10);
  PH_FLUSH(g_enc_phase, lane);
  out_len = op;
  return LZ4MI_OK;
}

// The launch arguments as they lie in the kernel argument segment.  The block loop keeps its
// control state in scalar registers and needs nearly all of them; holding the arguments there too
// made the compiler spill scalars inside the sequence loop, so they are read again from the
// segment at the few places that need them.
struct EncodeKArgs {
  BatchArgs a;
  uint32_t sh;
  EncodeScratch sc;
  uint32_t *counter;
};
typedef const volatile __attribute__((address_space(4))) EncodeKArgs *EncodeKArgsPtr;

// Wavefronts of one workgroup share nothing but the workgroup's LDS allocation, which the CU gets back only
// when the last of them has ended.
#ifndef LZ4MI_ENC_WPW
#define LZ4MI_ENC_WPW 4
#endif
constexpr uint32_t kEncWpw = LZ4MI_ENC_WPW;  // wavefronts per workgroup
#ifndef LZ4MI_ENC_WPE
#define LZ4MI_ENC_WPE 4  // resident wavefronts per SIMD the register budget is set for (with the ring, sixteen per CU)
#endif

template <bool HAS_DICT, bool LINKED, bool WIDE = false, bool TAG = false>
__global__ __launch_bounds__(64 * kEncWpw) __attribute__((amdgpu_waves_per_eu(LZ4MI_ENC_WPE, LZ4MI_ENC_WPE))) void encode_fast_kernel(BatchArgs, uint32_t, EncodeScratch, uint32_t *) {
  __shared__ __attribute__((aligned(16))) uint8_t s_win[kEncWpw][kWinBytes + kWinPad];
  __shared__ uint32_t s_claim[kEncWpw][kClaimSlots];
  EncodeKArgsPtr ka = (EncodeKArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
  const uint32_t lane = lane_id();
  const uint32_t wave = uni(threadIdx.x / kWave);
  const uint32_t slot = uni(blockIdx.x * (blockDim.x / kWave) + wave);
#ifdef LZ4MI_PROFILE
  const unsigned long long life0 = __builtin_amdgcn_s_memrealtime();
#endif
  uint32_t *T = ka->sc.tables + (size_t)slot * kHashEntries;
  uint32_t gen = uni(ka->sc.slot_gen[slot]);
  for (uint32_t k = lane; k < kClaimSlots; k += kWave) s_claim[wave][k] = 0;
  const uint32_t sh = ka->sh;

  // probe schedule: offA = off(lane), offB = off(lane + kFirst)
  uint32_t offA = 0, offB = 0;
  {
    uint32_t o = 0;
    for (uint32_t t = 0; t < kWave + kFirst; t++) {
      if (t == lane) offA = o;
      if (t == lane + kFirst) offB = o;
      o = sched_next(o, sh);
    }
  }

  // with a wavefront per block the block index is the wavefront's own (no contended queue pops at start)
  const uint32_t nwaves = gridDim.x * (blockDim.x / kWave);
  const bool direct = ka->a.nblocks <= nwaves;
  // a wavefront's first block is its own index (thousands of wavefronts popping one queue head at
  // launch serialise on it); the queue hands out the blocks behind those
  uint32_t b = uni(blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave);
  for (;; b = direct ? 0xffffffffu : nwaves + next_block(ka->counter, lane)) {
    if (b >= ka->a.nblocks) break;
    if (WIDE || ++gen > kGenMax) {  // generation wrapped (or entries carry no generation): clear the slot
      for (uint32_t k = lane * 4; k < kHashEntries; k += kWave * 4) *(uint4 *)(T + k) = make_uint4(0, 0, 0, 0);
      gen = 1;
    }
    EncIn in;
    in.dict = ka->a.dict;
    in.D = ka->a.dict_len;
    const uint64_t *src_off = ka->a.src_off, *dst_off = ka->a.dst_off;
    const uint32_t *src_len_v = ka->a.src_len, *dst_cap = ka->a.dst_cap;
    in.src = ka->a.src + uni64(src_off ? src_off[b] : (uint64_t)b * ka->a.src_stride);
    if (LINKED) {  // lz4_frame_encoder.dart:176-180; the encoder keeps the last 65535 bytes (:50-61)
      in.dict = b == 0 ? in.dict : in.src - 65535;
      in.D = b == 0 ? in.D : 65535u;
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
    const uint32_t *dictT = ka->sc.dict_table;
    uint32_t out_len = 0;
#ifdef LZ4MI_ENC_SPANS
    if (lane == 0 && b < 65536) g_enc_span[2 * b] = __builtin_amdgcn_s_memrealtime();
#endif
    int32_t st = encode_block_fast<HAS_DICT, LINKED, WIDE, TAG>(in, out, cap, sh, T, dictT, gen, offA, offB,
                                                     (lds_u8 *)s_win[wave], (lds_u32 *)s_claim[wave], lane, out_len);
#ifdef LZ4MI_ENC_SPANS
    if (lane == 0 && b < 65536) g_enc_span[2 * b + 1] = __builtin_amdgcn_s_memrealtime();
#endif
    if (lane == 0) {
      ka->a.out_len[b] = out_len;
      ka->a.status[b] = st;
    }
  }
  // wide entries would read as tagged ones: the next block on this slot starts with a clear
  if (lane == 0) ka->sc.slot_gen[slot] = WIDE ? kGenMax : gen;
#ifdef LZ4MI_PROFILE
  if (lane == 0) atomicMax(&g_enc_phase[13], __builtin_amdgcn_s_memrealtime() - life0);
#endif
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

#ifdef LZ4MI_ENC_SPANS
extern "C" void lz4mi_debug_enc_spans(unsigned long long *out, int n) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_enc_span), sizeof(unsigned long long) * 2 * n);
}
#endif
#ifdef LZ4MI_PROFILE
extern "C" void lz4mi_debug_fast_exits(unsigned long long *out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fast_exit), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fast_exit), z, sizeof z);
  }
}
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
  // blocks whose positions need more than 24 bits (only with uniform lengths: one large block per call)
  const bool wide = !a.src_len && !a.linked && (uint64_t)a.src_len_uniform + a.dict_len > kPosMask - 2;
  // every position of every block fits the tagged entry's 17 bits (uniform block length known here)
  const bool tag = !a.src_len && (uint64_t)a.src_len_uniform + (a.linked ? 65535u : a.dict_len) <= kTagPosMask;
  const uint32_t wgs = grid * 4u / kEncWpw;  // `grid` counts groups of four wavefronts
  if (tag && a.linked)
    hipLaunchKernelGGL((encode_fast_kernel<true, true, false, true>), dim3(wgs), dim3(64 * kEncWpw), 0, s, a, accel_shift, sc, counter);
  else if (tag && a.dict_len)
    hipLaunchKernelGGL((encode_fast_kernel<true, false, false, true>), dim3(wgs), dim3(64 * kEncWpw), 0, s, a, accel_shift, sc, counter);
  else if (tag)
    hipLaunchKernelGGL((encode_fast_kernel<false, false, false, true>), dim3(wgs), dim3(64 * kEncWpw), 0, s, a, accel_shift, sc, counter);
  else if (wide && a.dict_len)
    hipLaunchKernelGGL((encode_fast_kernel<true, false, true>), dim3(wgs), dim3(64 * kEncWpw), 0, s, a, accel_shift, sc, counter);
  else if (wide)
    hipLaunchKernelGGL((encode_fast_kernel<false, false, true>), dim3(wgs), dim3(64 * kEncWpw), 0, s, a, accel_shift, sc, counter);
  else if (a.linked)
    hipLaunchKernelGGL((encode_fast_kernel<true, true>), dim3(wgs), dim3(64 * kEncWpw), 0, s, a, accel_shift, sc, counter);
  else if (a.dict_len)
    hipLaunchKernelGGL((encode_fast_kernel<true, false>), dim3(wgs), dim3(64 * kEncWpw), 0, s, a, accel_shift, sc, counter);
  else
    hipLaunchKernelGGL((encode_fast_kernel<false, false>), dim3(wgs), dim3(64 * kEncWpw), 0, s, a, accel_shift, sc, counter);
}

}  // namespace lz4mi
