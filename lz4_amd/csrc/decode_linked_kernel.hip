// Linked-block frames decoded without the chain: every block of the frame is parsed at once, and what a
// serial decoder resolves byte after byte (a match copies bytes that earlier matches wrote) is resolved for
// all bytes of the frame together by pointer doubling.
//
// The reference decodes a dependent block into a writer that starts with the last 64 KiB of the frame's
// output, topped up from the dictionary (_decodeDependentBlock, lib/src/frame/lz4_frame_decoder.dart:326-382;
// lz4BlockDecompressInto, lib/src/block/lz4_block_decoder.dart:78-119; ByteWriter.copyMatch,
// lib/src/internal/byte_writer.dart:93-121).  Output byte p of a match with distance d equals byte p - d;
// followed back far enough, every byte of the frame equals a literal (or a byte of a stored block, or of
// the dictionary).  So:
//   1. link_walk_kernel<false>: one wavefront per block walks the sequences and reports the decoded size
//      (the blocks' output offsets follow on the host, where the per-block checks of the frame layer run);
//   2. link_walk_kernel<true>: the same walk writes the literals to their place in the output and, for every
//      byte of a match, the distance into an array D of one word per output byte (0 = the byte is in place);
//   3. link_jump_kernel, repeated: D[p] += D[p - D[p]] while p - D[p] is itself a copied byte; the lengths
//      of the copy chains halve with every round, and a tile in which nothing changed is left out from then on;
//   4. link_gather_kernel: out[p] = out[p - D[p]] for the copied bytes.
// The walk takes the well-formed case only: anything the reference would throw on (and anything unusual) marks
// the block, and the caller runs the serial chain instead, which names the failure.  The frame is taken in
// chunks of blocks so that D stays bounded; bytes in front of a chunk are final when the chunk is resolved.
#include "device_common.h"
#include "kernels.h"

namespace lz4mi {

namespace {

constexpr uint32_t kLw = 4096;    // compressed bytes a wavefront holds in LDS
constexpr uint32_t kLwPad = 32;   // readable slack behind them
constexpr uint32_t kLwNear = 400; // a window with fewer valid bytes than this behind the position is loaded anew

// window bytes [0, wlim) = payload bytes [wb, wb + wlim)
__device__ __forceinline__ void lw_load(lds_u8 *cw, const uint8_t *__restrict__ in, uint32_t n_in, uint32_t ip,
                                        uint32_t lane, uint32_t &wb, uint32_t &wlim) {
  wb = ip;
  const uint32_t avail = n_in - ip;
  wlim = avail < kLw ? avail : kLw;
  const uint8_t *g = in + ip;
  U128U v[4];
#pragma unroll
  for (uint32_t k = 0; k < 4; k++) {
    const uint32_t off = 1024u * k + 16u * lane;
    v[k] = U128U{0, 0, 0, 0};
    if (off + 16u <= wlim) v[k] = ld128(g + off);
  }
  const uint32_t tail0 = wlim & ~15u;
  uint8_t tb = 0;
  if (lane < wlim - tail0) tb = g[tail0 + lane];
#pragma unroll
  for (uint32_t k = 0; k < 4; k++) {
    const uint32_t off = 1024u * k + 16u * lane;
    if (off + 16u <= wlim) *(lds_u128 *)(cw + off) = u32x4{v[k].a, v[k].b, v[k].c, v[k].d};
  }
  if (lane < wlim - tail0) cw[tail0 + lane] = tb;
  vm_quiesce();
}

// four bytes at window offset a, from aligned dwords (an LDS access that is not aligned to its size stalls
// the LDS pipe of the CU); offsets past the slack read the slack
__device__ __forceinline__ uint32_t lw32(const lds_u8 *cw, uint32_t a) {
  a = a < kLw + kLwPad - 8u ? a : kLw + kLwPad - 8u;
  const lds_u32 *q = (const lds_u32 *)(cw + (a & ~3u));
  return __builtin_amdgcn_alignbyte(q[1], q[0], a & 3u);
}

__device__ __forceinline__ void fill_words(uint32_t *d, uint32_t n, uint32_t v, uint32_t lane) {
  for (uint32_t k = lane; k < n; k += kWave) d[k] = v;
}

}  // namespace

// status written per block: 0 = walked to its end, 1 = not for this path
template <bool PLAN>
__global__ __launch_bounds__(256) void link_walk_kernel(LinkArgs a) {
  __shared__ __attribute__((aligned(16))) uint8_t s_cw[4][kLw + kLwPad];
  const uint32_t lane = lane_id();
  const uint32_t wave = uni(threadIdx.x / kWave);
  const uint32_t k = blockIdx.x * 4u + wave;
  if (k >= a.nblocks) return;
  const uint32_t b = a.first + k;
  lds_u8 *cw = (lds_u8 *)s_cw[wave];
  const uint8_t *__restrict__ in = a.src + a.payload_off[b];
  const uint32_t n_in = a.payload_len[b];
  const uint64_t start = PLAN ? a.start[b] : 0;
  uint8_t *out = PLAN ? a.dst + start : nullptr;
  uint32_t *D = PLAN ? a.dist + (start - a.chunk_start) : nullptr;
  if (a.raw[b]) {  // a stored block is copied; its bytes are in place for the blocks behind it
    if (PLAN) {
      if (n_in >= kCopyLong) wave_copy_long(out, in, n_in, lane);
      else wave_copy(out, in, n_in, lane);
    }
    if (lane == 0) {
      a.out_len[b] = n_in;
      a.status[b] = 0;
    }
    return;
  }
  // what a match of this block may reach in front of the block: 64 KiB of output, topped up from the dictionary
  // (without the block's offset, in the sizing walk, every distance passes here and is checked by the second walk)
  uint32_t back = 65536u;
  if (PLAN) {
    back = start < 65536u ? (uint32_t)start : 65536u;
    if (back < 65536u && a.dict_len) back += a.dict_len < 65536u - back ? a.dict_len : 65536u - back;
  }
  const uint32_t cap = a.block_max;
  uint32_t ip = 0, op = 0, wb = 0, wlim = 0;
  bool have = false, unfit = false;
  while (ip < n_in && !unfit) {
    if (!have || (ip - wb + kLwNear > wlim && wb + wlim < n_in)) {
      lw_load(cw, in, n_in, ip, lane, wb, wlim);
      have = true;
    }
    const uint32_t a0 = ip - wb;
    // ---- every lane reads the byte it stands on as a token, in the common form (each length with at most one
    // extension byte); the real tokens are then followed from lane 0 with one cross-lane read each
    const uint32_t at = a0 + lane;
    const uint32_t t4 = lw32(cw, at);
    const uint32_t ll0 = (t4 >> 4) & 15u, mc = t4 & 15u, e1 = (t4 >> 8) & 255u;
    const bool isl = ll0 == 15u, ism = mc == 15u;
    const uint32_t ll = ll0 + (isl ? e1 : 0u);
    const uint32_t ls = at + 1u + (isl ? 1u : 0u);
    const uint32_t q = ls + ll;
    const uint32_t o4 = lw32(cw, q);
    const uint32_t dist = o4 & 0xffffu, f1 = (o4 >> 16) & 255u;
    const uint32_t ml = mc + 4u + (ism ? f1 : 0u);
    const uint32_t nx = q + 2u + (ism ? 1u : 0u);
    const bool form = !(isl && e1 == 255u) && !(ism && f1 == 255u) && nx <= wlim;
    const uint32_t link = form ? nx - a0 : 0x8000u | lane;
    uint64_t starts = 0;
    uint32_t r = 0;
    do {
      starts |= 1ull << r;
      r = rdlane(link, r);
    } while (r < kWave);
    if (r & 0x8000u) {
      r &= 0x7fffu;
      starts &= ~(1ull << r);
    }
    if (starts) {
      const bool mine = (starts >> lane) & 1ull;
      const uint32_t sz = mine ? ll + ml : 0u;
      const uint32_t incl = wave_incl_scan(sz);
      const uint32_t total = rdlane(incl, kWave - 1u);
      const uint32_t o_lit = op + incl - sz, o_mat = o_lit + ll;
      // lz4_block_decoder.dart:96-118 with a writer limited to prefix + blockMaxSize; byte_writer.dart:97-99
      const bool bad = mine && (dist == 0u || dist > o_mat + back);
      if (ballot(bad) || total > cap - op) {
        unfit = true;
        break;
      }
      if (PLAN) {
        for (uint64_t m = starts; m; m &= m - 1ull) {
          const uint32_t j = (uint32_t)__builtin_ctzll(m);
          const uint32_t ls_j = rdlane(ls, j), ll_j = rdlane(ll, j), ol_j = rdlane(o_lit, j);
          const uint32_t ml_j = rdlane(ml, j), dist_j = rdlane(dist, j);
          for (uint32_t i = lane; i < ll_j; i += kWave) out[ol_j + i] = cw[ls_j + i];
          fill_words(D + ol_j + ll_j, ml_j, dist_j, lane);
        }
      }
      op += total;
      ip += r;
      continue;
    }
    // ---- the token at ip is not of that form, or does not lie inside the window with all it leads to
    if (a0 != 0u && wb + wlim < n_in) {  // a window loaded at the token may hold it
      have = false;
      continue;
    }
    // one sequence of any form, straight from memory (the last sequence of every block comes here)
    uint32_t pos = ip;
    const uint32_t token = in[pos++];
    uint64_t len = token >> 4;
    if (len == 15u) {
      uint32_t x;
      do {
        if (pos >= n_in) { unfit = true; break; }
        x = in[pos++];
        len += x;
      } while (x == 255u);
      if (unfit) break;
    }
    if (len > cap - op || len > n_in - pos) { unfit = true; break; }
    if (PLAN && len) {
      if (len >= kCopyLong) wave_copy_long(out + op, in + pos, (uint32_t)len, lane);
      else wave_copy(out + op, in + pos, (uint32_t)len, lane);
      vm_quiesce();
    }
    op += (uint32_t)len;
    pos += (uint32_t)len;
    if (pos == n_in) { ip = pos; break; }  // the block ends behind a literal run
    if (n_in - pos < 2u) { unfit = true; break; }
    const uint32_t d = (uint32_t)in[pos] | ((uint32_t)in[pos + 1] << 8);
    pos += 2;
    uint64_t mlen = (token & 15u) + 4u;
    if ((token & 15u) == 15u) {
      uint32_t x;
      do {
        if (pos >= n_in) { unfit = true; break; }
        x = in[pos++];
        mlen += x;
      } while (x == 255u);
      if (unfit) break;
    }
    if (d == 0u || d > op + back || mlen > cap - op) { unfit = true; break; }
    if (PLAN) fill_words(D + op, (uint32_t)mlen, d, lane);
    op += (uint32_t)mlen;
    ip = pos;
    have = have && ip - wb < wlim;
  }
  if (lane == 0) {
    a.out_len[b] = op;
    a.status[b] = unfit ? 1 : 0;
  }
}

// One round of pointer doubling over the chunk's distance array.  A word may be read while its owner replaces
// it: both values name a byte that equals this one, so any interleaving is right.  A word that is zero stays
// zero, so a tile in which no word changed has only final distances and is skipped from then on.
constexpr uint32_t kJumpTile = 4096;  // words per workgroup step
__global__ __launch_bounds__(256) void link_jump_kernel(uint32_t *D, uint64_t n, uint8_t *done, uint32_t *moved) {
  const uint64_t tiles = (n + kJumpTile - 1) / kJumpTile;
  for (uint64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    if (done[t]) continue;
    bool any = false;
#pragma unroll
    for (uint32_t s = 0; s < 4; s++) {
      const uint64_t i0 = t * kJumpTile + s * 1024u + 4u * threadIdx.x;
      if (i0 >= n) continue;
      u32x4 v = *(const u32x4 *)(D + i0);  // the array is padded to whole tiles
      bool ch = false;
#pragma unroll
      for (uint32_t e = 0; e < 4; e++) {
        const uint64_t i = i0 + e;
        const uint32_t d = v[e];
        if (d != 0u && d <= i && i < n) {
          const uint32_t dq = D[i - d];
          if (dq) {
            v[e] = d + dq;
            ch = true;
          }
        }
      }
      if (ch) *(u32x4 *)(D + i0) = v;
      any |= ch;
    }
    const int some = __syncthreads_or(any ? 1 : 0);
    if (threadIdx.x == 0) {
      if (some) *moved = 1u;
      else done[t] = 1;
    }
  }
}

// out[p] = out[p - D[p]] for the chunk's copied bytes; a source in front of the frame is a dictionary byte
__global__ __launch_bounds__(256) void link_gather_kernel(const uint32_t *D, uint64_t n, uint8_t *out, uint64_t chunk_start,
                                                           const uint8_t *dict, uint32_t dict_len) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t d = D[i];
    if (d == 0u) continue;
    const uint64_t p = chunk_start + i;
    out[p] = d <= p ? out[p - d] : dict[dict_len - (uint32_t)(d - p)];
  }
}

void launch_link_walk(const LinkArgs &a, bool plan, hipStream_t s) {
  if (!a.nblocks) return;
  const dim3 grid((a.nblocks + 3u) / 4u), block(256);
  if (plan) hipLaunchKernelGGL(link_walk_kernel<true>, grid, block, 0, s, a);
  else hipLaunchKernelGGL(link_walk_kernel<false>, grid, block, 0, s, a);
}

void launch_link_jump(uint32_t *D, uint64_t n, uint8_t *done, uint32_t *moved, hipStream_t s) {
  if (!n) return;
  const uint64_t tiles = (n + kJumpTile - 1) / kJumpTile;
  const uint32_t grid = tiles < 16384u ? (uint32_t)tiles : 16384u;
  hipLaunchKernelGGL(link_jump_kernel, dim3(grid), dim3(256), 0, s, D, n, done, moved);
}

void launch_link_gather(const uint32_t *D, uint64_t n, uint8_t *out, uint64_t chunk_start, const uint8_t *dict,
                        uint32_t dict_len, hipStream_t s) {
  if (!n) return;
  const uint64_t want = (n + 255u) / 256u;
  const uint32_t grid = want < 65536u ? (uint32_t)want : 65536u;
  hipLaunchKernelGGL(link_gather_kernel, dim3(grid), dim3(256), 0, s, D, n, out, chunk_start, dict, dict_len);
}

uint64_t link_dist_words(uint64_t n) { return (n + kJumpTile - 1) / kJumpTile * kJumpTile; }
uint64_t link_tiles(uint64_t n) { return (n + kJumpTile - 1) / kJumpTile; }

}  // namespace lz4mi
