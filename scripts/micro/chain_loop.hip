// Variants of the decoder's scalar token chain on a synthetic compressed window (diagnostic).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint16_t lds_u16;
struct __attribute__((packed)) DP32 { uint32_t v; };
struct __attribute__((packed)) DP64 { uint64_t v; };
typedef __attribute__((address_space(3))) DP32 lds_dp32;
typedef __attribute__((address_space(3))) DP64 lds_dp64;
__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return __builtin_amdgcn_readlane(v, l); }
#define REP 64
constexpr uint32_t kCw = 2048;

template <int V>
__global__ __launch_bounds__(64) void k(uint64_t *out, uint32_t *sink, const uint8_t *win, uint32_t wlim) {
  __shared__ __attribute__((aligned(16))) uint8_t s_cw[kCw + 16];
  __shared__ uint16_t s_tl[130];
  lds_u8 *cw = (lds_u8 *)s_cw;
  lds_u16 *tl = (lds_u16 *)s_tl;
  const uint32_t lane = threadIdx.x;
  for (uint32_t i = lane; i < kCw + 16; i += 64) s_cw[i] = win[i];
  __syncthreads();
  uint32_t acc = 0, ntok = 0;
  uint64_t t0 = __builtin_readcyclecounter();
  for (int rep = 0; rep < REP; rep++) {
    uint32_t a = 0, count = 0, tot = 0;
    if (V == 0) {  // LDS dword read per token, LDS list
      while (count < 64 && tot <= 1u << 20) {
        if (a + 4u > wlim) break;
        const uint32_t lo = uni(((const lds_dp32 *)(cw + a))->v);
        const uint32_t ll0 = (lo >> 4) & 15u, e1 = (lo >> 8) & 255u;
        const bool isl = ll0 == 15u;
        if (isl && e1 == 255u) break;
        const uint32_t ll = ll0 + (isl ? e1 : 0u);
        const uint32_t nx = a + 3u + (isl ? 1u : 0u) + ll + ((lo & 15u) == 15u ? 1u : 0u);
        if (nx > wlim) break;
        tl[count] = (uint16_t)a;
        count++;
        tot += ll + 64u;
        a = nx;
      }
      acc += a;
    } else if (V == 1) {  // same, token positions kept in a register (lane j: token j)
      uint32_t tp = 0;
      while (count < 64 && tot <= 1u << 20) {
        if (a + 4u > wlim) break;
        const uint32_t lo = uni(((const lds_dp32 *)(cw + a))->v);
        const uint32_t ll0 = (lo >> 4) & 15u, e1 = (lo >> 8) & 255u;
        const bool isl = ll0 == 15u;
        if (isl && e1 == 255u) break;
        const uint32_t ll = ll0 + (isl ? e1 : 0u);
        const uint32_t nx = a + 3u + (isl ? 1u : 0u) + ll + ((lo & 15u) == 15u ? 1u : 0u);
        if (nx > wlim) break;
        tp = lane == count ? a : tp;
        count++;
        tot += ll + 64u;
        a = nx;
      }
      acc += a + tp;
    } else if (V == 2) {  // window held in registers (dword per lane per 256 bytes), token fetched with two readlanes
      uint32_t tp = 0;
      uint32_t hot = ((const lds_dp32 *)(cw + 4u * lane))->v;  // bytes [hb, hb + 256)
      uint32_t hb = 0;
      while (count < 64 && tot <= 1u << 20) {
        if (a + 4u > wlim) break;
        if (a - hb > 248u) {  // token dword (and the next) not inside the hot register
          hb = a & ~3u;
          hot = ((const lds_dp32 *)(cw + hb + 4u * lane))->v;
        }
        const uint32_t i = (a - hb) >> 2;
        const uint64_t two = ((uint64_t)rdlane(hot, i + 1u) << 32) | rdlane(hot, i);
        const uint32_t lo = (uint32_t)(two >> (8u * (a & 3u)));
        const uint32_t ll0 = (lo >> 4) & 15u, e1 = (lo >> 8) & 255u;
        const bool isl = ll0 == 15u;
        if (isl && e1 == 255u) break;
        const uint32_t ll = ll0 + (isl ? e1 : 0u);
        const uint32_t nx = a + 3u + (isl ? 1u : 0u) + ll + ((lo & 15u) == 15u ? 1u : 0u);
        if (nx > wlim) break;
        tp = lane == count ? a : tp;
        count++;
        tot += ll + 64u;
        a = nx;
      }
      acc += a + tp;
    }
    ntok = count;
  }
  uint64_t t1 = __builtin_readcyclecounter();
  if (lane == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = ntok; }
  sink[blockIdx.x * 64 + lane] = acc + s_tl[lane];
}

int main(int argc, char **argv) {
  int wpc = argc > 1 ? atoi(argv[1]) : 1;
  int gap = argc > 2 ? atoi(argv[2]) : 60;   // mean literal length
  uint64_t *d_out; uint32_t *d_sink; uint8_t *d_win;
  (void)hipMalloc(&d_out, 64); (void)hipMalloc(&d_sink, 256 * 64 * 4 * 32); (void)hipMalloc(&d_win, kCw + 16);
  uint8_t h[kCw + 16];
  uint32_t x = 777, p = 0, n = 0;
  for (uint32_t i = 0; i < kCw + 16; i++) { x = x * 1664525u + 1013904223u; h[i] = (uint8_t)(x >> 24); if (h[i] == 255) h[i] = 1; }
  while (p + 300 < kCw) {
    x = x * 1664525u + 1013904223u;
    uint32_t ll = (x >> 8) % (2 * gap);
    x = x * 1664525u + 1013904223u;
    uint32_t m = 4 + (x >> 8) % 120;
    uint32_t tok = ((ll < 15 ? ll : 15) << 4) | (m - 4 < 15 ? m - 4 : 15);
    h[p++] = (uint8_t)tok;
    if (ll >= 15) h[p++] = (uint8_t)(ll - 15);
    p += ll;
    h[p++] = 7; h[p++] = 0;
    if (m - 4 >= 15) h[p++] = (uint8_t)(m - 19);
    n++;
  }
  (void)hipMemcpy(d_win, h, kCw + 16, hipMemcpyHostToDevice);
  printf("%u tokens in %u bytes, %d waves/CU\n", n, p, wpc);
#define RUN(V, name)                                                                          \
  for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k<V>), dim3(256 * wpc), dim3(64), 0, 0, d_out, d_sink, d_win, p); \
  (void)hipDeviceSynchronize();                                                               \
  { uint64_t r[2]; (void)hipMemcpy(r, d_out, 16, hipMemcpyDeviceToHost); printf("%-44s %7.1f cycles/token (%llu tokens)\n", name, (double)r[0] / REP / r[1], (unsigned long long)r[1]); }
  RUN(0, "LDS dword per token, LDS list");
  RUN(1, "LDS dword per token, register list");
  RUN(2, "window in a register, two readlanes per token");
  return 0;
}
