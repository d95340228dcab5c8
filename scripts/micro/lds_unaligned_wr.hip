// Microbenchmark: throughput of LDS writes by width and alignment, 1 and 16 wavefronts per CU
// (per-lane addresses a byte apart, like a literal run copied lane-parallel lands in the staging buffer).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
struct __attribute__((packed)) P32 { uint32_t v; };
struct __attribute__((packed)) P64 { uint64_t v; };
typedef __attribute__((address_space(3))) P32 lds_p32;
typedef __attribute__((address_space(3))) P64 lds_p64;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x2 lds_u64a;

// MODE 0: ds_write_b64 at base + 9 * lane (unaligned)   MODE 1: ds_write_b64 at base + 8 * lane (aligned)
// MODE 2: ds_write_b32 at base + 5 * lane (unaligned)   MODE 3: ds_write_b32 at base + 4 * lane (aligned)
// MODE 4: ds_write_b8 at base + lane                    MODE 5: 8 x ds_write_b8 (eight bytes at base + 9 * lane)
template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned long long *cycles, uint32_t *sink, uint32_t iters) {
  __shared__ __attribute__((aligned(16))) uint8_t s[32 * 1024];
  lds_u8 *base = (lds_u8 *)s + 2048 * (threadIdx.x / 64);
  const uint32_t lane = threadIdx.x & 63;
  uint32_t acc = lane;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
    for (uint32_t u = 0; u < 8; u++) {
      const uint32_t r = ((it * 8 + u) * 72u & 511u);
      acc = acc * 5u + 1u;
      if (MODE == 0) ((lds_p64 *)(base + r + 9 * lane + 1))->v = ((uint64_t)acc << 32) | acc;
      else if (MODE == 1) *(lds_u64a *)(base + (r & ~7u) + 8 * lane) = u32x2{acc, acc};
      else if (MODE == 2) ((lds_p32 *)(base + r + 5 * lane + 1))->v = acc;
      else if (MODE == 3) *(lds_u32 *)(base + (r & ~3u) + 4 * lane) = acc;
      else if (MODE == 4) base[r + lane] = (uint8_t)acc;
      else {
#pragma unroll
        for (uint32_t b = 0; b < 8; b++) base[r + 9 * lane + 1 + b] = (uint8_t)(acc >> (4 * b));
      }
    }
  }
  __syncthreads();
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[0] = t1 - t0;
  sink[threadIdx.x] = acc + s[threadIdx.x];
}
template <int MODE> void run(const char *nm, uint32_t waves) {
  unsigned long long *d_c; uint32_t *d_s;
  (void)hipMalloc(&d_c, 8); (void)hipMalloc(&d_s, 4096);
  for (int r = 0; r < 2; r++) hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(64 * waves), 0, 0, d_c, d_s, 2000u);
  (void)hipDeviceSynchronize();
  unsigned long long c; (void)hipMemcpy(&c, d_c, 8, hipMemcpyDeviceToHost);
  printf("%-44s %2u waves: %6.1f cycles per write per wave\n", nm, waves, c / 16000.0);
  (void)hipFree(d_c); (void)hipFree(d_s);
}
int main() {
  for (uint32_t w : {1u, 16u}) {
    run<1>("ds_write_b64 aligned", w);
    run<0>("ds_write_b64 unaligned (9 * lane + 1)", w);
    run<3>("ds_write_b32 aligned", w);
    run<2>("ds_write_b32 unaligned (5 * lane + 1)", w);
    run<4>("ds_write_b8 (lane)", w);
    run<5>("8 x ds_write_b8 (9 * lane + 1 ..)", w);
  }
  return 0;
}
