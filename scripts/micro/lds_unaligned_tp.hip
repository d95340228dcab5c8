// Microbenchmark: throughput of LDS reads by width and alignment, 1 and 16 wavefronts per CU
// (independent reads, 8 in flight per lane; addresses stride one byte per lane like the kernels' byte-granular reads).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
struct __attribute__((packed)) P32 { uint32_t v; };
struct __attribute__((packed)) P64 { uint64_t v; };
typedef const __attribute__((address_space(3))) P32 lds_p32;
typedef const __attribute__((address_space(3))) P64 lds_p64;

// MODE 0: ds_read_b32 at byte address base + lane (unaligned for 3 of 4 lanes)
// MODE 1: two aligned dwords (ds_read2_b32) + v_alignbyte: the same four bytes
// MODE 2: ds_read_b32 at base + 4 * lane (aligned, for reference)
// MODE 3: ds_read_b64 at base + lane (unaligned)
// MODE 4: three aligned dwords + two v_alignbyte: the same eight bytes
// MODE 5: ds_read_u8 at base + lane
template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned long long *cycles, uint32_t *sink, uint32_t iters) {
  __shared__ __attribute__((aligned(16))) uint8_t s[16 * 1024];
  for (uint32_t i = threadIdx.x; i < 16 * 1024; i += blockDim.x) s[i] = (uint8_t)(i * 37u >> 3);
  __syncthreads();
  lds_u8 *base = (lds_u8 *)s + 1024 * (threadIdx.x / 64);
  const uint32_t lane = threadIdx.x & 63;
  uint32_t acc = 0;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
    for (uint32_t u = 0; u < 8; u++) {
      const uint32_t a = ((it * 8 + u) * 72u & 511u) + lane;
      if (MODE == 0) acc += ((lds_p32 *)(base + a))->v;
      else if (MODE == 1) {
        const uint32_t al = a & ~3u;
        const uint32_t lo = *(const lds_u32 *)(base + al), hi = *(const lds_u32 *)(base + al + 4);
        acc += __builtin_amdgcn_alignbyte(hi, lo, a & 3u);
      } else if (MODE == 2) acc += *(const lds_u32 *)(base + ((a - lane) & ~3u) + 4 * lane);
      else if (MODE == 3) { const uint64_t q = ((lds_p64 *)(base + a))->v; acc += (uint32_t)q ^ (uint32_t)(q >> 32); }
      else if (MODE == 4) {
        const uint32_t al = a & ~3u;
        const uint32_t w0 = *(const lds_u32 *)(base + al), w1 = *(const lds_u32 *)(base + al + 4), w2 = *(const lds_u32 *)(base + al + 8);
        acc += __builtin_amdgcn_alignbyte(w1, w0, a & 3u) ^ __builtin_amdgcn_alignbyte(w2, w1, a & 3u);
      } else acc += base[a];
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[0] = t1 - t0;
  sink[threadIdx.x] = acc;
}
template <int MODE> void run(const char *nm, uint32_t waves) {
  unsigned long long *d_c; uint32_t *d_s;
  (void)hipMalloc(&d_c, 8); (void)hipMalloc(&d_s, 4096);
  for (int r = 0; r < 2; r++) hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(64 * waves), 0, 0, d_c, d_s, 2000u);
  (void)hipDeviceSynchronize();
  unsigned long long c; (void)hipMemcpy(&c, d_c, 8, hipMemcpyDeviceToHost);
  printf("%-44s %2u waves: %6.1f cycles per read per wave\n", nm, waves, c / 16000.0);
  (void)hipFree(d_c); (void)hipFree(d_s);
}
int main() {
  for (uint32_t w : {1u, 16u}) {
    run<2>("ds_read_b32 aligned (4 * lane)", w);
    run<0>("ds_read_b32 at base + lane (unaligned)", w);
    run<1>("ds_read2_b32 aligned + v_alignbyte", w);
    run<3>("ds_read_b64 at base + lane (unaligned)", w);
    run<4>("3 aligned dwords + 2 v_alignbyte", w);
    run<5>("ds_read_u8 at base + lane", w);
  }
  return 0;
}
