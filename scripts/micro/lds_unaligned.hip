// Microbenchmark: dependent-chain latency of LDS reads by width and alignment (one wavefront).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) uint8_t lds_u8;
struct __attribute__((packed)) P32 { uint32_t v; };
struct __attribute__((packed)) P64 { uint64_t v; };
template <int MODE>
__global__ void k(unsigned long long *cycles, uint32_t *sink, uint32_t iters, uint32_t misalign) {
  __shared__ __attribute__((aligned(16))) uint8_t s[8192 + 16];
  for (uint32_t i = threadIdx.x; i < 8192 + 16; i += 64) s[i] = (uint8_t)(i * 37u >> 3);
  __syncthreads();
  lds_u8 *base = (lds_u8 *)s;
  uint32_t pos = threadIdx.x * 113u;
  uint32_t acc = 0;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (uint32_t it = 0; it < iters; it++) {
    uint32_t a = ((pos & 8191u) & ~7u) + misalign;
    uint32_t v;
    if (MODE == 0) v = base[a];
    else if (MODE == 1) v = ((const __attribute__((address_space(3))) P32 *)(base + a))->v;
    else { uint64_t q = ((const __attribute__((address_space(3))) P64 *)(base + a))->v; v = (uint32_t)q ^ (uint32_t)(q >> 32); }
    acc += v;
    pos = pos * 5u + v + 1;   // dependent
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { cycles[0] = t1 - t0; }
  sink[threadIdx.x] = acc;
}
template <int MODE> void run(const char *nm, uint32_t mis) {
  unsigned long long *d_c; uint32_t *d_s;
  (void)hipMalloc(&d_c, 8); (void)hipMalloc(&d_s, 256);
  for (int r = 0; r < 2; r++) hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(64), 0, 0, d_c, d_s, 4000u, mis);
  (void)hipDeviceSynchronize();
  unsigned long long c; (void)hipMemcpy(&c, d_c, 8, hipMemcpyDeviceToHost);
  printf("%-14s misalign %u: %6.1f cycles per dependent read (incl. ~6 ALU)\n", nm, mis, c / 4000.0);
  (void)hipFree(d_c); (void)hipFree(d_s);
}
int main() {
  for (uint32_t mis : {0u, 1u, 3u, 5u}) { run<0>("ds_read_u8", mis); run<1>("ds_read_b32", mis); run<2>("ds_read_b64", mis); }
  return 0;
}
