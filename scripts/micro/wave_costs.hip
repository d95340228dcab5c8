// Cycle costs of the building blocks of the block decoder on one wavefront (diagnostic, not part of the library).
// Build: hipcc --offload-arch=gfx950 -O3 -o wave_costs wave_costs.hip ; run: ./wave_costs [waves_per_cu]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((address_space(3))) uint8_t lds_u8;
#define N 1024

__global__ __launch_bounds__(64) void k(uint64_t *out, uint32_t *sink, uint32_t seed) {
  __shared__ __attribute__((aligned(16))) uint8_t buf[8192];
  lds_u8 *s = (lds_u8 *)buf;
  const uint32_t lane = threadIdx.x;
  for (uint32_t i = lane; i < 8192; i += 64) buf[i] = (uint8_t)(i * 7 + seed);
  __syncthreads();
  uint64_t t[8];
  uint32_t acc = 0;
  // (0) dependent ds_read_u8 -> ds_write_b8 chain (each read depends on the previous write through the address)
  t[0] = __builtin_readcyclecounter();
  {
    uint32_t a = lane;
    for (int i = 0; i < N; i++) {
      uint32_t v = s[a];
      s[(a + 1000) & 8191] = (uint8_t)v;
      a = (a + 64 + (v & 1)) & 8191;
    }
    acc += a;
  }
  t[1] = __builtin_readcyclecounter();
  // (1) independent read then write, no data-dependent address
  {
    uint32_t a = lane;
    for (int i = 0; i < N; i++) {
      uint32_t v = s[a];
      s[(a + 1000) & 8191] = (uint8_t)v;
      a = (a + 64) & 8191;
    }
    acc += a;
  }
  t[2] = __builtin_readcyclecounter();
  // (2) readlane hop chain: r = readlane(nx, r)
  {
    uint32_t nx = (lane * 5 + 3 + seed) & 63;
    uint32_t r = 0, tot = 0;
    for (int i = 0; i < N; i++) {
      tot += __builtin_amdgcn_readlane(nx, r) ;
      r = __builtin_amdgcn_readlane(nx ^ 1, r) & 63;
    }
    acc += r + tot;
  }
  t[3] = __builtin_readcyclecounter();
  // (3) 16 dependent VALU adds per iteration
  {
    uint32_t x = lane + seed;
    for (int i = 0; i < N; i++) {
#pragma unroll
      for (int u = 0; u < 16; u++) x = x * 3 + u;
    }
    acc += x;
  }
  t[4] = __builtin_readcyclecounter();
  // (4) dword LDS: unaligned ds_read_b32 -> ds_write_b32
  {
    uint32_t a = lane * 4 + 1;
    for (int i = 0; i < N; i++) {
      uint32_t v;
      __builtin_memcpy(&v, (const void *)(buf + a), 4);
      __builtin_memcpy((void *)(buf + ((a + 1000) & 8191 & ~3u) + 1), &v, 4);
      a = (a + 256) & 4095;
    }
    acc += a;
  }
  t[5] = __builtin_readcyclecounter();
  // (5) scalar branchy loop: data-dependent taken branches
  {
    uint32_t x = __builtin_amdgcn_readfirstlane(seed), c = 0;
    for (int i = 0; i < N; i++) {
      if (x & 1) c += 3; else c ^= x;
      if (x & 2) c += 5; else c ^= (x >> 1);
      x = x * 1664525u + 1013904223u;
      x = __builtin_amdgcn_readfirstlane(x);
    }
    acc += c;
  }
  t[6] = __builtin_readcyclecounter();
  if (lane == 0 && blockIdx.x == 0)
    for (int i = 0; i < 6; i++) out[i] = t[i + 1] - t[i];
  sink[blockIdx.x * 64 + lane] = acc;
}

int main(int argc, char **argv) {
  int wpc = argc > 1 ? atoi(argv[1]) : 1;
  uint64_t *d_out; uint32_t *d_sink;
  hipMalloc(&d_out, 64); hipMalloc(&d_sink, 256 * 64 * 4 * 32);
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k, dim3(256 * wpc), dim3(64), 0, 0, d_out, d_sink, 12345u);
    hipDeviceSynchronize();
  }
  uint64_t h[8];
  hipMemcpy(h, d_out, 48, hipMemcpyDeviceToHost);
  const char *nm[6] = {"dependent ds_read_u8->ds_write_b8 (addr dep)", "ds_read_u8->ds_write_b8", "2 readlane hop", "16 dependent VALU",
                       "ds_read_b32->ds_write_b32 unaligned", "scalar branchy iteration"};
  for (int i = 0; i < 6; i++) printf("%-50s %8.1f cycles/iter (%d waves/CU)\n", nm[i], (double)h[i] / N, wpc);
  return 0;
}
