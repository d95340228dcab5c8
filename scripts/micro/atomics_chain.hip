// Microbenchmark: cost of "A scattered no-return atomics, then a dependent 8-lane gather" per iteration,
// one wavefront per workgroup, table of 64K dwords per wavefront (the encoder's access pattern).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

template <int A, int MODE>  // MODE 0: atomic umax, 1: plain store, 2: nothing
__global__ void chain(uint32_t *tables, unsigned long long *cycles, uint32_t iters, uint32_t *sink) {
  uint32_t *T = tables + (size_t)blockIdx.x * 65536;
  uint32_t lane = threadIdx.x;
  uint32_t x = lane * 2654435761u + blockIdx.x * 40503u + 1;
  uint32_t acc = 0;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
    for (int a = 0; a < A; a++) {
      x = x * 1664525u + 1013904223u;
      uint32_t h = (x >> 16);
      if (MODE == 0) (void)__hip_atomic_fetch_max(T + h, it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (MODE == 1) __hip_atomic_store(T + h, it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    x = x * 1664525u + 1013904223u + acc;
    uint32_t h = (x >> 16);
    uint32_t v = 0;
    if (lane < 8) v = __hip_atomic_load(T + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    acc += __builtin_amdgcn_readfirstlane(v) & 1;  // dependent chain
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) { cycles[blockIdx.x] = t1 - t0; sink[blockIdx.x] = acc; }
}

template <int A, int MODE>
void run(const char *name, uint32_t *tables, int nwg, uint32_t iters) {
  unsigned long long *d_c; uint32_t *d_s;
  hipMalloc(&d_c, nwg * 8); hipMalloc(&d_s, nwg * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((chain<A, MODE>), dim3(nwg), dim3(64), 0, 0, tables, d_c, iters, d_s);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long *c = (unsigned long long *)malloc(nwg * 8);
  hipMemcpy(c, d_c, nwg * 8, hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < nwg; i++) s += c[i];
  printf("%-28s nwg %5d  A=%d  %8.1f cycles/iter  %8.1f ns/iter\n", name, nwg, A, s / nwg / iters, ms * 1e6 / iters);
  free(c); hipFree(d_c); hipFree(d_s);
}

int main() {
  const int maxwg = 4096;
  uint32_t *tables; hipMalloc(&tables, (size_t)maxwg * 65536 * 4);
  hipMemset(tables, 0, (size_t)maxwg * 65536 * 4);
  const uint32_t iters = 2000;
  for (int nwg : {64, 1024, 4096}) {
    run<0, 2>("gather only", tables, nwg, iters);
    run<1, 0>("1 atomic + gather", tables, nwg, iters);
    run<3, 0>("3 atomics + gather", tables, nwg, iters);
    run<1, 1>("1 store + gather", tables, nwg, iters);
    run<3, 1>("3 stores + gather", tables, nwg, iters);
  }
  return 0;
}
