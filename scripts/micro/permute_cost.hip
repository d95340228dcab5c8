// Cycles per ds_bpermute_b32 / ds_permute_b32: one dependent chain, and four chains side by side.
// build: hipcc --offload-arch=gfx950 -O2 scripts/micro/permute_cost.hip -o scripts/micro/permute_cost.bin
#include <hip/hip_runtime.h>
#include <cstdio>
template <int W, bool FWD>
__global__ void k(uint32_t *out, unsigned long long *cyc, uint32_t seed) {
  const uint32_t lane = threadIdx.x;
  uint32_t x[W];
  for (int w = 0; w < W; w++) x[w] = (lane * 7u + seed + w) & 63u;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 1000; it++) {
#pragma unroll
    for (int w = 0; w < W; w++)
      x[w] = FWD ? (uint32_t)__builtin_amdgcn_ds_permute((int)(4u * x[w]), (int)x[w]) & 63u
                 : (uint32_t)__builtin_amdgcn_ds_bpermute((int)(4u * x[w]), (int)x[w]);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  uint32_t s = 0;
  for (int w = 0; w < W; w++) s += x[w];
  out[lane] = s;
  if (lane == 0) cyc[0] = t1 - t0;
}
template <int W, bool FWD>
static void run(const char *name) {
  uint32_t *d; unsigned long long *c, h;
  hipMalloc(&d, 256); hipMalloc(&c, 8);
  hipLaunchKernelGGL((k<W, FWD>), dim3(1), dim3(64), 0, 0, d, c, 3u);
  hipLaunchKernelGGL((k<W, FWD>), dim3(1), dim3(64), 0, 0, d, c, 5u);
  hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("%s: %d chain(s): %.1f cycles per round, %.1f per operation\n", name, W, h / 1000.0, h / 1000.0 / W);
  hipFree(d); hipFree(c);
}
int main() {
  run<1, false>("ds_bpermute"); run<4, false>("ds_bpermute"); run<8, false>("ds_bpermute");
  run<1, true>("ds_permute"); run<4, true>("ds_permute"); run<8, true>("ds_permute");
  return 0;
}
