// What a lane that no lane sends to receives from ds_permute_b32 (forward permute): expected 0.
// build: hipcc --offload-arch=gfx950 -O2 scripts/micro/permute_unwritten.hip -o /tmp/permute_unwritten
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(uint32_t *out) {
  const uint32_t lane = threadIdx.x;
  const bool push = (lane % 2 == 0) && lane < 32;
  // senders name lane + 7, everybody else lane 0; every lane sends 0xABCD0000 | lane
  const uint32_t got = (uint32_t)__builtin_amdgcn_ds_permute((int)(push ? 4u * (lane + 7u) : 0u), (int)(0xABCD0000u | lane));
  out[lane] = got;
}
int main() {
  uint32_t *d, h[64];
  if (hipMalloc(&d, sizeof h) != hipSuccess) return 1;
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  int bad = 0;
  for (int l = 0; l < 64; l++) {
    const bool target = l >= 7 && (l - 7) % 2 == 0 && l - 7 < 32;
    const uint32_t want = target ? 0xABCD0000u | (uint32_t)(l - 7) : 0u;
    if (l != 0 && h[l] != want) bad++;
    printf("%08x%c", h[l], l % 8 == 7 ? '\n' : ' ');
  }
  printf("lanes that differ from the expectation (lane 0 aside): %d\n", bad);
  return bad != 0;
}
