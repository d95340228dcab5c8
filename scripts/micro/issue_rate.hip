// Issue cost of dependent / independent scalar and vector instructions and of taken branches, one wavefront (diagnostic).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define R64(x) R16(x) R16(x) R16(x) R16(x)
__global__ __launch_bounds__(64) void k(uint64_t *out, uint32_t *sink) {
  uint64_t t[10];
  uint32_t s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3, v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3;
  t[0] = __builtin_readcyclecounter();
  for (int i = 0; i < 16; i++) asm volatile(R64("s_add_u32 %0, %0, 1\n") : "+s"(s0));
  t[1] = __builtin_readcyclecounter();
  for (int i = 0; i < 16; i++) asm volatile(R16("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n") : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3));
  t[2] = __builtin_readcyclecounter();
  for (int i = 0; i < 16; i++) asm volatile(R64("v_add_u32 %0, %0, 1\n") : "+v"(v0));
  t[3] = __builtin_readcyclecounter();
  for (int i = 0; i < 16; i++) asm volatile(R16("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
  t[4] = __builtin_readcyclecounter();
  // s_cmp + taken branch pairs
  for (int i = 0; i < 16; i++) asm volatile(R64("s_cmp_eq_u32 %0, %0\n s_cbranch_scc1 1f\n s_nop 0\n 1:\n") : "+s"(s0));
  t[5] = __builtin_readcyclecounter();
  // s_cmp + not-taken branch pairs
  for (int i = 0; i < 16; i++) asm volatile(R64("s_cmp_lg_u32 %0, %0\n s_cbranch_scc1 1f\n 1:\n") : "+s"(s0));
  t[6] = __builtin_readcyclecounter();
  // alternating scalar / vector dependent through v_readfirstlane
  for (int i = 0; i < 16; i++) asm volatile(R64("v_add_u32 %0, %1, %0\n s_nop 0\n v_readfirstlane_b32 %1, %0\n") : "+v"(v0), "+s"(s0));
  t[7] = __builtin_readcyclecounter();
  // v_cmp -> vcc -> v_cndmask dependent pairs
  for (int i = 0; i < 16; i++) asm volatile(R64("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n") : "+v"(v0), "+v"(v1) : : "vcc");
  t[8] = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) for (int i = 0; i < 8; i++) out[i] = t[i + 1] - t[i];
  sink[blockIdx.x * 64 + threadIdx.x] = s0 + s1 + s2 + s3 + v0 + v1 + v2 + v3;
}
int main(int argc, char **argv) {
  int wpc = argc > 1 ? atoi(argv[1]) : 1;
  uint64_t *d_out; uint32_t *d_sink;
  (void)hipMalloc(&d_out, 128); (void)hipMalloc(&d_sink, 256 * 64 * 4 * 32);
  for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k, dim3(256 * wpc), dim3(64), 0, 0, d_out, d_sink);
  (void)hipDeviceSynchronize();
  uint64_t h[8]; (void)hipMemcpy(h, d_out, 64, hipMemcpyDeviceToHost);
  const char *nm[8] = {"s_add dependent", "s_add 4 independent chains (per instr)", "v_add dependent", "v_add 4 independent chains (per instr)",
                       "s_cmp + taken branch (pair)", "s_cmp + not-taken branch (pair)", "v_add + s_nop + v_readfirstlane (triple)", "v_cmp + v_cndmask (pair)"};
  const double div[8] = {1024, 1024, 1024, 1024, 1024, 1024, 1024, 1024};
  for (int i = 0; i < 8; i++) printf("%-44s %6.2f cycles (%d waves/CU)\n", nm[i], h[i] / div[i], wpc);
  return 0;
}
