// Calibration of FETCH_SIZE / WRITE_SIZE for the encoder's access pattern: random dword gathers and random dword
// stores at workgroup scope over a table far larger than the caches, next to a wide streaming read of known size.
// Run under `rocprofv3 --pmc FETCH_SIZE` (and WRITE_SIZE, and the raw TCC_EA0 counters) and divide the per-kernel
// counter by the number of accesses printed here.
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_fetch gather_fetch.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

constexpr int K = 64;  // accesses per thread

__global__ __launch_bounds__(256) void gather_kernel(const uint32_t *table, uint32_t mask, uint32_t *out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (int k = 0; k < K; k += 4) {  // four independent gathers in flight per lane
    uint32_t v[4];
#pragma unroll
    for (int u = 0; u < 4; u++)
      v[u] = __hip_atomic_load(table + (mix(t * K + k + u) & mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
    for (int u = 0; u < 4; u++) acc += v[u];
  }
  out[t] = acc;
}

__global__ __launch_bounds__(256) void scatter_kernel(uint32_t *table, uint32_t mask) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  for (int k = 0; k < K; k++)
    __hip_atomic_store(table + (mix(t * K + k) & mask), t + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__global__ __launch_bounds__(256) void stream_kernel(const uint4 *src, size_t n16, uint32_t *out) {
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = src[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char **argv) {
  const size_t bytes = (argc > 1 ? (size_t)atoi(argv[1]) : 2048) << 20;  // table size in MiB (a power of two)
  const uint32_t mask = (uint32_t)(bytes / 4 - 1);
  uint32_t *table, *out;
  if (hipMalloc(&table, bytes) != hipSuccess || hipMalloc(&out, 4u << 20) != hipSuccess) return 1;
  hipMemset(table, 0, bytes);
  const int grid = 4096, block = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  printf("table %zu MiB\n", bytes >> 20);
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0); hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(block), 0, 0, table, mask, out); hipEventRecord(e1);
    hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("gather: %zu dword gathers in %.3f ms (%.2f G/s)\n", (size_t)grid * block * K, ms, grid * (double)block * K / ms / 1e6);
    hipEventRecord(e0); hipLaunchKernelGGL(scatter_kernel, dim3(grid), dim3(block), 0, 0, table, mask); hipEventRecord(e1);
    hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("scatter: %zu dword stores in %.3f ms (%.2f G/s)\n", (size_t)grid * block * K, ms, grid * (double)block * K / ms / 1e6);
    hipEventRecord(e0); hipLaunchKernelGGL(stream_kernel, dim3(grid), dim3(block), 0, 0, (const uint4 *)table, bytes / 16, out); hipEventRecord(e1);
    hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("stream: %zu bytes read in %.3f ms (%.2f TB/s)\n", bytes, ms, bytes / ms / 1e9);
  }
  hipDeviceSynchronize();
  return 0;
}
