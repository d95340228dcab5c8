// Microbenchmark: chip-wide rate of copying 64 KiB blocks, one wavefront per block out of a work queue (the shape of the
// decoder's literal-only and stored blocks), by source alignment, bytes in flight per round trip and wavefronts per CU,
// against a plain grid-stride 16-byte copy of the same bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
struct __attribute__((packed)) U128U { uint32_t a, b, c, d; };
__device__ __forceinline__ U128U ld128(const uint8_t *p) { return *(const U128U *)p; }
__device__ __forceinline__ void st128(uint8_t *p, U128U v) { *(U128U *)p = v; }
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: K loads of 16 bytes per lane at the (possibly unaligned) source address, then K aligned stores
// MODE 1: K + 1 aligned loads, bytes shifted into place with v_alignbyte, K aligned stores
// MODE 2: as 0 with nontemporal loads and stores
template <int K, int MODE>
__global__ __launch_bounds__(256) void copy_blocks(uint8_t *dst, const uint8_t *src, uint32_t stride, uint32_t mis, uint32_t nblocks,
                                                   uint32_t *counter) {
  static_assert(65536 % (K * 1024) == 0, "whole rounds only: a round past the block's end would leave the buffers");
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t nwaves = gridDim.x * (blockDim.x / 64);
  uint32_t b = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
  while (b < nblocks) {
    const uint8_t *s = src + (size_t)b * stride + mis;
    uint8_t *d = dst + (size_t)b * 65536;
    for (uint32_t off = 0; off < 65536; off += K * 1024) {
      if (MODE == 1) {
        const uint32_t sh = mis & 3u, wsh = (mis >> 2) & 3u;
        const uint8_t *sa = s - mis;  // aligned
        u32x4 v[K + 1];
#pragma unroll
        for (int k = 0; k < K; k++) v[k] = *(const u32x4 *)(sa + off + 16 * lane + 1024 * k);
        // the sixteen bytes behind this lane's piece: the next lane's piece (lane 63: one more load)
        u32x4 w[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
          u32x4 nx;
          nx.x = __shfl_down(v[k].x, 1); nx.y = __shfl_down(v[k].y, 1); nx.z = __shfl_down(v[k].z, 1); nx.w = __shfl_down(v[k].w, 1);
          if (lane == 63) nx = *(const u32x4 *)(sa + off + 1024 * (k + 1));
          uint32_t q[8] = {v[k].x, v[k].y, v[k].z, v[k].w, nx.x, nx.y, nx.z, nx.w};
          uint32_t r[4];
#pragma unroll
          for (int i = 0; i < 4; i++) {
            uint32_t lo = wsh == 0 ? q[i] : wsh == 1 ? q[i + 1] : wsh == 2 ? q[i + 2] : q[i + 3];
            uint32_t hi = wsh == 0 ? q[i + 1] : wsh == 1 ? q[i + 2] : wsh == 2 ? q[i + 3] : q[i + 4];
            r[i] = __builtin_amdgcn_alignbyte(hi, lo, sh);
          }
          w[k] = u32x4{r[0], r[1], r[2], r[3]};
        }
#pragma unroll
        for (int k = 0; k < K; k++) *(u32x4 *)(d + off + 16 * lane + 1024 * k) = w[k];
      } else {
        U128U v[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
          if (MODE == 2) { const u32x4 t = __builtin_nontemporal_load((const u32x4 *)(s + off + 16 * lane + 1024 * k)); v[k] = U128U{t.x, t.y, t.z, t.w}; }
          else v[k] = ld128(s + off + 16 * lane + 1024 * k);
        }
#pragma unroll
        for (int k = 0; k < K; k++) {
          if (MODE == 2) __builtin_nontemporal_store(u32x4{v[k].a, v[k].b, v[k].c, v[k].d}, (u32x4 *)(d + off + 16 * lane + 1024 * k));
          else st128(d + off + 16 * lane + 1024 * k, v[k]);
        }
      }
    }
    uint32_t nx = 0;
    if (lane == 0) nx = atomicAdd(counter, 1u);
    b = nwaves + __shfl(nx, 0);
  }
}

__global__ __launch_bounds__(256) void copy_flat(u32x4 *dst, const u32x4 *src, size_t n16) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

template <int K, int MODE>
static void run(const char *nm, uint8_t *d, const uint8_t *s, uint32_t stride, uint32_t mis, uint32_t nb, uint32_t wpc, uint32_t *cnt) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 5; r++) {
    (void)hipMemsetAsync(cnt, 0, 4, 0);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((copy_blocks<K, MODE>), dim3(256 * wpc / 4), dim3(256), 0, 0, d, s, stride, mis, nb, cnt);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("%-34s K=%2d mis=%2u %2u waves/CU: %.3f ms, %.2f TB/s\n", nm, K, mis, wpc, best, 2.0 * nb * 65536 / best / 1e9);
}

int main() {
  const uint32_t nb = 16384, stride = 65536 + 320;
  uint8_t *s, *d; uint32_t *cnt;
  (void)hipMalloc(&s, (size_t)nb * stride + 4096); (void)hipMalloc(&d, (size_t)nb * 65536); (void)hipMalloc(&cnt, 4);
  (void)hipMemset(s, 1, (size_t)nb * stride + 4096);
  {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (uint32_t g : {2048u, 4096u, 8192u}) {
      float best = 1e9f;
      for (int r = 0; r < 5; r++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(copy_flat, dim3(g), dim3(256), 0, 0, (u32x4 *)d, (const u32x4 *)s, (size_t)nb * 4096);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      printf("flat 16-byte grid-stride copy, %u workgroups: %.3f ms, %.2f TB/s\n", g, best, 2.0 * nb * 65536 / best / 1e9);
    }
  }
  for (uint32_t wpc : {8u, 16u, 32u}) {
    run<8, 0>("blocks, plain", d, s, stride, 0, nb, wpc, cnt);
    run<8, 0>("blocks, plain", d, s, stride, 2, nb, wpc, cnt);
    run<8, 0>("blocks, plain", d, s, stride, 258, nb, wpc, cnt);
    run<8, 1>("blocks, aligned loads + alignbyte", d, s, stride, 258, nb, wpc, cnt);
    run<8, 2>("blocks, nontemporal", d, s, stride, 258, nb, wpc, cnt);
    run<4, 0>("blocks, plain", d, s, stride, 258, nb, wpc, cnt);
    run<16, 0>("blocks, plain", d, s, stride, 258, nb, wpc, cnt);
  }
  return 0;
}
