// Variants of the decoder's in-order match copy loop on synthetic descriptors (diagnostic).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) uint8_t lds_u8;
__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return __builtin_amdgcn_readlane(v, l); }
#define REP 16

template <int V>
__global__ __launch_bounds__(64) void k(uint64_t *out, uint32_t *sink, const uint32_t *dA, const uint32_t *dB) {
  __shared__ __attribute__((aligned(16))) uint8_t buf[8224 + 64];
  lds_u8 *stg = (lds_u8 *)buf;
  const uint32_t lane = threadIdx.x;
  for (uint32_t i = lane; i < 8224 + 64; i += 64) buf[i] = (uint8_t)(i * 7);
  __syncthreads();
  const uint32_t vA = dA[lane], vB = dB[lane];
  const uint32_t dsafe_ = (vB >> 16) ? (vB >> 16) : 1u;
  const uint32_t vC = (65536u / dsafe_ + 1u) | ((64u % dsafe_) << 17);
  const uint32_t nb = 64;
  const float lanef = (float)lane + 0.5f;
  uint64_t t0 = __builtin_readcyclecounter();
  for (int rep = 0; rep < REP; rep++) {
    if (V == 0) {  // as in the library
      uint32_t pa = 0, pb = 0, pdst = 0, pcnt = 0;
      for (uint32_t j = 0; j < nb; j++) {
        const uint32_t B = rdlane(vB, j);
        const uint32_t m = B & 0xffffu;
        if (!m) continue;
        const uint32_t d = B >> 16, A = rdlane(vA, j);
        const uint32_t ms = A & 0xffffu, md = (A >> 16) + lane;
        uint32_t r = lane, step = 64, per = 0x40000000u;
        if (d < m) {
          const float rc = __builtin_amdgcn_rcpf((float)d);
          r = lane - __umul24((uint32_t)(lanef * rc), d);
          step = 64u - __umul24((uint32_t)(64.5f * rc), d);
          per = d;
        }
        for (uint32_t k0 = 0; k0 < m; k0 += 128) {
          if (lane < pcnt) stg[pdst] = (uint8_t)pa;
          if (lane + 64 < pcnt) stg[pdst + 64] = (uint8_t)pb;
          const uint32_t ra = r; r += step; r = r >= per ? r - per : r;
          const uint32_t rb = r; r += step; r = r >= per ? r - per : r;
          pa = stg[ms + ra]; pb = stg[ms + rb];
          pdst = md + k0; pcnt = m - k0;
        }
      }
      if (lane < pcnt) stg[pdst] = (uint8_t)pa;
      if (lane + 64 < pcnt) stg[pdst + 64] = (uint8_t)pb;
    } else if (V == 1) {  // simplest: read, write, no pipelining, one piece per step
      for (uint32_t j = 0; j < nb; j++) {
        const uint32_t B = rdlane(vB, j);
        const uint32_t m = B & 0xffffu;
        if (!m) continue;
        const uint32_t d = B >> 16, A = rdlane(vA, j);
        const uint32_t ms = A & 0xffffu, md = (A >> 16) + lane;
        uint32_t r = lane, step = 64, per = 0x40000000u;
        if (d < m) {
          const float rc = __builtin_amdgcn_rcpf((float)d);
          r = lane - __umul24((uint32_t)(lanef * rc), d);
          step = 64u - __umul24((uint32_t)(64.5f * rc), d);
          per = d;
        }
        for (uint32_t k0 = 0; k0 < m; k0 += 64) {
          const uint32_t v = stg[ms + r];
          if (k0 + lane < m) stg[md + k0] = (uint8_t)v;
          r += step; r = r >= per ? r - per : r;
        }
      }
    } else if (V == 2) {  // plain copies only (no modulo), to see what the set-up costs
      for (uint32_t j = 0; j < nb; j++) {
        const uint32_t B = rdlane(vB, j);
        const uint32_t m = B & 0xffffu;
        if (!m) continue;
        const uint32_t A = rdlane(vA, j);
        const uint32_t ms = (A & 0xffffu) + lane, md = (A >> 16) + lane;
        for (uint32_t k0 = 0; k0 < m; k0 += 64) {
          const uint32_t v = stg[ms + k0];
          if (k0 + lane < m) stg[md + k0] = (uint8_t)v;
        }
      }
    } else if (V == 4) {  // full functionality, one piece per step, dummy-address masking
      for (uint32_t j = 0; j < nb; j++) {
        const uint32_t B = rdlane(vB, j);
        const uint32_t m = B & 0xffffu;
        if (!m) continue;
        const uint32_t d = B >> 16, A = rdlane(vA, j);
        const uint32_t ms = A & 0xffffu, md = (A >> 16) + lane;
        uint32_t r = lane, step = 64, per = 0x40000000u;
        if (d < m) {
          const float rc = __builtin_amdgcn_rcpf((float)d);
          r = lane - __umul24((uint32_t)(lanef * rc), d);
          step = 64u - __umul24((uint32_t)(64.5f * rc), d);
          per = d;
        }
        for (uint32_t k0 = 0; k0 < m; k0 += 64) {
          const uint32_t v = stg[ms + r];
          const uint32_t a = k0 + lane < m ? md + k0 : 8224u + lane;
          stg[a] = (uint8_t)v;
          r += step; r = r >= per ? r - per : r;
        }
      }
    } else if (V == 5) {  // write-behind, one piece per step, dummy-address masking
      uint32_t pv = 0, pa = 8224u + lane;
      for (uint32_t j = 0; j < nb; j++) {
        const uint32_t B = rdlane(vB, j);
        const uint32_t m = B & 0xffffu;
        if (!m) continue;
        const uint32_t d = B >> 16, A = rdlane(vA, j);
        const uint32_t ms = A & 0xffffu, md = (A >> 16) + lane;
        uint32_t r = lane, step = 64, per = 0x40000000u;
        if (d < m) {
          const float rc = __builtin_amdgcn_rcpf((float)d);
          r = lane - __umul24((uint32_t)(lanef * rc), d);
          step = 64u - __umul24((uint32_t)(64.5f * rc), d);
          per = d;
        }
        for (uint32_t k0 = 0; k0 < m; k0 += 64) {
          stg[pa] = (uint8_t)pv;
          pv = stg[ms + r];
          pa = k0 + lane < m ? md + k0 : 8224u + lane;
          r += step; r = r >= per ? r - per : r;
        }
      }
      stg[pa] = (uint8_t)pv;
    } else if (V == 6) {  // as 4, branch-free modulo set-up (always computed)
      for (uint32_t j = 0; j < nb; j++) {
        const uint32_t B = rdlane(vB, j);
        const uint32_t m = B & 0xffffu;
        if (!m) continue;
        const uint32_t d = B >> 16, A = rdlane(vA, j);
        const uint32_t ms = A & 0xffffu, md = (A >> 16) + lane;
        const float rc = __builtin_amdgcn_rcpf((float)d);
        uint32_t r = lane - __umul24((uint32_t)(lanef * rc), d);
        const uint32_t step = 64u - __umul24((uint32_t)(64.5f * rc), d);
        const uint32_t per = d < m ? d : 0x40000000u;
        r = d < m ? r : lane;
        for (uint32_t k0 = 0; k0 < m; k0 += 64) {
          const uint32_t v = stg[ms + r];
          const uint32_t a = k0 + lane < m ? md + k0 : 8224u + lane;
          stg[a] = (uint8_t)v;
          r += d < m ? step : 64u; r = r >= per ? r - per : r;
        }
      }
    } else if (V == 7) {  // magic modulo, two pieces straight-line, loop only past 128 bytes, no skip branch
      for (uint32_t j = 0; j < nb; j++) {
        const uint32_t B = rdlane(vB, j), A = rdlane(vA, j), C = rdlane(vC, j);
        const uint32_t m = B & 0xffffu, d = B >> 16;
        const uint32_t ms = A & 0xffffu, md = (A >> 16) + lane;
        const uint32_t per = d < m ? d : 0x40000000u;
        const uint32_t step = d < m ? C >> 17 : 64u;
        uint32_t r = lane - __umul24(__umul24(lane, C & 0x1ffffu) >> 16, d);
        uint32_t r2 = r + step; r2 = r2 >= per ? r2 - per : r2;
        const uint32_t va = stg[ms + r], vb = stg[ms + r2];
        stg[lane < m ? md : 8224u + lane] = (uint8_t)va;
        stg[lane + 64 < m ? md + 64 : 8224u + lane] = (uint8_t)vb;
        if (m > 128) {
          r = r2 + step; r = r >= per ? r - per : r;
          for (uint32_t k0 = 128; k0 < m; k0 += 64) {
            const uint32_t v = stg[ms + r];
            stg[k0 + lane < m ? md + k0 : 8224u + lane] = (uint8_t)v;
            r += step; r = r >= per ? r - per : r;
          }
        }
      }
    } else if (V == 8) {  // as 7, software pipelined: the reads of match j are issued, then match j + 1 is set up, then j is written
      uint32_t B = rdlane(vB, 0), A = rdlane(vA, 0), C = rdlane(vC, 0);
      for (uint32_t j = 0; j < nb; j++) {
        const uint32_t m = B & 0xffffu, d = B >> 16;
        const uint32_t ms = A & 0xffffu, md = (A >> 16) + lane;
        const uint32_t per = d < m ? d : 0x40000000u;
        const uint32_t step = d < m ? C >> 17 : 64u;
        uint32_t r = lane - __umul24(__umul24(lane, C & 0x1ffffu) >> 16, d);
        uint32_t r2 = r + step; r2 = r2 >= per ? r2 - per : r2;
        const uint32_t va = stg[ms + r], vb = stg[ms + r2];
        const uint32_t wa = lane < m ? md : 8224u + lane, wb = lane + 64 < m ? md + 64 : 8224u + lane;
        const uint32_t jn = j + 1 < nb ? j + 1 : j;
        B = rdlane(vB, jn); A = rdlane(vA, jn); C = rdlane(vC, jn);
        stg[wa] = (uint8_t)va;
        stg[wb] = (uint8_t)vb;
        if (m > 128) {
          r = r2 + step; r = r >= per ? r - per : r;
          for (uint32_t k0 = 128; k0 < m; k0 += 64) {
            const uint32_t v = stg[ms + r];
            stg[k0 + lane < m ? md + k0 : 8224u + lane] = (uint8_t)v;
            r += step; r = r >= per ? r - per : r;
          }
        }
      }
    } else if (V == 3) {  // plain copies, dummy-address masking instead of exec masking
      for (uint32_t j = 0; j < nb; j++) {
        const uint32_t B = rdlane(vB, j);
        const uint32_t m = B & 0xffffu;
        if (!m) continue;
        const uint32_t A = rdlane(vA, j);
        const uint32_t ms = (A & 0xffffu) + lane, md = (A >> 16) + lane;
        for (uint32_t k0 = 0; k0 < m; k0 += 64) {
          const uint32_t v = stg[ms + k0];
          const uint32_t a = k0 + lane < m ? md + k0 : 8224u + lane;
          stg[a] = (uint8_t)v;
        }
      }
    }
  }
  uint64_t t1 = __builtin_readcyclecounter();
  if (lane == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  sink[blockIdx.x * 64 + lane] = buf[lane * 3];
}

int main(int argc, char **argv) {
  int wpc = argc > 1 ? atoi(argv[1]) : 1;
  uint64_t *d_out; uint32_t *d_sink, *dA, *dB;
  (void)hipMalloc(&d_out, 64); (void)hipMalloc(&d_sink, 256 * 64 * 4 * 32); (void)hipMalloc(&dA, 256); (void)hipMalloc(&dB, 256);
  uint32_t hA[64], hB[64];
  uint32_t pos = 4200, x = 12345;
  int pieces = 0;
  for (int j = 0; j < 64; j++) {
    x = x * 1664525u + 1013904223u;
    uint32_t ll = (x >> 8) % 40;
    x = x * 1664525u + 1013904223u;
    uint32_t m = 4 + (x >> 8) % 120;          // mean ~64
    x = x * 1664525u + 1013904223u;
    uint32_t bits = 1 + (x >> 8) % 12;
    x = x * 1664525u + 1013904223u;
    uint32_t d = 1 + ((x >> 8) & ((1u << bits) - 1));
    pos += ll;
    if (d > pos) d = pos;
    hA[j] = (pos - d) | (pos << 16);
    hB[j] = m | (d << 16);
    pos += m;
    pieces += (m + 63) / 64;
  }
  (void)hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
  printf("64 matches, %d pieces, end offset %u, %d waves/CU\n", pieces, pos, wpc);
#define RUN(V, name)                                                                          \
  for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k<V>), dim3(256 * wpc), dim3(64), 0, 0, d_out, d_sink, dA, dB); \
  (void)hipDeviceSynchronize();                                                               \
  { uint64_t h; (void)hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost); printf("%-40s %7.1f cycles/match\n", name, (double)h / REP / 64); }
  RUN(0, "library loop (write-behind, 2 pieces)");
  RUN(1, "read-write per piece");
  RUN(2, "plain copies only");
  RUN(3, "plain, dummy-address masking");
  RUN(4, "full, 1 piece/step, dummy masking");
  RUN(5, "full, write-behind, dummy masking");
  RUN(6, "full, branch-free modulo, dummy masking");
  RUN(7, "magic modulo, 2 pieces straight-line");
  RUN(8, "magic modulo, 2 pieces, pipelined set-up");
  return 0;
}
