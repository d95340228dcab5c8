// Optional in-kernel phase timing (diagnostic builds only: make prof).  Each stamp adds the
// cycles since the previous stamp to a per-wavefront LDS slot (one column per lane, so no
// conflicts and no scalar-register pressure); lane 0's column is added to a device array at the
// end of a block.  No output depends on the stamps.
#pragma once
#ifdef LZ4MI_PROFILE
#define PH_SLOTS 32
namespace lz4mi {
__device__ __forceinline__ __attribute__((address_space(3))) uint32_t *ph_column() {
  __shared__ uint32_t slots[4 * PH_SLOTS * 64];
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x / 64);
  return (__attribute__((address_space(3))) uint32_t *)slots + wave * PH_SLOTS * 64 + (threadIdx.x & 63);
}
}  // namespace lz4mi
// the stamps' state as one object, for code outside the function that declared it
struct PhState {
  __attribute__((address_space(3))) uint32_t *c;
  unsigned long long t;
};
#define PHS(st, i)                                               \
  do {                                                           \
    unsigned long long n_ = __builtin_readcyclecounter();        \
    (st).c[(i) * 64] += (uint32_t)(n_ - (st).t);                 \
    (st).t = n_;                                                 \
  } while (0)
#define PH_DECL                                                              \
  __attribute__((address_space(3))) uint32_t *ph_c_ = lz4mi::ph_column();    \
  for (int q_ = 0; q_ < PH_SLOTS; q_++) ph_c_[q_ * 64] = 0;                  \
  unsigned long long ph_t_ = __builtin_readcyclecounter();                   \
  const unsigned long long ph_t0_ = ph_t_, ph_r0_ = __builtin_amdgcn_s_memrealtime();
#define PH(i)                                                    \
  do {                                                           \
    unsigned long long n_ = __builtin_readcyclecounter();        \
    ph_c_[(i) * 64] += (uint32_t)(n_ - ph_t_);                   \
    ph_t_ = n_;                                                  \
  } while (0)
#define PH_CNT(i) do { ph_c_[(i) * 64] += 1; } while (0)
#define PH_FLUSH(arr, lane)                                                                          \
  do {                                                                                               \
    ph_c_[15 * 64] = (uint32_t)(__builtin_readcyclecounter() - ph_t0_);                              \
    if ((lane) == 0)                                                                                 \
      for (int q_ = 0; q_ < PH_SLOTS; q_++)                                                          \
        if (ph_c_[q_ * 64]) atomicAdd(&(arr)[q_], (unsigned long long)ph_c_[q_ * 64]);               \
  } while (0)
#else
#define PH_DECL
#define PH(i) do {} while (0)
#define PH_CNT(i) do {} while (0)
#define PH_FLUSH(arr, lane) do {} while (0)
#define PHS(st, i) do {} while (0)
struct PhState {};
#endif
