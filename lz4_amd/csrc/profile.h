// Optional in-kernel phase timing (diagnostic builds only: make prof).  Stamps are summed per
// wavefront in registers and added to a device array of their own; no output depends on them.
#pragma once
#ifdef LZ4MI_PROFILE
#define PH_SLOTS 16
#define PH_DECL unsigned long long ph_t_ = __builtin_readcyclecounter(); unsigned long long ph_a_[PH_SLOTS] = {0};
#define PH(i) do { unsigned long long n_ = __builtin_readcyclecounter(); ph_a_[i] += n_ - ph_t_; ph_t_ = n_; } while (0)
#define PH_CNT(i) do { ph_a_[i] += 1; } while (0)
#define PH_FLUSH(arr, lane) do { if ((lane) == 0) for (int q_ = 0; q_ < PH_SLOTS; q_++) if (ph_a_[q_]) atomicAdd(&(arr)[q_], ph_a_[q_]); } while (0)
#else
#define PH_DECL
#define PH(i) do {} while (0)
#define PH_CNT(i) do {} while (0)
#define PH_FLUSH(arr, lane) do {} while (0)
#endif
