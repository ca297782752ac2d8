// Shared host/device helpers for libpfpp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "pfpp.h"

#define PFPP_WAVE 64

namespace pfpp {

// thread-local "last error" text behind pfpp_last_error()
void set_error(const char* fmt, ...);

// Every entry point converts its stream right before launching: also drop any sticky error an
// earlier, unrelated HIP call of the process left behind, so check_launch() reports only ours.
inline hipStream_t as_stream(pfpp_stream_t s) {
  (void)hipGetLastError();
  return reinterpret_cast<hipStream_t>(s);
}

// after a launch: turn a launch failure into PFPP_EHIP
int check_launch(const char* what);

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace pfpp

// x = hi + lo with hi = f16(x), lo = f16(x - hi): x - hi is exact in fp32, so the pair carries 22 bits of
// x for |x| >= 2^-3 and an absolute error <= 3e-8 below that (f16 subnormal spacing 6e-8) — the operand
// format of the PFPP_GEMM_F16X3 path (csrc/gemm.hip).  lo is NOT pre-scaled, so hi.hi, hi.lo and lo.hi
// accumulate into one fp32 accumulator.
struct pfpp_hl { _Float16 hi, lo; };
__device__ __forceinline__ pfpp_hl pfpp_split(float x) {
  pfpp_hl r;
  r.hi = (_Float16)x;
  r.lo = (_Float16)(x - (float)r.hi);
  return r;
}
// assigns into two targets (vector elements are not bindable to references)
#define PFPP_SPLIT_TO(x, HI, LO) do { const pfpp_hl _s = pfpp_split(x); (HI) = _s.hi; (LO) = _s.lo; } while (0)

// counter-based generator behind the dropout sites (pfpp_dropout / pfpp_geglu): splitmix64 finaliser
// of (seed, site, element index); forward and backward regenerate the same keep mask.
__host__ __device__ __forceinline__ uint32_t pfpp_rng_u32(uint64_t seed, uint32_t site, uint64_t idx) {
  uint64_t z = seed ^ ((uint64_t)site * 0xD6E8FEB86659FD93ull);
  z += (idx + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(z >> 32);
}
// keep threshold for drop probability p: keep iff rng >= thresh
inline uint32_t pfpp_drop_thresh(float p) {
  const double t = (double)p * 4294967296.0;
  return t <= 0.0 ? 0u : (t >= 4294967295.0 ? 4294967295u : (uint32_t)t);
}

#define PFPP_REQUIRE(cond, msg)                                   \
  do {                                                            \
    if (!(cond)) {                                                \
      pfpp::set_error("%s: %s", __func__, msg);                   \
      return PFPP_EINVAL;                                         \
    }                                                             \
  } while (0)

#define PFPP_SUPPORTED(cond, msg)                                 \
  do {                                                            \
    if (!(cond)) {                                                \
      pfpp::set_error("%s: unsupported: %s", __func__, msg);      \
      return PFPP_EUNSUPPORTED;                                   \
    }                                                             \
  } while (0)
