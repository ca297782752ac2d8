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

// process-wide arithmetic mode of the attention kernels (pfpp_set_attention_mode): -1 = each kernel's default (environment),
// 0 = exact fp32 matrix instructions, 1 = split-f16, 2 = single-pass fp16 (perf mode, never for parity)
int attn_mode();
inline bool attn_use_f16(bool env_default) { const int m = attn_mode(); return m < 0 ? env_default : m >= 1; }

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace pfpp

// internal (attention_bwd.hip): split-f16 forward of the per-fragment attention, launched by pfpp_attn_blockdiag / _split
int pfpp_attn_blockdiag_f16_launch(const float* qkv, float* out, void* out_hi, void* out_lo, int64_t pairs, int64_t L, int64_t H, float scale,
                                   hipStream_t st);

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

// optional split-f16 copy of a kernel's result (operand of the plane GEMM, csrc/gemm_pl.hip): planes of scale * value.
// Host side: pfpp_planes (include/pfpp.h) -> pfpp_planes_out by pfpp_planes_arg().
struct pfpp_planes_out { _Float16* hi; _Float16* lo; float scale; };
inline pfpp_planes_out pfpp_planes_arg(const pfpp_planes* p) {
  pfpp_planes_out o;
  o.hi = p ? reinterpret_cast<_Float16*>(p->hi) : nullptr;
  o.lo = p ? reinterpret_cast<_Float16*>(p->lo) : nullptr;
  o.scale = p ? p->scale : 1.0f;
  return o;
}
inline bool pfpp_planes_ok(const pfpp_planes* p) {
  return !p || (p->hi && p->lo && (reinterpret_cast<uintptr_t>(p->hi) & 7u) == 0 && (reinterpret_cast<uintptr_t>(p->lo) & 7u) == 0);
}
__device__ __forceinline__ void pfpp_store4_planes(const pfpp_planes_out& po, int64_t idx, float4 v) {
  typedef _Float16 h4_ __attribute__((ext_vector_type(4)));
  h4_ hi, lo;
  PFPP_SPLIT_TO(v.x * po.scale, hi[0], lo[0]);
  PFPP_SPLIT_TO(v.y * po.scale, hi[1], lo[1]);
  PFPP_SPLIT_TO(v.z * po.scale, hi[2], lo[2]);
  PFPP_SPLIT_TO(v.w * po.scale, hi[3], lo[3]);
  *reinterpret_cast<h4_*>(po.hi + idx) = hi;
  *reinterpret_cast<h4_*>(po.lo + idx) = lo;
}

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

// Wave priority of the kernels on the training step's dependency chain (round 5; -DPFPP_CHAIN_PRIO=n at build time, 0 = off):
// s_setprio only arbitrates instruction issue between the waves resident on one SIMD, so chain kernels that share CUs with the next
// iteration's encoder (CU-masked stream) or the weight-gradient stream issue first and the co-runners fill their stalls.  The
// iteration is the chain (4.93 ms alone) stretched by the encoder running under it (5.98 ms; the weight gradients are fully hidden:
// profiles/r05c_ab_masks_and_lab_skips.txt); priority 3 takes 1.3 % off (6.18 -> 6.10 ms alternating on one box,
// profiles/r05d_ab_chain_prio.txt).  Results are unchanged bit for bit (scheduling only).
#ifndef PFPP_CHAIN_PRIO
#define PFPP_CHAIN_PRIO 3
#endif
#ifdef __HIPCC__
__device__ __forceinline__ void pfpp_chain_prio() {
  if constexpr (PFPP_CHAIN_PRIO > 0) __builtin_amdgcn_s_setprio(PFPP_CHAIN_PRIO);
}
#endif
