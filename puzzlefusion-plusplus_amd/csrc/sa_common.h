// helpers shared by the fused set-abstraction kernels (sa_fused.hip: eval mode, sa_train.hip: train-mode BatchNorm by recomputation)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));


__device__ __forceinline__ void split1(float x, _Float16& hi, _Float16& lo) {
  const _Float16 h = (_Float16)x;
  hi = h;
  lo = (_Float16)(x - (float)h);
}

// accumulator tile (lane = sample, register e = channel (e&3) + 8*(e>>2) + 4*lhi) after scale/shift/ReLU -> the two
// 16-deep operand fragments of the next contraction (lane = sample, 8 consecutive channels at 8*lhi): lanes l and l+32
// exchange the quads the other one needs
__device__ __forceinline__ void tile_to_fragments(const f32x16 y, int lhi, half8 (&fh)[2], half8 (&fl)[2]) {
#pragma unroll
  for (int kl = 0; kl < 2; ++kl) {
    half4 lo_h, lo_l, up_h, up_l;        // "lower" quad e = 8*kl .. +3, "upper" quad e = 8*kl+4 .. +7
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      _Float16 a, b;
      split1(y[8 * kl + q], a, b); lo_h[q] = a; lo_l[q] = b;
      split1(y[8 * kl + 4 + q], a, b); up_h[q] = a; up_l[q] = b;
    }
    // lhi = 0 keeps its lower quad (channels 0-3) and needs the partner's lower quad (channels 4-7);
    // lhi = 1 keeps its upper quad (channels 12-15) and needs the partner's upper quad (channels 8-11)
    const half4 send_h = lhi ? lo_h : up_h, send_l = lhi ? lo_l : up_l;
    union { half4 h; int2 i; } sh, sl, rh, rl;
    sh.h = send_h; sl.h = send_l;
    rh.i.x = __shfl_xor(sh.i.x, 32); rh.i.y = __shfl_xor(sh.i.y, 32);
    rl.i.x = __shfl_xor(sl.i.x, 32); rl.i.y = __shfl_xor(sl.i.y, 32);
    const half4 a_h = lhi ? rh.h : lo_h, b_h = lhi ? up_h : rh.h;
    const half4 a_l = lhi ? rl.h : lo_l, b_l = lhi ? up_l : rl.h;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      fh[kl][q] = a_h[q]; fh[kl][4 + q] = b_h[q];
      fl[kl][q] = a_l[q]; fl[kl][4 + q] = b_l[q];
    }
  }
}

}  // namespace
