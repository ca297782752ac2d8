// a4 + a5 + a6 for a set-abstraction level WITHOUT input features (sa1 of the fragment encoder, vqvae/model/modules/pn2.py:16,
// utils/pn2_utils.py:127-151 grouping, :210-216 three [1x1 conv -> BatchNorm(eval, folded) -> ReLU] and the max over nsample)
// in ONE kernel: the level's activations (1.26 M rows x 64 / 64 / 128 channels at the benchmark shape, 1.3 GB of HBM traffic
// when each layer is its own GEMM) never leave the registers.
//
// One wave owns one neighbourhood (32 samples) at a time:
//   * the samples' offsets from the centroid (K = 3, padded to 16) are built in registers as an MFMA operand;
//   * layers 1 and 2 are computed TRANSPOSED (channels x samples = W . X^T): the accumulator then holds, per lane, one sample
//     and 16 of its channels — after the epilogue, one exchange between lanes l and l+32 turns that into the next layer's
//     operand fragment (8 consecutive channels per lane), so the activations go accumulator -> fragment without LDS;
//   * layer 3 runs in the normal orientation (samples x channels): the accumulator has one channel per lane and the 32
//     samples in registers, so the max over the neighbourhood is 16 fmaxf + one cross-half exchange, as in gemm_common.h.
// Arithmetic is the split-f16 contraction of gemm.hip with the same operand split, the same three products per 16-deep step
// in the same order and the same fused multiply-add epilogue, i.e. the result of pfpp_group_gather + 3 x pfpp_gemm.
// The weights' fp16 planes live in REGISTERS (208 per lane, at one wave per SIMD a wave may hold 512) for the lifetime of
// the persistent wave; the next neighbourhood's indices and coordinates are fetched while the current one is computed.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pfpp.h"
#include "pfpp_common.h"
#include "sa_common.h"

namespace {

struct SaP {
  const float* xyz; const float* ctr; const int32_t* idx;
  const _Float16* w0h; const _Float16* w0l; const _Float16* w1h; const _Float16* w1l; const _Float16* w2h; const _Float16* w2l;
  const float* s0; const float* t0; const float* s1; const float* t1; const float* s2; const float* t2;
  float* out;
  int N, S, G;
};

template <int C1, int C2, int C3>
__global__ __launch_bounds__(256, 1) void sa_mlp3_kernel(const SaP p) {
  // LDS: only the folded BatchNorm scale / shift vectors (the transposed layers index them by accumulator register)
  __shared__ __align__(16) float S0[C1], T0[C1], S1[C2], T1[C2];
  const int tid = threadIdx.x;
  for (int i = tid; i < C1; i += 256) { S0[i] = p.s0[i]; T0[i] = p.t0[i]; }
  for (int i = tid; i < C2; i += 256) { S1[i] = p.s1[i]; T1[i] = p.t1[i]; }
  __syncthreads();

  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;

  // ---- every weight fragment this lane will ever need, in registers for the lifetime of the (persistent) wave:
  // (2 + 8 + 16) fragments x 2 planes x 4 registers = 208 of the 512 a wave may hold at one wave per SIMD ----
  half8 w0h[C1 / 32], w0l[C1 / 32];
  half8 w1h[C2 / 32][C1 / 16], w1l[C2 / 32][C1 / 16];
  half8 w2h[C3 / 32][C2 / 16], w2l[C3 / 32][C2 / 16];
#pragma unroll
  for (int t = 0; t < C1 / 32; ++t) {
#pragma unroll
    for (int q = 0; q < 8; ++q) { w0h[t][q] = (_Float16)0.0f; w0l[t][q] = (_Float16)0.0f; }
    if (lhi == 0) {      // K = 3 (+ a zero) of 16: only the first 8-half group is populated
      w0h[t] = *reinterpret_cast<const half8*>(p.w0h + (t * 32 + l31) * 8);
      w0l[t] = *reinterpret_cast<const half8*>(p.w0l + (t * 32 + l31) * 8);
    }
  }
#pragma unroll
  for (int t = 0; t < C2 / 32; ++t)
#pragma unroll
    for (int ks = 0; ks < C1 / 16; ++ks) {
      w1h[t][ks] = *reinterpret_cast<const half8*>(p.w1h + (t * 32 + l31) * C1 + ks * 16 + lhi * 8);
      w1l[t][ks] = *reinterpret_cast<const half8*>(p.w1l + (t * 32 + l31) * C1 + ks * 16 + lhi * 8);
    }
#pragma unroll
  for (int n = 0; n < C3 / 32; ++n)
#pragma unroll
    for (int ks = 0; ks < C2 / 16; ++ks) {
      w2h[n][ks] = *reinterpret_cast<const half8*>(p.w2h + (n * 32 + l31) * C2 + ks * 16 + lhi * 8);
      w2l[n][ks] = *reinterpret_cast<const half8*>(p.w2l + (n * 32 + l31) * C2 + ks * 16 + lhi * 8);
    }
  float sc2[C3 / 32], sh2[C3 / 32];
#pragma unroll
  for (int n = 0; n < C3 / 32; ++n) { sc2[n] = p.s2[n * 32 + l31]; sh2[n] = p.t2[n * 32 + l31]; }

  // folded BatchNorm + ReLU of a transposed tile: channel of register e is c0 + (e&3) + 8*(e>>2) + 4*lhi
  auto bn_relu_t = [&](f32x16 acc, const float* sc, const float* sh, int c0) {
    f32x16 y;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 s4 = *reinterpret_cast<const float4*>(sc + c0 + 8 * q + 4 * lhi);
      const float4 t4 = *reinterpret_cast<const float4*>(sh + c0 + 8 * q + 4 * lhi);
      const float sv[4] = {s4.x, s4.y, s4.z, s4.w}, tv[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = __builtin_fmaf(acc[4 * q + r], sv[r], tv[r]);
        y[4 * q + r] = v > 0.0f ? v : 0.0f;
      }
    }
    return y;
  };

  // ---- software pipeline over the wave's neighbourhoods: indices two ahead, coordinates one ahead ----
  const int stride = gridDim.x * 4;
  const int g0 = blockIdx.x * 4 + wave;
  auto load_id = [&](int g) {
    const int gc = g < p.G ? g : p.G - 1;
    const int id = p.idx[(int64_t)gc * 32 + l31];
    return id < p.N ? id : p.N - 1;                   // memory safety only, as in group_gather_kernel
  };
  auto load_pt = [&](int g, int id, float (&q)[3], float (&c)[3]) {
    const int gc = g < p.G ? g : p.G - 1;
    const int f = gc / p.S;
    const float* q3 = p.xyz + ((int64_t)f * p.N + id) * 3;
    const float* c3 = p.ctr + (int64_t)gc * 3;
#pragma unroll
    for (int d = 0; d < 3; ++d) { q[d] = q3[d]; c[d] = c3[d]; }
  };
  float q_cur[3], c_cur[3], q_nxt[3], c_nxt[3];
  int id_nxt;
  load_pt(g0, load_id(g0), q_cur, c_cur);
  id_nxt = load_id(g0 + stride);

  for (int g = g0; g < p.G; g += stride) {
    load_pt(g + stride, id_nxt, q_nxt, c_nxt);
    id_nxt = load_id(g + 2 * stride);

    // ---- the neighbourhood's operand: lane l31 = sample, k = 0..2 its offset from the centroid (lhi = 0 half only) ----
    half8 xh, xl;
#pragma unroll
    for (int q = 0; q < 8; ++q) { xh[q] = (_Float16)0.0f; xl[q] = (_Float16)0.0f; }
    if (lhi == 0) {
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        _Float16 a, b;
        split1(__fsub_rn(q_cur[d], c_cur[d]), a, b);
        xh[d] = a; xl[d] = b;
      }
    }

    // ---- layer 1 (transposed): [C1 x 16] . [16 x 32 samples] ----
    half8 f1h[C1 / 16], f1l[C1 / 16];
#pragma unroll
    for (int t = 0; t < C1 / 32; ++t) {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0h[t], xl, acc, 0, 0, 0);     // x_lo . w_hi
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0l[t], xh, acc, 0, 0, 0);     // x_hi . w_lo
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0h[t], xh, acc, 0, 0, 0);     // x_hi . w_hi
      const f32x16 y = bn_relu_t(acc, S0, T0, t * 32);
      half8 fh[2], fl[2];
      tile_to_fragments(y, lhi, fh, fl);
      f1h[2 * t] = fh[0]; f1h[2 * t + 1] = fh[1];
      f1l[2 * t] = fl[0]; f1l[2 * t + 1] = fl[1];
    }

    // ---- layer 2 (transposed): [C2 x C1] . [C1 x 32 samples] ----
    half8 f2h[C2 / 16], f2l[C2 / 16];
#pragma unroll
    for (int t = 0; t < C2 / 32; ++t) {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < C1 / 16; ++ks) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h[t][ks], f1l[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1l[t][ks], f1h[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h[t][ks], f1h[ks], acc, 0, 0, 0);
      }
      const f32x16 y = bn_relu_t(acc, S1, T1, t * 32);
      half8 fh[2], fl[2];
      tile_to_fragments(y, lhi, fh, fl);
      f2h[2 * t] = fh[0]; f2h[2 * t + 1] = fh[1];
      f2l[2 * t] = fl[0]; f2l[2 * t + 1] = fl[1];
    }

    // ---- layer 3 (samples x channels) + max over the 32 samples ----
#pragma unroll
    for (int n = 0; n < C3 / 32; ++n) {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < C2 / 16; ++ks) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f2l[ks], w2h[n][ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f2h[ks], w2l[n][ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f2h[ks], w2h[n][ks], acc, 0, 0, 0);
      }
      float m = -__builtin_huge_valf();
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float v = __builtin_fmaf(acc[e], sc2[n], sh2[n]);
        v = v > 0.0f ? v : 0.0f;
        m = fmaxf(m, v);
      }
      m = fmaxf(m, __shfl_xor(m, 32));
      if (lhi == 0) p.out[(int64_t)g * C3 + n * 32 + l31] = m;
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) { q_cur[d] = q_nxt[d]; c_cur[d] = c_nxt[d]; }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Level with input features (sa2: 128 features + 3 coordinates -> 128 -> 128, nsample 64): grouping + the FIRST TWO folded
// conv/BN/ReLU in one kernel; the third layer (its 256 x 128 weights no longer fit next to the others) stays a GEMM with the
// max-pool epilogue.  Same scheme as above — a wave owns one neighbourhood (two 32-sample tiles), both layers transposed,
// accumulator -> operand fragment through the lane exchange — but the weights (147 KB of fp16 planes) sit in LDS, shared by
// the 4 waves of the one workgroup a CU holds, and the sample operand of layer 1 is read straight from the level's feature
// table: every lane fetches the 32-byte runs of ITS neighbour's row (by ball-query index) into registers and splits them
// there; the rows of the next neighbourhood are in flight while layer 2 of the current one computes.
// Removes the 646 MB write + 646 MB read of the first layer's activations (and the gather pass before it).
struct Sa2P {
  const float* feats; const float* xyz; const float* ctr; const int32_t* idx;
  const _Float16* w0h; const _Float16* w0l; const _Float16* w1h; const _Float16* w1l;
  const float* s0; const float* t0; const float* s1; const float* t1;
  float* out;        // [G*64, C2] (or null)
  _Float16* out_hi; _Float16* out_lo;   // the same rows as split-f16 planes (operand of the plane GEMM of layer 3), or null
  int N, S, G;
};

template <int D, int C1, int C2>
__global__ __launch_bounds__(256, 1) void sa_mlp2_kernel(const Sa2P p) {
  constexpr int KS0 = D / 16 + 1;                 // 16-deep steps of layer 1: the features, then [dx dy dz 0 ...]
  constexpr int KP0 = D + 8;                      // row length of the layer-1 planes in memory (K = D + 3, padded to 8)
  constexpr int LD0 = KS0 * 16 + 8, LD1 = C1 + 8; // LDS row strides in halfs (16-byte fragment reads conflict-free)
  extern __shared__ __align__(16) unsigned char sa2_smem[];
  _Float16* W0h = reinterpret_cast<_Float16*>(sa2_smem);
  _Float16* W0l = W0h + C1 * LD0;
  _Float16* W1h = W0l + C1 * LD0;
  _Float16* W1l = W1h + C2 * LD1;
  float* S0 = reinterpret_cast<float*>(W1l + C2 * LD1);
  float* T0 = S0 + C1;
  float* S1 = T0 + C1;
  float* T1 = S1 + C2;

  const int tid = threadIdx.x;
  for (int i = tid; i < C1 * (LD0 / 8); i += 256) {
    const int r = i / (LD0 / 8), c8 = i - r * (LD0 / 8);
    uint4 vh = make_uint4(0, 0, 0, 0), vl = vh;
    if (c8 * 8 < KP0) {
      vh = *reinterpret_cast<const uint4*>(p.w0h + (size_t)r * KP0 + c8 * 8);
      vl = *reinterpret_cast<const uint4*>(p.w0l + (size_t)r * KP0 + c8 * 8);
    }
    *reinterpret_cast<uint4*>(W0h + r * LD0 + c8 * 8) = vh;
    *reinterpret_cast<uint4*>(W0l + r * LD0 + c8 * 8) = vl;
  }
  for (int i = tid; i < C2 * (C1 / 8); i += 256) {
    const int r = i / (C1 / 8), c8 = i - r * (C1 / 8);
    *reinterpret_cast<uint4*>(W1h + r * LD1 + c8 * 8) = *reinterpret_cast<const uint4*>(p.w1h + (size_t)r * C1 + c8 * 8);
    *reinterpret_cast<uint4*>(W1l + r * LD1 + c8 * 8) = *reinterpret_cast<const uint4*>(p.w1l + (size_t)r * C1 + c8 * 8);
  }
  for (int i = tid; i < C1; i += 256) { S0[i] = p.s0[i]; T0[i] = p.t0[i]; }
  for (int i = tid; i < C2; i += 256) { S1[i] = p.s1[i]; T1[i] = p.t1[i]; }
  __syncthreads();

  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;

  auto bn_relu_t = [&](f32x16 acc, const float* sc, const float* sh, int c0) {
    f32x16 y;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 s4 = *reinterpret_cast<const float4*>(sc + c0 + 8 * q + 4 * lhi);
      const float4 t4 = *reinterpret_cast<const float4*>(sh + c0 + 8 * q + 4 * lhi);
      const float sv[4] = {s4.x, s4.y, s4.z, s4.w}, tv[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = __builtin_fmaf(acc[4 * q + r], sv[r], tv[r]);
        y[4 * q + r] = v > 0.0f ? v : 0.0f;
      }
    }
    return y;
  };
  auto split8 = [&](const float4 a, const float4 b, half8& hi, half8& lo) {
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      _Float16 h, l;
      split1(x[q], h, l);
      hi[q] = h; lo[q] = l;
    }
  };

  const int stride = gridDim.x * 4;
  const int g0 = blockIdx.x * 4 + wave;
  auto load_ids = [&](int g, int (&id)[2]) {
    const int gc = g < p.G ? g : p.G - 1;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const int v = p.idx[(int64_t)gc * 64 + st * 32 + l31];
      id[st] = v < p.N ? v : p.N - 1;               // memory safety only, as in group_gather_kernel
    }
  };
  // this lane's part of its two samples' feature rows: for every 16-deep step the 8 values at 8*lhi
  auto load_rows = [&](int g, const int (&id)[2], float4 (&raw)[2][D / 16][2], float (&q)[2][3], float (&c)[3]) {
    const int gc = g < p.G ? g : p.G - 1;
    const int f = gc / p.S;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const float* row = p.feats + ((int64_t)f * p.N + id[st]) * D + lhi * 8;
#pragma unroll
      for (int ks = 0; ks < D / 16; ++ks) {
        raw[st][ks][0] = *reinterpret_cast<const float4*>(row + ks * 16);
        raw[st][ks][1] = *reinterpret_cast<const float4*>(row + ks * 16 + 4);
      }
      const float* q3 = p.xyz + ((int64_t)f * p.N + id[st]) * 3;
#pragma unroll
      for (int d = 0; d < 3; ++d) q[st][d] = q3[d];
    }
    const float* c3 = p.ctr + (int64_t)gc * 3;
#pragma unroll
    for (int d = 0; d < 3; ++d) c[d] = c3[d];
  };

  float4 raw[2][D / 16][2];
  float qx[2][3], cx[3];
  int id_nxt[2];
  {
    int id0[2];
    load_ids(g0, id0);
    load_rows(g0, id0, raw, qx, cx);
    load_ids(g0 + stride, id_nxt);
  }

  for (int g = g0; g < p.G; g += stride) {
    // the weight fragments are loop-invariant LDS reads: without this fence the compiler hoists all 544 registers' worth
    // of them out of the loop and spills
    asm volatile("" ::: "memory");
    // ---- layer 1 (transposed): [C1 x (D+3)] . [(D+3) x 64 samples] ----
    f32x16 acc[C1 / 32][2];
#pragma unroll
    for (int t = 0; t < C1 / 32; ++t)
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][st][e] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < KS0; ++ks) {
      half8 xh[2], xl[2];
      if (ks < D / 16) {
#pragma unroll
        for (int st = 0; st < 2; ++st) split8(raw[st][ks < D / 16 ? ks : 0][0], raw[st][ks < D / 16 ? ks : 0][1], xh[st], xl[st]);
      } else {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
#pragma unroll
          for (int q = 0; q < 8; ++q) { xh[st][q] = (_Float16)0.0f; xl[st][q] = (_Float16)0.0f; }
          if (lhi == 0) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
              _Float16 a, b;
              split1(__fsub_rn(qx[st][d], cx[d]), a, b);
              xh[st][d] = a; xl[st][d] = b;
            }
          }
        }
      }
#pragma unroll
      for (int t = 0; t < C1 / 32; ++t) {
        const half8 wh = *reinterpret_cast<const half8*>(W0h + (t * 32 + l31) * LD0 + ks * 16 + lhi * 8);
        const half8 wl = *reinterpret_cast<const half8*>(W0l + (t * 32 + l31) * LD0 + ks * 16 + lhi * 8);
#pragma unroll
        for (int st = 0; st < 2; ++st) acc[t][st] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl[st], acc[t][st], 0, 0, 0);
#pragma unroll
        for (int st = 0; st < 2; ++st) acc[t][st] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[st], acc[t][st], 0, 0, 0);
#pragma unroll
        for (int st = 0; st < 2; ++st) acc[t][st] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[st], acc[t][st], 0, 0, 0);
      }
    }
    // the feature rows are consumed: fetch the next neighbourhood's while layer 2 computes
    {
      int id_cur[2] = {id_nxt[0], id_nxt[1]};
      load_rows(g + stride, id_cur, raw, qx, cx);
      load_ids(g + 2 * stride, id_nxt);
    }
    half8 f1h[C1 / 16][2], f1l[C1 / 16][2];
#pragma unroll
    for (int t = 0; t < C1 / 32; ++t)
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const f32x16 y = bn_relu_t(acc[t][st], S0, T0, t * 32);
        half8 fh[2], fl[2];
        tile_to_fragments(y, lhi, fh, fl);
        f1h[2 * t][st] = fh[0]; f1h[2 * t + 1][st] = fh[1];
        f1l[2 * t][st] = fl[0]; f1l[2 * t + 1][st] = fl[1];
      }

    // ---- layer 2 (transposed): [C2 x C1] . [C1 x 64 samples], written out as rows of [samples, C2] ----
#pragma unroll
    for (int t = 0; t < C2 / 32; ++t) {
      f32x16 a2[2];
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int e = 0; e < 16; ++e) a2[st][e] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < C1 / 16; ++ks) {
        const half8 wh = *reinterpret_cast<const half8*>(W1h + (t * 32 + l31) * LD1 + ks * 16 + lhi * 8);
        const half8 wl = *reinterpret_cast<const half8*>(W1l + (t * 32 + l31) * LD1 + ks * 16 + lhi * 8);
#pragma unroll
        for (int st = 0; st < 2; ++st) a2[st] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, f1l[ks][st], a2[st], 0, 0, 0);
#pragma unroll
        for (int st = 0; st < 2; ++st) a2[st] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, f1h[ks][st], a2[st], 0, 0, 0);
#pragma unroll
        for (int st = 0; st < 2; ++st) a2[st] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, f1h[ks][st], a2[st], 0, 0, 0);
      }
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const f32x16 y = bn_relu_t(a2[st], S1, T1, t * 32);
        const int64_t o = ((int64_t)g * 64 + st * 32 + l31) * C2 + t * 32 + 4 * lhi;
        if (p.out_hi) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            half4 hi, lo;
#pragma unroll
            for (int r = 0; r < 4; ++r) { _Float16 a, b; split1(y[4 * q + r], a, b); hi[r] = a; lo[r] = b; }
            *reinterpret_cast<half4*>(p.out_hi + o + 8 * q) = hi;
            *reinterpret_cast<half4*>(p.out_lo + o + 8 * q) = lo;
          }
        }
        if (p.out) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(p.out + o + 8 * q) = make_float4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
        }
      }
    }
  }
}

}  // namespace

extern "C" int pfpp_sa_mlp3_fused(const float* xyz, const float* new_xyz, const int32_t* idx,
                                  const void* w0_hi, const void* w0_lo, const void* w1_hi, const void* w1_lo,
                                  const void* w2_hi, const void* w2_lo, const float* s0, const float* t0, const float* s1,
                                  const float* t1, const float* s2, const float* t2, float* out, int64_t F, int64_t N,
                                  int64_t S, int64_t ns, int64_t C1, int64_t C2, int64_t C3, pfpp_stream_t stream) {
  PFPP_REQUIRE(xyz && new_xyz && idx && w0_hi && w0_lo && w1_hi && w1_lo && w2_hi && w2_lo && s0 && t0 && s1 && t1 && s2 && t2 && out,
               "null pointer");
  PFPP_REQUIRE(F >= 0 && N > 0 && S > 0, "bad sizes");
  PFPP_SUPPORTED(ns == 32 && C1 == 64 && C2 == 64 && C3 == 128, "fused set-abstraction MLP: nsample 32 and widths 64/64/128 only");
  PFPP_REQUIRE(F * S < (1ll << 31), "too many neighbourhoods");
  PFPP_REQUIRE(pfpp::aligned16(w1_hi) && pfpp::aligned16(w1_lo) && pfpp::aligned16(w2_hi) && pfpp::aligned16(w2_lo), "planes must be 16-byte aligned");
  if (F == 0) return PFPP_OK;
  SaP p;
  p.xyz = xyz; p.ctr = new_xyz; p.idx = idx;
  p.w0h = (const _Float16*)w0_hi; p.w0l = (const _Float16*)w0_lo; p.w1h = (const _Float16*)w1_hi; p.w1l = (const _Float16*)w1_lo;
  p.w2h = (const _Float16*)w2_hi; p.w2l = (const _Float16*)w2_lo;
  p.s0 = s0; p.t0 = t0; p.s1 = s1; p.t1 = t1; p.s2 = s2; p.t2 = t2;
  p.out = out;
  p.N = (int)N; p.S = (int)S; p.G = (int)(F * S);
  const int64_t wgs_needed = (p.G + 3) / 4;
  const unsigned grid = (unsigned)(wgs_needed < 256 ? wgs_needed : 256);      // persistent: one 4-wave workgroup per CU
  hipLaunchKernelGGL((sa_mlp3_kernel<64, 64, 128>), dim3(grid), dim3(256), 0, pfpp::as_stream(stream), p);
  return pfpp::check_launch("pfpp_sa_mlp3_fused");
}

extern "C" int pfpp_sa_mlp2_fused(const float* feats, const float* xyz, const float* new_xyz, const int32_t* idx,
                                  const void* w0_hi, const void* w0_lo, const void* w1_hi, const void* w1_lo,
                                  const float* s0, const float* t0, const float* s1, const float* t1, float* out,
                                  int64_t F, int64_t N, int64_t S, int64_t ns, int64_t D, int64_t C1, int64_t C2,
                                  pfpp_stream_t stream) {
  return pfpp_sa_mlp2_fused_p(feats, xyz, new_xyz, idx, w0_hi, w0_lo, w1_hi, w1_lo, s0, t0, s1, t1, out, nullptr, F, N, S, ns, D, C1, C2, stream);
}

extern "C" int pfpp_sa_mlp2_fused_p(const float* feats, const float* xyz, const float* new_xyz, const int32_t* idx,
                                    const void* w0_hi, const void* w0_lo, const void* w1_hi, const void* w1_lo,
                                    const float* s0, const float* t0, const float* s1, const float* t1, float* out,
                                    const pfpp_planes* out_planes, int64_t F, int64_t N, int64_t S, int64_t ns, int64_t D, int64_t C1,
                                    int64_t C2, pfpp_stream_t stream) {
  PFPP_REQUIRE(feats && xyz && new_xyz && idx && w0_hi && w0_lo && w1_hi && w1_lo && s0 && t0 && s1 && t1 && (out || out_planes), "null pointer");
  PFPP_REQUIRE(pfpp_planes_ok(out_planes) && (!out_planes || out_planes->scale == 1.0f), "out_planes: 8-byte aligned hi / lo, scale 1");
  PFPP_REQUIRE(F >= 0 && N > 0 && S > 0, "bad sizes");
  PFPP_SUPPORTED(ns == 64 && D == 128 && C1 == 128 && C2 == 128, "fused set-abstraction layers 1+2: nsample 64, 128 features, widths 128/128 only");
  PFPP_REQUIRE(F * S < (1ll << 25), "too many neighbourhoods");
  PFPP_REQUIRE(pfpp::aligned16(feats) && pfpp::aligned16(w0_hi) && pfpp::aligned16(w0_lo) && pfpp::aligned16(w1_hi) && pfpp::aligned16(w1_lo) &&
               (!out || pfpp::aligned16(out)), "16-byte alignment");
  if (F == 0) return PFPP_OK;
  Sa2P p;
  p.out_hi = out_planes ? reinterpret_cast<_Float16*>(out_planes->hi) : nullptr;
  p.out_lo = out_planes ? reinterpret_cast<_Float16*>(out_planes->lo) : nullptr;
  p.feats = feats; p.xyz = xyz; p.ctr = new_xyz; p.idx = idx;
  p.w0h = (const _Float16*)w0_hi; p.w0l = (const _Float16*)w0_lo; p.w1h = (const _Float16*)w1_hi; p.w1l = (const _Float16*)w1_lo;
  p.s0 = s0; p.t0 = t0; p.s1 = s1; p.t1 = t1;
  p.out = out;
  p.N = (int)N; p.S = (int)S; p.G = (int)(F * S);
  constexpr int d = 128, c1 = 128, c2 = 128;
  constexpr size_t smem = (size_t)2 * (c1 * ((d / 16 + 1) * 16 + 8) + c2 * (c1 + 8)) * sizeof(_Float16) + (size_t)2 * (c1 + c2) * sizeof(float);
  auto kern = sa_mlp2_kernel<d, c1, c2>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  const int64_t wgs_needed = (p.G + 3) / 4;
  const unsigned grid = (unsigned)(wgs_needed < 256 ? wgs_needed : 256);      // persistent: one 4-wave workgroup per CU
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, pfpp::as_stream(stream), p);
  return pfpp::check_launch("pfpp_sa_mlp2_fused");
}

