// Train-mode set abstraction WITHOUT the HBM round trips of its activations (a5 under train_denoiser.py:33-35: the frozen encoder
// stays in .train(), so utils/pn2_utils.py:203-216 runs conv -> BatchNorm(BATCH statistics) -> ReLU x3 -> max over nsample).
//
// Batch statistics put a global barrier between the layers: layer k's normalisation needs sum(y_k), sum(y_k^2) over ALL rows
// (1.26 M at the benchmark shape).  The layer-wise form therefore writes every layer's [rows, C] fp32 pre-activation and reads
// it back (sa1: 1.3 GB, sa2: 2.6 GB per encoder pass).  Here a level is a sequence of STAGES of the persistent chain kernels of
// sa_fused.hip; stage k recomputes the chain from the level's input through the already finalised layers 1..k-1 (registers
// only, as in eval mode) and runs layer k in the GEMM's orientation (samples x channels: a lane owns one channel), where the
// column sums are in-lane additions:
//   sa1 (no input features, 3 -> 64 -> 64 -> 128, nsample 32): stage 1 / 2 write NOTHING but the sums; stage 3 writes the sums and
//       the per-neighbourhood max and min of y_3 (max_p relu(a y_p + b) = relu(a (a >= 0 ? max y : min y) + b)): 42.8 GFLOP
//       instead of 31.5, no activation traffic at all;
//   sa2 (128 features + 3 -> 128 -> 128 -> 256, nsample 64): stage 1 = gather + layer 1 -> sums; stage 2 = gather + layer 1
//       + BN/ReLU + layer 2 -> sums + the raw y_2 rows (layer 3's weights do not fit next to the others in LDS); stage 3 =
//       those rows -> BN/ReLU -> layer 3 with ITS weights resident in LDS -> sums + max / min: y_1 is never written or read,
//       y_2 once each way (1.3 GB instead of 2.6 GB).
// Arithmetic per layer = the layer-wise path's: split-f16 contraction (lo.hi, hi.lo, hi.hi per 16-deep step), y = acc + bias,
// activation relu(fma(y, a_mul, a_add)); sums of 16 / 32 values in fp32, then fp64 per lane, fp64 atomics into the
// [copies][2][C] buffer pfpp_bn_finalize reads.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pfpp.h"
#include "pfpp_common.h"
#include "sa_common.h"

// lab only (tools/lab/sa_ablate.sh): what the rows kernels' time is made of.  1 = no matrix instructions, 2 = no normalise / split of the
// operand rows, 4 = no statistics / stores, 8 = the row loads of the first half only
#ifndef SA_ABL
#define SA_ABL 0
#endif

namespace {

struct SaTP {
  const float* xyz; const float* ctr; const float* feats; const int32_t* idx;
  const _Float16* wh[3]; const _Float16* wl[3];
  const float* bias[3];
  const float* am[2]; const float* aa[2];
  double* stats; int copies;
  float* y_out; float* out_max; float* out_min;
  int N, S, G;
  const float* u; int D;       // first layer applied per point (pfpp_sa_train_args.u_in); feature count of the level
  _Float16* e_hi; _Float16* e_lo;      // eval form: output planes
  const int32_t* sched;                // pfpp_sa_pad_schedule of idx (64-neighbour levels) or null
};

// Padding-aware walk over the neighbourhoods of a 64-neighbour level.  The ball query (utils/pn2_utils.py:103-123) lists the points in range
// in ascending order and fills the rest of the nsample slots with the FIRST of them, so with at most 32 points in range rows 32..63 of the
// neighbourhood are 32 copies of its row 0 (the schedule tests exactly that — slots 32..63 all equal to slot 0 — not the count): their pre-activations equal row 0's in every layer, they add 32 y_0 and 32 y_0^2 to the batch
// sums and nothing to the max / min.  The rows kernels take such a neighbourhood as ONE half; its second half is neither computed nor
// (stage 2) written nor (stage 3) read.  On the benchmarked batch that is 48 % of level 2's neighbourhoods and 12 % of level 3's.
// sched [G + 1] (pfpp_sa_pad_schedule): the two-half neighbourhoods, then the one-half ones (ascending inside each class), and their
// split point in sched[G].  Wave k of S takes positions k, k + S, ... of the first class and goes on round-robin through the second where
// the first stopped, so every wave's work differs by at most one half from its neighbours' — a static, run-to-run identical partition.
struct PadWalk {
  const int32_t* order;
  int G, n2, S, i, ib;
  __device__ __forceinline__ PadWalk(const int32_t* sched, int G_, int k, int S_) : order(sched), G(G_), S(S_), i(k) {
    n2 = sched ? sched[G_] : G_;
    const int r = n2 % S_;
    ib = n2 + (k >= r ? k - r : k - r + S_);
    if (i >= n2) i = ib;
  }
  __device__ __forceinline__ bool two() const { return i < n2; }
  // live slots of the neighbourhood at the current position: slots cnt .. 63 repeat slot 0 (sched's third stretch, in schedule order so
  // that the entry does not wait for g()); 64 without a schedule.  Stage 2 stores and stage 3 loads rows [0, cnt) only: the raw rows
  // of the copies are never written and a reader takes row 0 in their place
  __device__ __forceinline__ int cnt() const { return (order && i < G) ? order[2 * G + 1 + i] : 64; }
  // neighbourhood at the current position (G = past the end: the callers' loaders clamp it)
  __device__ __forceinline__ int g() const { return i < G ? (order ? order[i] : i) : G; }
  __device__ __forceinline__ void next() {
    const int j = i + S;
    i = (i < n2 && j >= n2) ? ib : j;
  }
};

// train-mode BatchNorm + ReLU of a transposed tile (lane = sample): channel of register e is c0 + (e&3) + 8*(e>>2) + 4*lhi;
// y = acc + bias (the stored pre-activation of the layer-wise path), then relu(fma(y, a_mul, a_add))
__device__ __forceinline__ f32x16 bn_train_relu_t(const f32x16 acc, const float* B, const float* M, const float* A, int c0, int lhi) {
  f32x16 y;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 b4 = *reinterpret_cast<const float4*>(B + c0 + 8 * q + 4 * lhi);
    const float4 m4 = *reinterpret_cast<const float4*>(M + c0 + 8 * q + 4 * lhi);
    const float4 a4 = *reinterpret_cast<const float4*>(A + c0 + 8 * q + 4 * lhi);
    const float bv[4] = {b4.x, b4.y, b4.z, b4.w}, mv[4] = {m4.x, m4.y, m4.z, m4.w}, av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = __builtin_fmaf(acc[4 * q + r] + bv[r], mv[r], av[r]);
      y[4 * q + r] = v > 0.0f ? v : 0.0f;
    }
  }
  return y;
}

__device__ __forceinline__ void flush_stats(double* stats, int copies, int C, int c, int lhi, double s, double q) {
  const double a = s + __shfl_xor(s, 32), b = q + __shfl_xor(q, 32);
  if (lhi == 0) {
    double* st = stats + (size_t)((blockIdx.x * 4 + (threadIdx.x >> 6)) % copies) * 2 * C;
    unsafeAtomicAdd(st + c, a);
    unsafeAtomicAdd(st + C + c, b);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// level without input features (sa1): STAGE = index of the layer whose batch statistics this launch produces
template <int C1, int C2, int C3, int STAGE>
__global__ __launch_bounds__(256, 1) void sa1_train_kernel(const SaTP p) {
  constexpr int CS = STAGE == 1 ? C1 : (STAGE == 2 ? C2 : C3);
  constexpr int NT1 = C1 / 32, NT2 = STAGE >= 2 ? C2 / 32 : 1, NT3 = STAGE >= 3 ? C3 / 32 : 1;
  __shared__ __align__(16) float B0[C1], M0[C1], A0[C1], B1[C2], M1[C2], A1[C2];
  // stage 3 holds every weight fragment in registers (208) and has none left for the fp64 running sums: they live in LDS, one
  // private slot per lane (no conflicts: consecutive lanes, consecutive 8-byte words)
  constexpr bool LDS_SUMS = STAGE == 3;
  __shared__ double SUMS[LDS_SUMS ? 2 * (CS / 32) : 1][LDS_SUMS ? 256 : 1];
  const int tid = threadIdx.x;
  if (LDS_SUMS)
    for (int i = 0; i < 2 * (CS / 32); ++i) SUMS[i][tid] = 0.0;
  if (STAGE >= 2)
    for (int i = tid; i < C1; i += 256) { B0[i] = p.bias[0][i]; M0[i] = p.am[0][i]; A0[i] = p.aa[0][i]; }
  if (STAGE >= 3)
    for (int i = tid; i < C2; i += 256) { B1[i] = p.bias[1][i]; M1[i] = p.am[1][i]; A1[i] = p.aa[1][i]; }
  __syncthreads();

  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;

  half8 w0h[NT1], w0l[NT1];
  half8 w1h[NT2][C1 / 16], w1l[NT2][C1 / 16];
  half8 w2h[NT3][C2 / 16], w2l[NT3][C2 / 16];
#pragma unroll
  for (int t = 0; t < NT1; ++t) {
#pragma unroll
    for (int q = 0; q < 8; ++q) { w0h[t][q] = (_Float16)0.0f; w0l[t][q] = (_Float16)0.0f; }
    if (lhi == 0) {      // K = 3 (+ a zero) of 16: only the first 8-half group is populated
      w0h[t] = *reinterpret_cast<const half8*>(p.wh[0] + (t * 32 + l31) * 8);
      w0l[t] = *reinterpret_cast<const half8*>(p.wl[0] + (t * 32 + l31) * 8);
    }
  }
  if (STAGE >= 2) {
#pragma unroll
    for (int t = 0; t < NT2; ++t)
#pragma unroll
      for (int ks = 0; ks < C1 / 16; ++ks) {
        w1h[t][ks] = *reinterpret_cast<const half8*>(p.wh[1] + (t * 32 + l31) * C1 + ks * 16 + lhi * 8);
        w1l[t][ks] = *reinterpret_cast<const half8*>(p.wl[1] + (t * 32 + l31) * C1 + ks * 16 + lhi * 8);
      }
  }
  if (STAGE >= 3) {
#pragma unroll
    for (int n = 0; n < NT3; ++n)
#pragma unroll
      for (int ks = 0; ks < C2 / 16; ++ks) {
        w2h[n][ks] = *reinterpret_cast<const half8*>(p.wh[2] + (n * 32 + l31) * C2 + ks * 16 + lhi * 8);
        w2l[n][ks] = *reinterpret_cast<const half8*>(p.wl[2] + (n * 32 + l31) * C2 + ks * 16 + lhi * 8);
      }
  }
  float bs[CS / 32];
  double ss[CS / 32], sq[CS / 32];
#pragma unroll
  for (int n = 0; n < CS / 32; ++n) { bs[n] = p.bias[STAGE - 1][n * 32 + l31]; ss[n] = 0.0; sq[n] = 0.0; }

  // the statistics layer, GEMM orientation: lane = channel n*32 + l31, register e = sample (e&3) + 8*(e>>2) + 4*lhi
  auto stat = [&](const f32x16& acc, int n, int g) {
    float s = 0.0f, q = 0.0f, mx = -__builtin_huge_valf(), mn = __builtin_huge_valf();
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float y = acc[e] + bs[n];
      s += y;
      q = __builtin_fmaf(y, y, q);
      if (STAGE == 3) { mx = fmaxf(mx, y); mn = fminf(mn, y); }
    }
    if (LDS_SUMS) {
      SUMS[2 * n][tid] += (double)s;
      SUMS[2 * n + 1][tid] += (double)q;
    } else {
      ss[n] += (double)s;
      sq[n] += (double)q;
    }
    if (STAGE == 3) {
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      mn = fminf(mn, __shfl_xor(mn, 32));
      if (lhi == 0) {
        p.out_max[(int64_t)g * C3 + n * 32 + l31] = mx;
        p.out_min[(int64_t)g * C3 + n * 32 + l31] = mn;
      }
    }
  };

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

    if (STAGE == 1) {
#pragma unroll
      for (int n = 0; n < NT1; ++n) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, w0h[n], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, w0l[n], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, w0h[n], acc, 0, 0, 0);
        stat(acc, n, g);
      }
    } else {
      // ---- layer 1 (transposed, normalised with its finalised statistics) -> operand fragments ----
      half8 f1h[C1 / 16], f1l[C1 / 16];
#pragma unroll
      for (int t = 0; t < NT1; ++t) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0h[t], xl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0l[t], xh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0h[t], xh, acc, 0, 0, 0);
        const f32x16 y = bn_train_relu_t(acc, B0, M0, A0, t * 32, lhi);
        half8 fh[2], fl[2];
        tile_to_fragments(y, lhi, fh, fl);
        f1h[2 * t] = fh[0]; f1h[2 * t + 1] = fh[1];
        f1l[2 * t] = fl[0]; f1l[2 * t + 1] = fl[1];
      }
      if (STAGE == 2) {
#pragma unroll
        for (int n = 0; n < NT2; ++n) {
          f32x16 acc;
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
          for (int ks = 0; ks < C1 / 16; ++ks) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1l[ks], w1h[n][ks], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1h[ks], w1l[n][ks], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1h[ks], w1h[n][ks], acc, 0, 0, 0);
          }
          stat(acc, n, g);
        }
      } else {
        // scheduling fence: without it the compiler overlaps the layers' accumulators and spills 41 of the 512 registers
        asm volatile("" ::: "memory");
        half8 f2h[C2 / 16], f2l[C2 / 16];
#pragma unroll
        for (int t = 0; t < NT2; ++t) {
          f32x16 acc;
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
          for (int ks = 0; ks < C1 / 16; ++ks) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h[t][ks], f1l[ks], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1l[t][ks], f1h[ks], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h[t][ks], f1h[ks], acc, 0, 0, 0);
          }
          const f32x16 y = bn_train_relu_t(acc, B1, M1, A1, t * 32, lhi);
          half8 fh[2], fl[2];
          tile_to_fragments(y, lhi, fh, fl);
          f2h[2 * t] = fh[0]; f2h[2 * t + 1] = fh[1];
          f2l[2 * t] = fl[0]; f2l[2 * t + 1] = fl[1];
        }
#pragma unroll
        for (int n = 0; n < NT3; ++n) {
          f32x16 acc;
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
          for (int ks = 0; ks < C2 / 16; ++ks) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f2l[ks], w2h[n][ks], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f2h[ks], w2l[n][ks], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f2h[ks], w2h[n][ks], acc, 0, 0, 0);
          }
          stat(acc, n, g);
        }
      }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) { q_cur[d] = q_nxt[d]; c_cur[d] = c_nxt[d]; }
  }
#pragma unroll
  for (int n = 0; n < CS / 32; ++n)
    flush_stats(p.stats, p.copies, CS, n * 32 + l31, lhi, LDS_SUMS ? SUMS[LDS_SUMS ? 2 * n : 0][LDS_SUMS ? tid : 0] : ss[n],
                LDS_SUMS ? SUMS[LDS_SUMS ? 2 * n + 1 : 0][LDS_SUMS ? tid : 0] : sq[n]);
}

// ---------------------------------------------------------------------------------------------------------------------
// level with D input features (sa2), nsample 64.  STAGE 1: gather + layer 1 -> sums.  STAGE 2: gather + layer 1 + BN/ReLU +
// layer 2 -> sums + raw y_2 rows [G*64, C2].
template <int D, int C1, int C2, int STAGE>
__global__ __launch_bounds__(256, 1) void sa2_train_kernel(const SaTP p) {
  constexpr int KS0 = D / 16 + 1;
  constexpr int KP0 = D + 8;
  constexpr int LD0 = KS0 * 16 + 8, LD1 = C1 + 8;
  constexpr int CS = STAGE == 1 ? C1 : C2;
  extern __shared__ __align__(16) unsigned char sa2t_smem[];
  _Float16* W0h = reinterpret_cast<_Float16*>(sa2t_smem);
  _Float16* W0l = W0h + C1 * LD0;
  _Float16* W1h = W0l + C1 * LD0;
  _Float16* W1l = W1h + C2 * LD1;
  float* B0 = reinterpret_cast<float*>(W1l + C2 * LD1);
  float* M0 = B0 + C1;
  float* A0 = M0 + C1;

  const int tid = threadIdx.x;
  for (int i = tid; i < C1 * (LD0 / 8); i += 256) {
    const int r = i / (LD0 / 8), c8 = i - r * (LD0 / 8);
    uint4 vh = make_uint4(0, 0, 0, 0), vl = vh;
    if (c8 * 8 < KP0) {
      vh = *reinterpret_cast<const uint4*>(p.wh[0] + (size_t)r * KP0 + c8 * 8);
      vl = *reinterpret_cast<const uint4*>(p.wl[0] + (size_t)r * KP0 + c8 * 8);
    }
    *reinterpret_cast<uint4*>(W0h + r * LD0 + c8 * 8) = vh;
    *reinterpret_cast<uint4*>(W0l + r * LD0 + c8 * 8) = vl;
  }
  if (STAGE == 2) {
    for (int i = tid; i < C2 * (C1 / 8); i += 256) {
      const int r = i / (C1 / 8), c8 = i - r * (C1 / 8);
      *reinterpret_cast<uint4*>(W1h + r * LD1 + c8 * 8) = *reinterpret_cast<const uint4*>(p.wh[1] + (size_t)r * C1 + c8 * 8);
      *reinterpret_cast<uint4*>(W1l + r * LD1 + c8 * 8) = *reinterpret_cast<const uint4*>(p.wl[1] + (size_t)r * C1 + c8 * 8);
    }
    for (int i = tid; i < C1; i += 256) { B0[i] = p.bias[0][i]; M0[i] = p.am[0][i]; A0[i] = p.aa[0][i]; }
  }
  __syncthreads();

  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;

  float bs[CS / 32];
  double ss[CS / 32], sq[CS / 32];
#pragma unroll
  for (int n = 0; n < CS / 32; ++n) { bs[n] = p.bias[STAGE - 1][n * 32 + l31]; ss[n] = 0.0; sq[n] = 0.0; }

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
      id[st] = v < p.N ? v : p.N - 1;
    }
  };
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
    asm volatile("" ::: "memory");      // keeps the loop-invariant LDS weight reads inside the loop (sa_fused.hip)
    // ---- layer 1: STAGE 1 in the GEMM's orientation (lane = channel: in-lane sums), STAGE 2 transposed (lane = sample) ----
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
        if (STAGE == 1) {
#pragma unroll
          for (int st = 0; st < 2; ++st) acc[t][st] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl[st], wh, acc[t][st], 0, 0, 0);
#pragma unroll
          for (int st = 0; st < 2; ++st) acc[t][st] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[st], wl, acc[t][st], 0, 0, 0);
#pragma unroll
          for (int st = 0; st < 2; ++st) acc[t][st] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[st], wh, acc[t][st], 0, 0, 0);
        } else {
#pragma unroll
          for (int st = 0; st < 2; ++st) acc[t][st] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl[st], acc[t][st], 0, 0, 0);
#pragma unroll
          for (int st = 0; st < 2; ++st) acc[t][st] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[st], acc[t][st], 0, 0, 0);
#pragma unroll
          for (int st = 0; st < 2; ++st) acc[t][st] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[st], acc[t][st], 0, 0, 0);
        }
      }
    }
    // the feature rows are consumed: fetch the next neighbourhood's
    {
      int id_cur[2] = {id_nxt[0], id_nxt[1]};
      load_rows(g + stride, id_cur, raw, qx, cx);
      load_ids(g + 2 * stride, id_nxt);
    }
    if (STAGE == 1) {
#pragma unroll
      for (int n = 0; n < C1 / 32; ++n) {
        float s = 0.0f, q = 0.0f;
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float y = acc[n][st][e] + bs[n];
            s += y;
            q = __builtin_fmaf(y, y, q);
          }
        ss[n] += (double)s;
        sq[n] += (double)q;
      }
    } else {
      half8 f1h[C1 / 16][2], f1l[C1 / 16][2];
#pragma unroll
      for (int t = 0; t < C1 / 32; ++t)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          const f32x16 y = bn_train_relu_t(acc[t][st], B0, M0, A0, t * 32, lhi);
          half8 fh[2], fl[2];
          tile_to_fragments(y, lhi, fh, fl);
          f1h[2 * t][st] = fh[0]; f1h[2 * t + 1][st] = fh[1];
          f1l[2 * t][st] = fl[0]; f1l[2 * t + 1][st] = fl[1];
        }
      // ---- layer 2 in the GEMM's orientation: lane = channel n*32 + l31, register e = sample row; raw rows out + sums ----
#pragma unroll
      for (int n = 0; n < C2 / 32; ++n) {
        f32x16 a2[2];
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int e = 0; e < 16; ++e) a2[st][e] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < C1 / 16; ++ks) {
          const half8 wh = *reinterpret_cast<const half8*>(W1h + (n * 32 + l31) * LD1 + ks * 16 + lhi * 8);
          const half8 wl = *reinterpret_cast<const half8*>(W1l + (n * 32 + l31) * LD1 + ks * 16 + lhi * 8);
#pragma unroll
          for (int st = 0; st < 2; ++st) a2[st] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1l[ks][st], wh, a2[st], 0, 0, 0);
#pragma unroll
          for (int st = 0; st < 2; ++st) a2[st] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1h[ks][st], wl, a2[st], 0, 0, 0);
#pragma unroll
          for (int st = 0; st < 2; ++st) a2[st] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1h[ks][st], wh, a2[st], 0, 0, 0);
        }
        float s = 0.0f, q = 0.0f;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          float* orow = p.y_out + ((int64_t)g * 64 + st * 32 + 4 * lhi) * C2 + n * 32 + l31;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float y = a2[st][e] + bs[n];
            s += y;
            q = __builtin_fmaf(y, y, q);
            orow[(int64_t)((e & 3) + 8 * (e >> 2)) * C2] = y;       // 128 contiguous bytes per half wave
          }
        }
        ss[n] += (double)s;
        sq[n] += (double)q;
      }
    }
  }
#pragma unroll
  for (int n = 0; n < CS / 32; ++n) flush_stats(p.stats, p.copies, CS, n * 32 + l31, lhi, ss[n], sq[n]);
}

// ---------------------------------------------------------------------------------------------------------------------
// last layer of the level with input features (sa2 stage 3): the raw second-layer rows y_2 [G*64, K] written by stage 2 ->
// relu(fma(y_2, a_mul, a_add)) -> third convolution -> sums of y_3 + per-neighbourhood max / min.  The layer-wise form of this is
// the fp32-A plane GEMM (one workgroup per 128 x 128 tile: 2 column tiles re-read and re-convert every A element, a DMA ring
// prologue + epilogue per 4-K-tile contraction: 505 us).  Here the 256 x 128 weight planes stay in LDS for the lifetime of one
// persistent workgroup per CU; a wave owns a neighbourhood's 64 rows: every lane fetches the 32-byte runs of ITS row, normalises
// and splits them once, and keeps the fragments in registers for all 8 column tiles.
template <int K, int N>
__global__ __launch_bounds__(256, 1) void sa_rows_train_kernel(const SaTP p) {
  constexpr int LD = K + 8;
  extern __shared__ __align__(16) unsigned char sar_smem[];
  _Float16* Wh = reinterpret_cast<_Float16*>(sar_smem);
  _Float16* Wl = Wh + N * LD;
  float* M = reinterpret_cast<float*>(Wl + N * LD);
  float* A = M + K;
  const int tid = threadIdx.x;
  for (int i = tid; i < N * (K / 8); i += 256) {
    const int r = i / (K / 8), c8 = i - r * (K / 8);
    *reinterpret_cast<uint4*>(Wh + r * LD + c8 * 8) = *reinterpret_cast<const uint4*>(p.wh[2] + (size_t)r * K + c8 * 8);
    *reinterpret_cast<uint4*>(Wl + r * LD + c8 * 8) = *reinterpret_cast<const uint4*>(p.wl[2] + (size_t)r * K + c8 * 8);
  }
  for (int i = tid; i < K; i += 256) { M[i] = p.am[1][i]; A[i] = p.aa[1][i]; }
  __syncthreads();

  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  float bs[N / 32];
  double ss[N / 32], sq[N / 32];
#pragma unroll
  for (int n = 0; n < N / 32; ++n) { bs[n] = p.bias[2][n * 32 + l31]; ss[n] = 0.0; sq[n] = 0.0; }

  const int stride = gridDim.x * 4;
  const int g0 = blockIdx.x * 4 + wave;
  const float* rows = p.y_out;
  auto load_rows = [&](int g, float4 (&raw)[2][K / 16][2]) {
    const int gc = g < p.G ? g : p.G - 1;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const float* row = rows + ((int64_t)gc * 64 + st * 32 + l31) * K + lhi * 8;
#pragma unroll
      for (int ks = 0; ks < K / 16; ++ks) {
        raw[st][ks][0] = *reinterpret_cast<const float4*>(row + ks * 16);
        raw[st][ks][1] = *reinterpret_cast<const float4*>(row + ks * 16 + 4);
      }
    }
  };
  float4 raw[2][K / 16][2];
  load_rows(g0, raw);

  for (int g = g0; g < p.G; g += stride) {
    asm volatile("" ::: "memory");
    half8 fh[K / 16][2], fl[K / 16][2];
#pragma unroll
    for (int ks = 0; ks < K / 16; ++ks) {
      const float4 m0 = *reinterpret_cast<const float4*>(M + ks * 16 + lhi * 8), m1 = *reinterpret_cast<const float4*>(M + ks * 16 + lhi * 8 + 4);
      const float4 a0 = *reinterpret_cast<const float4*>(A + ks * 16 + lhi * 8), a1 = *reinterpret_cast<const float4*>(A + ks * 16 + lhi * 8 + 4);
      const float mv[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w}, av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const float4 r0 = raw[st][ks][0], r1 = raw[st][ks][1];
        const float x[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float v = fmaxf(__builtin_fmaf(x[q], mv[q], av[q]), 0.0f);     // relu(batch-norm(y_2)), as the GEMM's A loader
          _Float16 h, l;
          split1(v, h, l);
          fh[ks][st][q] = h; fl[ks][st][q] = l;
        }
      }
    }
    load_rows(g + stride, raw);          // the next neighbourhood's rows are in flight during this one's contraction
    // the column-tile loop stays rolled: unrolled, the compiler overlaps the tiles' accumulators and weight fragments and spills
    // 150 registers
#pragma unroll 1
    for (int n = 0; n < N / 32; ++n) {
      f32x16 acc[2];
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[st][e] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < K / 16; ++ks) {
        const half8 wh = *reinterpret_cast<const half8*>(Wh + (n * 32 + l31) * LD + ks * 16 + lhi * 8);
        const half8 wl = *reinterpret_cast<const half8*>(Wl + (n * 32 + l31) * LD + ks * 16 + lhi * 8);
#pragma unroll
        for (int st = 0; st < 2; ++st) acc[st] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[ks][st], wh, acc[st], 0, 0, 0);
#pragma unroll
        for (int st = 0; st < 2; ++st) acc[st] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[ks][st], wl, acc[st], 0, 0, 0);
#pragma unroll
        for (int st = 0; st < 2; ++st) acc[st] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[ks][st], wh, acc[st], 0, 0, 0);
      }
      float s = 0.0f, q = 0.0f, mx = -__builtin_huge_valf(), mn = __builtin_huge_valf();
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float y = acc[st][e] + bs[n];
          s += y;
          q = __builtin_fmaf(y, y, q);
          mx = fmaxf(mx, y); mn = fminf(mn, y);
        }
      ss[n] += (double)s;
      sq[n] += (double)q;
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      mn = fminf(mn, __shfl_xor(mn, 32));
      if (lhi == 0) {
        p.out_max[(int64_t)g * N + n * 32 + l31] = mx;
        p.out_min[(int64_t)g * N + n * 32 + l31] = mn;
      }
    }
  }
#pragma unroll
  for (int n = 0; n < N / 32; ++n) flush_stats(p.stats, p.copies, N, n * 32 + l31, lhi, ss[n], sq[n]);
}

// The same stage with EIGHT waves per workgroup, two per SIMD: a wave takes half a neighbourhood (32 rows) at a time, so its raw rows,
// operand fragments and one accumulator tile fit in 256 registers and the split / statistics arithmetic of one wave runs under the
// matrix instructions of the other wave of its SIMD (with one wave per SIMD they alternate: 37 % of the matrix peak).  The weight
// fragments are read from LDS once per 32 rows instead of once per 64; the neighbourhood's max / min are carried across its two halves.
template <int K, int N>
__global__ __launch_bounds__(512, 1) void sa_rows8_train_kernel(const SaTP p) {
  constexpr int LD = K + 8;
  constexpr int NWAVE = 8;
  extern __shared__ __align__(16) unsigned char sar_smem[];
  _Float16* Wh = reinterpret_cast<_Float16*>(sar_smem);
  _Float16* Wl = Wh + N * LD;
  float* M = reinterpret_cast<float*>(Wl + N * LD);
  float* A = M + K;
  const int tid = threadIdx.x;
  for (int i = tid; i < N * (K / 8); i += 64 * NWAVE) {
    const int r = i / (K / 8), c8 = i - r * (K / 8);
    *reinterpret_cast<uint4*>(Wh + r * LD + c8 * 8) = *reinterpret_cast<const uint4*>(p.wh[2] + (size_t)r * K + c8 * 8);
    *reinterpret_cast<uint4*>(Wl + r * LD + c8 * 8) = *reinterpret_cast<const uint4*>(p.wl[2] + (size_t)r * K + c8 * 8);
  }
  for (int i = tid; i < K; i += 64 * NWAVE) { M[i] = p.am[1][i]; A[i] = p.aa[1][i]; }
  __syncthreads();

  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  float bs[N / 32];
  double ss[N / 32], sq[N / 32];
#pragma unroll
  for (int n = 0; n < N / 32; ++n) { bs[n] = p.bias[2][n * 32 + l31]; ss[n] = 0.0; sq[n] = 0.0; }

  const int stride = gridDim.x * NWAVE;
  const int g0 = blockIdx.x * NWAVE + wave;
  const float* rows = p.y_out;
  // half-neighbourhood h (0, 1) of neighbourhood g: rows g * 64 + 32 h + l31
  auto load_rows = [&](int g, int h, int live, float4 (&raw)[K / 16][2]) {
    const int gc = g < p.G ? g : p.G - 1;
    const int j = h * 32 + l31;
    const float* row = rows + ((int64_t)gc * 64 + (j < live ? j : 0)) * K + lhi * 8;
#pragma unroll
    for (int ks = 0; ks < K / 16; ++ks) {
      raw[ks][0] = *reinterpret_cast<const float4*>(row + ks * 16);
      raw[ks][1] = *reinterpret_cast<const float4*>(row + ks * 16 + 4);
    }
  };
  float4 raw[K / 16][2];
  // the walk runs two neighbourhoods ahead of the rows (the schedule entry of g2 is in flight while g is multiplied)
  PadWalk wk(p.sched, p.G, g0, stride);
  int g = wk.g(), c = wk.cnt(); wk.next();
  int g1 = wk.g(), c1 = wk.cnt(); wk.next();
  int g2 = wk.g(), c2 = wk.cnt();
  load_rows(g, 0, c, raw);
  float mx[N / 32], mn[N / 32];

  for (; g < p.G; g = g1, c = c1, g1 = g2, c1 = c2, wk.next(), g2 = wk.g(), c2 = wk.cnt()) {
    const int nh = c > 32 ? 2 : 1;
#pragma unroll 1
    for (int h = 0; h < nh; ++h) {
      asm volatile("" ::: "memory");
      half8 fh[K / 16], fl[K / 16];
#pragma unroll
      for (int ks = 0; ks < K / 16; ++ks) {
        const float4 m0 = *reinterpret_cast<const float4*>(M + ks * 16 + lhi * 8), m1 = *reinterpret_cast<const float4*>(M + ks * 16 + lhi * 8 + 4);
        const float4 a0 = *reinterpret_cast<const float4*>(A + ks * 16 + lhi * 8), a1 = *reinterpret_cast<const float4*>(A + ks * 16 + lhi * 8 + 4);
        const float mv[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w}, av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const float4 r0 = raw[ks][0], r1 = raw[ks][1];
        const float x[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if (SA_ABL & 2) { fh[ks][q] = (_Float16)x[q]; fl[ks][q] = (_Float16)mv[q]; continue; }
          const float v = fmaxf(__builtin_fmaf(x[q], mv[q], av[q]), 0.0f);     // relu(batch-norm(y_2)), as the GEMM's A loader
          _Float16 hh, ll;
          split1(v, hh, ll);
          fh[ks][q] = hh; fl[ks][q] = ll;
        }
      }
      // the next half's rows are in flight during this one's contraction
      if (!(SA_ABL & 8)) {
        if (h + 1 < nh) load_rows(g, 1, c, raw);
        else load_rows(g1, 0, c1, raw);
      }
#pragma unroll 1
      for (int n = 0; n < N / 32; ++n) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < K / 16; ++ks) {
          const half8 wh = *reinterpret_cast<const half8*>(Wh + (n * 32 + l31) * LD + ks * 16 + lhi * 8);
          const half8 wl = *reinterpret_cast<const half8*>(Wl + (n * 32 + l31) * LD + ks * 16 + lhi * 8);
          if (SA_ABL & 1) { asm volatile("" :: "v"(fl[ks]), "v"(fh[ks]), "v"(wh), "v"(wl)); continue; }
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[ks], wh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[ks], wl, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[ks], wh, acc, 0, 0, 0);
        }
        if (SA_ABL & 4) { asm volatile("" :: "v"(acc)); continue; }
        float s = 0.0f, q = 0.0f, hi = -__builtin_huge_valf(), lo = __builtin_huge_valf();
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float y = acc[e] + bs[n];
          s += y;
          q = __builtin_fmaf(y, y, q);
          hi = fmaxf(hi, y); lo = fminf(lo, y);
        }
        ss[n] += (double)s;
        sq[n] += (double)q;
        if (nh == 1) {      // rows 32..63 are copies of row 0 (register 0 of the lower lane half)
          const double y0 = lhi == 0 ? (double)(acc[0] + bs[n]) : 0.0;
          ss[n] += 32.0 * y0;
          sq[n] += 32.0 * y0 * y0;
        }
        if (h + 1 < nh) { mx[n] = hi; mn[n] = lo; }
        else {
          if (h == 1) { hi = fmaxf(hi, mx[n]); lo = fminf(lo, mn[n]); }
          hi = fmaxf(hi, __shfl_xor(hi, 32));
          lo = fminf(lo, __shfl_xor(lo, 32));
          if (lhi == 0) {
            p.out_max[(int64_t)g * N + n * 32 + l31] = hi;
            p.out_min[(int64_t)g * N + n * 32 + l31] = lo;
          }
        }
      }
    }
  }
#pragma unroll
  for (int n = 0; n < N / 32; ++n) flush_stats(p.stats, p.copies, N, n * 32 + l31, lhi, ss[n], sq[n]);
}

// ---------------------------------------------------------------------------------------------------------------------
// Wide level (sa3: 256 features + 3 -> 256 -> 256 -> 512, nsample 64, 246 K rows): no two of its weight matrices fit in LDS together,
// so it stays one launch per layer with the [rows, 256] pre-activations between them — but each layer as a ROWS kernel like the one
// above instead of a tiled GEMM: a workgroup keeps a 128-column slice of the layer's weight planes (135-143 KB) in LDS for its
// lifetime; a wave owns 32 rows at a time: every lane fetches the 32-byte runs of ITS row (gathered by ball-query index for layer 1,
// contiguous raw rows of the previous layer otherwise), applies relu(fma(y, a_mul, a_add)) of the previous layer's finalised
// statistics, splits once and keeps the 16 (17) operand fragments in registers for the slice's four column tiles.  The column
// slices of a row group run in workgroups of the same XCD (ids a multiple of 8 apart), so the rows are fetched from HBM once.
//   LAYER 1: gather + conv -> raw rows + sums;  LAYER 2: rows -> conv -> raw rows + sums;  LAYER 3: rows -> conv -> sums + max / min.
// UG (LAYER 2 only): the layer's input rows are not read back but GATHERED from the per-point table U = conv1([feats | xyz]) + b1
// (p.u, [F * N, K]): conv1 is linear, so its value on a grouped row is U[point] - W1_xyz . centroid, and relu(bn(.)) of it is
// relu(fma(U[point], a_mul, a_add - a_mul * (W1_xyz . centroid))) — the same fma as the rows path with a per-NEIGHBOURHOOD add
// vector, which every wave keeps in its own LDS slot.  The first layer's grouped convolution is never computed.
// EVAL (with UG): the eval-mode form of the same launch (pfpp_sa_mlp2_table_p) — the affines are the FOLDED BatchNorm scale / shift of
// layers 1 and 2, no statistics; the output is relu(fma(y2, s1, t1)) as split-f16 planes for the third layer's plane GEMM.
template <int K, int LAYER, bool UG = false, bool EVAL = false>
__global__ __launch_bounds__(256, 1) void sa_wide_train_kernel(const SaTP p, const float* __restrict__ y_in, int n_total) {
  static_assert(!UG || LAYER == 2, "the per-point table feeds the second layer");
  static_assert(!EVAL || UG, "the eval form exists for the table-fed second layer only");
  constexpr bool GATHER = LAYER == 1;
  constexpr int KS = K / 16 + (GATHER ? 1 : 0);          // 16-deep steps: the features, then [dx dy dz 0 ...]
  constexpr int KP = GATHER ? K + 8 : K;                 // row length of the weight planes in memory
  constexpr int LD = KS * 16 + 8;                        // LDS row stride in halfs
  constexpr int NSL = 128;                               // columns per workgroup
  extern __shared__ __align__(16) unsigned char saw_smem[];
  _Float16* Wh = reinterpret_cast<_Float16*>(saw_smem);
  _Float16* Wl = Wh + NSL * LD;
  float* M = reinterpret_cast<float*>(Wl + NSL * LD);
  float* A = M + K;
  const int tid = threadIdx.x;
  const int n_slices = n_total / NSL;
  const int row_wgs = gridDim.x / n_slices;               // workgroups per column slice
  const int slice = blockIdx.x / row_wgs, wg_row = blockIdx.x - slice * row_wgs;
  const _Float16* wh_g = p.wh[LAYER - 1] + (size_t)slice * NSL * KP;
  const _Float16* wl_g = p.wl[LAYER - 1] + (size_t)slice * NSL * KP;
  for (int i = tid; i < NSL * (LD / 8); i += 256) {
    const int r = i / (LD / 8), c8 = i - r * (LD / 8);
    uint4 vh = make_uint4(0, 0, 0, 0), vl = vh;
    if (c8 * 8 < KP) {
      vh = *reinterpret_cast<const uint4*>(wh_g + (size_t)r * KP + c8 * 8);
      vl = *reinterpret_cast<const uint4*>(wl_g + (size_t)r * KP + c8 * 8);
    }
    *reinterpret_cast<uint4*>(Wh + r * LD + c8 * 8) = vh;
    *reinterpret_cast<uint4*>(Wl + r * LD + c8 * 8) = vl;
  }
  if (!GATHER)
    for (int i = tid; i < K; i += 256) { M[i] = p.am[LAYER - 2][i]; A[i] = p.aa[LAYER - 2][i]; }
  if (UG) {
    const int kp1 = p.D + 8;
    for (int i = tid; i < 3 * K; i += 256) {
      const int d = i / K, ch = i - d * K;
      const size_t o = (size_t)ch * kp1 + p.D + d;
      (A + 5 * K)[i] = (float)p.wh[0][o] + (float)p.wl[0][o];
    }
  }
  __syncthreads();

  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int col0 = slice * NSL;
  // UG: this lane's channels of the add vector (K / 64 of them): multiplier, add and the three xyz weights of the first layer
  constexpr int CPL = K / 64;
  float* AS = UG ? A + K + wave * K : A;             // the wave's add vector of the current neighbourhood
  float* WX = A + 5 * K;                             // UG: [3][K] xyz weights of the first layer (shared, read-only after the barrier)
  float bs[4];
  double ss[4], sq[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) { bs[n] = p.bias[LAYER - 1] ? p.bias[LAYER - 1][col0 + n * 32 + l31] : 0.0f; ss[n] = 0.0; sq[n] = 0.0; }
  float es[EVAL ? 4 : 1], et[EVAL ? 4 : 1];
  if (EVAL) {
#pragma unroll
    for (int n = 0; n < 4; ++n) { es[n] = p.am[1][col0 + n * 32 + l31]; et[n] = p.aa[1][col0 + n * 32 + l31]; }
  }

  // a wave walks neighbourhoods (64 rows) in two halves of 32 rows
  const int stride = row_wgs * 4;
  const int g0 = wg_row * 4 + wave;
  auto load_half = [&](int g, int half, int live, float4 (&raw)[K / 16][2], float (&q)[3], float (&c)[3]) {
    const int gc = g < p.G ? g : p.G - 1;
    if (GATHER) {
      const int f = gc / p.S;
      int id = p.idx[(int64_t)gc * 64 + half * 32 + l31];
      id = id < p.N ? id : p.N - 1;
      const float* row = p.feats + ((int64_t)f * p.N + id) * K + lhi * 8;
#pragma unroll
      for (int ks = 0; ks < K / 16; ++ks) {
        raw[ks][0] = *reinterpret_cast<const float4*>(row + ks * 16);
        raw[ks][1] = *reinterpret_cast<const float4*>(row + ks * 16 + 4);
      }
      const float* q3 = p.xyz + ((int64_t)f * p.N + id) * 3;
      const float* c3 = p.ctr + (int64_t)gc * 3;
#pragma unroll
      for (int d = 0; d < 3; ++d) { q[d] = q3[d]; c[d] = c3[d]; }
    } else if (UG) {
      const int f = gc / p.S;
      int id = p.idx[(int64_t)gc * 64 + half * 32 + l31];
      id = id < p.N ? id : p.N - 1;
      const float* row = p.u + ((int64_t)f * p.N + id) * K + lhi * 8;
#pragma unroll
      for (int ks = 0; ks < K / 16; ++ks) {
        raw[ks][0] = *reinterpret_cast<const float4*>(row + ks * 16);
        raw[ks][1] = *reinterpret_cast<const float4*>(row + ks * 16 + 4);
      }
      if (half == 0) {
        const float* c3 = p.ctr + (int64_t)gc * 3;
#pragma unroll
        for (int d = 0; d < 3; ++d) c[d] = c3[d];
      }
    } else {
      const int j = half * 32 + l31;
      const float* row = y_in + ((int64_t)gc * 64 + (j < live ? j : 0)) * K + lhi * 8;
#pragma unroll
      for (int ks = 0; ks < K / 16; ++ks) {
        raw[ks][0] = *reinterpret_cast<const float4*>(row + ks * 16);
        raw[ks][1] = *reinterpret_cast<const float4*>(row + ks * 16 + 4);
      }
    }
  };
  float4 raw[K / 16][2];
  float qx[3] = {0.f, 0.f, 0.f}, cx[3] = {0.f, 0.f, 0.f};
  // padding-aware walk (PadWalk), two neighbourhoods ahead of the rows; the eval form writes every row for the next layer's plane GEMM
  PadWalk wk(EVAL ? nullptr : p.sched, p.G, g0, stride);
  int g = wk.g(), lv = wk.cnt(); wk.next();
  int g1 = wk.g(), lv1 = wk.cnt(); wk.next();
  int g2 = wk.g(), lv2 = wk.cnt();
  load_half(g, 0, lv, raw, qx, cx);

  for (; g < p.G; g = g1, lv = lv1, g1 = g2, lv1 = lv2, wk.next(), g2 = wk.g(), lv2 = wk.cnt()) {
    const int nh = lv > 32 ? 2 : 1;
    float mx[4], mn[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) { mx[n] = -__builtin_huge_valf(); mn[n] = __builtin_huge_valf(); }
    if (UG) {
      // this neighbourhood's add vector a_add - a_mul * (W1_xyz . centroid) (cx = its centroid, fetched with the rows of its first half)
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int ch = lane * CPL + c;
        const float v = __builtin_fmaf(WX[2 * K + ch], cx[2], __builtin_fmaf(WX[K + ch], cx[1], WX[ch] * cx[0]));
        AS[ch] = __builtin_fmaf(-M[ch], v, A[ch]);
      }
    }
#pragma unroll 1
    for (int half = 0; half < nh; ++half) {
      asm volatile("" ::: "memory");
      half8 fh[KS], fl[KS];
#pragma unroll
      for (int ks = 0; ks < K / 16; ++ks) {
        const float4 r0 = raw[ks][0], r1 = raw[ks][1];
        float x[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
        if (!GATHER) {
          const float4 m0 = *reinterpret_cast<const float4*>(M + ks * 16 + lhi * 8), m1 = *reinterpret_cast<const float4*>(M + ks * 16 + lhi * 8 + 4);
          const float4 a0 = *reinterpret_cast<const float4*>(AS + ks * 16 + lhi * 8), a1 = *reinterpret_cast<const float4*>(AS + ks * 16 + lhi * 8 + 4);
          const float mv[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w}, av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
          for (int q8 = 0; q8 < 8; ++q8) x[q8] = fmaxf(__builtin_fmaf(x[q8], mv[q8], av[q8]), 0.0f);      // relu(batch-norm(y)), as the GEMM's A loader
        }
#pragma unroll
        for (int q8 = 0; q8 < 8; ++q8) {
          _Float16 h, l;
          split1(x[q8], h, l);
          fh[ks][q8] = h; fl[ks][q8] = l;
        }
      }
      if (SA_ABL & 2) {
#pragma unroll
        for (int ks = 0; ks < K / 16; ++ks)
#pragma unroll
          for (int q8 = 0; q8 < 8; ++q8) { fh[ks][q8] = (_Float16)reinterpret_cast<const float*>(&raw[ks][0])[q8]; fl[ks][q8] = fh[ks][q8]; }
      }
      if (GATHER) {
#pragma unroll
        for (int q8 = 0; q8 < 8; ++q8) { fh[KS - 1][q8] = (_Float16)0.0f; fl[KS - 1][q8] = (_Float16)0.0f; }
        if (lhi == 0) {
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            _Float16 a, b;
            split1(__fsub_rn(qx[d], cx[d]), a, b);
            fh[KS - 1][d] = a; fl[KS - 1][d] = b;
          }
        }
      }
      // the rows of the next half (or of the next neighbourhood) travel during this half's contraction
      if (!(SA_ABL & 8)) { if (half + 1 < nh) load_half(g, 1, lv, raw, qx, cx); else load_half(g1, 0, lv1, raw, qx, cx); }
      f32x16 acc[4];
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[n][e] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        half8 wh[4], wl[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          wh[n] = *reinterpret_cast<const half8*>(Wh + (n * 32 + l31) * LD + ks * 16 + lhi * 8);
          wl[n] = *reinterpret_cast<const half8*>(Wl + (n * 32 + l31) * LD + ks * 16 + lhi * 8);
        }
        if (SA_ABL & 1) { asm volatile("" :: "v"(fl[ks]), "v"(fh[ks]), "v"(wh[0]), "v"(wl[0]), "v"(wh[1]), "v"(wl[1]), "v"(wh[2]), "v"(wl[2]), "v"(wh[3]), "v"(wl[3])); continue; }
        // term-major: consecutive MFMAs on different accumulators (the three products of one accumulator keep their order)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[ks], wh[n], acc[n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[ks], wl[n], acc[n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[ks], wh[n], acc[n], 0, 0, 0);
      }
      if (SA_ABL & 4) { asm volatile("" :: "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3])); continue; }
      if (EVAL) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const int64_t o0 = ((int64_t)g * 64 + half * 32 + 4 * lhi) * n_total + col0 + n * 32 + l31;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float v = fmaxf(__builtin_fmaf(acc[n][e] + bs[n], es[n], et[n]), 0.0f);
            _Float16 h, l;
            split1(v, h, l);
            const int64_t o = o0 + (int64_t)((e & 3) + 8 * (e >> 2)) * n_total;
            p.e_hi[o] = h; p.e_lo[o] = l;
          }
        }
        continue;
      }
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        float s = 0.0f, q = 0.0f;
        float* orow = LAYER < 3 ? p.y_out + ((int64_t)g * 64 + half * 32 + 4 * lhi) * n_total + col0 + n * 32 + l31 : nullptr;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float y = acc[n][e] + bs[n];
          s += y;
          q = __builtin_fmaf(y, y, q);
          if (LAYER == 3) { mx[n] = fmaxf(mx[n], y); mn[n] = fminf(mn[n], y); }
          else if (half * 32 + 4 * lhi + (e & 3) + 8 * (e >> 2) < lv) orow[(int64_t)((e & 3) + 8 * (e >> 2)) * n_total] = y;      // live rows only
        }
        ss[n] += (double)s;
        sq[n] += (double)q;
        if (nh == 1) {      // rows 32..63 are copies of row 0 (register 0 of the lower lane half): 32 y_0, 32 y_0^2
          const double y0 = lhi == 0 ? (double)(acc[n][0] + bs[n]) : 0.0;
          ss[n] += 32.0 * y0;
          sq[n] += 32.0 * y0 * y0;
        }
      }
    }
    if (LAYER == 3) {
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const float a = fmaxf(mx[n], __shfl_xor(mx[n], 32)), b = fminf(mn[n], __shfl_xor(mn[n], 32));
        if (lhi == 0) {
          p.out_max[(int64_t)g * n_total + col0 + n * 32 + l31] = a;
          p.out_min[(int64_t)g * n_total + col0 + n * 32 + l31] = b;
        }
      }
    }
  }
#pragma unroll
  for (int n = 0; n < 4; ++n)
    if (!EVAL) flush_stats(p.stats, p.copies, n_total, col0 + n * 32 + l31, lhi, ss[n], sq[n]);
}

// Batch statistics of the FIRST layer of a level with input features, from the per-point table (see UG above): the layer's value on
// the grouped row (neighbourhood s, neighbour j) is U[idx[s, j]] - W1_xyz . centroid_s.  One wave per neighbourhood, lanes over the
// channels (K / 64 each), 64 gathered rows of K floats (contiguous per row: every load instruction is one full row), nothing written
// but the sums.  646 MB of L2 / MALL reads at level 2 instead of 42 GFLOP of split-f16 matrix work.
template <int K>
__global__ __launch_bounds__(256) void sa_first_stats_kernel(const SaTP p) {
  constexpr int CPL = K / 64;
  typedef float vecf __attribute__((ext_vector_type(CPL)));
  __shared__ double red[2][4][K];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kp1 = p.D + 8;
  float uw[CPL][3];
#pragma unroll
  for (int c = 0; c < CPL; ++c)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const size_t o = (size_t)(lane * CPL + c) * kp1 + p.D + d;
      uw[c][d] = (float)p.wh[0][o] + (float)p.wl[0][o];
    }
  double ss[CPL], sq[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) { ss[c] = 0.0; sq[c] = 0.0; }
  for (int g = blockIdx.x * 4 + wave; g < p.G; g += gridDim.x * 4) {
    const int f = g / p.S;
    int id = p.idx[(int64_t)g * 64 + lane];
    id = id < p.N ? id : p.N - 1;
    const float* c3 = p.ctr + (int64_t)g * 3;
    const float cx = c3[0], cy = c3[1], cz = c3[2];
    float v[CPL], s[CPL], q[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      v[c] = __builtin_fmaf(uw[c][2], cz, __builtin_fmaf(uw[c][1], cy, uw[c][0] * cx));      // the order of the UG add vector
      s[c] = 0.0f; q[c] = 0.0f;
    }
    const float* ub = p.u + (int64_t)f * p.N * K + lane * CPL;
    // the points in range come first, the rest of the 64 slots repeat slot 0 (see PadWalk): the rows are gathered in runs of 16 (all
    // loads of a run in flight together) up to the last slot that differs from slot 0; the copies of row 0 beyond the last run enter
    // the sums as n y_0 and n y_0^2
    const int id0 = __shfl(id, 0);
    const unsigned long long live = __ballot(id != id0);
    const int runs = live ? (64 - __clzll(live) + 15) >> 4 : 1;
    const vecf r0 = *reinterpret_cast<const vecf*>(ub + (int64_t)id0 * K);
    float y0[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) y0[c] = r0[c] - v[c];
#pragma unroll 1
    for (int j0 = 0; j0 < runs * 16; j0 += 16) {
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const int idj = __shfl(id, j0 + jj);
        const vecf r = *reinterpret_cast<const vecf*>(ub + (int64_t)idj * K);
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          const float y = r[c] - v[c];
          s[c] += y;
          q[c] = __builtin_fmaf(y, y, q[c]);
        }
      }
    }
    const double np = (double)(64 - runs * 16);
#pragma unroll
    for (int c = 0; c < CPL; ++c) { ss[c] += (double)s[c] + np * (double)y0[c]; sq[c] += (double)q[c] + np * (double)y0[c] * (double)y0[c]; }
  }
#pragma unroll
  for (int c = 0; c < CPL; ++c) { red[0][wave][lane * CPL + c] = ss[c]; red[1][wave][lane * CPL + c] = sq[c]; }
  __syncthreads();
  double* st = p.stats + (size_t)(blockIdx.x % p.copies) * 2 * K;
  for (int i = tid; i < 2 * K; i += 256) {
    const int w = i / K, c = i - w * K;
    unsafeAtomicAdd(st + w * K + c, (red[w][0][c] + red[w][1][c]) + (red[w][2][c] + red[w][3][c]));
  }
}

// Eval-mode first layer of a level with features as an ELEMENTWISE pass over the per-point table: planes of
// relu(s0 * (u[idx] - W1_xyz . centroid) + t0) for every grouped row — what the fused-grouping GEMM with the folded BatchNorm epilogue
// writes for the next layer's plane GEMM, without the matrix work.  One wave per neighbourhood, lanes over the channels.
template <int K>
__global__ __launch_bounds__(256) void sa_table_apply_kernel(const SaTP p) {
  constexpr int CPL = K / 64;
  typedef float vecf __attribute__((ext_vector_type(CPL)));
  typedef _Float16 vech __attribute__((ext_vector_type(CPL)));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kp1 = p.D + 8;
  float uw[CPL][3], s0[CPL], t0[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int ch = lane * CPL + c;
    s0[c] = p.am[0][ch]; t0[c] = p.aa[0][ch];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const size_t o = (size_t)ch * kp1 + p.D + d;
      uw[c][d] = (float)p.wh[0][o] + (float)p.wl[0][o];
    }
  }
  for (int g = blockIdx.x * 4 + wave; g < p.G; g += gridDim.x * 4) {
    const int f = g / p.S;
    int id = p.idx[(int64_t)g * 64 + lane];
    id = id < p.N ? id : p.N - 1;
    const float* c3 = p.ctr + (int64_t)g * 3;
    const float cx = c3[0], cy = c3[1], cz = c3[2];
    float ad[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const float v = __builtin_fmaf(uw[c][2], cz, __builtin_fmaf(uw[c][1], cy, uw[c][0] * cx));
      ad[c] = __builtin_fmaf(-s0[c], v, t0[c]);
    }
    const float* ub = p.u + (int64_t)f * p.N * K + lane * CPL;
    _Float16* oh = p.e_hi + (int64_t)g * 64 * K + lane * CPL;
    _Float16* ol = p.e_lo + (int64_t)g * 64 * K + lane * CPL;
#pragma unroll 8
    for (int j = 0; j < 64; ++j) {
      const int idj = __shfl(id, j);
      const vecf r = *reinterpret_cast<const vecf*>(ub + (int64_t)idj * K);
      vech h, l;
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const float y = fmaxf(__builtin_fmaf(r[c], s0[c], ad[c]), 0.0f);
        _Float16 a, b;
        split1(y, a, b);
        h[c] = a; l[c] = b;
      }
      *reinterpret_cast<vech*>(oh + (int64_t)j * K) = h;
      *reinterpret_cast<vech*>(ol + (int64_t)j * K) = l;
    }
  }
}

// pfpp_sa_pad_schedule, two launches.  (1) one thread per neighbourhood, all CUs: its live-slot count = 1 + the highest slot that differs
// from slot 0 (for a ball query's list = the number of points in range; found slot by slot, so the skip is exact for ANY index list) —
// two cache lines per neighbourhood; a single workgroup fetching all of them through one CU's L2 port took 75 us at 19,712 neighbourhoods.  (2) one workgroup over the flags:
// thread t takes a contiguous run, an exclusive scan of the per-thread counts places the two-half neighbourhoods in ascending order at
// the front and the one-half ones behind them.
__global__ __launch_bounds__(256) void sa_pad_flags_kernel(const int32_t* __restrict__ idx, int G, int32_t* __restrict__ flags) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= G) return;
  const int4* sl = reinterpret_cast<const int4*>(idx + (int64_t)g * 64);      // two 128-byte lines
  const int id0 = idx[(int64_t)g * 64];
  int last = 0;                                // highest slot that differs from slot 0
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int4 v = sl[i];
    last = v.x != id0 ? 4 * i : last;
    last = v.y != id0 ? 4 * i + 1 : last;
    last = v.z != id0 ? 4 * i + 2 : last;
    last = v.w != id0 ? 4 * i + 3 : last;
  }
  flags[g] = last + 1;                         // live slots: cnt .. 63 repeat slot 0
}

__global__ __launch_bounds__(1024) void sa_pad_schedule_kernel(const int32_t* __restrict__ flags, int G, int32_t* __restrict__ sched) {
  int32_t* __restrict__ cnt_at = sched + 2 * G + 1;      // live-slot counts in schedule order
  __shared__ int wsum[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int per = (G + 1023) / 1024;
  const int b = t * per < G ? t * per : G, e = b + per < G ? b + per : G;
  // up to 32 neighbourhoods per thread (G <= 32,768: every level of the path) stay in a register mask between the count and the scatter
  constexpr int PER_REG = 32;
  unsigned mask = 0;
  int n = 0;
  int live[PER_REG];
  if (per <= PER_REG) {
#pragma unroll
    for (int i = 0; i < PER_REG; ++i) {
      live[i] = b + i < e ? flags[b + i] : 0;
      mask |= (live[i] > 32 ? 1u : 0u) << i;
    }
    n = __popc(mask);
  } else {
    for (int g = b; g < e; ++g) n += flags[g] > 32;
  }
  // exclusive scan of the per-thread counts: inside the wave by shuffles, across the 16 waves through 16 words of LDS
  int inc = n;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int v = __shfl_up(inc, d);
    if (lane >= d) inc += v;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int before = 0, total2 = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const int v = wsum[w];
    before += w < wave ? v : 0;
    total2 += v;
  }
  int a = before + inc - n;             // two-half neighbourhoods before this thread's run
  int o = total2 + (b - a);             // one-half ones before it, behind all the two-half ones
  if (per <= PER_REG) {
#pragma unroll
    for (int i = 0; i < PER_REG; ++i)
      if (b + i < e) {
        const int pos = (mask >> i) & 1u ? a++ : o++;
        sched[pos] = b + i;
        cnt_at[pos] = live[i];
      }
  } else {
    for (int g = b; g < e; ++g) {
      const int pos = flags[g] > 32 ? a++ : o++;
      sched[pos] = g;
      cnt_at[pos] = flags[g];
    }
  }
  if (t == 0) sched[G] = total2;
}

}  // namespace

extern "C" int pfpp_sa_pad_schedule(const int32_t* idx, int64_t G, int64_t ns, int32_t* sched, pfpp_stream_t stream) {
  PFPP_REQUIRE(idx && sched && G >= 0, "null pointer / negative size");
  PFPP_SUPPORTED(ns == 64, "the padding schedule is for 64-neighbour levels (two halves of 32 rows)");
  PFPP_REQUIRE(G < (1ll << 25), "too many neighbourhoods");
  if (G == 0) return PFPP_OK;
  int32_t* flags = sched + G + 1;      // scratch behind the schedule
  hipLaunchKernelGGL(sa_pad_flags_kernel, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, pfpp::as_stream(stream), idx, (int)G, flags);
  hipLaunchKernelGGL(sa_pad_schedule_kernel, dim3(1), dim3(1024), 0, pfpp::as_stream(stream), (const int32_t*)flags, (int)G, sched);
  return pfpp::check_launch("pfpp_sa_pad_schedule");
}

extern "C" int pfpp_sa_train_stage(const pfpp_sa_train_args* a, pfpp_stream_t stream) {
  PFPP_REQUIRE(a, "null args");
  PFPP_REQUIRE(a->xyz && a->new_xyz && a->idx && a->stats, "null pointer");
  PFPP_REQUIRE(a->F >= 0 && a->N > 0 && a->S > 0 && a->stats_copies >= 1, "bad sizes");
  const bool lvl1 = a->feats == nullptr;
  PFPP_REQUIRE(a->stage >= 1 && a->stage <= 3, "stage out of range (1..3)");
  if (a->u_in && a->stage <= 2) {
    // first layer by linearity: stage 1 = statistics of U[idx] - W1_xyz . centroid, stage 2 = second layer from the gathered rows
    PFPP_REQUIRE(!lvl1, "the per-point table belongs to a level with input features");
    PFPP_SUPPORTED(a->ns == 64 && ((a->D == 128 && a->C1 == 128 && a->C2 == 128) || (a->D == 256 && a->C1 == 256 && a->C2 == 256)),
                   "per-point first layer: nsample 64, (128 -> 128 -> 128) or (256 -> 256 -> 256) only");
    PFPP_REQUIRE(a->w_hi[0] && a->w_lo[0] && pfpp::aligned16(a->u_in), "first-layer planes / table alignment");
    PFPP_REQUIRE(a->F * a->S < (1ll << 25), "too many neighbourhoods");
    if (a->F == 0) return PFPP_OK;
    SaTP p;
    p.xyz = a->xyz; p.ctr = a->new_xyz; p.feats = a->feats; p.idx = a->idx;
    for (int i = 0; i < 3; ++i) { p.wh[i] = (const _Float16*)a->w_hi[i]; p.wl[i] = (const _Float16*)a->w_lo[i]; p.bias[i] = a->bias[i]; }
    for (int i = 0; i < 2; ++i) { p.am[i] = a->a_mul[i]; p.aa[i] = a->a_add[i]; }
    p.stats = a->stats; p.copies = (int)a->stats_copies;
    p.y_out = a->y_out; p.out_max = a->out_max; p.out_min = a->out_min;
    p.N = (int)a->N; p.S = (int)a->S; p.G = (int)(a->F * a->S);
    p.u = a->u_in; p.D = (int)a->D;
    p.e_hi = nullptr; p.e_lo = nullptr; p.sched = a->sched;
    hipStream_t st = pfpp::as_stream(stream);
    int64_t cap = a->max_workgroups > 0 ? a->max_workgroups : 256;
    if (a->stage == 1) {
      const int64_t need = (p.G + 3) / 4;
      const unsigned grid = (unsigned)(need < 4 * cap ? need : 4 * cap);          // light workgroups: four per CU
      if (a->D == 128) hipLaunchKernelGGL((sa_first_stats_kernel<128>), dim3(grid), dim3(256), 0, st, p);
      else hipLaunchKernelGGL((sa_first_stats_kernel<256>), dim3(grid), dim3(256), 0, st, p);
      return pfpp::check_launch("pfpp_sa_train_stage");
    }
    PFPP_REQUIRE(a->w_hi[1] && a->w_lo[1] && a->bias[1] && pfpp::aligned16(a->w_hi[1]) && pfpp::aligned16(a->w_lo[1]) && a->a_mul[0] &&
                 a->a_add[0] && a->y_out && pfpp::aligned16(a->y_out), "second-layer operands / first-layer affine / y_out missing");
    const int n_total = (int)a->C2;
    const int n_slices = n_total / 128;
    cap = cap / (8 * n_slices) * (8 * n_slices);
    if (cap < n_slices) cap = n_slices;
    constexpr size_t smem_128 = (size_t)2 * 128 * (128 + 8) * sizeof(_Float16) + (2 + 4 + 3) * 128 * sizeof(float);
    constexpr size_t smem_256 = (size_t)2 * 128 * (256 + 8) * sizeof(_Float16) + (2 + 4 + 3) * 256 * sizeof(float);
    static bool attr_u = false;
    if (!attr_u) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sa_wide_train_kernel<128, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_128);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sa_wide_train_kernel<256, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_256);
      attr_u = true;
    }
    if (a->D == 128) hipLaunchKernelGGL((sa_wide_train_kernel<128, 2, true>), dim3((unsigned)cap), dim3(256), smem_128, st, p, (const float*)nullptr, n_total);
    else hipLaunchKernelGGL((sa_wide_train_kernel<256, 2, true>), dim3((unsigned)cap), dim3(256), smem_256, st, p, (const float*)nullptr, n_total);
    return pfpp::check_launch("pfpp_sa_train_stage");
  }
  if (!lvl1 && a->D == 256) {
    // wide level (sa3): one rows launch per layer, the [rows, 256] pre-activations in between
    PFPP_SUPPORTED(a->ns == 64 && a->C1 == 256 && a->C2 == 256 && a->C3 == 512, "wide train-mode level: nsample 64, 256 features, widths 256/256/512 only");
    const int L = a->stage;
    PFPP_REQUIRE(a->w_hi[L - 1] && a->w_lo[L - 1] && a->bias[L - 1] && pfpp::aligned16(a->w_hi[L - 1]) && pfpp::aligned16(a->w_lo[L - 1]),
                 "weights / bias of the layer this stage computes are missing");
    PFPP_REQUIRE(L == 1 || (a->a_mul[L - 2] && a->a_add[L - 2] && a->y_in && pfpp::aligned16(a->y_in)), "the previous layer's rows / affine are missing");
    PFPP_REQUIRE(L == 3 ? (a->out_max && a->out_min) : (a->y_out != nullptr), "output missing");
    PFPP_REQUIRE(pfpp::aligned16(a->feats) && a->F * a->S < (1ll << 25), "alignment / too many neighbourhoods");
    if (a->F == 0) return PFPP_OK;
    SaTP p;
    p.xyz = a->xyz; p.ctr = a->new_xyz; p.feats = a->feats; p.idx = a->idx;
    for (int i = 0; i < 3; ++i) { p.wh[i] = (const _Float16*)a->w_hi[i]; p.wl[i] = (const _Float16*)a->w_lo[i]; p.bias[i] = a->bias[i]; }
    for (int i = 0; i < 2; ++i) { p.am[i] = a->a_mul[i]; p.aa[i] = a->a_add[i]; }
    p.stats = a->stats; p.copies = (int)a->stats_copies;
    p.y_out = a->y_out; p.out_max = a->out_max; p.out_min = a->out_min;
    p.N = (int)a->N; p.S = (int)a->S; p.G = (int)(a->F * a->S);
    p.u = nullptr; p.D = (int)a->D;
    p.e_hi = nullptr; p.e_lo = nullptr; p.sched = a->sched;
    PFPP_SUPPORTED(!(a->sched && L == 1), "the grouped first layer of the wide level takes no padding schedule (use the per-point table: u_in)");
    const int n_total = L == 3 ? 512 : 256;
    const int n_slices = n_total / 128;
    int64_t cap = a->max_workgroups > 0 ? a->max_workgroups : 256;
    cap = cap / (8 * n_slices) * (8 * n_slices);           // a whole number of 8-workgroup rounds per column slice: the slices of a row group share an XCD
    if (cap < n_slices) cap = n_slices;
    const unsigned grid = (unsigned)cap;
    constexpr size_t smem_g = (size_t)2 * 128 * ((256 / 16 + 1) * 16 + 8) * sizeof(_Float16) + 2 * 256 * sizeof(float);
    constexpr size_t smem_r = (size_t)2 * 128 * (256 + 8) * sizeof(_Float16) + 2 * 256 * sizeof(float);
    static bool attr_w = false;
    if (!attr_w) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sa_wide_train_kernel<256, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_g);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sa_wide_train_kernel<256, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_r);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sa_wide_train_kernel<256, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_r);
      attr_w = true;
    }
    hipStream_t st = pfpp::as_stream(stream);
    if (L == 1) hipLaunchKernelGGL((sa_wide_train_kernel<256, 1>), dim3(grid), dim3(256), smem_g, st, p, a->y_in, n_total);
    else if (L == 2) hipLaunchKernelGGL((sa_wide_train_kernel<256, 2>), dim3(grid), dim3(256), smem_r, st, p, a->y_in, n_total);
    else hipLaunchKernelGGL((sa_wide_train_kernel<256, 3>), dim3(grid), dim3(256), smem_r, st, p, a->y_in, n_total);
    return pfpp::check_launch("pfpp_sa_train_stage");
  }
  const bool rows3 = !lvl1 && a->stage == 3;          // reads the raw rows stage 2 wrote: only layer 3's operands are needed
  for (int i = rows3 ? 2 : 0; i < a->stage; ++i) {
    PFPP_REQUIRE(a->w_hi[i] && a->w_lo[i] && a->bias[i], "weights / bias of a layer this stage computes are missing");
    PFPP_REQUIRE(pfpp::aligned16(a->w_hi[i]) && pfpp::aligned16(a->w_lo[i]), "planes must be 16-byte aligned");
  }
  for (int i = rows3 ? 1 : 0; i + 1 < a->stage; ++i)
    PFPP_REQUIRE(a->a_mul[i] && a->a_add[i], "finalised BatchNorm affine of an earlier layer is missing");
  if (lvl1) {
    PFPP_SUPPORTED(!a->sched, "no padding schedule for the 32-neighbour level");
    PFPP_SUPPORTED(a->ns == 32 && a->C1 == 64 && a->C2 == 64 && a->C3 == 128, "train-mode chain without features: nsample 32, widths 64/64/128 only");
    PFPP_REQUIRE(a->stage < 3 || (a->out_max && a->out_min), "stage 3 writes the per-neighbourhood max and min");
    PFPP_REQUIRE(a->F * a->S < (1ll << 31), "too many neighbourhoods");
  } else {
    PFPP_SUPPORTED(a->ns == 64 && a->D == 128 && a->C1 == 128 && a->C2 == 128 && (a->stage < 3 || a->C3 == 256),
                   "train-mode chain with features: nsample 64, 128 features, widths 128/128/256 only");
    PFPP_REQUIRE(a->stage < 2 || (a->y_out && pfpp::aligned16(a->y_out)), "stage 2 writes / stage 3 reads the raw layer-2 rows");
    PFPP_REQUIRE(a->stage < 3 || (a->out_max && a->out_min), "stage 3 writes the per-neighbourhood max and min");
    PFPP_REQUIRE(pfpp::aligned16(a->feats), "16-byte alignment");
    PFPP_REQUIRE(a->F * a->S < (1ll << 25), "too many neighbourhoods");
  }
  if (a->F == 0) return PFPP_OK;
  SaTP p;
  p.xyz = a->xyz; p.ctr = a->new_xyz; p.feats = a->feats; p.idx = a->idx;
  for (int i = 0; i < 3; ++i) {
    p.wh[i] = (const _Float16*)a->w_hi[i]; p.wl[i] = (const _Float16*)a->w_lo[i]; p.bias[i] = a->bias[i];
  }
  for (int i = 0; i < 2; ++i) { p.am[i] = a->a_mul[i]; p.aa[i] = a->a_add[i]; }
  p.stats = a->stats; p.copies = (int)a->stats_copies;
  p.y_out = a->y_out; p.out_max = a->out_max; p.out_min = a->out_min;
  p.N = (int)a->N; p.S = (int)a->S; p.G = (int)(a->F * a->S);
  p.u = nullptr; p.D = (int)a->D;
  p.e_hi = nullptr; p.e_lo = nullptr; p.sched = a->sched;
  const int64_t cap = a->max_workgroups > 0 ? a->max_workgroups : 256;       // persistent: one 4-wave workgroup per CU the stream may use
  const int64_t wgs_needed = (p.G + 3) / 4;
  const unsigned grid = (unsigned)(wgs_needed < cap ? wgs_needed : cap);
  hipStream_t st = pfpp::as_stream(stream);
  if (lvl1) {
    if (a->stage == 1) hipLaunchKernelGGL((sa1_train_kernel<64, 64, 128, 1>), dim3(grid), dim3(256), 0, st, p);
    else if (a->stage == 2) hipLaunchKernelGGL((sa1_train_kernel<64, 64, 128, 2>), dim3(grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((sa1_train_kernel<64, 64, 128, 3>), dim3(grid), dim3(256), 0, st, p);
  } else {
    constexpr int d = 128, c1 = 128, c2 = 128;
    constexpr size_t smem1 = (size_t)2 * c1 * ((d / 16 + 1) * 16 + 8) * sizeof(_Float16);
    constexpr size_t smem2 = smem1 + (size_t)2 * c2 * (c1 + 8) * sizeof(_Float16) + (size_t)3 * c1 * sizeof(float);
    constexpr int c3 = 256;
    constexpr size_t smem3 = (size_t)2 * c3 * (c2 + 8) * sizeof(_Float16) + (size_t)2 * c2 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sa_rows_train_kernel<c2, c3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem3);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sa_rows8_train_kernel<c2, c3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem3);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sa2_train_kernel<d, c1, c2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem1);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sa2_train_kernel<d, c1, c2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2);
      attr_set = true;
    }
    PFPP_SUPPORTED(!a->sched || a->stage == 3, "stages 1 and 2 take the padding schedule only in their table-fed form (u_in)");
    if (a->stage == 1) hipLaunchKernelGGL((sa2_train_kernel<d, c1, c2, 1>), dim3(grid), dim3(256), smem1, st, p);
    else if (a->stage == 2) hipLaunchKernelGGL((sa2_train_kernel<d, c1, c2, 2>), dim3(grid), dim3(256), smem2, st, p);
    else {
      // two waves per SIMD (8-wave workgroups, half a neighbourhood per wave step) unless PFPP_SA_ROWS8=0
      static const bool rows8 = !(getenv("PFPP_SA_ROWS8") && atoi(getenv("PFPP_SA_ROWS8")) == 0);
      PFPP_SUPPORTED(!a->sched || rows8, "the padding schedule needs the 8-wave rows kernel (PFPP_SA_ROWS8=0 is set)");
      const int64_t need8 = (p.G + 7) / 8;
      if (rows8) hipLaunchKernelGGL((sa_rows8_train_kernel<c2, c3>), dim3((unsigned)(need8 < cap ? need8 : cap)), dim3(512), smem3, st, p);
      else hipLaunchKernelGGL((sa_rows_train_kernel<c2, c3>), dim3(grid), dim3(256), smem3, st, p);
    }
  }
  return pfpp::check_launch("pfpp_sa_train_stage");
}

// Eval-mode level with features (sa2), layers 1 and 2 from the per-point table: u [F*N, C1] = [feats | xyz] . W1^T (NO bias: the folded
// shift t0 carries it), (s0, t0) / (s1, t1) the folded BatchNorm scale / shift of layers 1 / 2 (utils/pn2_utils.py:210-216 in .eval()).
// out = relu(s1 * conv2(relu(s0 * (u[idx] - W1_xyz . centroid) + t0)) + t1) as split-f16 planes [F*S*ns, C2] — what pfpp_sa_mlp2_fused_p
// produces from the grouped first convolution, without computing it.
extern "C" int pfpp_sa_mlp2_table_p(const float* u, const float* new_xyz, const int32_t* idx, const void* w0_hi, const void* w0_lo,
                                    const void* w1_hi, const void* w1_lo, const float* s0, const float* t0, const float* s1,
                                    const float* t1, const pfpp_planes* out, int64_t F, int64_t N, int64_t S, int64_t ns, int64_t D,
                                    int64_t C1, int64_t C2, int64_t max_workgroups, pfpp_stream_t stream) {
  PFPP_REQUIRE(u && new_xyz && idx && w0_hi && w0_lo && w1_hi && w1_lo && s0 && t0 && s1 && t1 && pfpp_planes_ok(out) && out, "null pointer");
  PFPP_SUPPORTED(ns == 64 && D == 128 && C1 == 128 && C2 == 128, "table-fed eval level: nsample 64, 128 features, widths 128/128 only");
  PFPP_SUPPORTED(out->scale == 1.0f, "unit plane scale only");
  PFPP_REQUIRE(pfpp::aligned16(u) && pfpp::aligned16(w1_hi) && pfpp::aligned16(w1_lo) && F >= 0 && N > 0 && S > 0 && F * S < (1ll << 25),
               "alignment / sizes");
  if (F == 0) return PFPP_OK;
  SaTP p = {};
  p.ctr = new_xyz; p.idx = idx;
  p.wh[0] = (const _Float16*)w0_hi; p.wl[0] = (const _Float16*)w0_lo;
  p.wh[1] = (const _Float16*)w1_hi; p.wl[1] = (const _Float16*)w1_lo;
  p.am[0] = s0; p.aa[0] = t0; p.am[1] = s1; p.aa[1] = t1;
  p.N = (int)N; p.S = (int)S; p.G = (int)(F * S);
  p.u = u; p.D = (int)D;
  p.e_hi = (_Float16*)out->hi; p.e_lo = (_Float16*)out->lo;
  int64_t cap = max_workgroups > 0 ? max_workgroups : 256;
  cap = cap / 8 * 8;
  if (cap < 1) cap = 1;
  const int64_t need = (p.G + 3) / 4;
  if (need < cap) cap = need;
  constexpr size_t smem = (size_t)2 * 128 * (128 + 8) * sizeof(_Float16) + (2 + 4 + 3) * 128 * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sa_wide_train_kernel<128, 2, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr = true;
  }
  hipLaunchKernelGGL((sa_wide_train_kernel<128, 2, true, true>), dim3((unsigned)cap), dim3(256), smem, pfpp::as_stream(stream), p, (const float*)nullptr, 128);
  return pfpp::check_launch("pfpp_sa_mlp2_table_p");
}

// Eval-mode first layer of a level with features from the per-point table (see pfpp_sa_mlp2_table_p for u): out = relu(s0 * (u[idx] -
// W1_xyz . new_xyz) + t0) as split-f16 planes [F*S*ns, C1] — the A operand of the level's second convolution.  (D, C1) = (256, 256) or
// (128, 128), ns == 64.
extern "C" int pfpp_sa_table_planes(const float* u, const float* new_xyz, const int32_t* idx, const void* w0_hi, const void* w0_lo,
                                    const float* s0, const float* t0, const pfpp_planes* out, int64_t F, int64_t N, int64_t S, int64_t ns,
                                    int64_t D, int64_t C1, pfpp_stream_t stream) {
  PFPP_REQUIRE(u && new_xyz && idx && w0_hi && w0_lo && s0 && t0 && out && pfpp_planes_ok(out), "null pointer");
  PFPP_SUPPORTED(ns == 64 && ((D == 256 && C1 == 256) || (D == 128 && C1 == 128)), "table-fed first layer: nsample 64, (256 -> 256) or (128 -> 128)");
  PFPP_SUPPORTED(out->scale == 1.0f, "unit plane scale only");
  PFPP_REQUIRE(pfpp::aligned16(u) && F >= 0 && N > 0 && S > 0 && F * S < (1ll << 25), "alignment / sizes");
  if (F == 0) return PFPP_OK;
  SaTP p = {};
  p.ctr = new_xyz; p.idx = idx;
  p.wh[0] = (const _Float16*)w0_hi; p.wl[0] = (const _Float16*)w0_lo;
  p.am[0] = s0; p.aa[0] = t0;
  p.N = (int)N; p.S = (int)S; p.G = (int)(F * S);
  p.u = u; p.D = (int)D;
  p.e_hi = (_Float16*)out->hi; p.e_lo = (_Float16*)out->lo;
  const int64_t need = (p.G + 3) / 4;
  const unsigned grid = (unsigned)(need < 2048 ? need : 2048);
  hipStream_t st = pfpp::as_stream(stream);
  if (D == 256) hipLaunchKernelGGL((sa_table_apply_kernel<256>), dim3(grid), dim3(256), 0, st, p);
  else hipLaunchKernelGGL((sa_table_apply_kernel<128>), dim3(grid), dim3(256), 0, st, p);
  return pfpp::check_launch("pfpp_sa_table_planes");
}
