// The six transformer blocks of the DenoiserTransformer (denoiser_transformer.py:173-185 = 6 x EncoderLayer, attention.py:77-140:
// AdaLN -> per-fragment self-attention -> +res -> AdaLN -> key-masked global attention -> +res -> LayerNorm -> GEGLU feed-forward
// -> +res) for SMALL token counts (<= 512 tokens: one puzzle in flight, auto_aggl.py:136-151) as ONE persistent kernel.
//
// Why: at 50-500 tokens every one of the 66 launches of the layer loop is a 5-17 us dependent step (weights stream in cold, 16-250
// workgroups, DESIGN.md 3.1.1 "single-puzzle launches"): 1.0 ms per DDPM step for 13 GFLOP.  Here one workgroup per CU stays
// resident for all 48 phases; the phases are separated by a software grid barrier (1.7 us measured across the 8 XCDs,
// tools/lab/gridsync_probe.hip) instead of a kernel boundary.
//
// Cross-XCD visibility without cache maintenance: the L2s of the 8 XCDs are not coherent with each other, and the release /
// acquire fences that make ordinary stores visible (buffer_wbl2 / buffer_inv) cost 6 us per barrier with 256 workgroups.  Every
// buffer that one workgroup writes and another reads (h, qkv, att, u) is therefore accessed ONLY through agent-scope relaxed
// atomics (global_load / global_store with sc1: served by the memory side, coherent by themselves); the barrier then needs
// nothing but workgroup-scope fences.  Weights, biases and the AdaLN rows are read-only here and use ordinary cached loads.
//
// GEMM phases: a workgroup owns 32 x 32 output tiles (or a value / gate pair of them for GEGLU); its four waves split the
// contraction (each K / 4) and meet through LDS, so every wave has ONE batch of fragment loads in flight per tile - the phases are
// round trips, not bandwidth.  A operands: the AdaLN / LayerNorm of the 32 rows is computed by the workgroup itself into LDS planes
// (QKV and GEGLU projections), or read as fp32 fragments straight from the buffer the previous phase wrote (out-projections,
// second feed-forward linear).  W fragments go global -> registers (each lane its own 16-byte runs), no staging.
// Arithmetic: the split-f16 contraction of gemm.hip (lo.hi, hi.lo, hi.hi per 16-deep step into one fp32 accumulator).
// Attention: block-diagonal = attn_blockdiag_mfma_kernel's scheme (one wave per fragment and head, fp32 MFMA, no LDS); global =
// S^T = K.Q^T per 32-key tile with the keys of a sequence split over the four waves (online softmax per wave, merged in LDS).
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <stdint.h>
#include <stdlib.h>

#include "pfpp.h"
#include "pfpp_common.h"
#include "sa_common.h"

namespace {

constexpr int TB_C = 512, TB_H = 8, TB_DH = 64, TB_INNER = 2048;
constexpr int TB_LDA = TB_C + 8;            // LDS row stride of the normalised A tile in halfs (16-byte fragment reads conflict-free)

struct TbLayerD {
  const _Float16 *wqkv1_h, *wqkv1_l, *wo1_h, *wo1_l, *wqkv2_h, *wqkv2_l, *wo2_h, *wo2_l, *w1_h, *w1_l, *w2_h, *w2_l;
  const float *bo1, *bo2, *g3, *be3, *b1, *b2;
  float a_qkv1, a_o1, a_qkv2, a_o2, a_w1, a_w2;      // 1 / plane scale of each weight
};

struct TbP {
  float* h; float* qkv; float* att; float* u;
  const float* mods;                  // [2 * layers, B, 2C]
  const int32_t* frag_b;              // [Fv] AdaLN row (puzzle) of every fragment
  const int32_t* seq_off; const int32_t* seq_len;   // [B] token range of every puzzle's sequence
  unsigned* bar;
  unsigned gen_base;
  int M, B, Fv, L, n_layers;
  int skip;                           // diagnostic (PFPP_TBLOCK_SKIP): bit i = leave phase i of every layer out (timing only)
  float att_scale, eps;
  TbLayerD layer[8];
};

// ---- agent-coherent accesses to the buffers the workgroups exchange ----
__device__ __forceinline__ float2 ldc2(const float* p) {
  const uint64_t v = __hip_atomic_load(reinterpret_cast<uint64_t*>(const_cast<float*>(p)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  union { uint64_t u; float2 f; } c;
  c.u = v;
  return c.f;
}
__device__ __forceinline__ float ldc1(const float* p) {
  return __hip_atomic_load(const_cast<float*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void stc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stc2(float* p, float a, float b) {
  union { uint64_t u; float2 f; } c;
  c.f = make_float2(a, b);
  __hip_atomic_store(reinterpret_cast<uint64_t*>(p), c.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 8 consecutive floats of a row (32 bytes, 8-byte aligned)
__device__ __forceinline__ void ldc8(const float* p, float (&x)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float2 v = ldc2(p + 2 * i);
    x[2 * i] = v.x; x[2 * i + 1] = v.y;
  }
}

// bar (unsigned words, each counter on its own 256-byte line): [0] root, [64] release generation, [128 + 64 g] group g of 16.
// Counters only grow: the launch passes the number of barriers all earlier launches executed.
__device__ __forceinline__ void grid_sync(unsigned* bar, unsigned nwg, unsigned& gen) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    constexpr unsigned NG = 16;
    const unsigned gsz = nwg / NG;
    const unsigned g = blockIdx.x / gsz;
    const unsigned t = __hip_atomic_fetch_add(bar + 128 + 64 * g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool last = false;
    if (t == (gen + 1) * gsz - 1) {
      const unsigned r = __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = r == (gen + 1) * NG - 1;
    }
    if (last) __hip_atomic_store(bar + 64, gen + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else while (__hip_atomic_load(bar + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen + 1) __builtin_amdgcn_s_sleep(2);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  ++gen;
  __syncthreads();
}

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ void split8f(const float (&x)[8], half8& hi, half8& lo) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    _Float16 a, b;
    split1(x[q], a, b);
    hi[q] = a; lo[q] = b;
  }
}

// ---- LDS ----
struct __align__(16) TbSmem {
  union {
    struct { _Float16 ah[32 * TB_LDA]; _Float16 al[32 * TB_LDA]; float red[3][2][16][64]; } g;     // GEMM phases: 66,560 + 24,576 B
    struct { float m[4][64]; float l[4][64]; float o[4][2][16][64]; } a;                           // global attention merge: 34,816 B
  };
};

// normalised rows row0 .. row0+31 of h as split planes in LDS: (x - mean) * rstd, then * (1 + scale) + shift with the AdaLN row of
// the token's puzzle (mod != null) or * gamma + beta.  Each wave 8 rows, a lane 8 consecutive channels.
__device__ __forceinline__ void build_norm_tile(TbSmem& sm, const TbP& p, int row0, const float* mod, const float* gamma, const float* beta) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave * 8; r < wave * 8 + 8; ++r) {
    const int row = row0 + r;
    half8 hi, lo;
    if (row < p.M) {
      float x[8];
      ldc8(p.h + (int64_t)row * TB_C + lane * 8, x);
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < 8; ++i) s += x[i];
      const float mean = wave_sum64(s) / (float)TB_C;
      float q = 0.0f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float d = x[i] - mean; q += d * d; }
      const float var = wave_sum64(q) / (float)TB_C;
      const float rstd = 1.0f / sqrtf(var + p.eps);
      float y[8];
      if (mod) {
        const float* mr = mod + (int64_t)p.frag_b[row / p.L] * (2 * TB_C) + lane * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) y[i] = (x[i] - mean) * rstd * (1.0f + mr[i]) + mr[TB_C + i];
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) y[i] = (x[i] - mean) * rstd * gamma[lane * 8 + i] + beta[lane * 8 + i];
      }
      split8f(y, hi, lo);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) { hi[i] = (_Float16)0.0f; lo[i] = (_Float16)0.0f; }
    }
    *reinterpret_cast<half8*>(sm.g.ah + r * TB_LDA + lane * 8) = hi;
    *reinterpret_cast<half8*>(sm.g.al + r * TB_LDA + lane * 8) = lo;
  }
}

enum { EPI_STORE = 0, EPI_RESIDUAL = 1, EPI_GEGLU = 2 };

// One GEMM phase: out[M, N] (+)= A[M, K] . W[N, K]^T, tiles of 32 x 32 (NT = 2: a value tile and its gate tile, GEGLU).
// A_NORM: A = the normalised rows of h built into LDS by this workgroup; else A = fp32 rows of `a_src` (ld = K) written by an
// earlier phase.  The 4 waves split K; wave 0 reduces and runs the epilogue.
template <int K, int NT, bool A_NORM, int EPI>
__device__ __forceinline__ void gemm_phase(TbSmem& sm, const TbP& p, const float* a_src, const _Float16* __restrict__ wh,
                                           const _Float16* __restrict__ wl, float alpha, const float* __restrict__ bias, float* out,
                                           int ldo, int n_out_tiles, const float* mod, const float* gamma, const float* beta) {
  constexpr int KW = K / 4;                 // contraction share of a wave
  constexpr int NKS = KW / 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int row_tiles = (p.M + 31) >> 5;
  const int n_tasks = row_tiles * n_out_tiles;
  const int per = (n_tasks + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t0 = blockIdx.x * per, t1 = min(n_tasks, t0 + per);
  int built = -1;
  for (int t = t0; t < t1; ++t) {
    const int rt = t / n_out_tiles, ct = t - rt * n_out_tiles;
    const int row0 = rt * 32;
    const int wrow0 = ct * 32 * NT;         // first weight row (= output column of the packed layout) of this task
    const int kb = wave * KW;
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;
    // the residual rows of the epilogue are fetched now (wave 0): their round trip to the memory side runs under the contraction
    float res[16];
    if (EPI == EPI_RESIDUAL && wave == 0) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = min(row0 + (e & 3) + 8 * (e >> 2) + 4 * lhi, p.M - 1);
        res[e] = ldc1(out + (int64_t)row * ldo + ct * 32 + l31);
      }
    }
#pragma unroll 1
    for (int k0 = 0; k0 < NKS; k0 += 8) {
      // one batch: every fragment load of 8 k-steps in flight, then the products.  The (cold) weight loads go out first; the
      // normalised A tile is built while they travel.
      half8 w_h[8][NT], w_l[8][NT], a_h[8], a_l[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int k = kb + (k0 + s) * 16 + lhi * 8;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          w_h[s][j] = *reinterpret_cast<const half8*>(wh + (int64_t)(wrow0 + j * 32 + l31) * K + k);
          w_l[s][j] = *reinterpret_cast<const half8*>(wl + (int64_t)(wrow0 + j * 32 + l31) * K + k);
        }
      }
      if (A_NORM) {
        if (built != rt) {
          build_norm_tile(sm, p, row0, mod, gamma, beta);
          built = rt;
          __syncthreads();
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const int k = kb + (k0 + s) * 16 + lhi * 8;
          a_h[s] = *reinterpret_cast<const half8*>(sm.g.ah + l31 * TB_LDA + k);
          a_l[s] = *reinterpret_cast<const half8*>(sm.g.al + l31 * TB_LDA + k);
        }
      } else {
        const int row = min(row0 + l31, p.M - 1);
        float x[8][8];
#pragma unroll
        for (int s = 0; s < 8; ++s) ldc8(a_src + (int64_t)row * K + kb + (k0 + s) * 16 + lhi * 8, x[s]);
#pragma unroll
        for (int s = 0; s < 8; ++s) split8f(x[s], a_h[s], a_l[s]);
      }
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_l[s], w_h[s][j], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h[s], w_l[s][j], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h[s], w_h[s][j], acc[j], 0, 0, 0);
        }
    }
    // ---- the four K shares meet in LDS; wave 0 finishes the tile ----
    if (wave > 0) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) sm.g.red[wave - 1][j][e][lane] = acc[j][e];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] += (sm.g.red[0][j][e][lane] + sm.g.red[1][j][e][lane]) + sm.g.red[2][j][e][lane];
      // accumulator: lane = output column l31, register e = row (e&3) + 8*(e>>2) + 4*lhi
      if (EPI == EPI_GEGLU) {
        const float bv = bias[wrow0 + l31], bg = bias[wrow0 + 32 + l31];
        const int col = ct * 32 + l31;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = row0 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
          const float v = acc[0][e] * alpha + bv;
          const float g = acc[NT - 1][e] * alpha + bg;
          const float r = v * (0.5f * g * (1.0f + erff(g * 0.70710678118654752440f)));
          if (row < p.M) stc1(out + (int64_t)row * ldo + col, r);
        }
      } else {
        const int col = ct * 32 + l31;
        const float b = bias ? bias[col] : 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = row0 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
          if (row < p.M) {
            float v = acc[0][e] * alpha + b;
            if (EPI == EPI_RESIDUAL) v += res[e];
            stc1(out + (int64_t)row * ldo + col, v);
          }
        }
      }
    }
    __syncthreads();          // red (and, before a rebuild, the A tile) are free again
  }
}

// per-fragment self-attention (attn_blockdiag_mfma_kernel, transformer_ops.hip): one wave per (fragment, head)
__device__ __forceinline__ void blockdiag_phase(const TbP& p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int L = p.L;
  const int64_t ld = 3 * TB_C;
  const int n_pairs = p.Fv * TB_H;
  for (int pair = blockIdx.x * 4 + wave; pair < n_pairs; pair += gridDim.x * 4) {
    const int frag = pair / TB_H, hd = pair - frag * TB_H;
    const float* base = p.qkv + (int64_t)frag * L * ld + hd * TB_DH;
    const int row = l31 < L ? l31 : L - 1;
    const float* qp = base + row * ld + lhi * 4;
    const float* kp = qp + TB_C;
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.0f;
    float qv[8][4], kv[8][4];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float2 q0 = ldc2(qp + c * 8), q1 = ldc2(qp + c * 8 + 2), k0 = ldc2(kp + c * 8), k1 = ldc2(kp + c * 8 + 2);
      qv[c][0] = q0.x; qv[c][1] = q0.y; qv[c][2] = q1.x; qv[c][3] = q1.y;
      kv[c][0] = k0.x; kv[c][1] = k0.y; kv[c][2] = k1.x; kv[c][3] = k1.y;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[c][i], qv[c][i], s, 0, 0, 0);
    float mx = -__builtin_huge_valf();
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = (e & 3) + 8 * (e >> 2) + 4 * lhi;
      s[e] = key < L ? s[e] * p.att_scale : -__builtin_huge_valf();
      mx = fmaxf(mx, s[e]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.0f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = (e & 3) + 8 * (e >> 2) + 4 * lhi;
      s[e] = key < L ? __expf(s[e] - mx) : 0.0f;
      sum += s[e];
    }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] *= inv;
    const float* vp = base + 2 * TB_C + l31;
    float* orow = p.att + ((int64_t)frag * L + l31) * TB_C + hd * TB_DH;
#pragma unroll
    for (int tile = 0; tile < 2; ++tile) {
      float vv[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int key = (t & 3) + 8 * (t >> 2) + 4 * lhi;
        vv[t] = ldc1(vp + (int64_t)(key < L ? key : L - 1) * ld + tile * 32);
      }
      f32x16 o;
#pragma unroll
      for (int e = 0; e < 16; ++e) o[e] = 0.0f;
#pragma unroll
      for (int t = 0; t < 16; ++t) o = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[t], s[t], o, 0, 0, 0);
      if (l31 < L) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float* d = orow + tile * 32 + 8 * q + 4 * lhi;
          stc2(d, o[4 * q], o[4 * q + 1]);
          stc2(d + 2, o[4 * q + 2], o[4 * q + 3]);
        }
      }
    }
  }
}

// key-masked global attention over a puzzle's sequence (attention.py:77-85 with the mask of denoiser_transformer.py:163-164): one
// workgroup per (sequence, head, 32-query tile); wave w takes the key tiles w, w+4, ... with its own online softmax.
__device__ __forceinline__ void dense_phase(TbSmem& sm, const TbP& p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int64_t ld = 3 * TB_C;
  // units in a fixed order: for every sequence its query tiles x heads
  int unit0 = 0;
  for (int b = 0; b < p.B; ++b) {
    const int off = p.seq_off[b], len = p.seq_len[b];
    const int qts = (len + 31) >> 5, nkt = qts;
    const int n_units = qts * TB_H;
    for (int u = 0; u < n_units; ++u) {
      if ((unit0 + u) % (int)gridDim.x != (int)blockIdx.x) continue;
      const int qt = u / TB_H, hd = u - qt * TB_H;
      const int qrow = off + min(qt * 32 + l31, len - 1);
      // Q fragments: lane = query, 8 consecutive head dims per 16-deep step
      half8 qh[4], ql[4];
      {
        float x[4][8];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ldc8(p.qkv + (int64_t)qrow * ld + hd * TB_DH + ks * 16 + lhi * 8, x[ks]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) split8f(x[ks], qh[ks], ql[ks]);
      }
      float m = -__builtin_huge_valf(), l = 0.0f;
      f32x16 o[2];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[dt][e] = 0.0f;
      for (int kt = wave; kt < nkt; kt += 4) {
        const int krow = off + min(kt * 32 + l31, len - 1);
        float kx[4][8];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ldc8(p.qkv + (int64_t)krow * ld + TB_C + hd * TB_DH + ks * 16 + lhi * 8, kx[ks]);
        // V^T fragments: lane = head dim (of tile dt), 8 consecutive keys per 16-deep step
        float vx[2][2][8];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int key = min(kt * 32 + kk * 16 + lhi * 8 + j, len - 1);
              vx[dt][kk][j] = ldc1(p.qkv + (int64_t)(off + key) * ld + 2 * TB_C + hd * TB_DH + dt * 32 + l31);
            }
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          half8 kh, kl;
          split8f(kx[ks], kh, kl);
          s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], s, 0, 0, 0);
          s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], s, 0, 0, 0);
          s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], s, 0, 0, 0);
        }
        // s: lane = query, register e = key (e&3) + 8*(e>>2) + 4*lhi of this tile
        float tmax = -__builtin_huge_valf();
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
          s[e] = key < len ? s[e] * p.att_scale : -__builtin_huge_valf();
          tmax = fmaxf(tmax, s[e]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m, tmax);
        const float corr = __expf(m - m_new);
        float psum = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          s[e] = __expf(s[e] - m_new);
          psum += s[e];
        }
        psum += __shfl_xor(psum, 32);
        l = l * corr + psum;
        m = m_new;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int e = 0; e < 16; ++e) o[dt][e] *= corr;
        half8 ph[2], pl[2];
        tile_to_fragments(s, lhi, ph, pl);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            half8 vh, vl;
            split8f(vx[dt][kk], vh, vl);
            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[kk], o[dt], 0, 0, 0);
            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[kk], o[dt], 0, 0, 0);
            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[kk], o[dt], 0, 0, 0);
          }
      }
      // ---- merge the four waves' partial softmaxes ----
      sm.a.m[wave][lane] = m;
      sm.a.l[wave][lane] = l;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) sm.a.o[wave][dt][e][lane] = o[dt][e];
      __syncthreads();
      {
        float mm = sm.a.m[0][lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) mm = fmaxf(mm, sm.a.m[w][lane]);
        float f[4], lt = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          f[w] = __expf(sm.a.m[w][lane] - mm);
          lt += sm.a.l[w][lane] * f[w];
        }
        const float inv = 1.0f / lt;
        // wave w finishes head-dim tile w >> 1, registers 8 (w & 1) .. + 8: two quads of 4 consecutive head dims
        const int dt = wave >> 1;
        const int q_row = qt * 32 + l31;
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const int q = (wave & 1) * 2 + qq;
          float r[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int e = 4 * q + i;
            r[i] = ((sm.a.o[0][dt][e][lane] * f[0] + sm.a.o[1][dt][e][lane] * f[1]) + (sm.a.o[2][dt][e][lane] * f[2] + sm.a.o[3][dt][e][lane] * f[3])) * inv;
          }
          if (q_row < len) {
            float* d = p.att + (int64_t)(off + q_row) * TB_C + hd * TB_DH + dt * 32 + 8 * q + 4 * lhi;
            stc2(d, r[0], r[1]);
            stc2(d + 2, r[2], r[3]);
          }
        }
      }
      __syncthreads();
    }
    unit0 += n_units;
  }
}

__global__ __launch_bounds__(256, 1) void tblock_small_kernel(const TbP p) {
  __shared__ TbSmem sm;
  unsigned gen = p.gen_base;
  const unsigned nwg = gridDim.x;
  for (int i = 0; i < p.n_layers; ++i) {
    const TbLayerD& ly = p.layer[i];
    const float* mod1 = p.mods + (int64_t)(2 * i) * p.B * (2 * TB_C);
    const float* mod2 = p.mods + (int64_t)(2 * i + 1) * p.B * (2 * TB_C);
    if (!(p.skip & 1)) gemm_phase<TB_C, 1, true, EPI_STORE>(sm, p, nullptr, ly.wqkv1_h, ly.wqkv1_l, ly.a_qkv1, nullptr, p.qkv, 3 * TB_C, 3 * TB_C / 32, mod1, nullptr, nullptr);
    grid_sync(p.bar, nwg, gen);
    if (!(p.skip & 2)) blockdiag_phase(p);
    grid_sync(p.bar, nwg, gen);
    if (!(p.skip & 4)) gemm_phase<TB_C, 1, false, EPI_RESIDUAL>(sm, p, p.att, ly.wo1_h, ly.wo1_l, ly.a_o1, ly.bo1, p.h, TB_C, TB_C / 32, nullptr, nullptr, nullptr);
    grid_sync(p.bar, nwg, gen);
    if (!(p.skip & 8)) gemm_phase<TB_C, 1, true, EPI_STORE>(sm, p, nullptr, ly.wqkv2_h, ly.wqkv2_l, ly.a_qkv2, nullptr, p.qkv, 3 * TB_C, 3 * TB_C / 32, mod2, nullptr, nullptr);
    grid_sync(p.bar, nwg, gen);
    if (!(p.skip & 16)) dense_phase(sm, p);
    grid_sync(p.bar, nwg, gen);
    if (!(p.skip & 32)) gemm_phase<TB_C, 1, false, EPI_RESIDUAL>(sm, p, p.att, ly.wo2_h, ly.wo2_l, ly.a_o2, ly.bo2, p.h, TB_C, TB_C / 32, nullptr, nullptr, nullptr);
    grid_sync(p.bar, nwg, gen);
    if (!(p.skip & 64)) gemm_phase<TB_C, 2, true, EPI_GEGLU>(sm, p, nullptr, ly.w1_h, ly.w1_l, ly.a_w1, ly.b1, p.u, TB_INNER, TB_INNER / 32, nullptr, ly.g3, ly.be3);
    grid_sync(p.bar, nwg, gen);
    if (!(p.skip & 128)) gemm_phase<TB_INNER, 1, false, EPI_RESIDUAL>(sm, p, p.u, ly.w2_h, ly.w2_l, ly.a_w2, ly.b2, p.h, TB_C, TB_C / 32, nullptr, nullptr, nullptr);
    grid_sync(p.bar, nwg, gen);
  }
}

}  // namespace

extern "C" int64_t pfpp_tblock_small_barrier_words(void) { return 128 + 64 * 16; }

extern "C" int pfpp_tblock_small(const pfpp_tblock_args* a, pfpp_stream_t stream) {
  PFPP_REQUIRE(a, "null args");
  PFPP_REQUIRE(a->h && a->qkv && a->att && a->u && a->mods && a->frag_b && a->seq_off && a->seq_len && a->barrier, "null pointer");
  PFPP_SUPPORTED(a->C == TB_C && a->H == TB_H && a->inner == TB_INNER, "small-token transformer kernel: C 512, 8 heads, GEGLU inner 2048 only");
  PFPP_REQUIRE(a->n_layers >= 1 && a->n_layers <= 8, "1..8 layers");
  PFPP_REQUIRE(a->M >= 1 && a->M <= 512 && a->L >= 1 && a->L <= 32 && a->Fv * a->L == a->M && a->B >= 1, "M = Fv * L <= 512, L <= 32");
  PFPP_REQUIRE(a->workgroups >= 16 && a->workgroups % 16 == 0, "workgroups: a multiple of 16");
  TbP p;
  p.h = a->h; p.qkv = a->qkv; p.att = a->att; p.u = a->u; p.mods = a->mods; p.frag_b = a->frag_b;
  p.seq_off = a->seq_off; p.seq_len = a->seq_len; p.bar = reinterpret_cast<unsigned*>(a->barrier);
  p.gen_base = (unsigned)a->barrier_generation;
  p.M = (int)a->M; p.B = (int)a->B; p.Fv = (int)a->Fv; p.L = (int)a->L; p.n_layers = (int)a->n_layers;
  p.att_scale = a->att_scale; p.eps = a->eps;
  { const char* e = getenv("PFPP_TBLOCK_SKIP"); p.skip = e ? atoi(e) : 0; }
  for (int i = 0; i < p.n_layers; ++i) {
    const pfpp_tblock_layer& s = a->layer[i];
    PFPP_REQUIRE(s.wqkv1_hi && s.wqkv1_lo && s.wo1_hi && s.wo1_lo && s.wqkv2_hi && s.wqkv2_lo && s.wo2_hi && s.wo2_lo && s.w1_hi && s.w1_lo &&
                 s.w2_hi && s.w2_lo && s.bo1 && s.bo2 && s.norm3_gamma && s.norm3_beta && s.b1 && s.b2, "layer operand missing");
    TbLayerD& d = p.layer[i];
    d.wqkv1_h = (const _Float16*)s.wqkv1_hi; d.wqkv1_l = (const _Float16*)s.wqkv1_lo; d.wo1_h = (const _Float16*)s.wo1_hi; d.wo1_l = (const _Float16*)s.wo1_lo;
    d.wqkv2_h = (const _Float16*)s.wqkv2_hi; d.wqkv2_l = (const _Float16*)s.wqkv2_lo; d.wo2_h = (const _Float16*)s.wo2_hi; d.wo2_l = (const _Float16*)s.wo2_lo;
    d.w1_h = (const _Float16*)s.w1_hi; d.w1_l = (const _Float16*)s.w1_lo; d.w2_h = (const _Float16*)s.w2_hi; d.w2_l = (const _Float16*)s.w2_lo;
    d.bo1 = s.bo1; d.bo2 = s.bo2; d.g3 = s.norm3_gamma; d.be3 = s.norm3_beta; d.b1 = s.b1; d.b2 = s.b2;
    d.a_qkv1 = 1.0f / s.scale_qkv1; d.a_o1 = 1.0f / s.scale_o1; d.a_qkv2 = 1.0f / s.scale_qkv2; d.a_o2 = 1.0f / s.scale_o2;
    d.a_w1 = 1.0f / s.scale_w1; d.a_w2 = 1.0f / s.scale_w2;
  }
  void* kargs[] = {(void*)&p};
  // cooperative launch: every workgroup must be resident for the grid barrier (the runtime refuses a grid that cannot be)
  const hipError_t rc = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(tblock_small_kernel), dim3((unsigned)a->workgroups), dim3(256),
                                                   kargs, 0, pfpp::as_stream(stream));
  if (rc != hipSuccess) {
    pfpp::set_error("pfpp_tblock_small: cooperative launch failed: %s", hipGetErrorString(rc));
    (void)hipGetLastError();
    return PFPP_EHIP;
  }
  return PFPP_OK;
}

extern "C" int64_t pfpp_tblock_small_barriers(int64_t n_layers) { return 8 * n_layers; }
