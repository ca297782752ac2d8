// Token embedding of the Denoiser for FEW tokens (one to a few puzzles in flight), one launch:
//   tok[(f, l), :] = shape_embedding([latent | PE(xyz) | PE(scale)]) + param_fc(PE(x_f)) + ref_part_emb[ref_f] + pe[p_f]
//
// Reference: DenoiserTransformer._gen_cond / _add_ref_part_emb / forward (denoiser/model/modules/denoiser_transformer.py:117-135,
// 150-156, 173-185) with EmbedderNerf (utils/model_utils.py:68-69), eval mode, for the compacted fragment list of
// pfpp_hip.denoiser.denoiser_forward_compact.
//
// Why: with <= 2,048 tokens the four launches it replaces (pfpp_token_features, two skinny GEMMs with K = 147 / 148, pfpp_token_combine)
// are 34 us of kernel time and three launch gaps for 0.06 GFLOP.  Here a workgroup builds the features of its 32 token rows in LDS as
// split-f16 planes — [shape features (148) | pose features of the row's fragment (147, evaluated once per fragment of the tile) | 0] = 320
// columns — and contracts them with the CONCATENATED weight [W_shape | W_param | 0] (fragment-blocked planes, as csrc/gemm_small.hip:
// every wave a 32-column unit over the whole K): both linear layers are one accumulator chain.  Feature arithmetic: token_features_kernel's
// (csrc/transformer_ops.hip: same arguments; sin and cos of one argument come from one sincosf).  The two GEMMs of the reference are rounded separately
// and then added; here their products share one fp32 accumulation: equal to the four-launch path to fp32 rounding (tested at 2e-6).
#include "pfpp_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int FS = 148;          // shape features: 64 latent + 63 PE(xyz) + 21 PE(scale)
constexpr int FP = 147;          // pose features: PE(x), 7 x 21
constexpr int KE = 320;          // FS + FP padded to a multiple of 64
constexpr int LKP = KE + 8;      // LDS row stride of a plane in halfs
constexpr int NS = KE / 16;      // MFMA steps
constexpr int MAXF = 4;          // fragments a 32-row tile can touch (L >= 11)

// EmbedderNerf.embed (utils/model_utils.py:68-69): [v | sin(2^0 v) | cos(2^0 v) | ... | sin(2^9 v) | cos(2^9 v)], blocks as wide as v
__device__ __forceinline__ float es_pe(const float* v, int d, int c) {
  const int blk = c / d, comp = c - blk * d;
  const float x = v[comp];
  if (blk == 0) return x;
  const int fi = (blk - 1) >> 1;
  const float arg = x * (float)(1 << fi);
  return ((blk - 1) & 1) ? cosf(arg) : sinf(arg);
}

struct EsP {
  const float *latent, *xyz, *scale, *x;     // [slots, L, 64], [slots, L, 3], [slots], [slots, 7]
  const int32_t* slot;                       // listed fragment -> slot (or null: identity)
  const half8 *fh, *fl; float inv_scale;     // fragment-blocked planes of w_scale * [W_shape | W_param | 0]  [C, KE]
  const float* bias;                         // [C] = shape bias + param bias
  const float* ref_emb;                      // [2, C]
  const uint8_t* ref_part;                   // [slots]
  const float* pe;                           // [max_len, C]
  const int32_t* frag_pos;                   // listed fragment -> position in its puzzle
  float* tok;                                // [n L, C]
  int M, L, C;
};

__global__ __launch_bounds__(256) void embed_small_kernel(EsP p) {
  extern __shared__ __align__(16) char es_smem[];
  _Float16* sh = reinterpret_cast<_Float16*>(es_smem);
  _Float16* sl = sh + 32 * LKP;
  float* posef = reinterpret_cast<float*>(sl + 32 * LKP);         // [MAXF][FS]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lhi = lane >> 5;
  const int r0 = blockIdx.x * 32;
  const int u = blockIdx.y * 4 + wave;
  const bool any = u < p.C / 32;

  // the unit's weights: 20 steps in two halves, the first requested before the features are built
  half8 st[2][2][NS / 2];
  const size_t blk0 = (size_t)(any ? u : 0) * NS;
  auto fetch = [&](const int b) {
#pragma unroll
    for (int i = 0; i < NS / 2; ++i) {
      st[b][0][i] = p.fh[(blk0 + (NS / 2) * b + i) * 64 + lane];
      st[b][1][i] = p.fl[(blk0 + (NS / 2) * b + i) * 64 + lane];
    }
  };
  if (any) { fetch(0); fetch(1); }
  __builtin_amdgcn_sched_barrier(0);

  const int f0 = r0 / p.L;
  const int f1 = min(r0 + 31, p.M - 1) / p.L;
  // ---- raw inputs of the tile -> LDS with ONE round of loads (slot index, then the rows): feature loops that fetched their own inputs
  // paid two dependent memory round trips per feature (30 us for the launch)
  float* raw = posef + MAXF * FS;                  // [32][68]: 64 latent | xyz | scale of the row's fragment
  float* xraw = raw + 32 * 68;                     // [MAXF][8]: pose of the tile's fragments
  {
    const int row = tid >> 3, part = tid & 7;
    const int64_t grow = min(r0 + row, p.M - 1);
    const int64_t f = grow / p.L;
    const int64_t sf_ = p.slot ? (int64_t)p.slot[f] : f;
    const int64_t srow = sf_ * p.L + (grow - f * p.L);
    const float4 a = *reinterpret_cast<const float4*>(p.latent + srow * 64 + part * 8);
    const float4 b = *reinterpret_cast<const float4*>(p.latent + srow * 64 + part * 8 + 4);
    float extra = 0.0f;
    if (part < 3) extra = p.xyz[srow * 3 + part];
    else if (part == 3) extra = p.scale[sf_];
    float xv = 0.0f;
    if (tid < (f1 - f0 + 1) * 8 && (tid & 7) < 7) {
      const int64_t fs = p.slot ? (int64_t)p.slot[f0 + (tid >> 3)] : (int64_t)(f0 + (tid >> 3));
      xv = p.x[fs * 7 + (tid & 7)];
    }
    *reinterpret_cast<float4*>(raw + row * 68 + part * 8) = a;
    *reinterpret_cast<float4*>(raw + row * 68 + part * 8 + 4) = b;
    if (part < 4) raw[row * 68 + 64 + part] = extra;
    if (tid < MAXF * 8) xraw[tid] = xv;
  }
  __syncthreads();
  // pose features once per fragment of the tile: the identity block, then one sincosf per (component, frequency) — sin and cos of an
  // argument are neighbouring blocks of the encoding
  for (int i = tid; i < (f1 - f0 + 1) * 80; i += 256) {
    const int fl_ = i / 80, j = i - fl_ * 80;
    const float* xv = xraw + fl_ * 8;
    float* pf_ = posef + fl_ * FS;
    if (j < 70) {
      const int fi = j / 7, comp = j - fi * 7;
      float sn, cs;
      sincosf(xv[comp] * (float)(1 << fi), &sn, &cs);
      pf_[(1 + 2 * fi) * 7 + comp] = sn;
      pf_[(2 + 2 * fi) * 7 + comp] = cs;
    } else if (j < 77) {
      pf_[j - 70] = xv[j - 70];
    } else if (j == 77) {
      pf_[FP] = 0.0f;
    }
  }
  // shape features: thread -> row tid % 32, items tid / 32, + 8, ...: 40 (component, frequency) pairs and the 68 copied columns
  {
    const int row = tid & 31;
    const float* rr = raw + row * 68;
    _Float16* rh = sh + row * LKP;
    _Float16* rl = sl + row * LKP;
    auto put = [&](int c, float v) {
      _Float16 hi, lo;
      PFPP_SPLIT_TO(v, hi, lo);
      rh[c] = hi;
      rl[c] = lo;
    };
    for (int j = tid >> 5; j < 40; j += 8) {
      float sn, cs;
      if (j < 30) {
        const int fi = j / 3, comp = j - fi * 3;
        sincosf(rr[64 + comp] * (float)(1 << fi), &sn, &cs);
        put(64 + (1 + 2 * fi) * 3 + comp, sn);
        put(64 + (2 + 2 * fi) * 3 + comp, cs);
      } else {
        const int fi = j - 30;
        sincosf(rr[67] * (float)(1 << fi), &sn, &cs);
        put(127 + 1 + 2 * fi, sn);
        put(127 + 2 + 2 * fi, cs);
      }
    }
    for (int c = tid >> 5; c < 68; c += 8) put(c < 67 ? c : 127, rr[c]);       // latent, xyz (identity block), scale (identity block)
  }
  __syncthreads();
  {
    const int row = tid & 31;
    const int f = (int)(min(r0 + row, p.M - 1) / p.L) - f0;
    for (int c = tid >> 5; c < KE - FS; c += 8) {
      const float v = c < FS ? posef[f * FS + c] : 0.0f;
      _Float16 hi, lo;
      PFPP_SPLIT_TO(v, hi, lo);
      sh[row * LKP + FS + c] = hi;
      sl[row * LKP + FS + c] = lo;
    }
  }
  __syncthreads();
  if (!any) return;

  const half8* ah = reinterpret_cast<const half8*>(sh + l31 * LKP + 8 * lhi);
  const half8* al = reinterpret_cast<const half8*>(sl + l31 * LKP + 8 * lhi);
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int i = 0; i < NS / 2; ++i) {
      const int s = (NS / 2) * b + i;
      const half8 a_h = ah[2 * s], a_l = al[2 * s];
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_l, st[b][0][i], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h, st[b][1][i], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h, st[b][0][i], acc, 0, 0, 0);
    }
  const int col = 32 * u + l31;
  const float bb = p.bias[col];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = r0 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
    if (row < p.M) {
      const int f = row / p.L;
      const int64_t src = p.slot ? (int64_t)p.slot[f] : (int64_t)f;
      const float re = p.ref_emb[(p.ref_part[src] ? 1 : 0) * p.C + col];
      const float pp = p.pe[(int64_t)p.frag_pos[f] * p.C + col];
      p.tok[(int64_t)row * p.C + col] = ((acc[e] * p.inv_scale + bb) + re) + pp;
    }
  }
}

}  // namespace

extern "C" int pfpp_embed_tokens_small(const float* latent, const float* xyz, const float* scale, const float* x, const int32_t* slot,
                                       const pfpp_pw* w_cat, const float* bias, const float* ref_emb, const uint8_t* ref_part, const float* pe,
                                       const int32_t* frag_pos, float* tok, int64_t n, int64_t L, int64_t C, pfpp_stream_t stream) {
  PFPP_REQUIRE(pfpp::aligned16(latent), "latent: 16-byte alignment");
  PFPP_REQUIRE(latent && xyz && scale && x && w_cat && bias && ref_emb && ref_part && pe && frag_pos && tok, "null pointer");
  PFPP_REQUIRE(w_cat->fhi && w_cat->flo && pfpp::aligned16(w_cat->fhi) && pfpp::aligned16(w_cat->flo),
               "the concatenated weight's fragment-blocked planes (pfpp_pw.fhi / flo, K = 320) are required");
  PFPP_SUPPORTED(C % 32 == 0 && C >= 32 && L >= 11 && n >= 0 && n * L <= 0x7fffffff, "C % 32 != 0 or fewer than 11 tokens per fragment");
  if (n == 0) return PFPP_OK;
  EsP p;
  p.latent = latent; p.xyz = xyz; p.scale = scale; p.x = x; p.slot = slot;
  p.fh = (const half8*)w_cat->fhi; p.fl = (const half8*)w_cat->flo; p.inv_scale = 1.0f / w_cat->scale;
  p.bias = bias; p.ref_emb = ref_emb; p.ref_part = ref_part; p.pe = pe; p.frag_pos = frag_pos; p.tok = tok;
  p.M = (int)(n * L); p.L = (int)L; p.C = (int)C;
  const size_t smem = (size_t)2 * 32 * LKP * sizeof(_Float16) + (size_t)(MAXF * FS + 32 * 68 + MAXF * 8) * sizeof(float);
  const dim3 grid((unsigned)((p.M + 31) / 32), (unsigned)((C / 32 + 3) / 4));
  hipLaunchKernelGGL(embed_small_kernel, grid, dim3(256), smem, pfpp::as_stream(stream), p);
  return pfpp::check_launch(__func__);
}
