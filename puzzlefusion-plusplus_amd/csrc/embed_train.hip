// Training-side backward of the token embedding in ONE launch: the gradients of shape_embedding, param_fc and ref_part_emb.
//
// Reference: DenoiserTransformer._gen_cond / _add_ref_part_emb / forward (denoiser/model/modules/denoiser_transformer.py:117-135, 150-156,
// 173-185):  tok[(f, l)] = W_s . sf[(f, l)] + b_s + W_p . pf[f] + b_p + ref_emb[ref_f] + pe[p_f], sf = [latent | PE(xyz) | PE(scale)] (148),
// pf = PE(x_f) (147).  With dtok = d(loss) / d(tok) [M = fragments x L tokens, C]:
//   g_W_s += dtok^T . sf      g_W_p += (sum_l dtok)^T . pf      g_b_s, g_b_p += sum_m dtok[m]      g_ref[r] += sum over the tokens of ref = r
//
// All five are ONE contraction over the tokens with the EXTENDED feature row
//   F[m] = [ sf[m] (148) | pf[f(m)] (147) | [ref = 0] | [ref = 1] | 1 | 0 ... ]   (320 columns):   G = dtok^T . F   [C, 320]
// (the pose features repeat over a fragment's tokens, which turns the sum over l into part of the contraction).  The forward leaves F
// TRANSPOSED as split-f16 planes FT[k][m] (embed_feat_t_kernel; m padded to a multiple of 16 with zeros), so a feature fragment of
// v_mfma_f32_32x32x16_f16 is one 16-byte load; the dtok fragment is 8 dword loads of 128-byte row pieces, split in registers after the
// power-of-two gradient scale.  A workgroup owns a 32 (features) x 32 (channels) tile of G^T over ALL tokens: its eight waves take the
// 16-token steps round-robin (8 steps of loads in flight each), meet in LDS, and add into the gradient buffers — no atomics, no partial
// slabs, a fixed summation tree.  160 workgroups (320 / 32 x C / 32).
// Replaces: two tiled weight-gradient GEMMs (K = 3,850 / 154) + two adds of their padded results + two column sums + the per-fragment
// token sum with its atomics (7 launches, ~100 us at the end of the backward).
#include "pfpp_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int FS = 148;          // shape features
constexpr int FP = 147;          // pose features
constexpr int K_REF0 = FS + FP;  // 295: indicator of ref_part = 0
constexpr int K_REF1 = K_REF0 + 1;
constexpr int K_ONE = K_REF0 + 2;
constexpr int KE = 320;

// EmbedderNerf.embed (utils/model_utils.py:68-69), as csrc/transformer_ops.hip nerf_pe
__device__ __forceinline__ float et_pe(const float* v, int d, int c) {
  const int blk = c / d, comp = c - blk * d;
  const float x = v[comp];
  if (blk == 0) return x;
  const int fi = (blk - 1) >> 1;
  const float arg = x * (float)(1 << fi);
  return ((blk - 1) & 1) ? cosf(arg) : sinf(arg);
}

struct FtP {
  const float *latent, *xyz, *scale, *x;     // [slots, L, 64], [slots, L, 3], [slots], [slots, 7]
  const int32_t* slot;                       // listed fragment -> slot (or null)
  const uint8_t* ref_part;                   // [slots]
  _Float16 *hi, *lo;                         // [KE, Mp]
  int M, Mp, L;
};

// thread = (feature k, token m), m along the lanes: 128-byte plane writes; the values are token_features_kernel's
__global__ __launch_bounds__(256) void embed_feat_t_kernel(const FtP p) {
  const int m = blockIdx.x * 64 + (threadIdx.x & 63);
  const int k = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (m >= p.Mp) return;
  float v = 0.0f;
  if (m < p.M) {
    const int f = m / p.L;
    const int64_t sl = p.slot ? (int64_t)p.slot[f] : (int64_t)f;
    const int64_t row = sl * p.L + (m - f * p.L);
    if (k < 64) v = p.latent[row * 64 + k];
    else if (k < 127) v = et_pe(p.xyz + row * 3, 3, k - 64);
    else if (k < FS) { const float s = p.scale[sl]; v = et_pe(&s, 1, k - 127); }
    else if (k < K_REF0) v = et_pe(p.x + sl * 7, 7, k - FS);
    else if (k == K_REF0) v = p.ref_part[sl] ? 0.0f : 1.0f;
    else if (k == K_REF1) v = p.ref_part[sl] ? 1.0f : 0.0f;
    else if (k == K_ONE) v = 1.0f;
  }
  _Float16 h, l;
  PFPP_SPLIT_TO(v, h, l);
  p.hi[(int64_t)k * p.Mp + m] = h;
  p.lo[(int64_t)k * p.Mp + m] = l;
}

// [W_shape (148) | W_param (147) | 0] [C, 320] as fragment-blocked split-f16 planes (pfpp_pw.fhi / flo: piece ((n / 32) * 20 + k / 16) * 64 + lane
// = W[32 (n / 32) + lane % 32][16 (k / 16) + 8 (lane / 32) .. + 8)), the operand of pfpp_embed_tokens_small, straight from the fp32 parameters
// (training: the weights change every step), and the summed bias
__global__ __launch_bounds__(256) void embed_pack_kernel(const float* __restrict__ ws, const float* __restrict__ wp, const float* __restrict__ bs,
                                                         const float* __restrict__ bp, half8* __restrict__ fh, half8* __restrict__ fl,
                                                         float* __restrict__ bias, int C) {
  const int piece = blockIdx.x * 256 + threadIdx.x;
  if (piece < C) bias[piece] = bs[piece] + bp[piece];
  if (piece >= (C / 32) * (KE / 16) * 64) return;
  const int lane = piece & 63, blk = piece >> 6, nb = blk / (KE / 16), st = blk - nb * (KE / 16);
  const int n = 32 * nb + (lane & 31), k0 = 16 * st + 8 * (lane >> 5);
  half8 h, l;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = k0 + i;
    const float v = k < FS ? ws[(int64_t)n * FS + k] : (k < K_REF0 ? wp[(int64_t)n * FP + (k - FS)] : 0.0f);
    PFPP_SPLIT_TO(v, h[i], l[i]);
  }
  fh[piece] = h;
  fl[piece] = l;
}

struct EbP {
  const float* dtok;                         // [M, C]
  const _Float16 *fh, *fl;                   // [KE, Mp]
  float *g_ws, *g_wp, *g_bs, *g_bp, *g_ref;  // [C, 148], [C, 147], [C], [C], [2, C]   all +=
  float g_scale, inv_scale;
  int M, Mp, C;
};

constexpr int PD = 8;                        // 16-token steps a wave keeps in flight
constexpr int EW = 8;                        // waves per workgroup of embed_bwd_kernel (the token steps round-robin over them)

struct Step { half8 ah, al; float b[8]; };

__global__ __launch_bounds__(64 * EW) void embed_bwd_kernel(const EbP p) {
  __shared__ __align__(16) float red[EW][32][33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int n0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
  const int nsteps = p.Mp / 16;
  const _Float16* fh = p.fh + (int64_t)(k0 + l31) * p.Mp + 8 * lhi;
  const _Float16* fl = p.fl + (int64_t)(k0 + l31) * p.Mp + 8 * lhi;
  const float* dt = p.dtok + n0 + l31;
  auto load = [&](int s, Step& st) {
    if (s < nsteps) {
      st.ah = *reinterpret_cast<const half8*>(fh + 16 * s);
      st.al = *reinterpret_cast<const half8*>(fl + 16 * s);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int m = min(16 * s + 8 * lhi + i, p.M - 1);      // rows past M: their feature columns are zero
        st.b[i] = dt[(int64_t)m * p.C];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) { st.ah[i] = (_Float16)0.0f; st.al[i] = (_Float16)0.0f; st.b[i] = 0.0f; }
    }
  };
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
  auto compute = [&](const Step& st) {
    half8 bh, bl;
#pragma unroll
    for (int i = 0; i < 8; ++i) PFPP_SPLIT_TO(st.b[i] * p.g_scale, bh[i], bl[i]);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(st.al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(st.ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(st.ah, bh, acc, 0, 0, 0);
  };
  Step ring[PD];
#pragma unroll
  for (int i = 0; i < PD; ++i) load(wave + EW * i, ring[i]);
  for (int base = 0; wave + EW * base < nsteps; base += PD) {
#pragma unroll
    for (int i = 0; i < PD; ++i) {
      compute(ring[i]);
      load(wave + EW * (base + PD + i), ring[i]);
    }
  }
  // acc[e]: feature row (e & 3) + 8 (e >> 2) + 4 lhi of the tile, channel l31
#pragma unroll
  for (int e = 0; e < 16; ++e) red[wave][(e & 3) + 8 * (e >> 2) + 4 * lhi][l31] = acc[e];
  __syncthreads();
  if (tid < 256) {
    const int n = n0 + (tid >> 3);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kk = (tid & 7) * 4 + i, k = k0 + kk;
      float v = red[0][kk][tid >> 3];
#pragma unroll
      for (int w = 1; w < EW; ++w) v += red[w][kk][tid >> 3];
      v *= p.inv_scale;
      if (k < FS) p.g_ws[(int64_t)n * FS + k] += v;
      else if (k < K_REF0) p.g_wp[(int64_t)n * FP + (k - FS)] += v;
      else if (k == K_REF0) p.g_ref[n] += v;
      else if (k == K_REF1) p.g_ref[p.C + n] += v;
      else if (k == K_ONE) { p.g_bs[n] += v; p.g_bp[n] += v; }
    }
  }
}

}  // namespace

extern "C" int64_t pfpp_token_features_t_cols(int64_t n, int64_t L) { return (n * L + 15) / 16 * 16; }

extern "C" int pfpp_token_features_t(const float* latent, const float* xyz, const float* scale, const float* x, const int32_t* slot,
                                     const uint8_t* ref_part, void* ft_hi, void* ft_lo, int64_t n, int64_t L, pfpp_stream_t stream) {
  PFPP_REQUIRE(latent && xyz && scale && x && ref_part && ft_hi && ft_lo, "null pointer");
  PFPP_REQUIRE(n >= 1 && L >= 1 && n * L <= 0x7ffffff0, "sizes");
  FtP p;
  p.latent = latent; p.xyz = xyz; p.scale = scale; p.x = x; p.slot = slot; p.ref_part = ref_part;
  p.hi = (_Float16*)ft_hi; p.lo = (_Float16*)ft_lo;
  p.M = (int)(n * L); p.Mp = (int)pfpp_token_features_t_cols(n, L); p.L = (int)L;
  hipLaunchKernelGGL(embed_feat_t_kernel, dim3((unsigned)((p.Mp + 63) / 64), KE / 4), dim3(256), 0, pfpp::as_stream(stream), p);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_embed_pack_weights(const float* w_shape, const float* w_param, const float* b_shape, const float* b_param, void* fhi,
                                       void* flo, float* bias, int64_t C, pfpp_stream_t stream) {
  PFPP_REQUIRE(w_shape && w_param && b_shape && b_param && fhi && flo && bias, "null pointer");
  PFPP_SUPPORTED(C >= 32 && C % 32 == 0 && C <= 0x7fffff, "C % 32 != 0");
  PFPP_REQUIRE(pfpp::aligned16(fhi) && pfpp::aligned16(flo), "16-byte aligned planes");
  const int64_t pieces = (C / 32) * (KE / 16) * 64;
  hipLaunchKernelGGL(embed_pack_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, pfpp::as_stream(stream), w_shape, w_param,
                     b_shape, b_param, (half8*)fhi, (half8*)flo, bias, (int)C);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_token_embed_bwd(const float* dtok, const void* ft_hi, const void* ft_lo, float* g_shape_w, float* g_shape_b,
                                    float* g_param_w, float* g_param_b, float* g_ref_emb, int64_t n, int64_t L, int64_t C, float g_scale,
                                    pfpp_stream_t stream) {
  PFPP_REQUIRE(dtok && ft_hi && ft_lo && g_shape_w && g_shape_b && g_param_w && g_param_b && g_ref_emb, "null pointer");
  PFPP_REQUIRE(n >= 1 && L >= 1 && n * L <= 0x7ffffff0 && g_scale > 0.0f, "sizes / gradient scale");
  PFPP_SUPPORTED(C >= 32 && C % 32 == 0 && C <= 0x7fffffff, "C % 32 != 0");
  PFPP_REQUIRE(pfpp::aligned16(ft_hi) && pfpp::aligned16(ft_lo), "16-byte aligned feature planes");
  EbP p;
  p.dtok = dtok; p.fh = (const _Float16*)ft_hi; p.fl = (const _Float16*)ft_lo;
  p.g_ws = g_shape_w; p.g_wp = g_param_w; p.g_bs = g_shape_b; p.g_bp = g_param_b; p.g_ref = g_ref_emb;
  p.g_scale = g_scale; p.inv_scale = 1.0f / g_scale;
  p.M = (int)(n * L); p.Mp = (int)pfpp_token_features_t_cols(n, L); p.C = (int)C;
  hipLaunchKernelGGL(embed_bwd_kernel, dim3((unsigned)(C / 32), KE / 32), dim3(64 * EW), 0, pfpp::as_stream(stream), p);
  return pfpp::check_launch(__func__);
}
