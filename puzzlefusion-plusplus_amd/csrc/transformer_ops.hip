// Non-GEMM kernels of the Denoiser / Verifier transformers and the DDPM update
// (gfx950).  All of them are HBM-bound streaming or tiny per-row work; the dense
// contractions go through pfpp_gemm.
#include "pfpp_common.h"

namespace {

// ---------------------------------------------------------------------------
// a9: NeRF positional features (EmbedderNerf.embed, utils/model_utils.py:68-69):
//   PE(v) = [v | sin(2^0 v) | cos(2^0 v) | ... | sin(2^9 v) | cos(2^9 v)],
// each block as wide as v.  v*2^k is exact in fp32; sinf/cosf are the accurate
// (range-reduced) device functions.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float nerf_pe(const float* v, int d, int c) {
  // c in [0, 21*d): block = c / d, component = c % d
  const int blk = c / d, comp = c - blk * d;
  const float x = v[comp];
  if (blk == 0) return x;
  const int fi = (blk - 1) >> 1;
  const float arg = x * (float)(1 << fi);
  return ((blk - 1) & 1) ? cosf(arg) : sinf(arg);
}

constexpr int TOK_LD = 148;

__global__ __launch_bounds__(256) void token_features_kernel(
    const float* __restrict__ latent, const float* __restrict__ xyz,
    const float* __restrict__ scale, const float* __restrict__ x,
    float* __restrict__ shape_feat, float* __restrict__ pose_feat, int64_t n, int L, const int32_t* __restrict__ slot) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n_shape = n * L * TOK_LD;
  if (gid < n_shape) {
    const int64_t orow = gid / TOK_LD;    // (listed fragment, l)
    const int c = (int)(gid - orow * TOK_LD);
    const int64_t f = orow / L;
    const int64_t row = slot ? (int64_t)slot[f] * L + (orow - f * L) : orow;     // (source fragment slot, l)
    float v;
    if (c < 64) {
      v = latent[row * 64 + c];
    } else if (c < 64 + 63) {
      v = nerf_pe(xyz + row * 3, 3, c - 64);
    } else if (c < 64 + 63 + 21) {
      const float s = scale[row / L];
      v = nerf_pe(&s, 1, c - 127);
    } else {
      v = 0.0f;
    }
    shape_feat[gid] = v;
    return;
  }
  const int64_t g2 = gid - n_shape;
  if (g2 < n * TOK_LD) {
    const int64_t orow = g2 / TOK_LD;
    const int c = (int)(g2 - orow * TOK_LD);
    const int64_t row = slot ? (int64_t)slot[orow] : orow;
    pose_feat[g2] = c < 147 ? nerf_pe(x + row * 7, 7, c) : 0.0f;
  }
}

// tok = shape_emb + x_emb[bp] + ref_emb[ref] + pe[p]; evaluation order of the
// reference: (x_emb + ref_emb) then + shape_emb then + pe
// (denoiser_transformer.py:155,183-185)
__global__ __launch_bounds__(256) void token_combine_kernel(
    const float* __restrict__ shape_emb, const float* __restrict__ x_emb,
    const float* __restrict__ ref_emb, const uint8_t* __restrict__ ref_part,
    const float* __restrict__ pe, const int32_t* __restrict__ frag_pos, float* __restrict__ tok,
    int64_t total4, int P, int L, int C4, const int32_t* __restrict__ slot) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total4) return;
  const int64_t row = gid / C4;          // (b,p,l)
  const int c4 = (int)(gid - row * C4);
  const int64_t bp = row / L;
  const int p = frag_pos ? frag_pos[bp] : (int)(bp % P);
  const float4 s = reinterpret_cast<const float4*>(shape_emb)[gid];
  const float4 xe = reinterpret_cast<const float4*>(x_emb)[bp * C4 + c4];
  const float4 re = reinterpret_cast<const float4*>(ref_emb)[(ref_part[slot ? slot[bp] : bp] ? 1 : 0) * C4 + c4];
  const float4 pp = reinterpret_cast<const float4*>(pe)[p * C4 + c4];
  float4 o;
  o.x = ((xe.x + re.x) + s.x) + pp.x;
  o.y = ((xe.y + re.y) + s.y) + pp.y;
  o.z = ((xe.z + re.z) + s.z) + pp.z;
  o.w = ((xe.w + re.w) + s.w) + pp.w;
  reinterpret_cast<float4*>(tok)[gid] = o;
}

// ---------------------------------------------------------------------------
// a11: AdaLN
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void silu_embed_kernel(const float* __restrict__ tables,
                                                         const int64_t* __restrict__ t,
                                                         float* __restrict__ out, int64_t total,
                                                         int64_t n_emb, int B, int C) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int c = (int)(gid % C);
  const int64_t ib = gid / C;
  const int b = (int)(ib % B);
  const int64_t i = ib / B;
  const float v = tables[(i * n_emb + t[b]) * C + c];
  out[gid] = v / (1.0f + expf(-v));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}

// one wave per row; VPL float4 per lane (C = 256*VPL)
template <int VPL, bool SPLIT = false>
__global__ __launch_bounds__(256) void layernorm_kernel(
    const float* __restrict__ x, float* __restrict__ y, _Float16* __restrict__ y_hi, _Float16* __restrict__ y_lo,
    const float* __restrict__ mod,
    int64_t ld_mod, const float* __restrict__ gamma, const float* __restrict__ beta, int64_t rows,
    int rows_per_batch, float eps, const int32_t* __restrict__ group_batch = nullptr, int group_rows = 1) {
  pfpp_chain_prio();
  constexpr int C = 256 * VPL;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float4* xr = reinterpret_cast<const float4*>(x + row * C);
  float4 v[VPL];
  // the row first: nothing below depends on it until the reductions, and the batch index / modulation rows are a dependent chain
  // of their own (index load -> address -> rows) that would otherwise sit in front of it
#pragma unroll
  for (int k = 0; k < VPL; ++k) v[k] = xr[lane + 64 * k];
  // the (scale, shift) / (gamma, beta) rows do not depend on the reductions: fetched with the input row, in one branch-free sequence
  const int64_t grp = rows <= 0x7fffffffll ? (int64_t)((uint32_t)row / (uint32_t)(group_batch ? group_rows : rows_per_batch))
                                           : row / (group_batch ? group_rows : rows_per_batch);
  const int64_t b = group_batch ? (int64_t)group_batch[grp] : grp;
  float4 m_a[VPL], m_b[VPL];
  {
    const float* pa = mod ? mod + b * ld_mod : gamma;
    const float* pb = mod ? mod + b * ld_mod + C : beta;
    if (pa) {
#pragma unroll
      for (int k = 0; k < VPL; ++k) {
        m_a[k] = reinterpret_cast<const float4*>(pa)[lane + 64 * k];
        m_b[k] = reinterpret_cast<const float4*>(pb)[lane + 64 * k];
      }
    }
  }
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < VPL; ++k) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
  const float mean = wave_sum(s) / (float)C;
  float q = 0.0f;
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    const float a = v[k].x - mean, b = v[k].y - mean, c = v[k].z - mean, d = v[k].w - mean;
    q += (a * a + b * b) + (c * c + d * d);
  }
  const float var = wave_sum(q) / (float)C;
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    const int c4 = lane + 64 * k;
    float4 o;
    o.x = (v[k].x - mean) * rstd;
    o.y = (v[k].y - mean) * rstd;
    o.z = (v[k].z - mean) * rstd;
    o.w = (v[k].w - mean) * rstd;
    if (mod) {
      const float4 sc = m_a[k], sh = m_b[k];
      o.x = o.x * (1.0f + sc.x) + sh.x;
      o.y = o.y * (1.0f + sc.y) + sh.y;
      o.z = o.z * (1.0f + sc.z) + sh.z;
      o.w = o.w * (1.0f + sc.w) + sh.w;
    } else if (gamma) {
      const float4 g = m_a[k], be = m_b[k];
      o.x = o.x * g.x + be.x;
      o.y = o.y * g.y + be.y;
      o.z = o.z * g.z + be.z;
      o.w = o.w * g.w + be.w;
    }
    if constexpr (SPLIT) {
      typedef _Float16 half4 __attribute__((ext_vector_type(4)));
      half4 hi, lo;
      PFPP_SPLIT_TO(o.x, hi[0], lo[0]);
      PFPP_SPLIT_TO(o.y, hi[1], lo[1]);
      PFPP_SPLIT_TO(o.z, hi[2], lo[2]);
      PFPP_SPLIT_TO(o.w, hi[3], lo[3]);
      reinterpret_cast<half4*>(y_hi + row * C)[c4] = hi;
      reinterpret_cast<half4*>(y_lo + row * C)[c4] = lo;
    } else {
      reinterpret_cast<float4*>(y + row * C)[c4] = o;
    }
  }
}

// ---------------------------------------------------------------------------
// a12: block-diagonal self-attention.  One wave per (fragment, head); lane =
// query row (L <= 32 of the 64 lanes are busy), K and V of the fragment/head sit
// in LDS and every lane reads the same key at the same time (LDS broadcast).
// 80 kMAC per wave: negligible next to the projections, so plain VALU.
// ---------------------------------------------------------------------------
constexpr int AB_LMAX = 32;
constexpr int AB_DH = 64;

__global__ __launch_bounds__(256) void attn_blockdiag_kernel(const float* __restrict__ qkv,
                                                             float* __restrict__ out,
                                                             _Float16* __restrict__ out_hi,
                                                             _Float16* __restrict__ out_lo,
                                                             int64_t n_pairs, int L, int H,
                                                             float scale) {
  extern __shared__ __align__(16) float ab_smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  float* sk = ab_smem + wave * (2 * AB_LMAX * AB_DH);
  float* sv = sk + AB_LMAX * AB_DH;
  const int64_t pair = (int64_t)blockIdx.x * 4 + wave;   // frag * H + h
  if (pair >= n_pairs) return;                           // whole wave exits together
  const int64_t frag = pair / H;
  const int h = (int)(pair - frag * H);
  const int64_t ld = 3ll * H * AB_DH;
  const float* base = qkv + frag * L * ld + h * AB_DH;
  // stage K and V: L rows x 16 float4
  for (int i = lane; i < L * 16; i += 64) {
    const int r = i >> 4, c4 = i & 15;
    reinterpret_cast<float4*>(sk)[r * 16 + c4] =
        *reinterpret_cast<const float4*>(base + r * ld + H * AB_DH + c4 * 4);
    reinterpret_cast<float4*>(sv)[r * 16 + c4] =
        *reinterpret_cast<const float4*>(base + r * ld + 2 * H * AB_DH + c4 * 4);
  }
  __builtin_amdgcn_s_waitcnt(0);  // LDS writes of this wave visible to itself
  __builtin_amdgcn_wave_barrier();
  const bool active = lane < L;
  const int qi = active ? lane : 0;
  float q[AB_DH];
#pragma unroll
  for (int c4 = 0; c4 < 16; ++c4) {
    const float4 v = *reinterpret_cast<const float4*>(base + qi * ld + c4 * 4);
    q[4 * c4 + 0] = v.x; q[4 * c4 + 1] = v.y; q[4 * c4 + 2] = v.z; q[4 * c4 + 3] = v.w;
  }
  float s[AB_LMAX];
  float m = -__builtin_huge_valf();
#pragma unroll
  for (int j = 0; j < AB_LMAX; ++j) {
    s[j] = -__builtin_huge_valf();
    if (j < L) {
      float acc = 0.0f;
#pragma unroll
      for (int c4 = 0; c4 < 16; ++c4) {
        const float4 k = reinterpret_cast<const float4*>(sk)[j * 16 + c4];
        acc = fmaf(q[4 * c4 + 0], k.x, acc);
        acc = fmaf(q[4 * c4 + 1], k.y, acc);
        acc = fmaf(q[4 * c4 + 2], k.z, acc);
        acc = fmaf(q[4 * c4 + 3], k.w, acc);
      }
      s[j] = acc * scale;
      m = fmaxf(m, s[j]);
    }
  }
  float sum = 0.0f;
#pragma unroll
  for (int j = 0; j < AB_LMAX; ++j) {
    if (j < L) {
      s[j] = expf(s[j] - m);
      sum += s[j];
    }
  }
  const float inv = 1.0f / sum;
  float o[AB_DH];
#pragma unroll
  for (int d = 0; d < AB_DH; ++d) o[d] = 0.0f;
#pragma unroll
  for (int j = 0; j < AB_LMAX; ++j) {
    if (j < L) {
      const float pj = s[j] * inv;
#pragma unroll
      for (int c4 = 0; c4 < 16; ++c4) {
        const float4 v = reinterpret_cast<const float4*>(sv)[j * 16 + c4];
        o[4 * c4 + 0] = fmaf(pj, v.x, o[4 * c4 + 0]);
        o[4 * c4 + 1] = fmaf(pj, v.y, o[4 * c4 + 1]);
        o[4 * c4 + 2] = fmaf(pj, v.z, o[4 * c4 + 2]);
        o[4 * c4 + 3] = fmaf(pj, v.w, o[4 * c4 + 3]);
      }
    }
  }
  if (active) {
    const int64_t off = (frag * L + lane) * (int64_t)(H * AB_DH) + h * AB_DH;
    if (out_hi) {
      typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#pragma unroll
      for (int c8 = 0; c8 < 8; ++c8) {
        half8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) PFPP_SPLIT_TO(o[8 * c8 + e], hi[e], lo[e]);
        *reinterpret_cast<half8*>(out_hi + off + c8 * 8) = hi;
        *reinterpret_cast<half8*>(out_lo + off + c8 * 8) = lo;
      }
    } else {
      float* op = out + off;
#pragma unroll
      for (int c4 = 0; c4 < 16; ++c4)
        *reinterpret_cast<float4*>(op + c4 * 4) =
            make_float4(o[4 * c4 + 0], o[4 * c4 + 1], o[4 * c4 + 2], o[4 * c4 + 3]);
    }
  }
}

// The same on the matrix cores (exact fp32 products, v_mfma_f32_32x32x2_f32), no LDS: one wave per (fragment, head).
//   S^T = K . Q^T   keys x queries: lane = query, 16 registers = the keys of this lane half; the softmax over a query's keys is
//                   16 in-lane values + one exchange with lane ^ 32
//   O^T = V^T . P^T the normalised accumulator IS the B operand (key slot t of lane half lhi = register t), V is read
//                   straight from global memory, one coalesced row segment per key
// 64 MFMAs per pair (4096 cycles) against ~4800 dependent FMAs per lane with 25 of 64 lanes working: 23 -> 8 us.
typedef float abm_f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void attn_blockdiag_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                  _Float16* __restrict__ out_hi, _Float16* __restrict__ out_lo,
                                                                  int64_t n_pairs, int L, int H, float scale) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int64_t pair = (int64_t)blockIdx.x * 4 + wave;
  if (pair >= n_pairs) return;
  const int64_t frag = pair / H;
  const int h = (int)(pair - frag * H);
  const int C = H * AB_DH;
  const int64_t ld = 3ll * C;
  const float* base = qkv + frag * L * ld + h * AB_DH;
  const int row = l31 < L ? l31 : L - 1;
  const float* qp = base + row * ld + lhi * 4;
  const float* kp = qp + C;

  abm_f32x16 s;
#pragma unroll
  for (int e = 0; e < 16; ++e) s[e] = 0.0f;
#pragma unroll
  for (int c = 0; c < AB_DH / 8; ++c) {
    const float4 q4 = *reinterpret_cast<const float4*>(qp + c * 8);
    const float4 k4 = *reinterpret_cast<const float4*>(kp + c * 8);
    s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.x, q4.x, s, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.y, q4.y, s, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.z, q4.z, s, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.w, q4.w, s, 0, 0, 0);
  }
  float mx = -__builtin_huge_valf();
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int key = (e & 3) + 8 * (e >> 2) + 4 * lhi;
    s[e] = key < L ? s[e] * scale : -__builtin_huge_valf();
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

  const float* vp = base + 2 * C + l31;
  const int64_t orow = (frag * L + l31) * (int64_t)C + h * AB_DH;
#pragma unroll
  for (int tile = 0; tile < AB_DH / 32; ++tile) {
    abm_f32x16 o;
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] = 0.0f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int key = (t & 3) + 8 * (t >> 2) + 4 * lhi;
      const float v = vp[(int64_t)(key < L ? key : L - 1) * ld + tile * 32];      // masked keys carry p = 0
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(v, s[t], o, 0, 0, 0);
    }
    if (l31 < L) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t off = orow + tile * 32 + 8 * q + 4 * lhi;
        if (out_hi) {
#pragma unroll
          for (int r = 0; r < 4; ++r) PFPP_SPLIT_TO(o[4 * q + r], out_hi[off + r], out_lo[off + r]);
        } else {
          *reinterpret_cast<float4*>(out + off) = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// a13/a18: masked row softmax, one wave per row, three passes over a row that
// stays in L1/L2 (T <= a few thousand)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ S,
                                                           const uint8_t* __restrict__ key_valid,
                                                           int64_t rows_total, int rows_per_batch,
                                                           int T, int ld, float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows_total) return;
  float* s = S + row * ld;
  const uint8_t* kv = key_valid ? key_valid + (row / rows_per_batch) * T : nullptr;
  float m = -__builtin_huge_valf();
  for (int j = lane; j < T; j += 64)
    if (!kv || kv[j]) m = fmaxf(m, s[j] * scale);
  m = wave_max(m);
  float sum = 0.0f;
  for (int j = lane; j < T; j += 64) {
    float e = 0.0f;
    if (!kv || kv[j]) e = expf(s[j] * scale - m);
    s[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  const float inv = sum > 0.0f ? 1.0f / sum : 0.0f;
  for (int j = lane; j < ld; j += 64) s[j] = j < T ? s[j] * inv : 0.0f;
}

// a15: mean over the L tokens of a fragment (sum in l order, then / L)
__global__ __launch_bounds__(256) void mean_pool_kernel(const float* __restrict__ x,
                                                        float* __restrict__ out, int64_t total,
                                                        int L, int C) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int64_t n = gid / C;
  const int c = (int)(gid - n * C);
  const float* p = x + n * L * C + c;
  float s = 0.0f;
  for (int l = 0; l < L; ++l) s += p[(int64_t)l * C];
  out[gid] = s / (float)L;
}

// ---------------------------------------------------------------------------
// a16: scheduler
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ddpm_step_kernel(
    const float* __restrict__ x, const float* __restrict__ eps, const float* __restrict__ noise,
    const uint8_t* __restrict__ ref_part, const float* __restrict__ reference,
    float* __restrict__ out, int64_t total, float c_eps, float c_div, float c_x0, float c_x,
    float c_noise) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int64_t row = gid / 7;
  float v;
  if (ref_part && ref_part[row]) {
    v = reference[gid];
  } else {
    const float xv = x[gid];
    const float x0 = (xv - c_eps * eps[gid]) / c_div;  // -ffp-contract=off; IEEE division
    v = __fadd_rn(__fmul_rn(c_x0, x0), __fmul_rn(c_x, xv));
    if (noise) v = __fadd_rn(v, __fmul_rn(c_noise, noise[gid]));
  }
  out[gid] = v;
}

__global__ __launch_bounds__(256) void add_noise_kernel(const float* __restrict__ x0,
                                                        const float* __restrict__ noise,
                                                        const float* __restrict__ sa,
                                                        const float* __restrict__ sb,
                                                        float* __restrict__ out, int64_t total,
                                                        int per_batch) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int64_t b = gid / per_batch;
  out[gid] = __fadd_rn(__fmul_rn(sa[b], x0[gid]), __fmul_rn(sb[b], noise[gid]));
}

// a18: verifier token = feat_emb + [pe[i0] | pe[i1]]
__global__ __launch_bounds__(256) void verifier_embed_kernel(const float* __restrict__ feat_emb,
                                                             const int64_t* __restrict__ edge_idx,
                                                             const float* __restrict__ pe,
                                                             float* __restrict__ tok,
                                                             int64_t total, int C, int max_len) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int64_t row = gid / C;
  const int c = (int)(gid - row * C);
  const int half = C >> 1;
  int64_t node = edge_idx[row * 2 + (c >= half ? 1 : 0)];
  node = node < 0 ? 0 : (node >= max_len ? max_len - 1 : node);
  tok[gid] = pe[node * half + (c >= half ? c - half : c)] + feat_emb[gid];
}

inline unsigned blocks_for(int64_t total, int per) { return (unsigned)((total + per - 1) / per); }

}  // namespace

extern "C" int pfpp_token_features_slots(const float* latent, const float* xyz, const float* scale, const float* x,
                                         const int32_t* slot, float* shape_feat, float* pose_feat, int64_t n, int64_t L,
                                         pfpp_stream_t stream) {
  PFPP_REQUIRE(latent && xyz && scale && x && slot && shape_feat && pose_feat, "null pointer");
  const int64_t total = n * L * TOK_LD + n * TOK_LD;
  if (total == 0) return PFPP_OK;
  hipLaunchKernelGGL(token_features_kernel, dim3(blocks_for(total, 256)), dim3(256), 0,
                     pfpp::as_stream(stream), latent, xyz, scale, x, shape_feat, pose_feat, n, (int)L, slot);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_token_combine_slots(const float* shape_emb, const float* x_emb, const float* ref_emb,
                                        const uint8_t* ref_part, const float* pe, const int32_t* frag_pos, const int32_t* slot,
                                        float* tok, int64_t n, int64_t L, int64_t C, pfpp_stream_t stream) {
  PFPP_REQUIRE(shape_emb && x_emb && ref_emb && ref_part && pe && frag_pos && slot && tok, "null pointer");
  PFPP_REQUIRE(C % 4 == 0, "C % 4 != 0");
  const int64_t total4 = n * L * (C / 4);
  if (total4 == 0) return PFPP_OK;
  hipLaunchKernelGGL(token_combine_kernel, dim3(blocks_for(total4, 256)), dim3(256), 0,
                     pfpp::as_stream(stream), shape_emb, x_emb, ref_emb, ref_part, pe, frag_pos, tok, total4, 1,
                     (int)L, (int)(C / 4), slot);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_token_features(const float* latent, const float* xyz, const float* scale,
                                   const float* x, float* shape_feat, float* pose_feat, int64_t n,
                                   int64_t L, pfpp_stream_t stream) {
  PFPP_REQUIRE(latent && xyz && scale && x && shape_feat && pose_feat, "null pointer");
  const int64_t total = n * L * TOK_LD + n * TOK_LD;
  if (total == 0) return PFPP_OK;
  hipLaunchKernelGGL(token_features_kernel, dim3(blocks_for(total, 256)), dim3(256), 0,
                     pfpp::as_stream(stream), latent, xyz, scale, x, shape_feat, pose_feat, n, (int)L, (const int32_t*)nullptr);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_token_combine(const float* shape_emb, const float* x_emb, const float* ref_emb,
                                  const uint8_t* ref_part, const float* pe, float* tok, int64_t B,
                                  int64_t P, int64_t L, int64_t C, pfpp_stream_t stream) {
  PFPP_REQUIRE(shape_emb && x_emb && ref_emb && ref_part && pe && tok, "null pointer");
  PFPP_REQUIRE(C % 4 == 0, "C % 4 != 0");
  const int64_t total4 = B * P * L * (C / 4);
  if (total4 == 0) return PFPP_OK;
  hipLaunchKernelGGL(token_combine_kernel, dim3(blocks_for(total4, 256)), dim3(256), 0,
                     pfpp::as_stream(stream), shape_emb, x_emb, ref_emb, ref_part, pe, nullptr, tok, total4,
                     (int)P, (int)L, (int)(C / 4), (const int32_t*)nullptr);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_token_combine_list(const float* shape_emb, const float* x_emb, const float* ref_emb,
                                       const uint8_t* ref_part, const float* pe, const int32_t* frag_pos,
                                       float* tok, int64_t n, int64_t L, int64_t C, pfpp_stream_t stream) {
  PFPP_REQUIRE(shape_emb && x_emb && ref_emb && ref_part && pe && frag_pos && tok, "null pointer");
  PFPP_REQUIRE(C % 4 == 0, "C % 4 != 0");
  const int64_t total4 = n * L * (C / 4);
  if (total4 == 0) return PFPP_OK;
  hipLaunchKernelGGL(token_combine_kernel, dim3(blocks_for(total4, 256)), dim3(256), 0,
                     pfpp::as_stream(stream), shape_emb, x_emb, ref_emb, ref_part, pe, frag_pos, tok, total4, 1,
                     (int)L, (int)(C / 4), (const int32_t*)nullptr);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_silu_embed(const float* tables, const int64_t* t, float* out, int64_t n_tab,
                               int64_t n_emb, int64_t B, int64_t C, pfpp_stream_t stream) {
  PFPP_REQUIRE(tables && t && out, "null pointer");
  const int64_t total = n_tab * B * C;
  if (total == 0) return PFPP_OK;
  hipLaunchKernelGGL(silu_embed_kernel, dim3(blocks_for(total, 256)), dim3(256), 0,
                     pfpp::as_stream(stream), tables, t, out, total, n_emb, (int)B, (int)C);
  return pfpp::check_launch(__func__);
}

static int layernorm_impl(const float* x, float* y, _Float16* y_hi, _Float16* y_lo, const float* mod,
                          int64_t ld_mod, const float* gamma, const float* beta, int64_t rows, int64_t C,
                          int64_t rows_per_batch, float eps, pfpp_stream_t stream,
                          const int32_t* group_batch = nullptr, int64_t group_rows = 1);

extern "C" int pfpp_layernorm_grouped(const float* x, float* y, const float* mod, int64_t ld_mod,
                                      const int32_t* group_batch, int64_t group_rows, int64_t rows, int64_t C,
                                      float eps, pfpp_stream_t stream) {
  PFPP_REQUIRE(x && y && mod && group_batch && group_rows >= 1, "null pointer / bad group size");
  return layernorm_impl(x, y, nullptr, nullptr, mod, ld_mod, nullptr, nullptr, rows, C, 1, eps, stream, group_batch,
                        group_rows);
}

extern "C" int pfpp_layernorm_grouped_split(const float* x, void* y_hi, void* y_lo, const float* mod, int64_t ld_mod,
                                            const int32_t* group_batch, int64_t group_rows, int64_t rows, int64_t C,
                                            float eps, pfpp_stream_t stream) {
  PFPP_REQUIRE(x && y_hi && y_lo && mod && group_batch && group_rows >= 1, "null pointer / bad group size");
  return layernorm_impl(x, nullptr, (_Float16*)y_hi, (_Float16*)y_lo, mod, ld_mod, nullptr, nullptr, rows, C, 1, eps, stream,
                        group_batch, group_rows);
}

extern "C" int pfpp_layernorm(const float* x, float* y, const float* mod, int64_t ld_mod,
                              const float* gamma, const float* beta, int64_t rows, int64_t C,
                              int64_t rows_per_batch, float eps, pfpp_stream_t stream) {
  PFPP_REQUIRE(x && y, "null pointer");
  return layernorm_impl(x, y, nullptr, nullptr, mod, ld_mod, gamma, beta, rows, C, rows_per_batch, eps, stream);
}

extern "C" int pfpp_layernorm_split(const float* x, void* y_hi, void* y_lo, const float* mod, int64_t ld_mod,
                                    const float* gamma, const float* beta, int64_t rows, int64_t C,
                                    int64_t rows_per_batch, float eps, pfpp_stream_t stream) {
  PFPP_REQUIRE(x && y_hi && y_lo, "null pointer");
  return layernorm_impl(x, nullptr, (_Float16*)y_hi, (_Float16*)y_lo, mod, ld_mod, gamma, beta, rows, C,
                        rows_per_batch, eps, stream);
}

static int layernorm_impl(const float* x, float* y, _Float16* y_hi, _Float16* y_lo, const float* mod,
                          int64_t ld_mod, const float* gamma, const float* beta, int64_t rows, int64_t C,
                          int64_t rows_per_batch, float eps, pfpp_stream_t stream, const int32_t* group_batch,
                          int64_t group_rows) {
  const char* __func__name = "pfpp_layernorm";
  (void)__func__name;
  PFPP_REQUIRE(!gamma || beta, "gamma without beta");
  PFPP_REQUIRE(rows_per_batch >= 1, "rows_per_batch < 1");
  PFPP_SUPPORTED(C == 256 || C == 512 || C == 1024, "C not in {256, 512, 1024}");
  if (rows == 0) return PFPP_OK;
  hipStream_t st = pfpp::as_stream(stream);
  const dim3 grid(blocks_for(rows, 4));
  const int rpb = (int)rows_per_batch;
  const int gr = (int)group_rows;
  if (y_hi) {
    if (C == 256) hipLaunchKernelGGL((layernorm_kernel<1, true>), grid, dim3(256), 0, st, x, y, y_hi, y_lo, mod, ld_mod, gamma, beta, rows, rpb, eps, group_batch, gr);
    else if (C == 512) hipLaunchKernelGGL((layernorm_kernel<2, true>), grid, dim3(256), 0, st, x, y, y_hi, y_lo, mod, ld_mod, gamma, beta, rows, rpb, eps, group_batch, gr);
    else hipLaunchKernelGGL((layernorm_kernel<4, true>), grid, dim3(256), 0, st, x, y, y_hi, y_lo, mod, ld_mod, gamma, beta, rows, rpb, eps, group_batch, gr);
  } else {
    if (C == 256) hipLaunchKernelGGL((layernorm_kernel<1, false>), grid, dim3(256), 0, st, x, y, y_hi, y_lo, mod, ld_mod, gamma, beta, rows, rpb, eps, group_batch, gr);
    else if (C == 512) hipLaunchKernelGGL((layernorm_kernel<2, false>), grid, dim3(256), 0, st, x, y, y_hi, y_lo, mod, ld_mod, gamma, beta, rows, rpb, eps, group_batch, gr);
    else hipLaunchKernelGGL((layernorm_kernel<4, false>), grid, dim3(256), 0, st, x, y, y_hi, y_lo, mod, ld_mod, gamma, beta, rows, rpb, eps, group_batch, gr);
  }
  return pfpp::check_launch("pfpp_layernorm");
}

static int attn_blockdiag_impl(const float* qkv, float* out, _Float16* out_hi, _Float16* out_lo, int64_t n_frag,
                               int64_t L, int64_t H, int64_t dh, float scale, pfpp_stream_t stream);

extern "C" int pfpp_attn_blockdiag(const float* qkv, float* out, int64_t n_frag, int64_t L,
                                   int64_t H, int64_t dh, float scale, pfpp_stream_t stream) {
  PFPP_REQUIRE(qkv && out, "null pointer");
  return attn_blockdiag_impl(qkv, out, nullptr, nullptr, n_frag, L, H, dh, scale, stream);
}

extern "C" int pfpp_attn_blockdiag_split(const float* qkv, void* out_hi, void* out_lo, int64_t n_frag, int64_t L,
                                         int64_t H, int64_t dh, float scale, pfpp_stream_t stream) {
  PFPP_REQUIRE(qkv && out_hi && out_lo, "null pointer");
  return attn_blockdiag_impl(qkv, nullptr, (_Float16*)out_hi, (_Float16*)out_lo, n_frag, L, H, dh, scale, stream);
}

static int attn_blockdiag_impl(const float* qkv, float* out, _Float16* out_hi, _Float16* out_lo, int64_t n_frag,
                               int64_t L, int64_t H, int64_t dh, float scale, pfpp_stream_t stream) {
  PFPP_SUPPORTED(dh == AB_DH, "dim_head != 64");
  PFPP_SUPPORTED(L >= 1 && L <= AB_LMAX, "L outside [1, 32]");
  PFPP_REQUIRE(pfpp::aligned16(qkv) && pfpp::aligned16(out) && pfpp::aligned16(out_hi) && pfpp::aligned16(out_lo),
               "16-byte alignment");
  const int64_t pairs = n_frag * H;
  if (pairs == 0) return PFPP_OK;
  static const bool use_mfma = !(getenv("PFPP_ATTN_BD_MFMA") && atoi(getenv("PFPP_ATTN_BD_MFMA")) == 0);
  // split-f16 contraction (attention_bwd.hip: attn_blockdiag_f16_kernel) unless PFPP_ATTN_BD_F16X3=0 (then the exact fp32 matrix instructions)
  static const bool use_f16 = !(getenv("PFPP_ATTN_BD_F16X3") && atoi(getenv("PFPP_ATTN_BD_F16X3")) == 0);
  if (use_mfma && pfpp::attn_use_f16(use_f16) && dh == 64 && pairs <= 0x7fffffff)
    return pfpp_attn_blockdiag_f16_launch(qkv, out, out_hi, out_lo, pairs, L, H, scale, pfpp::as_stream(stream));
  if (use_mfma) {
    hipLaunchKernelGGL(attn_blockdiag_mfma_kernel, dim3(blocks_for(pairs, 4)), dim3(256), 0, pfpp::as_stream(stream), qkv, out,
                       out_hi, out_lo, pairs, (int)L, (int)H, scale);
    return pfpp::check_launch("pfpp_attn_blockdiag");
  }
  const size_t smem = 4 * 2 * AB_LMAX * AB_DH * sizeof(float);
  hipLaunchKernelGGL(attn_blockdiag_kernel, dim3(blocks_for(pairs, 4)), dim3(256), smem,
                     pfpp::as_stream(stream), qkv, out, out_hi, out_lo, pairs, (int)L, (int)H, scale);
  return pfpp::check_launch("pfpp_attn_blockdiag");
}

extern "C" int pfpp_softmax_rows(float* S, const uint8_t* key_valid, int64_t rows_total,
                                 int64_t rows_per_batch, int64_t T, int64_t ld, float scale,
                                 pfpp_stream_t stream) {
  PFPP_REQUIRE(S, "null pointer");
  PFPP_REQUIRE(T >= 1 && ld >= T && rows_per_batch >= 1, "bad sizes");
  if (rows_total == 0) return PFPP_OK;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(blocks_for(rows_total, 4)), dim3(256), 0,
                     pfpp::as_stream(stream), S, key_valid, rows_total, (int)rows_per_batch, (int)T,
                     (int)ld, scale);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_mean_pool(const float* x, float* out, int64_t n, int64_t L, int64_t C,
                              pfpp_stream_t stream) {
  PFPP_REQUIRE(x && out, "null pointer");
  const int64_t total = n * C;
  if (total == 0) return PFPP_OK;
  hipLaunchKernelGGL(mean_pool_kernel, dim3(blocks_for(total, 256)), dim3(256), 0,
                     pfpp::as_stream(stream), x, out, total, (int)L, (int)C);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_ddpm_step(const float* x, const float* eps, const float* noise,
                              const uint8_t* ref_part, const float* reference, float* out,
                              int64_t n, float c_eps, float c_div, float c_x0, float c_x,
                              float c_noise, pfpp_stream_t stream) {
  PFPP_REQUIRE(x && eps && out, "null pointer");
  PFPP_REQUIRE(!ref_part || reference, "ref_part without reference");
  const int64_t total = n * 7;
  if (total == 0) return PFPP_OK;
  hipLaunchKernelGGL(ddpm_step_kernel, dim3(blocks_for(total, 256)), dim3(256), 0,
                     pfpp::as_stream(stream), x, eps, noise, ref_part, reference, out, total, c_eps,
                     c_div, c_x0, c_x, c_noise);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_add_noise(const float* x0, const float* noise, const float* sqrt_ab,
                              const float* sqrt_1mab, float* out, int64_t B, int64_t per_batch,
                              pfpp_stream_t stream) {
  PFPP_REQUIRE(x0 && noise && sqrt_ab && sqrt_1mab && out, "null pointer");
  const int64_t total = B * per_batch;
  if (total == 0) return PFPP_OK;
  hipLaunchKernelGGL(add_noise_kernel, dim3(blocks_for(total, 256)), dim3(256), 0,
                     pfpp::as_stream(stream), x0, noise, sqrt_ab, sqrt_1mab, out, total,
                     (int)per_batch);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_verifier_embed(const float* feat_emb, const int64_t* edge_idx, const float* pe,
                                   float* tok, int64_t n, int64_t C, int64_t max_len,
                                   pfpp_stream_t stream) {
  PFPP_REQUIRE(feat_emb && edge_idx && pe && tok, "null pointer");
  PFPP_REQUIRE(C % 2 == 0, "C must be even");
  const int64_t total = n * C;
  if (total == 0) return PFPP_OK;
  hipLaunchKernelGGL(verifier_embed_kernel, dim3(blocks_for(total, 256)), dim3(256), 0,
                     pfpp::as_stream(stream), feat_emb, edge_idx, pe, tok, total, (int)C,
                     (int)max_len);
  return pfpp::check_launch(__func__);
}
