// a13 / a18: dense masked self-attention, fused (scores never leave the registers).
//
// Replaces  QK^T GEMM -> [B,H,T,T] scores in HBM -> softmax kernel -> P.V GEMM  of the global
// attention of the denoiser (attention.py:82-85, key-padding mask of denoiser_transformer.py:163-164)
// and of the verifier's encoder layers (verifier_transformer.py:62, src_key_padding_mask).
//
// One workgroup = 4 waves = 128 queries of one (sequence, head); each wave owns 32 queries and walks
// the keys in tiles of 32 with an online softmax.  fp32 MFMA (exact products) in the "swapped" form:
//     S^T[key][query] = K_tile . Q^T          (A = K rows from LDS, B = Q held in registers)
//     O^T[d][query]  += V_tile^T . P^T        (A = V from LDS,  B = P)
// With v_mfma_f32_32x32x2_f32 and the lane-half k split used by the GEMM (lanes 0-31 feed k 0-3,
// lanes 32-63 feed k 4-7 of each 8-chunk), accumulator element e = 4*kc + s of S^T is exactly the
// B-operand element the PV product needs for key kc*8 + 4*(lane>>5) + s — so P is consumed in
// place, no LDS round trip or cross-lane shuffle.  Each lane holds ONE query column: the row max /
// sum are in-lane reductions plus one exchange with lane^32, and the online rescale of O is a
// per-lane scalar multiply.  K/V tiles are shared by the 4 waves through a double-buffered LDS
// stage (K rows padded to DH+4 floats: conflict-free 16-byte reads).
// Sequences may have different lengths (seq_off / seq_len): padded fragments can be dropped.
#include <stdlib.h>

#include "pfpp_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int DH>
__global__ __launch_bounds__(256) void attn_dense_kernel(
    const float* __restrict__ qkv, float* __restrict__ out, _Float16* __restrict__ out_hi,
    _Float16* __restrict__ out_lo, const int32_t* __restrict__ seq_off,
    const int32_t* __restrict__ seq_len, const uint8_t* __restrict__ key_valid, int64_t kv_stride,
    int H, float scale, float* __restrict__ lse) {
  constexpr int KT = 32;
  constexpr int LDK = DH + 4;
  constexpr int NCH = DH / 8;      // 8-wide k chunks of the head dimension
  constexpr int NDT = DH / 32;     // 32-wide output column tiles
  constexpr int F4 = KT * DH / 4 / 256;   // float4 per thread and tensor per key tile
  __shared__ __align__(16) float Ks[2][KT * LDK];
  __shared__ __align__(16) float Vs[2][KT * DH];

  const int b = blockIdx.z, h = blockIdx.y;
  const int T = seq_len[b];
  const int q_base = blockIdx.x * 128;
  if (q_base >= T) return;                       // uniform for the workgroup, before any barrier
  const int64_t row0 = seq_off[b];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int C = H * DH;
  const int64_t ld = 3ll * C;
  const float* base = qkv + row0 * ld + h * DH;
  const uint8_t* kv = key_valid ? key_valid + (int64_t)b * kv_stride : nullptr;

  // Q fragments of this lane's query: d = kc*8 + lhi*4 + (0..3)
  const int q_row = q_base + wave * 32 + l31;
  const float* qp = base + (int64_t)min(q_row, T - 1) * ld + lhi * 4;
  float4 qf[NCH];
#pragma unroll
  for (int kc = 0; kc < NCH; ++kc) qf[kc] = *reinterpret_cast<const float4*>(qp + kc * 8);

  f32x16 o_acc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) o_acc[dt][e] = 0.0f;
  float m_run = -1e30f, l_run = 0.0f;

  float4 rk[F4], rv[F4];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int it = 0; it < F4; ++it) {
      const int idx = tid + 256 * it;
      const int r = idx / (DH / 4), c4 = idx % (DH / 4);
      const float* src = base + (int64_t)min(k0 + r, T - 1) * ld + C + c4 * 4;
      rk[it] = *reinterpret_cast<const float4*>(src);
      rv[it] = *reinterpret_cast<const float4*>(src + C);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int it = 0; it < F4; ++it) {
      const int idx = tid + 256 * it;
      const int r = idx / (DH / 4), c4 = idx % (DH / 4);
      *reinterpret_cast<float4*>(&Ks[buf][r * LDK + c4 * 4]) = rk[it];
      *reinterpret_cast<float4*>(&Vs[buf][r * DH + c4 * 4]) = rv[it];
    }
  };

  const int nt = (T + KT - 1) / KT;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    const int k0 = t * KT;
    load_tile(min(k0 + KT, (nt - 1) * KT));   // unconditional (a conditional load makes the compiler wait for it right here)

    // validity of the 32 keys of this tile as a bit mask (uniform)
    const int kidx = k0 + l31;
    const bool kval = kidx < T && (!kv || kv[kidx] != 0);
    const unsigned kmask = (unsigned)(__ballot(kval) & 0xffffffffull);

    // ---- S^T = K . Q^T ---------------------------------------------------------------------------
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.0f;
    const float* kp = &Ks[buf][l31 * LDK + lhi * 4];
#pragma unroll
    for (int kc = 0; kc < NCH; ++kc) {
      const float4 a = *reinterpret_cast<const float4*>(kp + kc * 8);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qf[kc].x, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qf[kc].y, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qf[kc].z, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qf[kc].w, s, 0, 0, 0);
    }

    // ---- online softmax over this lane's query column -------------------------------------------
    float mx = -__builtin_huge_valf();
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = (e & 3) + 8 * (e >> 2) + 4 * lhi;
      const float v = (kmask >> key) & 1u ? s[e] * scale : -__builtin_huge_valf();
      s[e] = v;
      mx = fmaxf(mx, v);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = expf(m_run - m_new);
    float psum = 0.0f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float pe = expf(s[e] - m_new);       // exp(-inf) = 0 for masked keys
      s[e] = pe;
      psum += pe;
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) o_acc[dt][e] *= alpha;

    // ---- O^T += V^T . P^T  (P straight from the score registers) -------------------------------
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const float* vp = &Vs[buf][(lhi * 4) * DH + dt * 32 + l31];
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[(kc * 8 + 0) * DH], s[4 * kc + 0], o_acc[dt], 0, 0, 0);
        o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[(kc * 8 + 1) * DH], s[4 * kc + 1], o_acc[dt], 0, 0, 0);
        o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[(kc * 8 + 2) * DH], s[4 * kc + 2], o_acc[dt], 0, 0, 0);
        o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[(kc * 8 + 3) * DH], s[4 * kc + 3], o_acc[dt], 0, 0, 0);
      }
    }
    if (t + 1 < nt) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- normalise and store: lane holds O[query = l31][d = dt*32 + 8g + 4*lhi + (0..3)] ------------
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
  if (lse && q_row < T && lhi == 0) lse[(row0 + q_row) * H + h] = m_run + logf(l_tot);   // training: saved for the backward
  if (q_row < T) {
    const int64_t off = (row0 + q_row) * (int64_t)C + h * DH + lhi * 4;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float v0 = o_acc[dt][4 * g + 0] * inv, v1 = o_acc[dt][4 * g + 1] * inv;
        const float v2 = o_acc[dt][4 * g + 2] * inv, v3 = o_acc[dt][4 * g + 3] * inv;
        if (out_hi) {
          typedef _Float16 half4 __attribute__((ext_vector_type(4)));
          half4 hi, lo;
          PFPP_SPLIT_TO(v0, hi[0], lo[0]);
          PFPP_SPLIT_TO(v1, hi[1], lo[1]);
          PFPP_SPLIT_TO(v2, hi[2], lo[2]);
          PFPP_SPLIT_TO(v3, hi[3], lo[3]);
          *reinterpret_cast<half4*>(out_hi + off + dt * 32 + 8 * g) = hi;
          *reinterpret_cast<half4*>(out_lo + off + dt * 32 + 8 * g) = lo;
        }
        if (out) *reinterpret_cast<float4*>(out + off + dt * 32 + 8 * g) = make_float4(v0, v1, v2, v3);
      }
  }
}

// -------------------------------------------------------------------------------------------------------------------
// The same attention with the split-f16 contraction of the GEMMs (x = hi + lo, three v_mfma_f32_32x32x16_f16 per 16-deep
// step, fp32 accumulation, lo.lo dropped: 2^-22 relative) instead of exact-fp32 MFMAs: 24 matrix instructions of 32 cycles
// per 32-key tile instead of 64 of 64 cycles.  The kernel time of a step is set by the LONGEST sequence (a 20-fragment
// puzzle walks 16 key tiles while a 4-fragment one walks 4), so the per-tile latency is what matters.
//   S^T = K.Q^T   : K tile in LDS as fp16 planes [key][dim], Q fragments (8 consecutive dims per lane) split once
//   O^T += V^T.P^T: V tile in LDS TRANSPOSED as planes [dim][key] (the A operand wants 8 consecutive keys per lane),
//                   P^T from the score accumulator through the lane ^ 32 exchange (score_to_fragments)
typedef _Float16 ad_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 ad_half4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void ad_split(float x, _Float16& hi, _Float16& lo) {
  const _Float16 h = (_Float16)x;
  hi = h;
  lo = (_Float16)(x - (float)h);
}

// accumulator tile (lane = query, register e = key (e&3) + 8*(e>>2) + 4*lhi) -> two 16-deep B fragments (lane = query,
// 8 consecutive keys at 16*g + 8*lhi), hi and lo planes
__device__ __forceinline__ void score_to_fragments(const f32x16 y, int lhi, ad_half8 (&fh)[2], ad_half8 (&fl)[2]) {
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    ad_half4 lo_h, lo_l, up_h, up_l;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      _Float16 a, b;
      ad_split(y[8 * g + q], a, b); lo_h[q] = a; lo_l[q] = b;
      ad_split(y[8 * g + 4 + q], a, b); up_h[q] = a; up_l[q] = b;
    }
    const ad_half4 send_h = lhi ? lo_h : up_h, send_l = lhi ? lo_l : up_l;
    union { ad_half4 h; int2 i; } sh, sl, rh, rl;
    sh.h = send_h; sl.h = send_l;
    rh.i.x = __shfl_xor(sh.i.x, 32); rh.i.y = __shfl_xor(sh.i.y, 32);
    rl.i.x = __shfl_xor(sl.i.x, 32); rl.i.y = __shfl_xor(sl.i.y, 32);
    const ad_half4 a_h = lhi ? rh.h : lo_h, b_h = lhi ? up_h : rh.h;
    const ad_half4 a_l = lhi ? rl.h : lo_l, b_l = lhi ? up_l : rl.h;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      fh[g][q] = a_h[q]; fh[g][4 + q] = b_h[q];
      fl[g][q] = a_l[q]; fl[g][4 + q] = b_l[q];
    }
  }
}

// offset (halfs) of (dim, key) in the transposed V planes: keys in 16-byte chunks of 8, the chunk index XOR-swizzled by bits
// 4-5 of the dim (the 16 lanes that write one key pair for dims 4*c4 + j then hit 16 different banks)
__device__ __forceinline__ int ad_toff(int dim, int key) { return dim * (32 + 8) + ((((key >> 3) ^ (dim >> 4)) & 3) << 3) + (key & 7); }

// X1: single-pass fp16 (hi planes only, one matrix instruction per product) — the perf mode of BASELINE configs[4]
// (pfpp_set_attention_mode(2)); ~1e-3 relative error, never a parity mode
template <int DH, bool X1 = false>
__global__ __launch_bounds__(256) void attn_dense_f16_kernel(
    const float* __restrict__ qkv, float* __restrict__ out, _Float16* __restrict__ out_hi,
    _Float16* __restrict__ out_lo, const int32_t* __restrict__ seq_off,
    const int32_t* __restrict__ seq_len, const uint8_t* __restrict__ key_valid, int64_t kv_stride,
    int H, float scale, float* __restrict__ lse) {
  pfpp_chain_prio();
  constexpr int KT = 32, NDT = DH / 32;      // NDT: 32-wide tiles of the head dimension
  constexpr int LDKH = DH + 8;          // halfs per K row  (16-byte fragment reads conflict-free)
  constexpr int LDVH = KT + 8;          // halfs per V^T row
  constexpr int F4 = KT * DH / 4 / 256;
  __shared__ __align__(16) _Float16 Kh[2][KT * LDKH], Kl[X1 ? 1 : 2][X1 ? 8 : KT * LDKH];
  __shared__ __align__(16) _Float16 Vh[2][DH * LDVH], Vl[X1 ? 1 : 2][X1 ? 8 : DH * LDVH];

  const int b = blockIdx.z, h = blockIdx.y;
  const int T = seq_len[b];
  const int q_base = blockIdx.x * 128;
  if (q_base >= T) return;
  const int64_t row0 = seq_off[b];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int C = H * DH;
  const int64_t ld = 3ll * C;
  const float* base = qkv + row0 * ld + h * DH;
  const uint8_t* kv = key_valid ? key_valid + (int64_t)b * kv_stride : nullptr;

  // Q fragments of this lane's query: dims 16c + 8*lhi .. +7
  const int q_row = q_base + wave * 32 + l31;
  const float* qp = base + (int64_t)min(q_row, T - 1) * ld + lhi * 8;
  // the softmax runs in the log2 domain: q is pre-multiplied by scale * log2(e), so a probability is ONE v_exp_f32 of (score - max)
  // — the per-score multiply by the scale and the one inside expf() are gone (the walk over the key tiles is VALU-bound for
  // dim_head 32: ~180 vector instructions against 4 matrix instructions per tile)
  const float qs = scale * 1.44269504088896340736f;
  ad_half8 qh[DH / 16], ql[DH / 16];
#pragma unroll
  for (int c = 0; c < DH / 16; ++c) {
    const float4 a = *reinterpret_cast<const float4*>(qp + c * 16);
    const float4 bq = *reinterpret_cast<const float4*>(qp + c * 16 + 4);
    const float x[8] = {a.x * qs, a.y * qs, a.z * qs, a.w * qs, bq.x * qs, bq.y * qs, bq.z * qs, bq.w * qs};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      _Float16 hh, ll;
      ad_split(x[e], hh, ll);
      qh[c][e] = hh; ql[c][e] = ll;
    }
  }

  f32x16 o_acc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) o_acc[dt][e] = 0.0f;
  float m_run = -1e30f, l_run = 0.0f;

  float4 rk[F4], rv[F4];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int it = 0; it < F4; ++it) {
      const int idx = tid + 256 * it;
      const int r = idx / (DH / 4), c4 = idx % (DH / 4);
      const float* src = base + (int64_t)min(k0 + r, T - 1) * ld + C + c4 * 4;
      rk[it] = *reinterpret_cast<const float4*>(src);
      rv[it] = *reinterpret_cast<const float4*>(src + C);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int it = 0; it < F4; ++it) {
      const int idx = tid + 256 * it;
      const int r = idx / (DH / 4), c4 = idx % (DH / 4);
      const float kx[4] = {rk[it].x, rk[it].y, rk[it].z, rk[it].w};
      const float vx[4] = {rv[it].x, rv[it].y, rv[it].z, rv[it].w};
      ad_half4 kh4, kl4, vh4, vl4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        _Float16 hh, ll;
        ad_split(kx[j], hh, ll);
        kh4[j] = hh; kl4[j] = ll;
        ad_split(vx[j], hh, ll);
        vh4[j] = hh; vl4[j] = ll;
      }
      *reinterpret_cast<ad_half4*>(&Kh[buf][r * LDKH + c4 * 4]) = kh4;
      if constexpr (!X1) *reinterpret_cast<ad_half4*>(&Kl[buf][r * LDKH + c4 * 4]) = kl4;
      // V transposed ([dim][key]): lanes r and r+1 (16 apart) swap halves, each then writes two keys of two dims as words
      union { ad_half4 h; int2 i; } uh, ul;
      uh.h = vh4; ul.h = vl4;
      const bool odd = r & 1;
      const int keep_h = odd ? uh.i.y : uh.i.x, send_h = odd ? uh.i.x : uh.i.y;
      const int keep_l = odd ? ul.i.y : ul.i.x, send_l = odd ? ul.i.x : ul.i.y;
      const int got_h = __shfl_xor(send_h, DH / 4), got_l = __shfl_xor(send_l, DH / 4);      // rows r and r + 1 are DH / 4 lanes apart
      const int a_h = odd ? got_h : keep_h, b_h = odd ? keep_h : got_h;
      const int a_l = odd ? got_l : keep_l, b_l = odd ? keep_l : got_l;
      const int d0 = c4 * 4 + (odd ? 2 : 0), r0 = r & ~1;
      *reinterpret_cast<int*>(&Vh[buf][ad_toff(d0, r0)]) = (a_h & 0xffff) | (b_h << 16);
      *reinterpret_cast<int*>(&Vh[buf][ad_toff(d0 + 1, r0)]) = ((unsigned)a_h >> 16) | (b_h & 0xffff0000);
      if constexpr (!X1) {
        *reinterpret_cast<int*>(&Vl[buf][ad_toff(d0, r0)]) = (a_l & 0xffff) | (b_l << 16);
        *reinterpret_cast<int*>(&Vl[buf][ad_toff(d0 + 1, r0)]) = ((unsigned)a_l >> 16) | (b_l & 0xffff0000);
      }
    }
  };

  const int nt = (T + KT - 1) / KT;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    const int k0 = t * KT;
    load_tile(min(k0 + KT, (nt - 1) * KT));   // unconditional (a conditional load makes the compiler wait for it right here)

    const int kidx = k0 + l31;
    const bool kval = kidx < T && (!kv || kv[kidx] != 0);
    const unsigned kmask = (unsigned)(__ballot(kval) & 0xffffffffull);

    // ---- S^T = K . Q^T ----
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.0f;
#pragma unroll
    for (int c = 0; c < DH / 16; ++c) {
      const ad_half8 kh = *reinterpret_cast<const ad_half8*>(&Kh[buf][l31 * LDKH + c * 16 + lhi * 8]);
      if constexpr (!X1) {
        const ad_half8 kl = *reinterpret_cast<const ad_half8*>(&Kl[buf][l31 * LDKH + c * 16 + lhi * 8]);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[c], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[c], s, 0, 0, 0);
      }
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[c], s, 0, 0, 0);
    }

    // ---- online softmax over this lane's query column (log2 domain) ----
    float mx = -__builtin_huge_valf();
    if (kmask == 0xffffffffu) {                       // every key of the tile counts (all but the last tile of an unmasked sequence)
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[e]);
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = (e & 3) + 8 * (e >> 2) + 4 * lhi;
        const float v = (kmask >> key) & 1u ? s[e] : -__builtin_huge_valf();
        s[e] = v;
        mx = fmaxf(mx, v);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    float psum = 0.0f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float pe = __builtin_amdgcn_exp2f(s[e] - m_new);
      s[e] = pe;
      psum += pe;
    }
    if (__ballot(m_new != m_run)) {                   // the running maximum moved for some query of the wave: rescale what was summed so far
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) o_acc[dt][e] *= alpha;
    }
    l_run += psum;
    m_run = m_new;

    // ---- O^T += V^T . P^T ----
    ad_half8 ph[2], pl[2];
    score_to_fragments(s, lhi, ph, pl);
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const ad_half8 vh = *reinterpret_cast<const ad_half8*>(&Vh[buf][ad_toff(dt * 32 + l31, g * 16 + lhi * 8)]);
        if constexpr (!X1) {
          const ad_half8 vl = *reinterpret_cast<const ad_half8*>(&Vl[buf][ad_toff(dt * 32 + l31, g * 16 + lhi * 8)]);
          o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[g], o_acc[dt], 0, 0, 0);
          o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[g], o_acc[dt], 0, 0, 0);
        }
        o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[g], o_acc[dt], 0, 0, 0);
      }
    if (t + 1 < nt) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- normalise and store: lane holds O[query = l31][d = dt*32 + 8g + 4*lhi + (0..3)] ----
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
  if (lse && q_row < T && lhi == 0) lse[(row0 + q_row) * H + h] = m_run * 0.69314718055994530942f + logf(l_tot);     // back to natural log
  if (q_row < T) {
    const int64_t off = (row0 + q_row) * (int64_t)C + h * DH + lhi * 4;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float v0 = o_acc[dt][4 * g + 0] * inv, v1 = o_acc[dt][4 * g + 1] * inv;
        const float v2 = o_acc[dt][4 * g + 2] * inv, v3 = o_acc[dt][4 * g + 3] * inv;
        if (out_hi) {
          ad_half4 hi, lo;
          PFPP_SPLIT_TO(v0, hi[0], lo[0]);
          PFPP_SPLIT_TO(v1, hi[1], lo[1]);
          PFPP_SPLIT_TO(v2, hi[2], lo[2]);
          PFPP_SPLIT_TO(v3, hi[3], lo[3]);
          *reinterpret_cast<ad_half4*>(out_hi + off + dt * 32 + 8 * g) = hi;
          *reinterpret_cast<ad_half4*>(out_lo + off + dt * 32 + 8 * g) = lo;
        }
        if (out) *reinterpret_cast<float4*>(out + off + dt * 32 + 8 * g) = make_float4(v0, v1, v2, v3);
      }
  }
}


// ------------------------------------------------------------------------------------------------------------------------
// SHORT sequences, few of them (one puzzle in flight: 100-500 tokens, one sequence): the kernel above gives a head ONE workgroup
// per 128 queries whose four waves walk the key tiles together, a barrier per tile — 8-16 workgroups on the chip and 4-16
// dependent tile steps (12-19 us per launch, six launches per DDPM step).  Here a workgroup owns 32 queries and its four waves
// split the KEYS: wave w takes key tiles w, w + 4, ... with a private LDS slot (it loads, converts and reads back its own tile:
// no workgroup barrier in the walk), then the four partial (max, sum, O) are merged in wave order through LDS — the
// online-softmax merge, deterministic.  4x the workgroups, a quarter of the dependent steps.  Same products as the kernel
// above, another association of the sums: equal to it to fp32 rounding.
template <int DH>
__global__ __launch_bounds__(256) void attn_dense_short_kernel(
    const float* __restrict__ qkv, float* __restrict__ out, _Float16* __restrict__ out_hi,
    _Float16* __restrict__ out_lo, const int32_t* __restrict__ seq_off,
    const int32_t* __restrict__ seq_len, const uint8_t* __restrict__ key_valid, int64_t kv_stride,
    int H, float scale) {
  constexpr int KT = 32, NDT = DH / 32;
  constexpr int LDKH = DH + 8, LDVH = KT + 8;
  constexpr int F4 = KT * DH / 4 / 64;               // float4 pieces of a K (or V) tile per lane
  constexpr int SLOT = 2 * KT * LDKH + 2 * DH * LDVH;        // halfs per wave slot: Kh | Kl | Vh | Vl
  constexpr int OP = DH + 4;                         // floats per query row of the merge buffer
  extern __shared__ __align__(16) char as_smem[];
  _Float16* slots = reinterpret_cast<_Float16*>(as_smem);
  float* ob = reinterpret_cast<float*>(as_smem + (size_t)4 * SLOT * sizeof(_Float16));     // [4 waves][32 queries][OP]
  float* ms = ob + 4 * 32 * OP;                      // [4][32] running maxima, [4][32] sums
  float* ls = ms + 4 * 32;

  const int b = blockIdx.z, h = blockIdx.y;
  const int T = seq_len[b];
  const int q_base = blockIdx.x * 32;
  if (q_base >= T) return;
  const int64_t row0 = seq_off[b];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int C = H * DH;
  const int64_t ld = 3ll * C;
  const float* base = qkv + row0 * ld + h * DH;
  const uint8_t* kv = key_valid ? key_valid + (int64_t)b * kv_stride : nullptr;
  _Float16* Kh = slots + wave * SLOT;
  _Float16* Kl = Kh + KT * LDKH;
  _Float16* Vh = Kl + KT * LDKH;
  _Float16* Vl = Vh + DH * LDVH;

  const int q_row = q_base + l31;                    // every wave holds the same 32 queries
  const float* qp = base + (int64_t)min(q_row, T - 1) * ld + lhi * 8;
  const float qs = scale * 1.44269504088896340736f;  // log2 domain, as above
  ad_half8 qh[DH / 16], ql[DH / 16];
#pragma unroll
  for (int c = 0; c < DH / 16; ++c) {
    const float4 a = *reinterpret_cast<const float4*>(qp + c * 16);
    const float4 bq = *reinterpret_cast<const float4*>(qp + c * 16 + 4);
    const float x[8] = {a.x * qs, a.y * qs, a.z * qs, a.w * qs, bq.x * qs, bq.y * qs, bq.z * qs, bq.w * qs};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      _Float16 hh, ll;
      ad_split(x[e], hh, ll);
      qh[c][e] = hh; ql[c][e] = ll;
    }
  }
  f32x16 o_acc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) o_acc[dt][e] = 0.0f;
  float m_run = -1e30f, l_run = 0.0f;

  float4 rk[F4], rv[F4];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int it = 0; it < F4; ++it) {
      const int idx = lane + 64 * it;
      const int r = idx / (DH / 4), c4 = idx % (DH / 4);
      const float* src = base + (int64_t)min(k0 + r, T - 1) * ld + C + c4 * 4;
      rk[it] = *reinterpret_cast<const float4*>(src);
      rv[it] = *reinterpret_cast<const float4*>(src + C);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int it = 0; it < F4; ++it) {
      const int idx = lane + 64 * it;
      const int r = idx / (DH / 4), c4 = idx % (DH / 4);
      const float kx[4] = {rk[it].x, rk[it].y, rk[it].z, rk[it].w};
      const float vx[4] = {rv[it].x, rv[it].y, rv[it].z, rv[it].w};
      ad_half4 kh4, kl4, vh4, vl4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        _Float16 hh, ll;
        ad_split(kx[j], hh, ll);
        kh4[j] = hh; kl4[j] = ll;
        ad_split(vx[j], hh, ll);
        vh4[j] = hh; vl4[j] = ll;
      }
      *reinterpret_cast<ad_half4*>(&Kh[r * LDKH + c4 * 4]) = kh4;
      *reinterpret_cast<ad_half4*>(&Kl[r * LDKH + c4 * 4]) = kl4;
      // V transposed ([dim][key]) exactly as in attn_dense_f16_kernel: rows r and r + 1 are DH / 4 lanes apart
      union { ad_half4 h; int2 i; } uh, ul;
      uh.h = vh4; ul.h = vl4;
      const bool odd = r & 1;
      const int keep_h = odd ? uh.i.y : uh.i.x, send_h = odd ? uh.i.x : uh.i.y;
      const int keep_l = odd ? ul.i.y : ul.i.x, send_l = odd ? ul.i.x : ul.i.y;
      const int got_h = __shfl_xor(send_h, DH / 4), got_l = __shfl_xor(send_l, DH / 4);
      const int a_h = odd ? got_h : keep_h, b_h = odd ? keep_h : got_h;
      const int a_l = odd ? got_l : keep_l, b_l = odd ? keep_l : got_l;
      const int d0 = c4 * 4 + (odd ? 2 : 0), r0 = r & ~1;
      *reinterpret_cast<int*>(&Vh[ad_toff(d0, r0)]) = (a_h & 0xffff) | (b_h << 16);
      *reinterpret_cast<int*>(&Vh[ad_toff(d0 + 1, r0)]) = ((unsigned)a_h >> 16) | (b_h & 0xffff0000);
      *reinterpret_cast<int*>(&Vl[ad_toff(d0, r0)]) = (a_l & 0xffff) | (b_l << 16);
      *reinterpret_cast<int*>(&Vl[ad_toff(d0 + 1, r0)]) = ((unsigned)a_l >> 16) | (b_l & 0xffff0000);
    }
  };

  const int nt = (T + KT - 1) / KT;
  if (wave < nt) load_tile(wave * KT);
  for (int t = wave; t < nt; t += 4) {
    const int k0 = t * KT;
    store_tile();
    __builtin_amdgcn_wave_barrier();
    load_tile(min(k0 + 4 * KT, (nt - 1) * KT));      // unconditional: the next tile of this wave (or a harmless re-read)

    const int kidx = k0 + l31;
    const bool kval = kidx < T && (!kv || kv[kidx] != 0);
    const unsigned kmask = (unsigned)(__ballot(kval) & 0xffffffffull);
    f32x16 sc;
#pragma unroll
    for (int e = 0; e < 16; ++e) sc[e] = 0.0f;
#pragma unroll
    for (int c = 0; c < DH / 16; ++c) {
      const ad_half8 kh = *reinterpret_cast<const ad_half8*>(&Kh[l31 * LDKH + c * 16 + lhi * 8]);
      const ad_half8 kl = *reinterpret_cast<const ad_half8*>(&Kl[l31 * LDKH + c * 16 + lhi * 8]);
      sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[c], sc, 0, 0, 0);
      sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[c], sc, 0, 0, 0);
      sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[c], sc, 0, 0, 0);
    }
    float mx = -__builtin_huge_valf();
    if (kmask == 0xffffffffu) {
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sc[e]);
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = (e & 3) + 8 * (e >> 2) + 4 * lhi;
        const float v = (kmask >> key) & 1u ? sc[e] : -__builtin_huge_valf();
        sc[e] = v;
        mx = fmaxf(mx, v);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    float psum = 0.0f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float pe = __builtin_amdgcn_exp2f(sc[e] - m_new);
      sc[e] = pe;
      psum += pe;
    }
    if (__ballot(m_new != m_run)) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) o_acc[dt][e] *= alpha;
    }
    l_run += psum;
    m_run = m_new;
    ad_half8 ph[2], pl[2];
    score_to_fragments(sc, lhi, ph, pl);
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const ad_half8 vh = *reinterpret_cast<const ad_half8*>(&Vh[ad_toff(dt * 32 + l31, g * 16 + lhi * 8)]);
        const ad_half8 vl = *reinterpret_cast<const ad_half8*>(&Vl[ad_toff(dt * 32 + l31, g * 16 + lhi * 8)]);
        o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[g], o_acc[dt], 0, 0, 0);
        o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[g], o_acc[dt], 0, 0, 0);
        o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[g], o_acc[dt], 0, 0, 0);
      }
    __builtin_amdgcn_wave_barrier();
  }

  // ---- merge the four waves' partial softmaxes (wave order): lane holds O[query = l31][d = dt*32 + 8g + 4*lhi + (0..3)]
  const float l_w = l_run + __shfl_xor(l_run, 32);
  if (lhi == 0) { ms[wave * 32 + l31] = m_run; ls[wave * 32 + l31] = l_w; }
  __syncthreads();
  {
    const float m0 = ms[l31], m1 = ms[32 + l31], m2 = ms[64 + l31], m3 = ms[96 + l31];
    const float m_all = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
    const float f = __builtin_amdgcn_exp2f(m_run - m_all);        // 0 for a wave that saw no key (m_run = -1e30)
    float* orow = ob + (wave * 32 + l31) * OP;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(orow + dt * 32 + 8 * g + 4 * lhi) =
            make_float4(o_acc[dt][4 * g + 0] * f, o_acc[dt][4 * g + 1] * f, o_acc[dt][4 * g + 2] * f, o_acc[dt][4 * g + 3] * f);
  }
  __syncthreads();
  // thread -> query tid / 8, dims (DH / 8) * (tid % 8) .. + DH / 8
  constexpr int DPT = DH / 8;
  const int q = tid >> 3, d0 = (tid & 7) * DPT;
  if (q_base + q < T) {
    float mw[4], lw[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) { mw[w] = ms[w * 32 + q]; lw[w] = ls[w * 32 + q]; }
    const float m_all = fmaxf(fmaxf(mw[0], mw[1]), fmaxf(mw[2], mw[3]));
    float l_tot = 0.0f;
#pragma unroll
    for (int w = 0; w < 4; ++w) l_tot += lw[w] * __builtin_amdgcn_exp2f(mw[w] - m_all);
    const float inv = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
    const int64_t off = (row0 + q_base + q) * (int64_t)C + h * DH + d0;
#pragma unroll
    for (int j = 0; j < DPT; j += 4) {
      float4 v = *reinterpret_cast<const float4*>(ob + q * OP + d0 + j);
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const float4 t4 = *reinterpret_cast<const float4*>(ob + (w * 32 + q) * OP + d0 + j);
        v.x += t4.x; v.y += t4.y; v.z += t4.z; v.w += t4.w;
      }
      v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
      if (out_hi) {
        ad_half4 hi, lo;
        PFPP_SPLIT_TO(v.x, hi[0], lo[0]); PFPP_SPLIT_TO(v.y, hi[1], lo[1]); PFPP_SPLIT_TO(v.z, hi[2], lo[2]); PFPP_SPLIT_TO(v.w, hi[3], lo[3]);
        *reinterpret_cast<ad_half4*>(out_hi + off + j) = hi;
        *reinterpret_cast<ad_half4*>(out_lo + off + j) = lo;
      }
      if (out) *reinterpret_cast<float4*>(out + off + j) = v;
    }
  }
}

}  // namespace

static int attn_dense_impl(const float* qkv, float* out, _Float16* out_hi, _Float16* out_lo, const int32_t* seq_off,
                           const int32_t* seq_len, const uint8_t* key_valid, int64_t kv_stride, int64_t n_seq,
                           int64_t max_len, int64_t H, int64_t dh, float scale, pfpp_stream_t stream,
                           float* lse = nullptr);

extern "C" int pfpp_attn_dense_train(const float* qkv, float* out, float* lse, const int32_t* seq_off,
                                     const int32_t* seq_len, const uint8_t* key_valid, int64_t kv_stride,
                                     int64_t n_seq, int64_t max_len, int64_t H, int64_t dh, float scale,
                                     pfpp_stream_t stream) {
  PFPP_REQUIRE(out && lse, "null pointer");
  return attn_dense_impl(qkv, out, nullptr, nullptr, seq_off, seq_len, key_valid, kv_stride, n_seq, max_len, H, dh,
                         scale, stream, lse);
}

// training forward with the output additionally (or only) as split-f16 planes for the out-projection GEMM
extern "C" int pfpp_attn_dense_train_p(const float* qkv, float* out, float* lse, const int32_t* seq_off,
                                       const int32_t* seq_len, const uint8_t* key_valid, int64_t kv_stride,
                                       int64_t n_seq, int64_t max_len, int64_t H, int64_t dh, float scale,
                                       const pfpp_planes* out_planes, pfpp_stream_t stream) {
  PFPP_REQUIRE(lse && (out || out_planes) && pfpp_planes_ok(out_planes), "null pointer");
  PFPP_REQUIRE(!out_planes || out_planes->scale == 1.0f, "forward planes are unscaled");
  return attn_dense_impl(qkv, out, out_planes ? (_Float16*)out_planes->hi : nullptr, out_planes ? (_Float16*)out_planes->lo : nullptr,
                         seq_off, seq_len, key_valid, kv_stride, n_seq, max_len, H, dh, scale, stream, lse);
}

extern "C" int pfpp_attn_dense(const float* qkv, float* out, const int32_t* seq_off, const int32_t* seq_len,
                               const uint8_t* key_valid, int64_t kv_stride, int64_t n_seq, int64_t max_len,
                               int64_t H, int64_t dh, float scale, pfpp_stream_t stream) {
  PFPP_REQUIRE(out, "null pointer");
  return attn_dense_impl(qkv, out, nullptr, nullptr, seq_off, seq_len, key_valid, kv_stride, n_seq, max_len, H, dh,
                         scale, stream);
}

extern "C" int pfpp_attn_dense_split(const float* qkv, void* out_hi, void* out_lo, const int32_t* seq_off,
                                     const int32_t* seq_len, const uint8_t* key_valid, int64_t kv_stride,
                                     int64_t n_seq, int64_t max_len, int64_t H, int64_t dh, float scale,
                                     pfpp_stream_t stream) {
  PFPP_REQUIRE(out_hi && out_lo, "null pointer");
  return attn_dense_impl(qkv, nullptr, (_Float16*)out_hi, (_Float16*)out_lo, seq_off, seq_len, key_valid, kv_stride,
                         n_seq, max_len, H, dh, scale, stream);
}

static int attn_dense_impl(const float* qkv, float* out, _Float16* out_hi, _Float16* out_lo, const int32_t* seq_off,
                           const int32_t* seq_len, const uint8_t* key_valid, int64_t kv_stride, int64_t n_seq,
                           int64_t max_len, int64_t H, int64_t dh, float scale, pfpp_stream_t stream, float* lse) {
  PFPP_REQUIRE(qkv && seq_off && seq_len, "null pointer");
  PFPP_REQUIRE(n_seq >= 0 && max_len >= 1 && H >= 1, "bad sizes");
  PFPP_SUPPORTED(dh == 64 || dh == 32, "dim_head must be 32 or 64");
  PFPP_SUPPORTED(n_seq <= 65535 && H <= 65535, "too many sequences / heads for one launch");
  PFPP_REQUIRE(pfpp::aligned16(qkv) && pfpp::aligned16(out) && pfpp::aligned16(out_hi) && pfpp::aligned16(out_lo),
               "16-byte alignment");
  if (n_seq == 0) return PFPP_OK;
  const dim3 grid((unsigned)((max_len + 127) / 128), (unsigned)H, (unsigned)n_seq);
  hipStream_t st = pfpp::as_stream(stream);
  // split-f16 kernel (default for dim_head 64): 44.6 -> 26.3 us on the compacted (ragged) token list, where the longest
  // sequence's walk over its key tiles is the critical path, and 226 -> 127 us on the all-slots form (32 x 500 keys, masked).
  // (Its first version stored V^T with 2-byte LDS writes: 661 us there; pairs of keys as swizzled 4-byte words fixed it.)
  // PFPP_ATTN_F16X3=0: the exact-fp32 MFMA kernel.
  static const int f16_mode = getenv("PFPP_ATTN_F16X3") ? atoi(getenv("PFPP_ATTN_F16X3")) : 1;
  const bool f16x3 = pfpp::attn_use_f16(f16_mode != 0);
  if (pfpp::attn_mode() == 2 && !lse) {      // single-pass fp16 (inference only: the training forward keeps its fp32-grade statistics)
    if (dh == 64)
      hipLaunchKernelGGL((attn_dense_f16_kernel<64, true>), grid, dim3(256), 0, st, qkv, out, out_hi, out_lo, seq_off, seq_len,
                         key_valid, kv_stride, (int)H, scale, lse);
    else
      hipLaunchKernelGGL((attn_dense_f16_kernel<32, true>), grid, dim3(256), 0, st, qkv, out, out_hi, out_lo, seq_off, seq_len,
                         key_valid, kv_stride, (int)H, scale, lse);
    return pfpp::check_launch("pfpp_attn_dense");
  }
  // few short sequences (one puzzle in flight): keys split over the waves of 32-query workgroups (attn_dense_short_kernel)
  const char* short_env = getenv("PFPP_ATTN_SHORT_MAX");          // (read per call: the tests compare both kernels in one process)
  const int short_max = short_env ? atoi(short_env) : 512;
  if (dh == 64 && f16x3 && !lse && max_len <= short_max && (int64_t)grid.x * H * n_seq <= 64) {
    constexpr size_t smem = (size_t)4 * (2 * 32 * (64 + 8) + 2 * 64 * (32 + 8)) * sizeof(_Float16) + (size_t)(4 * 32 * (64 + 4) + 8 * 32) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute((const void*)attn_dense_short_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return pfpp::check_launch("pfpp_attn_dense");
      attr_set = true;
    }
    const dim3 g32((unsigned)((max_len + 31) / 32), (unsigned)H, (unsigned)n_seq);
    hipLaunchKernelGGL(attn_dense_short_kernel<64>, g32, dim3(256), smem, st, qkv, out, out_hi, out_lo, seq_off, seq_len, key_valid, kv_stride,
                       (int)H, scale);
    return pfpp::check_launch("pfpp_attn_dense");
  }
  if (dh == 64 && f16x3)
    hipLaunchKernelGGL(attn_dense_f16_kernel<64>, grid, dim3(256), 0, st, qkv, out, out_hi, out_lo, seq_off, seq_len,
                       key_valid, kv_stride, (int)H, scale, lse);
  else if (dh == 32 && f16x3 && max_len >= 1024)
    // the verifier's attention over thousands of candidate edges (100-fragment puzzles: 4,950 keys per query): 3 f16 MFMAs per
    // 16-deep step instead of 8 fp32 ones; the 190 edges of the reference's 20-fragment puzzles keep the exact-fp32 kernel
    hipLaunchKernelGGL(attn_dense_f16_kernel<32>, grid, dim3(256), 0, st, qkv, out, out_hi, out_lo, seq_off, seq_len,
                       key_valid, kv_stride, (int)H, scale, lse);
  else if (dh == 64)
    hipLaunchKernelGGL(attn_dense_kernel<64>, grid, dim3(256), 0, st, qkv, out, out_hi, out_lo, seq_off, seq_len,
                       key_valid, kv_stride, (int)H, scale, lse);
  else
    hipLaunchKernelGGL(attn_dense_kernel<32>, grid, dim3(256), 0, st, qkv, out, out_hi, out_lo, seq_off, seq_len,
                       key_valid, kv_stride, (int)H, scale, lse);
  return pfpp::check_launch("pfpp_attn_dense");
}
