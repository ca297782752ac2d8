// a17: backward of the two attentions of an EncoderLayer (attention.py:77-85).
//
// Dense (global) attention — two passes, no atomics, probabilities recomputed from q, k and the
// log-sum-exp rows the training forward saved (pfpp_attn_dense_train):
//   pass 1 (per 32-query block per wave, keys streamed through LDS like the forward):
//       S^T = K.Q^T, P = exp(S*scale - lse), dP^T = V.dO^T, dS^T = P*(dP^T - D)*scale, dQ^T += K^T.dS^T
//       with D[q] = dO[q].O[q]  (also written out for pass 2)
//   pass 2 (per 32-key block per wave, queries streamed):
//       S = Q.K^T, P, dP = dO.V^T, dS as above,  dV^T += dO^T.P,  dK^T += Q^T.dS
// Both use the forward's register trick: with v_mfma_f32_32x32x2_f32 and the lane-half split of the
// contraction index, the 16 accumulator values a lane holds of the first product are exactly the
// B-operand values the second product needs, so P / dS never leave the registers.  Exact fp32
// products (gradients need no operand scaling here).
//
// Block-diagonal self-attention (L <= 32 tokens per fragment): one wave per (fragment, head), all
// five small products on the VALU out of LDS.
#include <stdlib.h>

#include "pfpp_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KT = 32;

// tools/lab/attn_bwd_probe.hip compiles this file with AB_PROBE: shader-clock stamps between the phases of the dk/dv tile loop,
// summed over the walk of workgroup (0, 0, 0) wave 0 and left in ab_probe_out
#ifdef AB_PROBE
__device__ long long ab_probe_out[16];
#define AB_STAMP_(i, W)                                                        \
  do {                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                         \
    long long now_;                                                            \
    asm volatile(W "s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory"); \
    ab_t[i] += now_ - ab_last;                                                 \
    ab_last = now_;                                                            \
    __builtin_amdgcn_sched_barrier(0);                                         \
  } while (0)
#define AB_STAMP(i) AB_STAMP_(i, "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t")
#define AB_STAMP_NOWAIT(i) AB_STAMP_(i, "")
#else
#define AB_STAMP(i)
#define AB_STAMP_NOWAIT(i)
#endif

// ---------------------------------------------------------------------------------------------------
// pass 1: dQ (and D)
// ---------------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void attn_dense_bwd_dq_kernel(
    const float* __restrict__ qkv, const float* __restrict__ out, const float* __restrict__ dout,
    const float* __restrict__ lse, float* __restrict__ dvec, float* __restrict__ dqkv,
    const int32_t* __restrict__ seq_off, const int32_t* __restrict__ seq_len,
    const uint8_t* __restrict__ key_valid, int64_t kv_stride, int H, float scale) {
  constexpr int LDK = DH + 4;
  constexpr int NCH = DH / 8;
  constexpr int NDT = DH / 32;
  constexpr int F4 = KT * DH / 4 / 256;
  __shared__ __align__(16) float Ks[2][KT * LDK];
  __shared__ __align__(16) float Vs[2][KT * LDK];

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

  const int q_row = q_base + wave * 32 + l31;
  const int q_cl = min(q_row, T - 1);
  const float* qp = base + (int64_t)q_cl * ld + lhi * 4;
  const float* op = out + (row0 + q_cl) * (int64_t)C + h * DH + lhi * 4;
  const float* dop = dout + (row0 + q_cl) * (int64_t)C + h * DH + lhi * 4;
  float4 qf[NCH], dof[NCH];
  float dpart = 0.0f;
#pragma unroll
  for (int kc = 0; kc < NCH; ++kc) {
    qf[kc] = *reinterpret_cast<const float4*>(qp + kc * 8);
    dof[kc] = *reinterpret_cast<const float4*>(dop + kc * 8);
    const float4 o = *reinterpret_cast<const float4*>(op + kc * 8);
    dpart += (dof[kc].x * o.x + dof[kc].y * o.y) + (dof[kc].z * o.z + dof[kc].w * o.w);
  }
  const float Dq = dpart + __shfl_xor(dpart, 32);
  const float lse_q = lse[(row0 + q_cl) * H + h];
  if (dvec && q_row < T && lhi == 0) dvec[(row0 + q_row) * H + h] = Dq;

  f32x16 dq_acc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) dq_acc[dt][e] = 0.0f;

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
      *reinterpret_cast<float4*>(&Vs[buf][r * LDK + c4 * 4]) = rv[it];
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

    f32x16 s, dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) { s[e] = 0.0f; dp[e] = 0.0f; }
    const float* kp = &Ks[buf][l31 * LDK + lhi * 4];
    const float* vp = &Vs[buf][l31 * LDK + lhi * 4];
#pragma unroll
    for (int kc = 0; kc < NCH; ++kc) {
      const float4 a = *reinterpret_cast<const float4*>(kp + kc * 8);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qf[kc].x, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qf[kc].y, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qf[kc].z, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qf[kc].w, s, 0, 0, 0);
      const float4 c = *reinterpret_cast<const float4*>(vp + kc * 8);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c.x, dof[kc].x, dp, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c.y, dof[kc].y, dp, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c.z, dof[kc].z, dp, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c.w, dof[kc].w, dp, 0, 0, 0);
    }
    // dS^T[key][query] in place of s
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = (e & 3) + 8 * (e >> 2) + 4 * lhi;
      const float p = (kmask >> key) & 1u ? expf(s[e] * scale - lse_q) : 0.0f;
      s[e] = p * (dp[e] - Dq) * scale;
    }
    // dQ^T[d][query] += K^T[d][key] . dS^T[key][query]
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const float* kc_p = &Ks[buf][(lhi * 4) * LDK + dt * 32 + l31];
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        dq_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kc_p[(kc * 8 + 0) * LDK], s[4 * kc + 0], dq_acc[dt], 0, 0, 0);
        dq_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kc_p[(kc * 8 + 1) * LDK], s[4 * kc + 1], dq_acc[dt], 0, 0, 0);
        dq_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kc_p[(kc * 8 + 2) * LDK], s[4 * kc + 2], dq_acc[dt], 0, 0, 0);
        dq_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kc_p[(kc * 8 + 3) * LDK], s[4 * kc + 3], dq_acc[dt], 0, 0, 0);
      }
    }
    if (t + 1 < nt) store_tile(buf ^ 1);
    __syncthreads();
  }

  if (q_row < T) {
    float* dst = dqkv + (row0 + q_row) * ld + h * DH + lhi * 4;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(dst + dt * 32 + 8 * g) =
            make_float4(dq_acc[dt][4 * g + 0], dq_acc[dt][4 * g + 1], dq_acc[dt][4 * g + 2], dq_acc[dt][4 * g + 3]);
  }
}

// D[row, h] = sum_d dO . O alone (same lane mapping and summation order as in pass 1, hence the same bits): with D
// precomputed, pass 1 and pass 2 no longer depend on each other and can run on two streams
template <int DH>
__global__ __launch_bounds__(256) void attn_dense_bwd_d_kernel(
    const float* __restrict__ out, const float* __restrict__ dout, float* __restrict__ dvec,
    const int32_t* __restrict__ seq_off, const int32_t* __restrict__ seq_len, int H) {
  constexpr int NCH = DH / 8;
  const int b = blockIdx.z, h = blockIdx.y;
  const int T = seq_len[b];
  const int q_base = blockIdx.x * 128;
  if (q_base >= T) return;
  const int64_t row0 = seq_off[b];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int C = H * DH;
  const int q_row = q_base + wave * 32 + l31;
  const int q_cl = min(q_row, T - 1);
  const float* op = out + (row0 + q_cl) * (int64_t)C + h * DH + lhi * 4;
  const float* dop = dout + (row0 + q_cl) * (int64_t)C + h * DH + lhi * 4;
  float dpart = 0.0f;
#pragma unroll
  for (int kc = 0; kc < NCH; ++kc) {
    const float4 d = *reinterpret_cast<const float4*>(dop + kc * 8);
    const float4 o = *reinterpret_cast<const float4*>(op + kc * 8);
    dpart += (d.x * o.x + d.y * o.y) + (d.z * o.z + d.w * o.w);
  }
  const float Dq = dpart + __shfl_xor(dpart, 32);
  if (q_row < T && lhi == 0) dvec[(row0 + q_row) * H + h] = Dq;
}

// ---------------------------------------------------------------------------------------------------
// pass 2: dK, dV
// ---------------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void attn_dense_bwd_dkv_kernel(
    const float* __restrict__ qkv, const float* __restrict__ dout, const float* __restrict__ lse,
    const float* __restrict__ dvec, float* __restrict__ dqkv, const int32_t* __restrict__ seq_off,
    const int32_t* __restrict__ seq_len, const uint8_t* __restrict__ key_valid, int64_t kv_stride, int H,
    float scale) {
  constexpr int LDK = DH + 4;
  constexpr int NCH = DH / 8;
  constexpr int NDT = DH / 32;
  constexpr int F4 = KT * DH / 4 / 256;
  __shared__ __align__(16) float Qs[2][KT * LDK];
  __shared__ __align__(16) float Gs[2][KT * LDK];     // dO tile
  __shared__ float Ls[2][KT];                         // lse of the query tile
  __shared__ float Ds[2][KT];                         // D of the query tile

  const int b = blockIdx.z, h = blockIdx.y;
  const int T = seq_len[b];
  const int k_base = blockIdx.x * 128;
  if (k_base >= T) return;
  const int64_t row0 = seq_off[b];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int C = H * DH;
  const int64_t ld = 3ll * C;
  const float* base = qkv + row0 * ld + h * DH;
  const uint8_t* kv = key_valid ? key_valid + (int64_t)b * kv_stride : nullptr;

  const int k_row = k_base + wave * 32 + l31;
  const bool key_ok = k_row < T && (!kv || kv[k_row] != 0);
  const float* kp = base + (int64_t)min(k_row, T - 1) * ld + C + lhi * 4;
  float4 kf[NCH], vf[NCH];
#pragma unroll
  for (int kc = 0; kc < NCH; ++kc) {
    kf[kc] = *reinterpret_cast<const float4*>(kp + kc * 8);
    vf[kc] = *reinterpret_cast<const float4*>(kp + C + kc * 8);
  }
  f32x16 dk_acc[NDT], dv_acc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) { dk_acc[dt][e] = 0.0f; dv_acc[dt][e] = 0.0f; }

  float4 rq[F4], rg[F4];
  float rl = 0.0f, rd = 0.0f;
  auto load_tile = [&](int q0) {
#pragma unroll
    for (int it = 0; it < F4; ++it) {
      const int idx = tid + 256 * it;
      const int r = idx / (DH / 4), c4 = idx % (DH / 4);
      const int64_t qr = min(q0 + r, T - 1);
      rq[it] = *reinterpret_cast<const float4*>(base + qr * ld + c4 * 4);
      rg[it] = *reinterpret_cast<const float4*>(dout + (row0 + qr) * (int64_t)C + h * DH + c4 * 4);
    }
    if (tid < KT) {
      const int64_t qr = min(q0 + tid, T - 1);
      rl = lse[(row0 + qr) * H + h];
      rd = dvec[(row0 + qr) * H + h];
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int it = 0; it < F4; ++it) {
      const int idx = tid + 256 * it;
      const int r = idx / (DH / 4), c4 = idx % (DH / 4);
      *reinterpret_cast<float4*>(&Qs[buf][r * LDK + c4 * 4]) = rq[it];
      *reinterpret_cast<float4*>(&Gs[buf][r * LDK + c4 * 4]) = rg[it];
    }
    if (tid < KT) { Ls[buf][tid] = rl; Ds[buf][tid] = rd; }
  };

  const int nt = (T + KT - 1) / KT;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    const int q0 = t * KT;
    load_tile(min(q0 + KT, (nt - 1) * KT));   // unconditional (a conditional load makes the compiler wait for it right here)

    // S[q][key] = Q.K^T,  dP[q][key] = dO.V^T
    f32x16 s, dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) { s[e] = 0.0f; dp[e] = 0.0f; }
    const float* qp = &Qs[buf][l31 * LDK + lhi * 4];
    const float* gp = &Gs[buf][l31 * LDK + lhi * 4];
#pragma unroll
    for (int kc = 0; kc < NCH; ++kc) {
      const float4 a = *reinterpret_cast<const float4*>(qp + kc * 8);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, kf[kc].x, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, kf[kc].y, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, kf[kc].z, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, kf[kc].w, s, 0, 0, 0);
      const float4 c = *reinterpret_cast<const float4*>(gp + kc * 8);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c.x, vf[kc].x, dp, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c.y, vf[kc].y, dp, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c.z, vf[kc].z, dp, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c.w, vf[kc].w, dp, 0, 0, 0);
    }
    // p in s, dS in dp
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int qi = (e & 3) + 8 * (e >> 2) + 4 * lhi;
      const bool ok = key_ok && (q0 + qi) < T;
      const float p = ok ? expf(s[e] * scale - Ls[buf][qi]) : 0.0f;
      s[e] = p;
      dp[e] = p * (dp[e] - Ds[buf][qi]) * scale;
    }
    // dV^T[d][key] += dO^T[d][q] . P[q][key];  dK^T[d][key] += Q^T[d][q] . dS[q][key]
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const float* gc = &Gs[buf][(lhi * 4) * LDK + dt * 32 + l31];
      const float* qc = &Qs[buf][(lhi * 4) * LDK + dt * 32 + l31];
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dv_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(gc[(kc * 8 + j) * LDK], s[4 * kc + j], dv_acc[dt], 0, 0, 0);
          dk_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(qc[(kc * 8 + j) * LDK], dp[4 * kc + j], dk_acc[dt], 0, 0, 0);
        }
      }
    }
    if (t + 1 < nt) store_tile(buf ^ 1);
    __syncthreads();
  }

  if (k_row < T) {
    float* dst = dqkv + (row0 + k_row) * ld + C + h * DH + lhi * 4;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<float4*>(dst + dt * 32 + 8 * g) =
            make_float4(dk_acc[dt][4 * g + 0], dk_acc[dt][4 * g + 1], dk_acc[dt][4 * g + 2], dk_acc[dt][4 * g + 3]);
        *reinterpret_cast<float4*>(dst + C + dt * 32 + 8 * g) =
            make_float4(dv_acc[dt][4 * g + 0], dv_acc[dt][4 * g + 1], dv_acc[dt][4 * g + 2], dv_acc[dt][4 * g + 3]);
      }
  }
}

// ---------------------------------------------------------------------------------------------------
// The two passes with the split-f16 contraction (three v_mfma_f32_32x32x16_f16 per 16-deep step, see attention.hip's
// attn_dense_f16_kernel): 36 / 48 matrix instructions of 32 cycles per 32 x 32 tile pair instead of 96 / 128 of 64 cycles.
// The step time of the backward is the walk of the LONGEST sequence over its tiles, so this is its critical path.
// Gradient operands are lifted into the normal fp16 range by powers of two before the split (dO by 2^12, dS by 2^14)
// and the accumulators are scaled back at the end.  Second-stage products read their A operand from TRANSPOSED tiles.
// ---------------------------------------------------------------------------------------------------
typedef _Float16 ab_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 ab_half4 __attribute__((ext_vector_type(4)));
constexpr float AB_GS = 4096.0f;        // dO
constexpr float AB_DS = 16384.0f;       // dS
constexpr int AB_LDR = 64 + 8;          // halfs per row of a [row][dim] tile
constexpr int AB_LDT = KT + 8;          // halfs per row of a transposed [dim][row] tile

__device__ __forceinline__ void ab_split(float x, _Float16& hi, _Float16& lo) {
  // x is pinned as a ROUNDED fp32 value first: when x is a product, the compiler may otherwise take hi from a fused multiply-convert
  // (one rounding of the exact product) and lo from x - fp16(RN32(product)) (two roundings) — two different hi's, and in the rare
  // double-rounding case (about 1e-4 of the values) hi + lo is off by a whole fp16 ulp of hi
  asm("" : "+v"(x));
  const _Float16 h = (_Float16)x;
  hi = h;
  lo = (_Float16)(x - (float)h);
}

// 8 consecutive floats (x scale) -> hi / lo operand fragment
__device__ __forceinline__ void ab_frag8(const float* p, float sc, ab_half8& hi, ab_half8& lo) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    _Float16 h, l;
    ab_split(x[e] * sc, h, l);
    hi[e] = h; lo[e] = l;
  }
}

// accumulator tile (lane = column n, register e = row (e&3) + 8*(e>>2) + 4*lhi) -> two 16-deep B fragments
// (lane = n, 8 consecutive rows at 16*g + 8*lhi); lanes l and l+32 exchange the quads the other one needs
__device__ __forceinline__ void ab_acc_to_fragments(const f32x16 y, int lhi, ab_half8 (&fh)[2], ab_half8 (&fl)[2]) {
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    ab_half4 lo_h, lo_l, up_h, up_l;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      _Float16 a, b;
      ab_split(y[8 * g + q], a, b); lo_h[q] = a; lo_l[q] = b;
      ab_split(y[8 * g + 4 + q], a, b); up_h[q] = a; up_l[q] = b;
    }
    const ab_half4 send_h = lhi ? lo_h : up_h, send_l = lhi ? lo_l : up_l;
    union { ab_half4 h; int2 i; } sh, sl, rh, rl;
    sh.h = send_h; sl.h = send_l;
    rh.i.x = __shfl_xor(sh.i.x, 32); rh.i.y = __shfl_xor(sh.i.y, 32);
    rl.i.x = __shfl_xor(sl.i.x, 32); rl.i.y = __shfl_xor(sl.i.y, 32);
    const ab_half4 a_h = lhi ? rh.h : lo_h, b_h = lhi ? up_h : rh.h;
    const ab_half4 a_l = lhi ? rl.h : lo_l, b_l = lhi ? up_l : rl.h;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      fh[g][q] = a_h[q]; fh[g][4 + q] = b_h[q];
      fl[g][q] = a_l[q]; fl[g][4 + q] = b_l[q];
    }
  }
}

// acc += A(lo).B(hi) + A(hi).B(lo) + A(hi).B(hi) for the 16-deep fragments read at (row l31, offset) of the planes ah / al
__device__ __forceinline__ f32x16 ab_mma3(const _Float16* ah, const _Float16* al, const ab_half8 bh, const ab_half8 bl, f32x16 acc) {
  const ab_half8 xh = *reinterpret_cast<const ab_half8*>(ah);
  const ab_half8 xl = *reinterpret_cast<const ab_half8*>(al);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, bh, acc, 0, 0, 0);
  return acc;
}

// offset (halfs) of (dim, row) in a transposed plane: rows in 16-byte chunks of 8, the chunk index XOR-swizzled by bits 4-5
// of the dim so that the 16 lanes that write the same row pair for dims 4*c4 + j land in 16 different banks
__device__ __forceinline__ int ab_toff(int dim, int row) { return dim * AB_LDT + ((((row >> 3) ^ (dim >> 4)) & 3) << 3) + (row & 7); }

// one float4 of a [row r][dims c4*4..] tile (thread idx = r*16 + c4) -> row planes (8-byte store) and, if th, transposed
// planes: lanes r and r+1 (16 apart) swap halves first, so each writes TWO rows of two dims as 4-byte words
__device__ __forceinline__ void ab_store4(const float4 v, float sc, int r, int c4, _Float16* rh, _Float16* rl, _Float16* th, _Float16* tl) {
  const float x[4] = {v.x * sc, v.y * sc, v.z * sc, v.w * sc};
  ab_half4 h4, l4;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    _Float16 h, l;
    ab_split(x[j], h, l);
    h4[j] = h; l4[j] = l;
  }
  *reinterpret_cast<ab_half4*>(rh + r * AB_LDR + c4 * 4) = h4;
  *reinterpret_cast<ab_half4*>(rl + r * AB_LDR + c4 * 4) = l4;
  if (th) {
    union { ab_half4 h; int2 i; } uh, ul;
    uh.h = h4; ul.h = l4;
    const bool odd = r & 1;
    // even rows keep dims 0,1 (word .x) and send dims 2,3 (.y); odd rows keep dims 2,3 and send dims 0,1
    const int keep_h = odd ? uh.i.y : uh.i.x, send_h = odd ? uh.i.x : uh.i.y;
    const int keep_l = odd ? ul.i.y : ul.i.x, send_l = odd ? ul.i.x : ul.i.y;
    const int got_h = __shfl_xor(send_h, 16), got_l = __shfl_xor(send_l, 16);
    // (own row, partner row) in row order: even lane = (r, r+1), odd lane = (r-1, r)
    const int a_h = odd ? got_h : keep_h, b_h = odd ? keep_h : got_h;
    const int a_l = odd ? got_l : keep_l, b_l = odd ? keep_l : got_l;
    const int d0 = c4 * 4 + (odd ? 2 : 0), r0 = r & ~1;
    // word for dim d0: low half = row r0, high half = row r0+1 (low halves of a, b); dim d0+1: the high halves
    const int w0_h = (a_h & 0xffff) | (b_h << 16), w1_h = ((unsigned)a_h >> 16) | (b_h & 0xffff0000);
    const int w0_l = (a_l & 0xffff) | (b_l << 16), w1_l = ((unsigned)a_l >> 16) | (b_l & 0xffff0000);
    *reinterpret_cast<int*>(th + ab_toff(d0, r0)) = w0_h;
    *reinterpret_cast<int*>(th + ab_toff(d0 + 1, r0)) = w1_h;
    *reinterpret_cast<int*>(tl + ab_toff(d0, r0)) = w0_l;
    *reinterpret_cast<int*>(tl + ab_toff(d0 + 1, r0)) = w1_l;
  }
}

// LDS of the two passes, carved from one buffer each
typedef _Float16 AbRowTile[KT * AB_LDR];       // [row][dim] planes of a 32-row tile
typedef _Float16 AbTrTile[64 * AB_LDT];        // [dim][row] planes
constexpr int AB_DQ_SMEM = 8 * (int)sizeof(AbRowTile) + 4 * (int)sizeof(AbTrTile);
constexpr int AB_DKV_SMEM = 8 * (int)sizeof(AbRowTile) + 8 * (int)sizeof(AbTrTile) + 4 * KT * (int)sizeof(float);

__device__ __forceinline__ void ab_dq_f16_body(
    char* smem, const int blk, const float* __restrict__ qkv, const float* __restrict__ out, const float* __restrict__ dout,
    const float* __restrict__ lse, float* __restrict__ dvec, float* __restrict__ dqkv,
    const int32_t* __restrict__ seq_off, const int32_t* __restrict__ seq_len, int H, float scale,
    const pfpp_planes_out po = pfpp_planes_out{nullptr, nullptr, 1.0f}) {
  constexpr int DH = 64;
  constexpr int F4 = KT * DH / 4 / 256;
  AbRowTile* Kh = reinterpret_cast<AbRowTile*>(smem);
  AbRowTile* Kl = Kh + 2;
  AbRowTile* Vh = Kl + 2;
  AbRowTile* Vl = Vh + 2;
  AbTrTile* Kth = reinterpret_cast<AbTrTile*>(Vl + 2);
  AbTrTile* Ktl = Kth + 2;

  const int b = blockIdx.z, h = blockIdx.y;
  const int T = seq_len[b];
  const int q_base = blk * 128;
  if (q_base >= T) return;
  const int64_t row0 = seq_off[b];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int C = H * DH;
  const int64_t ld = 3ll * C;
  const float* base = qkv + row0 * ld + h * DH;

  const int q_row = q_base + wave * 32 + l31;
  const int q_cl = min(q_row, T - 1);
  // D = rowsum(dO . O) in fp32, with the lane mapping of the exact kernels (same bits)
  float Dq;
  {
    const float* op = out + (row0 + q_cl) * (int64_t)C + h * DH + lhi * 4;
    const float* dop = dout + (row0 + q_cl) * (int64_t)C + h * DH + lhi * 4;
    float dpart = 0.0f;
#pragma unroll
    for (int kc = 0; kc < DH / 8; ++kc) {
      const float4 d = *reinterpret_cast<const float4*>(dop + kc * 8);
      const float4 o = *reinterpret_cast<const float4*>(op + kc * 8);
      dpart += (d.x * o.x + d.y * o.y) + (d.z * o.z + d.w * o.w);
    }
    Dq = dpart + __shfl_xor(dpart, 32);
    if (dvec && q_row < T && lhi == 0) dvec[(row0 + q_row) * H + h] = Dq;
  }
  // P = exp2(S * (scale log2 e) - lse log2 e): one fma + v_exp_f32 per entry
  constexpr float LOG2E = 1.4426950408889634f;
  const float lse_q = lse[(row0 + q_cl) * H + h] * LOG2E;
  const float sc2 = scale * LOG2E;
  ab_half8 qh[4], ql[4], gh[4], gl[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    ab_frag8(base + (int64_t)q_cl * ld + c * 16 + lhi * 8, 1.0f, qh[c], ql[c]);
    ab_frag8(dout + (row0 + q_cl) * (int64_t)C + h * DH + c * 16 + lhi * 8, AB_GS, gh[c], gl[c]);
  }
  f32x16 dq_acc[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) dq_acc[dt][e] = 0.0f;

  float4 rk[F4], rv[F4];
  int tile_k0 = 0;
  const float* kbase = base + C;
  const uint32_t ld32 = (uint32_t)ld;
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int it = 0; it < F4; ++it) {
      const int idx = tid + 256 * it;
      const int r = idx / (DH / 4), c4 = idx % (DH / 4);
      const float* src = kbase + (uint32_t)min(k0 + r, T - 1) * ld32 + c4 * 4;      // 32-bit offsets inside the sequence
      rk[it] = *reinterpret_cast<const float4*>(src);
      rv[it] = *reinterpret_cast<const float4*>(src + C);
    }
    tile_k0 = k0;
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int it = 0; it < F4; ++it) {
      const int idx = tid + 256 * it;
      const int r = idx / (DH / 4), c4 = idx % (DH / 4);
      // rows past the end of the sequence are staged as zero keys: S = 0 -> a finite P, and K^T . dS^T gets nothing from them,
      // so the tile loop needs no per-entry key mask
      ab_store4(rk[it], tile_k0 + r < T ? 1.0f : 0.0f, r, c4, Kh[buf], Kl[buf], Kth[buf], Ktl[buf]);
      ab_store4(rv[it], 1.0f, r, c4, Vh[buf], Vl[buf], nullptr, nullptr);
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

    f32x16 s, dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) { s[e] = 0.0f; dp[e] = 0.0f; }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int off = l31 * AB_LDR + c * 16 + lhi * 8;
      s = ab_mma3(&Kh[buf][off], &Kl[buf][off], qh[c], ql[c], s);        // S^T[key][query]
      dp = ab_mma3(&Vh[buf][off], &Vl[buf][off], gh[c], gl[c], dp);      // dP^T * 2^12
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float pv = __builtin_amdgcn_exp2f(fmaf(s[e], sc2, -lse_q));
      s[e] = pv * (dp[e] * (1.0f / AB_GS) - Dq) * (scale * AB_DS);      // dS^T * 2^14
    }
    ab_half8 sh[2], sl[2];
    ab_acc_to_fragments(s, lhi, sh, sl);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int off = ab_toff(dt * 32 + l31, g * 16 + lhi * 8);
        dq_acc[dt] = ab_mma3(&Kth[buf][off], &Ktl[buf][off], sh[g], sl[g], dq_acc[dt]);   // dQ^T[d][query] * 2^14
      }
    if (t + 1 < nt) store_tile(buf ^ 1);
    __syncthreads();
  }

  if (q_row < T) {
    const int64_t d0 = (row0 + q_row) * ld + h * DH + lhi * 4;
    constexpr float inv = 1.0f / AB_DS;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 v = make_float4(dq_acc[dt][4 * g + 0] * inv, dq_acc[dt][4 * g + 1] * inv, dq_acc[dt][4 * g + 2] * inv, dq_acc[dt][4 * g + 3] * inv);
        if (dqkv) *reinterpret_cast<float4*>(dqkv + d0 + dt * 32 + 8 * g) = v;
        if (po.hi) pfpp_store4_planes(po, d0 + dt * 32 + 8 * g, v);
      }
  }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_dense_bwd_dq_f16_kernel(
    const float* __restrict__ qkv, const float* __restrict__ out, const float* __restrict__ dout,
    const float* __restrict__ lse, float* __restrict__ dvec, float* __restrict__ dqkv,
    const int32_t* __restrict__ seq_off, const int32_t* __restrict__ seq_len, int H, float scale, pfpp_planes_out po) {
  pfpp_chain_prio();
  __shared__ __align__(16) char smem[AB_DQ_SMEM];
  ab_dq_f16_body(smem, blockIdx.x, qkv, out, dout, lse, dvec, dqkv, seq_off, seq_len, H, scale, po);
}

__device__ __forceinline__ void ab_dkv_f16_body(
    char* smem, const int blk, const float* __restrict__ qkv, const float* __restrict__ dout, const float* __restrict__ lse,
    const float* __restrict__ dvec, float* __restrict__ dqkv, const int32_t* __restrict__ seq_off,
    const int32_t* __restrict__ seq_len, int H, float scale,
    const pfpp_planes_out po = pfpp_planes_out{nullptr, nullptr, 1.0f}) {
  constexpr int DH = 64;
  constexpr int F4 = KT * DH / 4 / 256;
  AbRowTile* Qh = reinterpret_cast<AbRowTile*>(smem);
  AbRowTile* Ql = Qh + 2;
  AbRowTile* Gh = Ql + 2;
  AbRowTile* Gl = Gh + 2;
  AbTrTile* Qth = reinterpret_cast<AbTrTile*>(Gl + 2);
  AbTrTile* Qtl = Qth + 2;
  AbTrTile* Gth = Qtl + 2;
  AbTrTile* Gtl = Gth + 2;
  typedef float AbVec[KT];
  AbVec* Ls = reinterpret_cast<AbVec*>(Gtl + 2);
  AbVec* Ds = Ls + 2;

  const int b = blockIdx.z, h = blockIdx.y;
  const int T = seq_len[b];
  const int k_base = blk * 128;
  if (k_base >= T) return;
  const int64_t row0 = seq_off[b];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int C = H * DH;
  const int64_t ld = 3ll * C;
  const float* base = qkv + row0 * ld + h * DH;

  const int k_row = k_base + wave * 32 + l31;
  const float* kp = base + (int64_t)min(k_row, T - 1) * ld + C + lhi * 8;
  ab_half8 kh[4], kl[4], vh[4], vl[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    ab_frag8(kp + c * 16, 1.0f, kh[c], kl[c]);
    ab_frag8(kp + C + c * 16, 1.0f, vh[c], vl[c]);
  }
  f32x16 dk_acc[2], dv_acc[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) { dk_acc[dt][e] = 0.0f; dv_acc[dt][e] = 0.0f; }

  float4 rq[F4], rg[F4];
  float rl = 0.0f, rd = 0.0f;
  bool rl_ok = false;
  constexpr float LOG2E = 1.4426950408889634f;
  const float sc2 = scale * LOG2E;
  const uint32_t ld32 = (uint32_t)ld;
  const float* gbase = dout + row0 * (int64_t)C + h * DH;
  const float* lbase = lse + row0 * H + h;
  const float* dbase = dvec + row0 * H + h;
  auto load_tile = [&](int q0) {
#pragma unroll
    for (int it = 0; it < F4; ++it) {
      const int idx = tid + 256 * it;
      const int r = idx / (DH / 4), c4 = idx % (DH / 4);
      const uint32_t qr = (uint32_t)min(q0 + r, T - 1);                             // 32-bit offsets inside the sequence
      rq[it] = *reinterpret_cast<const float4*>(base + qr * ld32 + c4 * 4);
      rg[it] = *reinterpret_cast<const float4*>(gbase + qr * (uint32_t)C + c4 * 4);
    }
    if (tid < KT) {
      const uint32_t qr = (uint32_t)min(q0 + tid, T - 1);
      // query rows past the end are staged with lse = +inf: P = exp2(-inf) = 0 without a per-entry mask
      rl = lbase[qr * (uint32_t)H];
      rd = dbase[qr * (uint32_t)H];
      rl_ok = q0 + tid < T;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int it = 0; it < F4; ++it) {
      const int idx = tid + 256 * it;
      const int r = idx / (DH / 4), c4 = idx % (DH / 4);
      ab_store4(rq[it], 1.0f, r, c4, Qh[buf], Ql[buf], Qth[buf], Qtl[buf]);
      ab_store4(rg[it], AB_GS, r, c4, Gh[buf], Gl[buf], Gth[buf], Gtl[buf]);
    }
    if (tid < KT) { Ls[buf][tid] = rl_ok ? rl * LOG2E : __builtin_huge_valf(); Ds[buf][tid] = rd; }
  };

  const int nt = (T + KT - 1) / KT;
#ifdef AB_PROBE
  long long ab_t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long ab_last;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ab_last)::"memory");
  const long long ab_first = ab_last;
#endif
  load_tile(0);
  store_tile(0);
  __syncthreads();
  AB_STAMP(0);
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    const int q0 = t * KT;
    load_tile(min(q0 + KT, (nt - 1) * KT));   // unconditional (a conditional load makes the compiler wait for it right here)
    AB_STAMP_NOWAIT(1);

    f32x16 s, dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) { s[e] = 0.0f; dp[e] = 0.0f; }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int off = l31 * AB_LDR + c * 16 + lhi * 8;
      s = ab_mma3(&Qh[buf][off], &Ql[buf][off], kh[c], kl[c], s);        // S[query][key]
      dp = ab_mma3(&Gh[buf][off], &Gl[buf][off], vh[c], vl[c], dp);      // dP * 2^12
    }
    AB_STAMP_NOWAIT(2);
    // no masks here: rows past the end carry lse = +inf (P = 0), and a key lane past the end only feeds its own columns of
    // dK / dV, which are never stored
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 l4 = *reinterpret_cast<const float4*>(&Ls[buf][8 * g + 4 * lhi]);     // rows (e & 3) + 8 (e >> 2) + 4 lhi
      const float4 d4 = *reinterpret_cast<const float4*>(&Ds[buf][8 * g + 4 * lhi]);
      const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq_[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = 4 * g + j;
        const float pv = __builtin_amdgcn_exp2f(fmaf(s[e], sc2, -lq[j]));
        s[e] = pv;
        dp[e] = pv * (dp[e] * (1.0f / AB_GS) - dq_[j]) * (scale * AB_DS);     // dS * 2^14
      }
    }
    AB_STAMP_NOWAIT(3);
    ab_half8 ph[2], pl[2], sh[2], sl[2];
    ab_acc_to_fragments(s, lhi, ph, pl);
    ab_acc_to_fragments(dp, lhi, sh, sl);
    AB_STAMP_NOWAIT(4);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int off = ab_toff(dt * 32 + l31, g * 16 + lhi * 8);
        dv_acc[dt] = ab_mma3(&Gth[buf][off], &Gtl[buf][off], ph[g], pl[g], dv_acc[dt]);   // dV^T[d][key] * 2^12
        dk_acc[dt] = ab_mma3(&Qth[buf][off], &Qtl[buf][off], sh[g], sl[g], dk_acc[dt]);   // dK^T[d][key] * 2^14
      }
    AB_STAMP_NOWAIT(5);
    if (t + 1 < nt) store_tile(buf ^ 1);
    AB_STAMP(6);
    __syncthreads();
    AB_STAMP(7);
  }

  if (k_row < T) {
    const int64_t d0 = (row0 + k_row) * ld + C + h * DH + lhi * 4;
    constexpr float ik = 1.0f / AB_DS, iv = 1.0f / AB_GS;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 vk = make_float4(dk_acc[dt][4 * g + 0] * ik, dk_acc[dt][4 * g + 1] * ik, dk_acc[dt][4 * g + 2] * ik, dk_acc[dt][4 * g + 3] * ik);
        const float4 vv = make_float4(dv_acc[dt][4 * g + 0] * iv, dv_acc[dt][4 * g + 1] * iv, dv_acc[dt][4 * g + 2] * iv, dv_acc[dt][4 * g + 3] * iv);
        if (dqkv) {
          *reinterpret_cast<float4*>(dqkv + d0 + dt * 32 + 8 * g) = vk;
          *reinterpret_cast<float4*>(dqkv + d0 + C + dt * 32 + 8 * g) = vv;
        }
        if (po.hi) {
          pfpp_store4_planes(po, d0 + dt * 32 + 8 * g, vk);
          pfpp_store4_planes(po, d0 + C + dt * 32 + 8 * g, vv);
        }
      }
  }
#ifdef AB_PROBE
  AB_STAMP(8);
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0) {
    for (int i = 0; i < 9; ++i) ab_probe_out[i] = ab_t[i];
    ab_probe_out[9] = ab_last - ab_first;
    ab_probe_out[10] = nt;
  }
#endif
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_dense_bwd_dkv_f16_kernel(
    const float* __restrict__ qkv, const float* __restrict__ dout, const float* __restrict__ lse,
    const float* __restrict__ dvec, float* __restrict__ dqkv, const int32_t* __restrict__ seq_off,
    const int32_t* __restrict__ seq_len, int H, float scale, pfpp_planes_out po) {
  pfpp_chain_prio();
  __shared__ __align__(16) char smem[AB_DKV_SMEM];
  ab_dkv_f16_body(smem, blockIdx.x, qkv, dout, lse, dvec, dqkv, seq_off, seq_len, H, scale, po);
}

// ---------------------------------------------------------------------------------------------------
// block-diagonal attention backward: one 64-thread workgroup per (fragment, head)
// ---------------------------------------------------------------------------------------------------
constexpr int BD_L = 32;
constexpr int BD_DH = 64;
constexpr int BD_LD = BD_DH + 4;    // 16-byte aligned rows, conflict-light

// 256 threads per (fragment, head): the four waves split the score entries (phase 1) and the output rows (phase 3),
// so a workgroup's ~30 k LDS reads are spread over four waves that hide each other's latency; LDS is sized for the
// actual L (dynamic), which lets 4-5 workgroups share a CU instead of 3.
__global__ __launch_bounds__(256) void attn_blockdiag_bwd_kernel(const float* __restrict__ qkv,
                                                                 const float* __restrict__ dout,
                                                                 float* __restrict__ dqkv, int L, int H, float scale) {
  extern __shared__ __align__(16) float bd_smem[];
  const int LP = L + 1;
  float* sq = bd_smem;                 // [L][BD_LD]
  float* sk = sq + L * BD_LD;
  float* sv = sk + L * BD_LD;
  float* sg = sv + L * BD_LD;
  float* sp = sg + L * BD_LD;          // [L][LP] probabilities
  float* sd = sp + L * LP;             // [L][LP] dP then dS
  const int tid = threadIdx.x;
  const int64_t pair = blockIdx.x;
  const int64_t frag = pair / H;
  const int h = (int)(pair - frag * H);
  const int C = H * BD_DH;
  const int64_t ld = 3ll * C;
  const float* base = qkv + frag * L * ld + h * BD_DH;
  const float* gbase = dout + frag * L * (int64_t)C + h * BD_DH;
  for (int i = tid; i < L * 16; i += 256) {
    const int r = i >> 4, c4 = i & 15;
    *reinterpret_cast<float4*>(&sq[r * BD_LD + c4 * 4]) = *reinterpret_cast<const float4*>(base + r * ld + c4 * 4);
    *reinterpret_cast<float4*>(&sk[r * BD_LD + c4 * 4]) = *reinterpret_cast<const float4*>(base + r * ld + C + c4 * 4);
    *reinterpret_cast<float4*>(&sv[r * BD_LD + c4 * 4]) = *reinterpret_cast<const float4*>(base + r * ld + 2 * C + c4 * 4);
    *reinterpret_cast<float4*>(&sg[r * BD_LD + c4 * 4]) = *reinterpret_cast<const float4*>(gbase + r * (int64_t)C + c4 * 4);
  }
  __syncthreads();
  // phase 1: scores and dP, one (i, j) entry per thread and pass
  for (int idx = tid; idx < L * L; idx += 256) {
    const int i = idx / L, j = idx - i * L;
    float a = 0.0f, g = 0.0f;
#pragma unroll
    for (int c4 = 0; c4 < 16; ++c4) {
      const float4 q = *reinterpret_cast<const float4*>(&sq[i * BD_LD + c4 * 4]);
      const float4 k = *reinterpret_cast<const float4*>(&sk[j * BD_LD + c4 * 4]);
      const float4 o = *reinterpret_cast<const float4*>(&sg[i * BD_LD + c4 * 4]);
      const float4 v = *reinterpret_cast<const float4*>(&sv[j * BD_LD + c4 * 4]);
      a = fmaf(q.x, k.x, a); a = fmaf(q.y, k.y, a); a = fmaf(q.z, k.z, a); a = fmaf(q.w, k.w, a);
      g = fmaf(o.x, v.x, g); g = fmaf(o.y, v.y, g); g = fmaf(o.z, v.z, g); g = fmaf(o.w, v.w, g);
    }
    sp[i * LP + j] = a * scale;
    sd[i * LP + j] = g;
  }
  __syncthreads();
  // phase 2: softmax rows, then dS = P*(dP - sum_j P*dP)*scale (one thread per row: L <= 32)
  if (tid < L) {
    float m = -__builtin_huge_valf();
    for (int j = 0; j < L; ++j) m = fmaxf(m, sp[tid * LP + j]);
    float sum = 0.0f;
    for (int j = 0; j < L; ++j) {
      const float e = expf(sp[tid * LP + j] - m);
      sp[tid * LP + j] = e;
      sum += e;
    }
    const float inv = 1.0f / sum;
    float dsum = 0.0f;
    for (int j = 0; j < L; ++j) {
      const float pj = sp[tid * LP + j] * inv;
      sp[tid * LP + j] = pj;
      dsum += pj * sd[tid * LP + j];
    }
    for (int j = 0; j < L; ++j) sd[tid * LP + j] = sp[tid * LP + j] * (sd[tid * LP + j] - dsum) * scale;
  }
  __syncthreads();
  // phase 3: wave w takes rows w, w+4, ...; lane = head-dim column
  const int d = tid & 63;
  float* obase = dqkv + frag * L * ld + h * BD_DH + d;
  for (int i = tid >> 6; i < L; i += 4) {
    float dq = 0.0f, dk = 0.0f, dv = 0.0f;
    for (int j = 0; j < L; ++j) {
      dq = fmaf(sd[i * LP + j], sk[j * BD_LD + d], dq);       // dQ[i] = sum_j dS[i][j] K[j]
      dk = fmaf(sd[j * LP + i], sq[j * BD_LD + d], dk);       // dK[i] = sum_j dS[j][i] Q[j]
      dv = fmaf(sp[j * LP + i], sg[j * BD_LD + d], dv);       // dV[i] = sum_j P[j][i] dO[j]
    }
    obase[i * ld] = dq;
    obase[i * ld + C] = dk;
    obase[i * ld + 2 * C] = dv;
  }
}

// The same on the matrix cores, one wave per (fragment, head), no LDS.  The four rows a lane loads (its query's q and dO
// row, its key's k and v row: the 4 dims at 8c + 4*lhi of every 8-chunk) serve as the A or the B operand alike, so both
// orientations of every 32x32 product come from the same registers:
//   queries on lanes:  S^T = K.Q^T, dP^T = V.dO^T  ->  P^T, D, dS^T  ->  dQ^T = K^T.dS^T      (accumulator = B operand)
//   keys on lanes:     S = Q.K^T,   dP = dO.V^T    ->  P, dS          ->  dV^T = dO^T.P,  dK^T = Q^T.dS
// the per-query softmax statistics (max, 1/sum, D) of the first orientation reach the second through lane reads.
// 224 MFMAs (14 k cycles) per pair; the LDS/VALU version took 45 us for the 1232 pairs of the benchmark step.
typedef float bdm_f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void attn_blockdiag_bwd_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                      float* __restrict__ dqkv, int64_t n_pairs, int L, int H,
                                                                      float scale, pfpp_planes_out po) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int64_t pair = (int64_t)blockIdx.x * 4 + wave;
  if (pair >= n_pairs) return;
  const int64_t frag = pair / H;
  const int h = (int)(pair - frag * H);
  const int C = H * BD_DH;
  const int64_t ld = 3ll * C;
  const float* base = qkv + frag * L * ld + h * BD_DH;                     // q of row 0; k at +C, v at +2C
  const float* dob = dout + frag * L * (int64_t)C + h * BD_DH;
  const int64_t gb = frag * L * ld + h * BD_DH;           // element offset of this pair's dq rows in dqkv (dk at +C, dv at +2C)
  const int row = l31 < L ? l31 : L - 1;

  // ---- this lane's row pieces (re-read for every product: they stay in L1/L2, and holding all four rows for the whole
  // kernel costs 128 registers = one wave per SIMD and nothing to hide the gather latency of the second-stage products) ----
  const float* qrow = base + row * ld + lhi * 4;
  const float* krow = qrow + C;
  const float* vrow = qrow + 2 * C;
  const float* grow = dob + row * (int64_t)C + lhi * 4;
  auto prod = [&](const float* a, const float* b) {       // [rows of a] x [rows of b], contraction over the 64 dims
    bdm_f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float4 av = *reinterpret_cast<const float4*>(a + c * 8);
      const float4 bv = *reinterpret_cast<const float4*>(b + c * 8);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
    }
    return acc;
  };
  // [64 dims] x [32 lanes] product with the accumulator `b` as the B operand: out^T[dim][lane] = sum_t src[idx(t)][dim] * b[t]
  auto second = [&](const float* src, int64_t src_ld, const bdm_f32x16& b, int64_t dst) {
#pragma unroll
    for (int tile = 0; tile < 2; ++tile) {
      bdm_f32x16 o;
#pragma unroll
      for (int e = 0; e < 16; ++e) o[e] = 0.0f;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int idx = (t & 3) + 8 * (t >> 2) + 4 * lhi;
        const float a = src[(int64_t)(idx < L ? idx : L - 1) * src_ld + tile * 32 + l31];     // rows >= L meet b[t] = 0
        o = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[t], o, 0, 0, 0);
      }
      if (l31 < L) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
          const int64_t idx = dst + l31 * ld + tile * 32 + 8 * q + 4 * lhi;
          if (dqkv) *reinterpret_cast<float4*>(dqkv + idx) = v;
          if (po.hi) pfpp_store4_planes(po, idx, v);
        }
      }
    }
  };

  // ---- queries on lanes ----
  bdm_f32x16 st = prod(krow, qrow);            // st[e]: key (e&3)+8*(e>>2)+4*lhi, query l31
  float mx = -__builtin_huge_valf();
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int key = (e & 3) + 8 * (e >> 2) + 4 * lhi;
    st[e] = key < L ? st[e] * scale : -__builtin_huge_valf();
    mx = fmaxf(mx, st[e]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float sum = 0.0f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int key = (e & 3) + 8 * (e >> 2) + 4 * lhi;
    st[e] = key < L ? __expf(st[e] - mx) : 0.0f;
    sum += st[e];
  }
  sum += __shfl_xor(sum, 32);
  const float inv = 1.0f / sum;
  bdm_f32x16 dpt = prod(vrow, grow);           // dP^T: key x query
  float dsum = 0.0f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    st[e] *= inv;                          // P^T
    dsum += st[e] * dpt[e];
  }
  dsum += __shfl_xor(dsum, 32);            // D[query]
#pragma unroll
  for (int e = 0; e < 16; ++e) dpt[e] = st[e] * (dpt[e] - dsum) * scale;      // dS^T (0 at masked keys)
  second(base + C, ld, dpt, gb);                                               // dQ^T = K^T . dS^T  -> dq rows

  // ---- keys on lanes ----
  bdm_f32x16 sk = prod(qrow, krow);            // sk[e]: query (e&3)+8*(e>>2)+4*lhi, key l31
  bdm_f32x16 dp = prod(grow, vrow);            // dP: query x key
  bdm_f32x16 pk;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int qi = (e & 3) + 8 * (e >> 2) + 4 * lhi;
    const float m_q = __shfl(mx, qi), i_q = __shfl(inv, qi), d_q = __shfl(dsum, qi);
    const float pv = qi < L ? __expf(sk[e] * scale - m_q) * i_q : 0.0f;          // queries >= L do not exist
    pk[e] = pv;
    dp[e] = pv * (dp[e] - d_q) * scale;                                         // dS
  }
  second(dob, C, pk, gb + 2 * C);                                              // dV^T = dO^T . P   -> dv rows
  second(base, ld, dp, gb + C);                                                // dK^T = Q^T . dS   -> dk rows
}


// The same with the split-f16 contraction (three v_mfma_f32_32x32x16_f16 per 16-deep step: 84 matrix instructions of 32 cycles per
// pair instead of 224 of 64).  One wave = one workgroup = one (fragment, head).  The lane's row pieces of q, k, v, dO are split
// once into operand fragments that serve both orientations of the four first-stage products; the second-stage products need
// q, k, dO TRANSPOSED (dims on lanes, tokens along the contraction): their row-major planes go to LDS once and come back
// through ds_read_b64_tr_b16 (each lane of a 16-lane group gets the 4 token-consecutive halfs of its own dim).  Gradient
// operands are lifted like in the dense passes (dO by 2^12, dS by 2^14).
constexpr int BDF_LD = BD_DH + 8;                      // halfs per token row of an LDS plane
constexpr int BDF_PLANE = BD_L * BDF_LD * 2;           // bytes

// 8 transposed operand fragments (2 dim tiles x 2 token halves, hi and lo) of the planes at LDS byte address `ad` (hi; lo one plane up)
__device__ __forceinline__ void bdf_read_tr(uint32_t ad, ab_half8 (&ah)[2][2], ab_half8 (&al)[2][2]) {
  ab_half4 h[2][2][2], l[2][2][2];      // [tile][g][t]
#define BDF_OFF(tile, g, t, pl) ((pl) * BDF_PLANE + ((16 * (g) + 4 * (t)) * BDF_LD + 32 * (tile)) * 2)
  asm volatile(
      "ds_read_b64_tr_b16 %0, %16 offset:%17\n\tds_read_b64_tr_b16 %1, %16 offset:%18\n\t"
      "ds_read_b64_tr_b16 %2, %16 offset:%19\n\tds_read_b64_tr_b16 %3, %16 offset:%20\n\t"
      "ds_read_b64_tr_b16 %4, %16 offset:%21\n\tds_read_b64_tr_b16 %5, %16 offset:%22\n\t"
      "ds_read_b64_tr_b16 %6, %16 offset:%23\n\tds_read_b64_tr_b16 %7, %16 offset:%24\n\t"
      "ds_read_b64_tr_b16 %8, %16 offset:%25\n\tds_read_b64_tr_b16 %9, %16 offset:%26\n\t"
      "ds_read_b64_tr_b16 %10, %16 offset:%27\n\tds_read_b64_tr_b16 %11, %16 offset:%28\n\t"
      "ds_read_b64_tr_b16 %12, %16 offset:%29\n\tds_read_b64_tr_b16 %13, %16 offset:%30\n\t"
      "ds_read_b64_tr_b16 %14, %16 offset:%31\n\tds_read_b64_tr_b16 %15, %16 offset:%32\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(h[0][0][0]), "=&v"(h[0][0][1]), "=&v"(h[0][1][0]), "=&v"(h[0][1][1]), "=&v"(h[1][0][0]), "=&v"(h[1][0][1]),
        "=&v"(h[1][1][0]), "=&v"(h[1][1][1]), "=&v"(l[0][0][0]), "=&v"(l[0][0][1]), "=&v"(l[0][1][0]), "=&v"(l[0][1][1]),
        "=&v"(l[1][0][0]), "=&v"(l[1][0][1]), "=&v"(l[1][1][0]), "=&v"(l[1][1][1])
      : "v"(ad), "n"(BDF_OFF(0, 0, 0, 0)), "n"(BDF_OFF(0, 0, 1, 0)), "n"(BDF_OFF(0, 1, 0, 0)), "n"(BDF_OFF(0, 1, 1, 0)),
        "n"(BDF_OFF(1, 0, 0, 0)), "n"(BDF_OFF(1, 0, 1, 0)), "n"(BDF_OFF(1, 1, 0, 0)), "n"(BDF_OFF(1, 1, 1, 0)),
        "n"(BDF_OFF(0, 0, 0, 1)), "n"(BDF_OFF(0, 0, 1, 1)), "n"(BDF_OFF(0, 1, 0, 1)), "n"(BDF_OFF(0, 1, 1, 1)),
        "n"(BDF_OFF(1, 0, 0, 1)), "n"(BDF_OFF(1, 0, 1, 1)), "n"(BDF_OFF(1, 1, 0, 1)), "n"(BDF_OFF(1, 1, 1, 1))
      : "memory");
#undef BDF_OFF
#pragma unroll
  for (int tile = 0; tile < 2; ++tile)
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        ah[tile][g][q] = h[tile][g][0][q]; ah[tile][g][4 + q] = h[tile][g][1][q];
        al[tile][g][q] = l[tile][g][0][q]; al[tile][g][4 + q] = l[tile][g][1][q];
      }
}

__device__ __forceinline__ f32x16 bdf_mma3(const ab_half8 ah, const ab_half8 al, const ab_half8 bh, const ab_half8 bl, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
  return acc;
}

__global__ __launch_bounds__(64) void attn_blockdiag_bwd_f16_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                    float* __restrict__ dqkv, int64_t n_pairs, int L, int H,
                                                                    float scale, pfpp_planes_out po) {
  pfpp_chain_prio();
  __shared__ __align__(16) _Float16 planes[3][2][BD_L * BDF_LD];      // q, k, dO x (hi, lo), row-major [token][dim]
  const int lane = threadIdx.x;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int64_t pair = blockIdx.x;
  const int64_t frag = pair / H;
  const int h = (int)(pair - frag * H);
  const int C = H * BD_DH;
  const int64_t ld = 3ll * C;
  const float* base = qkv + frag * L * ld + h * BD_DH;                     // q of row 0; k at +C, v at +2C
  const float* dob = dout + frag * L * (int64_t)C + h * BD_DH;
  const int64_t gb = frag * L * ld + h * BD_DH;           // element offset of this pair's dq rows in dqkv (dk at +C, dv at +2C)
  const int row = l31 < L ? l31 : L - 1;

  // ---- this lane's row pieces -> operand fragments (registers) and, for q / k / dO, row-major planes in LDS ----
  ab_half8 qh[4], ql[4], kh[4], kl[4], vh[4], vl[4], gh[4], gl[4];
  {
    const float* qrow = base + row * ld + lhi * 8;
    const float* grow = dob + row * (int64_t)C + lhi * 8;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      ab_frag8(qrow + c * 16, 1.0f, qh[c], ql[c]);
      ab_frag8(qrow + C + c * 16, 1.0f, kh[c], kl[c]);
      ab_frag8(qrow + 2 * C + c * 16, 1.0f, vh[c], vl[c]);
      ab_frag8(grow + c * 16, AB_GS, gh[c], gl[c]);
      const int off = l31 * BDF_LD + c * 16 + lhi * 8;
      *reinterpret_cast<ab_half8*>(&planes[0][0][off]) = qh[c];
      *reinterpret_cast<ab_half8*>(&planes[0][1][off]) = ql[c];
      *reinterpret_cast<ab_half8*>(&planes[1][0][off]) = kh[c];
      *reinterpret_cast<ab_half8*>(&planes[1][1][off]) = kl[c];
      *reinterpret_cast<ab_half8*>(&planes[2][0][off]) = gh[c];
      *reinterpret_cast<ab_half8*>(&planes[2][1][off]) = gl[c];
    }
  }
  // transposing reads: lane (q4 = lane >> 4, j = lane & 15) supplies token row 8 (q4 >> 1) + (j >> 2), dims 16 (q4 & 1) + 4 (j & 3)
  const int q4 = lane >> 4, j = lane & 15;
  const uint32_t tr_off = (uint32_t)(((8 * (q4 >> 1) + (j >> 2)) * BDF_LD + 16 * (q4 & 1) + 4 * (j & 3)) * 2);
  const uint32_t lds_q = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&planes[0][0][0] + tr_off;
  const uint32_t lds_k = lds_q + 2 * BDF_PLANE, lds_g = lds_q + 4 * BDF_PLANE;

  auto prod = [&](const ab_half8 (&ah)[4], const ab_half8 (&al)[4], const ab_half8 (&bh)[4], const ab_half8 (&bl)[4]) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) acc = bdf_mma3(ah[c], al[c], bh[c], bl[c], acc);
    return acc;
  };
  // out^T[dim][lane] = sum_t plane[t][dim] * b[t][lane], written to the rows of dqkv at dst; b comes as accumulator (token rows t >= L
  // carry zeros there, so the clamped duplicates in the planes do not count)
  auto second = [&](uint32_t lds_ad, const f32x16& b, float out_scale, int64_t dst) {
    ab_half8 bh[2], bl[2], ah[2][2], al[2][2];
    ab_acc_to_fragments(b, lhi, bh, bl);
    bdf_read_tr(lds_ad, ah, al);
#pragma unroll
    for (int tile = 0; tile < 2; ++tile) {
      f32x16 o;
#pragma unroll
      for (int e = 0; e < 16; ++e) o[e] = 0.0f;
#pragma unroll
      for (int g = 0; g < 2; ++g) o = bdf_mma3(ah[tile][g], al[tile][g], bh[g], bl[g], o);
      if (l31 < L) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = make_float4(o[4 * q] * out_scale, o[4 * q + 1] * out_scale, o[4 * q + 2] * out_scale, o[4 * q + 3] * out_scale);
          const int64_t idx = dst + l31 * ld + tile * 32 + 8 * q + 4 * lhi;
          if (dqkv) *reinterpret_cast<float4*>(dqkv + idx) = v;
          if (po.hi) pfpp_store4_planes(po, idx, v);
        }
      }
    }
  };

  // ---- queries on lanes ----
  f32x16 st = prod(kh, kl, qh, ql);            // st[e]: key (e&3)+8*(e>>2)+4*lhi, query l31
  float mx = -__builtin_huge_valf();
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int key = (e & 3) + 8 * (e >> 2) + 4 * lhi;
    st[e] = key < L ? st[e] * scale : -__builtin_huge_valf();
    mx = fmaxf(mx, st[e]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float sum = 0.0f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    st[e] = __expf(st[e] - mx);                 // masked keys: exp(-inf) = 0
    sum += st[e];
  }
  sum += __shfl_xor(sum, 32);
  const float inv = 1.0f / sum;
  f32x16 dpt = prod(vh, vl, gh, gl);           // dP^T * 2^12: key x query
  float dsum = 0.0f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    st[e] *= inv;                          // P^T
    dpt[e] *= 1.0f / AB_GS;
    dsum += st[e] * dpt[e];
  }
  dsum += __shfl_xor(dsum, 32);            // D[query]
#pragma unroll
  for (int e = 0; e < 16; ++e) dpt[e] = st[e] * (dpt[e] - dsum) * (scale * AB_DS);      // dS^T * 2^14 (0 at masked keys)
  second(lds_k, dpt, 1.0f / AB_DS, gb);                                        // dQ^T = K^T . dS^T  -> dq rows

  // ---- keys on lanes ----
  f32x16 sk = prod(qh, ql, kh, kl);            // sk[e]: query (e&3)+8*(e>>2)+4*lhi, key l31
  f32x16 dp = prod(gh, gl, vh, vl);            // dP * 2^12: query x key
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int qi = (e & 3) + 8 * (e >> 2) + 4 * lhi;
    const float m_q = __shfl(mx, qi), i_q = __shfl(inv, qi), d_q = __shfl(dsum, qi);
    const float ev = __expf(sk[e] * scale - m_q) * i_q;
    const float pv = qi < L ? ev : 0.0f;                                        // queries >= L do not exist
    sk[e] = pv;
    dp[e] = pv * (dp[e] * (1.0f / AB_GS) - d_q) * (scale * AB_DS);              // dS * 2^14
  }
  second(lds_g, sk, 1.0f / AB_GS, gb + 2 * C);                                 // dV^T = dO^T . P   -> dv rows
  second(lds_q, dp, 1.0f / AB_DS, gb + C);                                     // dK^T = Q^T . dS   -> dk rows
}

// Forward of the same attention with the split-f16 contraction (attention.py:77-85 per fragment; the exact-fp32 form is
// attn_blockdiag_mfma_kernel in transformer_ops.hip: 64 matrix instructions of 64 cycles, this one 24 of 32).  One wave = one workgroup =
// one (fragment, head): S^T = K . Q^T from the lanes' row fragments, softmax over the keys in registers, O^T = V^T . P^T with V^T read
// back transposed from its row-major LDS planes and P^T taken from the accumulator.  The probabilities are lifted by 2^11 before the
// split (unlifted, the lo halves of every p < 1/8 are fp16 subnormals: twice the mean error).
constexpr float BDF_PS = 2048.0f;
__global__ __launch_bounds__(64) void attn_blockdiag_f16_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                _Float16* __restrict__ out_hi, _Float16* __restrict__ out_lo,
                                                                int64_t n_pairs, int L, int H, float scale) {
  pfpp_chain_prio();
  __shared__ __align__(16) _Float16 planes[2][BD_L * BDF_LD];       // v (hi, lo), row-major [token][dim]
  const int lane = threadIdx.x;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int64_t pair = blockIdx.x;
  const int64_t frag = pair / H;
  const int h = (int)(pair - frag * H);
  const int C = H * BD_DH;
  const int64_t ld = 3ll * C;
  const float* base = qkv + frag * L * ld + h * BD_DH;
  const int row = l31 < L ? l31 : L - 1;
  ab_half8 qh[4], ql[4], kh[4], kl[4];
  {
    const float* qrow = base + row * ld + lhi * 8;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      ab_half8 vh, vl;
      ab_frag8(qrow + c * 16, 1.0f, qh[c], ql[c]);
      ab_frag8(qrow + C + c * 16, 1.0f, kh[c], kl[c]);
      ab_frag8(qrow + 2 * C + c * 16, 1.0f, vh, vl);
      const int off = l31 * BDF_LD + c * 16 + lhi * 8;
      *reinterpret_cast<ab_half8*>(&planes[0][off]) = vh;
      *reinterpret_cast<ab_half8*>(&planes[1][off]) = vl;
    }
  }
  const int q4 = lane >> 4, j = lane & 15;
  const uint32_t lds_v = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&planes[0][0] +
                         (uint32_t)(((8 * (q4 >> 1) + (j >> 2)) * BDF_LD + 16 * (q4 & 1) + 4 * (j & 3)) * 2);
  f32x16 st;
#pragma unroll
  for (int e = 0; e < 16; ++e) st[e] = 0.0f;
#pragma unroll
  for (int c = 0; c < 4; ++c) st = bdf_mma3(kh[c], kl[c], qh[c], ql[c], st);      // st[e]: key (e&3)+8*(e>>2)+4*lhi, query l31
  float mx = -__builtin_huge_valf();
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int key = (e & 3) + 8 * (e >> 2) + 4 * lhi;
    st[e] = key < L ? st[e] * scale : -__builtin_huge_valf();
    mx = fmaxf(mx, st[e]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float sum = 0.0f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    st[e] = __expf(st[e] - mx);                 // masked keys: exp(-inf) = 0
    sum += st[e];
  }
  sum += __shfl_xor(sum, 32);
  const float inv = BDF_PS / sum;
#pragma unroll
  for (int e = 0; e < 16; ++e) st[e] *= inv;
  ab_half8 bh[2], bl[2], ah[2][2], al[2][2];
  ab_acc_to_fragments(st, lhi, bh, bl);
  bdf_read_tr(lds_v, ah, al);
  const int64_t orow = (frag * L + l31) * (int64_t)C + h * BD_DH;
#pragma unroll
  for (int tile = 0; tile < 2; ++tile) {
    f32x16 o;
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] = 0.0f;
#pragma unroll
    for (int g = 0; g < 2; ++g) o = bdf_mma3(ah[tile][g], al[tile][g], bh[g], bl[g], o);
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] *= 1.0f / BDF_PS;
    if (l31 < L) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t off = orow + tile * 32 + 8 * q + 4 * lhi;
        if (out_hi) {
          ab_half4 hi4, lo4;
#pragma unroll
          for (int r = 0; r < 4; ++r) { _Float16 a, b; ab_split(o[4 * q + r], a, b); hi4[r] = a; lo4[r] = b; }
          *reinterpret_cast<ab_half4*>(out_hi + off) = hi4;
          *reinterpret_cast<ab_half4*>(out_lo + off) = lo4;
        } else {
          *reinterpret_cast<float4*>(out + off) = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
        }
      }
    }
  }
}

}  // namespace

extern "C" int pfpp_attn_blockdiag_bwd(const float* qkv, const float* dout, float* dqkv, int64_t n_frag, int64_t L,
                                       int64_t H, int64_t dh, float scale, pfpp_stream_t stream) {
  PFPP_REQUIRE(dqkv, "null pointer");
  return pfpp_attn_blockdiag_bwd_p(qkv, dout, dqkv, n_frag, L, H, dh, scale, nullptr, stream);
}

extern "C" int pfpp_attn_blockdiag_bwd_p(const float* qkv, const float* dout, float* dqkv, int64_t n_frag, int64_t L,
                                         int64_t H, int64_t dh, float scale, const pfpp_planes* dqkv_planes,
                                         pfpp_stream_t stream) {
  PFPP_REQUIRE(qkv && dout && (dqkv || dqkv_planes) && pfpp_planes_ok(dqkv_planes), "null pointer");
  PFPP_SUPPORTED(dh == BD_DH, "dim_head != 64");
  PFPP_SUPPORTED(L >= 1 && L <= BD_L, "L outside [1, 32]");
  PFPP_REQUIRE(pfpp::aligned16(qkv) && pfpp::aligned16(dout), "16-byte alignment");
  const int64_t pairs = n_frag * H;
  if (pairs == 0) return PFPP_OK;
  static const bool use_mfma = !(getenv("PFPP_ATTN_BD_MFMA") && atoi(getenv("PFPP_ATTN_BD_MFMA")) == 0);
  // split-f16 contraction unless PFPP_ATTN_BD_F16X3=0 (then the exact fp32 matrix instructions)
  static const bool use_f16 = !(getenv("PFPP_ATTN_BD_F16X3") && atoi(getenv("PFPP_ATTN_BD_F16X3")) == 0);
  if (use_f16 && (use_mfma || dqkv_planes)) {
    PFPP_SUPPORTED(pairs <= 0x7fffffff, "too many (fragment, head) pairs for one launch");
    hipLaunchKernelGGL(attn_blockdiag_bwd_f16_kernel, dim3((unsigned)pairs), dim3(64), 0, pfpp::as_stream(stream), qkv, dout, dqkv,
                       pairs, (int)L, (int)H, scale, pfpp_planes_arg(dqkv_planes));
    return pfpp::check_launch("pfpp_attn_blockdiag_bwd");
  }
  if (use_mfma || dqkv_planes) {
    hipLaunchKernelGGL(attn_blockdiag_bwd_mfma_kernel, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, pfpp::as_stream(stream), qkv,
                       dout, dqkv, pairs, (int)L, (int)H, scale, pfpp_planes_arg(dqkv_planes));
    return pfpp::check_launch("pfpp_attn_blockdiag_bwd");
  }
  const size_t smem = (size_t)(4 * L * BD_LD + 2 * L * (L + 1)) * sizeof(float);
  hipLaunchKernelGGL(attn_blockdiag_bwd_kernel, dim3((unsigned)pairs), dim3(256), smem, pfpp::as_stream(stream), qkv, dout,
                     dqkv, (int)L, (int)H, scale);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_attn_dense_bwd(const float* qkv, const float* out, const float* dout, const float* lse,
                                   float* dvec, float* dqkv, const int32_t* seq_off, const int32_t* seq_len,
                                   const uint8_t* key_valid, int64_t kv_stride, int64_t n_seq, int64_t max_len,
                                   int64_t H, int64_t dh, float scale, pfpp_stream_t stream) {
  PFPP_REQUIRE(dqkv, "null pointer");
  return pfpp_attn_dense_bwd_p(qkv, out, dout, lse, dvec, dqkv, seq_off, seq_len, key_valid, kv_stride, n_seq, max_len, H, dh,
                               scale, nullptr, stream);
}

extern "C" int pfpp_attn_dense_bwd_p(const float* qkv, const float* out, const float* dout, const float* lse,
                                     float* dvec, float* dqkv, const int32_t* seq_off, const int32_t* seq_len,
                                     const uint8_t* key_valid, int64_t kv_stride, int64_t n_seq, int64_t max_len,
                                     int64_t H, int64_t dh, float scale, const pfpp_planes* dqkv_planes,
                                     pfpp_stream_t stream) {
  PFPP_REQUIRE(qkv && out && dout && lse && dvec && (dqkv || dqkv_planes) && seq_off && seq_len && pfpp_planes_ok(dqkv_planes),
               "null pointer");
  static const int f16_mode_p = getenv("PFPP_ATTN_F16X3") ? atoi(getenv("PFPP_ATTN_F16X3")) : 1;
  PFPP_SUPPORTED(!dqkv_planes || (dh == 64 && f16_mode_p >= 1 && key_valid == nullptr),
                 "plane output of the attention backward needs the split-f16 passes (dim_head 64, no key mask)");
  const pfpp_planes_out po = pfpp_planes_arg(dqkv_planes);
  PFPP_REQUIRE(n_seq >= 0 && max_len >= 1 && H >= 1, "bad sizes");
  PFPP_SUPPORTED(dh == 64 || dh == 32, "dim_head must be 32 or 64");
  PFPP_SUPPORTED(n_seq <= 65535 && H <= 65535, "too many sequences / heads for one launch");
  PFPP_REQUIRE(pfpp::aligned16(qkv) && pfpp::aligned16(out) && pfpp::aligned16(dout) && pfpp::aligned16(dqkv),
               "16-byte alignment");
  if (n_seq == 0) return PFPP_OK;
  const dim3 grid((unsigned)((max_len + 127) / 128), (unsigned)H, (unsigned)n_seq);
  hipStream_t st = pfpp::as_stream(stream);
  // split-f16 passes for the unmasked (ragged) launches, like pfpp_attn_dense (PFPP_ATTN_F16X3: 0 never, 1 unmasked, 2 -)
  static const int f16_mode = getenv("PFPP_ATTN_F16X3") ? atoi(getenv("PFPP_ATTN_F16X3")) : 1;
  if (dh == 64 && f16_mode >= 1 && key_valid == nullptr) {
    // (both passes in one launch was built and measured twice — rounds 2 and 3: no gain, the longest sequence's workgroups of the two
    // passes then share SIMDs and each walks its tiles more slowly than alone — and removed in round 4)
    hipLaunchKernelGGL(attn_dense_bwd_dq_f16_kernel, grid, dim3(256), 0, st, qkv, out, dout, lse, dvec, dqkv, seq_off, seq_len,
                       (int)H, scale, po);
    hipLaunchKernelGGL(attn_dense_bwd_dkv_f16_kernel, grid, dim3(256), 0, st, qkv, dout, lse, dvec, dqkv, seq_off, seq_len,
                       (int)H, scale, po);
    return pfpp::check_launch("pfpp_attn_dense_bwd");
  }
  if (dh == 64) {
    hipLaunchKernelGGL(attn_dense_bwd_dq_kernel<64>, grid, dim3(256), 0, st, qkv, out, dout, lse, dvec, dqkv, seq_off,
                       seq_len, key_valid, kv_stride, (int)H, scale);
    hipLaunchKernelGGL(attn_dense_bwd_dkv_kernel<64>, grid, dim3(256), 0, st, qkv, dout, lse, dvec, dqkv, seq_off,
                       seq_len, key_valid, kv_stride, (int)H, scale);
  } else {
    hipLaunchKernelGGL(attn_dense_bwd_dq_kernel<32>, grid, dim3(256), 0, st, qkv, out, dout, lse, dvec, dqkv, seq_off,
                       seq_len, key_valid, kv_stride, (int)H, scale);
    hipLaunchKernelGGL(attn_dense_bwd_dkv_kernel<32>, grid, dim3(256), 0, st, qkv, dout, lse, dvec, dqkv, seq_off,
                       seq_len, key_valid, kv_stride, (int)H, scale);
  }
  return pfpp::check_launch(__func__);
}

// The same backward in separately launchable parts (bit 0: D = rowsum(dO . O) -> dvec, bit 1: dq, bit 2: dk/dv), so that a
// caller can put dq and dk/dv on two streams once D is there (they write disjoint columns of dqkv).
extern "C" int pfpp_attn_dense_bwd_parts(const float* qkv, const float* out, const float* dout, const float* lse,
                                         float* dvec, float* dqkv, const int32_t* seq_off, const int32_t* seq_len,
                                         const uint8_t* key_valid, int64_t kv_stride, int64_t n_seq, int64_t max_len,
                                         int64_t H, int64_t dh, float scale, int parts, pfpp_stream_t stream) {
  PFPP_REQUIRE(qkv && out && dout && lse && dvec && dqkv && seq_off && seq_len, "null pointer");
  PFPP_REQUIRE(n_seq >= 0 && max_len >= 1 && H >= 1 && parts >= 1 && parts <= 7, "bad sizes / parts");
  PFPP_SUPPORTED(dh == 64 || dh == 32, "dim_head must be 32 or 64");
  PFPP_SUPPORTED(n_seq <= 65535 && H <= 65535, "too many sequences / heads for one launch");
  PFPP_REQUIRE(pfpp::aligned16(qkv) && pfpp::aligned16(out) && pfpp::aligned16(dout) && pfpp::aligned16(dqkv),
               "16-byte alignment");
  if (n_seq == 0) return PFPP_OK;
  const dim3 grid((unsigned)((max_len + 127) / 128), (unsigned)H, (unsigned)n_seq);
  hipStream_t st = pfpp::as_stream(stream);
  const int Hi = (int)H;
  static const int f16_mode = getenv("PFPP_ATTN_F16X3") ? atoi(getenv("PFPP_ATTN_F16X3")) : 1;
  if (dh == 64 && f16_mode >= 1 && key_valid == nullptr) {      // same kernel choice as pfpp_attn_dense_bwd
    if (parts & 1) hipLaunchKernelGGL(attn_dense_bwd_d_kernel<64>, grid, dim3(256), 0, st, out, dout, dvec, seq_off, seq_len, Hi);
    const pfpp_planes_out none = pfpp_planes_arg(nullptr);
    if (parts & 2) hipLaunchKernelGGL(attn_dense_bwd_dq_f16_kernel, grid, dim3(256), 0, st, qkv, out, dout, lse, (float*)nullptr, dqkv,
                                      seq_off, seq_len, Hi, scale, none);
    if (parts & 4) hipLaunchKernelGGL(attn_dense_bwd_dkv_f16_kernel, grid, dim3(256), 0, st, qkv, dout, lse, dvec, dqkv, seq_off,
                                      seq_len, Hi, scale, none);
    return pfpp::check_launch(__func__);
  }
  if (dh == 64) {
    if (parts & 1) hipLaunchKernelGGL(attn_dense_bwd_d_kernel<64>, grid, dim3(256), 0, st, out, dout, dvec, seq_off, seq_len, Hi);
    if (parts & 2) hipLaunchKernelGGL(attn_dense_bwd_dq_kernel<64>, grid, dim3(256), 0, st, qkv, out, dout, lse, (float*)nullptr, dqkv,
                                      seq_off, seq_len, key_valid, kv_stride, Hi, scale);
    if (parts & 4) hipLaunchKernelGGL(attn_dense_bwd_dkv_kernel<64>, grid, dim3(256), 0, st, qkv, dout, lse, dvec, dqkv, seq_off,
                                      seq_len, key_valid, kv_stride, Hi, scale);
  } else {
    if (parts & 1) hipLaunchKernelGGL(attn_dense_bwd_d_kernel<32>, grid, dim3(256), 0, st, out, dout, dvec, seq_off, seq_len, Hi);
    if (parts & 2) hipLaunchKernelGGL(attn_dense_bwd_dq_kernel<32>, grid, dim3(256), 0, st, qkv, out, dout, lse, (float*)nullptr, dqkv,
                                      seq_off, seq_len, key_valid, kv_stride, Hi, scale);
    if (parts & 4) hipLaunchKernelGGL(attn_dense_bwd_dkv_kernel<32>, grid, dim3(256), 0, st, qkv, dout, lse, dvec, dqkv, seq_off,
                                      seq_len, key_valid, kv_stride, Hi, scale);
  }
  return pfpp::check_launch(__func__);
}

// internal: the split-f16 forward of the per-fragment attention (called by pfpp_attn_blockdiag / _split in transformer_ops.hip)
int pfpp_attn_blockdiag_f16_launch(const float* qkv, float* out, void* out_hi, void* out_lo, int64_t pairs, int64_t L, int64_t H, float scale,
                                   hipStream_t st) {
  hipLaunchKernelGGL(attn_blockdiag_f16_kernel, dim3((unsigned)pairs), dim3(64), 0, st, qkv, out, (_Float16*)out_hi, (_Float16*)out_lo, pairs,
                     (int)L, (int)H, scale);
  return pfpp::check_launch("pfpp_attn_blockdiag");
}
