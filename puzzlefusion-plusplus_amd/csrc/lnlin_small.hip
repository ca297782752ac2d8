// LayerNorm (AdaLN or affine) fused into the linear layer that follows it, for SMALL token counts (one to a few puzzles in flight:
// 25-2,000 tokens; pfpp_hip hands over to the tiled GEMMs above 2,048, measured in the auto_aggl loop with 2-32 puzzles in flight).
//
// Reference: MyAdaLayerNorm / nn.LayerNorm followed by to_q|to_k|to_v resp. the GEGLU projection in EncoderLayer.forward
// (denoiser/model/modules/attention.py:21-25, 77-90), eval mode, as issued by pfpp_hip.denoiser.denoiser_forward_compact.
//
// Why: with one puzzle in flight the DDPM step is ~100 dependent launches of 5-17 us; 18 of them are LayerNorms of 5 us + a launch gap
// each, every one followed by a GEMM whose A operand they produce.  For few tokens the normalised rows need not exist in HBM at all
// (no backward, nothing else reads them): a workgroup normalises its 32 rows itself, keeps them in LDS as split-f16 planes and contracts
// them with its share of the weight columns.
//
// Shape of a launch (round 4, second form).  The first form gave every 64 weight columns a workgroup of their own (contraction cut over
// its four waves, partial sums through LDS): 168-1024 workgroups of ~130 KB LDS each, i.e. one per CU and up to four rounds of them, every
// one normalising its rows again — with all loads, MFMAs and stores ablated away the launch still took 8.2 us of its 15.9 (plain) / 24.7
// (GEGLU), and the twelve dependent ds_bpermute of each row's two wave sums another 2.5-3.6 us (tools/diag/lab_lnlin.sh).  Now:
//   * grid = (row tiles of 32, column groups), groups chosen on the host so that the launch is ONE round of <= 256 workgroups; a
//     workgroup normalises its rows once and its four waves then work through the group's column units independently — a unit = 32
//     output columns with the FULL contraction in one accumulator chain: no partial sums, no workgroup barrier after the LayerNorm;
//   * a wave streams its unit's weights in 128-deep (GEGLU: 64-deep, two tiles) chunks, chunk c + 1 (or the next unit's first)
//     requested before chunk c is multiplied.  The weight planes are read in their FRAGMENT-BLOCKED copy (include/pfpp.h pfpp_pw.fhi /
//     flo, packing.PW.frag(): eval weights are static): one load instruction = 1 KB contiguous = the 64 lanes' B operands of one MFMA,
//     straight into the operand registers.  (Row-major planes needed a pass through LDS to reach that layout: registers -> swizzled
//     patch -> fragments, 3 LDS operations per 16 bytes, and the contraction ran LDS-bound at 2,700 cycles per 128-deep chunk against
//     770 of MFMA work; fragments loaded directly from row-major rows — every lane its own row, 32 lines per instruction, 16 bytes of
//     each — ran at ~9 GB/s per workgroup.)
//   * the eight rows of a wave share ONE modulation row when they belong to one batch element (always, with one puzzle in flight): 4
//     loads instead of 32 — with the x rows and the weights a workgroup pulled 256 KB through its L1 before the LayerNorm could start;
//   * the wave sums of the LayerNorm use v_permlane32_swap / v_permlane16_swap / row-rotate DPP adds: the same butterfly (32, 16, 8, 4,
//     2, 1) and therefore the same bits as layernorm_kernel's __shfl_xor chain (a rotation by r inside a row of 16 reaches a lane that
//     holds the same partial sum as lane ^ r at that stage; tools/lab/wavesum/wave_sum_check.hip), without the LDS crossbar's latency.
// LayerNorm arithmetic: exactly layernorm_kernel's (csrc/transformer_ops.hip: one wave per row) — the planes in LDS are bit-identical
// to the ones that kernel writes.  Contraction: the split-f16 products of the GEMMs (small terms first), one k-ordered chain per output.
// GEGLU form: packed weights (32 value rows | 32 gate rows interleaved), u = (v + b_v) * gelu(g + b_g) written as planes.
#include "pfpp_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int LC = 512;          // width (= contraction length)
constexpr int LKP = LC + 8;      // LDS row stride of a plane in halfs
constexpr int LNW = 4;           // waves per workgroup

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int iv = __float_as_int(v);
  return v + __int_as_float(__builtin_amdgcn_update_dpp(iv, iv, CTRL, 0xf, 0xf, false));
}

// v + shfl_xor(v, 32), then 16, 8, 4, 2, 1 — the butterfly of layernorm_kernel, lane for lane the same operands (see the header)
__device__ __forceinline__ float wave_sum(float v) {
  {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  v = dpp_add<0x128>(v);   // row_ror:8
  v = dpp_add<0x124>(v);   // row_ror:4
  v = dpp_add<0x122>(v);   // row_ror:2
  v = dpp_add<0x121>(v);   // row_ror:1
  return v;
}

struct LnLinP {
  const float* x;                 // [M, LC]
  const float* mod; int64_t ld_mod;   // AdaLN rows [B, 2 LC] (scale | shift) or null
  const float *gamma, *beta;      // affine LayerNorm or null
  const int32_t* group_batch; int group_rows;    // row -> batch map for mod
  const half8 *fh, *fl; float inv_scale;    // fragment-blocked planes of scale * W [N, LC] (include/pfpp.h pfpp_pw.fhi / flo)
  const float* bias;              // [N] or null
  float* out; int64_t ldc;        // fp32 [M, N] (plain form)
  _Float16 *uh, *ul; int64_t ldu; // GEGLU form: planes [M, N / 2]
  int M, N;
  int units_per_group;            // column units (32 outputs each) of one blockIdx.y
  float eps;
};

template <bool GEGLU>
__global__ __launch_bounds__(64 * LNW) void lnlin_small_kernel(LnLinP p) {
  extern __shared__ __align__(16) char ll_smem[];
  _Float16* sh = reinterpret_cast<_Float16*>(ll_smem);
  _Float16* sl = sh + 32 * LKP;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lhi = lane >> 5;
  const int r0 = blockIdx.x * 32;
  constexpr int NT = GEGLU ? 2 : 1;                    // weight-row tiles of 32 per unit (GEGLU: value rows | gate rows)
  constexpr int KC = GEGLU ? 64 : 128;                 // contraction depth of one streamed weight chunk (two tiles: half the depth, same registers)
  constexpr int NI = KC / 16;                          // load instructions (= MFMA steps) per tile and plane of a chunk
  constexpr int NCH = LC / KC;                         // chunks per unit
  const int n_units = p.N / (32 * NT);
  const int u_begin = blockIdx.y * p.units_per_group + wave;
  const int u_end = min(n_units, (int)(blockIdx.y + 1) * p.units_per_group);

  // ---- weight chunk (unit u, contraction [KC c, KC c + KC)) -> registers.  The planes are stored fragment-blocked: block (row tile,
  // k-step) = the 64 lanes' 16-byte B operands of one MFMA, 1 KB contiguous — a load instruction is fully coalesced and lands in
  // the operand registers as it is (no LDS pass)
  half8 st[2][NT][2][NI];                              // [buffer][tile][plane][step]
  auto fetch = [&](const int b, int u, int c) {        // (b is a constant after unrolling: the buffers stay in registers)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const size_t blk = (size_t)(NT * u + j) * (LC / 16) + NI * c + i;
        st[b][j][0][i] = p.fh[blk * 64 + lane];
        st[b][j][1][i] = p.fl[blk * 64 + lane];
      }
  };
  const bool any = u_begin < u_end;
  // the rows' modulation batch (wave-uniform scalars), requested first: the modulation loads below depend on them
  const int row_first = min(r0 + 8 * wave, p.M - 1), row_last = min(r0 + 8 * wave + 7, p.M - 1);
  int b_first = 0, b_last = 0;
  if (p.mod) { b_first = p.group_batch[row_first / p.group_rows]; b_last = p.group_batch[row_last / p.group_rows]; }
  float4 v[8][2];
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) {
    const int64_t row = min((int64_t)(r0 + 8 * wave + rr), (int64_t)p.M - 1);
    const float4* xr = reinterpret_cast<const float4*>(p.x + row * LC);
    v[rr][0] = xr[lane]; v[rr][1] = xr[lane + 64];
  }
  if (any) fetch(0, u_begin, 0);
  __builtin_amdgcn_sched_barrier(0);                   // requested before the LayerNorm starts: its latency hides behind it

  // ---- LayerNorm of rows r0 .. r0 + 31: wave w takes rows 8 w .. 8 w + 7, one row at a time across the wave (layernorm_kernel<2>'s
  // arithmetic).  All eight rows and their modulation rows are requested first: issued row by row, every row paid a memory round trip
  // of its own (8 x ~1.5 us)
  {
    // one modulation row for the wave's eight rows when they belong to one batch element (always, with one puzzle in flight);
    // otherwise row by row
    const bool one_mod = !p.mod || b_first == b_last;
    float4 m_a[8][2], m_b[8][2];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      if (rr > 0 && one_mod) {
#pragma unroll
        for (int k = 0; k < 2; ++k) { m_a[rr][k] = m_a[0][k]; m_b[rr][k] = m_b[0][k]; }
        continue;
      }
      const int64_t row = min((int64_t)(r0 + 8 * wave + rr), (int64_t)p.M - 1);
      const float* pa = p.gamma;
      const float* pb = p.beta;
      if (p.mod) {
        const int64_t b = rr == 0 ? (int64_t)b_first : (int64_t)p.group_batch[row / p.group_rows];
        pa = p.mod + b * p.ld_mod;
        pb = pa + LC;
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        m_a[rr][k] = reinterpret_cast<const float4*>(pa)[lane + 64 * k];
        m_b[rr][k] = reinterpret_cast<const float4*>(pb)[lane + 64 * k];
      }
    }
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int lrow = 8 * wave + rr;
      half4 hi[2], lo[2];
      float s = 0.0f;
#pragma unroll
      for (int k = 0; k < 2; ++k) s += (v[rr][k].x + v[rr][k].y) + (v[rr][k].z + v[rr][k].w);
      const float mean = wave_sum(s) / (float)LC;
      float q = 0.0f;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float a = v[rr][k].x - mean, b = v[rr][k].y - mean, c = v[rr][k].z - mean, d = v[rr][k].w - mean;
        q += (a * a + b * b) + (c * c + d * d);
      }
      const float var = wave_sum(q) / (float)LC;
      const float rstd = 1.0f / sqrtf(var + p.eps);
      const bool live = r0 + lrow < p.M;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        float4 o;
        o.x = (v[rr][k].x - mean) * rstd; o.y = (v[rr][k].y - mean) * rstd; o.z = (v[rr][k].z - mean) * rstd; o.w = (v[rr][k].w - mean) * rstd;
        if (p.mod) {
          o.x = o.x * (1.0f + m_a[rr][k].x) + m_b[rr][k].x; o.y = o.y * (1.0f + m_a[rr][k].y) + m_b[rr][k].y;
          o.z = o.z * (1.0f + m_a[rr][k].z) + m_b[rr][k].z; o.w = o.w * (1.0f + m_a[rr][k].w) + m_b[rr][k].w;
        } else {
          o.x = o.x * m_a[rr][k].x + m_b[rr][k].x; o.y = o.y * m_a[rr][k].y + m_b[rr][k].y;
          o.z = o.z * m_a[rr][k].z + m_b[rr][k].z; o.w = o.w * m_a[rr][k].w + m_b[rr][k].w;
        }
        if (!live) o = make_float4(0.f, 0.f, 0.f, 0.f);          // rows past the end: zero operand rows (their outputs are not stored)
        PFPP_SPLIT_TO(o.x, hi[k][0], lo[k][0]); PFPP_SPLIT_TO(o.y, hi[k][1], lo[k][1]);
        PFPP_SPLIT_TO(o.z, hi[k][2], lo[k][2]); PFPP_SPLIT_TO(o.w, hi[k][3], lo[k][3]);
        *reinterpret_cast<half4*>(sh + lrow * LKP + 4 * (lane + 64 * k)) = hi[k];
        *reinterpret_cast<half4*>(sl + lrow * LKP + 4 * (lane + 64 * k)) = lo[k];
      }
    }
  }
  __syncthreads();                                     // the only workgroup barrier: from here on every wave is on its own
  if (!any) return;

  const half8* ah = reinterpret_cast<const half8*>(sh + l31 * LKP + 8 * lhi);
  const half8* al = reinterpret_cast<const half8*>(sl + l31 * LKP + 8 * lhi);
  f32x16 acc[NT];
  // one chunk: KC / 16 steps of 16, the A fragments (shared by the unit's tiles) from the planes in LDS
  auto multiply = [&](const int b, int c) {
#pragma unroll
    for (int s8 = 0; s8 < NI; ++s8) {
      const half8 a_h = ah[(KC / 8) * c + 2 * s8], a_l = al[(KC / 8) * c + 2 * s8];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_l, st[b][j][0][s8], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h, st[b][j][1][s8], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h, st[b][j][0][s8], acc[j], 0, 0, 0);
      }
    }
  };

  for (int u = u_begin; u < u_end; u += LNW) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;
    const int un = u + LNW < u_end ? u + LNW : u;      // the next unit of this wave (the last one re-requests its own first chunk: unused)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {                    // (NCH is even: a unit starts and ends on buffer 0 / 1)
      if (c + 1 < NCH) fetch((c + 1) & 1, u, c + 1);
      else fetch(0, un, 0);
      __builtin_amdgcn_sched_barrier(0);
      multiply(c & 1, c);
    }

    // ---- epilogue straight from the accumulators: lane = column l31 of the unit, element e = row (e & 3) + 8 (e >> 2) + 4 lhi
    if constexpr (!GEGLU) {
      const int col = 32 * u + l31;
      const float bb = p.bias ? p.bias[col] : 0.0f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = r0 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
        if (row < p.M) p.out[(int64_t)row * p.ldc + col] = acc[0][e] * p.inv_scale + bb;
      }
    } else {
      // packed columns of unit u: [64 u, 64 u + 32) value, [64 u + 32, 64 u + 64) gate of output columns 32 u .. 32 u + 31
      const float bv = p.bias[64 * u + l31], bg = p.bias[64 * u + 32 + l31];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = r0 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
        const float vv = acc[0][e] * p.inv_scale + bv, gg = acc[1][e] * p.inv_scale + bg;
        const float gate = 0.5f * gg * (1.0f + erff(gg * 0.70710678118654752440f));
        _Float16 hi, lo;
        PFPP_SPLIT_TO(vv * gate, hi, lo);
        if (row < p.M) {
          const int64_t idx = (int64_t)row * p.ldu + 32 * u + l31;
          p.uh[idx] = hi;
          p.ul[idx] = lo;
        }
      }
    }
  }
}

}  // namespace

extern "C" int pfpp_layernorm_linear_small(const float* x, const float* mod, int64_t ld_mod, const float* gamma, const float* beta,
                                           const int32_t* group_batch, int64_t group_rows, const pfpp_pw* w, const float* bias, float* out,
                                           int64_t ldc, const pfpp_planes* u_planes, int64_t ldu, int64_t M, int64_t N, int64_t C, float eps,
                                           pfpp_stream_t stream) {
  PFPP_REQUIRE(x && w && (out || u_planes), "null pointer");
  PFPP_REQUIRE(w->fhi && w->flo && pfpp::aligned16(w->fhi) && pfpp::aligned16(w->flo), "the weight's fragment-blocked planes (pfpp_pw.fhi / flo) are required");
  PFPP_REQUIRE(!(mod && gamma) && (mod || (gamma && beta)) && (!mod || group_batch) && group_rows >= 1, "mod (+ group_batch) or gamma / beta");
  PFPP_SUPPORTED(C == LC, "width != 512");
  PFPP_SUPPORTED(N % 64 == 0 && N >= 64 && M >= 1 && M <= 0x7fffffff, "N % 64 != 0");
  PFPP_REQUIRE(!u_planes || (bias && u_planes->hi && u_planes->lo && ldu >= N / 2 && ldu % 4 == 0), "GEGLU form: bias and planes [M, N / 2]");
  PFPP_REQUIRE(u_planes || (ldc >= N && ldc % 4 == 0), "ldc");
  PFPP_REQUIRE(pfpp::aligned16(x) && pfpp::aligned16(mod) && pfpp::aligned16(gamma) && pfpp::aligned16(beta) && pfpp::aligned16(bias) &&
               pfpp::aligned16(out) && ld_mod % 4 == 0, "16-byte alignment");
  LnLinP p;
  p.x = x; p.mod = mod; p.ld_mod = ld_mod; p.gamma = gamma; p.beta = beta; p.group_batch = group_batch; p.group_rows = (int)group_rows;
  p.fh = (const half8*)w->fhi; p.fl = (const half8*)w->flo; p.inv_scale = 1.0f / w->scale;
  p.bias = bias; p.out = out; p.ldc = ldc;
  p.uh = u_planes ? (_Float16*)u_planes->hi : nullptr; p.ul = u_planes ? (_Float16*)u_planes->lo : nullptr; p.ldu = ldu;
  p.M = (int)M; p.N = (int)N; p.eps = eps;
  const size_t smem = (size_t)2 * 32 * LKP * sizeof(_Float16);
  // one round of workgroups: column units (32 outputs) per group = what keeps row tiles x groups within the CUs, a multiple of the
  // four waves that share them
  const int row_tiles = (int)((M + 31) / 32);
  const int n_units = (int)(N / (u_planes ? 64 : 32));
  const int max_groups = row_tiles >= 256 ? 1 : 256 / row_tiles;
  int upg = (n_units + max_groups - 1) / max_groups;
  upg = (upg + LNW - 1) / LNW * LNW;
  p.units_per_group = upg;
  const dim3 grid((unsigned)row_tiles, (unsigned)((n_units + upg - 1) / upg));
  hipStream_t st = pfpp::as_stream(stream);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)lnlin_small_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess ||
        hipFuncSetAttribute((const void*)lnlin_small_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return pfpp::check_launch(__func__);
    attr_set = true;
  }
  if (u_planes) hipLaunchKernelGGL(lnlin_small_kernel<true>, grid, dim3(64 * LNW), smem, st, p);
  else hipLaunchKernelGGL(lnlin_small_kernel<false>, grid, dim3(64 * LNW), smem, st, p);
  return pfpp::check_launch(__func__);
}
