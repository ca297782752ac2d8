// LayerNorm (AdaLN or affine) fused into the linear layer that follows it, for SMALL token counts (one puzzle in flight: 25-500 tokens).
//
// Reference: MyAdaLayerNorm / nn.LayerNorm followed by to_q|to_k|to_v resp. the GEGLU projection in EncoderLayer.forward
// (denoiser/model/modules/attention.py:21-25, 77-90), eval mode, as issued by pfpp_hip.denoiser.denoiser_forward_compact.
//
// Why: with one puzzle in flight the DDPM step is ~100 dependent launches of 5-17 us; 18 of them are LayerNorms of 5 us + a launch gap
// each, every one followed by a GEMM whose A operand they produce.  For <= 512 tokens the normalised rows need not exist in HBM at all
// (no backward, nothing else reads them): a workgroup normalises its 32 rows itself — every column tile's workgroup redundantly, 64 KB
// of reads — keeps them in LDS as split-f16 planes and contracts them with its 64 weight columns.  LayerNorm arithmetic: exactly
// layernorm_kernel's (csrc/transformer_ops.hip: one wave per row, same reductions) — the planes in LDS are bit-identical to the
// ones that kernel writes.  Contraction: the split-f16 products of the GEMMs (small terms first) with the contraction cut over the
// four waves of the workgroup and the partial sums added in wave order (deterministic; a different summation order than the plane
// GEMM's, so results agree with the two-launch path to fp32 rounding, not bit for bit).  The weight fragments stream from their planes
// straight into the B operand registers (as in csrc/heads.hip), the first group requested BEFORE the LayerNorm so that its latency
// hides behind it.  GEGLU form: packed weights (32 value rows | 32 gate rows interleaved), u = (v + b_v) * gelu(g + b_g) written as planes.
#include "pfpp_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int LC = 512;          // width (= contraction length)
constexpr int LKP = LC + 8;      // LDS row stride of a plane in halfs
constexpr int LNW = 4;           // waves per workgroup = contraction slices

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

struct LnLinP {
  const float* x;                 // [M, LC]
  const float* mod; int64_t ld_mod;   // AdaLN rows [B, 2 LC] (scale | shift) or null
  const float *gamma, *beta;      // affine LayerNorm or null
  const int32_t* group_batch; int group_rows;    // row -> batch map for mod
  const _Float16 *wh, *wl; int64_t ldw; float inv_scale;    // planes of scale * W [N, LC]
  const float* bias;              // [N] or null
  float* out; int64_t ldc;        // fp32 [M, N] (plain form)
  _Float16 *uh, *ul; int64_t ldu; // GEGLU form: planes [M, N / 2]
  int M, N;
  float eps;
};

template <bool GEGLU>
__global__ __launch_bounds__(64 * LNW) void lnlin_small_kernel(LnLinP p) {
  extern __shared__ __align__(16) char ll_smem[];
  _Float16* sh = reinterpret_cast<_Float16*>(ll_smem);
  _Float16* sl = sh + 32 * LKP;
  _Float16* patches = sl + 32 * LKP;                            // LNW x [2 planes][32][128] halfs = 16 KB per wave
  float* part = reinterpret_cast<float*>(patches);              // [LNW][32][64] partial sums (after the contraction: the patches are dead)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int r0 = blockIdx.x * 32, n0 = blockIdx.y * 64;

  // ---- weights of this wave's contraction slice [128 wave, 128 wave + 128) for both column tiles.  Fetched COALESCED — one instruction
  // = 4 weight rows x 256 contiguous bytes — into registers before the LayerNorm (their latency hides behind it), then passed through a
  // wave-private LDS patch ([plane][32 rows][128 halfs], 16-byte chunks XOR-swizzled by the row) from which the MFMA operand fragments
  // (lane = weight row, 8 contraction-consecutive halfs) are read.  Loading the fragments straight from global memory (every lane its own
  // row: 32 lines per instruction, 16 bytes of each) ran at ~9 GB/s per workgroup: each 16-byte piece pulled its whole line from the L2.
  const int k0 = 128 * wave;
  const int srow = lane >> 4, schunk = lane & 15;          // staging: lane -> row 4 i + srow of the tile, chunk schunk (16 bytes)
  half8 st[2][2][8];                                       // [tile][plane][instruction]
  auto fetch_tile = [&](int j) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const size_t off = (size_t)(n0 + 32 * j + 4 * i + srow) * p.ldw + k0 + 8 * schunk;
      st[j][0][i] = *reinterpret_cast<const half8*>(p.wh + off);
      st[j][1][i] = *reinterpret_cast<const half8*>(p.wl + off);
    }
  };
  fetch_tile(0);
  fetch_tile(1);
  __builtin_amdgcn_sched_barrier(0);                   // all of it is requested before the LayerNorm starts

  // ---- LayerNorm of rows r0 .. r0 + 31: wave w takes rows 8 w .. 8 w + 7, one row at a time across the wave (layernorm_kernel<2>'s
  // arithmetic).  All eight rows and their modulation rows are requested first: issued row by row, every row paid a memory round trip
  // of its own (8 x ~1.5 us — the fused launch was then no faster than the two it replaces)
  float4 v[8][2], m_a[8][2], m_b[8][2];
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) {
    const int64_t row = min((int64_t)(r0 + 8 * wave + rr), (int64_t)p.M - 1);
    const float4* xr = reinterpret_cast<const float4*>(p.x + row * LC);
    v[rr][0] = xr[lane]; v[rr][1] = xr[lane + 64];
    const float* pa = p.gamma;
    const float* pb = p.beta;
    if (p.mod) {
      const int64_t b = (int64_t)p.group_batch[row / p.group_rows];
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
  __syncthreads();

  // ---- this wave's slice of the contraction
  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;
  const half8* ah = reinterpret_cast<const half8*>(sh + l31 * LKP + k0 + 8 * lhi);
  const half8* al = reinterpret_cast<const half8*>(sl + l31 * LKP + k0 + 8 * lhi);
  half8* patch = reinterpret_cast<half8*>(patches + wave * (2 * 32 * 128));
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    // tile j: registers -> patch (chunk c of row r at position c ^ (r & 15)), then 8 steps of 16
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = 4 * i + srow;
      patch[r * 16 + (schunk ^ (r & 15))] = st[j][0][i];
      patch[512 + r * 16 + (schunk ^ (r & 15))] = st[j][1][i];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      const int q = (2 * s8 + lhi) ^ (l31 & 15);
      const half8 b_h = patch[l31 * 16 + q], b_l = patch[512 + l31 * 16 + q];
      const half8 a_h = ah[2 * s8], a_l = al[2 * s8];
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_l, b_h, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h, b_l, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h, b_h, acc[j], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();                                     // every wave is done with its patch: the partial sums take their place
  // ---- partial sums -> LDS [wave][row][col], added in wave order
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * lhi;
      part[(wave * 32 + row) * 64 + 32 * j + l31] = acc[j][e];
    }
  __syncthreads();
  if constexpr (!GEGLU) {
    // thread -> row tid >> 3, columns 8 (tid & 7) .. + 7 (two float4)
    const int row = tid >> 3, c = (tid & 7) * 8;
    if (r0 + row < p.M) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float4 s = *reinterpret_cast<const float4*>(part + row * 64 + c + 4 * h);
#pragma unroll
        for (int w = 1; w < LNW; ++w) {
          const float4 t = *reinterpret_cast<const float4*>(part + (w * 32 + row) * 64 + c + 4 * h);
          s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        const int col = n0 + c + 4 * h;
        float4 o = make_float4(s.x * p.inv_scale, s.y * p.inv_scale, s.z * p.inv_scale, s.w * p.inv_scale);
        if (p.bias) {
          const float4 b = *reinterpret_cast<const float4*>(p.bias + col);
          o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
        }
        *reinterpret_cast<float4*>(p.out + (int64_t)(r0 + row) * p.ldc + col) = o;
      }
    }
  } else {
    // packed columns: [0, 32) value, [32, 64) gate of output columns n0 / 2 .. n0 / 2 + 31; thread -> row tid >> 3, 4 outputs
    const int row = tid >> 3, c = (tid & 7) * 4;
    if (r0 + row < p.M) {
      float4 sv = *reinterpret_cast<const float4*>(part + row * 64 + c);
      float4 sg = *reinterpret_cast<const float4*>(part + row * 64 + 32 + c);
#pragma unroll
      for (int w = 1; w < LNW; ++w) {
        const float4 tv = *reinterpret_cast<const float4*>(part + (w * 32 + row) * 64 + c);
        const float4 tg = *reinterpret_cast<const float4*>(part + (w * 32 + row) * 64 + 32 + c);
        sv.x += tv.x; sv.y += tv.y; sv.z += tv.z; sv.w += tv.w;
        sg.x += tg.x; sg.y += tg.y; sg.z += tg.z; sg.w += tg.w;
      }
      const float4 bv = *reinterpret_cast<const float4*>(p.bias + n0 + c);
      const float4 bg = *reinterpret_cast<const float4*>(p.bias + n0 + 32 + c);
      const float vv[4] = {sv.x * p.inv_scale + bv.x, sv.y * p.inv_scale + bv.y, sv.z * p.inv_scale + bv.z, sv.w * p.inv_scale + bv.w};
      const float gg[4] = {sg.x * p.inv_scale + bg.x, sg.y * p.inv_scale + bg.y, sg.z * p.inv_scale + bg.z, sg.w * p.inv_scale + bg.w};
      half4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gate = 0.5f * gg[e] * (1.0f + erff(gg[e] * 0.70710678118654752440f));
        const float u = vv[e] * gate;
        PFPP_SPLIT_TO(u, hi[e], lo[e]);
      }
      const int64_t idx = (int64_t)(r0 + row) * p.ldu + (n0 >> 1) + c;
      *reinterpret_cast<half4*>(p.uh + idx) = hi;
      *reinterpret_cast<half4*>(p.ul + idx) = lo;
    }
  }
}

}  // namespace

extern "C" int pfpp_layernorm_linear_small(const float* x, const float* mod, int64_t ld_mod, const float* gamma, const float* beta,
                                           const int32_t* group_batch, int64_t group_rows, const pfpp_pw* w, const float* bias, float* out,
                                           int64_t ldc, const pfpp_planes* u_planes, int64_t ldu, int64_t M, int64_t N, int64_t C, float eps,
                                           pfpp_stream_t stream) {
  PFPP_REQUIRE(x && w && w->hi && w->lo && (out || u_planes), "null pointer");
  PFPP_REQUIRE(!(mod && gamma) && (mod || (gamma && beta)) && (!mod || group_batch) && group_rows >= 1, "mod (+ group_batch) or gamma / beta");
  PFPP_SUPPORTED(C == LC && w->ldw >= LC && w->ldw % 8 == 0, "width != 512");
  PFPP_SUPPORTED(N % 64 == 0 && N >= 64 && M >= 1 && M <= 0x7fffffff, "N % 64 != 0");
  PFPP_REQUIRE(!u_planes || (bias && u_planes->hi && u_planes->lo && ldu >= N / 2 && ldu % 4 == 0), "GEGLU form: bias and planes [M, N / 2]");
  PFPP_REQUIRE(u_planes || (ldc >= N && ldc % 4 == 0), "ldc");
  PFPP_REQUIRE(pfpp::aligned16(x) && pfpp::aligned16(mod) && pfpp::aligned16(gamma) && pfpp::aligned16(beta) && pfpp::aligned16(bias) &&
               pfpp::aligned16(out) && ld_mod % 4 == 0, "16-byte alignment");
  LnLinP p;
  p.x = x; p.mod = mod; p.ld_mod = ld_mod; p.gamma = gamma; p.beta = beta; p.group_batch = group_batch; p.group_rows = (int)group_rows;
  p.wh = (const _Float16*)w->hi; p.wl = (const _Float16*)w->lo; p.ldw = w->ldw; p.inv_scale = 1.0f / w->scale;
  p.bias = bias; p.out = out; p.ldc = ldc;
  p.uh = u_planes ? (_Float16*)u_planes->hi : nullptr; p.ul = u_planes ? (_Float16*)u_planes->lo : nullptr; p.ldu = ldu;
  p.M = (int)M; p.N = (int)N; p.eps = eps;
  const size_t smem = (size_t)2 * 32 * LKP * sizeof(_Float16) + (size_t)LNW * 2 * 32 * 128 * sizeof(_Float16);     // >= the partial sums
  const dim3 grid((unsigned)((M + 31) / 32), (unsigned)(N / 64));
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
