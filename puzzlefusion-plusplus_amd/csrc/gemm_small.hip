// Plane GEMM for FEW rows (one to a few puzzles in flight: 25-2,000 tokens): out = A . W^T / scale + bias + residual.
//
// Reference: the out-projections of the two attentions and the second feed-forward linear of EncoderLayer.forward
// (denoiser/model/modules/attention.py:77-90: attn.to_out[0], ff.net[2], each followed by the residual add), eval mode, as sequenced
// by pfpp_tlayers_eval for M <= lnlin_max_rows.
//
// Why a kernel of its own: with 100-500 rows the tiled plane GEMM (gemm_pl.hip, 64 x 32 tiles, a three-stage LDS-DMA ring per
// workgroup) runs 32-128 workgroups of 16 or 64 dependent K-tiles: 10 us for K = 512, 20 us for K = 2048, bound by the latency of
// each ring stage.  Here (the contraction scheme of csrc/lnlin_small.hip):
//   * grid = (row tiles of 32, groups of four 32-column units); the A planes of the row tile are staged in LDS 512 deep at a time
//     (two buffers: the next 512 travel global -> registers -> LDS while the current ones are multiplied; one workgroup barrier per
//     512), every wave contracts ITS unit over the whole K in one accumulator chain — no partial sums;
//   * the weights come from the FRAGMENT-BLOCKED copy of their planes (include/pfpp.h pfpp_pw.fhi / flo): one load instruction = 1 KB
//     contiguous = the 64 lanes' B operands of one MFMA, no LDS pass; 128-deep chunks, chunk t + 1 requested before chunk t is multiplied;
//   * bias and residual are fetched before the contraction starts and added from registers (out may alias the residual).
// Arithmetic: the split-f16 products of pfpp_gemm (small terms first) in k order, one chain per output: equal to the tiled GEMM to fp32
// rounding (another association of the same products), deterministic.
#include <stdlib.h>

#include "pfpp_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KA = 512;          // depth of one staged A chunk
constexpr int LKP = KA + 8;      // LDS row stride of a plane in halfs
constexpr int KC = 128;          // depth of one weight chunk
constexpr int NI = KC / 16;      // MFMA steps (= load instructions per plane) of a weight chunk

struct GsP {
  const _Float16 *ah, *al; int64_t lda;      // planes of a_scale * A [M, K]
  const half8 *fh, *fl;                      // fragment-blocked planes of w_scale * W [N, K]
  float inv_scale;                           // 1 / (a_scale * w_scale)
  const float* bias;                         // [N] or null
  const float* res; int64_t ldr;             // [M, ldr] or null
  float* out; int64_t ldc;
  int M, N, K;
};

__global__ __launch_bounds__(256) void gemm_small_kernel(GsP p) {
  extern __shared__ __align__(16) char gs_smem[];
  _Float16* lds = reinterpret_cast<_Float16*>(gs_smem);          // [2 buffers][2 planes][32][LKP]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lhi = lane >> 5;
  const int r0 = blockIdx.x * 32;
  const int u = blockIdx.y * 4 + wave;                           // this wave's 32-column unit
  const bool any = u < p.N / 32;
  const int nchk = p.K / KA;
  const int nt = 4 * nchk;                                       // weight chunks of the unit

  // ---- A chunk ck: 32 rows x 512 halfs x 2 planes = 64 KB, piece q = tid + 256 i -> row q / 64, 16-byte piece q % 64 of the row
  half8 ar[2][8];
  auto load_a = [&](int ck) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = tid + 256 * i;
      const int64_t row = min(r0 + (q >> 6), p.M - 1);           // rows past the end repeat the last one (their outputs are not stored)
      const int64_t off = row * p.lda + (int64_t)KA * ck + 8 * (q & 63);
      ar[0][i] = *reinterpret_cast<const half8*>(p.ah + off);
      ar[1][i] = *reinterpret_cast<const half8*>(p.al + off);
    }
  };
  auto store_a = [&](int buf) {
    _Float16* b = lds + buf * (2 * 32 * LKP);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = tid + 256 * i;
      *reinterpret_cast<half8*>(b + (q >> 6) * LKP + 8 * (q & 63)) = ar[0][i];
      *reinterpret_cast<half8*>(b + 32 * LKP + (q >> 6) * LKP + 8 * (q & 63)) = ar[1][i];
    }
  };
  // ---- weight chunk t of the unit -> registers (blocks of the unit are contiguous over K)
  half8 st[2][2][NI];                                            // [buffer][plane][step]
  const size_t blk0 = (size_t)(any ? u : 0) * (p.K / 16);
  auto fetch = [&](const int b, int t) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const size_t blk = blk0 + (size_t)NI * t + i;
      st[b][0][i] = p.fh[blk * 64 + lane];
      st[b][1][i] = p.fl[blk * 64 + lane];
    }
  };

  load_a(0);
  if (any) fetch(0, 0);
  // bias / residual of this lane's outputs: column 32 u + l31, rows (e & 3) + 8 (e >> 2) + 4 lhi
  const int col = 32 * (any ? u : 0) + l31;
  float add[16];
  {
    const float bb = (any && p.bias) ? p.bias[col] : 0.0f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = r0 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
      add[e] = bb + ((any && p.res && row < p.M) ? p.res[(int64_t)row * p.ldr + col] : 0.0f);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  store_a(0);
  __syncthreads();

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
  for (int ck = 0; ck < nchk; ++ck) {
    const bool more = ck + 1 < nchk;
    if (more) load_a(ck + 1);                                    // in flight while this chunk is multiplied
    if (any) {
      const _Float16* b = lds + (ck & 1) * (2 * 32 * LKP);
      const half8* ah = reinterpret_cast<const half8*>(b + l31 * LKP + 8 * lhi);
      const half8* al = reinterpret_cast<const half8*>(b + 32 * LKP + l31 * LKP + 8 * lhi);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int t = 4 * ck + c;
        fetch((c + 1) & 1, t + 1 < nt ? t + 1 : t);              // (the last chunk re-requests itself: unused)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s8 = 0; s8 < NI; ++s8) {
          const half8 a_h = ah[(KC / 8) * c + 2 * s8], a_l = al[(KC / 8) * c + 2 * s8];
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_l, st[c & 1][0][s8], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h, st[c & 1][1][s8], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h, st[c & 1][0][s8], acc, 0, 0, 0);
        }
      }
    }
    if (more) {
      store_a((ck + 1) & 1);       // (that buffer was last read in chunk ck - 1: every wave is past it since the barrier that ended it)
      __syncthreads();
    }
  }
  if (!any) return;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = r0 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
    if (row < p.M) p.out[(int64_t)row * p.ldc + col] = acc[e] * p.inv_scale + add[e];
  }
}

// The same GEMM with the CONTRACTION split over the waves of a workgroup (round 6).  At 100-500 rows the kernel above runs 4-16 row tiles
// x N / 128 groups = 16-64 workgroups, and each of its waves walks the whole K in one accumulator chain: 96 (K = 512) to 384 (K = 2048)
// dependent matrix instructions behind four to sixteen rounds of weight loads, on a tenth of the chip.  Here a workgroup owns ONE 32 x 32
// output unit and NW waves (8 for K = 2048, the second feed-forward linear; 4 x 1 round would be K = 512): wave w takes k in [128 R w, 128 R (w + 1)) in R rounds of 128 (R = 1 / 2) —
// the 16 A fragments of a round (rows of the planes, 16 bytes per lane) and its 16 weight fragments (fragment-blocked planes: one load =
// the 64 lanes' operand) are requested in one go, 24 matrix instructions follow, the partial tile goes to LDS and the waves add the NW
// partials in wave order (fixed order: deterministic), scale, add bias / residual and store.  R global round trips, one barrier;
// N / 32 x M / 32 workgroups.  (16 waves of one round each for K = 2048 leave a wave 128 registers: 25 spilled.)
// Arithmetic: the same split-f16 products, small terms first, summed in 128-deep partial chains: equal to the kernel above to fp32 rounding.
template <int NW, int R>
__global__ __launch_bounds__(64 * NW) void gemm_small_ks_kernel(GsP p) {
  extern __shared__ __align__(16) char gs_smem[];
  float* red = reinterpret_cast<float*>(gs_smem);                // [NW][16][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lhi = lane >> 5;
  const int r0 = blockIdx.x * 32, u = blockIdx.y;
  const int64_t arow = min(r0 + l31, p.M - 1);                   // rows past the end repeat the last one (their outputs are not stored)
  const _Float16* ah = p.ah + arow * p.lda + 128 * R * wave + 8 * lhi;
  const _Float16* al = p.al + arow * p.lda + 128 * R * wave + 8 * lhi;
  const size_t blk0 = ((size_t)u * (p.K / 16) + 8 * R * wave) * 64 + lane;
  half8 fa[2][8], fw[2][8];
  auto fetch = [&](int r) {
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      fw[0][s8] = p.fh[blk0 + (size_t)(8 * r + s8) * 64];
      fw[1][s8] = p.fl[blk0 + (size_t)(8 * r + s8) * 64];
      fa[0][s8] = *reinterpret_cast<const half8*>(ah + 128 * r + 16 * s8);
      fa[1][s8] = *reinterpret_cast<const half8*>(al + 128 * r + 16 * s8);
    }
  };
  fetch(0);
  // this thread's outputs after the reduction: registers e = wave, wave + NW, ... of the tile (row (e & 3) + 8 (e >> 2) + 4 lhi, column l31)
  constexpr int EPW = 16 / NW;
  const int col = 32 * u + l31;
  float add[EPW];
  {
    const float bb = p.bias ? p.bias[col] : 0.0f;
#pragma unroll
    for (int j = 0; j < EPW; ++j) {
      const int e = wave + NW * j;
      const int row = r0 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
      add[j] = bb + ((p.res && row < p.M) ? p.res[(int64_t)row * p.ldr + col] : 0.0f);
    }
  }
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (r > 0) fetch(r);
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][s8], fw[0][s8], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][s8], fw[1][s8], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][s8], fw[0][s8], acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) red[(wave * 16 + e) * 64 + lane] = acc[e];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < EPW; ++j) {
    const int e = wave + NW * j;
    float v = red[e * 64 + lane];
#pragma unroll
    for (int w = 1; w < NW; ++w) v += red[(w * 16 + e) * 64 + lane];
    const int row = r0 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
    if (row < p.M) p.out[(int64_t)row * p.ldc + col] = v * p.inv_scale + add[j];
  }
}

}  // namespace

extern "C" int pfpp_gemm_small(const pfpp_planes* A, int64_t lda, const pfpp_pw* w, const float* bias, const float* residual, int64_t ldr,
                               float* out, int64_t ldc, int64_t M, int64_t N, int64_t K, pfpp_stream_t stream) {
  PFPP_REQUIRE(A && A->hi && A->lo && w && out, "null pointer");
  PFPP_REQUIRE(w->fhi && w->flo && pfpp::aligned16(w->fhi) && pfpp::aligned16(w->flo), "the weight's fragment-blocked planes (pfpp_pw.fhi / flo) are required");
  PFPP_SUPPORTED(K % KA == 0 && K >= KA && N % 32 == 0 && N >= 32, "K % 512 != 0 or N % 32 != 0");
  PFPP_REQUIRE(M >= 1 && M <= 0x7fffffff && lda >= K && lda % 8 == 0 && ldc >= N && (!residual || ldr >= N), "sizes / leading dimensions");
  PFPP_REQUIRE(pfpp::aligned16(A->hi) && pfpp::aligned16(A->lo), "16-byte aligned planes");
  GsP p;
  p.ah = (const _Float16*)A->hi; p.al = (const _Float16*)A->lo; p.lda = lda;
  p.fh = (const half8*)w->fhi; p.fl = (const half8*)w->flo;
  p.inv_scale = 1.0f / (A->scale * w->scale);
  p.bias = bias; p.res = residual; p.ldr = ldr; p.out = out; p.ldc = ldc;
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  // contraction split over the waves (gemm_small_ks_kernel) for the depths of the path, while the row tiles leave most of the chip idle;
  // PFPP_GEMM_SMALL_KS=0: the one-chain kernel everywhere (cross-check)
  const char* ks_env = getenv("PFPP_GEMM_SMALL_KS");
  // measured (tools/diag/small_time.py, dependent launches, N = 512): K = 2048 at 200 / 500 rows 20.7 -> 14.5 us, at 1,000 rows 21 -> 26
  // (every unit's workgroup re-reads the row tile's A planes); K = 512 is at the launch floor either way (7.5 - 8.2 us) and keeps the chain
  const bool ks = !(ks_env && atoi(ks_env) == 0) && K == 2048 && M <= 512;
  if (ks) {
    const dim3 gk((unsigned)((M + 31) / 32), (unsigned)(N / 32));
    hipLaunchKernelGGL((gemm_small_ks_kernel<8, 2>), gk, dim3(512), (size_t)8 * 16 * 64 * sizeof(float), pfpp::as_stream(stream), p);
    return pfpp::check_launch(__func__);
  }
  const size_t smem = (size_t)2 * 2 * 32 * LKP * sizeof(_Float16);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return pfpp::check_launch(__func__);
    attr_set = true;
  }
  const dim3 grid((unsigned)((M + 31) / 32), (unsigned)((N / 32 + 3) / 4));
  hipLaunchKernelGGL(gemm_small_kernel, grid, dim3(256), smem, pfpp::as_stream(stream), p);
  return pfpp::check_launch(__func__);
}
