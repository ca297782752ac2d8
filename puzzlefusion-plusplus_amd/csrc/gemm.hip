// GEMM with fused epilogues on the CDNA4 matrix cores (gfx950):  C = epilogue(A . op(W)).
//
// Two arithmetic paths behind one interface (pfpp_gemm_args.precision):
//
//  PFPP_GEMM_F32 — v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate (bitwise a
//      k-ordered fmaf chain).  Roofline: 157.3 TFLOP/s.
//
//  PFPP_GEMM_F16X3 — "split-f16": every fp32 operand x is written as hi + lo with
//      hi = f16(x), lo = f16(x - hi) (x - hi is exact in fp32, so hi/lo carry 22 bits of x; below
//      |x| = 2^-3 lo becomes f16-subnormal: absolute error <= 3e-8), and A.W is evaluated as
//      hi.hi + hi.lo + lo.hi  into ONE fp32 accumulator with three
//      v_mfma_f32_32x32x16_f16 per 16-deep step: f16 products are exact in the fp32 accumulator,
//      the dropped lo.lo term is 2^-22 relative — fp32-grade results (measured against the 1e-4
//      parity bar of the path) at 16/3 of the fp32-MFMA rate.  W's planes are pre-split once on
//      the host (weights), A — and W when it is an activation, e.g. K in Q.K^T — is split while it
//      is staged into LDS.  Requires |x| < 65504 (f16 range); the fp32 path has no such limit.
//
// Shared structure: 256 threads = 4 waves as 2x2, each wave owns (MT*32)x(NT*32) of a
// (64*MT)x(64*NT) tile; K advances in tiles of 32 through a register-staged LDS double buffer with
// one barrier per K-tile; the steady-state loop body (prefetch of the next full tile, fragment
// reads, MFMAs, LDS writes) is one branch-free basic block so the compiler can interleave memory
// and VALU work into the gaps between matrix instructions; rows outside M/N are clamped (their
// results are discarded) instead of predicated.  LDS rows are padded (36 floats / 40 halfs) so
// that the 16-byte fragment reads are bank-conflict free.  Tile ids are remapped XCD-aware.
#include <string.h>
#include <stdlib.h>

#include "gemm_common.h"

namespace pfpp_gemm_detail {
int launch_f16x3_planes(const GemmP& p, int batch, hipStream_t st, int group_m, int variant);   // gemm_pl.hip
int launch_f16x3_planes_af32(const GemmP& p, int batch, hipStream_t st, int group_m);
}

namespace {

using namespace pfpp_gemm_detail;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

// Diagnostic builds only (tools/diag/gemm_ablate.py): remove one ingredient of the K loop to see what bounds it.
//   1 = no global loads after the first tile   2 = no MFMAs   3 = no fragment reads from LDS   4 = no LDS stores / splits
//   5 = 1 + 4 (fragment reads + MFMAs + barrier only)   6 = 5 without the barrier
#ifndef PFPP_ABLATE
#define PFPP_ABLATE 0
#endif

constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;   // fp32 path: 36 floats = 144 B rows
constexpr int LDH = BK + 8;      // split path: 40 halfs = 80 B rows (5 x 16 B: conflict-free b128 reads)

// 16-byte global load of elements [k, k+4) of one row, zero beyond K
__device__ __forceinline__ float4 load_k4(const float* row, int k, int K) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (k < K) {
    v = *reinterpret_cast<const float4*>(row + k);
    if (k + 4 > K) {  // ragged tail: the padding of the row may hold anything
      if (k + 1 >= K) v.y = 0.f;
      if (k + 2 >= K) v.z = 0.f;
      v.w = 0.f;
    }
  }
  return v;
}

// =================================================================================================
// fp32-MFMA path
// =================================================================================================
template <int MT, int NT, bool WK>
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(const GemmP p) {
  constexpr int BM = 64 * MT;
  constexpr int BN = 64 * NT;
  constexpr int A_IT = BM / 32;            // float4 loads per thread for the A tile
  constexpr int W_IT = BN / 32;            // same for W (both layouts: BK*BN/4/256)
  constexpr int WLD = WK ? BN : LDS_LD;    // LDS leading dim of the W tile
  constexpr int W_TILE = WK ? BK * BN : BN * LDS_LD;
  extern __shared__ __align__(16) float gemm_smem[];
  float* As = gemm_smem;                   // [2][BM][LDS_LD]
  float* Ws = gemm_smem + 2 * BM * LDS_LD; // [2][W_TILE]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int tile = remap_tile(blockIdx.x, gridDim.x);
  int tm, tn;
  tile_coords(p, tile, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  const int z = blockIdx.z;
  const int z0 = z / p.zdiv, z1 = z - z0 * p.zdiv;
  const float* A = p.A + z0 * p.sA0 + z1 * p.sA1;
  const float* W = p.W + z0 * p.sW0 + z1 * p.sW1;
  const int64_t c_off = z0 * p.sC0 + z1 * p.sC1;
  const int64_t v_off = z0 * p.sV0 + z1 * p.sV1;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  // ---- global -> register staging (rows clamped into range: no predication in the main loop) ----
  float4 ra[A_IT], rw[W_IT];
  const int a_row = tid >> 3, a_c4 = tid & 7;  // 8 lanes cover one 128-byte row slice
  const float* a_ptr[A_IT];
  const float* w_ptr[W_IT];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int gm = min(m0 + a_row + 32 * it, p.M - 1);
    a_ptr[it] = A + (int64_t)gm * p.lda + a_c4 * 4;
  }
#pragma unroll
  for (int it = 0; it < W_IT; ++it) {
    const int gn = min(n0 + a_row + 32 * it, p.N - 1);
    w_ptr[it] = WK ? W : W + (int64_t)gn * p.ldw + a_c4 * 4;
  }
  auto load_full = [&](int k0) {   // a K-tile that lies completely inside K ([N,K] layout only)
#pragma unroll
    for (int it = 0; it < A_IT; ++it) ra[it] = *reinterpret_cast<const float4*>(a_ptr[it] + k0);
#pragma unroll
    for (int it = 0; it < W_IT; ++it) rw[it] = *reinterpret_cast<const float4*>(w_ptr[it] + k0);
  };
  auto load_wk = [&](int k0) {     // W stored [K,N]: rows are k (predicated on K), 16-byte column groups
    constexpr int C4 = BN / 4;
    constexpr int RPI = 256 / C4;
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
      const int gk = k0 + tid / C4 + RPI * it;
      const int gn = n0 + (tid % C4) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gk < p.K && gn < p.N) {
        const float* src = W + (int64_t)gk * p.ldw + gn;
        if (gn + 4 <= p.N) {
          v = *reinterpret_cast<const float4*>(src);
        } else {
          v.x = src[0];
          if (gn + 1 < p.N) v.y = src[1];
          if (gn + 2 < p.N) v.z = src[2];
        }
      }
      rw[it] = v;
    }
  };
  auto load_tail = [&](int k0) {   // a possibly ragged K-tile
#pragma unroll
    for (int it = 0; it < A_IT; ++it) ra[it] = load_k4(a_ptr[it] - a_c4 * 4, k0 + a_c4 * 4, p.K);
    if (WK) {
      load_wk(k0);
    } else {
#pragma unroll
      for (int it = 0; it < W_IT; ++it) rw[it] = load_k4(w_ptr[it] - a_c4 * 4, k0 + a_c4 * 4, p.K);
    }
  };
  auto store_tiles = [&](int buf) {
    float* as = As + buf * BM * LDS_LD;
#pragma unroll
    for (int it = 0; it < A_IT; ++it)
      *reinterpret_cast<float4*>(as + (a_row + 32 * it) * LDS_LD + a_c4 * 4) = ra[it];
    float* ws = Ws + buf * W_TILE;
    if (!WK) {
#pragma unroll
      for (int it = 0; it < W_IT; ++it)
        *reinterpret_cast<float4*>(ws + (a_row + 32 * it) * LDS_LD + a_c4 * 4) = rw[it];
    } else {
      constexpr int C4 = BN / 4;
      constexpr int RPI = 256 / C4;
#pragma unroll
      for (int it = 0; it < W_IT; ++it)
        *reinterpret_cast<float4*>(ws + (tid / C4 + RPI * it) * WLD + (tid % C4) * 4) = rw[it];
    }
  };

  const int l31 = lane & 31, lhi = lane >> 5;
  auto chunk = [&](const float* as, const float* ws, int kc) {
    float4 a[MT], b[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
      a[i] = *reinterpret_cast<const float4*>(as + i * 32 * LDS_LD + kc * 8);
    if (!WK) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
        b[j] = *reinterpret_cast<const float4*>(ws + (wn * 32 * NT + j * 32 + l31) * LDS_LD + lhi * 4 + kc * 8);
    } else {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float* bp = ws + (kc * 8 + lhi * 4) * WLD + wn * 32 * NT + j * 32 + l31;
        b[j].x = bp[0];
        b[j].y = bp[WLD];
        b[j].z = bp[2 * WLD];
        b[j].w = bp[3 * WLD];
      }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
      }
  };
  auto frag_base_a = [&](int buf) { return As + buf * BM * LDS_LD + (wm * 32 * MT + l31) * LDS_LD + lhi * 4; };

  const int nk_full = WK ? 0 : p.K / BK;               // the [K,N] layout always takes the predicated path
  const int nk = (p.K + BK - 1) / BK;

  if (nk_full > 0) load_full(0); else load_tail(0);
  store_tiles(0);
  __syncthreads();

  int kt = 0;
  // steady state: full tile kt in LDS, full tile kt+1 being fetched — one basic block
  for (; kt + 1 < nk_full; ++kt) {
    const int buf = kt & 1;
    load_full((kt + 1) * BK);
    const float* as = frag_base_a(buf);
    const float* ws = Ws + buf * W_TILE;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) chunk(as, ws, kc);
    store_tiles(buf ^ 1);
    __syncthreads();
  }
  // remaining tiles: the last full one (prefetching a ragged tail if there is one), then the tail
  for (; kt < nk; ++kt) {
    const int buf = kt & 1;
    const bool has_next = kt + 1 < nk;
    if (has_next) load_tail((kt + 1) * BK);
    const float* as = frag_base_a(buf);
    const float* ws = Ws + buf * W_TILE;
    const int krem = p.K - kt * BK;
    const int nchunk = krem >= BK ? 4 : (krem + 7) >> 3;
    for (int kc = 0; kc < nchunk; ++kc) chunk(as, ws, kc);
    if (has_next) store_tiles(buf ^ 1);
    __syncthreads();
  }

  epilogue<MT, NT>(p, acc, m0 + wm * 32 * MT, n0 + wn * 32 * NT, n0, wn, lane, c_off, v_off);
}

// =================================================================================================
// split-f16 x3 path
// =================================================================================================
// hi = f16(x), lo = f16(x - hi), two elements at a time so that every step is one packed instruction
// (v_cvt_pk_f16_f32, 2 x v_cvt_f32_f16, v_pk_add_f32, v_cvt_pk_f16_f32: 2.5 VALU per element instead of 4.3 — the
// element-wise form made the compiler convert every value twice)
typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2(const float2v x, half2v& hi, half2v& lo) {
  hi = __builtin_convertvector(x, half2v);
  const float2v back = __builtin_convertvector(hi, float2v);
  lo = __builtin_convertvector(x - back, half2v);
}
__device__ __forceinline__ void split4(const float4 v, half4& hi, half4& lo) {
  half2v h0, l0, h1, l1;
  split2(float2v{v.x, v.y}, h0, l0);
  split2(float2v{v.z, v.w}, h1, l1);
  hi = half4{h0[0], h0[1], h1[0], h1[1]};
  lo = half4{l0[0], l0[1], l1[0], l1[1]};
}

// AFF (PF2 variants only): the A operand always carries the fused BatchNorm+ReLU affine — a compile-time property there,
// because a data-dependent branch in the K loop splits the scheduling region the two-deep prefetch relies on
// APRE: the A operand arrives as pre-split fp16 planes (p.Ahi / p.Alo, written by the producing kernel's epilogue): it is
// staged exactly like a pre-split W — 16-byte loads, 16-byte LDS stores, no conversion instructions in the K loop
// PFD > 2: `PFD` K-tiles in flight in registers (latency-bound launches: a handful of workgroups, each walking its K range
// alone — the sampler step of a single puzzle, 250 rows): generalisation of the two-deep loop, PFD tiles per branch-free trip
template <int MT, int NT, bool WPRE, int WM, int WN, bool PF2, bool AFF, bool APRE, int PFD = 0>
__device__ __forceinline__ void gemm_f16x3_body(const GemmP& p) {
  constexpr int NTHR = 64 * WM * WN;
  constexpr int BM = 32 * MT * WM;
  constexpr int BN = 32 * NT * WN;
  constexpr int A_IT = BM / (NTHR / 8);    // float4 loads per thread (fp32 source, 8 lanes per row)
  constexpr int WF_IT = BN / (NTHR / 8);   // same for an fp32 W
  constexpr int WH_IT = BN / (NTHR / 4);   // 16-byte (8-half) loads per thread and plane for a pre-split W
  constexpr int PLANE_A = BM * LDH;        // halfs per plane
  constexpr int PLANE_W = BN * LDH;
  constexpr int STAGE = 2 * PLANE_A + 2 * PLANE_W;
  extern __shared__ __align__(16) _Float16 gemm_smem_h[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lhi = lane >> 5;

  const int tile = remap_tile(blockIdx.x, gridDim.x);
  int tm, tn;
  tile_coords(p, tile, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  const int z = blockIdx.z;
  const int z0 = z / p.zdiv, z1 = z - z0 * p.zdiv;
  const float* A = p.A + z0 * p.sA0 + z1 * p.sA1;
  const int64_t w_off = z0 * p.sW0 + z1 * p.sW1;
  const int64_t c_off = z0 * p.sC0 + z1 * p.sC1;
  const int64_t v_off = z0 * p.sV0 + z1 * p.sV1;

  f32x16 accM[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) accM[i][j][e] = 0.0f;

  // ---- staging registers and (clamped) source rows ---------------------------------------------
  constexpr int NWF = WPRE ? 1 : WF_IT;
  constexpr int NWH = WPRE ? WH_IT : 1;
  constexpr int AH_IT = BM / (NTHR / 4);   // 16-byte (8-half) loads per thread and plane for a pre-split A
  constexpr int NAF = APRE ? 1 : A_IT;
  constexpr int NAH = APRE ? AH_IT : 1;
  // one set of staging registers per K-tile in flight (PF2: two sets = prefetch distance 2, for grids of one or two
  // workgroups per CU where nothing else hides the load latency)
  struct Stage {
    float4 ra[NAF];
    uint4 rah[NAH], ral[NAH];
    float4 rwf[NWF];
    uint4 rwh[NWH], rwl[NWH];
    float4 r_mul, r_add;      // fused BN+ReLU on A
  };
  Stage s0, s1;              // s1 is dead (and eliminated) unless PF2
  s0.r_mul = make_float4(1.f, 1.f, 1.f, 1.f);
  s0.r_add = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (PF2) { s1.r_mul = s0.r_mul; s1.r_add = s0.r_add; }
  // the fused-BatchNorm operand is a run-time property only of the small one-deep tiles (set abstraction 1 in train mode);
  // everywhere else it is a template parameter (AFF) or absent — a data-dependent branch in the K loop costs the whole
  // loop its scheduling freedom (the 256x256 kernel's steady state had four of them)
  constexpr bool RT_AFF = !PF2 && !APRE && PFD == 0 && (MT * NT <= 4);
  const bool a_aff = RT_AFF && p.a_mul != nullptr;
  // staging rows: bits 0 and 2 of the row index are swapped, so the two rows a 16-lane (8-byte stores) or 8-lane
  // (16-byte stores) LDS store group touches are 4 apart — with 80-byte rows their bank ranges are then disjoint
  // (rows r, r+1 overlap in 4 of 32 banks: SQ_LDS_BANK_CONFLICT was 30 % of the LDS cycles)
  auto swap02 = [](int r) { return (r & ~5) | ((r & 1) << 2) | ((r >> 2) & 1); };
  const int a_row = swap02(tid >> 3), a_c4 = tid & 7;
  const int h_row = swap02(tid >> 2), h_c8 = tid & 3;    // pre-split W: 4 lanes x 8 halfs cover a 32-half row slice
  const float* a_ptr[NAF];
  const _Float16* ah_ptr[NAH];
  const _Float16* al_ptr[NAH];
  const float* wf_ptr[NWF];
  const _Float16* wh_ptr[NWH];
  const _Float16* wl_ptr[NWH];
  if constexpr (APRE) {
    const int64_t a_off = z0 * p.sA0 + z1 * p.sA1;
#pragma unroll
    for (int it = 0; it < AH_IT; ++it) {
      const int gm = min(m0 + h_row + (NTHR / 4) * it, p.M - 1);
      ah_ptr[it] = reinterpret_cast<const _Float16*>(p.Ahi) + a_off + (int64_t)gm * p.lda + h_c8 * 8;
      al_ptr[it] = reinterpret_cast<const _Float16*>(p.Alo) + a_off + (int64_t)gm * p.lda + h_c8 * 8;
    }
  } else
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int gm = min(m0 + a_row + (NTHR / 8) * it, p.M - 1);
    if (p.g_idx) {     // fused grouping: the row lives in the level's feature table at its ball-query index
      const int id = min(p.g_idx[gm], p.g_N - 1);
      const int f = gm / (p.g_S * p.g_ns);
      a_ptr[it] = A + ((int64_t)f * p.g_N + id) * p.lda + a_c4 * 4;
    } else {
      a_ptr[it] = A + (int64_t)gm * p.lda + a_c4 * 4;
    }
  }
  if constexpr (WPRE) {
#pragma unroll
    for (int it = 0; it < WH_IT; ++it) {
      const int gn = min(n0 + h_row + (NTHR / 4) * it, p.N - 1);
      wh_ptr[it] = reinterpret_cast<const _Float16*>(p.Whi) + w_off + (int64_t)gn * p.ldw + h_c8 * 8;
      wl_ptr[it] = reinterpret_cast<const _Float16*>(p.Wlo) + w_off + (int64_t)gn * p.ldw + h_c8 * 8;
    }
  } else {
#pragma unroll
    for (int it = 0; it < WF_IT; ++it) {
      const int gn = min(n0 + a_row + (NTHR / 8) * it, p.N - 1);
      wf_ptr[it] = p.W + w_off + (int64_t)gn * p.ldw + a_c4 * 4;
    }
  }
  auto load_full = [&](Stage& s, int k0) {
#if PFPP_ABLATE == 1 || PFPP_ABLATE == 5 || PFPP_ABLATE == 6
    if (k0 != 0) return;
#endif
    if constexpr (APRE) {
#pragma unroll
      for (int it = 0; it < AH_IT; ++it) {
        s.rah[it] = *reinterpret_cast<const uint4*>(ah_ptr[it] + k0);
        s.ral[it] = *reinterpret_cast<const uint4*>(al_ptr[it] + k0);
      }
    } else {
#pragma unroll
      for (int it = 0; it < A_IT; ++it) s.ra[it] = *reinterpret_cast<const float4*>(a_ptr[it] + k0);
    }
    if (!APRE && (AFF || (RT_AFF && a_aff))) {
      s.r_mul = *reinterpret_cast<const float4*>(p.a_mul + k0 + a_c4 * 4);
      s.r_add = *reinterpret_cast<const float4*>(p.a_add + k0 + a_c4 * 4);
    }
    if constexpr (WPRE) {
#pragma unroll
      for (int it = 0; it < WH_IT; ++it) {
        s.rwh[it] = *reinterpret_cast<const uint4*>(wh_ptr[it] + k0);
        s.rwl[it] = *reinterpret_cast<const uint4*>(wl_ptr[it] + k0);
      }
    } else {
#pragma unroll
      for (int it = 0; it < WF_IT; ++it) s.rwf[it] = *reinterpret_cast<const float4*>(wf_ptr[it] + k0);
    }
  };
  auto load_tail = [&](Stage& s, int k0) {
    if constexpr (APRE) {        // pre-split A needs K % 32 == 0 (validated on the host): there is no ragged tile
#pragma unroll
      for (int it = 0; it < AH_IT; ++it) { s.rah[it] = make_uint4(0, 0, 0, 0); s.ral[it] = make_uint4(0, 0, 0, 0); }
    } else if (p.g_idx) {
      // fused grouping: the last 4 columns are the neighbour's offset from its centroid (+ a zero); recomputed from
      // the row index here so that nothing extra stays live across the K loop
#pragma unroll
      for (int it = 0; it < A_IT; ++it) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a_c4 == 0) {
          const int gm = min(m0 + a_row + (NTHR / 8) * it, p.M - 1);
          const int id = min(p.g_idx[gm], p.g_N - 1);
          const int fs = gm / p.g_ns;
          const int f = fs / p.g_S;
          const float* q = p.g_xyz + ((int64_t)f * p.g_N + id) * 3;
          const float* c = p.g_ctr + (int64_t)fs * 3;
          v.x = __fsub_rn(q[0], c[0]); v.y = __fsub_rn(q[1], c[1]); v.z = __fsub_rn(q[2], c[2]);
        }
        s.ra[it] = v;
      }
    } else {
#pragma unroll
      for (int it = 0; it < A_IT; ++it) s.ra[it] = load_k4(a_ptr[it] - a_c4 * 4, k0 + a_c4 * 4, p.K);
    }
    if constexpr (WPRE) {
      // the planes are zero-padded to a multiple of 8 halfs per row (host packing): whole
      // 16-byte groups are either inside the padded row or skipped
      const bool ok = k0 + h_c8 * 8 < (int)p.ldw;
#pragma unroll
      for (int it = 0; it < WH_IT; ++it) {
        s.rwh[it] = ok ? *reinterpret_cast<const uint4*>(wh_ptr[it] + k0) : make_uint4(0, 0, 0, 0);
        s.rwl[it] = ok ? *reinterpret_cast<const uint4*>(wl_ptr[it] + k0) : make_uint4(0, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int it = 0; it < WF_IT; ++it) s.rwf[it] = load_k4(wf_ptr[it] - a_c4 * 4, k0 + a_c4 * 4, p.K);
    }
  };
  auto store_tiles = [&](const Stage& s, int buf, int part = 2) {      // part 0: A planes, 1: W planes, 2: both
#if PFPP_ABLATE == 4 || PFPP_ABLATE == 5 || PFPP_ABLATE == 6
    if (buf >= 0) return;
#endif
    _Float16* st = gemm_smem_h + buf * STAGE;
    _Float16* ahi = st, *alo = st + PLANE_A, *whi = st + 2 * PLANE_A, *wlo = st + 2 * PLANE_A + PLANE_W;
    if constexpr (APRE) {
      if (part != 1)
#pragma unroll
        for (int it = 0; it < AH_IT; ++it) {
          const int off = (h_row + (NTHR / 4) * it) * LDH + h_c8 * 8;
          *reinterpret_cast<uint4*>(ahi + off) = s.rah[it];
          *reinterpret_cast<uint4*>(alo + off) = s.ral[it];
        }
    } else if (part != 1)
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      half4 hi, lo;
      float4 v = s.ra[it];
      if (AFF || (RT_AFF && a_aff)) {   // relu(batch-norm(y)) of the previous layer, applied while the tile is staged
        v.x = fmaxf(v.x * s.r_mul.x + s.r_add.x, 0.0f); v.y = fmaxf(v.y * s.r_mul.y + s.r_add.y, 0.0f);
        v.z = fmaxf(v.z * s.r_mul.z + s.r_add.z, 0.0f); v.w = fmaxf(v.w * s.r_mul.w + s.r_add.w, 0.0f);
      }
      split4(v, hi, lo);
      const int off = (a_row + (NTHR / 8) * it) * LDH + a_c4 * 4;
      *reinterpret_cast<half4*>(ahi + off) = hi;
      *reinterpret_cast<half4*>(alo + off) = lo;
    }
    if (part == 0) return;
    if constexpr (WPRE) {
#pragma unroll
      for (int it = 0; it < WH_IT; ++it) {
        const int off = (h_row + (NTHR / 4) * it) * LDH + h_c8 * 8;
        *reinterpret_cast<uint4*>(whi + off) = s.rwh[it];
        *reinterpret_cast<uint4*>(wlo + off) = s.rwl[it];
      }
    } else {
#pragma unroll
      for (int it = 0; it < WF_IT; ++it) {
        half4 hi, lo;
        split4(s.rwf[it], hi, lo);
        const int off = (a_row + (NTHR / 8) * it) * LDH + a_c4 * 4;
        *reinterpret_cast<half4*>(whi + off) = hi;
        *reinterpret_cast<half4*>(wlo + off) = lo;
      }
    }
  };
  constexpr int IB = MT >= 2 ? 2 : 1;      // A fragments are read IB M-tiles at a time
  auto compute = [&](int buf, int ks_begin = 0, int ks_end = BK / 16) {
    const _Float16* st = gemm_smem_h + buf * STAGE;
    const _Float16* a_base = st + (wm * 32 * MT + l31) * LDH + lhi * 8;
    const _Float16* w_base = st + 2 * PLANE_A + (wn * 32 * NT + l31) * LDH + lhi * 8;
#ifndef PFPP_HOIST
#define PFPP_HOIST 0     // experiment (tools/diag/gemm_ablate.py): isolated GEMMs +0..10 % at 3850 rows, -4 % at 16000x1536x512;
                         // whole-step timings unchanged within noise, so the simpler loop stays
#endif
    if constexpr (PFPP_HOIST && MT == 2) {
      // all fragment reads of the K-tile up front: the second 16-deep step's reads land behind the first step's MFMAs
      half8 fa_h[2][MT], fa_l[2][MT], fb_h[2][NT], fb_l[2][NT];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          fa_h[ks][i] = *reinterpret_cast<const half8*>(a_base + i * 32 * LDH + ks * 16);
          fa_l[ks][i] = *reinterpret_cast<const half8*>(a_base + PLANE_A + i * 32 * LDH + ks * 16);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          fb_h[ks][j] = *reinterpret_cast<const half8*>(w_base + j * 32 * LDH + ks * 16);
          fb_l[ks][j] = *reinterpret_cast<const half8*>(w_base + PLANE_W + j * 32 * LDH + ks * 16);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) accM[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_l[ks][i], fb_h[ks][j], accM[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) accM[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_h[ks][i], fb_l[ks][j], accM[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) accM[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_h[ks][i], fb_h[ks][j], accM[i][j], 0, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      if (ks < ks_begin || ks >= ks_end) continue;
      half8 bh[NT], bl[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
#if PFPP_ABLATE == 3
        bh[j] = bl[j] = half8{(_Float16)(lane + j), 1, 2, 3, 4, 5, 6, 7};
#else
        bh[j] = *reinterpret_cast<const half8*>(w_base + j * 32 * LDH + ks * 16);
        bl[j] = *reinterpret_cast<const half8*>(w_base + PLANE_W + j * 32 * LDH + ks * 16);
#endif
      }
      // A fragments two M-tiles at a time (keeps the 128x64 wave tile of the 256x256 variant in registers)
#pragma unroll
      for (int i0 = 0; i0 < MT; i0 += IB) {
        half8 ah[IB], al[IB];
#pragma unroll
        for (int ii = 0; ii < IB; ++ii) {
#if PFPP_ABLATE == 3
          ah[ii] = al[ii] = half8{(_Float16)(lane + ii + ks), 1, 2, 3, 4, 5, 6, 7};
#else
          ah[ii] = *reinterpret_cast<const half8*>(a_base + (i0 + ii) * 32 * LDH + ks * 16);
          al[ii] = *reinterpret_cast<const half8*>(a_base + PLANE_A + (i0 + ii) * 32 * LDH + ks * 16);
#endif
        }
        // term-major order: the three MFMAs that feed one accumulator are 2*NT instructions apart, never back to back
        // (a dependent MFMA on the same accumulator waits for the previous one's passes; small terms first)
#if PFPP_ABLATE == 2
#pragma unroll
        for (int ii = 0; ii < IB; ++ii)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            accM[i0 + ii][j][0] += (float)al[ii][0] * (float)bh[j][0] + (float)ah[ii][1] * (float)bl[j][1];
#else
#pragma unroll
        for (int ii = 0; ii < IB; ++ii)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            accM[i0 + ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ii], bh[j], accM[i0 + ii][j], 0, 0, 0);
#pragma unroll
        for (int ii = 0; ii < IB; ++ii)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            accM[i0 + ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ii], bl[j], accM[i0 + ii][j], 0, 0, 0);
#pragma unroll
        for (int ii = 0; ii < IB; ++ii)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            accM[i0 + ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ii], bh[j], accM[i0 + ii][j], 0, 0, 0);
#endif
      }
    }
  };

  // K range of this workgroup (the whole K unless split_k > 1; chunks are multiples of BK, so only the last chunk can
  // have a ragged tile)
  const int kb = p.split_k > 1 ? blockIdx.y * p.k_chunk : 0;
  const int ke = p.split_k > 1 ? min(p.K, kb + p.k_chunk) : p.K;
  const int nk_full = (ke - kb) / BK;
  const int nk = (ke - kb + BK - 1) / BK;
  if constexpr (PFD > 2) {
    Stage sd[PFD];
    auto load_any = [&](Stage& s, int kt_) {
      if (kt_ < nk_full) load_full(s, kb + kt_ * BK); else load_tail(s, kb + kt_ * BK);
    };
    // invariant at the top of a trip (kt a multiple of PFD): register set d holds tile kt + d, LDS stage 0 holds tile kt
#pragma unroll
    for (int d = 0; d < PFD; ++d)
      if (d < nk) load_any(sd[d], d);
    store_tiles(sd[0], 0);
    __syncthreads();
    int kt = 0;
    for (; kt + 2 * PFD <= nk_full; kt += PFD) {
#pragma unroll
      for (int j = 0; j < PFD; ++j) {
        load_full(sd[j], kb + (kt + j + PFD) * BK);           // set j is free: its tile sits in LDS stage j & 1
        compute(j & 1, 0, 1);
        store_tiles(sd[(j + 1) % PFD], (j + 1) & 1, 0);
        compute(j & 1, 1, 2);
        store_tiles(sd[(j + 1) % PFD], (j + 1) & 1, 1);
        __syncthreads();
      }
    }
    for (; kt < nk; kt += PFD) {
#pragma unroll
      for (int j = 0; j < PFD; ++j) {
        if (kt + j >= nk) break;
        if (kt + j + PFD < nk) load_any(sd[j], kt + j + PFD);
        compute(j & 1, 0, 1);
        if (kt + j + 1 < nk) store_tiles(sd[(j + 1) % PFD], (j + 1) & 1, 0);
        compute(j & 1, 1, 2);
        if (kt + j + 1 < nk) store_tiles(sd[(j + 1) % PFD], (j + 1) & 1, 1);
        __syncthreads();
      }
    }
  } else if constexpr (!PF2) {
    if (nk_full > 0) load_full(s0, kb); else load_tail(s0, kb);
    store_tiles(s0, 0);
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nk_full; ++kt) {        // steady state: one basic block
      load_full(s0, kb + (kt + 1) * BK);
      compute(kt & 1);
      store_tiles(s0, (kt & 1) ^ 1);
#if PFPP_ABLATE != 6
      __syncthreads();
#endif
    }
    for (; kt < nk; ++kt) {
      const bool has_next = kt + 1 < nk;
      if (has_next) load_tail(s0, kb + (kt + 1) * BK);
      compute(kt & 1);                      // a ragged tile was zero-filled: the extra products are zeros
      if (has_next) store_tiles(s0, (kt & 1) ^ 1);
      __syncthreads();
    }
  } else {
    auto load_any = [&](Stage& s, int kt_) {
      if (kt_ < nk_full) load_full(s, kb + kt_ * BK); else load_tail(s, kb + kt_ * BK);
    };
    // invariant at the top of iteration kt: LDS stage kt&1 holds tile kt, set (kt+1)&1 holds tile kt+1 (in flight)
    load_any(s0, 0);
    if (nk > 1) load_any(s1, 1);
    store_tiles(s0, 0);
    __syncthreads();
    // The tile for kt+1 already sits in registers (loaded one iteration ago), so its split + LDS stores do not wait
    // on memory: they are placed BETWEEN the two 16-deep MFMA steps of tile kt to issue under the matrix pipe.
    // Steady state is branch-free (two K-tiles per trip, everything unconditional) so that it is ONE scheduling
    // region; the last tiles run through the guarded tail.
    int kt = 0;
    for (; kt + 3 < nk_full; kt += 2) {
      load_full(s0, kb + (kt + 2) * BK);
      compute(0, 0, 1);
      store_tiles(s1, 1, 0);
      compute(0, 1, 2);
      store_tiles(s1, 1, 1);
      __syncthreads();
      load_full(s1, kb + (kt + 3) * BK);
      compute(1, 0, 1);
      store_tiles(s0, 0, 0);
      compute(1, 1, 2);
      store_tiles(s0, 0, 1);
      __syncthreads();
    }
    for (; kt < nk; kt += 2) {
      if (kt + 2 < nk) load_any(s0, kt + 2);
      compute(0, 0, 1);
      if (kt + 1 < nk) store_tiles(s1, 1, 0);
      compute(0, 1, 2);
      if (kt + 1 < nk) store_tiles(s1, 1, 1);
      __syncthreads();
      if (kt + 1 >= nk) break;
      if (kt + 3 < nk) load_any(s1, kt + 3);
      compute(1, 0, 1);
      if (kt + 2 < nk) store_tiles(s0, 0, 0);
      compute(1, 1, 2);
      if (kt + 2 < nk) store_tiles(s0, 0, 1);
      __syncthreads();
    }
  }

  if (p.split_k > 1) {
    // park the partial tile (thread-contiguous layout: coalesced), take a ticket; the last arrival sums all chunks
    constexpr int ELEMS = MT * NT * 16;
    const int tile_lin = blockIdx.z * gridDim.x + blockIdx.x;
    const size_t n_tiles = (size_t)gridDim.x * gridDim.z;
    float* mine = p.split_ws + ((size_t)blockIdx.y * n_tiles + tile_lin) * ELEMS * NTHR + tid;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) mine[(size_t)((i * NT + j) * 16 + e) * NTHR] = accM[i][j][e];
    __threadfence();
    __syncthreads();
    int* s_ticket = reinterpret_cast<int*>(gemm_smem_h);     // the tile buffers are free now (no static LDS: the 256x256
                                                             // tile already uses all 160 KB)
    if (tid == 0) *s_ticket = atomicAdd(p.split_cnt + tile_lin, 1);
    __syncthreads();
    if (*s_ticket != p.split_k - 1) return;
    __threadfence();
    if (tid == 0) p.split_cnt[tile_lin] = 0;          // ready for the next launch
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) accM[i][j][e] = 0.0f;
    for (int sidx = 0; sidx < p.split_k; ++sidx) {
      const float* part = p.split_ws + ((size_t)sidx * n_tiles + tile_lin) * ELEMS * NTHR + tid;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) accM[i][j][e] += part[(size_t)((i * NT + j) * 16 + e) * NTHR];
    }
  }
  epilogue<MT, NT>(p, accM, m0 + wm * 32 * MT, n0 + wn * 32 * NT, n0, wn, lane, c_off, v_off);
}

template <int MT, int NT, bool WPRE, int WM = 2, int WN = 2, bool PF2 = false, bool AFF = false>
__global__ __launch_bounds__(64 * WM * WN, (MT * NT >= 16 ? 1 : 2)) void gemm_f16x3_kernel(const GemmP p) {
  gemm_f16x3_body<MT, NT, WPRE, WM, WN, PF2, AFF, false>(p);
}

// both operands pre-split (activations produced as fp16 planes by the previous kernel's epilogue, see ops.SplitAct)
template <int MT, int NT, int WM, int WN, bool PF2>
__global__ __launch_bounds__(64 * WM * WN, (MT * NT >= 16 ? 1 : 2)) void gemm_f16x3_apre_kernel(const GemmP p) {
  gemm_f16x3_body<MT, NT, true, WM, WN, PF2, false, true>(p);
}

// latency-bound launches: PFD K-tiles in flight per workgroup (W pre-split, fp32 A, no fused-BatchNorm operands)
template <int MT, int NT, int WM, int WN, int PFD>
__global__ __launch_bounds__(64 * WM * WN, 1) void gemm_f16x3_deep_kernel(const GemmP p) {
  gemm_f16x3_body<MT, NT, true, WM, WN, true, false, false, PFD>(p);
}

int gemm_group_m() {
  static const int v = getenv("PFPP_GEMM_GROUP_M") ? atoi(getenv("PFPP_GEMM_GROUP_M")) : 8;
  return v;
}

// =================================================================================================
// capacity of the caller's split-K workspace for the launch being dispatched (set by pfpp_gemm)
thread_local int64_t p_split_ws_bytes = 0;
thread_local int64_t p_split_cnt_len = 0;

template <typename K>
int launch(K kern, size_t smem, GemmP p, int BM, int BN, int batch, hipStream_t st, bool* attr_set, int nthr = 256,
           bool can_split = false, int min_chunk = 128, bool two_stage_layout = false) {
  if (!*attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    *attr_set = true;
  }
  const int env_group = gemm_group_m();
  static const int env_pad = getenv("PFPP_GEMM_LDS_PAD") ? atoi(getenv("PFPP_GEMM_LDS_PAD")) : 0;   // experiment knob
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  p.group_m = p.tiles_n > 1 ? env_group : 0;
  // skinny launches (fewer tiles than CUs, K >= 1024): split K over blockIdx.y when the caller lent a workspace.  Measured on
  // M = 125: 50.6 -> 25.4 us at K = 2048; at K = 512 the fix-up costs more than the four K-tiles it saves, hence the bound.
  p.split_k = 1;
  p.k_chunk = 0;
  static const bool split_on = !(getenv("PFPP_GEMM_SPLITK") && atoi(getenv("PFPP_GEMM_SPLITK")) == 0);
  const int64_t tiles = (int64_t)p.tiles_m * p.tiles_n * batch;
  if (can_split && split_on && p.split_ws && p.split_cnt && tiles < 192 && p.K >= 1024 && p.pool == 0 && !p.stats && !p.g_idx) {
    int want = (int)((384 + tiles - 1) / tiles);
    const int max_by_k = p.K / min_chunk;                    // at least min_chunk / 32 K-tiles per chunk (default 4)
    int splits = want < max_by_k ? want : max_by_k;
    if (splits > 16) splits = 16;
    const int64_t need = (int64_t)splits * tiles * BM * BN * (int64_t)sizeof(float);
    if (splits > 1 && need <= p_split_ws_bytes && tiles <= p_split_cnt_len) {
      int chunk = (p.K + splits - 1) / splits;
      chunk = (chunk + 31) / 32 * 32;
      p.split_k = (p.K + chunk - 1) / chunk;
      p.k_chunk = chunk;
    }
  }
  const dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)p.split_k, (unsigned)batch);
  // a contraction of one K-tile never touches the second LDS stage: allocate one, more workgroups fit a CU (the
  // [1.26 M x 4] first layer of the encoder: 128x64 tiles at 30 KB -> 3 per CU instead of 2, register-limited)
  static const bool one_stage = !(getenv("PFPP_GEMM_1STAGE") && atoi(getenv("PFPP_GEMM_1STAGE")) == 0);
  if (two_stage_layout && one_stage && p.K <= 32 && p.split_k == 1) smem /= 2;
  hipLaunchKernelGGL(kern, grid, dim3(nthr), smem + env_pad, st, p);
  return pfpp::check_launch("pfpp_gemm");
}

template <int MT, int NT, bool WK>
int launch_f32(const GemmP& p, int batch, hipStream_t st) {
  constexpr int BM = 64 * MT, BN = 64 * NT;
  constexpr size_t smem = (size_t)(2 * BM * LDS_LD + 2 * (WK ? BK * BN : BN * LDS_LD)) * sizeof(float);
  static bool attr_set = false;
  return launch(gemm_f32_mfma_kernel<MT, NT, WK>, smem, p, BM, BN, batch, st, &attr_set);
}

template <int MT, int NT, bool WPRE, int WM = 2, int WN = 2, bool PF2 = false, bool AFF = false>
int launch_f16x3(const GemmP& p, int batch, hipStream_t st) {
  constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
  constexpr size_t smem = (size_t)2 * (2 * BM * LDH + 2 * BN * LDH) * sizeof(_Float16);
  static bool attr_set = false;
  return launch(gemm_f16x3_kernel<MT, NT, WPRE, WM, WN, PF2, AFF>, smem, p, BM, BN, batch, st, &attr_set, 64 * WM * WN, true, 128, true);
}

template <int MT, int NT, int WM, int WN, int PFD>
int launch_f16x3_deep(const GemmP& p, int batch, hipStream_t st) {
  constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
  constexpr size_t smem = (size_t)2 * (2 * BM * LDH + 2 * BN * LDH) * sizeof(_Float16);
  static bool attr_set = false;
  // split K only into chunks that still fill the prefetch pipeline twice over (the fix-up reads one partial per chunk)
  return launch(gemm_f16x3_deep_kernel<MT, NT, WM, WN, PFD>, smem, p, BM, BN, batch, st, &attr_set, 64 * WM * WN, true, 512);
}

template <int MT, int NT, int WM, int WN, bool PF2>
int launch_f16x3_apre(const GemmP& p, int batch, hipStream_t st) {
  constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
  constexpr size_t smem = (size_t)2 * (2 * BM * LDH + 2 * BN * LDH) * sizeof(_Float16);
  static bool attr_set = false;
  return launch(gemm_f16x3_apre_kernel<MT, NT, WM, WN, PF2>, smem, p, BM, BN, batch, st, &attr_set, 64 * WM * WN, true);
}

}  // namespace

namespace pfpp_gemm_detail { namespace pl { extern thread_local char last_kernel[96]; } }

extern "C" int pfpp_gemm(const pfpp_gemm_args* a, pfpp_stream_t stream) {
  pfpp_gemm_detail::pl::last_kernel[0] = 0;
  PFPP_REQUIRE(a && (a->A || (a->a_hi && a->a_lo) || (a->gather_idx && a->lda == 0)) && (a->C || (a->c_hi && a->c_lo)), "null pointer");
  PFPP_REQUIRE(a->W || (a->w_hi && a->w_lo), "W (or its pre-split planes) missing");
  PFPP_REQUIRE(a->M >= 0 && a->N > 0 && a->K > 0, "bad sizes");
  PFPP_REQUIRE(a->M < (1ll << 31) && a->N < (1ll << 31) && a->K < (1ll << 31), "sizes exceed int32");
  const bool apre = a->a_hi != nullptr;
  if (apre) {
    PFPP_REQUIRE(a->a_lo && a->w_hi && a->w_lo && (a->precision == PFPP_GEMM_F16X3 || a->precision == PFPP_GEMM_F16) && !a->w_kmajor,
                 "pre-split A needs the f16x3 path with a pre-split [N,K] W");
    PFPP_REQUIRE(a->K % 32 == 0 && a->lda % 8 == 0 && a->lda >= a->K, "pre-split A: K % 32 == 0, lda % 8 == 0");
    PFPP_REQUIRE(pfpp::aligned16(a->a_hi) && pfpp::aligned16(a->a_lo) && a->sA0 % 8 == 0 && a->sA1 % 8 == 0,
                 "pre-split A planes must be 16-byte aligned");
    PFPP_SUPPORTED(a->N > 64 || a->act == PFPP_ACT_GEGLU, "pre-split A with N <= 64");
  } else if (a->gather_idx) {
    PFPP_REQUIRE(a->gather_xyz && a->gather_ctr && a->gather_N > 0 && a->gather_S > 0 && a->gather_ns > 0, "fused grouping: missing tables");
    PFPP_REQUIRE(a->lda % 32 == 0 && a->K == a->lda + 4 && (a->lda == 0 || pfpp::aligned16(a->A)), "fused grouping: K must be D + 4 with D = lda, D % 32 == 0");
    PFPP_REQUIRE(a->M % ((int64_t)a->gather_S * a->gather_ns) == 0, "fused grouping: M must be F*S*ns");
    PFPP_SUPPORTED(a->precision == PFPP_GEMM_F16X3 && a->w_hi && a->batch == 1 && !a->a_mul && !a->w_kmajor,
                   "fused grouping needs the f16x3 path with pre-split W, batch 1, no fused BatchNorm input");
  } else {
    PFPP_REQUIRE(a->lda % 4 == 0 && pfpp::aligned16(a->A), "lda must be a multiple of 4 and A 16-byte aligned");
    PFPP_REQUIRE(a->lda >= ((a->K + 3) & ~3ll), "lda smaller than K rounded up to 4");
  }
  PFPP_REQUIRE(!a->c_hi || (a->c_lo && a->pool == 0), "split output: c_lo missing or combined with pooling");
  PFPP_REQUIRE(a->batch >= 1 && a->zdiv >= 1, "batch/zdiv must be >= 1");
  PFPP_REQUIRE((a->sA0 % 4 == 0) && (a->sA1 % 4 == 0), "batch strides of A must keep 16-byte alignment");
  PFPP_REQUIRE(!a->scale || a->shift, "scale without shift");
  PFPP_REQUIRE(!a->residual || a->ldr > 0, "residual without ldr");
  PFPP_REQUIRE(a->pool == 0 || a->pool == 32 || a->pool == 64, "pool must be 0, 32 or 64");
  PFPP_REQUIRE(a->pool == 0 || (a->M % a->pool == 0 && !a->residual), "pool: M % pool != 0 or residual set");
  PFPP_REQUIRE(a->act >= PFPP_ACT_NONE && a->act <= PFPP_ACT_GEGLU, "unknown activation");
  PFPP_REQUIRE(a->act != PFPP_ACT_GEGLU || (a->N % 64 == 0 && a->pool == 0 && !a->scale && !a->residual),
               "GEGLU: N % 64 != 0 or unsupported epilogue combination");
  PFPP_REQUIRE(a->precision == PFPP_GEMM_F32 || a->precision == PFPP_GEMM_F16X3 || a->precision == PFPP_GEMM_F16, "unknown precision");
  PFPP_REQUIRE(a->precision != PFPP_GEMM_F16 || (a->a_hi && a->w_hi), "PFPP_GEMM_F16 needs both operands as pre-split planes");
  const bool fused_bn = a->a_mul || a->stats || a->c_min;
  if (fused_bn) {
    PFPP_REQUIRE(!a->a_mul == !a->a_add, "a_mul and a_add go together");
    PFPP_REQUIRE(!a->stats || a->stats_copies >= 1, "stats without stats_copies");
    PFPP_REQUIRE(!a->c_min || (a->pool != 0 && !a->c_hi), "c_min needs pooling and an fp32 output");
    PFPP_SUPPORTED(a->precision == PFPP_GEMM_F16X3 && a->w_hi && !apre && a->batch == 1 && a->act == PFPP_ACT_NONE &&
                   !a->scale && a->K % 32 == 0 || !a->a_mul,
                   "fused BatchNorm input needs the f16x3 path with pre-split W, K % 32 == 0, no activation");
    PFPP_SUPPORTED(a->precision == PFPP_GEMM_F16X3 && a->w_hi && !apre && a->batch == 1, "fused BatchNorm on this GEMM variant");
  }
  const bool pre = a->w_hi != nullptr;
  if (pre) {
    PFPP_REQUIRE((a->precision == PFPP_GEMM_F16X3 || a->precision == PFPP_GEMM_F16) && a->w_lo && !a->w_kmajor, "pre-split W needs the f16x3 path, [N,K] layout");
    PFPP_REQUIRE(a->ldw % 8 == 0 && a->ldw >= ((a->K + 7) & ~7ll), "pre-split W: ldw (halfs) must be K rounded up to 8");
    PFPP_REQUIRE(pfpp::aligned16(a->w_hi) && pfpp::aligned16(a->w_lo) && a->sW0 % 8 == 0 && a->sW1 % 8 == 0,
                 "pre-split W planes must be 16-byte aligned");
  } else {
    PFPP_REQUIRE(a->ldw % 4 == 0 && pfpp::aligned16(a->W), "ldw must be a multiple of 4 and W 16-byte aligned");
    PFPP_REQUIRE(a->w_kmajor ? a->ldw >= a->N : a->ldw >= ((a->K + 3) & ~3ll), "ldw too small");
    PFPP_REQUIRE((a->sW0 % 4 == 0) && (a->sW1 % 4 == 0), "batch strides of W must keep 16-byte alignment");
  }
  if (a->M == 0) return PFPP_OK;

  GemmP p;
  memset(&p, 0, sizeof(p));          // (fields only the plane entry point sets — csum, defer — must read as absent here)
  p.A = a->A; p.W = a->W; p.C = a->C; p.Whi = a->w_hi; p.Wlo = a->w_lo;
  p.Ahi = a->a_hi; p.Alo = a->a_lo; p.Chi = a->c_hi; p.Clo = a->c_lo;
  p.bias = a->bias; p.scale = a->scale; p.shift = a->shift; p.residual = a->residual;
  p.M = (int)a->M; p.N = (int)a->N; p.K = (int)a->K;
  p.lda = a->lda; p.ldw = a->ldw; p.ldc = a->ldc; p.ldr = a->ldr;
  p.act = a->act; p.pool = a->pool; p.zdiv = a->zdiv;
  p.sA0 = a->sA0; p.sA1 = a->sA1; p.sW0 = a->sW0; p.sW1 = a->sW1; p.sC0 = a->sC0; p.sC1 = a->sC1;
  p.sV0 = a->sV0; p.sV1 = a->sV1;
  p.alpha = a->alpha;
  p.a_mul = a->a_mul; p.a_add = a->a_add; p.stats = a->stats; p.stats_copies = a->stats_copies; p.Cmin = a->c_min;
  p.split_ws = a->split_ws; p.split_cnt = a->split_cnt; p.split_k = 1; p.k_chunk = 0;
  p.g_idx = a->gather_idx; p.g_xyz = a->gather_xyz; p.g_ctr = a->gather_ctr;
  p.g_N = a->gather_N; p.g_S = a->gather_S; p.g_ns = a->gather_ns;
  p_split_ws_bytes = a->split_ws ? a->split_ws_bytes : 0;
  p.ws_bytes = p_split_ws_bytes;
  p_split_cnt_len = a->split_cnt ? a->split_cnt_len : 0;
  p.tiles_n = 0;
  p.dbg = 0;
  p.accum = 0;
  p.x1 = a->precision == PFPP_GEMM_F16 ? 1 : 0;
  p.k_valid = (int)a->K;
  hipStream_t st = pfpp::as_stream(stream);

  // 128x128 tiles unless N is narrow (GEGLU and pool=64 need the 2-tile wave shape)
  const bool wide = a->N > 64 || a->act == PFPP_ACT_GEGLU;
  if ((a->precision == PFPP_GEMM_F16X3 || a->precision == PFPP_GEMM_F16) && !a->w_kmajor) {
    // LDS-DMA staged, software-pipelined plane kernel (gemm_pl.hip).  PFPP_GEMM_PL: 0 = off, 1..3 = force a tile, unset / -1 = by shape
    if (apre && !fused_bn && true) {
      const char* e = getenv("PFPP_GEMM_PL");
      const int v = e ? atoi(e) : -1;
      if (v != 0) return launch_f16x3_planes(p, a->batch, st, gemm_group_m(), v < 0 ? 0 : v);
    }
    if (apre) {
      // the register-staged kernels with A staged like W (no conversions in the loop); same tile choice as below
      static const bool big = !(getenv("PFPP_GEMM_BIG") && atoi(getenv("PFPP_GEMM_BIG")) == 0);
      if (big && a->M >= 8192 && a->N >= 1024 && a->pool == 0) return launch_f16x3_apre<4, 2, 2, 4, false>(p, a->batch, st);
      if (big && a->M >= 8192 && a->pool != 32) return launch_f16x3_apre<2, 2, 4, 2, true>(p, a->batch, st);
      const int64_t t128 = ((a->M + 127) / 128) * ((a->N + 127) / 128) * a->batch;
      if (t128 < 1024 && a->act != PFPP_ACT_GEGLU && a->pool == 0) return launch_f16x3_apre<2, 1, 2, 2, true>(p, a->batch, st);
      return launch_f16x3_apre<2, 2, 2, 2, true>(p, a->batch, st);
    }
    // big-M GEMMs with a short contraction (the train-mode set-abstraction MLPs: 1.26 M rows, K = 64..256, fused BatchNorm
    // operand / statistics / pooling): LDS-DMA staged plane kernel with the fp32 A operand converted at fragment-read time
    static const bool af32 = !(getenv("PFPP_GEMM_AF32") && atoi(getenv("PFPP_GEMM_AF32")) == 0);
    if (af32 && pre && !apre && !a->gather_idx && a->batch == 1 && a->M >= 65536 && a->K % 32 == 0 && a->K <= 256 && a->lda % 4 == 0 &&
        a->act != PFPP_ACT_GEGLU && !a->c_hi && (a->pool == 0 || a->pool == 32 || a->N > 64))
      return launch_f16x3_planes_af32(p, a->batch, st, gemm_group_m());
    static const bool big_tile = !(getenv("PFPP_GEMM_BIG") && atoi(getenv("PFPP_GEMM_BIG")) == 0);
    // 256x128 tile (8 waves): 1.33x more matrix work per byte staged; worth it when there are enough
    // row panels to fill the chip several times over
    // operand delivery from L2 is ~11 B/clk/CU whatever the load structure (profiles/README.md), so the
    // matrix pipe's utilisation is set by bytes per MFMA ~ (BM+BN)/(BM*BN): 256x256 (8 waves of 128x64)
    // where the grid still fills the chip, 256x128 for narrower N
    if (pre && wide && big_tile && a->M >= 8192 && a->N >= 1024 && a->pool == 0 && !a->a_mul)
      return launch_f16x3<4, 2, true, 2, 4>(p, a->batch, st);
    // train-mode encoder GEMMs (fused BatchNorm operands) run on their own stream UNDER the transformer's latency-bound
    // kernels: a tile whose LDS footprint leaves room for a second workgroup lets those co-reside (160 KB per CU:
    // 256x128 = 120 KB blocks a 61 KB transformer tile, 128x128 = 80 KB does not)
    static const int bn_tile = getenv("PFPP_GEMM_BN_TILE") ? atoi(getenv("PFPP_GEMM_BN_TILE")) : 0;
    if (pre && wide && fused_bn && bn_tile > 0 && a->M >= 8192) {
      if (bn_tile == 3) return launch_f16x3<2, 2, true>(p, a->batch, st);     // one K-tile in flight, more workgroups per CU
      if (bn_tile == 2 && a->pool == 0)
        return a->a_mul ? launch_f16x3<2, 1, true, 2, 2, true, true>(p, a->batch, st) : launch_f16x3<2, 1, true, 2, 2, true>(p, a->batch, st);
      return a->a_mul ? launch_f16x3<2, 2, true, 2, 2, true, true>(p, a->batch, st) : launch_f16x3<2, 2, true, 2, 2, true>(p, a->batch, st);
    }
    static const bool pf2_big = !(getenv("PFPP_GEMM_PF2BIG") && atoi(getenv("PFPP_GEMM_PF2BIG")) == 0);   // 16000x512x2048: 132 -> 111 us
    if (pre && wide && big_tile && a->M >= 8192 && a->pool != 32)
      return !pf2_big ? launch_f16x3<2, 2, true, 4, 2>(p, a->batch, st)
             : a->a_mul ? launch_f16x3<2, 2, true, 4, 2, true, true>(p, a->batch, st)
                        : launch_f16x3<2, 2, true, 4, 2, true>(p, a->batch, st);
    // small grids: a 128x128 tiling that cannot fill the 2 x 256 workgroup slots twice over runs as 128x64
    // tiles (twice the workgroups, same per-wave work shape) — GEGLU / pool=64 need the 2-tile-wide wave
    static const int small_thresh = getenv("PFPP_GEMM_SMALL") ? atoi(getenv("PFPP_GEMM_SMALL")) : 1024;
    const int64_t tiles128 = ((a->M + 127) / 128) * ((a->N + 127) / 128) * a->batch;
    // grids of at most ~two workgroups per CU: prefetch two K-tiles ahead (the load latency is all there is to hide)
    // latency-bound launches (see gemm_f16x3_deep_kernel).  PFPP_GEMM_DEEP: 0 off, 1 tiny grids only, 2 also the 3850-row grids
    // Measured (M = 250: one puzzle's tokens): 512x512 15.9 -> 12.3 us, 1536x512 16.2 -> 12.5, GEGLU 4096x512 24.7 -> 17.8,
    // 512x2048 (K split into 512-deep chunks instead of 128-deep ones) 40.8 -> 29.7 us; the single-puzzle auto_aggl loop
    // 4.39 -> 5.28 puzzles/s.  No gain on the wider 3850-row grids (PFPP_GEMM_DEEP=2 routes them here too).  0 = off.
    static const int deep_mode = getenv("PFPP_GEMM_DEEP") ? atoi(getenv("PFPP_GEMM_DEEP")) : 1;
    if (pre && deep_mode > 0 && !fused_bn && !a->gather_idx && a->pool == 0) {
      const int64_t t64 = ((a->M + 63) / 64) * ((a->N + 63) / 64) * a->batch;
      if (t64 <= 256) {      // 3850 x 512 (488 tiles) measured 19.1 us here vs 18.3 us with the 128x64 two-deep kernel
        if (a->act == PFPP_ACT_GEGLU) return launch_f16x3_deep<1, 2, 2, 2, 4>(p, a->batch, st);
        return launch_f16x3_deep<1, 1, 2, 2, 8>(p, a->batch, st);
      }
      if (deep_mode > 1 && wide && tiles128 < small_thresh && a->act != PFPP_ACT_GEGLU)
        return launch_f16x3_deep<2, 1, 2, 2, 4>(p, a->batch, st);
    }
    static const bool pf2 = !(getenv("PFPP_GEMM_PF2") && atoi(getenv("PFPP_GEMM_PF2")) == 0);   // two-deep prefetch, branch-free steady state: +10..23 % at 3850 rows
    const bool deep = pf2 && !fused_bn && tiles128 < 2 * small_thresh;
    if (pre && wide && tiles128 < small_thresh && a->act != PFPP_ACT_GEGLU && a->pool == 0)
      return deep ? launch_f16x3<2, 1, true, 2, 2, true>(p, a->batch, st) : launch_f16x3<2, 1, true>(p, a->batch, st);
    if (pre && deep)
      return wide ? launch_f16x3<2, 2, true, 2, 2, true>(p, a->batch, st) : launch_f16x3<2, 1, true, 2, 2, true>(p, a->batch, st);
    if (pre) return wide ? launch_f16x3<2, 2, true>(p, a->batch, st) : launch_f16x3<2, 1, true>(p, a->batch, st);
    return wide ? launch_f16x3<2, 2, false>(p, a->batch, st) : launch_f16x3<2, 1, false>(p, a->batch, st);
  }
  if (a->w_kmajor) {
    return wide ? launch_f32<2, 2, true>(p, a->batch, st) : launch_f32<2, 1, true>(p, a->batch, st);
  }
  return wide ? launch_f32<2, 2, false>(p, a->batch, st) : launch_f32<2, 1, false>(p, a->batch, st);
}
