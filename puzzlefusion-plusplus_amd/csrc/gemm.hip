// fp32 GEMM with fused epilogues on the CDNA4 matrix cores (gfx950).
//
//   C[M,N] = epilogue( A[M,K] . op(W) ),  op(W) = W^T for W[N,K] (torch Linear /
//   1x1-conv weight layout) or W for W[K,N] (attention P.V).
//
// v_mfma_f32_32x32x2_f32 computes exact fp32 products with fp32 accumulation
// (bitwise a k-ordered fmaf chain), which is what the 1e-4 parity bar on the
// predicted noise needs; its rate (157 TFLOP/s dense) is the roofline of every
// contraction on this path.
//
// Structure: 256 threads = 4 waves as 2x2; each wave owns an (MT*32)x(NT*32)
// sub-tile held in MT*NT accumulators of 16 VGPRs.  K advances in tiles of 32:
// the next tile is fetched HBM->VGPR (16-byte loads, coalesced along K) while
// the current one is multiplied out of LDS; one barrier per K-tile.  LDS rows are
// padded to 36 floats so the 16-byte fragment reads (ds_read_b128, one row per
// lane) are bank-conflict free.  Within an 8-wide K chunk lanes 0-31 feed
// k = 0..3 and lanes 32-63 feed k = 4..7 to the four MFMAs, so one ds_read_b128
// per operand serves four matrix instructions.
// Workgroup ids are remapped so that consecutive tiles (which share the A
// row-panel) run on the same XCD and hit its L2.
#include "pfpp_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;  // 36 floats = 144 B: 16-byte aligned, conflict-free b128 reads

struct GemmP {
  const float* A; const float* W; float* C;
  const float* bias; const float* scale; const float* shift; const float* residual;
  int M, N, K;
  int64_t lda, ldw, ldc, ldr;
  int act, pool, zdiv;
  int64_t sA0, sA1, sW0, sW1, sC0, sC1, sV0, sV1;
  float alpha;
  int tiles_n;
};

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case PFPP_ACT_RELU: return v > 0.0f ? v : 0.0f;
    case PFPP_ACT_SILU: return v / (1.0f + expf(-v));
    case PFPP_ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    default: return v;
  }
}

// 16-byte global load of A/W elements [k, k+4) of one row, zero beyond K
__device__ __forceinline__ float4 load_k4(const float* row, int k, int K, bool row_ok) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row_ok && k < K) {
    v = *reinterpret_cast<const float4*>(row + k);
    if (k + 4 > K) {  // ragged tail: the padding of the row may hold anything
      if (k + 1 >= K) v.y = 0.f;
      if (k + 2 >= K) v.z = 0.f;
      v.w = 0.f;
    }
  }
  return v;
}

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

  // XCD-aware bijective remap of the tile id (cdna guide T1)
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, local = bid >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int z = blockIdx.z;
  const int z0 = z / p.zdiv, z1 = z - z0 * p.zdiv;
  const float* A = p.A + z0 * p.sA0 + z1 * p.sA1;
  const float* W = p.W + z0 * p.sW0 + z1 * p.sW1;
  const int64_t c_off = z0 * p.sC0 + z1 * p.sC1;
  const int64_t v_off = z0 * p.sV0 + z1 * p.sV1;
  const float* bias = p.bias ? p.bias + v_off : nullptr;
  const float* scale = p.scale ? p.scale + v_off : nullptr;
  const float* shift = p.shift ? p.shift + v_off : nullptr;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  // ---- global -> register staging ---------------------------------------
  float4 ra[A_IT], rw[W_IT];
  const int a_row = tid >> 3, a_c4 = tid & 7;  // 8 lanes cover one 128-byte row slice
  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int row = a_row + 32 * it;
      const int gm = m0 + row;
      ra[it] = load_k4(A + (int64_t)gm * p.lda, k0 + a_c4 * 4, p.K, gm < p.M);
    }
    if (!WK) {
#pragma unroll
      for (int it = 0; it < W_IT; ++it) {
        const int row = a_row + 32 * it;
        const int gn = n0 + row;
        rw[it] = load_k4(W + (int64_t)gn * p.ldw, k0 + a_c4 * 4, p.K, gn < p.N);
      }
    } else {
      constexpr int C4 = BN / 4;             // float4 per k-row
      constexpr int RPI = 256 / C4;          // k-rows per iteration
#pragma unroll
      for (int it = 0; it < W_IT; ++it) {
        const int kr = tid / C4 + RPI * it;
        const int c4 = tid % C4;
        const int gk = k0 + kr;
        const int gn = n0 + c4 * 4;
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

  const int nk = (p.K + BK - 1) / BK;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();

  const int l31 = lane & 31, lhi = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tiles((kt + 1) * BK);
    const float* as = As + buf * BM * LDS_LD + (wm * 32 * MT + l31) * LDS_LD + lhi * 4;
    const float* ws = Ws + buf * W_TILE;
    const int krem = p.K - kt * BK;
    const int nchunk = krem >= BK ? 4 : (krem + 7) >> 3;
    for (int kc = 0; kc < nchunk; ++kc) {
      float4 a[MT], b[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i)
        a[i] = *reinterpret_cast<const float4*>(as + i * 32 * LDS_LD + kc * 8);
      if (!WK) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
          b[j] = *reinterpret_cast<const float4*>(ws + (wn * 32 * NT + j * 32 + l31) * LDS_LD +
                                                  lhi * 4 + kc * 8);
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
    }
    if (kt + 1 < nk) store_tiles(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue -------------------------------------------------------------
  // accumulator element e of a 32x32 tile: col = lane&31, row = (e&3)+8*(e>>2)+4*(lane>>5)
  float* C = p.C + c_off;
  const float* R = p.residual ? p.residual + c_off : nullptr;
  const float alpha = p.alpha;

  if (p.act == PFPP_ACT_GEGLU) {
    if constexpr (NT == 2) {
      // value columns in tile j=0, their gate columns in tile j=1 (host packing)
      const int ncol = n0 + wn * 64 + l31;            // packed value column
      const int ocol = (n0 >> 1) + wn * 32 + l31;     // output column
      const bool col_ok = ncol + 32 < p.N;
      const float bu = (bias && col_ok) ? bias[ncol] : 0.0f;
      const float bg = (bias && col_ok) ? bias[ncol + 32] : 0.0f;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = m0 + wm * 32 * MT + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
          if (row < p.M && col_ok) {
            const float u = acc[i][0][e] * alpha + bu;
            const float g = acc[i][1][e] * alpha + bg;
            C[(int64_t)row * p.ldc + ocol] = u * act_apply(g, PFPP_ACT_GELU);
          }
        }
    }
    return;
  }

#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = n0 + wn * 32 * NT + j * 32 + l31;
    const bool col_ok = col < p.N;
    float sc = 1.0f, sh = 0.0f;
    if (col_ok) {
      if (scale) { sc = scale[col]; sh = shift[col]; }
      else if (bias) { sh = bias[col]; }
    }
    if (p.pool == 0) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = m0 + wm * 32 * MT + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
          if (row < p.M && col_ok) {
            float v = acc[i][j][e] * alpha;
            v = scale ? v * sc + sh : v + sh;
            v = act_apply(v, p.act);
            if (R) v += R[(int64_t)row * p.ldr + col];
            C[(int64_t)row * p.ldc + col] = v;
          }
        }
    } else {
      // max over groups of `pool` consecutive rows (pool = 32: one MFMA tile,
      // pool = 64: both M-tiles of the wave); groups never straddle M
      float mx[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        float m = -__builtin_huge_valf();
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          float v = acc[i][j][e] * alpha;
          v = scale ? v * sc + sh : v + sh;
          v = act_apply(v, p.act);
          m = fmaxf(m, v);
        }
        m = fmaxf(m, __shfl_xor(m, 32));
        mx[i] = m;
      }
      if (p.pool == 64) {
        if constexpr (MT == 2) {
          const int row0 = m0 + wm * 64;
          if (lhi == 0 && col_ok && row0 < p.M)
            C[(int64_t)(row0 >> 6) * p.ldc + col] = fmaxf(mx[0], mx[1]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int row0 = m0 + wm * 32 * MT + i * 32;
          if (lhi == 0 && col_ok && row0 < p.M) C[(int64_t)(row0 >> 5) * p.ldc + col] = mx[i];
        }
      }
    }
  }
}

template <int MT, int NT, bool WK>
int launch_gemm(const GemmP& p, int batch, hipStream_t st) {
  constexpr int BM = 64 * MT, BN = 64 * NT;
  constexpr size_t smem = (size_t)(2 * BM * LDS_LD + 2 * (WK ? BK * BN : BN * LDS_LD)) * sizeof(float);
  static bool attr_set = false;
  auto kern = gemm_f32_mfma_kernel<MT, NT, WK>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  GemmP q = p;
  const int tiles_m = (p.M + BM - 1) / BM;
  q.tiles_n = (p.N + BN - 1) / BN;
  const dim3 grid((unsigned)(tiles_m * q.tiles_n), 1, (unsigned)batch);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, q);
  return pfpp::check_launch("pfpp_gemm");
}

}  // namespace

extern "C" int pfpp_gemm(const pfpp_gemm_args* a, pfpp_stream_t stream) {
  PFPP_REQUIRE(a && a->A && a->W && a->C, "null pointer");
  PFPP_REQUIRE(a->M >= 0 && a->N > 0 && a->K > 0, "bad sizes");
  PFPP_REQUIRE(a->M < (1ll << 31) && a->N < (1ll << 31) && a->K < (1ll << 31), "sizes exceed int32");
  PFPP_REQUIRE(a->lda % 4 == 0 && a->ldw % 4 == 0, "lda/ldw must be multiples of 4");
  PFPP_REQUIRE(pfpp::aligned16(a->A) && pfpp::aligned16(a->W), "A/W must be 16-byte aligned");
  PFPP_REQUIRE(a->lda >= ((a->K + 3) & ~3ll), "lda smaller than K rounded up to 4");
  PFPP_REQUIRE(a->w_kmajor ? a->ldw >= a->N : a->ldw >= ((a->K + 3) & ~3ll), "ldw too small");
  PFPP_REQUIRE(a->batch >= 1 && a->zdiv >= 1, "batch/zdiv must be >= 1");
  PFPP_REQUIRE((a->sA0 % 4 == 0) && (a->sA1 % 4 == 0) && (a->sW0 % 4 == 0) && (a->sW1 % 4 == 0),
               "batch strides of A/W must keep 16-byte alignment");
  PFPP_REQUIRE(!a->scale || a->shift, "scale without shift");
  PFPP_REQUIRE(!a->residual || a->ldr > 0, "residual without ldr");
  PFPP_REQUIRE(a->pool == 0 || a->pool == 32 || a->pool == 64, "pool must be 0, 32 or 64");
  PFPP_REQUIRE(a->pool == 0 || (a->M % a->pool == 0 && !a->residual), "pool: M % pool != 0 or residual set");
  PFPP_REQUIRE(a->act >= PFPP_ACT_NONE && a->act <= PFPP_ACT_GEGLU, "unknown activation");
  PFPP_REQUIRE(a->act != PFPP_ACT_GEGLU || (a->N % 64 == 0 && a->pool == 0 && !a->scale && !a->residual),
               "GEGLU: N % 64 != 0 or unsupported epilogue combination");
  if (a->M == 0) return PFPP_OK;

  GemmP p;
  p.A = a->A; p.W = a->W; p.C = a->C;
  p.bias = a->bias; p.scale = a->scale; p.shift = a->shift; p.residual = a->residual;
  p.M = (int)a->M; p.N = (int)a->N; p.K = (int)a->K;
  p.lda = a->lda; p.ldw = a->ldw; p.ldc = a->ldc; p.ldr = a->ldr;
  p.act = a->act; p.pool = a->pool; p.zdiv = a->zdiv;
  p.sA0 = a->sA0; p.sA1 = a->sA1; p.sW0 = a->sW0; p.sW1 = a->sW1; p.sC0 = a->sC0; p.sC1 = a->sC1;
  p.sV0 = a->sV0; p.sV1 = a->sV1;
  p.alpha = a->alpha;
  p.tiles_n = 0;
  hipStream_t st = pfpp::as_stream(stream);

  // 128x128 tiles unless N is narrow (or GEGLU / pool=64 need the 2-tile wave shape)
  const bool wide = a->N > 64 || a->act == PFPP_ACT_GEGLU;
  if (a->w_kmajor) {
    return wide ? launch_gemm<2, 2, true>(p, a->batch, st) : launch_gemm<2, 1, true>(p, a->batch, st);
  }
  return wide ? launch_gemm<2, 2, false>(p, a->batch, st) : launch_gemm<2, 1, false>(p, a->batch, st);
}
