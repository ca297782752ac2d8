// Shared pieces of the two GEMM kernels (fp32-MFMA exact path, split-f16 x3 path): launch
// parameters, tile-id remap, and the fused epilogue (bias / BN scale+shift / activation / GEGLU gate /
// residual / max over row groups).  Accumulator layout of a 32x32 MFMA tile (dtype independent on
// gfx950): element e of lane l is C[row = (e&3) + 8*(e>>2) + 4*(l>>5)][col = l&31].
#pragma once
#include "pfpp_common.h"

namespace pfpp_gemm_detail {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmP {
  const float* A; const float* W; float* C;
  const void* Whi; const void* Wlo;          // pre-split fp16 planes of W (split path) or null
  const void* Ahi; const void* Alo;          // pre-split fp16 planes of A or null
  void* Chi; void* Clo;                      // write the result as split planes instead of C
  const float* bias; const float* scale; const float* shift; const float* residual;
  int M, N, K;
  int64_t lda, ldw, ldc, ldr;
  int act, pool, zdiv;
  int64_t sA0, sA1, sW0, sW1, sC0, sC1, sV0, sV1;
  float alpha;
  int tiles_n, tiles_m, group_m;
  // train-mode BatchNorm fusion (bn_train.hip): A is read as relu(A*a_mul[k] + a_add[k]); per-column sum / sum of
  // squares of the (bias-added) result go to stats[copy][0..1][N] (fp64 atomics, copy = workgroup % stats_copies);
  // with pooling the per-group minimum goes to Cmin next to the maximum in C
  const float* a_mul; const float* a_add;
  double* stats; int stats_copies;
  float* Cmin;
  // split-K with a fix-up (skinny GEMMs: a handful of output tiles and a long K): blockIdx.y walks K chunks of
  // k_chunk; every workgroup parks its accumulators in split_ws, the last one to arrive at a tile (ticket from
  // split_cnt) adds the parked partials in chunk order (deterministic) and runs the normal epilogue
  float* split_ws; int* split_cnt; int split_k; int k_chunk;
  // fused grouping: row r of A is [A[f*g_N + g_idx[r]] (lda = D floats) | g_xyz[f, idx] - g_ctr[r / g_ns] | 0]
  const int* g_idx; const float* g_xyz; const float* g_ctr; int g_N, g_S, g_ns;
  int64_t ws_bytes;   // capacity of split_ws in bytes
  int k_valid; // gemm_pl.hip: contraction rows that exist (k-major operands; K is rounded up to the K-tile, the rest reads zeros)
  int x1;     // gemm_pl.hip: single-pass fp16 (hi planes only) — PFPP_GEMM_F16
  float* csum; float* csum_ws; float csum_alpha;   // gemm_pl.hip (k-major A): csum[m] += csum_alpha * sum_k A[k][m] — the bias gradient riding in dW = dY^T . X
  int accum;  // gemm_pl.hip: C += alpha * acc with fp32 atomics (set by the launcher for split-K / gradient accumulation)
  pfpp_slab_job* defer;   // host side only (gemm_pl.hip launch_pl): hand the slab reduction back instead of launching it
  int dbg;    // gemm_pl.hip ablation switches (PFPP_GEMM_DBG; developer runs only): 1 no epilogue, 2 no DMA after the prologue, 4 no barrier / DMA wait
};

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case PFPP_ACT_RELU: return v > 0.0f ? v : 0.0f;
    case PFPP_ACT_SILU: return v / (1.0f + expf(-v));
    case PFPP_ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    default: return v;
  }
}

// XCD-aware bijective remap of the workgroup id: consecutive tiles (same A row panel) land on
// the same XCD and share its L2 (cdna guide T1, bijective form)
__device__ __forceinline__ int remap_tile(int bid, int nwg) {
  const int xcd = bid & 7, local = bid >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

// tile id -> (tm, tn).  group_m > 0: ids walk group_m consecutive row panels column by column, so
// the ~64 tiles an XCD runs at once span group_m panels of A and 64/group_m panels of W (L2 reuse of
// both operands); group_m == 0: plain row-major order.
__device__ __forceinline__ void tile_coords(const GemmP& p, int tile, int& tm, int& tn) {
  if (p.group_m > 0) {
    const int group = p.group_m * p.tiles_n;
    const int g = tile / group;
    const int first_m = g * p.group_m;
    const int gsz = min(p.tiles_m - first_m, p.group_m);
    const int r = tile - g * group;
    tm = first_m + r % gsz;
    tn = r / gsz;
  } else {
    tm = tile / p.tiles_n;
    tn = tile - tm * p.tiles_n;
  }
}

__device__ __forceinline__ void store_out(const GemmP& p, int64_t idx, float v) {
  if (p.Chi) {
    _Float16 hi, lo;
    PFPP_SPLIT_TO(v, hi, lo);
    reinterpret_cast<_Float16*>(p.Chi)[idx] = hi;
    reinterpret_cast<_Float16*>(p.Clo)[idx] = lo;
  } else {
    p.C[idx] = v;
  }
}

// activation over one accumulator tile: the switch sits outside the element loop, so the fully unrolled
// epilogue touches each accumulator exactly once (a per-element switch, or one epilogue copy per
// activation, keeps the 256-register accumulator array of the 4x4-tile waves in scratch)
__device__ __forceinline__ f32x16 act_tile(f32x16 t, int act) {
  switch (act) {
    case PFPP_ACT_RELU:
#pragma unroll
      for (int e = 0; e < 16; ++e) t[e] = t[e] > 0.0f ? t[e] : 0.0f;
      break;
    case PFPP_ACT_SILU:
#pragma unroll
      for (int e = 0; e < 16; ++e) t[e] = t[e] / (1.0f + expf(-t[e]));
      break;
    case PFPP_ACT_GELU:
#pragma unroll
      for (int e = 0; e < 16; ++e) t[e] = 0.5f * t[e] * (1.0f + erff(t[e] * 0.70710678118654752440f));
      break;
    default: break;
  }
  return t;
}

// acc[i][j]: MT x NT tiles of the wave whose top-left element is (row_w, col_w)
template <int MT, int NT>
__device__ __forceinline__ void epilogue(const GemmP& p, f32x16 (&acc)[MT][NT], int row_w, int col_w,
                                         int n0, int wn, int lane, int64_t c_off, int64_t v_off) {
  (void)n0; (void)wn;
  const int l31 = lane & 31, lhi = lane >> 5;
  const float* R = p.residual ? p.residual + c_off : nullptr;
  const float* bias = p.bias ? p.bias + v_off : nullptr;
  const float* scale = p.scale ? p.scale + v_off : nullptr;
  const float* shift = p.shift ? p.shift + v_off : nullptr;
  const float alpha = p.alpha;
  const bool geglu = p.act == PFPP_ACT_GEGLU;
  const int pool = p.pool;

#pragma unroll
  for (int j = 0; j < NT; ++j) {
    // GEGLU: value columns in the even tiles, their gate columns in the next odd tile (host packing:
    // 32 value columns then their 32 gate columns per 64 packed columns); the odd tile's pass gates
    // what the even tile's pass left in `held`
    const int col = col_w + j * 32 + l31;
    const bool col_ok = col < p.N;
    float sc = 1.0f, sh = 0.0f;
    if (col_ok) {
      if (scale) { sc = scale[col]; sh = shift[col]; }
      else if (bias) { sh = bias[col]; }
    }
    const int ocol = geglu ? ((col_w + (j & ~1) * 32) >> 1) + l31 : col;
    float mx[MT], mn[MT];
    float st_s = 0.0f, st_q = 0.0f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      f32x16 t = acc[i][j];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float v = t[e] * alpha;
        t[e] = scale ? __builtin_fmaf(v, sc, sh) : v + sh;      // explicit: sa_fused.hip reproduces this epilogue bit for bit
      }
      if (p.stats) {      // BatchNorm batch statistics of the pre-activation (act is NONE on this path)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = row_w + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
          const float v = row < p.M ? t[e] : 0.0f;
          st_s += v;
          st_q += v * v;
        }
      }
      if (geglu) {
        if constexpr (NT >= 2) {
          if ((j & 1) == 0) { acc[i][j] = t; continue; }       // value tile: keep u for the gate pass
          t = act_tile(t, PFPP_ACT_GELU);
          const f32x16 u = acc[i][j - 1];
#pragma unroll
          for (int e = 0; e < 16; ++e) t[e] = u[e] * t[e];
        }
      } else {
        t = act_tile(t, p.act);
      }
      if (pool == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = row_w + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
          if (row < p.M && col_ok) {
            float v = t[e];
            if (R) v += R[(int64_t)row * p.ldr + col];
            store_out(p, c_off + (int64_t)row * p.ldc + ocol, v);
          }
        }
      } else {
        // max over groups of `pool` consecutive rows (pool = 32: one MFMA tile, pool = 64: both
        // M-tiles of the wave); groups never straddle M
        float m = -__builtin_huge_valf(), n = __builtin_huge_valf();
#pragma unroll
        for (int e = 0; e < 16; ++e) { m = fmaxf(m, t[e]); n = fminf(n, t[e]); }
        mx[i] = fmaxf(m, __shfl_xor(m, 32));
        mn[i] = fminf(n, __shfl_xor(n, 32));
      }
    }
    if (p.stats) {
      st_s += __shfl_xor(st_s, 32);
      st_q += __shfl_xor(st_q, 32);
      if (lhi == 0 && col_ok) {
        double* st = p.stats + (size_t)(blockIdx.x % p.stats_copies) * 2 * p.N;
        unsafeAtomicAdd(st + col, (double)st_s);
        unsafeAtomicAdd(st + p.N + col, (double)st_q);
      }
    }
    if (pool == 64) {
      if constexpr (MT == 2) {
        if (lhi == 0 && col_ok && row_w < p.M) {
          store_out(p, c_off + (int64_t)(row_w >> 6) * p.ldc + col, fmaxf(mx[0], mx[1]));
          if (p.Cmin) p.Cmin[c_off + (int64_t)(row_w >> 6) * p.ldc + col] = fminf(mn[0], mn[1]);
        }
      }
    } else if (pool == 32) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row0 = row_w + i * 32;
        if (lhi == 0 && col_ok && row0 < p.M) {
          store_out(p, c_off + (int64_t)(row0 >> 5) * p.ldc + col, mx[i]);
          if (p.Cmin) p.Cmin[c_off + (int64_t)(row0 >> 5) * p.ldc + col] = mn[i];
        }
      }
    }
  }
}

}  // namespace pfpp_gemm_detail
