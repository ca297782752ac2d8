// a17: the memory-bound pieces of the training step — dropout, GEGLU forward/backward, LayerNorm
// backward, bias gradients, pooling / token / embedding backward, the loss and AdamW.  Everything here
// streams its operands once (float4 where the layout allows); column reductions finish with fp32
// hardware atomics.  The contractions live in gemm_grad.hip, the attention backward in
// attention_bwd.hip.
#include "pfpp_common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

inline unsigned blocks_for(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

__device__ __forceinline__ float gelu_f(float g) { return 0.5f * g * (1.0f + erff(g * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad(float g) {
  return 0.5f * (1.0f + erff(g * 0.70710678118654752440f)) + g * 0.39894228040143267794f * expf(-0.5f * g * g);
}
__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + expf(-v)); }
__device__ __forceinline__ float silu_grad(float v) {
  const float s = 1.0f / (1.0f + expf(-v));
  return s * (1.0f + v * (1.0f - s));
}

// ---------------------------------------------------------------------------------------------------
// column sums
// ---------------------------------------------------------------------------------------------------
// block = 64 column lanes (float4 each) x 4 row phases; grid (col chunks of 256, row chunks, batch)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                     int64_t rows, int cols, int64_t ld, int rows_per_block,
                                                     int64_t sx, int64_t so) {
  __shared__ float4 red[4][64];
  const int cl = threadIdx.x & 63, rp = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + cl) * 4;
  const float* xb = x + blockIdx.z * sx;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = min(rows, r0 + rows_per_block);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < cols) {
    for (int64_t r = r0 + rp; r < r1; r += 4) {
      const float4 v = *reinterpret_cast<const float4*>(xb + r * ld + c);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  red[rp][cl] = a;
  __syncthreads();
  if (rp == 0 && c < cols) {
    const float4 b = red[1][cl], d = red[2][cl], e = red[3][cl];
    float* o = out + blockIdx.z * so + c;
    unsafeAtomicAdd(o + 0, (a.x + b.x) + (d.x + e.x));
    unsafeAtomicAdd(o + 1, (a.y + b.y) + (d.y + e.y));
    unsafeAtomicAdd(o + 2, (a.z + b.z) + (d.z + e.z));
    unsafeAtomicAdd(o + 3, (a.w + b.w) + (d.w + e.w));
  }
}

// unaligned / narrow fallback (output head: 3 and 4 columns inside a 7-wide buffer): one wave per column
__global__ __launch_bounds__(64) void colsum_narrow_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                           int64_t rows, int64_t ld, int64_t sx, int64_t so) {
  const int c = blockIdx.x;
  const float* xb = x + blockIdx.z * sx;
  float a = 0.0f;
  for (int64_t r = threadIdx.x; r < rows; r += 64) a += xb[r * ld + c];
  a = wave_sum(a);
  if (threadIdx.x == 0) unsafeAtomicAdd(out + blockIdx.z * so + c, a);
}

// ---------------------------------------------------------------------------------------------------
// dropout
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                      float* __restrict__ out, int64_t n, uint32_t thresh,
                                                      float inv_keep, uint64_t seed, uint32_t site) {
  const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= n) return;
  if (i4 + 4 <= n) {
    float4 v = *reinterpret_cast<const float4*>(x + i4);
    float4 r = res ? *reinterpret_cast<const float4*>(res + i4) : make_float4(0.f, 0.f, 0.f, 0.f);
    r.x += pfpp_rng_u32(seed, site, i4 + 0) >= thresh ? v.x * inv_keep : 0.0f;
    r.y += pfpp_rng_u32(seed, site, i4 + 1) >= thresh ? v.y * inv_keep : 0.0f;
    r.z += pfpp_rng_u32(seed, site, i4 + 2) >= thresh ? v.z * inv_keep : 0.0f;
    r.w += pfpp_rng_u32(seed, site, i4 + 3) >= thresh ? v.w * inv_keep : 0.0f;
    *reinterpret_cast<float4*>(out + i4) = r;
  } else {
    for (int64_t i = i4; i < n; ++i)
      out[i] = (res ? res[i] : 0.0f) + (pfpp_rng_u32(seed, site, i) >= thresh ? x[i] * inv_keep : 0.0f);
  }
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t* __restrict__ keep, int64_t n, uint32_t thresh,
                                                           uint64_t seed, uint32_t site) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) keep[i] = pfpp_rng_u32(seed, site, i) >= thresh ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------
// GEGLU
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void geglu_kernel(const float* __restrict__ z, float* __restrict__ u, int64_t rows,
                                                    int inner, uint32_t thresh, float inv_keep, uint64_t seed,
                                                    uint32_t site, pfpp_planes_out po) {
  pfpp_chain_prio();
  const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;   // index into u [rows, inner]
  if (i4 >= rows * inner) return;
  const int64_t r = i4 / inner;
  const int c = (int)(i4 - r * inner);
  const float4 v = *reinterpret_cast<const float4*>(z + r * 2 * inner + c);
  const float4 g = *reinterpret_cast<const float4*>(z + r * 2 * inner + inner + c);
  float4 o;
  o.x = v.x * gelu_f(g.x); o.y = v.y * gelu_f(g.y); o.z = v.z * gelu_f(g.z); o.w = v.w * gelu_f(g.w);
  if (thresh) {
    o.x = pfpp_rng_u32(seed, site, i4 + 0) >= thresh ? o.x * inv_keep : 0.0f;
    o.y = pfpp_rng_u32(seed, site, i4 + 1) >= thresh ? o.y * inv_keep : 0.0f;
    o.z = pfpp_rng_u32(seed, site, i4 + 2) >= thresh ? o.z * inv_keep : 0.0f;
    o.w = pfpp_rng_u32(seed, site, i4 + 3) >= thresh ? o.w * inv_keep : 0.0f;
  }
  if (u) *reinterpret_cast<float4*>(u + i4) = o;
  if (po.hi) pfpp_store4_planes(po, i4, o);
}

__global__ __launch_bounds__(256) void geglu_bwd_kernel(const float* __restrict__ z, const float* __restrict__ du,
                                                        float* __restrict__ dz, int64_t rows, int inner,
                                                        uint32_t thresh, float inv_keep, uint64_t seed, uint32_t site,
                                                        pfpp_planes_out po) {
  pfpp_chain_prio();
  const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= rows * inner) return;
  const int64_t r = i4 / inner;
  const int c = (int)(i4 - r * inner);
  const float4 v = *reinterpret_cast<const float4*>(z + r * 2 * inner + c);
  const float4 g = *reinterpret_cast<const float4*>(z + r * 2 * inner + inner + c);
  float4 d = *reinterpret_cast<const float4*>(du + i4);
  if (thresh) {
    d.x = pfpp_rng_u32(seed, site, i4 + 0) >= thresh ? d.x * inv_keep : 0.0f;
    d.y = pfpp_rng_u32(seed, site, i4 + 1) >= thresh ? d.y * inv_keep : 0.0f;
    d.z = pfpp_rng_u32(seed, site, i4 + 2) >= thresh ? d.z * inv_keep : 0.0f;
    d.w = pfpp_rng_u32(seed, site, i4 + 3) >= thresh ? d.w * inv_keep : 0.0f;
  }
  float4 dv, dg;
  dv.x = d.x * gelu_f(g.x); dv.y = d.y * gelu_f(g.y); dv.z = d.z * gelu_f(g.z); dv.w = d.w * gelu_f(g.w);
  dg.x = d.x * v.x * gelu_grad(g.x); dg.y = d.y * v.y * gelu_grad(g.y);
  dg.z = d.z * v.z * gelu_grad(g.z); dg.w = d.w * v.w * gelu_grad(g.w);
  if (dz) {
    *reinterpret_cast<float4*>(dz + r * 2 * inner + c) = dv;
    *reinterpret_cast<float4*>(dz + r * 2 * inner + inner + c) = dg;
  }
  if (po.hi) {
    pfpp_store4_planes(po, r * 2 * inner + c, dv);
    pfpp_store4_planes(po, r * 2 * inner + inner + c, dg);
  }
}

// ---------------------------------------------------------------------------------------------------
// split-f16 planes of an fp32 tensor, and column sums of a tensor given as planes
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, int64_t n, pfpp_planes_out po) {
  const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= n) return;
  pfpp_store4_planes(po, i4, *reinterpret_cast<const float4*>(x + i4));
}

// block = 64 column lanes (4 columns each) x 4 row phases, like colsum_kernel; out[c] += out_scale * sum_r (hi + lo)[r, c]
__global__ __launch_bounds__(256) void colsum_planes_kernel(const _Float16* __restrict__ hi, const _Float16* __restrict__ lo,
                                                            float* __restrict__ out, int64_t rows, int cols, int64_t ld,
                                                            int rows_per_block, float out_scale) {
  typedef _Float16 h4_ __attribute__((ext_vector_type(4)));
  __shared__ float4 red[4][64];
  const int cl = threadIdx.x & 63, rp = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + cl) * 4;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = min(rows, r0 + rows_per_block);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < cols) {
    for (int64_t r = r0 + rp; r < r1; r += 4) {
      const h4_ h = *reinterpret_cast<const h4_*>(hi + r * ld + c);
      const h4_ l = *reinterpret_cast<const h4_*>(lo + r * ld + c);
      a.x += (float)h[0] + (float)l[0]; a.y += (float)h[1] + (float)l[1];
      a.z += (float)h[2] + (float)l[2]; a.w += (float)h[3] + (float)l[3];
    }
  }
  red[rp][cl] = a;
  __syncthreads();
  if (rp == 0 && c < cols) {
    const float4 b = red[1][cl], d = red[2][cl], e = red[3][cl];
    float* o = out + c;
    unsafeAtomicAdd(o + 0, ((a.x + b.x) + (d.x + e.x)) * out_scale);
    unsafeAtomicAdd(o + 1, ((a.y + b.y) + (d.y + e.y)) * out_scale);
    unsafeAtomicAdd(o + 2, ((a.z + b.z) + (d.z + e.z)) * out_scale);
    unsafeAtomicAdd(o + 3, ((a.w + b.w) + (d.w + e.w)) * out_scale);
  }
}

// ---------------------------------------------------------------------------------------------------
// activations
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void act_kernel(const float* __restrict__ pre, const float* __restrict__ dy,
                                                  float* __restrict__ out, int64_t n, int act) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = pre[i];
  float r;
  if (dy) {   // backward
    const float d = dy[i];
    switch (act) {
      case PFPP_ACT_RELU: r = v > 0.0f ? d : 0.0f; break;
      case PFPP_ACT_SILU: r = d * silu_grad(v); break;
      case PFPP_ACT_GELU: r = d * gelu_grad(v); break;
      default: r = d; break;
    }
  } else {
    switch (act) {
      case PFPP_ACT_RELU: r = v > 0.0f ? v : 0.0f; break;
      case PFPP_ACT_SILU: r = silu_f(v); break;
      case PFPP_ACT_GELU: r = gelu_f(v); break;
      default: r = v; break;
    }
  }
  out[i] = r;
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm backward: one workgroup per group of rows that share (scale, shift); wave w takes rows
// w, w+4, ...; a lane owns the same 4*VPL columns in every row, so the column sums of the group stay
// in registers until the end (LDS across the 4 waves, then one atomic per column).
// ---------------------------------------------------------------------------------------------------
template <int VPL>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mod, int64_t ld_mod,
    const float* __restrict__ gamma, const int32_t* __restrict__ group_batch, int group_rows, int rows_per_batch,
    float* __restrict__ dx, float* __restrict__ dmult, float* __restrict__ dadd, int64_t ld_d, int64_t rows,
    float eps, float* __restrict__ drop_out, uint32_t thresh, float inv_keep, uint64_t seed, uint32_t site,
    int do_drop, pfpp_planes_out po_ret, pfpp_planes_out po_dx) {
  pfpp_chain_prio();
  // po_ret: planes of the value the backward chain continues with (the dropped-out gradient when do_drop, else the updated
  // dx); po_dx: planes of the updated dx.  drop_out may be null when only the planes of the dropped-out gradient are wanted.
  constexpr int C = 256 * VPL;
  __shared__ float red[2][4][C];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // gridDim.y workgroups share a group: each takes a slice of its rows and adds its column sums with the same atomics
  // (a group of 25 rows on one workgroup is 7 dependent row passes per wave and 154 workgroups on 256 CUs)
  const int64_t g_begin = (int64_t)blockIdx.x * group_rows;
  const int per = (group_rows + (int)gridDim.y - 1) / (int)gridDim.y;
  const int64_t r_begin = g_begin + (int64_t)blockIdx.y * per;
  const int64_t r_end = min(min(rows, g_begin + group_rows), r_begin + per);
  if (r_begin >= r_end) return;
  const int64_t b = group_batch ? (int64_t)group_batch[blockIdx.x] : (mod ? g_begin / rows_per_batch : 0);

  // a wave's rows are 4 apart; the next row's three inputs are requested before this row's four reductions (unconditionally, clamped
  // to the wave's last row: a conditional load makes the compiler wait for it on the spot)
  float4 nv[VPL], nd[VPL], no[VPL];
  auto fetch = [&](int64_t row) {
    const float4* xr = reinterpret_cast<const float4*>(x + row * C);
    const float4* dr = reinterpret_cast<const float4*>(dy + row * C);
    const float4* dxr = reinterpret_cast<const float4*>(dx + row * C);
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      nv[k] = xr[lane + 64 * k];
      nd[k] = dr[lane + 64 * k];
      no[k] = dxr[lane + 64 * k];          // the running gradient this row is added to
    }
  };
  fetch(r_begin + wave < r_end ? r_begin + wave : r_begin);       // first, so that it is in flight under the multiplier loads below

  // multipliers: ONE branch-free load sequence (per-k branches on mod / gamma serialise the loads, each with its own wait)
  float4 mult[VPL], accA[VPL], accB[VPL];
  {
    const float* mp = mod ? mod + b * ld_mod : gamma;
    const float one = mod ? 1.0f : 0.0f;
    float4 raw[VPL];
#pragma unroll
    for (int k = 0; k < VPL; ++k) raw[k] = mp ? reinterpret_cast<const float4*>(mp)[lane + 64 * k] : make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      mult[k] = make_float4(one + raw[k].x, one + raw[k].y, one + raw[k].z, one + raw[k].w);
      accA[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      accB[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  for (int64_t row = r_begin + wave; row < r_end; row += 4) {
    float4 v[VPL], d[VPL], o0[VPL];
    float4* dxr = reinterpret_cast<float4*>(dx + row * C);
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      v[k] = nv[k]; d[k] = nd[k]; o0[k] = no[k];
      s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    fetch(row + 4 < r_end ? row + 4 : row);
    const float mean = wave_sum(s) / (float)C;
    float q = 0.0f;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      v[k].x -= mean; v[k].y -= mean; v[k].z -= mean; v[k].w -= mean;
      q += (v[k].x * v[k].x + v[k].y * v[k].y) + (v[k].z * v[k].z + v[k].w * v[k].w);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    float s1 = 0.0f, s2 = 0.0f;
    float4 g[VPL];
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      v[k].x *= rstd; v[k].y *= rstd; v[k].z *= rstd; v[k].w *= rstd;          // xhat
      g[k] = make_float4(d[k].x * mult[k].x, d[k].y * mult[k].y, d[k].z * mult[k].z, d[k].w * mult[k].w);
      s1 += (g[k].x + g[k].y) + (g[k].z + g[k].w);
      s2 += (g[k].x * v[k].x + g[k].y * v[k].y) + (g[k].z * v[k].z + g[k].w * v[k].w);
      accA[k].x += d[k].x * v[k].x; accA[k].y += d[k].y * v[k].y;
      accA[k].z += d[k].z * v[k].z; accA[k].w += d[k].w * v[k].w;
      accB[k].x += d[k].x; accB[k].y += d[k].y; accB[k].z += d[k].z; accB[k].w += d[k].w;
    }
    const float c1 = wave_sum(s1) / (float)C;
    const float c2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      float4 o = o0[k];
      o.x += rstd * (g[k].x - c1 - v[k].x * c2);
      o.y += rstd * (g[k].y - c1 - v[k].y * c2);
      o.z += rstd * (g[k].z - c1 - v[k].z * c2);
      o.w += rstd * (g[k].w - c1 - v[k].w * c2);
      dxr[lane + 64 * k] = o;
      const uint64_t i4 = (uint64_t)row * C + (uint64_t)(lane + 64 * k) * 4;
      if (po_dx.hi) pfpp_store4_planes(po_dx, (int64_t)i4, o);
      if (do_drop) {       // the dropout that follows in the backward chain (same mask as pfpp_dropout over [rows, C])
        float4 dd;
        dd.x = pfpp_rng_u32(seed, site, i4 + 0) >= thresh ? o.x * inv_keep : 0.0f;
        dd.y = pfpp_rng_u32(seed, site, i4 + 1) >= thresh ? o.y * inv_keep : 0.0f;
        dd.z = pfpp_rng_u32(seed, site, i4 + 2) >= thresh ? o.z * inv_keep : 0.0f;
        dd.w = pfpp_rng_u32(seed, site, i4 + 3) >= thresh ? o.w * inv_keep : 0.0f;
        if (drop_out) reinterpret_cast<float4*>(drop_out + row * C)[lane + 64 * k] = dd;
        if (po_ret.hi) pfpp_store4_planes(po_ret, (int64_t)i4, dd);
      } else if (po_ret.hi) {
        pfpp_store4_planes(po_ret, (int64_t)i4, o);
      }
    }
  }
  if (!dmult) return;      // uniform
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    const int c = (lane + 64 * k) * 4;
    *reinterpret_cast<float4*>(&red[0][wave][c]) = accA[k];
    *reinterpret_cast<float4*>(&red[1][wave][c]) = accB[k];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const float a = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
    const float bb = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
    unsafeAtomicAdd(dmult + b * ld_d + c, a);
    unsafeAtomicAdd(dadd + b * ld_d + c, bb);
  }
}


// ---------------------------------------------------------------------------------------------------
// forward: h = (res or 0) + dropout(y), n = LayerNorm(h) in one pass (one wave per row, the arithmetic of
// pfpp_dropout followed by pfpp_layernorm, one read of y/res instead of three passes)
// ---------------------------------------------------------------------------------------------------
template <int VPL>
__global__ __launch_bounds__(256) void dropout_layernorm_kernel(
    const float* __restrict__ y, const float* __restrict__ res, float* __restrict__ h_out, float* __restrict__ n_out,
    const float* __restrict__ mod, int64_t ld_mod, const float* __restrict__ gamma, const float* __restrict__ beta,
    int64_t rows, int rows_per_batch, float eps, const int32_t* __restrict__ group_batch, int group_rows,
    uint32_t thresh, float inv_keep, uint64_t seed, uint32_t site, pfpp_planes_out po) {
  pfpp_chain_prio();
  constexpr int C = 256 * VPL;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float4 v[VPL];
  // the row's inputs first (all of them before any arithmetic): the batch index / modulation rows are a dependent chain of their own
  float4 ya[VPL], ra[VPL];
  {
    const float* rp = res ? res : y;        // uniform; without a residual the second load is a repeat that is not used
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      ya[k] = reinterpret_cast<const float4*>(y + row * C)[lane + 64 * k];
      ra[k] = reinterpret_cast<const float4*>(rp + row * C)[lane + 64 * k];
    }
  }
  // the (scale, shift) / (gamma, beta) rows do not depend on the reductions: fetched with the inputs, in one branch-free sequence
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
  for (int k = 0; k < VPL; ++k) {
    const int c4 = lane + 64 * k;
    const float4 a = ya[k];
    float4 r = res ? ra[k] : make_float4(0.f, 0.f, 0.f, 0.f);
    const uint64_t i4 = (uint64_t)row * C + (uint64_t)c4 * 4;
    r.x += pfpp_rng_u32(seed, site, i4 + 0) >= thresh ? a.x * inv_keep : 0.0f;
    r.y += pfpp_rng_u32(seed, site, i4 + 1) >= thresh ? a.y * inv_keep : 0.0f;
    r.z += pfpp_rng_u32(seed, site, i4 + 2) >= thresh ? a.z * inv_keep : 0.0f;
    r.w += pfpp_rng_u32(seed, site, i4 + 3) >= thresh ? a.w * inv_keep : 0.0f;
    reinterpret_cast<float4*>(h_out + row * C)[c4] = r;
    v[k] = r;
    s += (r.x + r.y) + (r.z + r.w);
  }
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
    if (n_out) reinterpret_cast<float4*>(n_out + row * C)[c4] = o;
    if (po.hi) pfpp_store4_planes(po, row * C + (int64_t)c4 * 4, o);
  }
}

// ---------------------------------------------------------------------------------------------------
// pooling / token / embedding backward
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mean_pool_bwd_kernel(const float* __restrict__ dp, float* __restrict__ dx,
                                                            int64_t n, int L, int C) {
  const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;    // into dx [n*L, C]
  if (i4 >= n * L * C) return;
  const int64_t row = i4 / C;
  const int c = (int)(i4 - row * C);
  const int64_t f = row / L;
  float4 v = *reinterpret_cast<const float4*>(dp + f * C + c);
  const float inv = 1.0f / (float)L;
  v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
  *reinterpret_cast<float4*>(dx + i4) = v;
}

// block per fragment: dx_emb[f, c] = sum_l dtok[(f,l), c]; dref[ref[f], c] += that
__global__ __launch_bounds__(128) void token_combine_bwd_kernel(const float* __restrict__ dtok,
                                                                const uint8_t* __restrict__ ref,
                                                                float* __restrict__ dx_emb, float* __restrict__ dref,
                                                                int L, int C, const int32_t* __restrict__ slot) {
  const int64_t f = blockIdx.x;
  const int r = ref[slot ? slot[f] : f] ? 1 : 0;
  for (int c = threadIdx.x * 4; c < C; c += 128 * 4) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l = 0; l < L; ++l) {
      const float4 v = *reinterpret_cast<const float4*>(dtok + (f * L + l) * C + c);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    *reinterpret_cast<float4*>(dx_emb + f * C + c) = a;
    float* d = dref + (int64_t)r * C + c;
    unsafeAtomicAdd(d + 0, a.x); unsafeAtomicAdd(d + 1, a.y);
    unsafeAtomicAdd(d + 2, a.z); unsafeAtomicAdd(d + 3, a.w);
  }
}

// dtables[tab, t[b], :] += dse[tab, b, :] * silu'(tables[tab, t[b], :]), puzzles that drew the same timestep summed IN INDEX ORDER by the
// one thread that owns the row's first occurrence: no atomics, so the result is a function of the inputs only — data-parallel ranks
// that scatter the same gathered (rows, timesteps) list (GradExchange.gather_rows) end up with bit-identical table gradients.
// One workgroup per (table, b); the scan over t[] is wave-uniform (scalar loads).
__global__ __launch_bounds__(128) void silu_embed_bwd_kernel(const float* __restrict__ tables,
                                                             const int64_t* __restrict__ t,
                                                             const float* __restrict__ dse, float* __restrict__ dtables,
                                                             int64_t n_emb, int64_t B, int C, uint32_t* __restrict__ active) {
  const int64_t b = blockIdx.x, tab = blockIdx.y;
  const int64_t tb = t[b];
  for (int64_t j = 0; j < b; ++j)
    if (t[j] == tb) return;                            // an earlier puzzle owns this row
  // rows that ever received a gradient (pfpp_adamw_rows_active): a monotone bitmap, so the order of the atomics does not matter
  if (active && tab == 0 && threadIdx.x == 0) atomicOr(&active[tb >> 5], 1u << (tb & 31));
  const int64_t row = (tab * n_emb + tb) * C;
  for (int c = threadIdx.x; c < C; c += 128) {
    const float sg = silu_grad(tables[row + c]);
    float acc = dtables[row + c];
    for (int64_t j = b; j < B; ++j)
      if (t[j] == tb) acc += dse[(tab * B + j) * C + c] * sg;
    dtables[row + c] = acc;
  }
}

// ---------------------------------------------------------------------------------------------------
// loss
// ---------------------------------------------------------------------------------------------------
// sel[r] != 0 selects row r; with `valid` given instead, a row counts iff valid[r] != 0 and ref[r] == 0 (Denoiser._loss's
// part_valids & ~ref_part, denoiser.py:118-126, without materialising the mask).  amax (optional): max |dpred| of this call, for the
// gradient-scale tracking of the training engine (replaces an abs + max reduction over dpred)
__global__ __launch_bounds__(1024) void mse_loss_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                        const uint8_t* __restrict__ sel, float* __restrict__ loss,
                                                        float* __restrict__ dpred, int64_t n, int width, float grad_out,
                                                        const float* __restrict__ valid, const uint8_t* __restrict__ ref,
                                                        float* __restrict__ amax) {
  __shared__ float s_sum[16];
  __shared__ float s_cnt[16];
  __shared__ float s_tot[2];
  float sum = 0.0f, cnt = 0.0f;
  auto picked = [&](int64_t r) { return valid ? (valid[r] != 0.0f && ref[r] == 0) : (sel[r] != 0); };
  for (int64_t r = threadIdx.x; r < n; r += 1024) {
    if (picked(r)) {
      cnt += 1.0f;
      for (int c = 0; c < width; ++c) {
        const float d = pred[r * width + c] - target[r * width + c];
        sum += d * d;
      }
    }
  }
  sum = wave_sum(sum);
  cnt = wave_sum(cnt);
  if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = sum; s_cnt[threadIdx.x >> 6] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.0f, b = 0.0f;
    for (int w = 0; w < 16; ++w) { a += s_sum[w]; b += s_cnt[w]; }
    s_tot[0] = a; s_tot[1] = b;
    loss[0] = a / (b * (float)width);       // 0/0 = NaN like torch's mean over an empty selection
  }
  __syncthreads();
  if (!dpred) return;
  const float k = grad_out * 2.0f / (s_tot[1] * (float)width);
  float mx = 0.0f;
  for (int64_t i = threadIdx.x; i < n * width; i += 1024) {
    const int64_t r = i / width;
    const float d = picked(r) ? k * (pred[i] - target[i]) : 0.0f;
    dpred[i] = d;
    mx = fmaxf(mx, fabsf(d));          // fmaxf drops a NaN operand, like torch.max would not: NaN gradients surface in the overflow guard
  }
  if (amax) {
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    __syncthreads();                                   // s_sum is free again
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
      float m = 0.0f;
      for (int w = 0; w < 16; ++w) m = fmaxf(m, s_sum[w]);
      amax[0] = m;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// AdamW
// ---------------------------------------------------------------------------------------------------
// GUARD: the loss-scaling guard of a mixed-precision optimizer, on the device (no host read): `overflow[0]` is the step's
// "a non-finite gradient was seen" flag, `overflow[1]` counts such elements.  An element whose own gradient is inf / NaN is left
// untouched (gradient still cleared when asked) and raises the flag — parameters and both moments can never be poisoned by an
// overflowed split-f16 gradient plane.  The decision is PER ELEMENT and depends on that element's gradient only: the flag is
// written, never read, by the kernel, so the outcome does not depend on workgroup scheduling and data-parallel replicas (which
// hold bit-identical all-reduced gradients) skip the same elements and stay equal.
template <bool GUARD>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    _Float16* __restrict__ hi, _Float16* __restrict__ lo, int64_t n,
                                                    float decay, float w1, float beta2, float w2, float eps,
                                                    float step_size, float inv_sqrt_bc2, float g_scale, int zero_g,
                                                    int* __restrict__ overflow) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool in = i < n;
  float gr = in ? g[i] * g_scale : 0.0f;
  if (in && zero_g) g[i] = 0.0f;                      // optimizer.zero_grad() in the same pass (no separate 230 MB memset)
  if (GUARD) {
    const bool bad = !(fabsf(gr) <= 3.0e38f);         // inf or NaN
    const unsigned long long mask = __ballot(bad);
    if (mask != 0ull && (threadIdx.x & 63) == 0) {
      atomicOr(overflow, 1);
      atomicAdd(overflow + 1, __popcll(mask));
    }
    if (bad) return;
  }
  if (!in) return;
  float pp = p[i] * decay;
  float mm = m[i];
  mm = mm + w1 * (gr - mm);                         // exp_avg.lerp_(grad, 1 - beta1)
  const float vv = v[i] * beta2 + w2 * (gr * gr);   // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
  const float denom = sqrtf(vv) * inv_sqrt_bc2 + eps;
  pp = pp - step_size * (mm / denom);
  p[i] = pp; m[i] = mm; v[i] = vv;
  if (hi) {
    const pfpp_hl s = pfpp_split(pp);
    hi[i] = s.hi; lo[i] = s.lo;
  }
}

// AdamW over a stack of embedding tables restricted by rows (see pfpp_adamw_rows): MODE 0 = one thread per element of the stack, rows
// listed in t skipped (a 3,072-bit row mask built in LDS per workgroup); MODE 1 = one thread per element of the listed rows, a row
// listed twice is taken by its first occurrence only; MODE 2 = one thread per element of the stack, rows whose bit is CLEAR in the bitmap
// `t` points at (reinterpreted: uint32 words) skipped.  Same arithmetic per element as adamw_kernel<true>.
template <int MODE, bool GUARD>
__global__ __launch_bounds__(256) void adamw_rows_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                                         int64_t n_tables, int rows_per_table, int C, const int64_t* __restrict__ t,
                                                         int n_t, float decay, float w1, float beta2, float w2, float eps,
                                                         float step_size, float inv_sqrt_bc2, float g_scale, int zero_g,
                                                         int* __restrict__ overflow) {
  int64_t i;
  if (MODE == 2) {
    __shared__ unsigned act[128];
    const uint32_t* bits = reinterpret_cast<const uint32_t*>(t);
    for (int k = threadIdx.x; k < 128; k += 256) act[k] = k < (rows_per_table + 31) / 32 ? bits[k] : 0u;
    __syncthreads();
    i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_tables * rows_per_table * (int64_t)C) return;
    const int r = (int)((i / C) % rows_per_table);
    if (!((act[r >> 5] >> (r & 31)) & 1u)) return;
  } else if (MODE == 0) {
    __shared__ unsigned mask[128];              // rows_per_table <= 4096
    for (int k = threadIdx.x; k < 128; k += 256) mask[k] = 0u;
    __syncthreads();
    for (int k = threadIdx.x; k < n_t; k += 256) {
      const int64_t r = t[k];
      if (r >= 0 && r < rows_per_table) atomicOr(&mask[r >> 5], 1u << (r & 31));
    }
    __syncthreads();
    i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_tables * rows_per_table * (int64_t)C) return;
    const int r = (int)((i / C) % rows_per_table);
    if ((mask[r >> 5] >> (r & 31)) & 1u) return;
  } else {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (table, listed entry, column)
    if (e >= n_tables * n_t * (int64_t)C) return;
    const int c = (int)(e % C);
    const int k = (int)((e / C) % n_t);
    const int64_t tab = e / ((int64_t)C * n_t);
    const int64_t r = t[k];
    if (r < 0 || r >= rows_per_table) return;
    for (int k2 = 0; k2 < k; ++k2)
      if (t[k2] == r) return;                   // a duplicate: its first occurrence does the row
    i = (tab * rows_per_table + r) * C + c;
  }
  float gr = g[i] * g_scale;
  if (zero_g) g[i] = 0.0f;
  if (GUARD) {
    if (!(fabsf(gr) <= 3.0e38f)) {              // inf or NaN: the element is left untouched (adamw_kernel<true>'s rule)
      atomicOr(overflow, 1);
      atomicAdd(overflow + 1, 1);
      return;
    }
  }
  float pp = p[i] * decay;
  float mm = m[i];
  mm = mm + w1 * (gr - mm);
  const float vv = v[i] * beta2 + w2 * (gr * gr);
  const float denom = sqrtf(vv) * inv_sqrt_bc2 + eps;
  pp = pp - step_size * (mm / denom);
  p[i] = pp; m[i] = mm; v[i] = vv;
  if (hi) {
    const pfpp_hl s = pfpp_split(pp);
    hi[i] = s.hi; lo[i] = s.lo;
  }
}

}  // namespace

// =====================================================================================================
extern "C" int pfpp_colsum(const float* x, float* out, int64_t rows, int64_t cols, int64_t ld, int64_t batch,
                           int64_t sx, int64_t so, int accumulate, pfpp_stream_t stream) {
  PFPP_REQUIRE(x && out, "null pointer");
  PFPP_REQUIRE(rows >= 0 && cols > 0 && ld >= cols && batch >= 1, "bad sizes");
  hipStream_t st = pfpp::as_stream(stream);
  if (!accumulate) {
    for (int64_t z = 0; z < batch; ++z)
      if (hipMemsetAsync(out + z * so, 0, (size_t)cols * sizeof(float), st) != hipSuccess) return pfpp::check_launch(__func__);
  }
  if (rows == 0) return PFPP_OK;
  const bool vec = cols % 4 == 0 && ld % 4 == 0 && sx % 4 == 0 && pfpp::aligned16(x);
  if (vec) {
    const int rpb = 64;      // row blocks of 64: enough workgroups for [3850 x 512] inputs; the tail is 4 atomics per column
    const dim3 grid(blocks_for(cols, 256), blocks_for(rows, rpb), (unsigned)batch);
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, st, x, out, rows, (int)cols, ld, rpb, sx, so);
  } else {
    hipLaunchKernelGGL(colsum_narrow_kernel, dim3((unsigned)cols, 1, (unsigned)batch), dim3(64), 0, st, x, out, rows, ld, sx, so);
  }
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_dropout(const float* x, const float* res, float* out, int64_t n, float p, uint64_t seed,
                            uint32_t site, pfpp_stream_t stream) {
  PFPP_REQUIRE(x && out, "null pointer");
  PFPP_REQUIRE(p >= 0.0f && p < 1.0f, "p outside [0, 1)");
  PFPP_REQUIRE(pfpp::aligned16(x) && pfpp::aligned16(out) && pfpp::aligned16(res), "16-byte alignment");
  if (n == 0) return PFPP_OK;
  hipLaunchKernelGGL(dropout_kernel, dim3(blocks_for(n, 1024)), dim3(256), 0, pfpp::as_stream(stream), x, res, out, n,
                     pfpp_drop_thresh(p), 1.0f / (1.0f - p), seed, site);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_dropout_mask(uint8_t* keep, int64_t n, float p, uint64_t seed, uint32_t site,
                                 pfpp_stream_t stream) {
  PFPP_REQUIRE(keep, "null pointer");
  PFPP_REQUIRE(p >= 0.0f && p < 1.0f, "p outside [0, 1)");
  if (n == 0) return PFPP_OK;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, pfpp::as_stream(stream), keep, n,
                     pfpp_drop_thresh(p), seed, site);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_geglu_p(const float* z, float* u, int64_t rows, int64_t inner, float p, uint64_t seed,
                            uint32_t site, const pfpp_planes* u_planes, pfpp_stream_t stream) {
  PFPP_REQUIRE(z && (u || u_planes), "null pointer");
  PFPP_REQUIRE(inner > 0 && inner % 4 == 0 && p >= 0.0f && p < 1.0f, "inner % 4 != 0 or p outside [0, 1)");
  PFPP_REQUIRE(pfpp::aligned16(z) && pfpp::aligned16(u) && pfpp_planes_ok(u_planes), "alignment");
  if (rows == 0) return PFPP_OK;
  hipLaunchKernelGGL(geglu_kernel, dim3(blocks_for(rows * inner, 1024)), dim3(256), 0, pfpp::as_stream(stream), z, u,
                     rows, (int)inner, pfpp_drop_thresh(p), 1.0f / (1.0f - p), seed, site, pfpp_planes_arg(u_planes));
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_geglu(const float* z, float* u, int64_t rows, int64_t inner, float p, uint64_t seed,
                          uint32_t site, pfpp_stream_t stream) {
  PFPP_REQUIRE(u, "null pointer");
  return pfpp_geglu_p(z, u, rows, inner, p, seed, site, nullptr, stream);
}

extern "C" int pfpp_geglu_bwd_p(const float* z, const float* du, float* dz, int64_t rows, int64_t inner, float p,
                                uint64_t seed, uint32_t site, const pfpp_planes* dz_planes, pfpp_stream_t stream) {
  PFPP_REQUIRE(z && du && (dz || dz_planes), "null pointer");
  PFPP_REQUIRE(inner > 0 && inner % 4 == 0 && p >= 0.0f && p < 1.0f, "inner % 4 != 0 or p outside [0, 1)");
  PFPP_REQUIRE(pfpp::aligned16(z) && pfpp::aligned16(du) && pfpp::aligned16(dz) && pfpp_planes_ok(dz_planes), "alignment");
  if (rows == 0) return PFPP_OK;
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(blocks_for(rows * inner, 1024)), dim3(256), 0, pfpp::as_stream(stream), z,
                     du, dz, rows, (int)inner, pfpp_drop_thresh(p), 1.0f / (1.0f - p), seed, site, pfpp_planes_arg(dz_planes));
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_geglu_bwd(const float* z, const float* du, float* dz, int64_t rows, int64_t inner, float p,
                              uint64_t seed, uint32_t site, pfpp_stream_t stream) {
  PFPP_REQUIRE(dz, "null pointer");
  return pfpp_geglu_bwd_p(z, du, dz, rows, inner, p, seed, site, nullptr, stream);
}

extern "C" int pfpp_split_planes(const float* x, int64_t n, const pfpp_planes* planes, pfpp_stream_t stream) {
  PFPP_REQUIRE(x && planes && pfpp_planes_ok(planes) && pfpp::aligned16(x), "null pointer / alignment");
  PFPP_REQUIRE(n >= 0 && n % 4 == 0, "n % 4 != 0");
  if (n == 0) return PFPP_OK;
  hipLaunchKernelGGL(split_planes_kernel, dim3(blocks_for(n, 1024)), dim3(256), 0, pfpp::as_stream(stream), x, n,
                     pfpp_planes_arg(planes));
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_colsum_planes(const void* hi, const void* lo, float* out, int64_t rows, int64_t cols, int64_t ld,
                                  float out_scale, pfpp_stream_t stream) {
  PFPP_REQUIRE(hi && lo && out, "null pointer");
  PFPP_REQUIRE(cols > 0 && cols % 4 == 0 && ld % 4 == 0 && ld >= cols, "cols / ld must be multiples of 4");
  if (rows == 0) return PFPP_OK;
  // enough row chunks for ~2 workgroups per CU (3850 x 512 at 128 rows per block was 62 workgroups: 14.6 us for 8 MB)
  const int64_t col_blocks = (cols + 255) / 256;
  int rpb = (int)((rows * col_blocks + 511) / 512);
  rpb = rpb < 16 ? 16 : (rpb > 128 ? 128 : (rpb + 3) / 4 * 4);
  const dim3 grid(blocks_for(cols, 256), blocks_for(rows, rpb));
  hipLaunchKernelGGL(colsum_planes_kernel, grid, dim3(256), 0, pfpp::as_stream(stream), (const _Float16*)hi, (const _Float16*)lo,
                     out, rows, (int)cols, ld, rpb, out_scale);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_act(const float* pre, float* out, int64_t n, int act, pfpp_stream_t stream) {
  PFPP_REQUIRE(pre && out, "null pointer");
  PFPP_REQUIRE(act >= PFPP_ACT_NONE && act <= PFPP_ACT_GELU, "unknown activation");
  if (n == 0) return PFPP_OK;
  hipLaunchKernelGGL(act_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, pfpp::as_stream(stream), pre,
                     (const float*)nullptr, out, n, act);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_act_bwd(const float* pre, const float* dy, float* dx, int64_t n, int act, pfpp_stream_t stream) {
  PFPP_REQUIRE(pre && dy && dx, "null pointer");
  PFPP_REQUIRE(act >= PFPP_ACT_NONE && act <= PFPP_ACT_GELU, "unknown activation");
  if (n == 0) return PFPP_OK;
  hipLaunchKernelGGL(act_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, pfpp::as_stream(stream), pre, dy, dx, n, act);
  return pfpp::check_launch(__func__);
}

namespace {
int layernorm_bwd_impl(const float* x, const float* dy, const float* mod, int64_t ld_mod,
                       const float* gamma, const int32_t* group_batch, int64_t group_rows,
                       int64_t rows_per_batch, float* dx, float* dmult, float* dadd, int64_t ld_d,
                       int64_t rows, int64_t C, float eps, float* drop_out, float p, uint64_t seed, uint32_t site,
                       pfpp_stream_t stream, int do_drop = -1, const pfpp_planes* ret_planes = nullptr,
                       const pfpp_planes* dx_planes = nullptr) {
  PFPP_REQUIRE(x && dy && dx, "null pointer");
  PFPP_REQUIRE(pfpp_planes_ok(ret_planes) && pfpp_planes_ok(dx_planes), "planes: null / misaligned");
  if (do_drop < 0) do_drop = drop_out != nullptr;
  const pfpp_planes_out po_ret = pfpp_planes_arg(ret_planes), po_dx = pfpp_planes_arg(dx_planes);
  PFPP_REQUIRE(!(mod && gamma), "mod and gamma are exclusive");
  PFPP_REQUIRE(!dmult == !dadd, "dmult and dadd go together");
  PFPP_SUPPORTED(C == 256 || C == 512, "C not in {256, 512}");
  PFPP_REQUIRE(group_rows >= 1 && rows_per_batch >= 1, "bad group sizes");
  PFPP_REQUIRE(group_batch || !mod || rows_per_batch % group_rows == 0, "rows_per_batch % group_rows != 0");
  PFPP_REQUIRE(pfpp::aligned16(x) && pfpp::aligned16(dy) && pfpp::aligned16(dx) && pfpp::aligned16(mod) &&
               pfpp::aligned16(gamma) && pfpp::aligned16(drop_out) && ld_mod % 4 == 0, "16-byte alignment");
  PFPP_REQUIRE(p >= 0.0f && p < 1.0f, "p outside [0, 1)");
  if (rows == 0) return PFPP_OK;
  static const int rows_per_wg = getenv("PFPP_LN_BWD_ROWS") ? atoi(getenv("PFPP_LN_BWD_ROWS")) : 8;
  const int split = rows_per_wg > 0 ? (int)((group_rows + rows_per_wg - 1) / rows_per_wg) : 1;
  const dim3 grid(blocks_for(rows, (int)group_rows), (unsigned)(split < 1 ? 1 : split));
  hipStream_t st = pfpp::as_stream(stream);
  const uint32_t thresh = pfpp_drop_thresh(p);
  const float inv_keep = 1.0f / (1.0f - p);
  if (C == 256)
    hipLaunchKernelGGL(layernorm_bwd_kernel<1>, grid, dim3(256), 0, st, x, dy, mod, ld_mod, gamma, group_batch,
                       (int)group_rows, (int)rows_per_batch, dx, dmult, dadd, ld_d, rows, eps, drop_out, thresh, inv_keep,
                       seed, site, do_drop, po_ret, po_dx);
  else
    hipLaunchKernelGGL(layernorm_bwd_kernel<2>, grid, dim3(256), 0, st, x, dy, mod, ld_mod, gamma, group_batch,
                       (int)group_rows, (int)rows_per_batch, dx, dmult, dadd, ld_d, rows, eps, drop_out, thresh, inv_keep,
                       seed, site, do_drop, po_ret, po_dx);
  return pfpp::check_launch("pfpp_layernorm_bwd");
}
}  // namespace

extern "C" int pfpp_layernorm_bwd(const float* x, const float* dy, const float* mod, int64_t ld_mod,
                                  const float* gamma, const int32_t* group_batch, int64_t group_rows,
                                  int64_t rows_per_batch, float* dx, float* dmult, float* dadd, int64_t ld_d,
                                  int64_t rows, int64_t C, float eps, pfpp_stream_t stream) {
  return layernorm_bwd_impl(x, dy, mod, ld_mod, gamma, group_batch, group_rows, rows_per_batch, dx, dmult, dadd, ld_d, rows, C,
                            eps, nullptr, 0.0f, 0, 0, stream);
}

extern "C" int pfpp_layernorm_bwd_dropout(const float* x, const float* dy, const float* mod, int64_t ld_mod,
                                          const float* gamma, const int32_t* group_batch, int64_t group_rows,
                                          int64_t rows_per_batch, float* dx, float* dmult, float* dadd, int64_t ld_d,
                                          int64_t rows, int64_t C, float eps, float* drop_out, float p, uint64_t seed,
                                          uint32_t site, pfpp_stream_t stream) {
  PFPP_REQUIRE(drop_out, "null pointer");
  return layernorm_bwd_impl(x, dy, mod, ld_mod, gamma, group_batch, group_rows, rows_per_batch, dx, dmult, dadd, ld_d, rows, C,
                            eps, drop_out, p, seed, site, stream);
}

extern "C" int pfpp_layernorm_bwd_p(const float* x, const float* dy, const float* mod, int64_t ld_mod,
                                    const float* gamma, const int32_t* group_batch, int64_t group_rows,
                                    int64_t rows_per_batch, float* dx, float* dmult, float* dadd, int64_t ld_d,
                                    int64_t rows, int64_t C, float eps, float* drop_out, float p, uint64_t seed,
                                    uint32_t site, int32_t dropout, const pfpp_planes* ret_planes,
                                    const pfpp_planes* dx_planes, pfpp_stream_t stream) {
  PFPP_REQUIRE(!dropout || drop_out || ret_planes, "dropout without an output for the dropped-out gradient");
  return layernorm_bwd_impl(x, dy, mod, ld_mod, gamma, group_batch, group_rows, rows_per_batch, dx, dmult, dadd, ld_d, rows, C,
                            eps, drop_out, dropout ? p : 0.0f, seed, site, stream, dropout ? 1 : 0, ret_planes, dx_planes);
}

static int dropout_layernorm_impl(const float* y, const float* res, float* h_out, float* n_out, const float* mod,
                                  int64_t ld_mod, const float* gamma, const float* beta, const int32_t* group_batch,
                                  int64_t group_rows, int64_t rows_per_batch, int64_t rows, int64_t C, float eps, float p,
                                  uint64_t seed, uint32_t site, const pfpp_planes* n_planes, pfpp_stream_t stream);

extern "C" int pfpp_dropout_layernorm(const float* y, const float* res, float* h_out, float* n_out, const float* mod,
                                      int64_t ld_mod, const float* gamma, const float* beta,
                                      const int32_t* group_batch, int64_t group_rows, int64_t rows_per_batch,
                                      int64_t rows, int64_t C, float eps, float p, uint64_t seed, uint32_t site,
                                      pfpp_stream_t stream) {
  PFPP_REQUIRE(n_out, "null pointer");
  return dropout_layernorm_impl(y, res, h_out, n_out, mod, ld_mod, gamma, beta, group_batch, group_rows, rows_per_batch, rows, C,
                                eps, p, seed, site, nullptr, stream);
}

extern "C" int pfpp_dropout_layernorm_p(const float* y, const float* res, float* h_out, float* n_out, const float* mod,
                                        int64_t ld_mod, const float* gamma, const float* beta,
                                        const int32_t* group_batch, int64_t group_rows, int64_t rows_per_batch,
                                        int64_t rows, int64_t C, float eps, float p, uint64_t seed, uint32_t site,
                                        const pfpp_planes* n_planes, pfpp_stream_t stream) {
  PFPP_REQUIRE(n_out || n_planes, "null pointer");
  return dropout_layernorm_impl(y, res, h_out, n_out, mod, ld_mod, gamma, beta, group_batch, group_rows, rows_per_batch, rows, C,
                                eps, p, seed, site, n_planes, stream);
}

static int dropout_layernorm_impl(const float* y, const float* res, float* h_out, float* n_out, const float* mod,
                                  int64_t ld_mod, const float* gamma, const float* beta, const int32_t* group_batch,
                                  int64_t group_rows, int64_t rows_per_batch, int64_t rows, int64_t C, float eps, float p,
                                  uint64_t seed, uint32_t site, const pfpp_planes* n_planes, pfpp_stream_t stream) {
  PFPP_REQUIRE(y && h_out && pfpp_planes_ok(n_planes), "null pointer");
  const pfpp_planes_out po = pfpp_planes_arg(n_planes);
  PFPP_REQUIRE(!(mod && gamma) && (!gamma == !beta), "mod and gamma/beta are exclusive; gamma and beta go together");
  PFPP_SUPPORTED(C == 256 || C == 512, "C not in {256, 512}");
  PFPP_REQUIRE(group_rows >= 1 && rows_per_batch >= 1, "bad group sizes");
  PFPP_REQUIRE(p >= 0.0f && p < 1.0f, "p outside [0, 1)");
  PFPP_REQUIRE(pfpp::aligned16(y) && pfpp::aligned16(res) && pfpp::aligned16(h_out) && pfpp::aligned16(n_out) &&
               pfpp::aligned16(mod) && pfpp::aligned16(gamma) && pfpp::aligned16(beta) && ld_mod % 4 == 0, "16-byte alignment");
  if (rows == 0) return PFPP_OK;
  const dim3 grid(blocks_for(rows, 4));
  hipStream_t st = pfpp::as_stream(stream);
  const uint32_t thresh = pfpp_drop_thresh(p);
  const float inv_keep = 1.0f / (1.0f - p);
  if (C == 256)
    hipLaunchKernelGGL(dropout_layernorm_kernel<1>, grid, dim3(256), 0, st, y, res, h_out, n_out, mod, ld_mod, gamma, beta, rows,
                       (int)rows_per_batch, eps, group_batch, (int)group_rows, thresh, inv_keep, seed, site, po);
  else
    hipLaunchKernelGGL(dropout_layernorm_kernel<2>, grid, dim3(256), 0, st, y, res, h_out, n_out, mod, ld_mod, gamma, beta, rows,
                       (int)rows_per_batch, eps, group_batch, (int)group_rows, thresh, inv_keep, seed, site, po);
  return pfpp::check_launch("pfpp_dropout_layernorm");
}

extern "C" int pfpp_mean_pool_bwd(const float* dpooled, float* dx, int64_t n, int64_t L, int64_t C,
                                  pfpp_stream_t stream) {
  PFPP_REQUIRE(dpooled && dx, "null pointer");
  PFPP_REQUIRE(C % 4 == 0 && L >= 1 && pfpp::aligned16(dpooled) && pfpp::aligned16(dx), "C % 4 != 0 or alignment");
  if (n == 0) return PFPP_OK;
  hipLaunchKernelGGL(mean_pool_bwd_kernel, dim3(blocks_for(n * L * C, 1024)), dim3(256), 0, pfpp::as_stream(stream),
                     dpooled, dx, n, (int)L, (int)C);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_token_combine_bwd(const float* dtok, const uint8_t* ref_part, float* dx_emb, float* dref_emb,
                                      int64_t n, int64_t L, int64_t C, pfpp_stream_t stream) {
  PFPP_REQUIRE(dtok && ref_part && dx_emb && dref_emb, "null pointer");
  PFPP_REQUIRE(C % 4 == 0 && L >= 1 && pfpp::aligned16(dtok) && pfpp::aligned16(dx_emb), "C % 4 != 0 or alignment");
  if (n == 0) return PFPP_OK;
  hipLaunchKernelGGL(token_combine_bwd_kernel, dim3((unsigned)n), dim3(128), 0, pfpp::as_stream(stream), dtok, ref_part,
                     dx_emb, dref_emb, (int)L, (int)C, (const int32_t*)nullptr);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_token_combine_bwd_slots(const float* dtok, const uint8_t* ref_part, const int32_t* slot, float* dx_emb,
                                            float* dref_emb, int64_t n, int64_t L, int64_t C, pfpp_stream_t stream) {
  PFPP_REQUIRE(dtok && ref_part && slot && dx_emb && dref_emb, "null pointer");
  PFPP_REQUIRE(C % 4 == 0 && L >= 1 && pfpp::aligned16(dtok) && pfpp::aligned16(dx_emb), "C % 4 != 0 or alignment");
  if (n == 0) return PFPP_OK;
  hipLaunchKernelGGL(token_combine_bwd_kernel, dim3((unsigned)n), dim3(128), 0, pfpp::as_stream(stream), dtok, ref_part,
                     dx_emb, dref_emb, (int)L, (int)C, slot);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_silu_embed_bwd(const float* tables, const int64_t* t, const float* dse, float* dtables,
                                   int64_t n_tab, int64_t n_emb, int64_t B, int64_t C, pfpp_stream_t stream) {
  PFPP_REQUIRE(tables && t && dse && dtables, "null pointer");
  const int64_t total = n_tab * B * C;
  if (total == 0) return PFPP_OK;
  PFPP_SUPPORTED(B <= 0x7fffffff && n_tab <= 65535, "too many rows / tables for one launch");
  hipLaunchKernelGGL(silu_embed_bwd_kernel, dim3((unsigned)B, (unsigned)n_tab), dim3(128), 0, pfpp::as_stream(stream), tables,
                     t, dse, dtables, n_emb, B, (int)C, (uint32_t*)nullptr);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_silu_embed_bwd_mark(const float* tables, const int64_t* t, const float* dse, float* dtables,
                                        int64_t n_tab, int64_t n_emb, int64_t B, int64_t C, uint32_t* active, pfpp_stream_t stream) {
  PFPP_REQUIRE(tables && t && dse && dtables && active, "null pointer");
  const int64_t total = n_tab * B * C;
  if (total == 0) return PFPP_OK;
  PFPP_SUPPORTED(B <= 0x7fffffff && n_tab <= 65535 && n_emb <= 4096, "too many rows / tables for one launch");
  hipLaunchKernelGGL(silu_embed_bwd_kernel, dim3((unsigned)B, (unsigned)n_tab), dim3(128), 0, pfpp::as_stream(stream), tables,
                     t, dse, dtables, n_emb, B, (int)C, active);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_mse_loss(const float* pred, const float* target, const uint8_t* sel, float* loss, float* dpred,
                             int64_t n, int64_t width, float grad_out, pfpp_stream_t stream) {
  PFPP_REQUIRE(pred && target && sel && loss, "null pointer");
  PFPP_REQUIRE(n >= 0 && width >= 1, "bad sizes");
  hipLaunchKernelGGL(mse_loss_kernel, dim3(1), dim3(1024), 0, pfpp::as_stream(stream), pred, target, sel, loss, dpred, n,
                     (int)width, grad_out, (const float*)nullptr, (const uint8_t*)nullptr, (float*)nullptr);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_mse_loss_masked(const float* pred, const float* target, const float* valid, const uint8_t* ref, float* loss,
                                    float* dpred, float* amax, int64_t n, int64_t width, float grad_out, pfpp_stream_t stream) {
  PFPP_REQUIRE(pred && target && valid && ref && loss, "null pointer");
  PFPP_REQUIRE(n >= 0 && width >= 1 && (!amax || dpred), "bad sizes (amax needs dpred)");
  hipLaunchKernelGGL(mse_loss_kernel, dim3(1), dim3(1024), 0, pfpp::as_stream(stream), pred, target, (const uint8_t*)nullptr, loss, dpred,
                     n, (int)width, grad_out, valid, ref, amax);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_adamw(float* p, const float* g, float* m, float* v, void* hi, void* lo, int64_t n, float lr,
                          float beta1, float beta2, float eps, float weight_decay, float bc1, float bc2,
                          float g_scale, pfpp_stream_t stream) {
  return pfpp_adamw_zero(p, const_cast<float*>(g), m, v, hi, lo, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, g_scale, 0, stream);
}

extern "C" int pfpp_adamw_zero(float* p, float* g, float* m, float* v, void* hi, void* lo, int64_t n, float lr,
                               float beta1, float beta2, float eps, float weight_decay, float bc1, float bc2,
                               float g_scale, int zero_grad, pfpp_stream_t stream) {
  return pfpp_adamw_guarded(p, g, m, v, hi, lo, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, g_scale, zero_grad, nullptr, stream);
}

extern "C" int pfpp_adamw_rows(float* p, float* g, float* m, float* v, void* hi, void* lo, int64_t n_tables, int64_t rows_per_table,
                               int64_t C, const int64_t* t, int64_t n_t, int mode, float lr, float beta1, float beta2, float eps,
                               float weight_decay, float bc1, float bc2, float g_scale, int zero_grad, int32_t* overflow,
                               pfpp_stream_t stream) {
  PFPP_REQUIRE(p && g && m && v && t, "null pointer");
  PFPP_REQUIRE(!hi == !lo, "hi and lo go together");
  PFPP_REQUIRE(bc1 > 0.0f && bc2 > 0.0f, "bias corrections must be positive");
  PFPP_REQUIRE(n_tables >= 1 && rows_per_table >= 1 && rows_per_table <= 4096 && C >= 1 && C < (1 << 30) && n_t >= 0 && n_t <= 4096, "sizes");
  PFPP_REQUIRE(mode == 0 || mode == 1, "mode: 0 = all rows but the listed ones, 1 = the listed rows");
  hipStream_t st = pfpp::as_stream(stream);
  const int64_t n = mode == 0 ? n_tables * rows_per_table * C : n_tables * n_t * C;
  if (n == 0) return PFPP_OK;
  const dim3 grid(blocks_for(n, 256));
  const float decay = 1.0f - lr * weight_decay, w1 = 1.0f - beta1, w2 = 1.0f - beta2, step = lr / bc1, isb = 1.0f / sqrtf(bc2);
#define PFPP_ROWS_LAUNCH(MODE, GUARD)                                                                                             \
  hipLaunchKernelGGL((adamw_rows_kernel<MODE, GUARD>), grid, dim3(256), 0, st, p, g, m, v, (_Float16*)hi, (_Float16*)lo, n_tables,  \
                     (int)rows_per_table, (int)C, t, (int)n_t, decay, w1, beta2, w2, eps, step, isb, g_scale, zero_grad ? 1 : 0,   \
                     (int*)overflow)
  if (mode == 0) { if (overflow) PFPP_ROWS_LAUNCH(0, true); else PFPP_ROWS_LAUNCH(0, false); }
  else { if (overflow) PFPP_ROWS_LAUNCH(1, true); else PFPP_ROWS_LAUNCH(1, false); }
#undef PFPP_ROWS_LAUNCH
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_adamw_rows_active(float* p, float* g, float* m, float* v, void* hi, void* lo, int64_t n_tables, int64_t rows_per_table,
                                      int64_t C, const uint32_t* active, float lr, float beta1, float beta2, float eps, float weight_decay,
                                      float bc1, float bc2, float g_scale, int zero_grad, int32_t* overflow, pfpp_stream_t stream) {
  PFPP_REQUIRE(p && g && m && v && active, "null pointer");
  PFPP_REQUIRE(!hi == !lo, "hi and lo go together");
  PFPP_REQUIRE(bc1 > 0.0f && bc2 > 0.0f, "bias corrections must be positive");
  PFPP_REQUIRE(n_tables >= 1 && rows_per_table >= 1 && rows_per_table <= 4096 && C >= 1 && C < (1 << 30), "sizes");
  const float decay = 1.0f - lr * weight_decay;
  // a row without a set bit has g = m = v = 0: its update is p * decay, the identity only when decay rounds to 1 — otherwise every row is taken
  if (decay != 1.0f)
    return pfpp_adamw_guarded(p, g, m, v, hi, lo, n_tables * rows_per_table * C, lr, beta1, beta2, eps, weight_decay, bc1, bc2, g_scale,
                              zero_grad, overflow, stream);
  hipStream_t st = pfpp::as_stream(stream);
  const int64_t n = n_tables * rows_per_table * C;
  const dim3 grid(blocks_for(n, 256));
  const float w1 = 1.0f - beta1, w2 = 1.0f - beta2, step = lr / bc1, isb = 1.0f / sqrtf(bc2);
  const int64_t* bits = reinterpret_cast<const int64_t*>(active);
  if (overflow)
    hipLaunchKernelGGL((adamw_rows_kernel<2, true>), grid, dim3(256), 0, st, p, g, m, v, (_Float16*)hi, (_Float16*)lo, n_tables,
                       (int)rows_per_table, (int)C, bits, 0, decay, w1, beta2, w2, eps, step, isb, g_scale, zero_grad ? 1 : 0, (int*)overflow);
  else
    hipLaunchKernelGGL((adamw_rows_kernel<2, false>), grid, dim3(256), 0, st, p, g, m, v, (_Float16*)hi, (_Float16*)lo, n_tables,
                       (int)rows_per_table, (int)C, bits, 0, decay, w1, beta2, w2, eps, step, isb, g_scale, zero_grad ? 1 : 0, (int*)overflow);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_adamw_guarded(float* p, float* g, float* m, float* v, void* hi, void* lo, int64_t n, float lr,
                                  float beta1, float beta2, float eps, float weight_decay, float bc1, float bc2,
                                  float g_scale, int zero_grad, int32_t* overflow, pfpp_stream_t stream) {
  PFPP_REQUIRE(p && g && m && v, "null pointer");
  PFPP_REQUIRE(!hi == !lo, "hi and lo go together");
  PFPP_REQUIRE(bc1 > 0.0f && bc2 > 0.0f, "bias corrections must be positive");
  if (n == 0) return PFPP_OK;
  const dim3 grid(blocks_for(n, 256));
  hipStream_t st = pfpp::as_stream(stream);
  if (overflow)
    hipLaunchKernelGGL(adamw_kernel<true>, grid, dim3(256), 0, st, p, g, m, v, (_Float16*)hi, (_Float16*)lo, n,
                       1.0f - lr * weight_decay, 1.0f - beta1, beta2, 1.0f - beta2, eps, lr / bc1, 1.0f / sqrtf(bc2), g_scale,
                       zero_grad ? 1 : 0, (int*)overflow);
  else
    hipLaunchKernelGGL(adamw_kernel<false>, grid, dim3(256), 0, st, p, g, m, v, (_Float16*)hi, (_Float16*)lo, n,
                       1.0f - lr * weight_decay, 1.0f - beta1, beta2, 1.0f - beta2, eps, lr / bc1, 1.0f / sqrtf(bc2), g_scale,
                       zero_grad ? 1 : 0, (int*)nullptr);
  return pfpp::check_launch(__func__);
}
