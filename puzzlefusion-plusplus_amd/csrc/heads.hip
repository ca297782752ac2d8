// The two output heads of the DenoiserTransformer as ONE launch each way.
//
// Reference: DenoiserTransformer._out (denoiser/model/modules/denoiser_transformer.py:138-147): the pooled fragment rows go through
//   mlp_out_trans = Linear(C, C) - SiLU - Linear(C, C/2) - SiLU - Linear(C/2, 3)   and   mlp_out_rot = ... - Linear(C/2, 4)
// (constructed at :58-61), pred = cat(trans, rot).  Backward: autograd of the same (Denoiser.training_step, denoiser.py:128-145).
//
// Before: 6 skinny GEMMs + 4 activation launches + a scatter forward (~115 us), ~35 launches backward (~350 us) — every one a
// 5-25 us dependent step at 3-6 TFLOP/s, all of them on the critical chain of the iteration.  Here a workgroup owns 32 rows of one
// head and walks the whole chain: the activations between the layers stay in LDS as split-f16 planes, the weights stream through
// the matrix cores straight from their planes in global memory (row-major for the forward; for the backward's dX = dY . W the
// k-major weight tile goes through a wave-private LDS patch and comes back through the transposing read, like csrc/gemm_pl.hip).
// Arithmetic: the split-f16 contraction of the GEMMs (3 x v_mfma_f32_32x32x16_f16 per 16-deep step, small terms first), the
// SiLU / SiLU' of csrc/train_ops.hip, the last (3- / 4-column) layer in fp32 FMAs.
#include "pfpp_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HC = 512;        // width
constexpr int HC2 = 256;       // hidden width of the second linear
constexpr int KP = HC + 8;     // LDS row stride of an activation plane in halfs: 1040 bytes -> rows 4 banks apart
constexpr int NW = 8;          // waves per workgroup

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + expf(-v)); }
__device__ __forceinline__ float silu_grad(float v) {
  const float s = 1.0f / (1.0f + expf(-v));
  return s * (1.0f + v * (1.0f - s));
}

struct HeadW {
  const _Float16 *w0h, *w0l, *w2h, *w2l;   // planes of scale * W0 [HC, HC], scale * W2 [HC2, HC] (row-major [out, in])
  const half8 *f0h, *f0l, *f2h, *f2l;      // the same planes fragment-blocked (pfpp_pw.fhi / flo layout) or null: static weights
  const float *w4, *b0, *b2, *b4;          // W4 [n_out, HC2] fp32, biases
  float inv_s0, inv_s2;                    // 1 / plane scale
  int n_out, c0;                           // 3 | 4 output columns starting at column c0 of the 7-wide prediction
};

struct HeadsFwdP {
  const float* x;                          // [R, HC] pooled rows
  HeadW w[2];
  float *a0, *v0, *a1, *v1;                // saved for the backward: [2, R, HC] / [2, R, HC2] (all or none)
  float* out;                              // [*, ldo]: row (slot ? slot[r] : r), columns c0 .. c0 + n_out - 1
  const int32_t* slot;
  int ldo, R;
};

// acc[j] += A[32 x K] (LDS planes, row stride KP) . W[n0 + 32 j + (0..31), 0..K)^T, W row-major planes with leading dimension ldw.
// The weights come straight from global memory into the B operand registers: lane (column n, half lhi) needs 8 contraction-consecutive
// halfs of row n per 16-deep step.  The contraction is walked in groups of 64 (one 128-byte line of every weight row): within a
// group lane half lhi owns the 64-byte half line [32 lhi, 32 lhi + 32) and step u of the group takes its u-th 16-byte piece — the four
// loads that share a line are issued back to back (the second to fourth hit the L1), and A is read from LDS in the same order.  One
// group is in flight while the previous one is multiplied; the loads are unconditional (a conditional prefetch makes the compiler
// wait for it on the spot): past the end they re-read the last group.
template <int NT>
__device__ __forceinline__ void contract_rowmajor(const _Float16* sh, const _Float16* sl, const _Float16* __restrict__ wh,
                                                  const _Float16* __restrict__ wl, int K, int ldw, int n0, f32x16 (&acc)[NT]) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, lhi = lane >> 5;
  const half8* ah = reinterpret_cast<const half8*>(sh + l31 * KP + 32 * lhi);
  const half8* al = reinterpret_cast<const half8*>(sl + l31 * KP + 32 * lhi);
  const half8 *bh[NT], *bl[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    bh[j] = reinterpret_cast<const half8*>(wh + (size_t)(n0 + 32 * j + l31) * ldw + 32 * lhi);
    bl[j] = reinterpret_cast<const half8*>(wl + (size_t)(n0 + 32 * j + l31) * ldw + 32 * lhi);
  }
  const int NG = K / 64;                   // groups of 4 steps; a group = 8 half8 along a row
  half8 fh[2][4][NT], fl[2][4][NT];
  auto fetch = [&](int g, int slot) {
    // one (tile, plane) after the other, its four pieces of a line back to back (alternating planes thrashed the L1)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
      for (int u = 0; u < 4; ++u) fh[slot][u][j] = bh[j][8 * g + u];
#pragma unroll
      for (int u = 0; u < 4; ++u) fl[slot][u][j] = bl[j][8 * g + u];
    }
  };
  fetch(0, 0);
  for (int g = 0; g < NG; g += 2) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int gg = g + half;
      if (gg < NG) {
        fetch(gg + 1 < NG ? gg + 1 : NG - 1, half ^ 1);
        __builtin_amdgcn_sched_barrier(0);         // the next group's 16 loads are issued HERE (the scheduler would sink them to their uses)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const half8 a_h = ah[8 * gg + u], a_l = al[8 * gg + u];
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_l, fh[half][u][j], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h, fl[half][u][j], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h, fh[half][u][j], acc[j], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

// The same contraction with the weights read from their fragment-blocked planes (eval: static weights, packing.PW.frag()): block (row
// tile, k-step) = the 64 lanes' B operands of one MFMA, 1 KB contiguous — one fully coalesced load per operand, where the row-major
// form above pulls 32 lines per instruction through the L1 (measured on 8-20 pooled rows: 56 us for the launch, two workgroups
// streaming 1.5 MB each).  tile0 = first 32-row tile of the weight; groups of four steps, one group in flight.
template <int NT>
__device__ __forceinline__ void contract_frag(const _Float16* sh, const _Float16* sl, const half8* __restrict__ fh, const half8* __restrict__ fl,
                                              int K, int tile0, f32x16 (&acc)[NT]) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, lhi = lane >> 5;
  const half8* ah = reinterpret_cast<const half8*>(sh + l31 * KP + 8 * lhi);
  const half8* al = reinterpret_cast<const half8*>(sl + l31 * KP + 8 * lhi);
  const int S = K / 16, NG = K / 64;
  half8 bh[2][4][NT], bl[2][4][NT];
  auto fetch = [&](int g, int slot) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t blk = (size_t)(tile0 + j) * S + 4 * g + u;
        bh[slot][u][j] = fh[blk * 64 + lane];
        bl[slot][u][j] = fl[blk * 64 + lane];
      }
  };
  fetch(0, 0);
  for (int g = 0; g < NG; g += 2) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int gg = g + half;
      if (gg < NG) {
        fetch(gg + 1 < NG ? gg + 1 : NG - 1, half ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const half8 a_h = ah[2 * (4 * gg + u)], a_l = al[2 * (4 * gg + u)];
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_l, bh[half][u][j], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h, bl[half][u][j], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h, bh[half][u][j], acc[j], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

__device__ __forceinline__ void zero(f32x16& a) {
#pragma unroll
  for (int e = 0; e < 16; ++e) a[e] = 0.0f;
}

__global__ __launch_bounds__(64 * NW) void heads_fwd_kernel(HeadsFwdP p) {
  extern __shared__ __align__(16) char hd_smem[];
  _Float16* sh = reinterpret_cast<_Float16*>(hd_smem);
  _Float16* sl = sh + 32 * KP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int hd = blockIdx.y, r0 = blockIdx.x * 32, R = p.R;
  const HeadW w = p.w[hd];
  // the 32 pooled rows as planes
  for (int i = tid; i < 32 * HC / 4; i += 64 * NW) {
    const int row = i / (HC / 4), c = (i - row * (HC / 4)) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + row < R) v = *reinterpret_cast<const float4*>(p.x + (size_t)(r0 + row) * HC + c);
    half4 h, l;
    PFPP_SPLIT_TO(v.x, h[0], l[0]); PFPP_SPLIT_TO(v.y, h[1], l[1]); PFPP_SPLIT_TO(v.z, h[2], l[2]); PFPP_SPLIT_TO(v.w, h[3], l[3]);
    *reinterpret_cast<half4*>(sh + row * KP + c) = h;
    *reinterpret_cast<half4*>(sl + row * KP + c) = l;
  }
  __syncthreads();
  // ---- Linear(C, C) + SiLU: wave w -> columns 64 w .. 64 w + 63
  {
    f32x16 acc[2];
    zero(acc[0]); zero(acc[1]);
    if (w.f0h) contract_frag<2>(sh, sl, w.f0h, w.f0l, HC, 2 * wave, acc);
    else contract_rowmajor<2>(sh, sl, w.w0h, w.w0l, HC, HC, 64 * wave, acc);
    __syncthreads();                                   // every wave has read the input planes
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = 64 * wave + 32 * j + l31;
      const float b = w.b0[col];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * lhi;
        const float a = acc[j][e] * w.inv_s0 + b;
        const float v = silu_f(a);
        if (p.a0 && r0 + row < R) {
          p.a0[((size_t)hd * R + r0 + row) * HC + col] = a;
          p.v0[((size_t)hd * R + r0 + row) * HC + col] = v;
        }
        _Float16 h, l;
        PFPP_SPLIT_TO(v, h, l);
        sh[row * KP + col] = h;
        sl[row * KP + col] = l;
      }
    }
    __syncthreads();
  }
  // ---- Linear(C, C/2) + SiLU: wave w -> columns 32 w .. 32 w + 31
  {
    f32x16 acc[1];
    zero(acc[0]);
    if (w.f2h) contract_frag<1>(sh, sl, w.f2h, w.f2l, HC, wave, acc);
    else contract_rowmajor<1>(sh, sl, w.w2h, w.w2l, HC, HC, 32 * wave, acc);
    __syncthreads();
    float* sv = reinterpret_cast<float*>(hd_smem);     // v1 [32][HC2 + 1] fp32 for the last layer
    const int col = 32 * wave + l31;
    const float b = w.b2[col];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * lhi;
      const float a = acc[0][e] * w.inv_s2 + b;
      const float v = silu_f(a);
      if (p.a1 && r0 + row < R) {
        p.a1[((size_t)hd * R + r0 + row) * HC2 + col] = a;
        p.v1[((size_t)hd * R + r0 + row) * HC2 + col] = v;
      }
      sv[row * (HC2 + 1) + col] = v;
    }
    __syncthreads();
    // ---- Linear(C/2, 3 | 4): one thread per (row, output column)
    if (tid < 32 * 4) {
      const int row = tid >> 2, c = tid & 3;
      if (c < w.n_out && r0 + row < R) {
        const float* wr = w.w4 + (size_t)c * HC2;
        float s = 0.0f;
#pragma unroll 8
        for (int k = 0; k < HC2; ++k) s = fmaf(sv[row * (HC2 + 1) + k], wr[k], s);
        const int64_t orow = p.slot ? p.slot[r0 + row] : (r0 + row);
        p.out[orow * p.ldo + w.c0 + c] = s + w.b4[c];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------------------------------
struct HeadsBwdP {
  const float* dout;                       // [*, 7] gradient of the prediction rows (unscaled); row r is dout[slot ? slot[r] : r]
  const int32_t* slot;
  HeadW w[2];
  const float *a0, *v0, *a1, *v1;          // saved by the forward
  float *da0, *da1;                        // out: [2, R, HC], [2, R, HC2] (unscaled; operands of the weight-gradient GEMMs)
  float* dp;                               // out: [2, R, HC] per-head gradient of the pooled rows
  float *gw4[2], *gb4[2], *gb2[2], *gb0[2];   // accumulated with atomics: dW4 [n_out, HC2], db4 [n_out], db2 [HC2], db0 [HC]
  float G;                                 // power of two lifting the gradient planes into the fp16 range
  int R;
};

__device__ __forceinline__ half4 lds_rd_tr(uint32_t addr) {
  half4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

// acc[j] += A[32 x K] (LDS planes of G * dY, row stride KP) . W[0..K) x [n0 + 32 j + (0..31)], W k-major = the weight as stored
// ([K = out, ldw = in] row-major planes).  A 16-row k-tile of the wave's 64 columns (2 KB per plane) is copied into the wave's own LDS
// patch [16][64] and fetched back with the transposing read: within a 16-lane group lane jj gets element jj of the four rows the
// group's lanes address (lanes 4 r .. 4 r + 3 supply row r, 4 halfs each) — lane (q = lane >> 4, jj) reads row 8 (q >> 1) + 4 t +
// (jj >> 2), columns 32 tile + 16 (q & 1) + 4 (jj & 3) .. + 3 and ends up with column 32 tile + (lane & 31), rows 8 lhi + 4 t + 0..3.
__device__ __forceinline__ void contract_kmajor(const _Float16* sh, const _Float16* sl, const _Float16* __restrict__ wh,
                                                const _Float16* __restrict__ wl, int K, int ldw, int n0, _Float16* patch,
                                                f32x16 (&acc)[2]) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, lhi = lane >> 5;
  const half8* ah = reinterpret_cast<const half8*>(sh + l31 * KP + 8 * lhi);
  const half8* al = reinterpret_cast<const half8*>(sl + l31 * KP + 8 * lhi);
  // copy: lane -> k-row lane >> 2 of the tile, 16-byte chunks (lane & 3) and (lane & 3) + 4 of its 64 columns
  const int krow = lane >> 2, ch = lane & 3;
  const half8* gh = reinterpret_cast<const half8*>(wh + (size_t)krow * ldw + n0 + 8 * ch);
  const half8* gl = reinterpret_cast<const half8*>(wl + (size_t)krow * ldw + n0 + 8 * ch);
  const size_t step = (size_t)16 * ldw / 8;          // half8 units per 16-row tile
  // patch: [2 buffers][2 planes][16 rows][64 cols] halfs = 8 KB per wave
  half8* pw = reinterpret_cast<half8*>(patch) + krow * 8 + ch;
  const int q = lane >> 4, jj = lane & 15;
  const uint32_t rd0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)patch +
                       (uint32_t)(((8 * (q >> 1) + (jj >> 2)) * 64 + 16 * (q & 1) + 4 * (jj & 3)) * 2);
  const int S = K / 16;
  constexpr int DEPTH = 4;                           // k-tiles in flight (registers); the LDS patch is double-buffered
  half8 r_h[DEPTH][2], r_l[DEPTH][2];                // [slot][chunk]
  auto fetch = [&](int s, int slot) {                // unconditional: past the end the last tile is read again
    const int sc = s < S ? s : S - 1;
    r_h[slot][0] = gh[sc * step]; r_h[slot][1] = gh[sc * step + 4];
    r_l[slot][0] = gl[sc * step]; r_l[slot][1] = gl[sc * step + 4];
  };
#pragma unroll
  for (int u = 0; u < DEPTH; ++u) fetch(u, u);
  for (int s0 = 0; s0 < S; s0 += DEPTH) {
#pragma unroll
   for (int u = 0; u < DEPTH; ++u) {
    const int s = s0 + u;
    const int slot = u & 1;
    half8* dst = pw + slot * 256;                    // buffer = 2 planes x 16 x 64 halfs = 256 half8 = 4 KB
    dst[0] = r_h[u][0]; dst[4] = r_h[u][1];
    dst[128] = r_l[u][0]; dst[132] = r_l[u][1];
    fetch(s + DEPTH, u);
    __builtin_amdgcn_sched_barrier(0);               // keep DEPTH tiles in flight (the scheduler would sink the loads to their uses)
    const half8 a_h = ah[2 * s], a_l = al[2 * s];
    const uint32_t rd = rd0 + slot * 4096;
    half4 t[2][2][2];                                // [tile][plane][t]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) t[j][pl][tt] = lds_rd_tr(rd + pl * 2048 + tt * 4 * 128 + j * 64);
    // the wait names every destination register: that is what orders the uses below behind it (the compiler believes an asm's
    // outputs are ready when the statement ends, and would otherwise assemble the fragments from registers still in flight)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(t[0][0][0]), "+v"(t[0][0][1]), "+v"(t[0][1][0]), "+v"(t[0][1][1]), "+v"(t[1][0][0]), "+v"(t[1][0][1]),
                   "+v"(t[1][1][0]), "+v"(t[1][1][1])
                 :
                 : "memory");
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      half8 b_h, b_l;
#pragma unroll
      for (int e = 0; e < 4; ++e) { b_h[e] = t[j][0][0][e]; b_h[4 + e] = t[j][0][1][e]; b_l[e] = t[j][1][0][e]; b_l[4 + e] = t[j][1][1][e]; }
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_l, b_h, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h, b_l, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h, b_h, acc[j], 0, 0, 0);
    }
   }
  }
}

__global__ __launch_bounds__(64 * NW) void heads_bwd_kernel(HeadsBwdP p) {
  extern __shared__ __align__(16) char hd_smem[];
  _Float16* sh = reinterpret_cast<_Float16*>(hd_smem);
  _Float16* sl = sh + 32 * KP;
  _Float16* patches = sl + 32 * KP;                    // NW x 8 KB
  __shared__ float s_do[32][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int hd = blockIdx.y, r0 = blockIdx.x * 32, R = p.R;
  const HeadW w = p.w[hd];
  const float G = p.G, invG = 1.0f / G;
  if (tid < 128) {
    const int row = tid >> 2, c = tid & 3;
    s_do[row][c] = (c < w.n_out && r0 + row < R) ? p.dout[(size_t)(p.slot ? p.slot[r0 + row] : r0 + row) * 7 + w.c0 + c] : 0.0f;
  }
  __syncthreads();
  // ---- last layer: db4, dW4 (contraction over this block's rows, then atomics), dv1 -> da1 = dv1 * silu'(a1)
  if (tid < 4) {
    float s = 0.0f;
    for (int r = 0; r < 32; ++r) s += s_do[r][tid];
    if (tid < w.n_out) unsafeAtomicAdd(p.gb4[hd] + tid, s);
  }
  {
    // thread -> column k = tid & 255 of v1, output columns c = 2 * (tid >> 8) and + 1
    const int k = tid & (HC2 - 1), cb = (tid >> 8) * 2;
    float s0 = 0.0f, s1 = 0.0f;
    for (int r = 0; r < 32; ++r) {
      if (r0 + r >= R) break;
      const float v = p.v1[((size_t)hd * R + r0 + r) * HC2 + k];
      s0 = fmaf(s_do[r][cb], v, s0);
      s1 = fmaf(s_do[r][cb + 1], v, s1);
    }
    if (cb < w.n_out) unsafeAtomicAdd(p.gw4[hd] + (size_t)cb * HC2 + k, s0);
    if (cb + 1 < w.n_out) unsafeAtomicAdd(p.gw4[hd] + (size_t)(cb + 1) * HC2 + k, s1);
  }
  {
    // da1 [32, HC2]: thread -> column k = tid & 255, rows (tid >> 8) * 16 .. + 15; column sums -> db2
    const int k = tid & (HC2 - 1), rb = (tid >> 8) * 16;
    float wk[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) wk[c] = c < w.n_out ? w.w4[(size_t)c * HC2 + k] : 0.0f;
    float cs = 0.0f;
    for (int r = rb; r < rb + 16; ++r) {
      float d = 0.0f;
      if (r0 + r < R) {
        float dv = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) dv = fmaf(s_do[r][c], wk[c], dv);
        d = dv * silu_grad(p.a1[((size_t)hd * R + r0 + r) * HC2 + k]);
        p.da1[((size_t)hd * R + r0 + r) * HC2 + k] = d;
      }
      cs += d;
      _Float16 h, l;
      PFPP_SPLIT_TO(d * G, h, l);
      sh[r * KP + k] = h;
      sl[r * KP + k] = l;
    }
    unsafeAtomicAdd(p.gb2[hd] + k, cs);
  }
  __syncthreads();
  // ---- dv0 = da1 . W2 (k-major), da0 = dv0 * silu'(a0): wave w -> columns 64 w .. + 63
  _Float16* patch = patches + wave * 4096;             // 8 KB per wave
  {
    f32x16 acc[2];
    zero(acc[0]); zero(acc[1]);
    contract_kmajor(sh, sl, w.w2h, w.w2l, HC2, HC, 64 * wave, patch, acc);
    __syncthreads();                                   // every wave has read the da1 planes
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = 64 * wave + 32 * j + l31;
      float cs = 0.0f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * lhi;
        float d = 0.0f;
        if (r0 + row < R) {
          d = acc[j][e] * (invG * w.inv_s2) * silu_grad(p.a0[((size_t)hd * R + r0 + row) * HC + col]);
          p.da0[((size_t)hd * R + r0 + row) * HC + col] = d;
        }
        cs += d;
        _Float16 h, l;
        PFPP_SPLIT_TO(d * G, h, l);
        sh[row * KP + col] = h;
        sl[row * KP + col] = l;
      }
      cs += __shfl_xor(cs, 32);
      if (lhi == 0) unsafeAtomicAdd(p.gb0[hd] + col, cs);
    }
    __syncthreads();
  }
  // ---- d pooled (this head's share) = da0 . W0 (k-major)
  {
    f32x16 acc[2];
    zero(acc[0]); zero(acc[1]);
    contract_kmajor(sh, sl, w.w0h, w.w0l, HC, HC, 64 * wave, patch, acc);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = 64 * wave + 32 * j + l31;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * lhi;
        if (r0 + row < R) p.dp[((size_t)hd * R + r0 + row) * HC + col] = acc[j][e] * (invG * w.inv_s0);
      }
    }
  }
}

// dx[(f, l), :] = (dp0[f, :] + dp1[f, :]) / L: mean_pool_bwd over the sum of the two heads' shares
__global__ __launch_bounds__(256) void pool_bwd2_kernel(const float* __restrict__ dp0, const float* __restrict__ dp1, float* __restrict__ dx,
                                                        int64_t n, int L, int C) {
  const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= n * L * C) return;
  const int64_t row = i4 / C;
  const int c = (int)(i4 - row * C);
  const int64_t f = row / L;
  const float4 a = *reinterpret_cast<const float4*>(dp0 + f * C + c);
  const float4 b = *reinterpret_cast<const float4*>(dp1 + f * C + c);
  const float inv = 1.0f / (float)L;
  float4 v;
  v.x = (a.x + b.x) * inv; v.y = (a.y + b.y) * inv; v.z = (a.z + b.z) * inv; v.w = (a.w + b.w) * inv;
  *reinterpret_cast<float4*>(dx + i4) = v;
}

bool fill_head(HeadW& w, const pfpp_head_params& h, int n_out, int c0) {
  if (!(h.w0.hi && h.w0.lo && h.w2.hi && h.w2.lo && h.w4 && h.b0 && h.b2 && h.b4)) return false;
  w.w0h = (const _Float16*)h.w0.hi; w.w0l = (const _Float16*)h.w0.lo;
  w.w2h = (const _Float16*)h.w2.hi; w.w2l = (const _Float16*)h.w2.lo;
  const bool frag = h.f0.hi && h.f0.lo && h.f2.hi && h.f2.lo;
  w.f0h = frag ? (const half8*)h.f0.hi : nullptr; w.f0l = frag ? (const half8*)h.f0.lo : nullptr;
  w.f2h = frag ? (const half8*)h.f2.hi : nullptr; w.f2l = frag ? (const half8*)h.f2.lo : nullptr;
  w.w4 = h.w4; w.b0 = h.b0; w.b2 = h.b2; w.b4 = h.b4;
  w.inv_s0 = 1.0f / h.w0.scale; w.inv_s2 = 1.0f / h.w2.scale;
  w.n_out = n_out; w.c0 = c0;
  return true;
}

}  // namespace

extern "C" int pfpp_heads_fwd(const float* pooled, const pfpp_head_params* trans, const pfpp_head_params* rot, int64_t R, int64_t C,
                              float* a0, float* v0, float* a1, float* v1, float* out, const int32_t* slot, int64_t ldo,
                              pfpp_stream_t stream) {
  PFPP_REQUIRE(pooled && trans && rot && out, "null pointer");
  PFPP_SUPPORTED(C == HC, "width != 512");
  PFPP_REQUIRE((!a0 && !v0 && !a1 && !v1) || (a0 && v0 && a1 && v1), "the saved activations come all or none");
  PFPP_REQUIRE(ldo >= 7 && pfpp::aligned16(pooled), "ldo < 7 or misaligned input");
  if (R == 0) return PFPP_OK;
  HeadsFwdP p;
  PFPP_REQUIRE(fill_head(p.w[0], *trans, 3, 0) && fill_head(p.w[1], *rot, 4, 3), "null weight pointer");
  p.x = pooled; p.a0 = a0; p.v0 = v0; p.a1 = a1; p.v1 = v1; p.out = out; p.slot = slot; p.ldo = (int)ldo; p.R = (int)R;
  const size_t smem = (size_t)2 * 32 * KP * sizeof(_Float16);
  static bool attr_set = false;
  if (!attr_set) {                                       // 65 KB of dynamic LDS: above the default cap
    if (hipFuncSetAttribute((const void*)heads_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return pfpp::check_launch(__func__);
    attr_set = true;
  }
  hipLaunchKernelGGL(heads_fwd_kernel, dim3((unsigned)((R + 31) / 32), 2), dim3(64 * NW), smem, pfpp::as_stream(stream), p);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_heads_bwd(const float* dout, const int32_t* slot, const pfpp_head_params* trans, const pfpp_head_params* rot, int64_t R, int64_t C,
                              const float* a0, const float* v0, const float* a1, const float* v1, float* da0, float* da1, float* dp,
                              const pfpp_head_grads* g_trans, const pfpp_head_grads* g_rot, float grad_scale, float* dx, int64_t L,
                              pfpp_stream_t stream) {
  PFPP_REQUIRE(dout && trans && rot && a0 && v0 && a1 && v1 && da0 && da1 && dp && g_trans && g_rot, "null pointer");
  PFPP_SUPPORTED(C == HC, "width != 512");
  PFPP_REQUIRE(grad_scale > 0.0f && L >= 1, "bad grad_scale / L");
  if (R == 0) return PFPP_OK;
  HeadsBwdP p;
  PFPP_REQUIRE(fill_head(p.w[0], *trans, 3, 0) && fill_head(p.w[1], *rot, 4, 3), "null weight pointer");
  const pfpp_head_grads* gs[2] = {g_trans, g_rot};
  for (int h = 0; h < 2; ++h) {
    PFPP_REQUIRE(gs[h]->w4 && gs[h]->b4 && gs[h]->b2 && gs[h]->b0, "null gradient pointer");
    p.gw4[h] = gs[h]->w4; p.gb4[h] = gs[h]->b4; p.gb2[h] = gs[h]->b2; p.gb0[h] = gs[h]->b0;
  }
  p.dout = dout; p.slot = slot; p.a0 = a0; p.v0 = v0; p.a1 = a1; p.v1 = v1; p.da0 = da0; p.da1 = da1; p.dp = dp; p.G = grad_scale; p.R = (int)R;
  hipStream_t st = pfpp::as_stream(stream);
  const size_t smem = (size_t)2 * 32 * KP * sizeof(_Float16) + (size_t)NW * 8192;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)heads_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return pfpp::check_launch(__func__);
    attr_set = true;
  }
  hipLaunchKernelGGL(heads_bwd_kernel, dim3((unsigned)((R + 31) / 32), 2), dim3(64 * NW), smem, st, p);
  if (dx) {
    const int64_t total = R * L * C;
    hipLaunchKernelGGL(pool_bwd2_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, dp, dp + R * C, dx, R, (int)L, (int)C);
  }
  return pfpp::check_launch(__func__);
}
