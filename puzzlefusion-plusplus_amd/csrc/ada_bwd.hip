// Backward of the twelve AdaLN modulation linears in two launches, fp32 FMA, no atomics.
//
// Reference: MyAdaLayerNorm.forward (denoiser/model/modules/attention.py:21-25): mods_j = Linear_j(silu(emb_j(t))) for the 2 * num_layers
// norms of the denoiser; se_j = silu(emb_j(t)) [B, C], W_j [2 C, C], dmods_j [B, 2 C] arrives from the LayerNorm backward kernels.
//   g_W_j[n, k] += sum_b dmods_j[b, n] se_j[b, k]        g_b_j[n] += sum_b dmods_j[b, n]        dse_j[b, k] = sum_n dmods_j[b, n] W_j[n, k]
//
// Why not the GEMM kernels: the contraction of the weight gradient is over the B = 32 puzzles of a step — twelve outer products of depth 32
// that rewrite 25 MB of gradient — and d(se) has 32 output rows.  The tiled gradient GEMMs took 49 + 27 us (+ 7 us column sums) for them at
// the exposed end of the backward; both are memory passes: 50 MB (read-modify-write of g_W) and 25 MB (W).
//   ada_dw_kernel   workgroup = (norm j, 32 rows n of W_j): the 32 x 32 block of dmods_j transposed into LDS ([n][b], also written out
//                   as dmT for the second kernel); a thread keeps se_j[0..31][4 columns] in registers and walks 16 rows: one 16-byte
//                   load / store of g_W per 128 FMAs, the dmods values as LDS broadcasts.  Column sums (g_b) from the same block.
//   ada_dse_kernel  workgroup = (norm j, 32 columns k): wave w of eight walks an eighth of the rows n, 8 rows per step (lane = row x 16-byte
//                   column piece); every lane accumulates its row's contribution to all 32 x 4 outputs, then a reduce-scatter over the
//                   8 rows of a step (3 exchange rounds, each halving what a lane holds), then the eight waves through LDS.  Fixed
//                   summation tree: deterministic, so data-parallel replicas stay bit-equal.
// Arithmetic is plain fp32 (b ascending for g_W / g_b); the tiled kernels' split-f16 products differ from it below 1e-6 relative.
#include "pfpp_common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BC = 32;            // puzzles per pass (chunk of B)
constexpr int RB = 32;            // rows of W_j per workgroup of ada_dw_kernel
constexpr int DLD = 36;           // LDS row stride of the transposed dmods block (floats; 16-byte aligned rows, write conflicts 8-way at most)

struct AdaP {
  const float* dmods;             // [n_ada, B, N2]
  const float* se;                // [n_ada, B, C]
  const float* w;                 // [n_ada, N2, C]
  float* gw;                      // [n_ada, N2, C]   +=
  float* gb;                      // [n_ada, N2]      +=
  float* dmt;                     // [n_ada, N2, BC]  scratch: dmods of the current chunk, transposed, zero beyond B
  float* dse;                     // [n_ada, B, C]    =
  int B, C, N2, b0;               // b0: first puzzle of this pass
};

__global__ __launch_bounds__(256) void ada_dw_kernel(const AdaP p) {
  __shared__ __align__(16) float dm[RB * DLD];
  const int tid = threadIdx.x, j = blockIdx.y, n0 = blockIdx.x * RB;
  const int nb = min(BC, p.B - p.b0);
  {   // dmods_j[b0 + b][n0 .. n0 + 32) -> dm[n][b]
    const int b = tid >> 3, r4 = (tid & 7) * 4;
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (b < nb) v = *reinterpret_cast<const f32x4*>(p.dmods + ((int64_t)j * p.B + p.b0 + b) * p.N2 + n0 + r4);
#pragma unroll
    for (int i = 0; i < 4; ++i) dm[(r4 + i) * DLD + b] = v[i];
  }
  __syncthreads();
  {   // the transposed block for ada_dse_kernel: 32 rows x 128 bytes, contiguous
    const int r = tid >> 3, b4 = (tid & 7) * 4;
    *reinterpret_cast<f32x4*>(p.dmt + ((int64_t)j * p.N2 + n0 + r) * BC + b4) = *reinterpret_cast<const f32x4*>(dm + r * DLD + b4);
  }
  if (tid < RB) {
    float s = 0.0f;
    for (int b = 0; b < nb; ++b) s += dm[tid * DLD + b];
    p.gb[(int64_t)j * p.N2 + n0 + tid] += s;
  }
  const int half = tid >> 7, q = tid & 127;
  for (int kq = q; kq < p.C / 4; kq += 128) {
    f32x4 s4[BC];
#pragma unroll
    for (int b = 0; b < BC; ++b) {
      s4[b] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (b < nb) s4[b] = *reinterpret_cast<const f32x4*>(p.se + ((int64_t)j * p.B + p.b0 + b) * p.C + kq * 4);
    }
    float* g = p.gw + ((int64_t)j * p.N2 + n0 + half * (RB / 2)) * p.C + kq * 4;
#pragma unroll 8
    for (int r = 0; r < RB / 2; ++r) {
      const float* d = dm + (half * (RB / 2) + r) * DLD;
      f32x4 acc = *reinterpret_cast<const f32x4*>(g + (int64_t)r * p.C);
      f32x4 sum = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int b4 = 0; b4 < BC; b4 += 4) {
        const f32x4 dv = *reinterpret_cast<const f32x4*>(d + b4);
#pragma unroll
        for (int i = 0; i < 4; ++i) sum += dv[i] * s4[b4 + i];
      }
      *reinterpret_cast<f32x4*>(g + (int64_t)r * p.C) = acc + sum;
    }
  }
}

__device__ __forceinline__ f32x4 xchg(const f32x4 v, const int mask) {
  f32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = __shfl_xor(v[i], mask, 64);
  return r;
}

constexpr int DW = 8;             // waves of ada_dse_kernel (each an eighth of the rows n: 16 steps of 8 rows, enough of them per SIMD to hide the loads)
__global__ __launch_bounds__(64 * DW) void ada_dse_kernel(const AdaP p) {
  __shared__ __align__(16) float red[DW][BC][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = blockIdx.y, k0 = blockIdx.x * 32;
  const int kq = lane & 7, rs = lane >> 3;
  const int rows_w = p.N2 / DW;
  const float* wrow = p.w + ((int64_t)j * p.N2 + wave * rows_w + rs) * p.C + k0 + kq * 4;
  const float* drow = p.dmt + ((int64_t)j * p.N2 + wave * rows_w + rs) * BC;
  f32x4 acc[BC];
#pragma unroll
  for (int b = 0; b < BC; ++b) acc[b] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 4
  for (int it = 0; it < rows_w / 8; ++it) {
    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wrow + (int64_t)it * 8 * p.C);
#pragma unroll
    for (int b4 = 0; b4 < BC; b4 += 4) {
      const f32x4 dv = *reinterpret_cast<const f32x4*>(drow + (int64_t)it * 8 * BC + b4);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[b4 + i] += dv[i] * w4;
    }
  }
  // reduce-scatter over the 8 rows of a step (lane bits 5, 4, 3): after the rounds a lane holds puzzles 16 [bit 5] + 8 [bit 4] + 4 [bit 3] + 0..3
  f32x4 a16[16], a8[8], a4[4];
  {
    const bool hi = lane & 32;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const f32x4 send = hi ? acc[i] : acc[i + 16], keep = hi ? acc[i + 16] : acc[i];
      a16[i] = keep + xchg(send, 32);
    }
  }
  {
    const bool hi = lane & 16;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x4 send = hi ? a16[i] : a16[i + 8], keep = hi ? a16[i + 8] : a16[i];
      a8[i] = keep + xchg(send, 16);
    }
  }
  {
    const bool hi = lane & 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 send = hi ? a8[i] : a8[i + 4], keep = hi ? a8[i + 4] : a8[i];
      a4[i] = keep + xchg(send, 8);
    }
  }
  const int bb = ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(&red[wave][bb + i][kq * 4]) = a4[i];
  __syncthreads();
  const int b = tid >> 3, c4 = (tid & 7) * 4;
  if (tid < 256 && p.b0 + b < p.B) {
    f32x4 s = *reinterpret_cast<const f32x4*>(&red[0][b][c4]);
#pragma unroll
    for (int w = 1; w < DW; ++w) s += *reinterpret_cast<const f32x4*>(&red[w][b][c4]);
    *reinterpret_cast<f32x4*>(p.dse + ((int64_t)j * p.B + p.b0 + b) * p.C + k0 + c4) = s;
  }
}

}  // namespace

extern "C" int64_t pfpp_ada_linear_bwd_scratch_floats(int64_t n_ada, int64_t N2) { return n_ada * N2 * BC; }

extern "C" int pfpp_ada_linear_bwd(const float* dmods, const float* se, const float* w, float* g_w, float* g_b, float* dse, float* scratch,
                                   int64_t n_ada, int64_t B, int64_t C, int64_t N2, pfpp_stream_t stream) {
  PFPP_REQUIRE(dmods && se && w && g_w && g_b && dse && scratch, "null pointer");
  PFPP_REQUIRE(n_ada >= 1 && n_ada <= 65535 && B >= 1 && B <= 0x7fffffff, "sizes");
  PFPP_SUPPORTED(C >= 32 && C % 32 == 0 && N2 >= 64 && N2 % 64 == 0 && C <= 0x7fffffff && N2 <= 0x7fffffff, "C % 32 != 0 or N2 % 64 != 0");
  PFPP_REQUIRE(pfpp::aligned16(dmods) && pfpp::aligned16(se) && pfpp::aligned16(w) && pfpp::aligned16(g_w) && pfpp::aligned16(dse) &&
               pfpp::aligned16(scratch), "16-byte aligned operands");
  AdaP p;
  p.dmods = dmods; p.se = se; p.w = w; p.gw = g_w; p.gb = g_b; p.dmt = scratch; p.dse = dse;
  p.B = (int)B; p.C = (int)C; p.N2 = (int)N2;
  hipStream_t st = pfpp::as_stream(stream);
  for (int64_t b0 = 0; b0 < B; b0 += BC) {      // more than 32 puzzles per step: one pass per 32 (the scratch block is reused in stream order)
    p.b0 = (int)b0;
    hipLaunchKernelGGL(ada_dw_kernel, dim3((unsigned)(N2 / RB), (unsigned)n_ada), dim3(256), 0, st, p);
    hipLaunchKernelGGL(ada_dse_kernel, dim3((unsigned)(C / 32), (unsigned)n_ada), dim3(64 * DW), 0, st, p);
  }
  return pfpp::check_launch(__func__);
}
