// Train-mode BatchNorm of the set-abstraction MLPs (utils/pn2_utils.py:211-214): the reference keeps the
// "frozen" encoder in .train() during Denoiser training, so its BatchNorm2d layers use batch statistics.
//
// Both kernels stream x [rows, C] once (HBM-bound; rows is F*S*nsample ~ 1e6).  Statistics are
// accumulated in fp64 like ATen's CPU kernel: per-thread double sums over a 2048-row slab, an LDS
// reduction over the row phases, one [2, C] double partial per slab, and a tiny second kernel that
// adds the partials in slab order (deterministic) and updates the running statistics.
#include "pfpp_common.h"

namespace {

constexpr int SLAB = 2048;

// block: CL = C/4 column lanes x RP = 256/CL row phases (C in {64,128,256,512,1024} -> CL <= 256)
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, int64_t rows, int C, int64_t ld,
                                                         double* __restrict__ part) {
  extern __shared__ double bn_red[];      // [RP][2][C]
  const int CL = C / 4;
  const int RP = 256 / CL;
  const int cl = threadIdx.x % CL, rp = threadIdx.x / CL;
  const int64_t r0 = (int64_t)blockIdx.x * SLAB;
  const int64_t r1 = min(rows, r0 + SLAB);
  double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  if (rp < RP) {
    for (int64_t r = r0 + rp; r < r1; r += RP) {
      const float4 v = *reinterpret_cast<const float4*>(x + r * ld + cl * 4);
      const double a = v.x, b = v.y, c = v.z, d = v.w;
      s[0] += a; s[1] += b; s[2] += c; s[3] += d;
      q[0] += a * a; q[1] += b * b; q[2] += c * c; q[3] += d * d;
    }
    double* o = bn_red + (size_t)rp * 2 * C;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[cl * 4 + e] = s[e]; o[C + cl * 4 + e] = q[e]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    double a = 0.0;
    for (int p = 0; p < RP; ++p) a += bn_red[(size_t)p * 2 * C + i];
    part[(size_t)blockIdx.x * 2 * C + i] = a;
  }
}

// one workgroup per 16 columns: 16 column lanes x 64 partial phases, tree over the phases in LDS (fixed order:
// deterministic), then the running-statistics update
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const double* __restrict__ part, int n_part, int64_t rows, int C,
                                                           float* __restrict__ mean, float* __restrict__ var,
                                                           float* __restrict__ rmean, float* __restrict__ rvar,
                                                           float momentum) {
  __shared__ double red[2][64][16];
  const int cl = threadIdx.x & 15, ph = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double s = 0.0, q = 0.0;
  if (c < C) {
    for (int p = ph; p < n_part; p += 64) {
      s += part[(size_t)p * 2 * C + c];
      q += part[(size_t)p * 2 * C + C + c];
    }
  }
  red[0][ph][cl] = s;
  red[1][ph][cl] = q;
  __syncthreads();
  for (int half = 32; half > 0; half >>= 1) {
    if (ph < half) {
      red[0][ph][cl] += red[0][ph + half][cl];
      red[1][ph][cl] += red[1][ph + half][cl];
    }
    __syncthreads();
  }
  if (ph != 0 || c >= C) return;
  s = red[0][0][cl];
  q = red[1][0][cl];
  const double n = (double)rows;
  const double m = s / n;
  double v = q / n - m * m;
  if (v < 0.0) v = 0.0;
  mean[c] = (float)m;
  var[c] = (float)v;
  if (rmean) {
    const double unbiased = rows > 1 ? v * n / (n - 1.0) : v;
    rmean[c] = (float)((1.0 - (double)momentum) * (double)rmean[c] + (double)momentum * m);
    rvar[c] = (float)((1.0 - (double)momentum) * (double)rvar[c] + (double)momentum * unbiased);
  }
}

// y = relu(x*a + b), optional max over `pool` consecutive rows.  thread = one float4 column group of one
// output row; coalesced along C.
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, int64_t out_rows, int C, int64_t ld,
                                                       const float* __restrict__ mean, const float* __restrict__ var,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps, float* __restrict__ y, int64_t ldy, int pool) {
  const int CL = C / 4;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= out_rows * CL) return;
  const int64_t orow = i / CL;
  const int c = (int)(i - orow * CL) * 4;
  const float4 m = *reinterpret_cast<const float4*>(mean + c);
  const float4 v = *reinterpret_cast<const float4*>(var + c);
  const float4 g = *reinterpret_cast<const float4*>(gamma + c);
  const float4 be = *reinterpret_cast<const float4*>(beta + c);
  float4 a, b;
  a.x = g.x * (1.0f / sqrtf(v.x + eps)); a.y = g.y * (1.0f / sqrtf(v.y + eps));
  a.z = g.z * (1.0f / sqrtf(v.z + eps)); a.w = g.w * (1.0f / sqrtf(v.w + eps));
  b.x = be.x - m.x * a.x; b.y = be.y - m.y * a.y; b.z = be.z - m.z * a.z; b.w = be.w - m.w * a.w;
  const int np = pool > 0 ? pool : 1;
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);          // relu output >= 0: 0 is the identity of the max
  for (int p = 0; p < np; ++p) {
    const float4 t = *reinterpret_cast<const float4*>(x + (orow * np + p) * ld + c);
    o.x = fmaxf(o.x, t.x * a.x + b.x); o.y = fmaxf(o.y, t.y * a.y + b.y);
    o.z = fmaxf(o.z, t.z * a.z + b.z); o.w = fmaxf(o.w, t.w * a.w + b.w);
  }
  *reinterpret_cast<float4*>(y + orow * ldy + c) = o;
}

// fused path: stats [copies][2][C] from the GEMM epilogues -> affine for the consumer; clears stats
__global__ __launch_bounds__(256) void bn_finalize_fused_kernel(double* __restrict__ stats, int copies, int64_t rows, int C,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float eps, float momentum, float* __restrict__ rmean,
                                                                float* __restrict__ rvar, float* __restrict__ mean,
                                                                float* __restrict__ var, float* __restrict__ a_mul,
                                                                float* __restrict__ a_add) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  // loads first (independent, pipelined; same summation order), the clearing stores after: interleaved, every store fenced
  // the next copy's loads off and the 64 copies were 64 round trips (20 us per finalize, nine per encoder pass)
  double s = 0.0, q = 0.0;
#pragma unroll 16
  for (int k = 0; k < copies; ++k) {
    s += stats[(size_t)k * 2 * C + c];
    q += stats[(size_t)k * 2 * C + C + c];
  }
  for (int k = 0; k < copies; ++k) {
    stats[(size_t)k * 2 * C + c] = 0.0;
    stats[(size_t)k * 2 * C + C + c] = 0.0;
  }
  const double n = (double)rows;
  const double m = s / n;
  double v = q / n - m * m;
  if (v < 0.0) v = 0.0;
  const float mf = (float)m, vf = (float)v;
  if (mean) { mean[c] = mf; var[c] = vf; }
  if (rmean) {
    const double unbiased = rows > 1 ? v * n / (n - 1.0) : v;
    rmean[c] = (float)((1.0 - (double)momentum) * (double)rmean[c] + (double)momentum * m);
    rvar[c] = (float)((1.0 - (double)momentum) * (double)rvar[c] + (double)momentum * unbiased);
  }
  const float a = gamma[c] * (1.0f / sqrtf(vf + eps));
  a_mul[c] = a;
  a_add[c] = beta[c] - mf * a;
}

__global__ __launch_bounds__(256) void bn_minmax_apply_kernel(const float* __restrict__ mx, const float* __restrict__ mn,
                                                              const float* __restrict__ a_mul, const float* __restrict__ a_add,
                                                              float* __restrict__ y, int64_t total, int C) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  const float a = a_mul[c];
  y[i] = fmaxf(a * (a >= 0.0f ? mx[i] : mn[i]) + a_add[c], 0.0f);
}

}  // namespace

extern "C" int pfpp_bn_finalize(double* stats, int64_t copies, int64_t rows, int64_t C, const float* gamma,
                                const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                float* mean, float* var, float* a_mul, float* a_add, pfpp_stream_t stream) {
  PFPP_REQUIRE(stats && gamma && beta && a_mul && a_add, "null pointer");
  PFPP_REQUIRE(copies >= 1 && rows >= 1 && C >= 1, "bad sizes");
  PFPP_REQUIRE(!running_mean == !running_var && !mean == !var, "mean/var pointers go in pairs");
  hipLaunchKernelGGL(bn_finalize_fused_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, pfpp::as_stream(stream), stats,
                     (int)copies, rows, (int)C, gamma, beta, eps, momentum, running_mean, running_var, mean, var, a_mul, a_add);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_bn_minmax_apply(const float* mx, const float* mn, const float* a_mul, const float* a_add, float* y,
                                    int64_t rows, int64_t C, pfpp_stream_t stream) {
  PFPP_REQUIRE(mx && mn && a_mul && a_add && y, "null pointer");
  const int64_t total = rows * C;
  if (total == 0) return PFPP_OK;
  hipLaunchKernelGGL(bn_minmax_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, pfpp::as_stream(stream), mx,
                     mn, a_mul, a_add, y, total, (int)C);
  return pfpp::check_launch(__func__);
}

extern "C" int64_t pfpp_bn_stats_workspace(int64_t rows, int64_t C) {
  const int64_t n_part = (rows + SLAB - 1) / SLAB;
  return n_part * 2 * C * (int64_t)sizeof(double);
}

extern "C" int pfpp_bn_stats(const float* x, int64_t rows, int64_t C, int64_t ld, float* mean, float* var,
                             float* running_mean, float* running_var, float momentum, void* workspace,
                             pfpp_stream_t stream) {
  PFPP_REQUIRE(x && mean && var && workspace, "null pointer");
  PFPP_REQUIRE(!running_mean == !running_var, "running_mean and running_var go together");
  PFPP_REQUIRE(rows >= 1 && ld >= C && ld % 4 == 0 && pfpp::aligned16(x), "bad sizes / alignment");
  PFPP_SUPPORTED(C == 64 || C == 128 || C == 256 || C == 512 || C == 1024, "C not in {64,128,256,512,1024}");
  hipStream_t st = pfpp::as_stream(stream);
  const int n_part = (int)((rows + SLAB - 1) / SLAB);
  const int RP = 256 / (int)(C / 4);
  const size_t smem = (size_t)RP * 2 * C * sizeof(double);
  hipLaunchKernelGGL(bn_partial_kernel, dim3(n_part), dim3(256), smem, st, x, rows, (int)C, ld, (double*)workspace);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)((C + 15) / 16)), dim3(1024), 0, st, (const double*)workspace,
                     n_part, rows, (int)C, mean, var, running_mean, running_var, momentum);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_bn_apply(const float* x, int64_t rows, int64_t C, int64_t ld, const float* mean, const float* var,
                             const float* gamma, const float* beta, float eps, float* y, int64_t ldy, int64_t pool,
                             pfpp_stream_t stream) {
  PFPP_REQUIRE(x && mean && var && gamma && beta && y, "null pointer");
  PFPP_REQUIRE(C % 4 == 0 && C <= 1024 && ld >= C && ldy >= C && ld % 4 == 0 && ldy % 4 == 0, "bad sizes");
  PFPP_REQUIRE(pool >= 0 && (pool == 0 || rows % pool == 0), "rows % pool != 0");
  PFPP_REQUIRE(pfpp::aligned16(x) && pfpp::aligned16(y) && pfpp::aligned16(mean) && pfpp::aligned16(var) &&
               pfpp::aligned16(gamma) && pfpp::aligned16(beta), "16-byte alignment");
  if (rows == 0) return PFPP_OK;
  const int64_t out_rows = pool > 0 ? rows / pool : rows;
  const int64_t total = out_rows * (C / 4);
  hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, pfpp::as_stream(stream), x,
                     out_rows, (int)C, ld, mean, var, gamma, beta, eps, y, ldy, (int)pool);
  return pfpp::check_launch(__func__);
}
