// Merge step of the auto-agglomerative loop: the device-side pieces of
// remove_intersect_points_and_fps_ds (utils/node_merge_utils.py:159-222).
//
//  * pfpp_estimate_normals — pytorch3d.ops.estimate_pointcloud_normals(neighborhood_size = K) restated
//    (pytorch3d is not in the reference tree): K nearest neighbours of every point
//    inside its own part (the point itself included), covariance of the neighbourhood about its mean,
//    eigenvector of the smallest eigenvalue, sign chosen so that at least half of the neighbours lie on
//    its positive side.  One thread per point, the part's points as SoA in LDS (broadcast reads), the K
//    best distances in a sorted register list (an insertion happens ~K ln(N/K) times per point), the
//    3x3 symmetric eigenproblem in fp64 (closed form + cross products).
//  * pfpp_merge_keep_mask — the pairwise filter of node_merge_utils.py:176-205: point k of part i is
//    dropped when for some other part j  nn(i->j)[k] + nn(j->i)[k] < threshold  and  n_i[k].n_j[k] < 0
//    (the reference pairs point k of part i with point k of part j).  d comes from pfpp_nn_dist.
#include "pfpp_common.h"

namespace {

constexpr int NRM_KMAX = 32;

__device__ __forceinline__ void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

// eigenvector of the smallest eigenvalue of the symmetric matrix (a00 a01 a02; a01 a11 a12; a02 a12 a22)
__device__ void smallest_eigvec(double a00, double a01, double a02, double a11, double a12, double a22, double* v) {
  const double p1 = a01 * a01 + a02 * a02 + a12 * a12;
  const double q = (a00 + a11 + a22) / 3.0;
  const double p2 = (a00 - q) * (a00 - q) + (a11 - q) * (a11 - q) + (a22 - q) * (a22 - q) + 2.0 * p1;
  double lam;
  if (p2 <= 0.0) {
    lam = q;
  } else {
    const double p = sqrt(p2 / 6.0);
    const double b00 = (a00 - q) / p, b11 = (a11 - q) / p, b22 = (a22 - q) / p;
    const double b01 = a01 / p, b02 = a02 / p, b12 = a12 / p;
    double r = 0.5 * (b00 * (b11 * b22 - b12 * b12) - b01 * (b01 * b22 - b12 * b02) + b02 * (b01 * b12 - b11 * b02));
    r = fmin(1.0, fmax(-1.0, r));
    const double phi = acos(r) / 3.0;
    lam = q + 2.0 * p * cos(phi + 2.0943951023931953);      // smallest eigenvalue
  }
  const double r0[3] = {a00 - lam, a01, a02}, r1[3] = {a01, a11 - lam, a12}, r2[3] = {a02, a12, a22 - lam};
  double c0[3], c1[3], c2[3];
  cross3(r0, r1, c0); cross3(r0, r2, c1); cross3(r1, r2, c2);
  const double n0 = c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2];
  const double n1 = c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2];
  const double n2 = c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2];
  const double* c = c0;
  double nn = n0;
  if (n1 > nn) { c = c1; nn = n1; }
  if (n2 > nn) { c = c2; nn = n2; }
  if (nn <= 1e-300) { v[0] = 1.0; v[1] = 0.0; v[2] = 0.0; return; }    // isotropic neighbourhood: any direction
  const double inv = 1.0 / sqrt(nn);
  v[0] = c[0] * inv; v[1] = c[1] * inv; v[2] = c[2] * inv;
}

template <int K>
__global__ __launch_bounds__(256) void estimate_normals_kernel(const float* __restrict__ pts, float* __restrict__ normals,
                                                               int N) {
  extern __shared__ float nrm_smem[];
  float* xs = nrm_smem;
  float* ys = xs + N;
  float* zs = ys + N;
  const int part = blockIdx.y;
  const float* src = pts + (int64_t)part * N * 3;
  for (int i = threadIdx.x; i < N; i += 256) {
    xs[i] = src[3 * i]; ys[i] = src[3 * i + 1]; zs[i] = src[3 * i + 2];
  }
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const float px = xs[i], py = ys[i], pz = zs[i];
  float bd[K];
  int bi[K];
#pragma unroll
  for (int t = 0; t < K; ++t) { bd[t] = __builtin_huge_valf(); bi[t] = 0; }
  for (int j = 0; j < N; ++j) {
    const float dx = px - xs[j], dy = py - ys[j], dz = pz - zs[j];
    const float d = (dx * dx + dy * dy) + dz * dz;
    if (d < bd[K - 1]) {
      bd[K - 1] = d; bi[K - 1] = j;
#pragma unroll
      for (int t = K - 1; t > 0; --t) {
        if (bd[t] < bd[t - 1]) {
          const float td = bd[t]; bd[t] = bd[t - 1]; bd[t - 1] = td;
          const int ti = bi[t]; bi[t] = bi[t - 1]; bi[t - 1] = ti;
        }
      }
    }
  }
  // neighbourhood mean and covariance (fp64: the smallest eigenvalue of a flat patch is a difference of
  // nearly equal numbers)
  double mx = 0.0, my = 0.0, mz = 0.0;
#pragma unroll
  for (int t = 0; t < K; ++t) { mx += xs[bi[t]]; my += ys[bi[t]]; mz += zs[bi[t]]; }
  mx /= K; my /= K; mz /= K;
  double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
#pragma unroll
  for (int t = 0; t < K; ++t) {
    const double dx = xs[bi[t]] - mx, dy = ys[bi[t]] - my, dz = zs[bi[t]] - mz;
    c00 += dx * dx; c01 += dx * dy; c02 += dx * dz; c11 += dy * dy; c12 += dy * dz; c22 += dz * dz;
  }
  double v[3];
  smallest_eigvec(c00 / K, c01 / K, c02 / K, c11 / K, c12 / K, c22 / K, v);
  // direction: at least half of the neighbours on the positive side (_disambiguate_vector_directions)
  int n_pos = 0;
#pragma unroll
  for (int t = 0; t < K; ++t) {
    const double proj = v[0] * (double)(xs[bi[t]] - px) + v[1] * (double)(ys[bi[t]] - py) + v[2] * (double)(zs[bi[t]] - pz);
    n_pos += proj > 0.0 ? 1 : 0;
  }
  const double sgn = (2 * n_pos < K) ? -1.0 : 1.0;
  float* o = normals + ((int64_t)part * N + i) * 3;
  o[0] = (float)(sgn * v[0]); o[1] = (float)(sgn * v[1]); o[2] = (float)(sgn * v[2]);
}

__global__ __launch_bounds__(256) void merge_keep_mask_kernel(const float* __restrict__ d, const float* __restrict__ normals,
                                                              uint8_t* __restrict__ keep, int P, int N, float thr) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)P * N) return;
  const int i = (int)(idx / N), k = (int)(idx - (int64_t)i * N);
  const float* ni = normals + ((int64_t)i * N + k) * 3;
  bool kp = true;
  for (int j = 0; j < P; ++j) {
    if (j == i) continue;
    const float cd = d[((int64_t)i * P + j) * N + k] + d[((int64_t)j * P + i) * N + k];
    if (cd < thr) {
      const float* nj = normals + ((int64_t)j * N + k) * 3;
      const float dot = (ni[0] * nj[0] + ni[1] * nj[1]) + ni[2] * nj[2];
      if (dot < 0.0f) kp = false;
    }
  }
  keep[idx] = kp ? 1 : 0;
}

}  // namespace

extern "C" int pfpp_estimate_normals(const float* pts, float* normals, int64_t P, int64_t N, int64_t K,
                                     pfpp_stream_t stream) {
  PFPP_REQUIRE(pts && normals, "null pointer");
  PFPP_REQUIRE(P >= 0 && N >= 1 && K >= 3 && K <= N, "need 3 <= K <= N");
  PFPP_SUPPORTED(K == 20 || K == 10 || K == 32, "neighbourhood sizes 10, 20 (the reference's) and 32");
  PFPP_SUPPORTED(N <= 8192 && P <= 65535, "N > 8192 points per part");
  if (P == 0) return PFPP_OK;
  const dim3 grid((unsigned)((N + 255) / 256), (unsigned)P);
  const size_t smem = (size_t)3 * N * sizeof(float);
  hipStream_t st = pfpp::as_stream(stream);
  if (K == 20) hipLaunchKernelGGL(estimate_normals_kernel<20>, grid, dim3(256), smem, st, pts, normals, (int)N);
  else if (K == 10) hipLaunchKernelGGL(estimate_normals_kernel<10>, grid, dim3(256), smem, st, pts, normals, (int)N);
  else hipLaunchKernelGGL(estimate_normals_kernel<32>, grid, dim3(256), smem, st, pts, normals, (int)N);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_merge_keep_mask(const float* d, const float* normals, uint8_t* keep, int64_t P, int64_t N,
                                    float threshold, pfpp_stream_t stream) {
  PFPP_REQUIRE(d && normals && keep, "null pointer");
  PFPP_REQUIRE(P >= 1 && N >= 1, "bad sizes");
  hipLaunchKernelGGL(merge_keep_mask_kernel, dim3((unsigned)((P * N + 255) / 256)), dim3(256), 0, pfpp::as_stream(stream), d,
                     normals, keep, (int)P, (int)N, threshold);
  return pfpp::check_launch(__func__);
}
