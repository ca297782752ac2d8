// Evaluation metrics: the two device-side pieces of
// puzzlefusion_plusplus/denoiser/evaluation/evaluator.py.
//
//  * pfpp_nn_dist — chamferdist's forward term (KNN-1, squared L2, knn_points semantics): for every
//    point of src[b] the squared distance to its nearest neighbour in dst[b].  calc_part_acc
//    (evaluator.py:88-121, per-part clouds of 1000 points) and calc_shape_cd (:124-153, whole shapes of
//    P*N = 20000 points) are this kernel called in both directions.  Brute force with the target set
//    streamed through LDS in tiles of 1024 points; every LDS read is a broadcast (all lanes scan the
//    same target point), 8 FLOP per 12 bytes of LDS — VALU bound, ~N*M/CU-count distance evaluations.
//  * pfpp_quat_to_euler_xyz — transform.quaternion_to_euler (transform.py:70-86): pytorch3d
//    quaternion_to_matrix then matrix_to_euler_angles(convention="XYZ"), optionally in degrees; used by
//    rot_metrics (evaluator.py:53-85).
#include "pfpp_common.h"

namespace {

constexpr int NN_TILE = 1024;

__global__ __launch_bounds__(256) void nn_dist_kernel(const float* __restrict__ src, const float* __restrict__ dst,
                                                      float* __restrict__ out, int64_t n, int64_t m) {
  __shared__ float tx[NN_TILE], ty[NN_TILE], tz[NN_TILE];
  const int64_t b = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const float* s = src + b * n * 3;
  const float* d = dst + b * m * 3;
  const bool ok = i < n;
  const float px = ok ? s[3 * i] : 0.0f, py = ok ? s[3 * i + 1] : 0.0f, pz = ok ? s[3 * i + 2] : 0.0f;
  float best = __builtin_huge_valf();
  for (int64_t j0 = 0; j0 < m; j0 += NN_TILE) {
    const int cnt = (int)min((int64_t)NN_TILE, m - j0);
    __syncthreads();
    for (int j = threadIdx.x; j < cnt; j += 256) {
      tx[j] = d[3 * (j0 + j)]; ty[j] = d[3 * (j0 + j) + 1]; tz[j] = d[3 * (j0 + j) + 2];
    }
    __syncthreads();
#pragma unroll 8
    for (int j = 0; j < cnt; ++j) {
      const float dx = px - tx[j], dy = py - ty[j], dz = pz - tz[j];
      best = fminf(best, (dx * dx + dy * dy) + dz * dz);
    }
  }
  if (ok) out[b * n + i] = best;
}

__global__ __launch_bounds__(256) void quat_to_euler_kernel(const float* __restrict__ q, float* __restrict__ out, int64_t n,
                                                            int to_degree) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float r = q[4 * i], x = q[4 * i + 1], y = q[4 * i + 2], z = q[4 * i + 3];
  // pytorch3d.transforms.quaternion_to_matrix
  const float two_s = 2.0f / (((r * r + x * x) + y * y) + z * z);
  const float m00 = 1.0f - two_s * (y * y + z * z);
  const float m01 = two_s * (x * y - z * r);
  const float m02 = two_s * (x * z + y * r);
  const float m12 = two_s * (y * z - x * r);
  const float m22 = 1.0f - two_s * (x * x + y * y);
  // matrix_to_euler_angles(convention="XYZ"): (atan2(-m12, m22), asin(m02), atan2(-m01, m00))
  float e0 = atan2f(-m12, m22), e1 = asinf(m02), e2 = atan2f(-m01, m00);
  if (to_degree) {
    const float k = 57.29577951308232f;     // torch.rad2deg: x * (180 / pi)
    e0 *= k; e1 *= k; e2 *= k;
  }
  out[3 * i] = e0; out[3 * i + 1] = e1; out[3 * i + 2] = e2;
}

}  // namespace

extern "C" int pfpp_nn_dist(const float* src, const float* dst, float* out, int64_t batch, int64_t n, int64_t m,
                            pfpp_stream_t stream) {
  PFPP_REQUIRE(src && dst && out, "null pointer");
  PFPP_REQUIRE(batch >= 0 && n >= 0 && m >= 1, "bad sizes (the target set must not be empty)");
  PFPP_SUPPORTED(batch <= 65535, "more than 65535 clouds per launch");
  if (batch == 0 || n == 0) return PFPP_OK;
  hipLaunchKernelGGL(nn_dist_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)batch), dim3(256), 0,
                     pfpp::as_stream(stream), src, dst, out, n, m);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_quat_to_euler_xyz(const float* quat, float* euler, int64_t n, int to_degree, pfpp_stream_t stream) {
  PFPP_REQUIRE(quat && euler, "null pointer");
  if (n == 0) return PFPP_OK;
  hipLaunchKernelGGL(quat_to_euler_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, pfpp::as_stream(stream), quat,
                     euler, n, to_degree);
  return pfpp::check_launch(__func__);
}
