// Verifier edge features, the step between denoiser and verifier in
// AutoAgglomerative.test_step:
//   * pfpp_pose_apply_points: get_final_pose_pts_dynamic (utils/node_merge_utils.py:16-41) — the
//     variable-size "by area" clouds of all parts are one flat list, every point carries the index
//     of the pose to apply (its part's pivot); quaternion_apply WITHOUT normalisation + translation.
//   * pfpp_edge_histogram: get_distance_for_matching_pts (:62-89) + _make_cd_to_bins
//     (auto_aggl.py:385-389): for every candidate edge the matched points (a_i, b_i), i < M, give
//     d_i = min_j |a_i - b_j|^2 + min_j |b_i - a_j|^2 (chamferdist, bidirectional, no reduction) which
//     is counted into the bins [0,1e-3) [1e-3,5e-3) [5e-3,1e-2) [1e-2,5e-2) [5e-2,1e-1) [1e-1,100).
// One workgroup per edge, both point sets in LDS, one thread per matched pair.
#include "pfpp_common.h"

namespace {

__global__ __launch_bounds__(256) void pose_apply_points_kernel(const float* __restrict__ pts,
                                                                const int32_t* __restrict__ pose_idx,
                                                                const float* __restrict__ pose,
                                                                float* __restrict__ out, int64_t n, int normalise) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* ps = pose + (int64_t)pose_idx[i] * 7;
  float w = ps[3], x = ps[4], y = ps[5], z = ps[6];
  if (normalise) {
    const float nrm = sqrtf(((w * w + x * x) + y * y) + z * z);
    w = w / nrm; x = x / nrm; y = y / nrm; z = z / nrm;
  }
  const float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
  // t = q * (0, p)
  const float tw = ((w * 0.0f - x * px) - y * py) - z * pz;
  const float tx = ((w * px + x * 0.0f) + y * pz) - z * py;
  const float ty = ((w * py - x * pz) + y * 0.0f) + z * px;
  const float tz = ((w * pz + x * py) - y * px) + z * 0.0f;
  // r = t * conj(q)
  const float cx = -x, cy = -y, cz = -z;
  const float rx = ((tw * cx + tx * w) + ty * cz) - tz * cy;
  const float ry = ((tw * cy - tx * cz) + ty * w) + tz * cx;
  const float rz = ((tw * cz + tx * cy) - ty * cx) + tz * w;
  out[3 * i] = rx + ps[0];
  out[3 * i + 1] = ry + ps[1];
  out[3 * i + 2] = rz + ps[2];
}

__global__ __launch_bounds__(256) void edge_histogram_kernel(const float* __restrict__ pts,
                                                             const int32_t* __restrict__ idx_a,
                                                             const int32_t* __restrict__ idx_b,
                                                             const int32_t* __restrict__ edge_off,
                                                             int32_t* __restrict__ hist, int max_m) {
  extern __shared__ __align__(16) float eh_smem[];
  float* a = eh_smem;                 // [3*max_m]
  float* b = eh_smem + 3 * max_m;     // [3*max_m]
  __shared__ int s_hist[6];
  const int e = blockIdx.x;
  const int o = edge_off[e];
  const int m = edge_off[e + 1] - o;
  const int tid = threadIdx.x;
  if (tid < 6) s_hist[tid] = 0;
  for (int i = tid; i < m; i += 256) {
    const int ia = idx_a[o + i], ib = idx_b[o + i];
    a[3 * i] = pts[3 * ia]; a[3 * i + 1] = pts[3 * ia + 1]; a[3 * i + 2] = pts[3 * ia + 2];
    b[3 * i] = pts[3 * ib]; b[3 * i + 1] = pts[3 * ib + 1]; b[3 * i + 2] = pts[3 * ib + 2];
  }
  __syncthreads();
  for (int i = tid; i < m; i += 256) {
    const float ax = a[3 * i], ay = a[3 * i + 1], az = a[3 * i + 2];
    const float bx = b[3 * i], by = b[3 * i + 1], bz = b[3 * i + 2];
    float fa = __builtin_huge_valf(), fb = __builtin_huge_valf();
    for (int j = 0; j < m; ++j) {
      float dx = ax - b[3 * j], dy = ay - b[3 * j + 1], dz = az - b[3 * j + 2];
      fa = fminf(fa, (dx * dx + dy * dy) + dz * dz);
      dx = bx - a[3 * j]; dy = by - a[3 * j + 1]; dz = bz - a[3 * j + 2];
      fb = fminf(fb, (dx * dx + dy * dy) + dz * dz);
    }
    const float d = fa + fb;
    // torch.bucketize(d, [0,1e-3,5e-3,1e-2,5e-2,1e-1,100], right=True) - 1, values outside dropped
    int bin = -1;
    if (d >= 0.0f) bin = 0;
    if (d >= 1e-3f) bin = 1;
    if (d >= 5e-3f) bin = 2;
    if (d >= 1e-2f) bin = 3;
    if (d >= 5e-2f) bin = 4;
    if (d >= 1e-1f) bin = 5;
    if (d >= 100.0f) bin = -1;
    if (bin >= 0) atomicAdd(&s_hist[bin], 1);
  }
  __syncthreads();
  if (tid < 6) hist[e * 6 + tid] = s_hist[tid];
}

}  // namespace

extern "C" int pfpp_pose_apply_points(const float* pts, const int32_t* pose_idx, const float* pose, float* out,
                                      int64_t n, int normalise, pfpp_stream_t stream) {
  PFPP_REQUIRE(pts && pose_idx && pose && out, "null pointer");
  if (n == 0) return PFPP_OK;
  hipLaunchKernelGGL(pose_apply_points_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     pfpp::as_stream(stream), pts, pose_idx, pose, out, n, normalise);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_edge_histogram(const float* pts, const int32_t* idx_a, const int32_t* idx_b,
                                   const int32_t* edge_off, int32_t* hist, int64_t n_edges, int64_t max_m,
                                   pfpp_stream_t stream) {
  PFPP_REQUIRE(pts && idx_a && idx_b && edge_off && hist, "null pointer");
  PFPP_SUPPORTED(max_m >= 0 && max_m <= 6000, "more than 6000 correspondences on one edge");
  if (n_edges == 0) return PFPP_OK;
  const size_t smem = (size_t)6 * (max_m > 0 ? max_m : 1) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(edge_histogram_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 6000 * 4);
    attr_set = true;
  }
  hipLaunchKernelGGL(edge_histogram_kernel, dim3((unsigned)n_edges), dim3(256), smem, pfpp::as_stream(stream), pts,
                     idx_a, idx_b, edge_off, hist, (int)max_m);
  return pfpp::check_launch(__func__);
}
