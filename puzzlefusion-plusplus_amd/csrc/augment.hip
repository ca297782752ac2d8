// GPU-side augmentation of GeometryLatentDataset.__getitem__ (SURVEY.md §8f rank 4;
// puzzlefusion_plusplus/denoiser/dataset/dataset.py:165-215) for a whole batch at once:
//     pts  = R(q_g)^T . part_pcs_gt                      random rotation of the whole assembly (:134-146)
//     pts -= centroid(pts[ref])                          recentre on the reference part       (:148-157)
//     per part:  t = centroid(pts[p]);  z = R(q_p)^T (pts[p] - t)                             (:112-132)
//     scale = max |z| (1 if 0);  part_pcs = z / scale                                         (:205-208)
// q_g / q_p are the STORED quaternions (pose_gt_r, part_rots: scalar first, they take the canonical data back), so
// the rotation applied here is their inverse.  The reference does this in float64 numpy per sample in the
// DataLoader workers and casts to float32 at the end; the kernel keeps the float64 arithmetic (means,
// rotations) and the float32 max / division, one workgroup per fragment, points held in registers.
#include "pfpp_common.h"

namespace {

__device__ __forceinline__ void quat_to_mat_t(const float* q, double* m) {
  // rows of R(q)^T = columns of pytorch3d/scipy's R(q); q need not be exactly unit (two_s normalises)
  const double r = q[0], i = q[1], j = q[2], k = q[3];
  const double two_s = 2.0 / (r * r + i * i + j * j + k * k);
  const double R00 = 1 - two_s * (j * j + k * k), R01 = two_s * (i * j - k * r), R02 = two_s * (i * k + j * r);
  const double R10 = two_s * (i * j + k * r), R11 = 1 - two_s * (i * i + k * k), R12 = two_s * (j * k - i * r);
  const double R20 = two_s * (i * k - j * r), R21 = two_s * (j * k + i * r), R22 = 1 - two_s * (i * i + j * j);
  m[0] = R00; m[1] = R10; m[2] = R20;
  m[3] = R01; m[4] = R11; m[5] = R21;
  m[6] = R02; m[7] = R12; m[8] = R22;
}

__device__ __forceinline__ double block_sum(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// raw centroid of every fragment in fp64: cent[b,p,:]
__global__ __launch_bounds__(256) void frag_centroid_kernel(const float* __restrict__ gt, double* __restrict__ cent, int N) {
  __shared__ double red[4];
  const int64_t f = blockIdx.x;
  const float* src = gt + f * N * 3;
  double sx = 0, sy = 0, sz = 0;
  for (int i = threadIdx.x; i < N; i += 256) { sx += src[3 * i]; sy += src[3 * i + 1]; sz += src[3 * i + 2]; }
  sx = block_sum(sx, red); sy = block_sum(sy, red); sz = block_sum(sz, red);
  if (threadIdx.x == 0) { cent[3 * f] = sx / N; cent[3 * f + 1] = sy / N; cent[3 * f + 2] = sz / N; }
}

template <int PPT>
__global__ __launch_bounds__(256) void frag_prepare_kernel(const float* __restrict__ gt, const double* __restrict__ cent,
                                                           const int32_t* __restrict__ num_parts,
                                                           const int32_t* __restrict__ ref_idx, const float* __restrict__ q_g,
                                                           const float* __restrict__ q_p, float* __restrict__ part_pcs,
                                                           float* __restrict__ part_trans, float* __restrict__ part_scale,
                                                           float* __restrict__ init_t, int P, int N) {
  __shared__ float redf[4];
  const int b = blockIdx.y, p = blockIdx.x;
  const int64_t f = (int64_t)b * P + p;
  float* out = part_pcs + f * N * 3;
  if (p >= num_parts[b]) {                      // padded slot: zeros, scale 1 (dataset.py:158-163, :206)
    for (int i = threadIdx.x; i < 3 * N; i += 256) out[i] = 0.0f;
    if (threadIdx.x == 0) {
      part_trans[3 * f] = part_trans[3 * f + 1] = part_trans[3 * f + 2] = 0.0f;
      part_scale[f] = 1.0f;
    }
    return;
  }
  double Rg[9], Rp[9];
  quat_to_mat_t(q_g + 4 * b, Rg);
  quat_to_mat_t(q_p + 4 * f, Rp);
  const double* cr = cent + ((int64_t)b * P + ref_idx[b]) * 3;
  const double* cp = cent + f * 3;
  // centroid of the reference part and of this part after the global rotation
  const double c_ref[3] = {Rg[0] * cr[0] + Rg[1] * cr[1] + Rg[2] * cr[2], Rg[3] * cr[0] + Rg[4] * cr[1] + Rg[5] * cr[2],
                           Rg[6] * cr[0] + Rg[7] * cr[1] + Rg[8] * cr[2]};
  const double c_p[3] = {Rg[0] * cp[0] + Rg[1] * cp[1] + Rg[2] * cp[2], Rg[3] * cp[0] + Rg[4] * cp[1] + Rg[5] * cp[2],
                         Rg[6] * cp[0] + Rg[7] * cp[1] + Rg[8] * cp[2]};
  const double t[3] = {c_p[0] - c_ref[0], c_p[1] - c_ref[1], c_p[2] - c_ref[2]};
  const float* src = gt + f * N * 3;
  float z[PPT][3];
  float mx = 0.0f;
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int i = threadIdx.x + 256 * k;
    z[k][0] = z[k][1] = z[k][2] = 0.0f;
    if (i < N) {
      const double x0 = src[3 * i], x1 = src[3 * i + 1], x2 = src[3 * i + 2];
      // y = Rg^T x - c_ref - t  (= the point relative to its part's centroid, in the rotated assembly frame)
      const double y0 = (Rg[0] * x0 + Rg[1] * x1 + Rg[2] * x2) - c_ref[0] - t[0];
      const double y1 = (Rg[3] * x0 + Rg[4] * x1 + Rg[5] * x2) - c_ref[1] - t[1];
      const double y2 = (Rg[6] * x0 + Rg[7] * x1 + Rg[8] * x2) - c_ref[2] - t[2];
      z[k][0] = (float)(Rp[0] * y0 + Rp[1] * y1 + Rp[2] * y2);
      z[k][1] = (float)(Rp[3] * y0 + Rp[4] * y1 + Rp[5] * y2);
      z[k][2] = (float)(Rp[6] * y0 + Rp[7] * y1 + Rp[8] * y2);
      mx = fmaxf(mx, fmaxf(fabsf(z[k][0]), fmaxf(fabsf(z[k][1]), fabsf(z[k][2]))));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((threadIdx.x & 63) == 0) redf[threadIdx.x >> 6] = mx;
  __syncthreads();
  float scale = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
  if (scale == 0.0f) scale = 1.0f;
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int i = threadIdx.x + 256 * k;
    if (i < N) { out[3 * i] = z[k][0] / scale; out[3 * i + 1] = z[k][1] / scale; out[3 * i + 2] = z[k][2] / scale; }
  }
  if (threadIdx.x == 0) {
    part_trans[3 * f] = (float)t[0]; part_trans[3 * f + 1] = (float)t[1]; part_trans[3 * f + 2] = (float)t[2];
    part_scale[f] = scale;
    if (p == 0) { init_t[3 * b] = (float)c_ref[0]; init_t[3 * b + 1] = (float)c_ref[1]; init_t[3 * b + 2] = (float)c_ref[2]; }
  }
}

}  // namespace

extern "C" int64_t pfpp_fragment_prepare_workspace(int64_t B, int64_t P) { return B * P * 3 * (int64_t)sizeof(double); }

extern "C" int pfpp_fragment_prepare(const float* part_pcs_gt, const int32_t* num_parts, const int32_t* ref_idx,
                                     const float* q_global, const float* q_part, float* part_pcs, float* part_trans,
                                     float* part_scale, float* init_pose_t, int64_t B, int64_t P, int64_t N,
                                     void* workspace, pfpp_stream_t stream) {
  PFPP_REQUIRE(part_pcs_gt && num_parts && ref_idx && q_global && q_part && part_pcs && part_trans && part_scale && init_pose_t &&
               workspace, "null pointer");
  PFPP_REQUIRE(B >= 0 && P >= 1 && N >= 1, "bad sizes");
  PFPP_SUPPORTED(N <= 2048 && B <= 65535, "N > 2048 points per fragment");
  if (B == 0) return PFPP_OK;
  hipStream_t st = pfpp::as_stream(stream);
  double* cent = (double*)workspace;
  hipLaunchKernelGGL(frag_centroid_kernel, dim3((unsigned)(B * P)), dim3(256), 0, st, part_pcs_gt, cent, (int)N);
  const dim3 grid((unsigned)P, (unsigned)B);
  if (N <= 1024)
    hipLaunchKernelGGL(frag_prepare_kernel<4>, grid, dim3(256), 0, st, part_pcs_gt, cent, num_parts, ref_idx, q_global, q_part,
                       part_pcs, part_trans, part_scale, init_pose_t, (int)P, (int)N);
  else
    hipLaunchKernelGGL(frag_prepare_kernel<8>, grid, dim3(256), 0, st, part_pcs_gt, cent, num_parts, ref_idx, q_global, q_part,
                       part_pcs, part_trans, part_scale, init_pose_t, (int)P, (int)N);
  return pfpp::check_launch(__func__);
}
