// a7/a8: vector quantisation of the PointNet++ latents + scatter into the padded
// per-puzzle tensors (gfx950).
//
// VectorQuantizer.forward (vqvae/model/modules/quantizer.py:26-71) builds a
// [F*100, 1024] distance matrix, a one-hot matrix and a one-hot @ codebook
// matmul.  Here one thread owns one 16-wide sub-vector, the 64 KB codebook and
// its squared norms live in LDS (every lane reads the same code at the same time
// -> LDS broadcast, no bank conflicts), the argmin is a running first-minimum
// and the selected code row is read straight back from LDS.
#include "pfpp_common.h"

namespace {

constexpr int VQ_DIM = 16;

// VQ_SPLIT = lanes per sub-vector (4, 16 or 64: the host takes more lanes when there are few sub-vectors — one puzzle in flight has 800,
// with four lanes each that was 13 workgroups scanning 256 codes per lane: 38 us)
template <int VQ_SPLIT>
__global__ __launch_bounds__(256) void vq_encode_kernel(
    const float* __restrict__ z_e, const float* __restrict__ codebook,
    const int32_t* __restrict__ slot, float* __restrict__ z_q, int32_t* __restrict__ codes,
    int64_t total, int rows_per_frag, int n_codes) {
  extern __shared__ __align__(16) float vq_smem[];
  float* s_cb = vq_smem;                       // [n_codes][16]
  float* s_ee = vq_smem + n_codes * VQ_DIM;    // [n_codes]
  const int tid = threadIdx.x;
  for (int i = tid; i < n_codes * VQ_DIM / 4; i += 256)
    reinterpret_cast<float4*>(s_cb)[i] = reinterpret_cast<const float4*>(codebook)[i];
  __syncthreads();
  for (int j = tid; j < n_codes; j += 256) {
    // torch.sum(E**2, dim=1): sequential over the 16 columns
    float s = 0.0f;
#pragma unroll
    for (int d = 0; d < VQ_DIM; ++d) s = __fadd_rn(s, __fmul_rn(s_cb[j * VQ_DIM + d], s_cb[j * VQ_DIM + d]));
    s_ee[j] = s;
  }
  __syncthreads();

  // VQ_SPLIT adjacent lanes share one sub-vector, each scans a contiguous quarter of the codebook; the quarters are in
  // index order and ties go to the lower index, so the merged result is torch.argmin's first minimum (4x the workgroups:
  // at 15 400 sub-vectors one lane per sub-vector used 61 of the 256 CUs)
  const int64_t tidg = (int64_t)blockIdx.x * 256 + tid;
  const int64_t gid = tidg / VQ_SPLIT;
  const int part = (int)(tidg - gid * VQ_SPLIT);
  const bool live = gid < total;
  const int64_t gl = live ? gid : total - 1;
  float z[VQ_DIM];
  const float4* zp = reinterpret_cast<const float4*>(z_e + gl * VQ_DIM);
#pragma unroll
  for (int d4 = 0; d4 < VQ_DIM / 4; ++d4) {
    const float4 v = zp[d4];
    z[4 * d4 + 0] = v.x; z[4 * d4 + 1] = v.y; z[4 * d4 + 2] = v.z; z[4 * d4 + 3] = v.w;
  }
  float zz = 0.0f;
#pragma unroll
  for (int d = 0; d < VQ_DIM; ++d) zz = __fadd_rn(zz, __fmul_rn(z[d], z[d]));

  float best = __builtin_huge_valf();
  const int per = (n_codes + VQ_SPLIT - 1) / VQ_SPLIT;
  const int j0 = part * per, j1 = min(n_codes, j0 + per);
  int bj = j0 < n_codes ? j0 : 0;
  for (int j = j0; j < j1; ++j) {
    const float4* e4 = reinterpret_cast<const float4*>(s_cb + j * VQ_DIM);
    // z @ E^T: k-ordered fma chain (what the fp32 matrix core and the CPU BLAS evaluate)
    float dot = 0.0f;
#pragma unroll
    for (int d4 = 0; d4 < VQ_DIM / 4; ++d4) {
      const float4 e = e4[d4];
      dot = __fmaf_rn(z[4 * d4 + 0], e.x, dot);
      dot = __fmaf_rn(z[4 * d4 + 1], e.y, dot);
      dot = __fmaf_rn(z[4 * d4 + 2], e.z, dot);
      dot = __fmaf_rn(z[4 * d4 + 3], e.w, dot);
    }
    // d = (|z|^2 + |e|^2) - 2*(z.e)   (quantizer.py:45-47)
    const float d = __fsub_rn(__fadd_rn(zz, s_ee[j]), __fmul_rn(2.0f, dot));
    if (d < best) { best = d; bj = j; }   // first minimum, like torch.argmin
  }

#pragma unroll
  for (int m = 1; m < VQ_SPLIT; m <<= 1) {
    const float ob = __shfl_xor(best, m);
    const int oj = __shfl_xor(bj, m);
    if (ob < best || (ob == best && oj < bj)) { best = ob; bj = oj; }
  }
  if (!live || part != 0) return;
  // scatter: sub-vector gid belongs to fragment f, row r, part c of the latent
  const int sub_per_frag = rows_per_frag;   // rows are already the 16-wide sub-vectors
  const int64_t f = gid / sub_per_frag;
  const int64_t r = gid - f * sub_per_frag;
  float* o = z_q + ((int64_t)slot[f] * sub_per_frag + r) * VQ_DIM;
  const float* e = s_cb + bj * VQ_DIM;
#pragma unroll
  for (int d4 = 0; d4 < VQ_DIM / 4; ++d4) {
    float4 v;
    // z + (z_q - z): the straight-through value, rounded like the reference (quantizer.py:63)
    v.x = __fadd_rn(z[4 * d4 + 0], __fsub_rn(e[4 * d4 + 0], z[4 * d4 + 0]));
    v.y = __fadd_rn(z[4 * d4 + 1], __fsub_rn(e[4 * d4 + 1], z[4 * d4 + 1]));
    v.z = __fadd_rn(z[4 * d4 + 2], __fsub_rn(e[4 * d4 + 2], z[4 * d4 + 2]));
    v.w = __fadd_rn(z[4 * d4 + 3], __fsub_rn(e[4 * d4 + 3], z[4 * d4 + 3]));
    reinterpret_cast<float4*>(o)[d4] = v;
  }
  if (codes) codes[gid] = bj;
}

__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ in,
                                                           const int32_t* __restrict__ slot,
                                                           float* __restrict__ out, int64_t total,
                                                           int row_elems) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int64_t f = gid / row_elems;
  const int64_t c = gid - f * row_elems;
  out[(int64_t)slot[f] * row_elems + c] = in[gid];
}

}  // namespace

extern "C" int pfpp_vq_encode(const float* z_e, const float* codebook, const int32_t* slot,
                              float* z_q, int32_t* codes, int64_t F, int64_t rows_per_frag,
                              int64_t dim, int64_t n_codes, pfpp_stream_t stream) {
  PFPP_REQUIRE(z_e && codebook && slot && z_q, "null pointer");
  PFPP_SUPPORTED(dim == VQ_DIM, "embedding_dim != 16");
  PFPP_SUPPORTED(n_codes >= 1 && n_codes <= 2048, "n_codes outside [1, 2048]");
  PFPP_REQUIRE(pfpp::aligned16(z_e) && pfpp::aligned16(codebook) && pfpp::aligned16(z_q),
               "16-byte alignment");
  const int64_t total = F * rows_per_frag;
  if (total == 0) return PFPP_OK;
  const size_t smem = (size_t)n_codes * (VQ_DIM + 1) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(vq_encode_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 2048 * (VQ_DIM + 1) * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(vq_encode_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 2048 * (VQ_DIM + 1) * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(vq_encode_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 2048 * (VQ_DIM + 1) * 4);
    attr_set = true;
  }
  hipStream_t st = pfpp::as_stream(stream);
  // lanes per sub-vector: enough of them that the launch fills the chip (the argmin is the first minimum for every choice)
  if (total <= 2048)
    hipLaunchKernelGGL(vq_encode_kernel<64>, dim3((unsigned)((total * 64 + 255) / 256)), dim3(256), smem, st, z_e, codebook, slot, z_q, codes, total,
                       (int)rows_per_frag, (int)n_codes);
  else if (total <= 8192)
    hipLaunchKernelGGL(vq_encode_kernel<16>, dim3((unsigned)((total * 16 + 255) / 256)), dim3(256), smem, st, z_e, codebook, slot, z_q, codes, total,
                       (int)rows_per_frag, (int)n_codes);
  else
    hipLaunchKernelGGL(vq_encode_kernel<4>, dim3((unsigned)((total * 4 + 255) / 256)), dim3(256), smem, st, z_e, codebook, slot, z_q, codes, total,
                       (int)rows_per_frag, (int)n_codes);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_scatter_rows(const float* in, const int32_t* slot, float* out, int64_t F,
                                 int64_t row_elems, pfpp_stream_t stream) {
  PFPP_REQUIRE(in && slot && out, "null pointer");
  const int64_t total = F * row_elems;
  if (total == 0) return PFPP_OK;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     pfpp::as_stream(stream), in, slot, out, total, (int)row_elems);
  return pfpp::check_launch(__func__);
}
