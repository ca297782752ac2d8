// How many bytes per clock can ONE workgroup per CU pull from L2/HBM with the GEMM's load pattern?
// hipcc --offload-arch=gfx950 -O3 tools/diag/load_bw.hip -o /tmp/load_bw && /tmp/load_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int DEPTH, int NTHR>
__global__ __launch_bounds__(NTHR) void stream_kernel(const float* __restrict__ A, int64_t lda, int rows_per_wg, int K,
                                                      float* __restrict__ out) {
  // tile: rows_per_wg rows x 32 floats per step, thread -> (row = tid>>3 (+ NTHR/8 * it), c4 = tid&7)
  const int tid = threadIdx.x;
  const float* base = A + (int64_t)blockIdx.x * rows_per_wg * lda + (tid & 7) * 4;
  const int its = rows_per_wg / (NTHR / 8);
  float4 acc = make_float4(0, 0, 0, 0);
  for (int k0 = 0; k0 < K; k0 += 32 * DEPTH) {
    float4 r[DEPTH][8];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int it = 0; it < 8; ++it)
        if (it < its) r[d][it] = *reinterpret_cast<const float4*>(base + (int64_t)((tid >> 3) + (NTHR / 8) * it) * lda + k0 + 32 * d);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int it = 0; it < 8; ++it)
        if (it < its) { acc.x += r[d][it].x; acc.y += r[d][it].y; acc.z += r[d][it].z; acc.w += r[d][it].w; }
  }
  if (acc.x == 123.456f) out[0] = acc.y + acc.z + acc.w;
}

template <int DEPTH, int NTHR>
void run(const float* A, int64_t lda, int rows, int K, int rows_per_wg, float* out, const char* tag) {
  const int wgs = rows / rows_per_wg;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream_kernel<DEPTH, NTHR>), dim3(wgs), dim3(NTHR), 0, 0, A, lda, rows_per_wg, K, out);
  hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream_kernel<DEPTH, NTHR>), dim3(wgs), dim3(NTHR), 0, 0, A, lda, rows_per_wg, K, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)rows * K * 4.0 * reps;
  const double s = ms * 1e-3;
  printf("%-28s wgs %5d threads %3d depth %d: %7.2f TB/s  = %6.1f B/clk/WG (2.4 GHz)\n", tag, wgs, NTHR, DEPTH, bytes / s / 1e12,
         bytes / s / 2.4e9 / wgs);
}

int main() {
  const int K = 4096;
  const int64_t lda = K;
  const int rows = 256 * 128 * 4;     // 131072 rows x 4096 floats = 2 GB
  float *A, *out;
  hipMalloc(&A, (size_t)rows * K * 4);
  hipMalloc(&out, 64);
  hipMemset(A, 0, (size_t)rows * K * 4);
  // (1) one WG per CU (256 WGs), 128-row panels: the GEMM's A-operand pattern
  run<1, 256>(A, lda, 256 * 128, K, 128, out, "256 WGs, 128 rows");
  run<2, 256>(A, lda, 256 * 128, K, 128, out, "256 WGs, 128 rows");
  run<4, 256>(A, lda, 256 * 128, K, 128, out, "256 WGs, 128 rows");
  run<1, 512>(A, lda, 256 * 128, K, 128, out, "256 WGs, 128 rows");
  run<2, 512>(A, lda, 256 * 128, K, 256, out, "128 WGs, 256 rows");
  // (2) many WGs
  run<1, 256>(A, lda, rows, K, 128, out, "1024 WGs, 128 rows");
  run<2, 256>(A, lda, rows, K, 128, out, "1024 WGs, 128 rows");
  run<2, 256>(A, lda, rows, K, 32, out, "4096 WGs, 32 rows");
  // (3) small working set (L2 resident: 48 WGs x 128 rows x 4096 = 100 MB no; use K=512: 48*128*512*4 = 12.6 MB)
  run<2, 256>(A, 512, 48 * 128, 512, 128, out, "48 WGs, K=512 (L2/MALL)");
  run<2, 256>(A, 512, 256 * 128, 512, 128, out, "256 WGs, K=512 (MALL)");
  return 0;
}
