// probe for tools/lab/gemm_ws.h (wave-specialised split-f16 GEMM: 4 consumer + 4 loader waves, LDS full / empty counters, 64-deep stages).
//   hipcc --offload-arch=gfx950 -O3 -Itools/lab -DABL=0 tools/lab/gemm_ws_probe.hip -o tools/lab/_bin/ws_a0
//   tools/lab/_bin/ws_a0 3850 512 512 [grid]
// prints the time per launch, the max error against float64 on 256 sampled outputs and an FNV hash of the whole result (the W-direct
// probe prints the same hash for the same seed: bit identity of the two kernels)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gemm_ws.h"

#ifndef ABL
#define ABL 0
#endif
#ifndef PRIO
#define PRIO 0
#endif
#ifndef PFD
#define PFD 0      // prefetch distance in stages (0 = no prefetch wave)
#endif
#ifndef NLD
#define NLD 4      // loader waves
#endif
#ifndef LPRIO
#define LPRIO 0
#endif
#define NTHR ((4 + NLD + (PFD > 0 ? 1 : 0)) * 64)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

using namespace pfpp_ws;

__global__ __launch_bounds__(NTHR) void ws_kernel(const WsP p) {
  extern __shared__ __align__(1024) char smem[];
  if (PRIO > 0 && threadIdx.x < 256) __builtin_amdgcn_s_setprio(PRIO);
  if (LPRIO > 0 && threadIdx.x >= 256) __builtin_amdgcn_s_setprio(LPRIO);
  gemm_ws_body<ABL, PFD, NLD>(p, smem);
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 3850, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 512;
  int grid = argc > 4 ? atoi(argv[4]) : 256;
  const int with_res = argc > 5 ? atoi(argv[5]) : 0;
  if (N % 128 || K % 64) { printf("N %% 128 or K %% 64\n"); return 1; }
  const int NB = N / 32, KB = K / 16;
  std::vector<float> A((size_t)M * K), W((size_t)N * K), R((size_t)M * N), Bv(N);
  unsigned s = 777;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : A) v = rnd();
  for (auto& v : W) v = rnd() / sqrtf((float)K);
  for (auto& v : R) v = with_res ? rnd() : 0.0f;
  for (auto& v : Bv) v = with_res ? rnd() : 0.0f;
  std::vector<_Float16> ahi(A.size()), alo(A.size()), whi((size_t)NB * KB * 512), wlo(whi.size());
  for (size_t i = 0; i < A.size(); ++i) { const _Float16 h = (_Float16)A[i]; ahi[i] = h; alo[i] = (_Float16)(A[i] - (float)h); }
  for (int rb = 0; rb < NB; ++rb) for (int kb = 0; kb < KB; ++kb) for (int ln = 0; ln < 64; ++ln) for (int q = 0; q < 8; ++q) {
    const float x = W[(size_t)(rb * 32 + (ln & 31)) * K + kb * 16 + (ln >> 5) * 8 + q];
    const _Float16 h = (_Float16)x;
    const size_t o = (((size_t)rb * KB + kb) * 64 + ln) * 8 + q;
    whi[o] = h; wlo[o] = (_Float16)(x - (float)h);
  }
  _Float16 *d_ah, *d_al, *d_wh, *d_wl; float *d_c, *d_r, *d_b;
  CK(hipMalloc(&d_ah, ahi.size() * 2)); CK(hipMalloc(&d_al, alo.size() * 2)); CK(hipMalloc(&d_wh, whi.size() * 2)); CK(hipMalloc(&d_wl, wlo.size() * 2));
  CK(hipMalloc(&d_c, (size_t)M * N * 4)); CK(hipMalloc(&d_r, (size_t)M * N * 4)); CK(hipMalloc(&d_b, N * 4));
  CK(hipMemcpy(d_ah, ahi.data(), ahi.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d_al, alo.data(), alo.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_wh, whi.data(), whi.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d_wl, wlo.data(), wlo.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_r, R.data(), R.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_b, Bv.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemset(d_c, 0xff, (size_t)M * N * 4));
  WsP p;
  p.ah = d_ah; p.al = d_al; p.lda = K; p.fh = (const half8*)d_wh; p.fl = (const half8*)d_wl; p.alpha = 1.0f;
  p.bias = with_res ? d_b : nullptr; p.res = with_res ? d_r : nullptr; p.ldr = N; p.out = d_c; p.ldc = N; p.M = M; p.N = N; p.K = K;
  p.tiles = ((M + BM - 1) / BM) * (N / BN);
  if (grid > p.tiles) grid = p.tiles;
  CK(hipFuncSetAttribute((const void*)ws_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&]() { hipLaunchKernelGGL(ws_kernel, dim3(grid), dim3(NTHR), SMEM, 0, p); };
  launch();
  CK(hipDeviceSynchronize());
  std::vector<float> Cc((size_t)M * N);
  CK(hipMemcpy(Cc.data(), d_c, Cc.size() * 4, hipMemcpyDeviceToHost));
  for (int i = 0; i < 5; ++i) launch();
  CK(hipEventRecord(e0));
  const int it = 50;
  for (int i = 0; i < it; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double worst = 0.0, scale = 0.0;
  for (int t = 0; t < 256; ++t) {
    const int m = t < 8 ? M - 1 - t : (int)((unsigned)(t * 2654435761u) % M), n = (int)((unsigned)(t * 40503u + 17) % N);
    double ref = 0.0;
    for (int k = 0; k < K; ++k) ref += (double)A[(size_t)m * K + k] * (double)W[(size_t)n * K + k];
    ref += Bv[n]; ref += R[(size_t)m * N + n];
    worst = fmax(worst, fabs(ref - Cc[(size_t)m * N + n])); scale = fmax(scale, fabs(ref));
  }
  unsigned long long h = 1469598103934665603ull;
  for (size_t i = 0; i < Cc.size(); ++i) { unsigned u; memcpy(&u, &Cc[i], 4); h = (h ^ u) * 1099511628211ull; }
  const double us = ms / it * 1e3;
  printf("WS ABL %d PRIO %d PF %d NL %d LPRIO %d grid %d | M %d N %d K %d: %.1f us per launch, %.1f TFLOP/s (fp32-grade), %d tiles, LDS %d B, max |err| %.2e of %.2e, hash %016llx\n",
         ABL, PRIO, PFD, NLD, LPRIO, grid, M, N, K, us, 2.0 * M * N * K / us * 1e-6, p.tiles, SMEM, worst, scale, h);
  return 0;
}
