// probe: issue rate of v_mfma_f32_32x32x16_f16 on one wave per SIMD against the number of INDEPENDENT accumulator chains the wave
// cycles through (the 64 x 32 wave tile of the token GEMMs has two), with and without filler instructions between them.
//   hipcc --offload-arch=gfx950 -O3 tools/lab/mfma_chain_probe.hip -o tools/lab/_run/mfma_chain && tools/lab/_run/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NC, int FILL, int WPS, bool SMALL>
__global__ __launch_bounds__(256 * WPS) void chains(const half8* __restrict__ in, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  half8 a = in[lane], b = in[64 + lane];
  f32x16 acc[NC];
  f32x4 acs[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[c][e] = 0.0f;
    acs[c] = f32x4{0, 0, 0, 0};
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 24 / NC; ++r)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (SMALL) acs[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acs[c], 0, 0, 0);
        else acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (FILL >= 1) asm volatile("s_nop 0");
        if constexpr (FILL >= 2) asm volatile("v_mov_b32 %0, %0" : "+v"(a[0]));   // not really: keeps a VALU between
      }
  }
  float s = 0.0f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[c][e];
    s += acs[c][0] + acs[c][1] + acs[c][2] + acs[c][3];
  }
  if (s == 1234.5f) out[0] = s;
}

static int g_grid = 256;
template <int NC, int FILL, int WPS, bool SMALL>
void run(const half8* in, float* out) {
  const int iters = 2000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((chains<NC, FILL, WPS, SMALL>), dim3(g_grid), dim3(256 * WPS), 0, 0, in, out, iters);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((chains<NC, FILL, WPS, SMALL>), dim3(g_grid), dim3(256 * WPS), 0, 0, in, out, iters);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double n = (double)iters * (24 / NC) * NC * WPS;     // matrix instructions per SIMD
  const double flops = SMALL ? 16.0 * 16 * 32 * 2 : 32.0 * 32 * 16 * 2;
  printf("grid %3d %s chains %d fill %d waves/SIMD %d: %.1f ns per instruction per SIMD = %.1f cycles at 2.4 GHz; chip %.0f TFLOP/s\n", g_grid, SMALL ? "16x16x32" : "32x32x16", NC, FILL, WPS,
         ms * 1e6 / n, ms * 1e6 / n * 2.4, n * 4 * g_grid * flops / (ms * 1e-3) * 1e-12);
}

int main() {
  half8* in; float* out;
  CK(hipMalloc(&in, 128 * 16)); CK(hipMalloc(&out, 64));
  _Float16 h[1024];
  unsigned s = 12345;
  for (int i = 0; i < 1024; ++i) { s = s * 1664525u + 1013904223u; h[i] = (_Float16)(((s >> 8) & 0xffff) / 32768.0f - 1.0f); }
  CK(hipMemcpy(in, h, 2048, hipMemcpyHostToDevice));
  for (int g : {64, 128, 160, 192, 224, 256}) { g_grid = g; run<4, 0, 1, false>(in, out); run<4, 0, 2, false>(in, out); }
  g_grid = 256;
  run<1, 0, 1, false>(in, out); run<2, 0, 1, false>(in, out); run<3, 0, 1, false>(in, out); run<4, 0, 1, false>(in, out); run<6, 0, 1, false>(in, out); run<8, 0, 1, false>(in, out);
  run<2, 1, 1, false>(in, out); run<2, 2, 1, false>(in, out); run<4, 2, 1, false>(in, out); run<6, 2, 1, false>(in, out);
  run<2, 0, 2, false>(in, out); run<2, 2, 2, false>(in, out); run<4, 0, 2, false>(in, out);
  run<2, 0, 1, true>(in, out); run<4, 0, 1, true>(in, out); run<8, 0, 1, true>(in, out); run<8, 2, 1, true>(in, out);
  return 0;
}
