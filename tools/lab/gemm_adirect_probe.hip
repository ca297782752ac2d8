// probe: a split-f16 GEMM main loop whose ACTIVATION operand never touches LDS.  C[M, N] = A[M, K] . W[N, K]^T with both operands stored
// fragment-blocked ([rows / 32][K / 16][64 lanes][8 halfs]: one MFMA operand fragment = one contiguous 1 KB, hi and lo planes apart).
// Workgroup = 4 waves = 128 x 64 of C; wave w owns rows 32 w .. 32 w + 31 and both 32-column tiles: its A fragments come straight from
// global memory (two 1 KB loads per 16-deep step, three K-tiles ahead), the weights go through a three-deep LDS ring shared by the waves.
// Per 32-deep K-tile and CU: 48 MFMAs (384 cycles / SIMD), 40 KB through LDS (the product kernel: 72 KB), 24 KB through the texture path.
//   hipcc --offload-arch=gfx950 -O3 tools/lab/gemm_adirect_probe.hip -o tools/lab/_bin/gemm_adirect_probe && tools/lab/_bin/gemm_adirect_probe 3850 512 512
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef DEPTH
#define DEPTH 3    // K-tiles in the pipeline (register slots of the A fragments = LDS stages of the weight ring); DEPTH - 1 are in flight
#endif
#ifndef SKEW
#define SKEW 0     // 1: every workgroup starts its K walk at a different K-tile (wraps around): concurrent workgroups do not ask for the same lines
#endif
#ifndef XCD
#define XCD 0      // 1: 1-D grid, workgroup ids dealt to the 8 XCDs round-robin are remapped so that an XCD owns consecutive tiles (row-major: the
#endif             //    column tiles of a row panel share an XCD's L2)
#ifndef ABL
#define ABL 0      // ablations (probe only): 1 no MFMA, 2 no A loads after the first, 4 no weight loads / staging, 8 no barrier, 16 term-major MFMA order, 32 no epilogue stores
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// blocked plane: fragment (rb, kb) at ((rb * KB + kb) * 64 + lane) * 8 halfs; lane = 32 * lhi + l31 holds row 32 rb + l31, k = 16 kb + 8 lhi + 0..7
__global__ __launch_bounds__(256) void adirect_kernel(const _Float16* __restrict__ a_hi, const _Float16* __restrict__ a_lo,
                                                      const _Float16* __restrict__ w_hi, const _Float16* __restrict__ w_lo,
                                                      float* __restrict__ C, int M, int N, int K) {
  __shared__ __align__(16) half8 ring[DEPTH][2][2][2][64];          // [stage][col tile][k16 step][plane][lane]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int KB = K / 16, NKT = K / 32;
#if XCD
  const int tiles_n = N / 64, nwg = gridDim.x;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, q_ = nwg >> 3, r_ = nwg & 7;
  const int tile = (xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_) + local;
  const int bx = tile / tiles_n, by = tile - bx * tiles_n;
#else
  const int bx = blockIdx.x, by = blockIdx.y;
#endif
  const int mb = bx * 4 + wave;                                  // this wave's 32-row block
  const int nb0 = by * 2;                                        // first of the two 32-column blocks
  const int mbc = mb < (M + 31) / 32 ? mb : (M + 31) / 32 - 1;
  const half8* ah = reinterpret_cast<const half8*>(a_hi) + (size_t)mbc * KB * 64 + lane;
  const half8* al = reinterpret_cast<const half8*>(a_lo) + (size_t)mbc * KB * 64 + lane;
  // weight tile of a K-tile: 8 fragments (col tile, step, plane) of 1 KB; thread t copies 16 bytes of fragments 2 (t / 64) and + 1... laid out so
  // that thread t handles fragments f0 = t / 64 and f0 + 4 (lane-contiguous 1 KB each)
  const int wf = tid >> 6;                                       // 0..3 -> (col tile, step) = (wf >> 1, wf & 1); plane 0 then plane 1
  const half8* wsrc_h = reinterpret_cast<const half8*>(w_hi) + ((size_t)(nb0 + (wf >> 1)) * KB + (wf & 1)) * 64 + lane;
  const half8* wsrc_l = reinterpret_cast<const half8*>(w_lo) + ((size_t)(nb0 + (wf >> 1)) * KB + (wf & 1)) * 64 + lane;

  const int skew = SKEW == 1 ? (int)((bx * 5u + by * 3u) % (unsigned)NKT) : SKEW == 2 ? (int)((by * 2u + (bx & 1u)) % (unsigned)NKT) : 0;
  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;

  half8 afh[DEPTH][2], afl[DEPTH][2], wrh[DEPTH], wrl[DEPTH];
  auto issue = [&](int kt, int slot) {                            // loads of K-tile kt into register slot
    int ktc = kt < NKT ? kt : NKT - 1;
    if (SKEW) { ktc += skew; ktc = ktc >= NKT ? ktc - NKT : ktc; }
    if (!(ABL & 2) || kt < 3) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        afh[slot][s] = ah[(size_t)(2 * ktc + s) * 64];
        afl[slot][s] = al[(size_t)(2 * ktc + s) * 64];
      }
    }
    if (!(ABL & 4) || kt < 3) {
      wrh[slot] = wsrc_h[(size_t)(2 * ktc) * 64];
      wrl[slot] = wsrc_l[(size_t)(2 * ktc) * 64];
    }
  };
  auto stage = [&](int slot) {                                    // this thread's two weight pieces -> LDS stage `slot`
    if (ABL & 4) return;
    ring[slot][wf >> 1][wf & 1][0][lane] = wrh[slot];
    ring[slot][wf >> 1][wf & 1][1][lane] = wrl[slot];
  };
  auto compute = [&](int slot) {
    if (ABL & 16) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        half8 bh[2], bl[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) { bh[j] = ring[slot][j][s][0][lane]; bl[j] = ring[slot][j][s][1][lane]; }
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afl[slot][s], bh[j], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afh[slot][s], bl[j], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afh[slot][s], bh[j], acc[j], 0, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const half8 bh = ring[slot][j][s][0][lane], bl = ring[slot][j][s][1][lane];
        if (ABL & 1) { acc[j][0] += (float)bh[0] + (float)bl[0] + (float)afl[slot][s][0] + (float)afh[slot][s][0]; continue; }
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afl[slot][s], bh, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afh[slot][s], bl, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afh[slot][s], bh, acc[j], 0, 0, 0);
      }
    }
  };
#pragma unroll
  for (int u = 0; u < DEPTH - 1; ++u) issue(u, u);
  stage(0);
  __syncthreads();
  // K-tile kt computes from slot kt % DEPTH while tile kt + DEPTH - 1 is requested and tile kt + 1 moves into LDS
  int kt = 0;
  for (; kt + DEPTH <= NKT; kt += DEPTH) {
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      issue(kt + u + DEPTH - 1, (u + DEPTH - 1) % DEPTH);
      stage((u + 1) % DEPTH);
      compute(u);
      if (!(ABL & 8)) __syncthreads();
    }
  }
#pragma unroll
  for (int u = 0; u < DEPTH - 1; ++u) {
    if (kt + u < NKT) {
      issue(kt + u + DEPTH - 1, (u + DEPTH - 1) % DEPTH);
      stage((u + 1) % DEPTH);
      compute(u);
      if (!(ABL & 8)) __syncthreads();
    }
  }
  // plain epilogue (probe): lane = column, register e = row (e & 3) + 8 (e >> 2) + 4 lhi
  const int l31 = lane & 31, lhi = lane >> 5;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = (nb0 + j) * 32 + l31;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = mb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
      if ((ABL & 32) ? (acc[j][e] == 12345.678f) : (row < M && col < N)) C[(size_t)row * N + col] = acc[j][e];      // 32: no stores
    }
  }
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 3850, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 512;
  const int MB = (M + 31) / 32, NB = N / 32, KB = K / 16;
  std::vector<float> A((size_t)MB * 32 * K, 0.0f), W((size_t)N * K);
  unsigned s = 777;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (int m = 0; m < M; ++m) for (int k = 0; k < K; ++k) A[(size_t)m * K + k] = rnd();
  for (auto& v : W) v = rnd() / sqrtf((float)K);
  auto block = [&](const std::vector<float>& X, int RB, std::vector<_Float16>& hi, std::vector<_Float16>& lo) {
    hi.resize((size_t)RB * KB * 512); lo.resize(hi.size());
    for (int rb = 0; rb < RB; ++rb) for (int kb = 0; kb < KB; ++kb) for (int ln = 0; ln < 64; ++ln) for (int q = 0; q < 8; ++q) {
      const float x = X[(size_t)(rb * 32 + (ln & 31)) * K + kb * 16 + (ln >> 5) * 8 + q];
      const _Float16 h = (_Float16)x;
      const size_t o = (((size_t)rb * KB + kb) * 64 + ln) * 8 + q;
      hi[o] = h; lo[o] = (_Float16)(x - (float)h);
    }
  };
  std::vector<_Float16> ahi, alo, whi, wlo;
  block(A, MB, ahi, alo); block(W, NB, whi, wlo);
  _Float16 *d_ah, *d_al, *d_wh, *d_wl; float* d_c;
  CK(hipMalloc(&d_ah, ahi.size() * 2)); CK(hipMalloc(&d_al, alo.size() * 2)); CK(hipMalloc(&d_wh, whi.size() * 2)); CK(hipMalloc(&d_wl, wlo.size() * 2));
  CK(hipMalloc(&d_c, (size_t)M * N * 4));
  CK(hipMemcpy(d_ah, ahi.data(), ahi.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d_al, alo.data(), alo.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_wh, whi.data(), whi.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d_wl, wlo.data(), wlo.size() * 2, hipMemcpyHostToDevice));
#if XCD
  const dim3 grid(((MB + 3) / 4) * (NB / 2), 1);
#else
  const dim3 grid((MB + 3) / 4, NB / 2);
#endif
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(adirect_kernel, grid, dim3(256), 0, 0, d_ah, d_al, d_wh, d_wl, d_c, M, N, K);
  CK(hipEventRecord(e0));
  const int it = 50;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL(adirect_kernel, grid, dim3(256), 0, 0, d_ah, d_al, d_wh, d_wl, d_c, M, N, K);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<float> Cc((size_t)M * N);
  CK(hipMemcpy(Cc.data(), d_c, Cc.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0.0, scale = 0.0;
  for (int t = 0; t < 64; ++t) {
    const int m = (int)((unsigned)(t * 2654435761u) % M), n = (int)((unsigned)(t * 40503u + 17) % N);
    double ref = 0.0;
    for (int k = 0; k < K; ++k) ref += (double)A[(size_t)m * K + k] * (double)W[(size_t)n * K + k];
    worst = fmax(worst, fabs(ref - Cc[(size_t)m * N + n])); scale = fmax(scale, fabs(ref));
  }
  const double us = ms / it * 1e3;
  printf("M %d N %d K %d: %.1f us per launch, %.1f TFLOP/s (fp32-grade), %d workgroups, max |err| %.2e of %.2e\n", M, N, K, us,
         2.0 * M * N * K / us * 1e-6, grid.x * grid.y, worst, scale);
  return 0;
}
