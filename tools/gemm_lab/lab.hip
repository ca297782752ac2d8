// Developer bench for the split-f16 GEMM kernels, through the C ABI of libpfpp_hip.so (no Python: a fresh GPU box
// spends minutes importing torch).  For every shape: the default kernel (fp32 A, pre-split W) is the reference;
// every plane variant (PFPP_GEMM_PL=1..3, and the register-staged pre-split kernel = 0) is compared with it bit for
// bit and timed with HIP events.
//   usage: lab [iters] M,N,K[,act] ...      act: 0 none, 3 gelu, r = residual + bias
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "pfpp.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void split_kernel(const float* x, _Float16* hi, _Float16* lo, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  const _Float16 h = (_Float16)v;
  hi[i] = h;
  lo[i] = (_Float16)(v - (float)h);
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static float frand() {
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return (float)((rng_state >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f;
}

struct Shape { int64_t M, N, K; int act; bool res; };

static float time_gemm(const pfpp_gemm_args& a, int iters, hipStream_t st) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) { int rc = pfpp_gemm(&a, (pfpp_stream_t)st); if (rc) { printf("pfpp_gemm rc %d: %s\n", rc, pfpp_last_error()); return -1.f; } }
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) pfpp_gemm(&a, (pfpp_stream_t)st);
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / iters;
}

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 20;
  std::vector<Shape> shapes;
  for (int i = 2; i < argc; ++i) {
    Shape s{0, 0, 0, 0, false};
    char actc[8] = "0";
    int n = sscanf(argv[i], "%ld,%ld,%ld,%7s", &s.M, &s.N, &s.K, actc);
    if (n < 3) { printf("bad shape %s\n", argv[i]); return 1; }
    if (actc[0] == 'r') s.res = true; else s.act = atoi(actc);
    shapes.push_back(s);
  }
  hipStream_t st;
  CK(hipStreamCreate(&st));
  for (const Shape& s : shapes) {
    const int64_t M = s.M, N = s.N, K = s.K;
    std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N), hR((size_t)M * N);
    for (auto& v : hA) v = frand();
    const float wsc = 1.0f / sqrtf((float)K);
    for (auto& v : hW) v = frand() * wsc * 1.7f;
    for (auto& v : hb) v = frand();
    for (auto& v : hR) v = frand();
    float *dA, *dW, *db, *dR, *dC0, *dC1;
    _Float16 *ah, *al, *wh, *wl;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&db, N * 4));
    CK(hipMalloc(&dR, hR.size() * 4)); CK(hipMalloc(&dC0, (size_t)M * N * 4)); CK(hipMalloc(&dC1, (size_t)M * N * 4));
    CK(hipMalloc(&ah, hA.size() * 2)); CK(hipMalloc(&al, hA.size() * 2));
    CK(hipMalloc(&wh, hW.size() * 2)); CK(hipMalloc(&wl, hW.size() * 2));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dR, hR.data(), hR.size() * 4, hipMemcpyHostToDevice));
    split_kernel<<<(unsigned)((hA.size() + 255) / 256), 256, 0, st>>>(dA, ah, al, (int64_t)hA.size());
    split_kernel<<<(unsigned)((hW.size() + 255) / 256), 256, 0, st>>>(dW, wh, wl, (int64_t)hW.size());
    CK(hipStreamSynchronize(st));

    pfpp_gemm_args a;
    memset(&a, 0, sizeof(a));
    a.A = dA; a.C = dC0; a.w_hi = wh; a.w_lo = wl; a.bias = db;
    a.M = M; a.N = N; a.K = K; a.lda = K; a.ldw = K; a.ldc = N; a.ldr = N;
    a.act = s.act; a.batch = 1; a.zdiv = 1; a.precision = PFPP_GEMM_F16X3; a.alpha = 1.0f;
    if (s.res) a.residual = dR;
    const double gflop = 2.0 * M * N * K * 1e-9;
    setenv("PFPP_GEMM_PL", "0", 1);
    float t_ref = time_gemm(a, iters, st);
    printf("M%-7ld N%-5ld K%-5ld act%d%s | default(fp32 A) %8.1f us %7.1f TF/s\n", M, N, K, s.act, s.res ? "+res" : "", t_ref, gflop / t_ref * 1e3);
    std::vector<float> c0((size_t)M * N), c1((size_t)M * N);
    CK(hipMemcpy(c0.data(), dC0, c0.size() * 4, hipMemcpyDeviceToHost));
    a.A = nullptr; a.a_hi = ah; a.a_lo = al; a.C = dC1;
    const char* only = getenv("LAB_VARIANTS");      // e.g. "13": variants 1 and 3 only; letters a.. = 10..
    for (int v = 0; v <= 12; ++v) {
      if (only && !strchr(only, v < 10 ? '0' + v : 'a' + v - 10)) continue;
      char buf[8];
      snprintf(buf, sizeof buf, "%d", v);
      setenv("PFPP_GEMM_PL", buf, 1);
      CK(hipMemsetAsync(dC1, 0xFF, (size_t)M * N * 4, st));
      float t = time_gemm(a, iters, st);
      CK(hipMemcpy(c1.data(), dC1, c1.size() * 4, hipMemcpyDeviceToHost));
      size_t bad = 0; double maxd = 0.0; size_t first = (size_t)-1;
      for (size_t i = 0; i < c0.size(); ++i) {
        if (memcmp(&c0[i], &c1[i], 4) != 0) { if (!bad) first = i; ++bad; double d = fabs((double)c0[i] - (double)c1[i]); if (!(d <= maxd)) maxd = d; }
      }
      printf("    planes variant %d %8.1f us %7.1f TF/s   mismatches %zu (max |d| %.3g, first at row %zu col %zu)\n", v, t,
             gflop / t * 1e3, bad, maxd, bad ? first / N : 0, bad ? first % N : 0);
    }
    hipFree(dA); hipFree(dW); hipFree(db); hipFree(dR); hipFree(dC0); hipFree(dC1); hipFree(ah); hipFree(al); hipFree(wh); hipFree(wl);
    fflush(stdout);
  }
  return 0;
}
