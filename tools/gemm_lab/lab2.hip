// Developer bench for pfpp_gemm_planes (nt / nn / tn operand layouts, split-K accumulate): checks a sample of the
// output against a float64 evaluation of the same three-term product on the host and times the launch.
//   usage: lab2 [iters] mode,M,N,K[,variant[,splits]] ...     mode: nt | nn | tn ; accumulate is implied by splits != 1 or tn
#include <hip/hip_runtime.h>
#include <math.h>
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
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static float frand() { return (float)((rnd() >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f; }

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 20;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  for (int ai = 2; ai < argc; ++ai) {
    char mode[8];
    long M, N, K; int variant = 0, splits = -1;
    int n = sscanf(argv[ai], "%2s,%ld,%ld,%ld,%d,%d", mode, &M, &N, &K, &variant, &splits);
    if (n < 4) { printf("bad spec %s\n", argv[ai]); return 1; }
    const bool ak = !strcmp(mode, "tn"), wk = ak || !strcmp(mode, "nn");
    const bool accum = ak || (splits >= 0 && splits != 1);
    if (splits < 0) splits = 0;
    const int64_t lda = ak ? M : K, ldw = wk ? N : K;
    const size_t na = (size_t)(ak ? K : M) * lda, nw = (size_t)(wk ? K : N) * ldw;
    std::vector<float> hA(na), hW(nw), hb(N);
    for (auto& v : hA) v = frand();
    const float wsc = 1.7f / sqrtf((float)K);
    for (auto& v : hW) v = frand() * wsc;
    for (auto& v : hb) v = frand();
    float *dA, *dW, *db, *dC;
    _Float16 *ah, *al, *wh, *wl;
    CK(hipMalloc(&dA, na * 4)); CK(hipMalloc(&dW, nw * 4)); CK(hipMalloc(&db, N * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMalloc(&ah, na * 2)); CK(hipMalloc(&al, na * 2)); CK(hipMalloc(&wh, nw * 2)); CK(hipMalloc(&wl, nw * 2));
    CK(hipMemcpy(dA, hA.data(), na * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), nw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
    split_kernel<<<(unsigned)((na + 255) / 256), 256, 0, st>>>(dA, ah, al, (int64_t)na);
    split_kernel<<<(unsigned)((nw + 255) / 256), 256, 0, st>>>(dW, wh, wl, (int64_t)nw);
    std::vector<_Float16> hah(na), hal(na), hwh(nw), hwl(nw);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(hah.data(), ah, na * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hal.data(), al, na * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hwh.data(), wh, nw * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hwl.data(), wl, nw * 2, hipMemcpyDeviceToHost));

    pfpp_gemm_planes_args a;
    memset(&a, 0, sizeof(a));
    a.a_hi = ah; a.a_lo = al; a.w_hi = wh; a.w_lo = wl; a.C = dC; a.bias = db;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldc = N;
    a.a_kmajor = ak; a.w_kmajor = wk; a.accumulate = accum; a.splits = splits; a.variant = variant; a.alpha = 0.25f;
    float* dws = nullptr;
    if (getenv("LAB_WS")) { CK(hipMalloc(&dws, (size_t)64 << 20)); a.ws = dws; a.ws_bytes = (int64_t)64 << 20; }
    CK(hipMemsetAsync(dC, 0, (size_t)M * N * 4, st));
    int rc = pfpp_gemm_planes(&a, (pfpp_stream_t)st);
    if (rc) { printf("%s: rc %d: %s\n", argv[ai], rc, pfpp_last_error()); continue; }
    std::vector<float> c((size_t)M * N);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0.0, maxref = 0.0; long worst_r = 0, worst_c = 0;
    const int samples = 4000;
    for (int sidx = 0; sidx < samples; ++sidx) {
      long r = (long)(rnd() % (uint64_t)M), col = (long)(rnd() % (uint64_t)N);
      if (sidx < 64) { r = sidx < 32 ? (sidx % M) : (M - 1 - (sidx % 32) % M); col = sidx % 2 ? N - 1 - (sidx % N) % N : sidx % N; }
      double acc = 0.0;
      for (long k = 0; k < K; ++k) {
        const size_t ia = ak ? (size_t)k * lda + r : (size_t)r * lda + k;
        const size_t iw = wk ? (size_t)k * ldw + col : (size_t)col * ldw + k;
        const double xh = (double)(float)hah[ia], xl = (double)(float)hal[ia], yh = (double)(float)hwh[iw], yl = (double)(float)hwl[iw];
        acc += xl * yh + xh * yl + xh * yh;
      }
      const double ref = acc * 0.25 + hb[col];
      const double err = fabs(ref - (double)c[(size_t)r * N + col]);
      if (err > maxerr) { maxerr = err; worst_r = r; worst_c = col; }
      if (fabs(ref) > maxref) maxref = fabs(ref);
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) pfpp_gemm_planes(&a, (pfpp_stream_t)st);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) pfpp_gemm_planes(&a, (pfpp_stream_t)st);
    CK(hipEventRecord(e1, st));
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters;
    printf("%s M%-6ld N%-6ld K%-6ld variant %d splits %d%s: %8.1f us %7.1f TF/s   max |err| %.3g (|ref| <= %.3g) at (%ld, %ld)  %s\n", mode, M, N, K,
           variant, splits, accum ? " accum" : "", us, 2.0 * M * N * K / us * 1e-6, maxerr, maxref, worst_r, worst_c,
           maxerr <= 2e-5 * (maxref > 1 ? maxref : 1) ? "ok" : "MISMATCH");
    fflush(stdout);
    if (dws) hipFree(dws);
    hipFree(dA); hipFree(dW); hipFree(db); hipFree(dC); hipFree(ah); hipFree(al); hipFree(wh); hipFree(wl);
  }
  return 0;
}
