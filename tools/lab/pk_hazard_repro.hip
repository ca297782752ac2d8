// Stand-alone reproducer attempt for DESIGN.md 6 (round 5): a dependent farthest-point chain whose distance updates compile to packed fp32
// instructions (v_pk_add_f32 / v_pk_mul_f32), run next to a matrix + LDS heavy kernel on a second stream; every launch's index chain is
// compared with the chain of an undisturbed launch.  Build both ways and compare the counts:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/lab/pk_hazard_repro.hip -o pk_on
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops tools/lab/pk_hazard_repro.hip -o pk_off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f2v __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int N = 1024, S = 256, W = 4, PPT = 4;      // points per fragment, selections, waves per fragment, points per thread

template <int CTRL>
__device__ __forceinline__ float dpp_max(float v) {
  const int iv = __float_as_int(v);
  return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(iv, iv, CTRL, 0xf, 0xf, false)));
}
__device__ __forceinline__ float wave_max(float v) {
  v = dpp_max<0xB1>(v); v = dpp_max<0x4E>(v); v = dpp_max<0x141>(v); v = dpp_max<0x140>(v);
  const int iv = __float_as_int(v);
  return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(iv, 0)), __int_as_float(__builtin_amdgcn_readlane(iv, 16))),
               fmaxf(__int_as_float(__builtin_amdgcn_readlane(iv, 32)), __int_as_float(__builtin_amdgcn_readlane(iv, 48))));
}
__global__ __launch_bounds__(256) void chain(const float* __restrict__ pts, int* __restrict__ out) {
  __shared__ float sx[N], sy[N], sz[N];
  __shared__ float wbest[2][W];
  __shared__ int widx[2][W];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* p = pts + (size_t)blockIdx.x * N * 3;
  for (int i = t; i < N; i += 256) { sx[i] = p[3 * i]; sy[i] = p[3 * i + 1]; sz[i] = p[3 * i + 2]; }
  __syncthreads();
  f2v x[2], y[2], z[2], d[2];
  for (int k = 0; k < PPT; ++k) { x[k >> 1][k & 1] = sx[t * PPT + k]; y[k >> 1][k & 1] = sy[t * PPT + k]; z[k >> 1][k & 1] = sz[t * PPT + k]; d[k >> 1][k & 1] = __builtin_huge_valf(); }
  int cur = 0;
  for (int s = 0; s < S; ++s) {
    if (t == 0) out[blockIdx.x * S + s] = cur;
    const f2v cx = {sx[cur], sx[cur]}, cy = {sy[cur], sy[cur]}, cz = {sz[cur], sz[cur]};
    float best = -1.0f;
    for (int q = 0; q < 2; ++q) {                 // two points at a time: the packed subtract / multiply / add of fps_kernel
      const f2v dx = x[q] - cx, dy = y[q] - cy, dz = z[q] - cz;
      const f2v dd = (dx * dx + dy * dy) + dz * dz;
      d[q][0] = fminf(d[q][0], dd[0]); d[q][1] = fminf(d[q][1], dd[1]);
      best = fmaxf(best, fmaxf(d[q][0], d[q][1]));
    }
    const float wm = wave_max(best);              // the product's reduction: DPP quad swaps / mirrors + v_readlane
    const int src = __builtin_ctzll(__ballot(best == wm));
    int kk = PPT - 1;
    for (int k = PPT - 2; k >= 0; --k) kk = d[k >> 1][k & 1] == wm ? k : kk;
    const int wi = __builtin_amdgcn_readlane(t * PPT + kk, src);
    if (lane == 0) { wbest[s & 1][wave] = wm; widx[s & 1][wave] = wi; }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    float bm = wbest[s & 1][0]; cur = widx[s & 1][0];
    for (int w = 1; w < W; ++w) if (wbest[s & 1][w] > bm) { bm = wbest[s & 1][w]; cur = widx[s & 1][w]; }
  }
}
__global__ __launch_bounds__(256) void load(const half8* in, float* sink, int iters) {      // the co-runner: matrix pipe + LDS traffic
  __shared__ half8 buf[1024];
  half8 a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
  f32x16 acc = {0};
  for (int i = 0; i < iters; ++i) {
    buf[(threadIdx.x + i) & 1023] = a;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    a = buf[(threadIdx.x * 7 + i) & 1023];
  }
  if (acc[0] == 123.0f) sink[0] = acc[1];
}
int main(int argc, char** argv) {
  const int F = 8, launches = argc > 1 ? atoi(argv[1]) : 20000;
  std::vector<float> h(F * N * 3);
  unsigned s = 3; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }
  float *pts, *sink; int *out, *ref; half8* in;
  hipMalloc(&pts, h.size() * 4); hipMemcpy(pts, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&out, F * S * 4); hipMalloc(&ref, F * S * 4); hipMalloc(&sink, 64); hipMalloc(&in, 4096); hipMemset(in, 0x3c, 4096);
  hipStream_t a, b; hipStreamCreate(&a); hipStreamCreate(&b);
  chain<<<F, 256, 0, a>>>(pts, ref); hipStreamSynchronize(a);
  std::vector<int> r(F * S), o(F * S); hipMemcpy(r.data(), ref, F * S * 4, hipMemcpyDeviceToHost);
  long bad = 0, bad_alone = 0;
  for (int pass = 0; pass < 2; ++pass)
    for (int i = 0; i < launches; ++i) {
      if (pass == 1 && i % 8 == 0) load<<<512, 256, 0, b>>>(in, sink, 4000);
      chain<<<F, 256, 0, a>>>(pts, out);
      hipMemcpyAsync(o.data(), out, F * S * 4, hipMemcpyDeviceToHost, a); hipStreamSynchronize(a);
      if (memcmp(o.data(), r.data(), F * S * 4)) (pass ? bad : bad_alone)++;
    }
  hipDeviceSynchronize();
  printf("chains that differ from the first launch: %ld of %d alone, %ld of %d next to the co-running matrix kernel\n", bad_alone, launches, bad, launches);
  return 0;
}
