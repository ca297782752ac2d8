#include <hip/hip_runtime.h>
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
  const int iv = __float_as_int(v);
  const int o = __builtin_amdgcn_update_dpp(iv, iv, CTRL, 0xf, 0xf, false);
  return v + __int_as_float(o);
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  {
    const unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  {
    const unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  v = dpp_add<0x128>(v);  // row_ror:8
  v = dpp_add<0x124>(v);  // row_ror:4
  v = dpp_add<0x122>(v);  // row_ror:2
  v = dpp_add<0x121>(v);  // row_ror:1
  return v;
}
__global__ void k(float* x, float* y) {
  float v = x[threadIdx.x];
  float a = wave_sum_dpp(v);
  float b = v;
  for (int off = 32; off >= 1; off >>= 1) b += __shfl_xor(b, off);
  y[threadIdx.x] = a; y[64 + threadIdx.x] = b;
}
int main() {
  float *x, *y; hipMalloc(&x, 256); hipMalloc(&y, 512);
  float h[64], o[128]; unsigned s = 12345;
  for (int i = 0; i < 64; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)(s >> 8) / 16777216.0f * 3.7f - 1.1f; }
  hipMemcpy(x, h, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(x, y);
  hipMemcpy(o, y, 512, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 64; ++i) bad += (o[i] != o[64 + i]);
  printf("mismatches %d  (%.9g %.9g)\n", bad, o[0], o[64]);
  return bad;
}
