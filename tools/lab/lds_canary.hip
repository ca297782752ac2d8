// lab: does anything else on the chip write into a workgroup's LDS?  Every workgroup fills its dynamic LDS with a pattern of its own and
// re-reads it `iters` times; words that changed are counted.  Run next to the kernel under suspicion on another stream
// (tools/diag/lds_canary.py).    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/lab/lds_canary.hip -o tools/lab/_bin/liblds_canary.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void canary_kernel(unsigned long long* bad, int iters, int words) {
  extern __shared__ unsigned cs[];
  const unsigned tag = 0x9e3779b9u * (blockIdx.x + 1);
  for (int i = threadIdx.x; i < words; i += 256) cs[i] = tag ^ (unsigned)i;
  __syncthreads();
  unsigned long long n = 0;
  for (int it = 0; it < iters; ++it) {
    for (int i = threadIdx.x; i < words; i += 256) {
      const unsigned v = *(volatile unsigned*)&cs[i];
      if (v != (tag ^ (unsigned)i)) { ++n; cs[i] = tag ^ (unsigned)i; }
    }
    __builtin_amdgcn_s_sleep(8);
  }
  if (n) atomicAdd(bad, n);
  if (threadIdx.x == 0) atomicAdd(bad + 1, 1ull);
}

extern "C" int canary_launch(void* stream, int wgs, int lds_bytes, int iters, unsigned long long* d_bad) {
  hipLaunchKernelGGL(canary_kernel, dim3(wgs), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, d_bad, iters, lds_bytes / 4);
  return (int)hipGetLastError();
}
