// probe: how many bytes per second ONE CU can pull from L2 (or L1), by instruction form, waves per CU, loads in flight, working set per
// CU and sharing between CUs.  One workgroup per CU (256 workgroups); every wave streams 1 KB per instruction (64 lanes x 16 bytes,
// contiguous) over a region of `span` bytes, again and again.
//   mode 0: global_load_dwordx4 into registers     mode 1: global_load_lds_dwordx4 into an LDS ring
//   share 0: every workgroup has its own region (base + wg * span)    share 1: all workgroups read the SAME region from offset 0
//   share 2: the same region, every workgroup starting at its own rotated offset
//   hipcc --offload-arch=gfx950 -O3 tools/lab/l2_delivery_probe.hip -o tools/lab/_run/l2probe && tools/lab/_run/l2probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int U>
__global__ __launch_bounds__(1024) void probe(const char* __restrict__ base, size_t span, int share, int iters, float* sink) {
  extern __shared__ __align__(1024) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const char* reg = base + (share == 0 ? (size_t)blockIdx.x * span : 0);
  const uint32_t pmask = (uint32_t)(span >> 10) - 1;     // 1 KB pieces in the region (a power of two)
  uint32_t pc = wave + (share == 2 ? blockIdx.x * 37u : 0u);
  f32x4 acc = {0, 0, 0, 0};
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem + wave * U * 1024;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      f32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t q = (pc + (uint32_t)u * nw) & pmask;
        v[u] = *reinterpret_cast<const f32x4*>(reg + (size_t)q * 1024 + lane * 16);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u];
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t q = (pc + (uint32_t)u * nw) & pmask;
        __builtin_amdgcn_global_load_lds((gbl_void*)(reg + (size_t)q * 1024 + lane * 16), (lds_void*)(uintptr_t)(lds0 + u * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    pc += (uint32_t)U * nw;
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

template <int MODE, int U>
void run(const char* d, size_t span, int share, int waves, float* sink, const char* what) {
  const int iters = 400;
  auto k = probe<MODE, U>;
  const size_t smem = MODE ? (size_t)waves * U * 1024 : 0;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), smem, 0, d, span, share, iters, sink);
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), smem, 0, d, span, share, iters, sink);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = 5.0 * 256 * waves * (double)U * 1024 * iters;
  printf("%-14s mode %d U %2d waves %d span %7zu KB share %d: %7.1f GB/s per CU, %6.2f TB/s chip\n", what, MODE, U, waves, span >> 10, share,
         bytes / (ms * 1e-3) / 256 * 1e-9, bytes / (ms * 1e-3) * 1e-12);
}

int main() {
  char* d; float* sink;
  const size_t total = (size_t)256 << 20;
  CK(hipMalloc(&d, total)); CK(hipMemset(d, 1, total)); CK(hipMalloc(&sink, 64));
  for (int waves : {4, 8, 16}) {
    run<0, 8>(d, 16 << 10, 0, waves, sink, "L1 private");
    run<0, 8>(d, 64 << 10, 0, waves, sink, "L2 private");
    run<0, 16>(d, 64 << 10, 0, waves, sink, "L2 private");
    run<1, 8>(d, 64 << 10, 0, waves, sink, "L2 private");
    run<1, 16>(d, 64 << 10, 0, waves, sink, "L2 private");
    run<0, 8>(d, 512 << 10, 0, waves, sink, "MALL private");
    run<1, 12>(d, 512 << 10, 0, waves, sink, "MALL private");
    run<0, 8>(d, 2 << 20, 1, waves, sink, "L2 shared");
    run<1, 12>(d, 2 << 20, 1, waves, sink, "L2 shared");
    run<0, 8>(d, 2 << 20, 2, waves, sink, "L2 shared rot");
    run<1, 12>(d, 2 << 20, 2, waves, sink, "L2 shared rot");
    run<1, 12>(d, 512 << 10, 2, waves, sink, "L2 shared rot");
  }
  return 0;
}
