// probe: latency and correctness of a software grid barrier across the 8 XCDs of gfx950 (one persistent workgroup per CU).
// Every round each workgroup writes a value, all synchronise, each reads its right neighbour's (an XCD away) value and checks it.
//   hipcc --offload-arch=gfx950 -O3 gridsync_probe.hip -o gridsync_probe && ./gridsync_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#ifndef VARIANT
#define VARIANT 0
#endif
// bar layout (unsigned words, every counter on its own 256-byte line): [0] root counter, [64] release flag (generation),
// [128 + 64 g] group counter g
__device__ __forceinline__ void grid_sync(unsigned* bar, unsigned nwg, unsigned& gen) {
  __syncthreads();
  if (threadIdx.x == 0) {
#if VARIANT >= 7
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
#elif VARIANT != 4
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
#if VARIANT == 0
    const unsigned target = (gen + 1) * nwg;
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
#elif VARIANT == 1      // flat counter, separate release flag written by the last arriver
    const unsigned t = __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == (gen + 1) * nwg - 1) __hip_atomic_store(bar + 64, gen + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else while (__hip_atomic_load(bar + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen + 1) __builtin_amdgcn_s_sleep(2);
#else                   // two levels: groups of GSZ workgroups (VARIANT 2: consecutive ids, 3: same XCD = id % 8), then the group leaders
    constexpr unsigned NG = (VARIANT == 3) ? 8 : 16;
    const unsigned g = (VARIANT == 3) ? (blockIdx.x % NG) : (blockIdx.x / (nwg / NG));
    const unsigned gsz = nwg / NG;
    const unsigned t = __hip_atomic_fetch_add(bar + 128 + 64 * g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool last = false;
    if (t == (gen + 1) * gsz - 1) {
      const unsigned r = __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = r == (gen + 1) * NG - 1;
    }
#if VARIANT == 5 || VARIANT == 6     // per-group release flags: 16 pollers per flag instead of 256
    unsigned* flag = bar + 2048 + 64 * g;
    if (last) { for (unsigned k = 0; k < NG; ++k) __hip_atomic_store(bar + 2048 + 64 * k, gen + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    else while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen + 1) __builtin_amdgcn_s_sleep(VARIANT == 6 ? 8 : 1);
#else
    if (last) __hip_atomic_store(bar + 64, gen + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else while (__hip_atomic_load(bar + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen + 1) __builtin_amdgcn_s_sleep(2);
#endif
#endif
#if VARIANT >= 7
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#elif VARIANT != 4
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
  }
  ++gen;
  __syncthreads();
}

#if VARIANT >= 7
// agent-coherent data accesses (sc1): visible across XCDs without L2 write-back / invalidate fences
__device__ __forceinline__ void st_coh(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_coh(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#else
__device__ __forceinline__ void st_coh(float* p, float v) { *p = v; }
__device__ __forceinline__ float ld_coh(const float* p) { return *p; }
#endif
template <int PAYLOAD>   // floats written per workgroup per round (x 256 threads)
__global__ __launch_bounds__(256) void probe(unsigned* bar, float* buf, int rounds, int* errors) {
  unsigned gen = 0;
  const unsigned nwg = gridDim.x;
  const int me = blockIdx.x, nb = (blockIdx.x + 1) % gridDim.x;
  int bad = 0;
  for (int r = 0; r < rounds; ++r) {
    for (int i = 0; i < PAYLOAD; ++i) st_coh(&buf[((size_t)me * PAYLOAD + i) * 256 + threadIdx.x], (float)(r * 7 + me + i));
    grid_sync(bar, nwg, gen);
    for (int i = 0; i < PAYLOAD; ++i) {
      const float v = ld_coh(&buf[((size_t)nb * PAYLOAD + i) * 256 + threadIdx.x]);
      if (v != (float)(r * 7 + nb + i)) ++bad;
    }
    grid_sync(bar, nwg, gen);
  }
  if (bad) atomicAdd(errors, bad);
}

template <int PAYLOAD>
void run(int nwg, int rounds) {
  unsigned* bar; float* buf; int* err;
  hipMalloc(&bar, 16384); hipMalloc(&buf, (size_t)nwg * PAYLOAD * 256 * 4 + 1024); hipMalloc(&err, 4);
  hipMemset(bar, 0, 16384); hipMemset(err, 0, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<PAYLOAD>, dim3(nwg), dim3(256), 0, 0, bar, buf, 10, err);
  hipDeviceSynchronize(); hipMemset(bar, 0, 16384);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<PAYLOAD>, dim3(nwg), dim3(256), 0, 0, bar, buf, rounds, err);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  int h; hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost);
  printf("nwg %3d payload %5d B/wg: %.2f us per barrier (2 per round, incl. the payload write + read), errors %d\n", nwg, PAYLOAD * 1024,
         ms * 1e3 / (2.0 * rounds), h);
  hipFree(bar); hipFree(buf); hipFree(err);
}

int main() {
  printf("variant %d\n", VARIANT);
  run<1>(256, 2000); run<1>(128, 2000);
  run<16>(256, 1000);
  return 0;
}
