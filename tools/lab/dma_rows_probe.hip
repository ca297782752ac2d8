// probe: L2 -> LDS delivery rate of global_load_lds_dwordx4 per CU against the length of the contiguous run a lane group fetches per
// row: the plane GEMM's K-tile of 32 halfs makes every DMA instruction touch 16 rows x 64 bytes (half a 128-byte line each);
// a K-tile of 64 halfs would touch 8 rows x 128 bytes.  One workgroup per CU streams a [ROWS x K] fp16 panel (L2-resident after the
// first pass) into a 3-stage LDS ring like the GEMM does; nothing consumes it.
//   hipcc --offload-arch=gfx950 -O3 dma_rows_probe.hip -o dma_rows_probe && ./dma_rows_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// RUN = bytes per row per tile (64 or 128 or 256); a tile = ROWS rows x RUN bytes; each wave issues (ROWS * RUN / 1024 / NW) DMA
// instructions per tile (64 lanes x 16 bytes = 1 KB each)
template <int ROWS, int RUN, int NW, int NS>
__global__ __launch_bounds__(NW * 64) void probe(const char* __restrict__ src, int64_t ld_bytes, int ntiles, int reps, int* sink, int npan) {
  extern __shared__ __align__(1024) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int TILE = ROWS * RUN;
  constexpr int PIECES = TILE / 1024;             // 1 KB DMA instructions per tile
  constexpr int PPW = PIECES / NW;
  constexpr int LPR = RUN / 16;                   // lanes per row
  const char* base = src + (size_t)(blockIdx.x % npan) * ROWS * ld_bytes;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem;
  for (int r = 0; r < reps; ++r) {
    for (int t = 0; t < ntiles; ++t) {
      const uint32_t stage = lds0 + (t % NS) * TILE;
#pragma unroll
      for (int j = 0; j < PPW; ++j) {
        const int piece = wave + NW * j;
        const int ci = piece * 64 + lane;          // 16-byte chunk index within the tile
        const int row = ci / LPR, chunk = ci % LPR;
        __builtin_amdgcn_global_load_lds((gbl_void*)(base + (size_t)row * ld_bytes + (size_t)t * RUN + chunk * 16),
                                         (lds_void*)(uintptr_t)(stage + piece * 1024), 16, 0, 0);
      }
      // keep NS - 1 tiles in flight
      if (t >= NS - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * PPW) : "memory");
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (tid == 0 && smem[0] == 77 && smem[5] == 33) *sink = 1;
}

template <int ROWS, int RUN, int NW, int NS>
void run(const char* name, int K_halfs, int nwg, int npan = 0) {
  if (npan == 0) npan = nwg;
  const int64_t ld = (int64_t)K_halfs * 2;
  char* src; int* sink;
  hipMalloc(&src, (size_t)nwg * ROWS * ld); hipMalloc(&sink, 4);
  hipMemset(src, 1, (size_t)nwg * ROWS * ld);
  const int ntiles = (int)(ld / RUN), reps = 200;
  const size_t smem = (size_t)NS * ROWS * RUN;
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<ROWS, RUN, NW, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL((probe<ROWS, RUN, NW, NS>), dim3(nwg), dim3(NW * 64), smem, 0, src, ld, ntiles, 3, sink, npan);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<ROWS, RUN, NW, NS>), dim3(nwg), dim3(NW * 64), smem, 0, src, ld, ntiles, reps, sink, npan);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)nwg * ROWS * ld * reps;
  printf("%-44s panels %3d rows %3d run %3d B waves %d stages %d K %4d wgs %3d: %7.1f GB/s per workgroup, %6.2f TB/s total, %.2f us per tile\n", name, npan, ROWS, RUN, NW,
         NS, K_halfs, nwg, bytes / (ms * 1e-3) / nwg / 1e9, bytes / (ms * 1e-3) / 1e12, ms * 1e3 / reps / ntiles);
  hipFree(src); hipFree(sink);
}

int main() {
  // HBM / MALL streaming (every workgroup its own panel)
  run<384, 64, 4, 3>("stream: BK 32, 3 stages", 512, 256);
  run<384, 128, 4, 3>("stream: BK 64, 3 stages", 512, 256);
  // L2-resident panels shared by workgroups of the same XCD (32 panels of 393 KB: 4 per XCD)
  run<384, 64, 4, 3>("L2: BK 32, 3 stages", 512, 256, 32);
  run<384, 128, 4, 3>("L2: BK 64, 3 stages", 512, 256, 32);
  run<384, 64, 4, 6>("L2: BK 32, 6 stages", 512, 256, 32);
  run<192, 128, 4, 3>("L2: BK 64, 192 rows (half tile), 3 stages", 512, 256, 32);
  run<384, 64, 4, 3>("L2: BK 32, 8 panels (one per XCD)", 512, 256, 8);
  run<384, 128, 4, 3>("L2: BK 64, 8 panels (one per XCD)", 512, 256, 8);
  run<384, 64, 4, 3>("L2: BK 32, K 2048, 8 panels", 2048, 256, 8);
  run<384, 128, 4, 3>("L2: BK 64, K 2048, 8 panels", 2048, 256, 8);
  return 0;
}
