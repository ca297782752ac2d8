// probe: where the dk/dv pass of the dense attention backward spends its tile step.  Compiles the product source with AB_PROBE
// (shader-clock stamps between the phases of workgroup (0,0,0) wave 0) and runs it on uniform sequences.
//   hipcc --offload-arch=gfx950 -O3 -I include -I puzzlefusion-plusplus_amd/csrc tools/lab/attn_bwd_probe.hip -o gpurun_out/attn_bwd_probe
#define AB_PROBE 1
#include "attention_bwd.hip"
#include <cstdio>
#include <vector>
#include <cmath>

// pfpp_common.h declares these; the probe is a stand-alone program, so it brings its own
namespace pfpp {
void set_error(const char*, ...) {}
int check_launch(const char*) { return hipGetLastError() == hipSuccess ? 0 : 1; }
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32, T = argc > 2 ? atoi(argv[2]) : 128, H = 8, DH = 64;
  const int64_t rows = (int64_t)B * T;
  std::vector<float> h_qkv(rows * 3 * H * DH), h_do(rows * H * DH), h_lse(rows * H, 3.0f), h_d(rows * H, 0.0f);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& x : h_qkv) x = rnd();
  for (auto& x : h_do) x = rnd() * 1e-3f;
  std::vector<int32_t> h_off(B), h_len(B, T);
  for (int b = 0; b < B; ++b) h_off[b] = b * T;
  float *qkv, *dout, *lse, *dvec, *dqkv;
  int32_t *off, *len;
  hipMalloc(&qkv, h_qkv.size() * 4); hipMalloc(&dout, h_do.size() * 4); hipMalloc(&lse, h_lse.size() * 4);
  hipMalloc(&dvec, h_d.size() * 4); hipMalloc(&dqkv, h_qkv.size() * 4); hipMalloc(&off, B * 4); hipMalloc(&len, B * 4);
  hipMemcpy(qkv, h_qkv.data(), h_qkv.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dout, h_do.data(), h_do.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(lse, h_lse.data(), h_lse.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dvec, h_d.data(), h_d.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(off, h_off.data(), B * 4, hipMemcpyHostToDevice);
  hipMemcpy(len, h_len.data(), B * 4, hipMemcpyHostToDevice);
  const dim3 grid((T + 127) / 128, H, B);
  const pfpp_planes_out none{nullptr, nullptr, 1.0f};
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i)
    hipLaunchKernelGGL(attn_dense_bwd_dkv_f16_kernel, grid, dim3(256), 0, 0, qkv, dout, lse, dvec, dqkv, off, len, H, 0.125f, none);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i)
    hipLaunchKernelGGL(attn_dense_bwd_dkv_f16_kernel, grid, dim3(256), 0, 0, qkv, dout, lse, dvec, dqkv, off, len, H, 0.125f, none);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  long long t[16];
  hipMemcpyFromSymbol(t, HIP_SYMBOL(ab_probe_out), sizeof(t));
  const char* names[9] = {"prologue (own k/v fragments, tile 0 load+store, barrier)", "issue next tile's global loads", "stage 1: S, dP (24 MFMA, LDS fragments)",
                          "exp / dS", "accumulators -> fragments (x2)", "stage 2: dV, dK (24 MFMA)", "split + store next tile (waits for the loads)",
                          "barrier", "epilogue"};
  printf("B %d T %d: dkv %.1f us per launch (with stamps); workgroup 0 wave 0: %lld clocks over %lld tiles\n", B, T, ms / 20 * 1e3, t[9], t[10]);
  for (int i = 0; i < 9; ++i) printf("  %8lld clocks (%5.1f %%)  %s\n", t[i], 100.0 * t[i] / t[9], names[i]);
  return 0;
}
