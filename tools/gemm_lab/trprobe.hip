// probe: lane -> element mapping of ds_read_b64_tr_b16 on gfx950 (LDS holds u16 value = its own index)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short* out, int stride_bytes) {
  __shared__ unsigned short sm[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) sm[i] = (unsigned short)i;
  __syncthreads();
  const int lane = threadIdx.x;
  // lane l supplies the address of 4 contiguous halfs: row (l & 15 ... ) chosen by the host pattern: addr = lane * stride
  uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)sm + lane * stride_bytes;
  u16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  for (int stride : {8, 64}) {
    probe<<<1, 64>>>(d, stride);
    unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("stride %d bytes (lane l reads halfs [l*%d, l*%d+4)):\n", stride, stride / 2, stride / 2);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  return 0;
}
