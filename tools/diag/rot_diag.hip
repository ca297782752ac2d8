// stage-by-stage GPU-vs-host comparison of the quaternion rotate arithmetic (developer diagnostic)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
struct Out { float n2, n, qw, qx, t_w, t_x, t_y, t_z, rx, ry, rz; };
__host__ __device__ inline void stages(const float* q4, const float* p3, Out* o) {
  const float w = q4[0], x = q4[1], y = q4[2], z = q4[3];
  const float n2 = ((w * w + x * x) + y * y) + z * z;
  const float n = sqrtf(n2);
  const float a0 = w / n, a1 = x / n, a2 = y / n, a3 = z / n;
  const float b0 = 0.0f, b1 = p3[0], b2 = p3[1], b3 = p3[2];
  const float tw = ((a0 * b0 - a1 * b1) - a2 * b2) - a3 * b3;
  const float tx = ((a0 * b1 + a1 * b0) + a2 * b3) - a3 * b2;
  const float ty = ((a0 * b2 - a1 * b3) + a2 * b0) + a3 * b1;
  const float tz = ((a0 * b3 + a1 * b2) - a2 * b1) + a3 * b0;
  const float c0 = a0, c1 = -a1, c2 = -a2, c3 = -a3;
  o->n2 = n2; o->n = n; o->qw = a0; o->qx = a1; o->t_w = tw; o->t_x = tx; o->t_y = ty; o->t_z = tz;
  o->rx = ((tw * c1 + tx * c0) + ty * c3) - tz * c2;
  o->ry = ((tw * c2 - tx * c3) + ty * c0) + tz * c1;
  o->rz = ((tw * c3 + tx * c2) - ty * c1) + tz * c0;
}
__global__ void k(const float* q, const float* p, Out* o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) stages(q + 4 * i, p + 3 * i, o + i);
}
int main() {
  const int n = 1 << 18;
  float* q = (float*)malloc(n * 16); float* p = (float*)malloc(n * 12);
  srand(1);
  for (int i = 0; i < 4 * n; ++i) q[i] = (float)rand() / RAND_MAX * 4.f - 2.f;
  for (int i = 0; i < 3 * n; ++i) p[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
  float *dq, *dp; Out* dout;
  hipMalloc(&dq, n * 16); hipMalloc(&dp, n * 12); hipMalloc(&dout, n * sizeof(Out));
  hipMemcpy(dq, q, n * 16, hipMemcpyHostToDevice); hipMemcpy(dp, p, n * 12, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dq, dp, dout, n);
  Out* g = (Out*)malloc(n * sizeof(Out));
  hipMemcpy(g, dout, n * sizeof(Out), hipMemcpyDeviceToHost);
  const char* names[11] = {"n2", "n", "qw", "qx", "t_w", "t_x", "t_y", "t_z", "rx", "ry", "rz"};
  long bad[11] = {0};
  for (int i = 0; i < n; ++i) {
    Out h; stages(q + 4 * i, p + 3 * i, &h);
    const float* a = (const float*)&h; const float* b = (const float*)&g[i];
    for (int s = 0; s < 11; ++s) if (memcmp(&a[s], &b[s], 4) != 0) bad[s]++;
  }
  for (int s = 0; s < 11; ++s) printf("%-4s mismatches %ld / %d\n", names[s], bad[s], n);
  return 0;
}
