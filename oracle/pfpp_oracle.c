/*
 * TEST INFRASTRUCTURE — not part of the product.
 *
 * Plain-C CPU restatement of the integer / argmax-chain pieces of the
 * PuzzleFusion++ fragment encoder, with every rounding step spelled out so the
 * result does not depend on a BLAS or on compiler contraction
 * (build: gcc -O2 -ffp-contract=off -fno-fast-math, see oracle/build.py).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  Paths cited are relative to the reference checkout.
 *
 * Parity status: torch_cluster.fps and pytorch3d are NOT in the reference tree
 * and not installed here -> oracle_fps / oracle_quat_apply follow
 * their documented semantics ("parity unpinned" for those two); ball query and
 * VQ are checked against the reference's own Python (tools/make_goldens.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- a1: Denoiser._apply_rots, puzzlefusion_plusplus/denoiser/model/denoiser.py:55-63
 * pytorch3d.transforms.quaternion_apply after q / q.norm(); each elementwise torch
 * op rounds once, sums are evaluated left to right.                              */
static void raw_mul(const float a[4], const float b[4], float o[4]) {
  o[0] = ((a[0] * b[0] - a[1] * b[1]) - a[2] * b[2]) - a[3] * b[3];
  o[1] = ((a[0] * b[1] + a[1] * b[0]) + a[2] * b[3]) - a[3] * b[2];
  o[2] = ((a[0] * b[2] - a[1] * b[3]) + a[2] * b[0]) + a[3] * b[1];
  o[3] = ((a[0] * b[3] + a[1] * b[2]) - a[2] * b[1]) + a[3] * b[0];
}

/* pts [n,3], q[4] (w,x,y,z) -> out [n,3]; normalise: q <- q / sqrt(((w*w+x*x)+y*y)+z*z) */
void oracle_quat_apply(const float* pts, const float* quat, float* out, int64_t n, int normalise) {
  float q[4] = {quat[0], quat[1], quat[2], quat[3]};
  if (normalise) {
    const float s = ((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3];
    const float nrm = sqrtf(s);
    for (int k = 0; k < 4; ++k) q[k] = q[k] / nrm;
  }
  const float qi[4] = {q[0], -q[1], -q[2], -q[3]};
  for (int64_t i = 0; i < n; ++i) {
    const float p[4] = {0.0f, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    float t[4], r[4];
    raw_mul(q, p, t);
    raw_mul(t, qi, r);
    out[3 * i] = r[1]; out[3 * i + 1] = r[2]; out[3 * i + 2] = r[3];
  }
}

/* ---- a2: torch_cluster.fps(random_start=False) as called at utils/pn2_utils.py:131-137
 * (SURVEY.md A1): start 0, dist = running min of (dx*dx+dy*dy)+dz*dz, next = first argmax. */
void oracle_fps(const float* xyz, int64_t F, int64_t N, int64_t S, int32_t* idx) {
#pragma omp parallel for schedule(dynamic)
  for (int64_t f = 0; f < F; ++f) {
    const float* p = xyz + f * N * 3;
    float* dist = (float*)malloc(sizeof(float) * (size_t)N);
    for (int64_t i = 0; i < N; ++i) dist[i] = INFINITY;
    int64_t cur = 0;
    for (int64_t s = 0; s < S; ++s) {
      idx[f * S + s] = (int32_t)cur;
      if (s + 1 == S) break;
      const float cx = p[3 * cur], cy = p[3 * cur + 1], cz = p[3 * cur + 2];
      float best = -1.0f;
      int64_t bi = 0;
      for (int64_t i = 0; i < N; ++i) {
        const float dx = p[3 * i] - cx, dy = p[3 * i + 1] - cy, dz = p[3 * i + 2] - cz;
        const float d = (dx * dx + dy * dy) + dz * dz;   /* -ffp-contract=off: no fma */
        const float nd = d < dist[i] ? d : dist[i];
        dist[i] = nd;
        if (nd > best) { best = nd; bi = i; }
      }
      cur = bi;
    }
    free(dist);
  }
}

/* ---- a3: square_distance + query_ball_point, utils/pn2_utils.py:21-42, 92-112
 * d = ((-2*dot) + |c|^2) + |p|^2, dot = fma(c2,p2,fma(c1,p1,c0*p0)) — the K=3 sgemm of
 * the CPU BLAS (verified against torch.matmul in tools/make_goldens.py);
 * keep unless d > r2; first nsample kept indices ascending, padded with the first. */
void oracle_ball_query(const float* xyz, const float* new_xyz, int64_t F, int64_t N, int64_t S,
                       int64_t ns, float r2, int32_t* idx) {
#pragma omp parallel for schedule(dynamic)
  for (int64_t f = 0; f < F; ++f) {
    const float* p = xyz + f * N * 3;
    float* pp = (float*)malloc(sizeof(float) * (size_t)N);
    for (int64_t i = 0; i < N; ++i)
      pp[i] = (p[3 * i] * p[3 * i] + p[3 * i + 1] * p[3 * i + 1]) + p[3 * i + 2] * p[3 * i + 2];
    for (int64_t s = 0; s < S; ++s) {
      const float* c = new_xyz + (f * S + s) * 3;
      const float nn = (c[0] * c[0] + c[1] * c[1]) + c[2] * c[2];
      int32_t* o = idx + (f * S + s) * ns;
      int64_t cnt = 0;
      for (int64_t i = 0; i < N && cnt < ns; ++i) {
        const float dot = fmaf(c[2], p[3 * i + 2], fmaf(c[1], p[3 * i + 1], c[0] * p[3 * i]));
        const float d = (-2.0f * dot + nn) + pp[i];
        if (!(d > r2)) o[cnt++] = (int32_t)i;
      }
      const int32_t first = cnt > 0 ? o[0] : (int32_t)N;
      for (int64_t j = cnt; j < ns; ++j) o[j] = first;
    }
    free(pp);
  }
}

/* ---- a7: VectorQuantizer.forward argmin, vqvae/model/modules/quantizer.py:45-50
 * d_j = (|z|^2 + |e_j|^2) - 2*(z.e_j); squared norms summed in column order, the dot
 * as a k-ordered fma chain; first minimum.  z [n,dim], cb [K,dim] -> code [n]      */
void oracle_vq_argmin(const float* z, const float* cb, int64_t n, int64_t K, int64_t dim,
                      int32_t* code) {
  float* ee = (float*)malloc(sizeof(float) * (size_t)K);
  for (int64_t j = 0; j < K; ++j) {
    float s = 0.0f;
    for (int64_t d = 0; d < dim; ++d) s = s + cb[j * dim + d] * cb[j * dim + d];
    ee[j] = s;
  }
#pragma omp parallel for
  for (int64_t i = 0; i < n; ++i) {
    const float* zi = z + i * dim;
    float zz = 0.0f;
    for (int64_t d = 0; d < dim; ++d) zz = zz + zi[d] * zi[d];
    float best = INFINITY;
    int32_t bj = 0;
    for (int64_t j = 0; j < K; ++j) {
      float dot = 0.0f;
      for (int64_t d = 0; d < dim; ++d) dot = fmaf(zi[d], cb[j * dim + d], dot);
      const float dj = (zz + ee[j]) - 2.0f * dot;
      if (dj < best) { best = dj; bj = (int32_t)j; }
    }
    code[i] = bj;
  }
  free(ee);
}
