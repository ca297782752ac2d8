// Point-cloud kernels of the fragment encoder and the SE(3) glue (gfx950).
//
// Everything here is index / byte work or an argmax chain whose results must be
// BIT-IDENTICAL to the CPU path, so this file is
// compiled with -ffp-contract=off and every place where the CPU evaluates a
// fused multiply-add says so with an explicit __fmaf_rn.
#include "pfpp_common.h"

namespace {

// ---------------------------------------------------------------------------
// quaternion helpers — pytorch3d.transforms operation order
// ---------------------------------------------------------------------------
struct Quat { float w, x, y, z; };

// quaternion_raw_multiply(a, b): left-to-right evaluation of each 4-term sum,
// every product and every sum rounded to fp32 on its own (torch elementwise ops).
__device__ __forceinline__ Quat quat_raw_mul(const Quat a, const Quat b) {
  Quat o;  // -ffp-contract=off: every product and sum below rounds on its own
  o.w = ((a.w * b.w - a.x * b.x) - a.y * b.y) - a.z * b.z;
  o.x = ((a.w * b.x + a.x * b.w) + a.y * b.z) - a.z * b.y;
  o.y = ((a.w * b.y - a.x * b.z) + a.y * b.w) + a.z * b.x;
  o.z = ((a.w * b.z + a.x * b.y) - a.y * b.x) + a.z * b.w;
  return o;
}

// q / ||q||, ||q|| = sqrt(((w*w + x*x) + y*y) + z*z)  (torch.norm over 4 values)
__device__ __forceinline__ Quat quat_normalise(const Quat q) {
  const float n2 = ((q.w * q.w + q.x * q.x) + q.y * q.y) + q.z * q.z;
  const float n = sqrtf(n2);   // correctly rounded (no -ffast-math, default hipcc sqrt/div)
  Quat o;
  o.w = q.w / n;
  o.x = q.x / n;
  o.y = q.y / n;
  o.z = q.z / n;
  return o;
}

// quaternion_apply(q, p) = (q * (0,p) * q^-1)[1:], q^-1 = q * (1,-1,-1,-1)
__device__ __forceinline__ void quat_apply(const Quat q, float px, float py, float pz,
                                           float& ox, float& oy, float& oz) {
  Quat p;
  p.w = 0.0f; p.x = px; p.y = py; p.z = pz;
  const Quat t = quat_raw_mul(q, p);
  Quat qi;
  qi.w = q.w; qi.x = -q.x; qi.y = -q.y; qi.z = -q.z;
  const Quat r = quat_raw_mul(t, qi);
  ox = r.x; oy = r.y; oz = r.z;
}

// ---------------------------------------------------------------------------
// a1: rotate + gather of the valid fragments
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void se3_rotate_gather_kernel(
    const float* __restrict__ part_pcs, const float* __restrict__ pose,
    const int32_t* __restrict__ slot, float* __restrict__ out, int64_t F, int64_t N) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= F * N) return;
  const int64_t f = gid / N;
  const int64_t i = gid - f * N;
  const int64_t s = slot[f];
  const float* ps = pose + s * 7;
  Quat q;
  q.w = ps[3]; q.x = ps[4]; q.y = ps[5]; q.z = ps[6];
  q = quat_normalise(q);
  const float* p = part_pcs + (s * N + i) * 3;
  float ox, oy, oz;
  quat_apply(q, p[0], p[1], p[2], ox, oy, oz);
  float* o = out + gid * 3;
  o[0] = ox; o[1] = oy; o[2] = oz;
}

// a19: R(q)p + t with optional pre-scale and optional normalisation of q
__global__ __launch_bounds__(256) void pose_apply_kernel(
    const float* __restrict__ pts, const float* __restrict__ pose,
    const float* __restrict__ scale, float* __restrict__ out, int64_t n, int64_t N,
    int normalise) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n * N) return;
  const int64_t f = gid / N;
  const float* ps = pose + f * 7;
  Quat q;
  q.w = ps[3]; q.x = ps[4]; q.y = ps[5]; q.z = ps[6];
  if (normalise) q = quat_normalise(q);
  const float* p = pts + gid * 3;
  float x = p[0], y = p[1], z = p[2];
  if (scale) {
    const float sc = scale[f];
    x = x * sc; y = y * sc; z = z * sc;
  }
  float ox, oy, oz;
  quat_apply(q, x, y, z, ox, oy, oz);
  float* o = out + gid * 3;
  o[0] = ox + ps[0];
  o[1] = oy + ps[1];
  o[2] = oz + ps[2];
}

// ---------------------------------------------------------------------------
// wave-wide max of a float through DPP (no LDS traffic): quad swaps, half-row
// mirror, row mirror, then the four row results through v_readlane.
// ---------------------------------------------------------------------------

template <int CTRL>
__device__ __forceinline__ float dpp_max(float v) {
  const int iv = __float_as_int(v);
  const int o = __builtin_amdgcn_update_dpp(iv, iv, CTRL, 0xf, 0xf, false);
  return fmaxf(v, __int_as_float(o));
}

__device__ __forceinline__ float wave_max_f32(float v) {
  v = dpp_max<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_max<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_max<0x141>(v);  // row_half_mirror
  v = dpp_max<0x140>(v);  // row_mirror  -> every lane holds its row-of-16 max
  const int iv = __float_as_int(v);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(iv, 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(iv, 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(iv, 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(iv, 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// ---------------------------------------------------------------------------
// One step of the farthest-point chain, shared by fps_kernel and the fused sampling kernel.  The step is a dependent chain (centroid
// -> distances -> argmax -> next centroid), 255 of them for level 1: what counts is its LATENCY.
//   * distances two points at a time (v_pk_add_f32 / v_pk_mul_f32: the same correctly rounded sub, mul, add per element, in the
//     reference's order ((dx^2 + dy^2) + dz^2); -ffp-contract=off keeps them apart);
//   * the lane's best is a max tree over its running minima (v_max3_f32), the index is NOT carried through the loop: after the wave
//     maximum, the lowest lane holding it is found by ballot (blocked ownership: lowest lane = lowest index block) and the first of its
//     points equal to the maximum by PPT compares — torch.argmax's first-maximum rule, as before;
//   * waves exchange (maximum, index) as ONE 8-byte LDS write each and two 16-byte broadcast reads, behind a raw `s_waitcnt lgkmcnt(0);
//     s_barrier`: __syncthreads() also drains vmcnt, i.e. waited every step for the global stores of the previous selection.
// Measured (one puzzle in flight, rocprofv3): 1,000 -> 256 points 107.9 -> 94.0 us, 256 -> 128 33.2 -> 28.5 us.  Tried on top and
// dropped (slower: 123 / 35 us): the winner's coordinates carried through v_readlane and a 16-byte exchange slot instead of the LDS
// fetch by index, single-instruction v_max_f32_dpp steps and v_min / v_max without the IEEE-mode canonicalisation via inline asm —
// the select chains and the opaque asm (an s_nop after every statement) cost more than the broadcast LDS reads they replace.
// ---------------------------------------------------------------------------
typedef float f2v __attribute__((ext_vector_type(2)));

#ifdef PFPP_FPS_FULL_BARRIER      // lab: the compiler's full barrier (waits for vmcnt as well) instead of the raw one
__device__ __forceinline__ void lds_barrier() { __syncthreads(); }
#else
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif
template <int PPT>
struct FpsLane {
  static constexpr int NP = (PPT + 1) / 2;
  f2v x[NP], y[NP], z[NP], d[NP];      // this lane's points [t * PPT, (t + 1) * PPT) and their running minimum distances (pads: -1)

  __device__ __forceinline__ void load(const float* xs, const float* ys, const float* zs, int stride, int t, int N) {
#pragma unroll
    for (int k = 0; k < 2 * NP; ++k) {
      const int i = t * PPT + k;
      const bool ok = k < PPT && i < N;
#ifdef PFPP_FPS_VOLATILE_LOAD   // lab: dword reads of the points only (no ds_read_b96 / b64 merges)
      x[k >> 1][k & 1] = ok ? *(const volatile float*)&xs[(size_t)stride * i] : 0.0f;
      y[k >> 1][k & 1] = ok ? *(const volatile float*)&ys[(size_t)stride * i] : 0.0f;
      z[k >> 1][k & 1] = ok ? *(const volatile float*)&zs[(size_t)stride * i] : 0.0f;
#else
      x[k >> 1][k & 1] = ok ? xs[(size_t)stride * i] : 0.0f;
      y[k >> 1][k & 1] = ok ? ys[(size_t)stride * i] : 0.0f;
      z[k >> 1][k & 1] = ok ? zs[(size_t)stride * i] : 0.0f;
#endif
      d[k >> 1][k & 1] = ok ? __builtin_huge_valf() : -1.0f;   // pads can never win
    }
  }
  // running minima against the new centroid -> this lane's maximum
  __device__ __forceinline__ float update(float cx, float cy, float cz) {
#ifdef PFPP_FPS_NOPK      // lab: one point at a time (no packed fp32 instructions)
#pragma unroll
    for (int k = 0; k < 2 * NP; ++k) {
      const float dx = __fsub_rn(x[k >> 1][k & 1], cx), dy = __fsub_rn(y[k >> 1][k & 1], cy), dz = __fsub_rn(z[k >> 1][k & 1], cz);
      const float dd = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      d[k >> 1][k & 1] = fminf(d[k >> 1][k & 1], dd);
    }
#else
    const f2v c_x = {cx, cx}, c_y = {cy, cy}, c_z = {cz, cz};
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const f2v dx = x[q] - c_x, dy = y[q] - c_y, dz = z[q] - c_z;
      f2v dd = (dx * dx + dy * dy) + dz * dz;
#ifdef PFPP_FPS_PKNOP      // lab: an explicit wait state between the packed add and its first consumer
      asm volatile("s_nop 1" : "+v"(dd));
#endif
      d[q][0] = fminf(d[q][0], dd[0]);
      d[q][1] = fminf(d[q][1], dd[1]);
    }
#endif
    float b = d[0][0];
#pragma unroll
    for (int k = 1; k < 2 * NP; ++k) b = fmaxf(b, d[k >> 1][k & 1]);
    return b;
  }
  // first of this lane's points whose running minimum equals v (meaningful in the lane that holds the wave maximum)
  __device__ __forceinline__ int first_equal(float v) const {
    int kk = PPT - 1;
#pragma unroll
    for (int k = PPT - 2; k >= 0; --k) kk = (d[k >> 1][k & 1] == v) ? k : kk;
    return kk;
  }
};

// LDS words of the exchange slots of G waves: [2][G] (maximum, index) pairs, 16-byte aligned
#ifdef PFPP_FPS_CHECK
#define PFPP_FPS_SLOT_WORDS(G) (8 * (G) + 4)
#else
#define PFPP_FPS_SLOT_WORDS(G) (4 * (G) + 4)
#endif

// argmax over the G waves' lanes -> selected index (every thread of the G waves calls it; one raw barrier when G > 1)
#ifdef PFPP_FPS_CHECK      // lab (tools/diag/fps_race.py): every exchange entry carries (step, wave); readers count entries of another step / wave
__device__ unsigned long long pfpp_fps_violations[4];      // [0] entries of an OLDER step, [1] of a NEWER step, [2] wrong wave tag, [3] reads checked
#endif
template <int G, int PPT>
__device__ __forceinline__ int fps_argmax(const FpsLane<PPT>& pts, float lane_best, int t, int s, float* slot) {
  const int lane = t & 63, wave = t >> 6;
  const float wmax = wave_max_f32(lane_best);
  const unsigned long long m = __ballot(lane_best == wmax);
  const int widx = __builtin_amdgcn_readlane(t * PPT + pts.first_equal(wmax), __builtin_ctzll(m));
  if (G == 1) return widx;
#ifdef PFPP_FPS_CHECK
  {
    int4* cur4 = reinterpret_cast<int4*>(slot) + (s & 1) * G;
    if (lane == 0) cur4[wave] = make_int4(__float_as_int(wmax), widx, s, wave);
    lds_barrier();
    float bm4 = -1.0f; int bidx4 = 0;
#pragma unroll
    for (int w = 0; w < G; ++w) {
      const int4 c = cur4[w];
      if (lane == 0) {
        if (c.z < s) atomicAdd(&pfpp_fps_violations[0], 1ull);
        if (c.z > s) atomicAdd(&pfpp_fps_violations[1], 1ull);
        if (c.w != w) atomicAdd(&pfpp_fps_violations[2], 1ull);
        if (wave == 0 && w == 0) atomicAdd(&pfpp_fps_violations[3], 1ull);
      }
      const float d2 = __int_as_float(c.x);
      if (w == 0 || d2 > bm4) { bm4 = d2; bidx4 = c.y; }
    }
    return bidx4;
  }
#endif
  int2* cur = reinterpret_cast<int2*>(slot) + (s & 1) * G;
  if (lane == 0) cur[wave] = make_int2(__float_as_int(wmax), widx);
  lds_barrier();
  float bm = __int_as_float(cur[0].x);
  int bidx = cur[0].y;
#pragma unroll
  for (int w = 1; w < G; ++w) {
    const int2 c = cur[w];
    const float d2 = __int_as_float(c.x);
    if (d2 > bm) { bm = d2; bidx = c.y; }
  }
  return bidx;
}

// ---------------------------------------------------------------------------
// a2: farthest point sampling.  One workgroup per fragment; the fragment's
// points live in registers (PPT per thread, blocked ownership: thread t owns
// indices [t*PPT, (t+1)*PPT) so "lowest lane with the maximum" == "lowest
// index with the maximum" == torch.argmax's first-maximum rule) and in LDS
// (for the broadcast fetch of the newly selected centroid).  One barrier per
// selected point when NWAVES > 1, none when the fragment fits one wave.
// ---------------------------------------------------------------------------
// LDSPTS = false (clouds beyond the LDS budget, the merged clouds of auto_aggl): the selected centroid is
// fetched from global memory instead.  `start` (may be NULL = 0) is the first selected index per fragment
// (torch_cluster.fps random_start, utils/node_merge_utils.py:219).
template <int NWAVES, int PPT, bool LDSPTS = true>
__global__ __launch_bounds__(NWAVES * 64) void fps_kernel(
    const float* __restrict__ xyz, int32_t* __restrict__ idx_out,
    float* __restrict__ new_xyz, int N, int S, const int32_t* __restrict__ start = nullptr) {
  extern __shared__ __align__(16) float fps_smem[];
  constexpr int NT = NWAVES * 64;
  const int f = blockIdx.x;
  const int tid = threadIdx.x;
  const float* src = xyz + (size_t)f * N * 3;
  const float* s_pts = LDSPTS ? fps_smem : src;             // [3*N] (+pad)
  float* s_slot = fps_smem + (LDSPTS ? ((3 * N + 3) & ~3) : 0);          // exchange slots of the waves (PFPP_FPS_SLOT_WORDS)
  if (LDSPTS) {
    for (int i = tid; i < 3 * N; i += NT) fps_smem[i] = src[i];
    __syncthreads();
  }
  FpsLane<PPT> pts;
  pts.load(s_pts, s_pts + 1, s_pts + 2, 3, tid, N);
  int sel = start ? min(max(start[f], 0), N - 1) : 0;
  float cx = s_pts[3 * sel + 0], cy = s_pts[3 * sel + 1], cz = s_pts[3 * sel + 2];
  int32_t* o_idx = idx_out + (size_t)f * S;
  float* o_xyz = new_xyz + (size_t)f * S * 3;
#ifdef PFPP_FPS_LDS_OUT
  // lab: the selections collect in LDS ([S] indices, [3 S] coordinates behind the exchange slots) and leave in one coalesced pass
  int32_t* st_idx = reinterpret_cast<int32_t*>(s_slot + PFPP_FPS_SLOT_WORDS(NWAVES));
  float* st_xyz = reinterpret_cast<float*>(st_idx + S);
#endif

  for (int s = 0;;) {
    if (tid == 0) {
#ifdef PFPP_FPS_LDS_OUT
      st_idx[s] = sel;
      st_xyz[3 * s + 0] = cx;
      st_xyz[3 * s + 1] = cy;
      st_xyz[3 * s + 2] = cz;
#else
      o_idx[s] = sel;
      o_xyz[3 * s + 0] = cx;
      o_xyz[3 * s + 1] = cy;
      o_xyz[3 * s + 2] = cz;
#endif
    }
    if (++s >= S) break;
    const float best = pts.update(cx, cy, cz);
    sel = fps_argmax<NWAVES, PPT>(pts, best, tid, s, s_slot);
    cx = s_pts[3 * sel + 0];
    cy = s_pts[3 * sel + 1];
    cz = s_pts[3 * sel + 2];
  }
#ifdef PFPP_FPS_LDS_OUT
  __syncthreads();
  for (int i = tid; i < S; i += NT) o_idx[i] = st_idx[i];
  for (int i = tid; i < 3 * S; i += NT) o_xyz[i] = st_xyz[i];
#endif
}

// ---------------------------------------------------------------------------
// a3: ball query.  One wave per centroid scans the fragment's points (SoA in
// LDS, with |p|^2 precomputed) 64 at a time in index order; kept indices are
// compacted in order with ballot + mbcnt; the scan stops at nsample hits.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ball_query_kernel(
    const float* __restrict__ xyz, const float* __restrict__ new_xyz,
    int32_t* __restrict__ idx, int N, int S, int ns, float r2, int cpb) {
  extern __shared__ __align__(16) float bq_smem[];
  float* xs = bq_smem;
  float* ys = xs + N;
  float* zs = ys + N;
  float* pp = zs + N;
  const int f = blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const float* src = xyz + (size_t)f * N * 3;
  for (int i = tid; i < N; i += 256) {
    const float x = src[3 * i + 0], y = src[3 * i + 1], z = src[3 * i + 2];
    xs[i] = x; ys[i] = y; zs[i] = z;
    pp[i] = __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
  }
  __syncthreads();
  const int c_begin = blockIdx.x * cpb;
  const int c_end = min(c_begin + cpb, S);
  for (int c = c_begin + wave; c < c_end; c += 4) {
    const float* cp = new_xyz + ((size_t)f * S + c) * 3;
    const float cx = cp[0], cy = cp[1], cz = cp[2];
    const float nn = __fadd_rn(__fadd_rn(__fmul_rn(cx, cx), __fmul_rn(cy, cy)), __fmul_rn(cz, cz));
    int32_t* out = idx + ((size_t)f * S + c) * ns;
    int cnt = 0;
    int first = N;  // reference value when nothing is in range (pn2_utils.py:106-111)
    for (int base = 0; base < N && cnt < ns; base += 64) {
      const int i = base + lane;
      const bool ok = i < N;
      const float x = ok ? xs[i] : 0.0f;
      const float y = ok ? ys[i] : 0.0f;
      const float z = ok ? zs[i] : 0.0f;
      const float q = ok ? pp[i] : 0.0f;
      // K=3 matmul on the CPU BLAS: fma chain over k = 0,1,2
      const float dot = __fmaf_rn(cz, z, __fmaf_rn(cy, y, __fmul_rn(cx, x)));
      const float d = __fadd_rn(__fadd_rn(__fmul_rn(-2.0f, dot), nn), q);
      const bool keep = ok && !(d > r2);
      const unsigned long long m = __ballot(keep);
      if (m != 0ull) {
        if (cnt == 0) first = base + __builtin_ctzll(m);
        const int prefix = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                     __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
        const int pos = cnt + prefix;
        if (keep && pos < ns) out[pos] = i;
        cnt += __builtin_popcountll(m);
      }
    }
    const int total = cnt < ns ? cnt : ns;
    if (lane >= total && lane < ns) out[lane] = first;
  }
}

// ---------------------------------------------------------------------------
// a2 + a3 for all three set-abstraction levels of one fragment in ONE kernel (PN2.encode's sampling chain, pn2.py:57-68:
// sample_and_group of sa1, sa2, sa3 depends on coordinates only, never on the features).  One 16-wave workgroup per
// fragment, the fragment's points as SoA (+ |p|^2) in LDS:
//   phase A  FPS of level 1 on four waves (the same blocked-ownership argmax chain as fps_kernel);
//   phase B  wave 0 runs the FPS chains of levels 2 and 3 on the level-1 / level-2 centroids (they fit one wave: no barrier),
//            while the other 15 waves scan the ball queries of level 1;
//   phase C  all waves: ball queries of levels 2 and 3.
// Same arithmetic, in the same order, as fps_kernel / ball_query_kernel: indices are bit-identical (tested).  Six launches
// and their gaps become one, and the ball queries of level 1 run in the shadow of the later FPS chains.
// ---------------------------------------------------------------------------
struct SampleLevelP {
  int S, ns;
  float r2;
  int32_t* fps_idx;   // [F, S] or null
  float* new_xyz;     // [F, S, 3]
  int32_t* ball;      // [F, S, ns]
};
struct SampleP {
  const float* xyz;
  int N;
  SampleLevelP lv[3];
};

// FPS over N points held as SoA in LDS by G * 64 threads (thread t of the group owns [t * PPT, (t + 1) * PPT)); writes the S
// selected indices / coordinates to global memory and the coordinates (+ |c|^2) to the LDS arrays of the next level.
// G > 1: every thread of the workgroup must call it (one __syncthreads per selected point).
template <int G, int PPT>
__device__ __forceinline__ void fps_chain(const float* xs, const float* ys, const float* zs, int N, int S, int t, int32_t* o_idx,
                                          float* o_xyz, float* nx, float* ny, float* nz, float* npp, float* slot) {
  FpsLane<PPT> pts;
  pts.load(xs, ys, zs, 1, t, N);
  int sel = 0;
  float cx = xs[0], cy = ys[0], cz = zs[0];
  for (int s = 0;;) {
    if (t == 0) {
      if (o_idx) o_idx[s] = sel;
      o_xyz[3 * s + 0] = cx; o_xyz[3 * s + 1] = cy; o_xyz[3 * s + 2] = cz;
      nx[s] = cx; ny[s] = cy; nz[s] = cz;
      npp[s] = __fadd_rn(__fadd_rn(__fmul_rn(cx, cx), __fmul_rn(cy, cy)), __fmul_rn(cz, cz));
    }
    if (++s >= S) break;
    const float best = pts.update(cx, cy, cz);
    sel = fps_argmax<G, PPT>(pts, best, t, s, slot);
    cx = xs[sel]; cy = ys[sel]; cz = zs[sel];
  }
}

// one wave: ball query of centroid (cx, cy, cz) over N points (SoA + |p|^2 in LDS) -> out[ns]
__device__ __forceinline__ void ball_scan(const float* xs, const float* ys, const float* zs, const float* pp, int N, float cx, float cy,
                                          float cz, int ns, float r2, int32_t* out, int lane) {
  const float nn = __fadd_rn(__fadd_rn(__fmul_rn(cx, cx), __fmul_rn(cy, cy)), __fmul_rn(cz, cz));
  int cnt = 0;
  int first = N;
  for (int base = 0; base < N && cnt < ns; base += 64) {
    const int i = base + lane;
    const bool ok = i < N;
    const float x = ok ? xs[i] : 0.0f;
    const float y = ok ? ys[i] : 0.0f;
    const float z = ok ? zs[i] : 0.0f;
    const float q = ok ? pp[i] : 0.0f;
    const float dot = __fmaf_rn(cz, z, __fmaf_rn(cy, y, __fmul_rn(cx, x)));
    const float d = __fadd_rn(__fadd_rn(__fmul_rn(-2.0f, dot), nn), q);
    const bool keep = ok && !(d > r2);
    const unsigned long long m = __ballot(keep);
    if (m != 0ull) {
      if (cnt == 0) first = base + __builtin_ctzll(m);
      const int prefix = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
      const int pos = cnt + prefix;
      if (keep && pos < ns) out[pos] = i;
      cnt += __builtin_popcountll(m);
    }
  }
  const int total = cnt < ns ? cnt : ns;
  if (lane >= total && lane < ns) out[lane] = first;
}

constexpr int SL_WAVES = 16;      // waves per fragment: 4 run the level-1 FPS chain, all of them share the ball queries
template <int PPT>
__global__ __launch_bounds__(SL_WAVES * 64) void sample_levels_kernel(const SampleP p) {
  extern __shared__ __align__(16) float sl_smem[];
  const int N = p.N, S1 = p.lv[0].S, S2 = p.lv[1].S, S3 = p.lv[2].S;
  float* xs = sl_smem;          float* ys = xs + N;  float* zs = ys + N;  float* pp = zs + N;
  float* c1x = pp + N;          float* c1y = c1x + S1; float* c1z = c1y + S1; float* c1p = c1z + S1;
  float* c2x = c1p + S1;        float* c2y = c2x + S2; float* c2z = c2y + S2; float* c2p = c2z + S2;
  float* c3x = c2p + S2;        float* c3y = c3x + S3; float* c3z = c3y + S3; float* c3p = c3z + S3;
  float* slot = sl_smem + (((c3p + S3) - sl_smem + 3) & ~3);      // exchange slots of the four FPS waves (PFPP_FPS_SLOT_WORDS(4)), 16-byte aligned
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* src = p.xyz + (size_t)f * N * 3;
  for (int i = tid; i < N; i += SL_WAVES * 64) {
    const float x = src[3 * i + 0], y = src[3 * i + 1], z = src[3 * i + 2];
    xs[i] = x; ys[i] = y; zs[i] = z;
    pp[i] = __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
  }
  __syncthreads();
  // phase A: level-1 FPS on waves 0-3; the others only keep the barrier count (one per selected point)
  if (wave < 4) {
    fps_chain<4, PPT>(xs, ys, zs, N, S1, tid, p.lv[0].fps_idx ? p.lv[0].fps_idx + (size_t)f * S1 : nullptr,
                      p.lv[0].new_xyz + (size_t)f * S1 * 3, c1x, c1y, c1z, c1p, slot);
  } else {
    for (int s = 1; s < S1; ++s) lds_barrier();
  }
  __syncthreads();
  // phase B: wave 0 -> FPS of levels 2 and 3; waves 1-3 -> ball queries of level 1
  if (wave == 0) {
    fps_chain<1, 4>(c1x, c1y, c1z, S1, S2, lane, p.lv[1].fps_idx ? p.lv[1].fps_idx + (size_t)f * S2 : nullptr,
                    p.lv[1].new_xyz + (size_t)f * S2 * 3, c2x, c2y, c2z, c2p, nullptr);
    __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): lane 0's LDS writes of c2 before the wave reads them back
    fps_chain<1, 2>(c2x, c2y, c2z, S2, S3, lane, p.lv[2].fps_idx ? p.lv[2].fps_idx + (size_t)f * S3 : nullptr,
                    p.lv[2].new_xyz + (size_t)f * S3 * 3, c3x, c3y, c3z, c3p, nullptr);
  } else {
    int32_t* out = p.lv[0].ball + (size_t)f * S1 * p.lv[0].ns;
    for (int c = wave - 1; c < S1; c += SL_WAVES - 1)
      ball_scan(xs, ys, zs, pp, N, c1x[c], c1y[c], c1z[c], p.lv[0].ns, p.lv[0].r2, out + (size_t)c * p.lv[0].ns, lane);
  }
  __syncthreads();
  // phase C: ball queries of levels 2 and 3, all waves
  {
    int32_t* out = p.lv[1].ball + (size_t)f * S2 * p.lv[1].ns;
    for (int c = wave; c < S2; c += SL_WAVES)
      ball_scan(c1x, c1y, c1z, c1p, S1, c2x[c], c2y[c], c2z[c], p.lv[1].ns, p.lv[1].r2, out + (size_t)c * p.lv[1].ns, lane);
    out = p.lv[2].ball + (size_t)f * S3 * p.lv[2].ns;
    for (int c = wave; c < S3; c += SL_WAVES)
      ball_scan(c2x, c2y, c2z, c2p, S2, c3x[c], c3y[c], c3z[c], p.lv[2].ns, p.lv[2].r2, out + (size_t)c * p.lv[2].ns, lane);
  }
}

// ---------------------------------------------------------------------------
// a4: grouping.  One thread per float4 of the output row
//   [ feats[f, id, 0:D] | xyz[f,id] - new_xyz[f,s] | 0 ]
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void group_gather_kernel(
    const float* __restrict__ xyz, const float* __restrict__ new_xyz,
    const float* __restrict__ feats, const int32_t* __restrict__ idx,
    float* __restrict__ out, int64_t total4, int N, int S, int ns, int D, int ldo4) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total4) return;
  const int64_t row = gid / ldo4;
  const int c4 = (int)(gid - row * ldo4);
  const int64_t fs = row / ns;      // f*S + s
  const int64_t f = fs / S;
  int id = idx[row];
  id = id < N ? id : N - 1;          // memory safety only; see ball_query_kernel
  float4 v;
  const int col = c4 * 4;
  if (col < D) {
    v = *reinterpret_cast<const float4*>(feats + ((size_t)f * N + id) * D + col);
  } else if (col == D) {
    const float* p = xyz + ((size_t)f * N + id) * 3;
    const float* c = new_xyz + fs * 3;
    v.x = __fsub_rn(p[0], c[0]);
    v.y = __fsub_rn(p[1], c[1]);
    v.z = __fsub_rn(p[2], c[2]);
    v.w = 0.0f;
  } else {
    v = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  *reinterpret_cast<float4*>(out + gid * 4) = v;
}

// ---------------------------------------------------------------------------
// a19: pose composition (get_param / extract_final_pred_trans_rots)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void quat_to_matrix(const float* q, float* m /*9*/) {
  const float r = q[0], i = q[1], j = q[2], k = q[3];
  const float two_s = 2.0f / (((r * r + i * i) + j * j) + k * k);
  m[0] = 1.0f - two_s * (j * j + k * k);
  m[1] = two_s * (i * j - k * r);
  m[2] = two_s * (i * k + j * r);
  m[3] = two_s * (i * j + k * r);
  m[4] = 1.0f - two_s * (i * i + k * k);
  m[5] = two_s * (j * k - i * r);
  m[6] = two_s * (i * k - j * r);
  m[7] = two_s * (j * k + i * r);
  m[8] = 1.0f - two_s * (i * i + j * j);
}

__device__ __forceinline__ float sqrt_pos(float x) { return x > 0.0f ? sqrtf(x) : 0.0f; }

__device__ __forceinline__ void matrix_to_quat(const float* m, float* q) {
  const float m00 = m[0], m01 = m[1], m02 = m[2], m10 = m[3], m11 = m[4], m12 = m[5],
              m20 = m[6], m21 = m[7], m22 = m[8];
  float qa[4];
  qa[0] = sqrt_pos(1.0f + m00 + m11 + m22);
  qa[1] = sqrt_pos(1.0f + m00 - m11 - m22);
  qa[2] = sqrt_pos(1.0f - m00 + m11 - m22);
  qa[3] = sqrt_pos(1.0f - m00 - m11 + m22);
  float cand[4][4] = {
      {qa[0] * qa[0], m21 - m12, m02 - m20, m10 - m01},
      {m21 - m12, qa[1] * qa[1], m10 + m01, m02 + m20},
      {m02 - m20, m10 + m01, qa[2] * qa[2], m12 + m21},
      {m10 - m01, m20 + m02, m21 + m12, qa[3] * qa[3]}};
  int best = 0;
#pragma unroll
  for (int a = 1; a < 4; ++a)
    if (qa[a] > qa[best]) best = a;   // first maximum, like argmax
  float o[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    if (a == best) {
      const float den = 2.0f * fmaxf(qa[a], 0.1f);
#pragma unroll
      for (int c = 0; c < 4; ++c) o[c] = cand[a][c] / den;
    }
  }
  // standardize_quaternion: non-negative real part
  const float sgn = o[0] < 0.0f ? -1.0f : 1.0f;
#pragma unroll
  for (int c = 0; c < 4; ++c) q[c] = sgn < 0.0f ? -o[c] : o[c];
}

__global__ __launch_bounds__(64) void pose_compose_kernel(
    const float* __restrict__ pose, const int32_t* __restrict__ pivot,
    const float* __restrict__ init_pose, const uint8_t* __restrict__ has_init,
    float* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* ps = pose + (int64_t)pivot[i] * 7;
  float R[9];
  quat_to_matrix(ps + 3, R);
  float t[3] = {ps[0], ps[1], ps[2]};
  float Ro[9], to[3];
  if (has_init && has_init[i]) {
    const float* I = init_pose + i * 16;
    // [R t; 0 1] @ I  (torch matmul of 4x4: k-ordered accumulation)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float acc = R[r * 3 + 0] * I[0 * 4 + c];
        acc = fmaf(R[r * 3 + 1], I[1 * 4 + c], acc);
        acc = fmaf(R[r * 3 + 2], I[2 * 4 + c], acc);
        acc = fmaf(t[r], I[3 * 4 + c], acc);
        Ro[r * 3 + c] = acc;
      }
      float acc = R[r * 3 + 0] * I[0 * 4 + 3];
      acc = fmaf(R[r * 3 + 1], I[1 * 4 + 3], acc);
      acc = fmaf(R[r * 3 + 2], I[2 * 4 + 3], acc);
      acc = fmaf(t[r], I[3 * 4 + 3], acc);
      to[r] = acc;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 9; ++k) Ro[k] = R[k];
    to[0] = t[0]; to[1] = t[1]; to[2] = t[2];
  }
  float q[4];
  matrix_to_quat(Ro, q);
  float* o = out + i * 7;
  o[0] = to[0]; o[1] = to[1]; o[2] = to[2];
  o[3] = q[0]; o[4] = q[1]; o[5] = q[2]; o[6] = q[3];
}

template <int NWAVES, int PPT, bool LDSPTS = true>
int launch_fps(const float* xyz, int32_t* idx, float* new_xyz, int64_t F, int N, int S,
               hipStream_t st, const int32_t* start = nullptr) {
  size_t smem = (size_t)((LDSPTS ? ((3 * N + 3) & ~3) : 0) + PFPP_FPS_SLOT_WORDS(NWAVES)) * sizeof(float);
#ifdef PFPP_FPS_LDS_OUT
  smem += (size_t)4 * S * sizeof(float);
#endif
  hipLaunchKernelGGL((fps_kernel<NWAVES, PPT, LDSPTS>), dim3((unsigned)F), dim3(NWAVES * 64), smem, st,
                     xyz, idx, new_xyz, N, S, start);
  return pfpp::check_launch("pfpp_fps");
}

}  // namespace

extern "C" int pfpp_se3_rotate_gather(const float* part_pcs, const float* pose,
                                      const int32_t* slot, float* out, int64_t F, int64_t N,
                                      pfpp_stream_t stream) {
  PFPP_REQUIRE(part_pcs && pose && slot && out, "null pointer");
  PFPP_REQUIRE(F >= 0 && N > 0, "bad sizes");
  if (F == 0) return PFPP_OK;
  const int64_t total = F * N;
  hipLaunchKernelGGL(se3_rotate_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     pfpp::as_stream(stream), part_pcs, pose, slot, out, F, N);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_pose_apply(const float* pts, const float* pose, const float* scale,
                               float* out, int64_t n, int64_t N, int normalise,
                               pfpp_stream_t stream) {
  PFPP_REQUIRE(pts && pose && out, "null pointer");
  PFPP_REQUIRE(n >= 0 && N > 0, "bad sizes");
  if (n == 0) return PFPP_OK;
  const int64_t total = n * N;
  hipLaunchKernelGGL(pose_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     pfpp::as_stream(stream), pts, pose, scale, out, n, N, normalise);
  return pfpp::check_launch(__func__);
}

static int fps_impl(const float* xyz, int32_t* idx, float* new_xyz, int64_t F, int64_t N, int64_t S,
                    const int32_t* start, pfpp_stream_t stream) {
  PFPP_REQUIRE(xyz && idx && new_xyz, "null pointer");
  PFPP_REQUIRE(F >= 0 && N >= 1 && S >= 1 && S <= N, "need 1 <= S <= N");
  PFPP_SUPPORTED(N <= 32768, "N > 32768");
  if (F == 0) return PFPP_OK;
  hipStream_t st = pfpp::as_stream(stream);
  const int n = (int)N, s = (int)S;
  if (N <= 64) return launch_fps<1, 1>(xyz, idx, new_xyz, F, n, s, st, start);
  if (N <= 128) return launch_fps<1, 2>(xyz, idx, new_xyz, F, n, s, st, start);
  if (N <= 256) return launch_fps<1, 4>(xyz, idx, new_xyz, F, n, s, st, start);
  if (N <= 512) return launch_fps<2, 4>(xyz, idx, new_xyz, F, n, s, st, start);
  if (N <= 1024) return launch_fps<4, 4>(xyz, idx, new_xyz, F, n, s, st, start);
  if (N <= 2048) return launch_fps<4, 8>(xyz, idx, new_xyz, F, n, s, st, start);
  if (N <= 4096) return launch_fps<4, 16>(xyz, idx, new_xyz, F, n, s, st, start);
  // merged clouds (up to 20 parts x 1000 points, node_merge_utils.py:212-220): points in registers only
  if (N <= 16384) return launch_fps<16, 16, false>(xyz, idx, new_xyz, F, n, s, st, start);
  return launch_fps<16, 32, false>(xyz, idx, new_xyz, F, n, s, st, start);
}

#ifdef PFPP_FPS_CHECK
extern "C" int pfpp_lab_fps_violations(unsigned long long* out4, int reset) {
  if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(pfpp_fps_violations), 32) != hipSuccess) return -1;
  if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(pfpp_fps_violations), z, 32) != hipSuccess) return -1; }
  return 0;
}
#endif

extern "C" int pfpp_fps(const float* xyz, int32_t* idx, float* new_xyz, int64_t F, int64_t N,
                        int64_t S, pfpp_stream_t stream) {
  return fps_impl(xyz, idx, new_xyz, F, N, S, nullptr, stream);
}

extern "C" int pfpp_fps_start(const float* xyz, int32_t* idx, float* new_xyz, int64_t F, int64_t N,
                              int64_t S, const int32_t* start, pfpp_stream_t stream) {
  return fps_impl(xyz, idx, new_xyz, F, N, S, start, stream);
}

extern "C" int pfpp_ball_query(const float* xyz, const float* new_xyz, int32_t* idx, int64_t F,
                               int64_t N, int64_t S, int64_t nsample, float r2,
                               pfpp_stream_t stream) {
  PFPP_REQUIRE(xyz && new_xyz && idx, "null pointer");
  PFPP_REQUIRE(F >= 0 && N >= 1 && S >= 1 && nsample >= 1, "bad sizes");
  PFPP_SUPPORTED(nsample <= 64, "nsample > 64");
  PFPP_SUPPORTED(N <= 8192, "N > 8192");
  PFPP_SUPPORTED(F <= 65535, "F > 65535 fragments per call");
  if (F == 0) return PFPP_OK;
  const int cpb = 16;  // centroids per workgroup (4 per wave)
  const dim3 grid((unsigned)((S + cpb - 1) / cpb), (unsigned)F);
  const size_t smem = (size_t)N * 4 * sizeof(float);
  hipLaunchKernelGGL(ball_query_kernel, grid, dim3(256), smem, pfpp::as_stream(stream), xyz,
                     new_xyz, idx, (int)N, (int)S, (int)nsample, r2, cpb);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_sample_levels(const float* xyz, int64_t F, int64_t N, const pfpp_sample_level* levels, pfpp_stream_t stream) {
  PFPP_REQUIRE(xyz && levels, "null pointer");
  PFPP_REQUIRE(F >= 0 && N >= 1, "bad sizes");
  SampleP p;
  p.xyz = xyz; p.N = (int)N;
  int64_t prev = N;
  for (int l = 0; l < 3; ++l) {
    const pfpp_sample_level& v = levels[l];
    PFPP_REQUIRE(v.new_xyz && v.ball_idx, "level outputs: new_xyz and ball_idx are required");
    PFPP_REQUIRE(v.S >= 1 && v.S <= prev && v.nsample >= 1, "need 1 <= S <= points of the level above, nsample >= 1");
    PFPP_SUPPORTED(v.nsample <= 64, "nsample > 64");
    p.lv[l].S = (int)v.S; p.lv[l].ns = (int)v.nsample; p.lv[l].r2 = v.r2;
    p.lv[l].fps_idx = v.fps_idx; p.lv[l].new_xyz = v.new_xyz; p.lv[l].ball = v.ball_idx;
    prev = v.S;
  }
  PFPP_SUPPORTED(N <= 2048 && levels[0].S <= 256 && levels[1].S <= 128, "fused sampling: N <= 2048, S1 <= 256, S2 <= 128 (levels 2 and 3 run on one wave)");
  if (F == 0) return PFPP_OK;
  const size_t smem = ((size_t)4 * (N + levels[0].S + levels[1].S + levels[2].S) + 4 + PFPP_FPS_SLOT_WORDS(4)) * sizeof(float);
  hipStream_t st = pfpp::as_stream(stream);
  const dim3 grid((unsigned)F), block(SL_WAVES * 64);
  if (N <= 512) hipLaunchKernelGGL(sample_levels_kernel<2>, grid, block, smem, st, p);
  else if (N <= 1024) hipLaunchKernelGGL(sample_levels_kernel<4>, grid, block, smem, st, p);
  else hipLaunchKernelGGL(sample_levels_kernel<8>, grid, block, smem, st, p);
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_group_gather(const float* xyz, const float* new_xyz, const float* feats,
                                 const int32_t* idx, float* out, int64_t F, int64_t N, int64_t S,
                                 int64_t ns, int64_t D, int64_t ldo, pfpp_stream_t stream) {
  PFPP_REQUIRE(xyz && new_xyz && idx && out, "null pointer");
  PFPP_REQUIRE(D == 0 || feats, "feats is NULL with D > 0");
  PFPP_REQUIRE(D % 4 == 0 && ldo % 4 == 0 && ldo >= D + 4, "need D%4==0, ldo%4==0, ldo>=D+4");
  PFPP_REQUIRE(pfpp::aligned16(out) && (D == 0 || pfpp::aligned16(feats)), "16-byte alignment");
  const int64_t total4 = F * S * ns * (ldo / 4);
  if (total4 == 0) return PFPP_OK;
  hipLaunchKernelGGL(group_gather_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0,
                     pfpp::as_stream(stream), xyz, new_xyz, feats, idx, out, total4, (int)N, (int)S,
                     (int)ns, (int)D, (int)(ldo / 4));
  return pfpp::check_launch(__func__);
}

extern "C" int pfpp_pose_compose(const float* pose, const int32_t* pivot, const float* init_pose,
                                 const uint8_t* has_init, float* out, int64_t n,
                                 pfpp_stream_t stream) {
  PFPP_REQUIRE(pose && pivot && out, "null pointer");
  PFPP_REQUIRE(!has_init || init_pose, "has_init without init_pose");
  if (n == 0) return PFPP_OK;
  hipLaunchKernelGGL(pose_compose_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0,
                     pfpp::as_stream(stream), pose, pivot, init_pose, has_init, out, n);
  return pfpp::check_launch(__func__);
}
