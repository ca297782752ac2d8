// Training-side GEMMs (a17, backward of the linear layers of denoiser_transformer.py / attention.py):
//
//     dX[M,Kin]   = dY[M,Nout] . W[Nout,Kin]          "NN": A row-major, W k-major
//     dW[Nout,Kin] = dY[M,Nout]^T . X[M,Kin]          "TN": both operands k-major (contraction over rows)
//
// Same split-f16 arithmetic, LDS layout and wave tiling as gemm.hip's forward kernel; what differs is
// how a tile reaches LDS.  A k-major operand (element (r,k) at base[k*ld + r]) is read as 16-byte
// loads along r for 4 consecutive k; the 4x4 block is transposed in registers (renaming only) and
// written as four 8-byte k-runs, so the fragment reads and the MFMA loop are unchanged — no transposed
// copy of dY, X or the weights ever exists in HBM.
//
// Gradient operands are small numbers (1e-3 .. 1e-7): `a_scale` / `w_scale` (powers of two, exact in
// fp32) lift them into the normal f16 range before the hi/lo split and the epilogue divides back.
//
// Weight gradients have few output tiles and a long contraction (all tokens): the K range is cut into
// `split_k` chunks over blockIdx.y which accumulate with hardware fp32 atomics (`accumulate`, also
// what lets several micro-batches add into one .grad buffer).
#include <stdlib.h>

#include "gemm_common.h"

namespace {

using namespace pfpp_gemm_detail;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;
constexpr int LDH = BK + 8;

struct GradP {
  const float* A; const float* W; float* C;
  int M, N, K;
  int64_t lda, ldw, ldc;
  int64_t sA, sW, sC;
  int k_chunk;              // K range per blockIdx.y (multiple of BK)
  int accumulate;           // C += result (else C = result)
  int atomic;               // the += goes through hardware atomics (several workgroups add into one tile: split_k > 1)
  float a_scale, w_scale, alpha;
  int tiles_n, tiles_m, group_m;
  int splits;               // grouped launch only: K chunks of this problem
  int kxcd;                 // split-K launch as a 1-D grid: an XCD walks ONE K chunk over many tiles (see gemm_grad_kernel)
  float* colsum;            // weight-gradient form: colsum[m] += sum_k A[m, k] (the bias gradient), taken from the A tiles as they pass
};

__device__ __forceinline__ void split4s(const float4 v, float s, half4& hi, half4& lo) {
  const float x[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const _Float16 h = (_Float16)x[e];
    hi[e] = h;
    lo[e] = (_Float16)(x[e] - (float)h);
  }
}

// staging rows with bits 0 and 2 swapped: the two rows of an LDS store group are 4 apart (disjoint banks with 80-byte rows)
__device__ __forceinline__ int swap02(int r) { return (r & ~5) | ((r & 1) << 2) | ((r >> 2) & 1); }

// One operand tile: R rows (output rows or columns) x BK of the contraction -> hi/lo planes [R][LDH].
template <int R, int NTHR, bool KM>
struct OperandLoader {
  // row-major: 8 lanes x float4 cover the 32 k of a row;  k-major: unit = (4 k) x (4 r)
  static constexpr int UNITS = KM ? (R / 4) * 8 : R * 8;
  static constexpr int IT = (UNITS + NTHR - 1) / NTHR;
  float4 reg[IT][KM ? 4 : 1];

  __device__ __forceinline__ void load(const float* base, int64_t ld, int r0, int rmax, int k0, int kend, int tid) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int u = tid + it * NTHR;
      if (UNITS % NTHR != 0 && u >= UNITS) break;
      if constexpr (KM) {
        const int kg = u & 7, c4 = u >> 3;
        const int r = r0 + c4 * 4;
        const bool r_ok = r < rmax;            // extents along r are multiples of 4 (checked on the host)
        const float* src = base + (int64_t)(k0 + 4 * kg) * ld + (r_ok ? r : 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool ok = r_ok && (k0 + 4 * kg + j) < kend;
          reg[it][j] = ok ? *reinterpret_cast<const float4*>(src + j * ld) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      } else {
        const int row = swap02(u >> 3), c4 = u & 7;
        const float* src = base + (int64_t)min(r0 + row, rmax - 1) * ld;
        const int k = k0 + c4 * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k + 4 <= kend) {
          v = *reinterpret_cast<const float4*>(src + k);
        } else if (k < kend) {
          v = *reinterpret_cast<const float4*>(src + k);   // inside the (4-padded) row; mask the tail
          if (k + 1 >= kend) v.y = 0.f;
          if (k + 2 >= kend) v.z = 0.f;
          v.w = 0.f;
        }
        reg[it][0] = v;
      }
    }
  }

  // interior K-tile: no bounds along the contraction, rows past the edge are read from row 0 instead (they only feed
  // output rows / columns the epilogue never stores) — no conditional loads, so the steady-state loop stays one region
  __device__ __forceinline__ void load_full(const float* base, int64_t ld, int r0, int rmax, int k0, int tid) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int u = tid + it * NTHR;
      if (UNITS % NTHR != 0 && u >= UNITS) break;
      if constexpr (KM) {
        const int kg = u & 7, c4 = u >> 3;
        const int r = r0 + c4 * 4;
        const float* src = base + (int64_t)(k0 + 4 * kg) * ld + (r < rmax ? r : 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) reg[it][j] = *reinterpret_cast<const float4*>(src + j * ld);
      } else {
        const int row = swap02(u >> 3), c4 = u & 7;
        reg[it][0] = *reinterpret_cast<const float4*>(base + (int64_t)min(r0 + row, rmax - 1) * ld + k0 + c4 * 4);
      }
    }
  }

  // CS: also add the tile's 4 k-rows into `cs[it]` (per-thread partial sums over the contraction for its 4 output rows: the
  // bias gradient rides along with the weight gradient, k-major operand only)
  template <bool CS = false>
  __device__ __forceinline__ void store(_Float16* hi_plane, _Float16* lo_plane, float scale, int tid, float4* cs = nullptr) const {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int u = tid + it * NTHR;
      if (UNITS % NTHR != 0 && u >= UNITS) break;
      if constexpr (KM) {
        const int kg = u & 7, c4 = u >> 3;
        const float4 a = reg[it][0], b = reg[it][1], c = reg[it][2], d = reg[it][3];
        if constexpr (CS) {
          cs[it].x += (a.x + b.x) + (c.x + d.x);
          cs[it].y += (a.y + b.y) + (c.y + d.y);
          cs[it].z += (a.z + b.z) + (c.z + d.z);
          cs[it].w += (a.w + b.w) + (c.w + d.w);
        }
        const float4 col[4] = {make_float4(a.x, b.x, c.x, d.x), make_float4(a.y, b.y, c.y, d.y),
                               make_float4(a.z, b.z, c.z, d.z), make_float4(a.w, b.w, c.w, d.w)};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          half4 hi, lo;
          split4s(col[e], scale, hi, lo);
          const int off = (c4 * 4 + e) * LDH + 4 * kg;
          *reinterpret_cast<half4*>(hi_plane + off) = hi;
          *reinterpret_cast<half4*>(lo_plane + off) = lo;
        }
      } else {
        const int row = swap02(u >> 3), c4 = u & 7;
        half4 hi, lo;
        split4s(reg[it][0], scale, hi, lo);
        const int off = row * LDH + c4 * 4;
        *reinterpret_cast<half4*>(hi_plane + off) = hi;
        *reinterpret_cast<half4*>(lo_plane + off) = lo;
      }
    }
  }
};

template <int MT, int NT, int WM, int WN, bool AKM, bool WKM>
__device__ __forceinline__ void grad_tile(const GradP& p, const int tile, const int ksplit, const int z) {
  constexpr int NTHR = 64 * WM * WN;
  constexpr int BM = 32 * MT * WM;
  constexpr int BN = 32 * NT * WN;
  constexpr int PLANE_A = BM * LDH;
  constexpr int PLANE_W = BN * LDH;
  constexpr int STAGE = 2 * PLANE_A + 2 * PLANE_W;
  extern __shared__ __align__(16) _Float16 grad_smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lhi = lane >> 5;

  int tm, tn;
  {
    GemmP g;
    g.group_m = p.group_m; g.tiles_n = p.tiles_n; g.tiles_m = p.tiles_m;
    tile_coords(g, tile, tm, tn);
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const float* A = p.A + z * p.sA;
  const float* W = p.W + z * p.sW;
  float* C = p.C + z * p.sC;
  const int kb = ksplit * p.k_chunk;
  const int ke = min(p.K, kb + p.k_chunk);

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  // Two register sets = prefetch distance of two K-tiles.  These GEMMs run at one or two workgroups per CU
  // (few output tiles, see dispatch()), so nothing but the prefetch depth hides the ~2 us a k-major tile takes
  // to arrive: with one tile in flight the loop measured 2.4 us per K-tile against 0.32 us of matrix work.
  OperandLoader<BM, NTHR, AKM> la0, la1;
  OperandLoader<BN, NTHR, WKM> lw0, lw1;
  constexpr int IT_A = OperandLoader<BM, NTHR, AKM>::IT;
  float4 cs[IT_A];                 // bias-gradient partial sums (k-major A only; always accumulated: 16 adds per K-tile, no branch)
#pragma unroll
  for (int it = 0; it < IT_A; ++it) cs[it] = make_float4(0.f, 0.f, 0.f, 0.f);

  auto compute = [&](int buf, int ks_begin = 0, int ks_end = BK / 16) {
    const _Float16* st = grad_smem + buf * STAGE;
    const _Float16* a_base = st + (wm * 32 * MT + l31) * LDH + lhi * 8;
    const _Float16* w_base = st + 2 * PLANE_A + (wn * 32 * NT + l31) * LDH + lhi * 8;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      if (ks < ks_begin || ks >= ks_end) continue;
      half8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        ah[i] = *reinterpret_cast<const half8*>(a_base + i * 32 * LDH + ks * 16);
        al[i] = *reinterpret_cast<const half8*>(a_base + PLANE_A + i * 32 * LDH + ks * 16);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        bh[j] = *reinterpret_cast<const half8*>(w_base + j * 32 * LDH + ks * 16);
        bl[j] = *reinterpret_cast<const half8*>(w_base + PLANE_W + j * 32 * LDH + ks * 16);
      }
      // term-major order: dependent MFMAs on one accumulator are MT*NT instructions apart
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    }
  };
#define GRAD_LOAD(SET, KT)                                              \
  do {                                                                  \
    la##SET.load(A, p.lda, m0, p.M, kb + (KT) * BK, ke, tid);           \
    lw##SET.load(W, p.ldw, n0, p.N, kb + (KT) * BK, ke, tid);           \
  } while (0)
#define GRAD_LOAD_FULL(SET, KT)                                         \
  do {                                                                  \
    la##SET.load_full(A, p.lda, m0, p.M, kb + (KT) * BK, tid);          \
    lw##SET.load_full(W, p.ldw, n0, p.N, kb + (KT) * BK, tid);          \
  } while (0)
#define GRAD_STORE(SET, BUF)                                                                  \
  do {                                                                                        \
    _Float16* st_ = grad_smem + (BUF) * STAGE;                                                \
    la##SET.template store<AKM>(st_, st_ + PLANE_A, p.a_scale, tid, cs);                       \
    lw##SET.store(st_ + 2 * PLANE_A, st_ + 2 * PLANE_A + PLANE_W, p.w_scale, tid);            \
  } while (0)
#define GRAD_STORE_A(SET, BUF)                                                                \
  do {                                                                                        \
    _Float16* st_ = grad_smem + (BUF) * STAGE;                                                \
    la##SET.template store<AKM>(st_, st_ + PLANE_A, p.a_scale, tid, cs);                       \
  } while (0)
#define GRAD_STORE_W(SET, BUF)                                                                \
  do {                                                                                        \
    _Float16* st_ = grad_smem + (BUF) * STAGE;                                                \
    lw##SET.store(st_ + 2 * PLANE_A, st_ + 2 * PLANE_A + PLANE_W, p.w_scale, tid);            \
  } while (0)

  const int nk = (ke - kb + BK - 1) / BK;
  if (nk > 0) GRAD_LOAD(0, 0);
  if (nk > 1) GRAD_LOAD(1, 1);
  if (nk > 0) GRAD_STORE(0, 0);
  __syncthreads();
  // invariant at the top of iteration kt: LDS stage kt&1 holds tile kt, register set (kt+1)&1 holds tile kt+1
  // (in flight), register set kt&1 is free.  Steady state: two K-tiles per trip, no branch inside (one scheduling
  // region), the split + LDS stores of tile kt+1 placed between the two 16-deep MFMA steps of tile kt.
  int kt = 0;
  const int nk_full = (ke - kb) / BK;          // tiles that lie entirely inside [kb, ke)
  for (; kt + 3 < nk_full; kt += 2) {
    GRAD_LOAD_FULL(0, kt + 2);
    compute(0, 0, 1);
    GRAD_STORE_A(1, 1);
    compute(0, 1, 2);
    GRAD_STORE_W(1, 1);
    __syncthreads();
    GRAD_LOAD_FULL(1, kt + 3);
    compute(1, 0, 1);
    GRAD_STORE_A(0, 0);
    compute(1, 1, 2);
    GRAD_STORE_W(0, 0);
    __syncthreads();
  }
  for (; kt < nk; kt += 2) {
    if (kt + 2 < nk) GRAD_LOAD(0, kt + 2);
    compute(0);
    if (kt + 1 < nk) GRAD_STORE(1, 1);
    __syncthreads();
    if (kt + 1 >= nk) break;
    if (kt + 3 < nk) GRAD_LOAD(1, kt + 3);
    compute(1);
    if (kt + 2 < nk) GRAD_STORE(0, 0);
    __syncthreads();
  }
#undef GRAD_LOAD
#undef GRAD_LOAD_FULL
#undef GRAD_STORE
#undef GRAD_STORE_A
#undef GRAD_STORE_W

  // ---- bias gradient: the tn == 0 tile of every (row panel, K chunk) adds its partial sums ------------
  if constexpr (AKM) {
    if (p.colsum != nullptr && tn == 0) {
#pragma unroll
      for (int it = 0; it < IT_A; ++it) {
        float4 v = cs[it];
#pragma unroll
        for (int sh = 1; sh < 8; sh <<= 1) {          // the 8 k-groups of a row quad sit in 8 neighbouring lanes
          v.x += __shfl_xor(v.x, sh); v.y += __shfl_xor(v.y, sh); v.z += __shfl_xor(v.z, sh); v.w += __shfl_xor(v.w, sh);
        }
        const int u = tid + it * NTHR;
        const int r = m0 + (u >> 3) * 4;
        if ((u & 7) == 0 && u < (BM / 4) * 8 && r < p.M) {      // M % 4 == 0 for k-major A (checked on the host)
          unsafeAtomicAdd(p.colsum + r + 0, v.x);
          unsafeAtomicAdd(p.colsum + r + 1, v.y);
          unsafeAtomicAdd(p.colsum + r + 2, v.z);
          unsafeAtomicAdd(p.colsum + r + 3, v.w);
        }
      }
    }
  }

  // ---- epilogue: plain store or atomic accumulate ------------------------------------------------
  const int row_w = m0 + wm * 32 * MT, col_w = n0 + wn * 32 * NT;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = col_w + j * 32 + l31;
    if (col >= p.N) continue;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = row_w + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
        if (row < p.M) {
          float* dst = C + (int64_t)row * p.ldc + col;
          const float v = acc[i][j][e] * p.alpha;
          if (p.atomic) unsafeAtomicAdd(dst, v);
          else if (p.accumulate) *dst += v;      // this workgroup is the only writer of the tile in this launch
          else *dst = v;
        }
      }
  }
}

template <int MT, int NT, int WM, int WN, bool AKM, bool WKM>
__global__ __launch_bounds__(64 * WM * WN, 2) void gemm_grad_kernel(const GradP p) {
  if (p.kxcd) {
    // chunk-major over the XCDs: the workgroups of one XCD (ids = x mod 8) take consecutive (chunk, tile) pairs, so
    // they walk the SAME K range of the operands and every panel enters that XCD's L2 once; with the tile-major
    // order each XCD reads its tiles' panels over the whole contraction and the panels are fetched by several XCDs
    const int tiles = p.tiles_m * p.tiles_n;
    const int flat = remap_tile(blockIdx.x, gridDim.x);
    const int ksplit = flat / tiles;
    grad_tile<MT, NT, WM, WN, AKM, WKM>(p, flat - ksplit * tiles, ksplit, blockIdx.z);
    return;
  }
  grad_tile<MT, NT, WM, WN, AKM, WKM>(p, remap_tile(blockIdx.x, gridDim.x), blockIdx.y, blockIdx.z);
}

// Several independent problems in ONE launch (the weight gradients of a transformer layer: 6 small outputs, long
// contractions).  Launched one by one each gives ~1 workgroup per CU unless its K range is split 4-15 ways (atomics, a
// pipeline fill per chunk); together they fill the chip with long K loops.  The descriptors travel as kernel arguments.
constexpr int MAX_GROUP = 8;
struct GradGroup {
  GradP p[MAX_GROUP];
  int wg_start[MAX_GROUP + 1];      // first workgroup of problem i (multiples of 8: the XCD round-robin restarts per problem)
  int count;
};

template <int MT, int NT, int WM, int WN, bool AKM, bool WKM>
__global__ __launch_bounds__(64 * WM * WN, 2) void gemm_grad_group_kernel(const GradGroup g) {
  const int b = blockIdx.x;
  int i = 0;
  while (i + 1 < g.count && b >= g.wg_start[i + 1]) ++i;
  const GradP p = g.p[i];
  const int local = b - g.wg_start[i];
  const int tiles = p.tiles_m * p.tiles_n;
  const int nwg = tiles * p.splits;
  if (local >= nwg) return;           // padding up to the next multiple of 8
  // split-major: the `tiles` workgroups of one K chunk are consecutive, remapped so neighbours share an XCD's L2
  const int ksplit = local / tiles;
  const int tile = remap_tile(local - ksplit * tiles, tiles);
  grad_tile<MT, NT, WM, WN, AKM, WKM>(p, tile, ksplit, 0);
}

template <int MT, int NT, int WM, int WN, bool AKM, bool WKM>
int launch_grad(GradP p, int batch, int splits, hipStream_t st) {
  constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
  constexpr size_t smem = (size_t)2 * (2 * BM * LDH + 2 * BN * LDH) * sizeof(_Float16);
  static bool attr_set = false;
  auto kern = gemm_grad_kernel<MT, NT, WM, WN, AKM, WKM>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  p.group_m = p.tiles_n > 1 ? 8 : 0;
  // chunk-major XCD assignment of split-K launches (training iteration 8.52 -> 8.36 ms, one stream 11.59 -> 11.36); 0 = tile-major
  static const bool kxcd = !(getenv("PFPP_GRAD_KXCD") && atoi(getenv("PFPP_GRAD_KXCD")) == 0);
  p.kxcd = kxcd && splits > 1;
  const dim3 grid = p.kxcd ? dim3((unsigned)(p.tiles_m * p.tiles_n * splits), 1u, (unsigned)batch)
                           : dim3((unsigned)(p.tiles_m * p.tiles_n), (unsigned)splits, (unsigned)batch);
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), smem, st, p);
  return pfpp::check_launch("pfpp_gemm_grad");
}

template <bool AKM, bool WKM>
int dispatch(GradP p, int batch, int split_k, hipStream_t st) {
  // tile shape by how many workgroups the output gives: 256x128 when that still fills the chip
  const auto tiles = [&](int bm, int bn) { return (int64_t)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * batch; };
  int bm = 128, bn = 64;
  static const bool big_split = getenv("PFPP_GRAD_BIG") && atoi(getenv("PFPP_GRAD_BIG")) == 1;   // experiment knob
  if (tiles(256, 128) >= 384 || (big_split && p.accumulate && split_k != 1 && p.M >= 256 && p.N >= 128)) { bm = 256; bn = 128; }
  else if (tiles(128, 128) >= 256 || p.N > 64) { bm = 128; bn = 128; }
  if (bm == 128 && bn == 128 && tiles(128, 128) < 192 && split_k == 1 && p.N <= 2048) { bn = 64; }
  static const int small_tiles = getenv("PFPP_GRAD_SMALL") ? atoi(getenv("PFPP_GRAD_SMALL")) : 40;   // [512x512]-sized outputs: 128x64 tiles (45 -> 40.6 us)
  if (bm == 128 && bn == 128 && p.accumulate && tiles(128, 128) < small_tiles) { bn = 64; }
  int splits = split_k;
  if (splits <= 0 && !p.accumulate) splits = 1;     // a plain store cannot be split
  if (splits <= 0) {
    // auto: ~1 workgroup per CU, chunks at least 1024 deep.  Measured on the [3850-token] training step after the K loop
    // became one scheduling region (each workgroup is faster, so fewer, deeper chunks win — less atomic traffic):
    // (384, 512) 8.66 ms, (256, 512) 8.48, (192, 512) 8.29, (256, 1024) 8.28, (320, 1024) 8.33, (256, 2048) 8.93 per iteration;
    // again with the chunk-major XCD assignment: (256, 1024) 8.27, (256, 768) 8.19, (224, 768) 8.28, (192, 896) 8.22,
    // (192, 768) 8.16, (192, 640) 8.15, (160, 768) 8.20, (128, 1024) 8.83
    static const int target_wg = getenv("PFPP_GRAD_WG") ? atoi(getenv("PFPP_GRAD_WG")) : 192;
    static const int min_k = getenv("PFPP_GRAD_MINK") ? atoi(getenv("PFPP_GRAD_MINK")) : 768;
    const int64_t t = tiles(bm, bn);
    int64_t want = (target_wg + t - 1) / t;
    const int64_t max_by_k = (p.K + min_k - 1) / min_k;
    if (want > max_by_k) want = max_by_k;
    splits = (int)(want < 1 ? 1 : want);
  }
  int chunk = (p.K + splits - 1) / splits;
  chunk = (chunk + BK - 1) / BK * BK;
  splits = (p.K + chunk - 1) / chunk;
  p.k_chunk = chunk;
  if (splits > 1 && !p.accumulate) {
    pfpp::set_error("pfpp_gemm_grad: split_k > 1 needs accumulate (zero-initialised output)");
    return PFPP_EINVAL;
  }
  if (bm == 256) return launch_grad<2, 2, 4, 2, AKM, WKM>(p, batch, splits, st);
  if (bn == 128) return launch_grad<2, 2, 2, 2, AKM, WKM>(p, batch, splits, st);
  return launch_grad<2, 1, 2, 2, AKM, WKM>(p, batch, splits, st);
}

}  // namespace

extern "C" int pfpp_gemm_grad(const pfpp_gemm_grad_args* a, pfpp_stream_t stream) {
  PFPP_REQUIRE(a && a->A && a->W && a->C, "null pointer");
  PFPP_REQUIRE(a->M > 0 && a->N > 0 && a->K >= 0, "bad sizes");
  PFPP_REQUIRE(a->M < (1ll << 31) && a->N < (1ll << 31) && a->K < (1ll << 31), "sizes exceed int32");
  PFPP_REQUIRE(a->lda % 4 == 0 && a->ldw % 4 == 0 && pfpp::aligned16(a->A) && pfpp::aligned16(a->W),
               "lda/ldw must be multiples of 4 and A/W 16-byte aligned");
  PFPP_REQUIRE(a->sA % 4 == 0 && a->sW % 4 == 0, "batch strides must keep 16-byte alignment");
  if (a->a_kmajor) PFPP_REQUIRE(a->M % 4 == 0 && a->lda >= a->M, "k-major A: M % 4 != 0 or lda < M");
  else PFPP_REQUIRE(a->lda >= ((a->K + 3) & ~3ll), "lda smaller than K rounded up to 4");
  if (a->w_kmajor) PFPP_REQUIRE(a->N % 4 == 0 && a->ldw >= a->N, "k-major W: N % 4 != 0 or ldw < N");
  else PFPP_REQUIRE(a->ldw >= ((a->K + 3) & ~3ll), "ldw smaller than K rounded up to 4");
  PFPP_REQUIRE(a->ldc >= a->N && a->batch >= 1 && a->split_k >= 0, "bad ldc / batch / split_k");
  PFPP_REQUIRE(a->a_scale > 0.0f && a->w_scale > 0.0f, "operand scales must be positive");
  PFPP_SUPPORTED(a->a_kmajor == 0 || a->w_kmajor != 0, "k-major A with row-major W");
  PFPP_SUPPORTED(!a->colsum || (a->a_kmajor && a->batch == 1), "colsum: weight-gradient form (k-major A), batch 1 only");

  GradP p;
  p.A = a->A; p.W = a->W; p.C = a->C;
  p.M = (int)a->M; p.N = (int)a->N; p.K = (int)a->K;
  p.lda = a->lda; p.ldw = a->ldw; p.ldc = a->ldc;
  p.sA = a->sA; p.sW = a->sW; p.sC = a->sC;
  p.accumulate = a->accumulate;
  p.atomic = a->accumulate;     // several launches (micro-batches, streams) may add into one buffer
  p.splits = 1; p.kxcd = 0; p.colsum = a->colsum;
  p.a_scale = a->a_scale; p.w_scale = a->w_scale;
  p.alpha = a->alpha / (a->a_scale * a->w_scale);
  p.k_chunk = 0;
  hipStream_t st = pfpp::as_stream(stream);
  if (a->a_kmajor) return dispatch<true, true>(p, a->batch, a->split_k, st);
  if (a->w_kmajor) return dispatch<false, true>(p, a->batch, a->split_k, st);
  return dispatch<false, false>(p, a->batch, a->split_k, st);
}

namespace {

int fill_problem(const pfpp_gemm_grad_args* a, GradP& p) {
  PFPP_REQUIRE(a->A && a->W && a->C, "null pointer");
  PFPP_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, "bad sizes");
  PFPP_REQUIRE(a->M < (1ll << 31) && a->N < (1ll << 31) && a->K < (1ll << 31), "sizes exceed int32");
  PFPP_REQUIRE(a->lda % 4 == 0 && a->ldw % 4 == 0 && pfpp::aligned16(a->A) && pfpp::aligned16(a->W),
               "lda/ldw must be multiples of 4 and A/W 16-byte aligned");
  PFPP_SUPPORTED(a->a_kmajor && a->w_kmajor && a->batch == 1, "grouped launch: weight-gradient form only (both operands k-major, batch 1)");
  PFPP_SUPPORTED(!a->colsum, "grouped launch: no fused column sums");
  PFPP_REQUIRE(a->M % 4 == 0 && a->lda >= a->M && a->N % 4 == 0 && a->ldw >= a->N, "k-major operands: M, N % 4 != 0 or ld too small");
  PFPP_REQUIRE(a->ldc >= a->N, "bad ldc");
  PFPP_REQUIRE(a->a_scale > 0.0f && a->w_scale > 0.0f, "operand scales must be positive");
  p.A = a->A; p.W = a->W; p.C = a->C;
  p.M = (int)a->M; p.N = (int)a->N; p.K = (int)a->K;
  p.lda = a->lda; p.ldw = a->ldw; p.ldc = a->ldc;
  p.sA = p.sW = p.sC = 0; p.kxcd = 0; p.colsum = nullptr;
  p.accumulate = a->accumulate;
  p.a_scale = a->a_scale; p.w_scale = a->w_scale;
  p.alpha = a->alpha / (a->a_scale * a->w_scale);
  return PFPP_OK;
}

}  // namespace

extern "C" int pfpp_gemm_grad_group(const pfpp_gemm_grad_args* args, int count, pfpp_stream_t stream) {
  PFPP_REQUIRE(args && count >= 1 && count <= MAX_GROUP, "1..8 problems per grouped launch");
  constexpr int BM = 128, BN = 128;
  GradGroup g;
  int64_t tiles_total = 0;
  for (int i = 0; i < count; ++i) {
    const int rc = fill_problem(args + i, g.p[i]);
    if (rc != PFPP_OK) return rc;
    GradP& p = g.p[i];
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    p.group_m = p.tiles_n > 1 ? 8 : 0;
    tiles_total += (int64_t)p.tiles_m * p.tiles_n;
  }
  // one K split factor for the whole group: enough workgroups for ~2 per CU, chunks at least `min_k` deep
  static const int target_wg = getenv("PFPP_GRAD_GROUP_WG") ? atoi(getenv("PFPP_GRAD_GROUP_WG")) : 1024;
  static const int min_k = getenv("PFPP_GRAD_GROUP_MINK") ? atoi(getenv("PFPP_GRAD_GROUP_MINK")) : 640;
  int start = 0;
  for (int i = 0; i < count; ++i) {
    GradP& p = g.p[i];
    int64_t want = (target_wg + tiles_total / 2) / tiles_total;
    const int64_t max_by_k = (p.K + min_k - 1) / min_k;
    if (want > max_by_k) want = max_by_k;
    if (want < 1 || !p.accumulate) want = 1;              // a plain store cannot be split
    if (args[i].split_k > 0) want = args[i].split_k;
    int chunk = (int)((p.K + want - 1) / want);
    chunk = (chunk + BK - 1) / BK * BK;
    p.splits = (p.K + chunk - 1) / chunk;
    p.k_chunk = chunk;
    PFPP_REQUIRE(p.splits == 1 || p.accumulate, "split_k > 1 needs accumulate (zero-initialised output)");
    p.atomic = p.splits > 1;
    g.wg_start[i] = start;
    start += (p.tiles_m * p.tiles_n * p.splits + 7) & ~7;
  }
  for (int i = count; i <= MAX_GROUP; ++i) g.wg_start[i] = start;
  g.count = count;
  constexpr size_t smem = (size_t)2 * (2 * BM * LDH + 2 * BN * LDH) * sizeof(_Float16);
  auto kern = gemm_grad_group_kernel<2, 2, 2, 2, true, true>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)start), dim3(256), smem, pfpp::as_stream(stream), g);
  return pfpp::check_launch("pfpp_gemm_grad_group");
}
